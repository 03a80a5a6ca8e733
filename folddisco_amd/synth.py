"""Seeded synthetic "AFDB-shaped" structures: benchmark / test input only.

Vectorised over structures with torch, so the same code runs on the CPU (small parity cases) and on
the GPU (bench-scale shards generated directly in HBM).  Chains are CA traces with 3.8 Å virtual
bonds whose bond/dihedral angles follow helix / strand / coil segments, confined to a sphere of
protein-like density so that the number of residue pairs within 20 Å per residue lands in the
50–120 range measured on the reference's fixtures; N and C are placed from the trace, CB by the
reference's ideal-tetrahedral construction (src/structure/coordinate.rs:167-186) plus noise;
coordinates are rounded to 3 decimals (PDB precision); residue types are iid from Swiss-Prot
background frequencies.  Inputs to the GPU path and to the CPU oracle are always the *same* arrays
(copied), so CPU/GPU libm differences inside this generator do not matter.
"""
from __future__ import annotations

import math
import os

import numpy as np
import torch

# Swiss-Prot amino-acid background (A R N D C Q E G H I L K M F P S T W Y V), release-like values
_AA_FREQ = np.array([8.25, 5.53, 4.06, 5.45, 1.37, 3.93, 6.75, 7.07, 2.27, 5.96, 9.66, 5.84, 2.42, 3.86, 4.70, 6.56,
                     5.34, 1.08, 2.92, 6.87], dtype=np.float64)
_AA_FREQ /= _AA_FREQ.sum()


def sample_lengths(n: int, seed: int, median: float = 270.0, sigma: float = 0.6, lo: int = 40, hi: int = 2700) -> np.ndarray:
    rng = np.random.Generator(np.random.PCG64(seed))
    L = np.exp(rng.normal(math.log(median), sigma, size=n))
    return np.clip(np.rint(L), lo, hi).astype(np.int64)


def _unit(v: torch.Tensor) -> torch.Tensor:
    return v / v.norm(dim=-1, keepdim=True).clamp_min(1e-6)


def _place(prev2, prev1, cur, theta, tau, bond=3.8):
    """next point from the last three (internal coordinates: bond, bond angle theta, dihedral tau)."""
    bc = _unit(cur - prev1)
    n = _unit(torch.cross(prev1 - prev2, bc, dim=-1))
    m = torch.cross(n, bc, dim=-1)
    d = torch.stack([-bond * torch.cos(theta), bond * torch.sin(theta) * torch.cos(tau), bond * torch.sin(theta) * torch.sin(tau)], dim=-1)
    return cur + d[..., 0:1] * bc + d[..., 1:2] * m + d[..., 2:3] * n


@torch.no_grad()
def generate(n_struct: int, seed: int = 20260927, device: str | torch.device = "cpu", lengths: np.ndarray | None = None,
             k_candidates: int = 6):
    """Returns dict of torch tensors on `device`:
    res_off u64[S+1] (as int64), n_xyz/ca_xyz/cb_xyz f32[R,3], aa u8[R], plddt f32[R]."""
    dev = torch.device(device)
    g = torch.Generator(device=dev)
    g.manual_seed(seed)
    if lengths is None:
        lengths = sample_lengths(n_struct, seed)
    lengths = np.asarray(lengths, dtype=np.int64)
    S = len(lengths)
    order = np.argsort(-lengths, kind="stable")          # process longest first: active set is a prefix
    Ls = torch.from_numpy(lengths[order]).to(dev)
    Lmax = int(lengths.max()) if S else 0
    radius = 3.5 * Ls.to(torch.float32).pow(1.0 / 3.0) + 2.0
    ca = torch.zeros((S, max(Lmax, 3), 3), dtype=torch.float32, device=dev)
    # first three points: a bent start near the centre
    ca[:, 1, 0] = 3.8
    ca[:, 2, 0] = 3.8 + 3.8 * math.cos(math.radians(70.0))
    ca[:, 2, 1] = 3.8 * math.sin(math.radians(70.0))
    ca[:, :3] -= ca[:, :3].mean(dim=1, keepdim=True)
    sse_type = torch.randint(0, 3, (S,), generator=g, device=dev)       # 0 helix, 1 strand, 2 coil
    sse_left = torch.randint(4, 12, (S,), generator=g, device=dev)
    deg = math.pi / 180.0
    for t in range(3, Lmax):
        na = int((Ls > t).sum().item())
        if na == 0:
            break
        st, sl = sse_type[:na], sse_left[:na]
        new_seg = sl <= 0
        if bool(new_seg.any()):
            nt = torch.randint(0, 3, (na,), generator=g, device=dev)
            nl = torch.where(nt == 0, torch.randint(8, 22, (na,), generator=g, device=dev),
                             torch.where(nt == 1, torch.randint(4, 10, (na,), generator=g, device=dev),
                                         torch.randint(2, 9, (na,), generator=g, device=dev)))
            st = torch.where(new_seg, nt, st)
            sl = torch.where(new_seg, nl, sl)
        # candidate (theta, tau): candidate 0 follows the segment type, the rest are coil-like
        K = k_candidates
        u1 = torch.rand((na, K), generator=g, device=dev)
        u2 = torch.rand((na, K), generator=g, device=dev)
        theta = (90.0 + 50.0 * u1) * deg
        tau = (-180.0 + 360.0 * u2) * deg
        jit = (torch.rand((na, 2), generator=g, device=dev) - 0.5) * (12.0 * deg)
        th0 = torch.where(st == 0, torch.full_like(jit[:, 0], 91.0 * deg), torch.where(st == 1, torch.full_like(jit[:, 0], 124.0 * deg), theta[:, 0]))
        ta0 = torch.where(st == 0, torch.full_like(jit[:, 1], 50.0 * deg), torch.where(st == 1, torch.full_like(jit[:, 1], -170.0 * deg), tau[:, 0]))
        theta[:, 0] = th0 + jit[:, 0]
        tau[:, 0] = ta0 + jit[:, 1]
        p2, p1, p0 = ca[:na, t - 3], ca[:na, t - 2], ca[:na, t - 1]
        cand = _place(p2[:, None, :], p1[:, None, :], p0[:, None, :], theta, tau)   # [na, K, 3]
        rad = cand.norm(dim=-1)
        inside = rad <= radius[:na, None]
        # first candidate that stays inside; otherwise the one closest to the centre
        first_in = torch.where(inside, torch.arange(K, device=dev)[None, :], torch.full((1, 1), K, device=dev)).min(dim=1).values
        pick = torch.where(first_in < K, first_in, rad.argmin(dim=1))
        ca[:na, t] = cand[torch.arange(na, device=dev), pick]
        sse_type[:na] = st
        sse_left[:na] = sl - 1
    # backbone N / C from the trace, CB ideal + noise
    idx = torch.arange(max(Lmax, 3), device=dev)[None, :]
    mask = idx < Ls[:, None]
    prev = torch.roll(ca, 1, dims=1)
    nxt = torch.roll(ca, -1, dims=1)
    prev[:, 0] = 2 * ca[:, 0] - ca[:, 1]
    last = (Ls - 1).clamp_min(1)
    ar = torch.arange(S, device=dev)
    nxt[ar, last] = 2 * ca[ar, last] - ca[ar, last - 1]
    up, un = _unit(prev - ca), _unit(nxt - ca)
    w = _unit(torch.cross(up, un, dim=-1) + 1e-3)
    n_at = ca + 1.46 * _unit(0.80 * up + 0.35 * w - 0.10 * un)
    c_at = ca + 1.52 * _unit(0.80 * un - 0.35 * w - 0.10 * up)
    v1, v2 = _unit(c_at - ca), _unit(n_at - ca)
    b1 = v2 + v1 / 3.0
    b2 = torch.cross(v1, b1, dim=-1)
    u1_, u2_ = _unit(b1), _unit(b2)
    v4 = (-0.5 * u1_ - (math.sqrt(3.0) / 2.0) * u2_) * (math.sqrt(8.0) / 3.0) - v1 / 3.0
    cb = ca + 1.5336 * v4 + (torch.rand(ca.shape, generator=g, device=dev) - 0.5) * 0.2
    # undo the length sort, flatten structure-major
    inv = torch.from_numpy(np.argsort(order, kind="stable")).to(dev)
    mask_o = mask[inv]
    def pack(x):
        return (torch.round(x[inv][mask_o] * 1000.0) / 1000.0).contiguous()
    n_flat, ca_flat, cb_flat = pack(n_at), pack(ca), pack(cb)
    R = int(lengths.sum())
    probs = torch.from_numpy(_AA_FREQ).to(dev, dtype=torch.float32)
    aa = torch.multinomial(probs, max(R, 1), replacement=True, generator=g)[:R].to(torch.uint8)
    plddt = 50.0 + 45.0 * torch.rand((R,), generator=g, device=dev)
    res_off = torch.zeros(S + 1, dtype=torch.int64, device=dev)
    res_off[1:] = torch.cumsum(torch.from_numpy(lengths).to(dev), dim=0)
    return dict(res_off=res_off, n_xyz=n_flat, ca_xyz=ca_flat, cb_xyz=cb_flat, aa=aa.contiguous(), plddt=plddt)


def to_packed(d) -> "PackedStructures":
    from .api import PackedStructures
    return PackedStructures(d["res_off"].cpu().numpy().astype(np.uint64), d["n_xyz"].cpu().numpy(), d["ca_xyz"].cpu().numpy(),
                            d["cb_xyz"].cpu().numpy(), d["aa"].cpu().numpy())


# ---------------------------------------------------------------------------------------------------------------- files for the ingest leg
_AA3 = ["ALA", "ARG", "ASN", "ASP", "CYS", "GLN", "GLU", "GLY", "HIS", "ILE", "LEU", "LYS", "MET", "PHE", "PRO", "SER", "THR", "TRP", "TYR", "VAL"]


def _pdb_text(n, ca, cb, aa, plddt) -> bytes:
    """one structure as PDB ATOM records (N, CA, C, O, CB per residue; C and O are fabricated from the trace — the ingest reads them like
    any backbone atom, the hot path uses N, CA and the explicit CB)"""
    L = len(aa)
    nxt = np.roll(ca, -1, axis=0)
    nxt[-1] = 2 * ca[-1] - ca[-2] if L > 1 else ca[-1] + 1.0
    u = nxt - ca
    u /= np.maximum(np.linalg.norm(u, axis=1, keepdims=True), 1e-6)
    c = ca + 1.52 * u
    o = c + np.array([0.0, 1.23, 0.0], np.float32)
    out = []
    k = 1
    for r in range(L):
        rn = _AA3[int(aa[r])] if aa[r] < 20 else "UNK"
        for nm, xyz, el in ((" N  ", n[r], "N"), (" CA ", ca[r], "C"), (" C  ", c[r], "C"), (" O  ", o[r], "O"), (" CB ", cb[r], "C")):
            out.append("ATOM  %5d %s %s A%4d    %8.3f%8.3f%8.3f  1.00%6.2f          %2s" % (k % 100000, nm, rn, (r + 1) % 10000, xyz[0], xyz[1], xyz[2], plddt[r], el))
            k += 1
    out.append("END")
    return ("\n".join(out) + "\n").encode()


def _write_pdb_range(args):
    import gzip
    directory, first, items = args
    for k, (n, ca, cb, aa, pl) in enumerate(items):
        with gzip.open("%s/AF-S%07d-F1-model_v4.pdb.gz" % (directory, first + k), "wb", compresslevel=1) as f:
            f.write(_pdb_text(n, ca, cb, aa, pl))
    return len(items)


def write_pdb_gz(d, directory: str, workers: int = 32) -> int:
    """the structures of generate()'s dict as gzipped PDB files, written by worker processes -> number of files"""
    import os
    from concurrent.futures import ProcessPoolExecutor
    os.makedirs(directory, exist_ok=True)
    off = d["res_off"].cpu().numpy()
    h = {k: d[k].cpu().numpy() for k in ("n_xyz", "ca_xyz", "cb_xyz", "aa", "plddt")}
    S = len(off) - 1
    per = max(1, -(-S // (workers * 4)))
    jobs = []
    for a in range(0, S, per):
        b = min(S, a + per)
        jobs.append((directory, a, [(h["n_xyz"][off[s]:off[s + 1]], h["ca_xyz"][off[s]:off[s + 1]], h["cb_xyz"][off[s]:off[s + 1]], h["aa"][off[s]:off[s + 1]],
                                     h["plddt"][off[s]:off[s + 1]]) for s in range(a, b)]))
    with ProcessPoolExecutor(workers) as ex:
        return sum(ex.map(_write_pdb_range, jobs))


def replicate_foldcomp_db(src_db: str, dst_db: str, n_entries: int) -> int:
    """a Foldcomp database of n_entries built by repeating the entries of src_db (DB, DB.index, DB.lookup, DB.dbtype): real compressed
    proteins to decode, as many as the ingest leg wants -> number of entries"""
    import shutil
    data = open(src_db, "rb").read()
    ents = [tuple(int(t) for t in line.split()) for line in open(src_db + ".index")]
    names = {}
    for line in open(src_db + ".lookup"):
        p = line.rstrip("\n").split("\t")
        names[int(p[0])] = p[1]
    ents.sort()
    with open(dst_db, "wb") as f, open(dst_db + ".index", "w") as fi, open(dst_db + ".lookup", "w") as fl:
        pos = 0
        for k in range(n_entries):
            key, start, length = ents[k % len(ents)]
            f.write(data[start:start + length])
            fi.write("%d\t%d\t%d\n" % (k, pos, length))
            fl.write("%d\t%s_%06d\t0\n" % (k, names.get(key, "entry"), k))
            pos += length
    if os.path.exists(src_db + ".dbtype"):
        shutil.copy(src_db + ".dbtype", dst_db + ".dbtype")
    return n_entries
