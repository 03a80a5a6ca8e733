"""`folddisco analyze -i PREFIX` — summary of the encoding distribution of an index.

Restates src/cli/workflows/analyze.rs:42-152 (summary branch) over src/controller/summary.rs:121-260, 490-541 and
HashType::{dist_bins, angle_bins, total_bins} (src/controller/feature.rs:293-345): host-side bookkeeping over the sparse
offset table (hashes[H], offsets[H+1]) — the "count" of an encoding is its posting list's BYTE length
(summary.rs:495-496), not the number of structures.  Writes PREFIX_summary_{stats.tsv, topN.tsv, aa_pairs.csv,
count_distribution.tsv}.

`folddisco analyze -i PREFIX -p DIR` — the enrichment branch (analyze_enrichment, summary.rs:262-480): the encodings of a structure
set tested against the index as background (right-tail hypergeometric test, fdgpu_hypergeom_enrichment), the enriched encodings
with the residue pairs that carry them, the positions supported by more than --min-support enriched encodings and a query string
per structure.  The GPU does the per-structure hashing (fdgpu_hash_batch: collect_hash_vec) and the (hash, i, j) stream of every
residue pair (fdgpu_hash_batch_rows: collect_hash_id_pos); the rest is bookkeeping.  Where the reference's output order is that
of a concurrent hash map (rows with equal sort keys in *_enriched_positions.tsv, the order of positions inside a row of
*_enriched_hashes.tsv under several threads) the order here is the single-thread one: structure order, then pair order.
"""
from __future__ import annotations

import numpy as np

from . import indexio
from ._lib import hash_type_index

AA3 = ["ALA", "ARG", "ASN", "ASP", "CYS", "GLN", "GLU", "GLY", "HIS", "ILE", "LEU", "LYS", "MET", "PHE", "PRO", "SER", "THR", "TRP", "TYR",
       "VAL"]                                                   # map_u8_to_aa, src/utils/convert.rs:169-193
_F = np.float32
_PI = _F(3.14159274)
_DEG = _F(57.2957795130823208767981548141051703)                # f32::to_degrees

# per encoding: (default dist bins, default angle bins, #dist fields, #angle fields, angle fields are sin/cos pairs)
_BINS = {0: (18, 9, 2, 1, False), 1: (8, 3, 2, 1, True), 3: (16, 4, 2, 3, True), 7: (8, 32, 2, 3, False), 8: (32, 16, 2, 3, False),
         4: (8, 3, 1, 3, True)}
# TrRosetta / TertiaryInteraction / Hybrid decode to 8-9 values; the reference's summary writes them into a 7-slot container
# (summary.rs:149, 209) and panics, so there is nothing to reproduce for them


_libm = None


def _atan2f(y, x) -> np.ndarray:
    """f32::atan2 = glibc atan2f on linux-gnu (numpy's float32 arctan2 differs from it in the last ulp, which shows in {:.4})"""
    global _libm
    if _libm is None:
        import ctypes as C
        import ctypes.util
        _libm = C.CDLL(ctypes.util.find_library("m") or "libm.so.6")
        _libm.atan2f.restype = C.c_float
        _libm.atan2f.argtypes = [C.c_float, C.c_float]
    y = np.asarray(y, np.float32); x = np.asarray(x, np.float32)
    return np.array([_libm.atan2f(float(a), float(b)) for a, b in zip(y.ravel(), x.ravel())], np.float32).reshape(y.shape)


def _aa(a: int) -> str:
    return AA3[a] if 0 <= a < 20 else "UNK"


def _cont(v, mn, mx, nb):
    """continuize_u32_value_into_f32 (src/utils/convert.rs:43-46), f32 arithmetic"""
    cont = (_F(mx) - _F(mn)) / (_F(nb) - _F(1.0))
    return np.asarray(v, dtype=np.float32) * cont + _F(mn)


def total_bins(htype: int, nbin_dist: int, nbin_angle: int) -> int:
    """HashType::total_bins (src/controller/feature.rs:330-345)"""
    dd, da, nd, na, sincos = _BINS[htype]
    nbd = dd if nbin_dist == 0 else nbin_dist
    nba = da if nbin_angle == 0 else nbin_angle
    dist_bins = nbd ** nd
    if htype == 8 and nba == 16:
        angle_bins = 2048                                       # hard-coded 8 * 16 * 16 (feature.rs:310-316)
    elif sincos:
        angle_bins = (nba * nba) ** na
    else:
        angle_bins = nba ** na
    return dist_bins * angle_bins * 400


def reverse_hash(htype: int, h, nbin_dist: int, nbin_angle: int, angles: bool = True) -> np.ndarray:
    """GeometricHash::reverse_hash (src/geometry/core.rs:425-...) for the built encodings -> f32[n, 7]
    (aa1, aa2, then the encoding's decoded fields; unused slots stay 0 like the reference's feature container).
    angles=False skips the atan2 fields of the sin/cos encodings (one libm call per value: only the top-N rows need them)."""
    h = np.atleast_1d(np.asarray(h, dtype=np.uint32))
    out = np.zeros((len(h), 7), np.float32)
    u = h.astype(np.uint32)
    if htype == 3:                                              # pdb_tr.rs:95-136
        out[:, 0] = (u >> 25) & 31; out[:, 1] = (u >> 20) & 31
        out[:, 2] = _cont((u >> 16) & 15, 2.0, 20.0, nbin_dist); out[:, 3] = _cont((u >> 12) & 15, 2.0, 20.0, nbin_dist)
        for k, (ss, cs) in enumerate(((10, 8), (6, 4), (2, 0))):
            if not angles:
                break
            s = _cont((u >> ss) & 3, -1.0, 1.0, nbin_angle); c = _cont((u >> cs) & 3, -1.0, 1.0, nbin_angle)
            out[:, 4 + k] = _atan2f(s, c) * _DEG
    elif htype == 0:                                            # pdb_motif.rs:56-72
        out[:, 0] = (u >> 20) & 31; out[:, 1] = (u >> 15) & 31
        out[:, 2] = _cont((u >> 10) & 31, 2.0, 20.0, nbin_dist); out[:, 3] = _cont((u >> 5) & 31, 2.0, 20.0, nbin_dist)
        out[:, 4] = _cont(u & 31, 0.0, 180.0, nbin_angle)
    elif htype == 1:                                            # pdb_motif_sincos.rs:59-81
        out[:, 0] = (u >> 21) & 31; out[:, 1] = (u >> 16) & 31
        out[:, 2] = _cont((u >> 12) & 15, 2.0, 20.0, nbin_dist); out[:, 3] = _cont((u >> 8) & 15, 2.0, 20.0, nbin_dist)
        if angles:
            s = _cont((u >> 4) & 15, -1.0, 1.0, nbin_angle); c = _cont(u & 15, -1.0, 1.0, nbin_angle)
            out[:, 4] = _atan2f(s, c) * _DEG
    elif htype == 4:                                            # ppf.rs:55-88
        out[:, 0] = (u >> 27) & 31; out[:, 1] = (u >> 22) & 31
        out[:, 2] = _cont((u >> 18) & 15, 2.0, 20.0, nbin_dist)
        if angles:
            for k in range(3):
                sn = _cont((u >> (15 - 6 * k)) & 7, -1.0, 1.0, nbin_angle); cs = _cont((u >> (12 - 6 * k)) & 7, -1.0, 1.0, nbin_angle)
                out[:, 3 + k] = _atan2f(sn, cs) * _DEG
    elif htype in (7, 8):                                       # folddisco_angle.rs:80-108, folddisco_dist.rs:73-102
        pair = (u >> 21) & 0x1ff
        out[:, 0] = pair // 20; out[:, 1] = pair % 20
        if htype == 7:
            ca, cb, th, p1, p2, cap = (u >> 18) & 7, (u >> 15) & 7, (u >> 10) & 31, (u >> 5) & 31, u & 31, 32.0
        else:
            ca, cb, th, p1, p2, cap = (u >> 16) & 31, (u >> 11) & 31, (u >> 8) & 7, (u >> 4) & 15, u & 15, 8.0
        out[:, 2] = _cont(ca, 2.0, 20.0, nbin_dist); out[:, 3] = _cont(cb, 2.0, 20.0, nbin_dist)
        out[:, 4] = _cont(th, 0.0, _PI, min(float(nbin_angle), cap)) * _DEG
        out[:, 5] = _cont(p1, -_PI, _PI, nbin_angle) * _DEG
        out[:, 6] = _cont(p2, -_PI, _PI, nbin_angle) * _DEG
    else:
        raise ValueError(f"analyze: hash type {htype} is not built")
    return out


def count_distribution(counts_desc: np.ndarray):
    """get_counts_from_hash_count_vec with default borders (summary.rs:502-541): powers of two up to the largest count, then the
    largest count itself; row = (border, number of encodings with previous border < count <= border)"""
    mx = int(counts_desc[0])
    borders, b = [], 1
    while b <= mx:
        borders.append(b); b *= 2
    if borders[-1] < mx:
        borders.append(mx)
    asc = counts_desc[::-1]
    n = len(asc)
    gt = lambda x: n - int(np.searchsorted(asc, x, side="right"))     # encodings with count > x
    rows = []
    for k, bd in enumerate(borders):
        rows.append((bd, n - gt(bd) if k == 0 else gt(borders[k - 1]) - gt(bd)))
    return rows


def summarize(prefix: str) -> dict:
    """count_encodings (summary.rs:125-171)"""
    cfg = indexio.load_type(prefix + ".type")
    htype = hash_type_index(cfg.get("hash_type", "PDBTrRosetta"))
    if htype not in _BINS:
        raise ValueError(f"analyze: hash type {cfg.get('hash_type')} is not built")
    nbd, nba = int(cfg.get("num_bin_dist", 0)), int(cfg.get("num_bin_angle", 0))
    nbd = _BINS[htype][0] if nbd == 0 else nbd
    nba = _BINS[htype][1] if nba == 0 else nba
    _, hashes, offsets = indexio.read_index_files(prefix)
    counts = np.diff(offsets.astype(np.int64)).astype(np.int64)
    order = np.argsort(-counts, kind="stable")                 # par_sort_by(|a, b| b.1.cmp(&a.1)): stable, descending by count
    h_sorted, c_sorted = hashes[order], counts[order]
    total = len(hashes)
    possible = total_bins(htype, nbd, nba)
    feats = reverse_hash(htype, hashes, nbd, nba, angles=False)
    a1, a2 = feats[:, 0].astype(np.int64), feats[:, 1].astype(np.int64)
    ok = (a1 < 20) & (a2 < 20)
    aa_pairs = np.bincount(a1[ok] * 20 + a2[ok], weights=counts[ok].astype(np.float64), minlength=400).astype(np.int64).reshape(20, 20)
    return dict(hash_type=htype, nbin_dist=nbd, nbin_angle=nba, total=total, possible=possible, empty=possible - total, nonempty=total,
                density=np.float32(np.float64(total) / np.float64(possible) * 100.0), hashes=h_sorted, counts=c_sorted, aa_pairs=aa_pairs)


def save_summary(st: dict, output_prefix: str, top_n: int = 10):
    """save_summary (summary.rs:173-260)"""
    with open(f"{output_prefix}_stats.tsv", "w") as f:
        f.write("metric\tvalue\n")
        f.write(f"total\t{st['total']}\npossible\t{st['possible']}\nempty\t{st['empty']}\nnonempty\t{st['nonempty']}\n")
        f.write("density\t%.4f\n" % float(st["density"]))
    with open(f"{output_prefix}_top{top_n}.tsv", "w") as f:
        f.write("rank\thash\tcount\taa1\taa2\tca_dist\tcb_dist\tca_cb_angle\tphi1\tphi2\n")
        k = min(top_n, len(st["hashes"]))
        feats = reverse_hash(st["hash_type"], st["hashes"][:k], st["nbin_dist"], st["nbin_angle"])
        for r in range(k):
            v = feats[r]
            f.write("%d\t%d\t%d\t%s\t%s\t%.4f\t%.4f\t%.4f\t%.4f\t%.4f\n" % (r + 1, int(st["hashes"][r]), int(st["counts"][r]), _aa(int(v[0]) & 255),
                                                                          _aa(int(v[1]) & 255), v[2], v[3], v[4], v[5], v[6]))
    with open(f"{output_prefix}_aa_pairs.csv", "w") as f:
        f.write("aa1_aa2" + "".join("," + a for a in AA3) + "\n")
        for i, a in enumerate(AA3):
            f.write(a + "".join(",%d" % int(x) for x in st["aa_pairs"][i]) + "\n")
    with open(f"{output_prefix}_count_distribution.tsv", "w") as f:
        f.write("frequency\tcount\n")
        if len(st["counts"]):
            for bd, c in count_distribution(st["counts"]):
                f.write(f"{bd}\t{c}\n")


def enrichment(ctx, index_prefix: str, paths, p_value: float = 1e-4, threads: int = 1):
    """analyze_enrichment up to the three tables (summary.rs:262-345).  -> dict(enriched=[(hash, p)], positions={hash: [(pdb_pos, pos1,
    pos2)]}, hash_type, nbin_dist, nbin_angle)"""
    import ctypes as C
    from . import structure
    from ._lib import HashParams, u32p, u64p
    from .api import PackedStructures, get_geometric_hash_as_u32
    cfg = indexio.load_type(index_prefix + ".type")
    htype = hash_type_index(cfg.get("hash_type", "PDBTrRosetta"))
    nbd, nba = int(cfg.get("num_bin_dist", 0)), int(cfg.get("num_bin_angle", 0))
    _, bg_hashes, offsets = indexio.read_index_files(index_prefix)
    bg_counts = np.diff(offsets.astype(np.int64)).astype(np.uint64)          # get_hash_count_vec: BYTE lengths (summary.rs:495-496)
    structs, ok = structure.read_compact_structures(list(paths), threads=threads)
    batch = ctx.upload(PackedStructures.concat([s.as_item() for s in structs]))
    # collect_hash_vec: sorted-unique hashes per structure -> number of structures that hold each encoding
    h, off = get_geometric_hash_as_u32(ctx, batch, nbin_dist=nbd, nbin_angle=nba, hash_type=htype)
    q_hashes, q_counts = np.unique(h, return_counts=True)
    total_query, total_bg = int(q_counts.sum()), int(bg_counts.sum())
    k = np.searchsorted(bg_hashes, q_hashes)
    kk = np.minimum(k, max(len(bg_hashes) - 1, 0))
    bg = np.where((k < len(bg_hashes)) & (bg_hashes[kk] == q_hashes), bg_counts[kk], 0).astype(np.uint64) if len(bg_hashes) else np.zeros(len(q_hashes), np.uint64)
    pv = np.zeros(len(q_hashes), np.float64)
    qc = np.ascontiguousarray(q_counts, np.uint64)
    bg = np.ascontiguousarray(bg, np.uint64)
    rc = ctx.L.fdgpu_hypergeom_enrichment(qc.ctypes.data_as(u64p), bg.ctypes.data_as(u64p), len(qc), total_query, total_bg, max(threads, 1),
                                          pv.ctypes.data_as(C.POINTER(C.c_double)))
    if rc:
        raise RuntimeError(f"fdgpu_hypergeom_enrichment failed ({rc})")
    sel = np.nonzero(pv < p_value)[0]
    sel = sel[np.argsort(pv[sel], kind="stable")]                            # par_sort_by p-value ascending (stable: ties stay in hash order)
    enriched = [(int(q_hashes[i]), float(pv[i])) for i in sel]
    # collect_hash_id_pos: (hash, i, j) of every residue pair with a feature; kept for the enriched encodings only
    p = HashParams(nbd, nba, 20.0, htype, None)
    hp, pj, ro = u32p(), u32p(), u64p()
    ctx.check(ctx.L.fdgpu_hash_batch_rows(ctx.h, batch.h, C.byref(p), C.byref(hp), C.byref(pj), C.byref(ro)))
    R = sum(s.n for s in structs)
    row_off = np.ctypeslib.as_array(ro, shape=(R + 1,)).copy()
    P = int(row_off[-1])
    raw = np.ctypeslib.as_array(hp, shape=(max(P, 1),))[:P].copy()
    part = np.ctypeslib.as_array(pj, shape=(max(P, 1),))[:P].copy()
    for ptr in (hp, pj, ro):
        ctx.L.fdgpu_free(ptr)
    res_struct = np.repeat(np.arange(len(structs)), [s.n for s in structs])
    labels = np.array([f"{chr(int(c))}{int(r)}" for s in structs for c, r in zip(s.chain, s.serial)], dtype=object)
    keep = np.nonzero(np.isin(raw, q_hashes[sel]))[0] if len(sel) else np.zeros(0, np.int64)
    res_i = np.searchsorted(row_off, keep, side="right") - 1                # the row (residue i) an entry belongs to
    positions = {}
    for e, i in zip(keep, res_i):
        positions.setdefault(int(raw[e]), []).append((int(res_struct[i]), labels[i], labels[int(part[e])]))
    return dict(enriched=enriched, positions=positions, hash_type=htype, nbin_dist=_BINS[htype][0] if nbd == 0 else nbd,
                nbin_angle=_BINS[htype][1] if nba == 0 else nba)


def save_enrichment(en: dict, paths, output_prefix: str, min_support: int = 4, max_pos: int = 32):
    """the three tables of analyze_enrichment (summary.rs:347-478)"""
    paths = list(paths)
    pos_hash, pdb_positions = {}, {}
    with open(f"{output_prefix}_enriched_hashes.tsv", "w") as f:
        f.write("hash\tp_value\tfeatures\tpositions\n")
        for h, pv in en["enriched"]:
            v = reverse_hash(en["hash_type"], [h], en["nbin_dist"], en["nbin_angle"])[0]
            feat = "%s,%s,%.4f,%.4f,%.4f,%.4f,%.4f" % (_aa(int(v[0]) & 255), _aa(int(v[1]) & 255), v[2], v[3], v[4], v[5], v[6])
            plist = []
            for pdb_pos, p1, p2 in en["positions"].get(h, []):
                plist.append(f"{paths[pdb_pos]}-{p1}-{p2}")
                pos_hash.setdefault((pdb_pos, p1), []).append(h)
                pos_hash.setdefault((pdb_pos, p2), []).append(h)
                pdb_positions.setdefault(pdb_pos, []).extend([p1, p2])
            f.write("%d\t%s\t%s\t%s\n" % (h, _rust_exp4(pv), feat, ",".join(plist)))
    pos_list = sorted(pos_hash.items(), key=lambda kv: (kv[0][0], -len(kv[1])))     # stable: pdb id, then count descending
    with open(f"{output_prefix}_enriched_positions.tsv", "w") as f:
        f.write("id\tpos\tcount\thash_list\n")
        for (pdb_pos, pos), hl in pos_list:
            if len(hl) <= min_support:
                continue
            f.write("%s\t%s\t%d\t%s\n" % (paths[pdb_pos], pos, len(hl), ",".join(str(x) for x in hl)))
    with open(f"{output_prefix}_query_summary.tsv", "w") as f:
        f.write("pdb_path\tpositions\n")
        for pdb_pos in sorted(pdb_positions):
            uniq = sorted(set(pdb_positions[pdb_pos]))
            withc = [(pos, len(pos_hash[(pdb_pos, pos)])) for pos in uniq if len(pos_hash.get((pdb_pos, pos), [])) > min_support]
            withc.sort(key=lambda t: -t[1])                                 # stable, count descending
            withc = withc[:max_pos]

            def key(pos):
                try:
                    idx = int(pos[1:])
                except ValueError:
                    idx = 0
                return (pos[:1] or " ", idx)
            final = sorted((pos for pos, _ in withc), key=key)
            if final:
                f.write("%s\t%s\n" % (paths[pdb_pos], ",".join(final)))


def _rust_exp4(x: float) -> str:
    """Rust `{:.4e}`: mantissa with four decimals, exponent without padding or plus sign (1.2346e-7, 0.0000e0)"""
    if x == 0.0:
        return "0.0000e0"
    m, e = ("%.4e" % x).split("e")
    return f"{m}e{int(e)}"
