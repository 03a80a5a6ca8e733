"""Host-side structure ingest: PDB ATOM records -> CompactStructure arrays.

Mirrors src/structure/io/pdb.rs:37-77 (reader), src/structure/io/parser.rs:3-56 (fixed columns) and
CompactStructure::build (src/structure/core.rs:70-214) including its quirks: residue
boundary = change of res_serial only; the flush also fires at the last atom index before that atom is
examined; chain / b-factor of residue k are read from the first atom of residue k+1; `c` is never reset,
so a virtual CB may use a stale C; GLY N is captured only through the GLY branch.
Text ingest is host work; the numeric hot path starts at PackedStructures.
"""
from __future__ import annotations

import gzip
import math
import os
from dataclasses import dataclass

import numpy as np

_AA3 = ["ALA", "ARG", "ASN", "ASP", "CYS", "GLN", "GLU", "GLY", "HIS", "ILE", "LEU", "LYS", "MET", "PHE", "PRO", "SER",
        "THR", "TRP", "TYR", "VAL"]
# src/utils/convert.rs:53-81
_AA_GROUPS = [
    "ALA ABA ORN DAL AIB ALC MDO MAA DAB", "ARG DAR CIR AGM", "ASN DSG MEN SNN", "ASP 0TD DAS IAS PHD BFD ASX",
    "CYS CSO CSD CME OCS CAS CSX CSS YCM DCY SMC SCH SCY CAF SNC SEC", "GLN DGN CRQ MEQ", "GLU PCA DGL CGU FGA B3E GLX",
    "GLY CR2 SAR GHP GL3", "HIS HIC DHI NEP CR8 MHS", "ILE DIL", "LEU DLE NLE MLE MK8",
    "LYS KCX LLP MLY M3L ALY MLZ DLY KPI PYL", "MET MSE FME NRQ CXM SME MHO MED", "PHE DPN PHI MEA PHL", "PRO HYP DPR",
    "SER CSH SEP DSN SAC GYS DHA OAS", "THR TPO CRO DTH BMT CRF", "TRP DTR TRQ TOX 0AF", "TYR PTR TYS TPQ DTY OMY",
    "VAL DVA MVA FVA"]
AA_MAP = {name: k for k, grp in enumerate(_AA_GROUPS) for name in grp.split()}


def map_aa_to_u8(name: str) -> int:
    return AA_MAP.get(name, 255)


def map_u8_to_aa(aa: int) -> str:
    return _AA3[aa] if aa < 20 else "UNK"


f32 = np.float32


def _norm3(v):
    n = f32(math.sqrt(f32(f32(f32(v[0] * v[0]) + f32(v[1] * v[1])) + f32(v[2] * v[2]))))
    return np.array([v[0] / n, v[1] / n, v[2] / n], dtype=f32)


def _cross(a, b):
    return np.array([f32(a[1] * b[2]) - f32(a[2] * b[1]), f32(a[2] * b[0]) - f32(a[0] * b[2]), f32(a[0] * b[1]) - f32(a[1] * b[0])], dtype=f32)


def approx_cb(ca, n, c):
    """src/structure/coordinate.rs:167-186, every operation rounded to f32 in the reference's order."""
    ca, n, c = (np.asarray(x, dtype=f32) for x in (ca, n, c))
    v1 = _norm3(c - ca)
    v2 = _norm3(n - ca)
    b1 = v2 + v1 * f32(f32(1.0) / f32(3.0))
    b2 = _cross(v1, b1)
    u1 = _norm3(b1)
    u2 = _norm3(b2)
    v4 = u1 * f32(f32(-1.0) / f32(2.0)) - u2 * f32(f32(math.sqrt(f32(3.0))) / f32(2.0))
    v4 = v4 * f32(f32(math.sqrt(f32(8.0))) / f32(3.0))
    v4 = v4 + v1 * f32(f32(-1.0) / f32(3.0))
    return ca + v4 * f32(1.5336)


@dataclass
class CompactStructure:
    n_xyz: np.ndarray
    ca_xyz: np.ndarray
    cb_xyz: np.ndarray
    cb_ok: np.ndarray
    aa: np.ndarray
    resname: list
    chain: np.ndarray      # u8 per residue (quirk: from the next residue's first atom)
    serial: np.ndarray     # u64
    bfac: np.ndarray
    chains: list           # chain ids in order of appearance
    num_residues_raw: int

    @property
    def n(self) -> int:
        return len(self.aa)

    def get_index(self, chain: int, serial: int):
        hit = np.nonzero((self.chain == chain) & (self.serial == serial))[0]
        return int(hit[0]) if len(hit) else None

    def avg_plddt(self) -> np.float32:
        s = f32(0.0)
        for b in self.bfac:
            s = f32(s + f32(b))
        with np.errstate(invalid="ignore", divide="ignore"):
            return f32(s / f32(self.n))

    def resname_std(self) -> np.ndarray:
        return np.array([1 if (a < 20 and r == _AA3[a]) else 0 for a, r in zip(self.aa, self.resname)], dtype=np.uint8)

    def as_item(self) -> dict:
        return dict(n_xyz=self.n_xyz, ca_xyz=self.ca_xyz, cb_xyz=self.cb_xyz, aa=self.aa, cb_ok=self.cb_ok)


def _parse_f32(s: str):
    s = s.strip()
    if not s or "x" in s.lower() or "_" in s:
        return None
    try:
        return f32(s)
    except ValueError:
        return None


def _parse_u64(s: str):
    s = s.strip()
    if s.startswith("+"):
        s = s[1:]
    return int(s) if s.isdigit() else None


def read_atoms(path: str):
    gz = path.endswith(".gz")
    opener = gzip.open if gz else open
    atoms = []
    model = 0
    with opener(path, "rt", errors="replace") as fh:
        for line in fh:
            line = line.rstrip("\r\n")
            if model > 1 and not gz:     # read_structure_from_gz (pdb.rs:79-124) keeps the ATOM records of every model
                break
            if len(line) < 6:
                continue
            rec = line[:6]
            if rec == "MODEL ":
                model += 1
                continue
            if rec != "ATOM  " or len(line) < 54:
                continue
            x, y, z = _parse_f32(line[30:38]), _parse_f32(line[38:46]), _parse_f32(line[46:54])
            aser, rser = _parse_u64(line[6:11]), _parse_u64(line[22:26])
            b = _parse_f32(line[60:66]) if len(line) >= 66 else f32(1.0)
            if None in (x, y, z, aser, rser, b):
                continue
            atoms.append((x, y, z, line[12:16], line[17:20], rser, ord(line[21]), b))
    return atoms


def build_compact(atoms) -> CompactStructure:
    chains, rec_chain, rec_serial, nres_raw = [], ord(" "), 0, 0
    for a in atoms:
        if rec_chain != a[6]:
            chains.append(a[6]); rec_chain = a[6]
        if rec_serial != a[5]:
            nres_raw += 1; rec_serial = a[5]
    N, CA, CB, OK, RN, CH, SE, BF = [], [], [], [], [], [], [], []
    prev_serial = None
    prev_name = None
    n = ca = cb = c = gly_n = gly_c = None
    na = len(atoms)
    for idx, a in enumerate(atoms):
        xyz = np.array(a[:3], dtype=f32)
        if prev_serial != a[5] or idx == na - 1:
            if n is not None and ca is not None:
                N.append(n); CA.append(ca); SE.append(prev_serial); RN.append(prev_name); CH.append(a[6]); BF.append(a[7])
                if cb is not None:
                    CB.append(cb); OK.append(1)
                elif prev_name == "GLY" and gly_n is not None and gly_c is not None:
                    CB.append(approx_cb(ca, gly_n, gly_c)); OK.append(1)
                elif c is not None:
                    CB.append(approx_cb(ca, n, c)); OK.append(1)
                else:
                    CB.append(np.zeros(3, f32)); OK.append(0)
            ca = cb = n = None
            prev_serial, prev_name = a[5], a[4]
        name, res = a[3], a[4]
        if name == " CA ":
            ca = xyz
        elif name == " CB ":
            cb = xyz
        elif name == " C  ":
            c = xyz
        elif name == " N  " and res != "GLY":
            n = xyz
        elif res == "GLY":
            if name == " N  ":
                gly_n = xyz; n = xyz
            elif name == " C  ":
                gly_c = xyz
    m = len(CA)
    arr = lambda v: (np.stack(v).astype(f32) if m else np.zeros((0, 3), f32))
    return CompactStructure(arr(N), arr(CA), arr(CB), np.array(OK, np.uint8), np.array([map_aa_to_u8(r) for r in RN], np.uint8), RN,
                            np.array(CH, np.uint8), np.array(SE, np.uint64), np.array(BF, f32), chains, nres_raw)


def read_compact_structure(path: str) -> CompactStructure:
    return build_compact(read_atoms(path))


class FoldcompDb:
    """A Foldcomp database (DB, DB.index, DB.lookup) as the index / query workflows see it (reference: FoldcompDbReader,
    src/structure/io/fcz.rs:41-74): entries in ascending key order, `keys` = the db_key column of the index's .lookup,
    `names` = the structure names.  Decoding happens in csrc/fd_fcz.cpp through fdgpu_parse_foldcomp_db."""

    def __init__(self, path: str):
        import ctypes as C
        from . import _lib
        L = _lib.load()
        keys, names, n = _lib.u64p(), C.c_void_p(), C.c_uint64()
        rc = L.fdgpu_foldcomp_db_list(os.fsencode(path), C.byref(keys), C.byref(names), C.byref(n))
        if rc != 0:
            raise RuntimeError(f"{path}: not a Foldcomp database (DB.index / DB.lookup unreadable, {rc})")
        self.path = path
        self.keys = np.ctypeslib.as_array(keys, shape=(max(n.value, 1),))[:n.value].astype(np.uint64, copy=True)
        txt = C.string_at(names.value).decode("utf-8", "replace")
        self.names = txt.split("\n")[:n.value]
        L.fdgpu_free(keys)
        L.fdgpu_free(names)
        self._by_name = None

    def __len__(self):
        return len(self.keys)

    def key_of(self, name: str) -> int:
        if self._by_name is None:
            self._by_name = {nm: int(k) for nm, k in zip(self.names, self.keys)}
        return self._by_name[name]


def is_foldcomp_db(path: str) -> bool:
    """cli/workflows/build_index.rs:109-123: an input that is a FILE (not a directory) is a Foldcomp database"""
    return os.path.isfile(path) and os.path.isfile(path + ".index") and os.path.isfile(path + ".lookup")


def _native_parse(paths, threads, max_residue, foldcomp):
    """-> POINTER(fd_parsed); paths = file paths, or (with foldcomp = FoldcompDb) database keys"""
    import ctypes as C
    from . import _lib
    L = _lib.load()
    out = C.POINTER(_lib.Parsed)()
    if foldcomp is not None:
        keys = np.ascontiguousarray(paths, np.uint64)
        if len(keys) == 0:
            keys = np.zeros(1, np.uint64)
            out_n = 0
        else:
            out_n = len(keys)
        if out_n == 0:
            rc = L.fdgpu_parse_structures((C.c_char_p * 1)(), 0, threads, max_residue, C.byref(out))
        else:
            rc = L.fdgpu_parse_foldcomp_db(os.fsencode(foldcomp.path), keys.ctypes.data_as(_lib.u64p), out_n, threads, max_residue, C.byref(out))
        if rc != 0:
            raise RuntimeError(f"fdgpu_parse_foldcomp_db failed ({rc})")
        return out
    arr = (C.c_char_p * max(len(paths), 1))(*[os.fsencode(p) for p in paths])
    rc = L.fdgpu_parse_structures(arr, len(paths), threads, max_residue, C.byref(out))
    if rc != 0:
        raise RuntimeError(f"fdgpu_parse_structures failed ({rc})")
    return out


def read_compact_structures(paths, threads: int = 0, max_residue: int = 0, foldcomp=None):
    """Multi-threaded native ingest (csrc/fd_ingest.cpp, fdgpu_parse_structures): PDB / mmCIF, optionally gzip — or, with
    foldcomp = FoldcompDb, the database entries whose keys are given (fdgpu_parse_foldcomp_db).
    -> (list of CompactStructure, ok flags).  Same arrays, bit for bit, as read_compact_structure()."""
    import ctypes as C
    from . import _lib
    L = _lib.load()
    out = _native_parse(paths, threads, max_residue, foldcomp)
    P = out.contents
    S, R = P.n_struct, P.n_res
    view = lambda ptr, n, dt: (np.ctypeslib.as_array(ptr, shape=(max(n, 1),))[:n].astype(dt, copy=True))
    off = view(P.res_off, S + 1, np.uint64)
    nx, cax, cbx = (view(p, 3 * R, np.float32).reshape(-1, 3) for p in (P.n_xyz, P.ca_xyz, P.cb_xyz))
    aa, ok_cb, chain, ser, bf = view(P.aa, R, np.uint8), view(P.cb_valid, R, np.uint8), view(P.chain, R, np.uint8), view(P.serial, R, np.uint64), view(P.bfac, R, np.float32)
    rn = C.string_at(P.resname, 3 * R).decode("latin-1") if R else ""
    raw, okf, fch = view(P.nres_raw, S, np.uint64), view(P.ok, S, np.uint8), view(P.first_chain, S, np.uint8)
    res = []
    for k in range(S):
        a, b = int(off[k]), int(off[k + 1])
        ch = chain[a:b]
        chains = [int(fch[k])]
        res.append(CompactStructure(nx[a:b], cax[a:b], cbx[a:b], ok_cb[a:b], aa[a:b], [rn[3 * r: 3 * r + 3] for r in range(a, b)], ch, ser[a:b],
                                    bf[a:b], chains, int(raw[k])))
    L.fdgpu_parsed_free(out)
    return res, okf


class _ParsedOwner:
    """keeps one fd_parsed alive for the numpy views read_packed hands out (freed when the last view goes)"""

    def __init__(self, L, out):
        self.L, self.out = L, out

    def __del__(self):
        try:
            if self.out is not None:
                self.L.fdgpu_parsed_free(self.out)
                self.out = None
        except Exception:      # interpreter shutdown: the library handle may already be gone
            pass


class _ParsedView:
    """one array of an fd_parsed as numpy sees it (the array's base is this object, which holds the owner)"""

    def __init__(self, owner, ptr, n, dtype):
        import ctypes as C
        self.owner = owner
        self.__array_interface__ = {"data": (C.cast(ptr, C.c_void_p).value or 0, False), "shape": (n,), "typestr": np.dtype(dtype).str, "version": 3}


def read_packed(paths, threads: int = 0, max_residue: int = 0, foldcomp=None):
    """Native ingest straight into the flat batch layout (no per-structure Python objects): -> (PackedStructures, nres u64[S],
    plddt f32[S], nres_raw u64[S], ok u8[S]).  What the index workflow needs: coordinates for the GPU, nres / plddt for .lookup.
    With foldcomp = FoldcompDb, paths are database keys.  The coordinate / residue-type arrays are views of the library's arrays (a copy
    of them was a fifth of the ingest's wall time: ~40 bytes per residue into fresh pages); the per-structure columns are copies."""
    from . import _lib
    from .api import PackedStructures
    L = _lib.load()
    out = _native_parse(paths, threads, max_residue, foldcomp)
    P = out.contents
    S, R = P.n_struct, P.n_res
    own = _ParsedOwner(L, out)
    zero = lambda ptr, n, dt: np.asarray(_ParsedView(own, ptr, n, dt)) if n else np.zeros(0, dt)
    view = lambda ptr, n, dt: (np.ctypeslib.as_array(ptr, shape=(max(n, 1),))[:n].astype(dt, copy=True))
    off = view(P.res_off, S + 1, np.uint64)
    ps = PackedStructures(off, zero(P.n_xyz, 3 * R, np.float32).reshape(-1, 3), zero(P.ca_xyz, 3 * R, np.float32).reshape(-1, 3),
                          zero(P.cb_xyz, 3 * R, np.float32).reshape(-1, 3), zero(P.aa, R, np.uint8), zero(P.cb_valid, R, np.uint8))
    nres = np.diff(off).astype(np.uint64)
    plddt, raw, okf = view(P.plddt, S, np.float32), view(P.nres_raw, S, np.uint64), view(P.ok, S, np.uint8)
    return ps, nres, plddt, raw, okf
