"""On-disk index files: PREFIX (value bytes), PREFIX.offset, PREFIX.lookup, PREFIX.type.

The two binary files are written by libfdgpu (`fdgpu_index_save`) or by `write_index_files` for a merged index;
`.lookup` (src/index/lookup.rs:35-56) and `.type` (src/cli/config.rs:66-97) are small text files written here."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib
from ._lib import u8p, u32p, u64p


def format_f32_display(v) -> str:
    """Rust `{}` for f32: shortest digits that round-trip, never exponent form, integral values without fraction."""
    v = np.float32(v)
    if np.isnan(v):
        return "NaN"
    if np.isinf(v):
        return "inf" if v > 0 else "-inf"
    for prec in range(1, 10):
        s = "%.*e" % (prec - 1, float(v))
        if np.float32(float(s)) == v:
            break
    mant, ex = s.split("e")
    neg = mant.startswith("-")
    digits = mant.lstrip("-").replace(".", "").rstrip("0") or "0"
    ex = int(ex)
    if ex < 0:
        out = "0." + "0" * (-ex - 1) + digits
    elif len(digits) <= ex + 1:
        out = digits + "0" * (ex + 1 - len(digits))
    else:
        out = digits[: ex + 1] + "." + digits[ex + 1:]
    return ("-" if neg else "") + out          # -0.0 -> "-0" like Rust


_ID_TYPES = {"Pdb": "pdb", "PDB": "pdb", "pdb": "pdb", "Afdb": "afdb", "AFDB": "afdb", "afdb": "afdb", "Uniprot": "uniprot", "UniProt": "uniprot",
             "uniprot": "uniprot", "BasenameWithoutExt": "stem", "basename_without_ext": "stem", "basename_no_ext": "stem", "filename": "stem",
             "BasenameWithExt": "name", "basename_with_ext": "name", "basename": "name", "file": "name", "AbsPath": "abs", "Abspath": "abs",
             "abspath": "abs", "absolute_path": "abs", "path": "abs", "RelPath": "rel", "Relpath": "rel", "relpath": "rel", "relative_path": "rel",
             "default": "rel"}


def parse_path_by_id_type(path: str, id_type: str) -> str:
    """--id of the index subcommand: IdType::get_with_str + parse_path_by_id_type (src/controller/mode.rs:18-30, 69-126).
    file_stem drops only the last extension (x.pdb.gz -> x.pdb), like std::path::Path::file_stem."""
    import re
    kind = _ID_TYPES.get(id_type, "other")
    name = os.path.basename(path.rstrip("/")) if kind != "abs" else ""
    stem = name[: name.rfind(".")] if name.rfind(".") > 0 else name
    if kind == "pdb":
        return stem[3:] if stem.startswith("pdb") else stem
    if kind in ("afdb", "uniprot"):
        m = re.search(r"AF-.+-model_v\d", stem)
        if not m:
            return stem
        return m.group(0) if kind == "afdb" else m.group(0).split("-")[1]
    if kind == "stem":
        return stem
    if kind == "name":
        return name
    if kind == "abs":
        return os.path.realpath(path)
    return path


def save_lookup(path: str, tids, nres, plddt, db_keys=None):
    """PREFIX.lookup through the library's writer (fdgpu_write_lookup: 20,500 lines cost 0.1 s of Python float formatting otherwise).  An id with a
    newline inside (no path has one) takes the Python writer, which is also what the tests compare the library's bytes with."""
    tids = [str(t) for t in tids]
    n = len(tids)
    if any("\n" in t for t in tids):
        return save_lookup_py(path, tids, nres, plddt, db_keys)
    nr = np.ascontiguousarray(nres, dtype=np.uint64)
    pl = np.ascontiguousarray(plddt, dtype=np.float32)
    dk = None if db_keys is None else np.ascontiguousarray(db_keys, dtype=np.uint64)
    assert len(nr) >= n and len(pl) >= n and (dk is None or len(dk) >= n)
    rc = _lib.load().fdgpu_write_lookup(os.fsencode(path), "\n".join(tids).encode(), n, nr.ctypes.data_as(u64p), pl.ctypes.data_as(_lib.f32p),
                                        None if dk is None else dk.ctypes.data_as(u64p))
    if rc != 0:
        raise IOError(f"cannot write {path}")


def save_lookup_py(path: str, tids, nres, plddt, db_keys=None):
    with open(path, "w") as f:
        for i, tid in enumerate(tids):
            f.write(f"{i}\t{tid}\t{int(nres[i])}\t{format_f32_display(plddt[i])}\t{i if db_keys is None else int(db_keys[i])}\n")


def load_lookup(path: str):
    tids, nres, plddt, keys = [], [], [], []
    with open(path) as f:
        for line in f:
            p = line.rstrip("\n").split("\t")
            tids.append(p[1]); nres.append(int(p[2])); plddt.append(np.float32(p[3])); keys.append(int(p[4]) if len(p) > 4 else int(p[0]))
    return tids, np.array(nres, np.uint64), np.array(plddt, np.float32), np.array(keys, np.uint64)


def save_type(path: str, n_structures: int, grid_width: float = 20.0, max_residue: int = 50000, nbin_angle: int = 0, nbin_dist: int = 0,
              input_format: str = "PDB", hash_type: str = "PDBTrRosetta", multiple_bins=None, foldcomp_db=None):
    """IndexConfig::to_toml (cli/config.rs:66-87): keys in alphabetical order (toml's table is a BTreeMap)"""
    gw = repr(float(grid_width))  # toml prints the f64; 20.0 -> "20.0"
    with open(path, "w") as f:
        f.write(f"chunk_size = {n_structures}\n" + (f"foldcomp_db = \"{foldcomp_db}\"\n" if foldcomp_db else "") + f"grid_width = {gw}\nhash_type = \"{hash_type}\"\ninput_format = \"{input_format}\"\n"
                f"max_residue = {max_residue}\n" + (("multiple_bin = [" + ", ".join(f"[{d}, {a}]" for d, a in multiple_bins) + "]\n") if multiple_bins else "") +
                f"num_bin_angle = {nbin_angle}\nnum_bin_dist = {nbin_dist}\n")


def load_type(path: str) -> dict:
    out = {}
    for line in open(path):
        if "=" in line:
            k, v = (t.strip() for t in line.split("=", 1))
            if v.startswith("[["):    # multiple_bin = [[16, 4], [8, 3]] (cli/config.rs:48-53, 78-84)
                out[k] = [tuple(int(x) for x in item.split(",")) for item in v.strip()[2:-2].split("], [")]
            else:
                out[k] = v.strip('"') if v.startswith('"') else (float(v) if "." in v else int(v))
    return out


def write_index_files(prefix: str, value: np.ndarray, hashes: np.ndarray, offsets: np.ndarray):
    """PREFIX and PREFIX.offset (u64 H | u32 hashes[H] | u64 offsets[H+1]), src/index/indextable.rs:297-326"""
    np.ascontiguousarray(value, np.uint8).tofile(prefix)
    with open(prefix + ".offset", "wb") as f:
        f.write(np.uint64(len(hashes)).tobytes())
        f.write(np.ascontiguousarray(hashes, np.uint32).tobytes())
        f.write(np.ascontiguousarray(offsets, np.uint64).tobytes())


def read_index_files(prefix: str):
    """-> (value, hashes, offsets); accepts the legacy PREFIX.value name (indextable.rs:333-337)"""
    vp = prefix + ".value" if os.path.exists(prefix + ".value") else prefix
    value = np.fromfile(vp, dtype=np.uint8)
    raw = np.fromfile(prefix + ".offset", dtype=np.uint8)
    H = int(raw[:8].view(np.uint64)[0])
    if len(raw) < 8 + 4 * H + 8 * (H + 1):
        raise ValueError("offset file is in an old format or corrupted")
    hashes = raw[8: 8 + 4 * H].view(np.uint32).copy()
    offsets = raw[8 + 4 * H: 8 + 4 * H + 8 * (H + 1)].copy().view(np.uint64)
    return value, hashes, offsets


def merge_subindices(parts):
    """parts: list of (value u8[], hashes u32[], offsets u64[]) over ascending id ranges -> merged (value, hashes, offsets)"""
    L = _lib.load()
    n = len(parts)
    vals = [np.ascontiguousarray(p[0], np.uint8) for p in parts]
    hs = [np.ascontiguousarray(p[1], np.uint32) for p in parts]
    offs = [np.ascontiguousarray(p[2], np.uint64) for p in parts]
    vp = (u8p * n)(*[v.ctypes.data_as(u8p) for v in vals])
    hp = (u32p * n)(*[h.ctypes.data_as(u32p) for h in hs])
    op = (u64p * n)(*[o.ctypes.data_as(u64p) for o in offs])
    nh = np.array([len(h) for h in hs], np.uint64)
    ov, oh, oo = u8p(), u32p(), u64p()
    vl, H = C.c_uint64(), C.c_uint64()
    rc = L.fdgpu_merge_subindices(n, vp, hp, op, nh.ctypes.data_as(u64p), C.byref(ov), C.byref(vl), C.byref(oh), C.byref(oo), C.byref(H))
    if rc != 0:
        raise RuntimeError(f"fdgpu_merge_subindices failed ({rc}): parts must cover ascending id ranges")
    v = np.ctypeslib.as_array(ov, shape=(max(vl.value, 1),))[: vl.value].copy()
    h = np.ctypeslib.as_array(oh, shape=(max(H.value, 1),))[: H.value].copy()
    o = np.ctypeslib.as_array(oo, shape=(H.value + 1,)).copy()
    for p in (ov, oh, oo):
        L.fdgpu_free(p)
    return v, h, o
