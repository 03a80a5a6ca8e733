"""ctypes binding of libfdgpu.so (include/fdgpu.h). The library is the product; there is no
Python or CPU fallback: if the shared object is missing or no MI355X answers, calls raise."""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libfdgpu.so")

u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
u64p = C.POINTER(C.c_uint64)
f32p = C.POINTER(C.c_float)
VP = C.c_void_p


class BatchDesc(C.Structure):
    _fields_ = [("n_struct", C.c_uint64), ("res_off", u64p), ("n_xyz", f32p), ("ca_xyz", f32p), ("cb_xyz", f32p),
                ("aa", u8p), ("cb_valid", u8p)]


class HashParams(C.Structure):
    _fields_ = [("nbin_dist", C.c_uint32), ("nbin_angle", C.c_uint32), ("dist_cutoff", C.c_float), ("hash_type", C.c_uint32),
                ("n_multiple_bins", C.c_uint32), ("multiple_bins", (C.c_uint32 * 2) * 8)]

    def __init__(self, nbin_dist=0, nbin_angle=0, dist_cutoff=20.0, hash_type=3, multiple_bins=None):
        super().__init__(int(nbin_dist), int(nbin_angle), float(dist_cutoff), int(hash_type))
        mb = list(multiple_bins or [])
        if len(mb) > 8:
            raise ValueError("at most 8 (dist, angle) bin pairs")
        self.n_multiple_bins = len(mb)
        for k, (d, a) in enumerate(mb):
            self.multiple_bins[k][0] = int(d)
            self.multiple_bins[k][1] = int(a)


def parse_multiple_bins(s):
    """parse_distance_angle_pairs (src/utils/cli.rs:1-16): "16-4,8-3" -> [(16, 4), (8, 3)]; malformed items are dropped"""
    out = []
    for item in (s or "").split(","):
        parts = item.strip().split("-")
        if len(parts) == 2:
            try:
                out.append((int(parts[0]), int(parts[1])))
            except ValueError:
                pass
    return out


# HashType::get_with_str / to_string (src/geometry/core.rs:42-75); only the encodings over the PDBTrRosetta descriptor are built
HASH_TYPE_NAMES = {0: "PDBMotif", 1: "PDBMotifSinCos", 2: "TrRosetta", 3: "PDBTrRosetta", 4: "PointPairFeature", 5: "TertiaryInteraction",
                   6: "Hybrid", 7: "FolddiscoAngle", 8: "FolddiscoDist"}
_HASH_TYPE_ALIASES = {"pyscomotif": 0, "orig_pdb": 0, "pdb": 1, "trrosetta": 2, "tr": 2, "pdbtr": 3, "default": 3, "folddisco": 3, "ppf": 4,
                      "tertiary": 5, "3di": 5, "hybrid": 6, "angle": 7, "folddisco_angle": 7, "distance": 8, "dist": 8, "folddisco_dist": 8}


def hash_type_index(name) -> int:
    """HashType::get_with_str: index, canonical name or alias -> HashType index; unknown -> ValueError (the reference's Other)"""
    if isinstance(name, int):
        return name
    s = str(name)
    if s.isdigit() and int(s) in HASH_TYPE_NAMES:
        return int(s)
    for k, v in HASH_TYPE_NAMES.items():
        if v == s:
            return k
    if s in _HASH_TYPE_ALIASES:
        return _HASH_TYPE_ALIASES[s]
    raise ValueError(f"unknown hash type {name!r}")


class CountRec(C.Structure):
    _fields_ = [("nid", C.c_uint32), ("total_match_count", C.c_uint32), ("node_count", C.c_uint32),
                ("edge_count", C.c_uint32), ("idf", C.c_float)]


class PairRec(C.Structure):
    _fields_ = [("cand", C.c_uint32), ("i", C.c_uint32), ("j", C.c_uint32), ("hash", C.c_uint32)]


class CandRec(C.Structure):
    _fields_ = [("cand", C.c_uint32), ("qi", C.c_uint32), ("i", C.c_uint32), ("j", C.c_uint32)]


class MatchQuery(C.Structure):
    _fields_ = [("hashes", u32p), ("n_hashes", C.c_uint64), ("aad_aa1", u8p), ("aad_aa2", u8p), ("aad_dist", f32p),
                ("aad_qi", u32p), ("n_aad", C.c_uint64), ("ca_distance_cutoff", C.c_float), ("use_aa_prefilter", C.c_int)]


class QueryMap(C.Structure):
    _fields_ = [("n", C.c_uint64), ("hash", u32p), ("qi", u32p), ("qj", u32p), ("is_primary", u8p), ("idf", f32p),
                ("n_indices", C.c_uint64), ("indices", u32p),
                ("n_aad", C.c_uint64), ("aad_aa1", u8p), ("aad_aa2", u8p), ("aad_dist", f32p), ("aad_qi", u32p), ("primary_hash", u32p),
                ("post_len", u64p), ("post_seg", u32p), ("post_index_uid", C.c_uint64), ("post_kidx", C.POINTER(C.c_longlong)),
                ("arena_bytes", C.c_uint64)]


class MatchRec(C.Structure):
    _fields_ = [("cand", C.c_uint32), ("same", C.c_uint32), ("idf", C.c_float), ("rmsd", C.c_float), ("rmsd_from_hash", C.c_float),
                ("rot", C.c_float * 9), ("tran", C.c_float * 3), ("metrics", C.c_float * 5),
                ("rot_from_hash", C.c_float * 9), ("tran_from_hash", C.c_float * 3), ("metrics_from_hash", C.c_float * 5)]


class FoldcompAtom(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("z", C.c_float), ("b", C.c_float), ("name", C.c_char * 4), ("res", C.c_char * 3),
                ("chain", C.c_uint8), ("rser", C.c_uint64)]


class Parsed(C.Structure):
    _fields_ = [("n_struct", C.c_uint64), ("n_res", C.c_uint64), ("res_off", u64p), ("n_xyz", f32p), ("ca_xyz", f32p), ("cb_xyz", f32p),
                ("aa", u8p), ("cb_valid", u8p), ("chain", u8p), ("resname_std", u8p), ("serial", u64p), ("bfac", f32p),
                ("resname", C.POINTER(C.c_char)), ("nres_raw", u64p), ("plddt", f32p), ("ok", u8p), ("first_chain", u8p)]


# every symbol include/fdgpu.h declares: (name, restype, argtypes)
SYMBOLS = [
    ("fdgpu_create", C.c_int, [C.c_int, C.POINTER(VP)]),
    ("fdgpu_destroy", None, [VP]),
    ("fdgpu_set_stream", C.c_int, [VP, VP]),
    ("fdgpu_synchronize", C.c_int, [VP]),
    ("fdgpu_release_workspaces", C.c_int, [VP]),
    ("fdgpu_last_error", C.c_char_p, [VP]),
    ("fdgpu_free", None, [VP]),
    ("fdgpu_version", C.c_char_p, []),
    ("fdgpu_batch_upload", C.c_int, [VP, C.POINTER(BatchDesc), C.POINTER(VP)]),
    ("fdgpu_batch_wrap_device", C.c_int, [VP, C.POINTER(BatchDesc), C.c_uint64, C.POINTER(VP)]),
    ("fdgpu_batch_destroy", None, [VP]),
    ("fdgpu_batch_num_structures", C.c_uint64, [VP]),
    ("fdgpu_batch_num_residues", C.c_uint64, [VP]),
    ("fdgpu_hash_batch", C.c_int, [VP, VP, C.POINTER(HashParams), C.c_int, C.POINTER(u32p), C.POINTER(u64p)]),
    ("fdgpu_index_build", C.c_int, [VP, VP, C.POINTER(HashParams), C.c_uint64, C.POINTER(VP)]),
    ("fdgpu_index_export", C.c_int, [VP, VP, C.POINTER(u8p), u64p, C.POINTER(u32p), C.POINTER(u64p), u64p]),
    ("fdgpu_index_load", C.c_int, [VP, u32p, u64p, C.c_uint64, u8p, C.c_uint64, C.c_uint64, C.POINTER(VP)]),
    ("fdgpu_index_destroy", None, [VP]),
    ("fdgpu_index_num_hashes", C.c_uint64, [VP]),
    ("fdgpu_index_value_len", C.c_uint64, [VP]),
    ("fdgpu_index_num_postings", C.c_uint64, [VP]),
    ("fdgpu_index_num_structures", C.c_uint64, [VP]),
    ("fdgpu_index_save", C.c_int, [VP, VP, C.c_char_p]),
    ("fdgpu_posting_lengths", C.c_int, [VP, VP, u32p, C.c_uint64, u64p]),
    ("fdgpu_count_query", C.c_int, [VP, VP, u32p, u32p, u32p, f32p, C.c_uint64, f32p, C.POINTER(C.POINTER(CountRec)), u64p]),
    ("fdgpu_count_query_batch", C.c_int, [VP, VP, C.c_uint64, u64p, u32p, u32p, u32p, f32p, f32p, C.POINTER(C.POINTER(CountRec)), C.POINTER(u64p)]),
    ("fdgpu_spec_fallbacks", C.c_int, [VP, u64p]),
    ("fdgpu_parse_structures", C.c_int, [C.POINTER(C.c_char_p), C.c_uint64, C.c_uint32, C.c_uint64, C.POINTER(C.POINTER(Parsed))]),
    ("fdgpu_parsed_free", None, [C.POINTER(Parsed)]),
    ("fdgpu_foldcomp_decode", C.c_int, [C.c_char_p, C.c_uint64, C.POINTER(C.POINTER(FoldcompAtom)), u64p]),
    ("fdgpu_foldcomp_db_list", C.c_int, [C.c_char_p, C.POINTER(u64p), C.POINTER(C.c_void_p), u64p]),
    ("fdgpu_parse_foldcomp_db", C.c_int, [C.c_char_p, u64p, C.c_uint64, C.c_uint32, C.c_uint64, C.POINTER(C.POINTER(Parsed))]),
    ("fdgpu_get_entries", C.c_int, [VP, VP, u32p, C.c_uint64, C.POINTER(u32p), C.POINTER(u64p)]),
    ("fdgpu_count_query_batch_top", C.c_int, [VP, VP, C.c_uint64, u64p, u32p, u32p, u32p, f32p, f32p, C.c_uint32, C.POINTER(C.POINTER(CountRec)), C.POINTER(u64p)]),
    ("fdgpu_index_set_penalty", C.c_int, [VP, VP, f32p]),
    ("fdgpu_count_query_maps_top", C.c_int, [VP, VP, C.c_uint64, C.c_void_p, f32p, C.c_float, C.c_uint32, C.POINTER(C.POINTER(CountRec)), C.POINTER(u64p)]),
    ("fdgpu_match_pairs", C.c_int, [VP, VP, u8p, u32p, C.c_uint64, C.POINTER(MatchQuery), C.POINTER(HashParams),
                                    C.POINTER(C.POINTER(PairRec)), u64p, C.POINTER(C.POINTER(CandRec)), u64p]),
    ("fdgpu_kabsch_batch", C.c_int, [VP, f32p, f32p, u64p, C.c_uint64, f32p, f32p, f32p]),
    ("fdgpu_lms_qcp_batch", C.c_int, [VP, f32p, f32p, u64p, C.c_uint64, f32p, f32p, f32p, u32p, u32p]),
    ("fdgpu_last_timings", C.c_int, [VP, C.POINTER(C.c_char_p), f32p, u64p, C.c_int]),
    ("fdgpu_enable_timing", C.c_int, [VP, C.c_int]),
    ("fdgpu_pair_features", C.c_int, [VP, VP, C.c_uint64, u32p, u32p, C.c_uint64, C.POINTER(HashParams), f32p, u8p]),
    ("fdgpu_hash_features", C.c_int, [VP, f32p, C.c_uint64, C.POINTER(HashParams), u32p]),
    ("fdgpu_make_query_map_batch", C.c_int, [VP, VP, C.c_uint64, u32p, u64p, u32p, C.POINTER(u8p), u32p, f32p, C.c_uint64, f32p, C.c_uint64,
                                              C.POINTER(HashParams), VP, C.c_float, C.POINTER(C.POINTER(QueryMap))]),
    ("fdgpu_make_query_map", C.c_int, [VP, VP, u32p, C.c_uint64, C.POINTER(u8p), u32p, f32p, C.c_uint64, f32p, C.c_uint64,
                                       C.POINTER(HashParams), VP, C.c_float, C.POINTER(C.POINTER(QueryMap))]),
    ("fdgpu_query_map_free", None, [C.POINTER(QueryMap)]),
    ("fdgpu_retrieve_batch", C.c_int, [VP, VP, u8p, C.c_uint64, u32p, u64p, C.POINTER(C.POINTER(QueryMap)), VP, u32p, C.POINTER(HashParams),
                                        C.c_float, C.c_uint32, C.c_uint32, C.POINTER(C.POINTER(MatchRec)), C.POINTER(u64p), C.POINTER(C.POINTER(C.c_int32)),
                                        C.POINTER(u64p)]),
    ("fdgpu_query_batch", C.c_int, [VP, VP, VP, u8p, VP, C.c_uint64, u32p, u64p, u32p, C.POINTER(u8p), u32p, f32p, C.c_uint64, f32p, C.c_uint64,
                                    C.POINTER(HashParams), C.c_float, f32p, C.c_uint32, C.c_uint32, C.c_float, C.c_uint32, C.POINTER(C.POINTER(QueryMap)),
                                    C.POINTER(C.POINTER(CountRec)), C.POINTER(u64p), C.POINTER(C.POINTER(MatchRec)), C.POINTER(u64p),
                                    C.POINTER(C.POINTER(C.c_int32)), C.POINTER(u64p)]),
    ("fdgpu_query_batch_submit", C.c_int, [VP, VP, VP, u8p, VP, C.c_uint64, u32p, u64p, u32p, C.POINTER(u8p), u32p, f32p, C.c_uint64, f32p, C.c_uint64,
                                           C.POINTER(HashParams), C.c_float, f32p, C.c_uint32, C.c_uint32, C.c_float, C.c_uint32, C.POINTER(VP)]),
    ("fdgpu_query_batch_wait", C.c_int, [VP, VP, C.POINTER(C.POINTER(QueryMap)), C.POINTER(C.POINTER(CountRec)), C.POINTER(u64p), C.POINTER(C.POINTER(MatchRec)),
                                         C.POINTER(u64p), C.POINTER(C.POINTER(C.c_int32)), C.POINTER(u64p)]),
    ("fdgpu_query_lanes", C.c_int, [VP, C.c_uint32]),
    ("fdgpu_trim", None, []),
    ("fdgpu_reserve_staging", C.c_int, [VP]),
    ("fdgpu_write_lookup", C.c_int, [C.c_char_p, C.c_char_p, C.c_uint64, u64p, f32p, u64p]),
    ("fdgpu_format_f32_display", C.c_int, [f32p, C.c_uint64, C.c_char_p]),
    ("fdgpu_retrieve", C.c_int, [VP, VP, u8p, u32p, C.c_uint64, C.POINTER(QueryMap), VP, C.POINTER(HashParams), C.c_float, C.c_uint32, C.c_uint32,
                                 C.POINTER(C.POINTER(MatchRec)), u64p, C.POINTER(C.POINTER(C.c_int32))]),
    ("fdgpu_matches_free", None, [C.POINTER(MatchRec), C.POINTER(C.c_int32)]),
    ("fdgpu_merge_subindices", C.c_int, [C.c_uint64, C.POINTER(u8p), C.POINTER(u32p), C.POINTER(u64p), u64p, C.POINTER(u8p), u64p,
                                         C.POINTER(u32p), C.POINTER(u64p), u64p]),
    ("fdgpu_index_merge", C.c_int, [VP, C.POINTER(VP), C.c_uint64, C.POINTER(VP)]),
    ("fdgpu_posting_bytes", C.c_int, [VP, VP, u32p, C.c_uint64, u64p]),
    ("fdgpu_index_set_first_id", C.c_int, [VP, C.c_uint64]),
    ("fdgpu_host_libm_matches", C.c_int, [VP]),
    ("fdgpu_metrics_batch", C.c_int, [VP, f32p, f32p, u64p, C.c_uint64, f32p, f32p, f32p]),
    ("fdgpu_hash_batch_rows", C.c_int, [VP, VP, C.POINTER(HashParams), C.POINTER(u32p), C.POINTER(u32p), C.POINTER(u64p)]),
    ("fdgpu_hypergeom_enrichment", C.c_int, [u64p, u64p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint32, C.POINTER(C.c_double)]),
    ("fdgpu_comm_unique_id", C.c_int, [u8p]),
    ("fdgpu_comm_init", C.c_int, [VP, u8p, C.c_int, C.c_int, C.POINTER(VP)]),
    ("fdgpu_comm_destroy", None, [VP]),
    ("fdgpu_comm_rank", C.c_int, [VP]),
    ("fdgpu_comm_world", C.c_int, [VP]),
    ("fdgpu_allreduce_lengths", C.c_int, [VP, VP, u64p, C.c_uint64]),
    ("fdgpu_sharded_count_query", C.c_int, [VP, VP, VP, C.c_uint64, u64p, u32p, u32p, u32p, f32p, C.c_uint64, C.c_uint32,
                                            C.POINTER(C.POINTER(CountRec)), C.POINTER(u64p)]),
    ("fdgpu_comm_stats", C.c_int, [VP, u64p, u64p]),
    ("fdgpu_index_range_bounds", C.c_int, [VP, VP, C.c_uint32, u32p]),
    ("fdgpu_index_slice", C.c_int, [VP, VP, C.c_uint64, C.c_uint64, C.POINTER(VP)]),
    ("fdgpu_comm_single_index", C.c_int, [VP, VP, VP, C.POINTER(VP), u64p, u64p, u64p, u64p]),
    ("fdgpu_index_save_part", C.c_int, [VP, VP, C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_int, C.c_int]),
    ("fdgpu_sharded_count_query_maps", C.c_int, [VP, VP, VP, C.c_uint64, C.c_void_p, f32p, C.c_uint64, C.c_uint32,
                                                 C.POINTER(C.POINTER(CountRec)), C.POINTER(u64p)]),
    ("fdgpu_sharded_retrieve", C.c_int, [VP, VP, VP, C.c_uint64, u8p, C.c_uint64, u32p, u64p, C.POINTER(C.POINTER(QueryMap)), VP, u32p,
                                         C.POINTER(HashParams), C.c_float, C.c_uint32, C.c_uint32, C.POINTER(C.POINTER(MatchRec)), C.POINTER(u64p),
                                         C.POINTER(C.POINTER(C.c_int32)), C.POINTER(u64p)]),
    ("fdgpu_query_maps_lengths", C.c_int, [VP, VP, C.c_uint64, C.c_void_p, u64p]),
    ("fdgpu_count_query_maps_top_global", C.c_int, [VP, VP, C.c_uint64, C.c_void_p, u64p, f32p, C.c_float, C.c_uint32,
                                                    C.POINTER(C.POINTER(CountRec)), C.POINTER(u64p)]),
    ("fdgpu_comm_message_bytes", C.c_uint64, [C.c_uint64, C.c_uint32]),
    ("fdgpu_debug_merge_gathered", C.c_int, [VP, C.c_uint32, C.c_uint64, C.c_uint32, u8p, C.POINTER(C.POINTER(CountRec)), C.POINTER(u64p)]),
    ("fdgpu_debug_libm", C.c_int, [VP, C.c_int, f32p, f32p, f32p, C.c_uint64]),
    ("fdgpu_ingest_stats", None, [C.POINTER(C.c_double), C.c_int]),
    ("fdgpu_debug_gunzip", C.c_int, [u8p, C.c_uint64, C.POINTER(u8p), u64p]),
    ("fdgpu_debug_host_components", C.c_int, [u32p, u32p, C.c_uint64, C.c_uint32, C.POINTER(u32p), C.POINTER(u64p), u64p]),
    ("fdgpu_debug_hash_is_symmetric", C.c_int, [C.c_uint32, u32p, C.c_uint64, u8p]),
    ("fdgpu_debug_merge_retrieved", C.c_int, [VP, C.c_uint32, C.c_uint64, u64p, C.POINTER(C.POINTER(MatchRec)), C.POINTER(C.POINTER(C.c_int32)), u64p,
                                              C.POINTER(C.POINTER(MatchRec)), C.POINTER(u64p), C.POINTER(C.POINTER(C.c_int32)), C.POINTER(u64p)]),
]

_lib = None


def load() -> C.CDLL:
    """Load libfdgpu.so. `import torch` first (if torch is installed) so that the HIP runtime torch
    bundles is the one both share (same SONAME libamdhip64.so.7)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -m folddisco_amd.build` (hipcc, gfx950). "
            "folddisco_amd has no CPU fallback.")
    try:
        import torch  # noqa: F401  (shares the HIP runtime; optional)
    except Exception:
        pass
    L = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(L, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = L
    return L


def owned_view(lib, ptr, nbytes, dtype):
    """numpy view of `nbytes` of a library-owned output array WITHOUT a copy: the array (and every slice of it) keeps the block alive and
    fdgpu_free takes it back when the last view is gone (the result arrays of a 128-query batch are megabytes: copying them out cost
    0.2 ms per batch).  ptr: ctypes pointer returned by the library (released here even when empty)."""
    import weakref
    import numpy as _np
    addr = C.cast(ptr, C.c_void_p).value
    if not nbytes or not addr:
        if addr:
            lib.fdgpu_free(C.c_void_p(addr))
        return _np.zeros(0, _np.uint8).view(dtype)
    buf = (C.c_uint8 * nbytes).from_address(addr)
    weakref.finalize(buf, lib.fdgpu_free, C.c_void_p(addr))
    return _np.frombuffer(buf, dtype=_np.uint8).view(dtype)
