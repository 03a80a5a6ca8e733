"""`python -m folddisco_amd index|query …` — the reference's two hot-path subcommands with its flag names and defaults
(src/cli/main.rs:26-110, src/cli/workflows/build_index.rs:64-241, src/cli/workflows/query_pdb.rs:144-519), driving the
GPU path through the C ABI.  Structure order = lexicographic path order (the reference uses readdir order, which is
filesystem dependent; SURVEY §7 hard part 3).  Only the default PDBTrRosetta encoding is supported; input is PDB or mmCIF, optionally gzip."""
from __future__ import annotations

import argparse
import os
import sys

import numpy as np

from folddisco_amd._lib import HASH_TYPE_NAMES, hash_type_index, parse_multiple_bins


def _load_paths(d: str, recursive: bool):
    out = []
    if os.path.isfile(d):
        return [d]
    for root, dirs, files in os.walk(d):
        for f in files:
            if f.lower().endswith((".pdb", ".ent", ".pdb.gz", ".ent.gz", ".cif", ".cif.gz", ".mmcif", ".mmcif.gz")):
                out.append(os.path.join(root, f))
        if not recursive:
            break
    return sorted(out)


def _init_dist(a):
    """torchrun / torch.distributed.run launch: one rank per GPU (backend nccl = RCCL; FD_BENCH_BACKEND=gloo puts several ranks on one
    GPU for plumbing tests).  -> (rank, world, torch device or None)"""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return 0, 1, None
    import torch
    import torch.distributed as dist
    rank, local = int(os.environ["RANK"]), int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("FD_BENCH_BACKEND", "nccl")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    if backend != "nccl":
        local = local % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local)
    a.device = local
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, torch.device("cuda", local)


LAST_TIMINGS = {}      # stage times of the last `index` run in this process (bench.py's drop-in leg reads them)


def _shard_prefix(prefix, rank, world):
    return f"{prefix}.shard{rank}of{world}"


def cmd_index(a):
    import folddisco_amd as fd
    from folddisco_amd import indexio, structure
    rank, world, _dev = _init_dist(a)
    # an input that is a file is a Foldcomp database (build_index.rs:109-123): structures = its entries in key order, names from
    # DB.lookup, ids for the reader = database keys
    a.fc = structure.FoldcompDb(a.pdbs) if structure.is_foldcomp_db(a.pdbs) else None
    if a.fc is not None:
        keep = [k for k, nm in enumerate(a.fc.names) if nm]
        all_paths, a.fc_keys = [a.fc.names[k] for k in keep], a.fc.keys[keep]
    else:
        all_paths, a.fc_keys = _load_paths(a.pdbs, a.recursive), None
    if not all_paths:
        sys.exit(f"[FAIL] no structures under {a.pdbs}")
    prefix = a.index or (_default_index_prefix(a.pdbs))
    if world > 1:
        return _cmd_index_sharded(a, fd, indexio, structure, all_paths, prefix, rank, world)
    paths = all_paths
    import time
    T = a.timings = {"ingest_s": 0.0, "gpu_build_s": 0.0, "merge_s": 0.0, "export_write_s": 0.0, "chunks": 0}
    t_all = time.perf_counter()
    ctx = fd.Context(a.device)
    # The reference walks its input in chunks and parses / hashes a chunk in parallel (controller/mod.rs:282-348).  Here a host thread
    # ingests chunk k + 1 (native, multi-threaded: csrc/fd_ingest.cpp; a structure above the reference's hard-wired 65,535 residues keeps its id but has no hashes,
    # nres 0 and plddt 0, controller/mod.rs:313-318) WHILE the GPU builds the sub-index of chunk k (one fdgpu_index_build call, < 2^32 residue
    # pairs); the sub-indices stay resident in HBM and are concatenated per hash on the device (fdgpu_index_merge, in rounds of 64), so
    # the index crosses the bus once, in the reference's on-disk layout.
    from concurrent.futures import ThreadPoolExecutor
    tid_pool = ThreadPoolExecutor(1)
    tids_f = tid_pool.submit(lambda: [indexio.parse_path_by_id_type(x, a.id) for x in paths])      # 20 ms of Python per 20,000 paths: under the ingest
    parts, nres, plddt = _build_chunks(a, fd, structure, ctx, paths, 0, resident=True)
    t0 = time.perf_counter()
    ix = _merge_resident(fd, parts)
    ctx.synchronize()
    T["merge_s"] = time.perf_counter() - t0
    t0 = time.perf_counter()
    ix.save(prefix)                               # the library writes PREFIX and PREFIX.offset itself (byte-identical to save_offset_to_file)
    T["save_s"] = time.perf_counter() - t0
    n_hashes, value_len = ix.num_hashes, ix.value_len
    indexio.save_lookup(prefix + ".lookup", tids_f.result(), nres, plddt, db_keys=a.fc_keys)
    tid_pool.shutdown()
    indexio.save_type(prefix + ".type", len(paths), grid_width=a.grid, max_residue=a.max_residue, nbin_angle=a.angle, nbin_dist=a.distance, hash_type=HASH_TYPE_NAMES[a.hash_type], multiple_bins=a.multi,
                      **(dict(input_format="FCZDB", foldcomp_db=a.pdbs) if a.fc is not None else {}))
    T["export_write_s"] = time.perf_counter() - t0
    T["total_s"] = time.perf_counter() - t_all
    T["structures"] = len(paths)
    LAST_TIMINGS.clear()
    LAST_TIMINGS.update(T)
    if a.verbose:
        print(f"[DONE] {len(paths)} structures, {n_hashes} hashes, {value_len} value bytes -> {prefix}", file=sys.stderr)
        print("[TIME] total %.2f s = %.0f structures/s; ingest %.2f s (host threads, overlapped with the GPU), GPU builds %.2f s, device merge %.2f s, "
              "export + files %.2f s" % (T["total_s"], len(paths) / max(T["total_s"], 1e-9), T["ingest_s"], T["gpu_build_s"], T["merge_s"], T["export_write_s"]), file=sys.stderr)


def _merge_resident(fd, parts):
    """sub-indices resident in HBM -> one resident index (fdgpu_index_merge takes at most 64 parts: merge in rounds)"""
    g = max(2, min(64, int(os.environ.get("FD_MERGE_GROUP", "64"))))      # parts per fdgpu_index_merge call (tests lower it to exercise the rounds)
    while len(parts) > 1:
        nxt = []
        for k in range(0, len(parts), g):
            grp = parts[k:k + g]
            nxt.append(fd.FolddiscoIndexSet(grp).merge() if len(grp) > 1 else grp[0])
        parts = nxt
    return parts[0]


def _default_index_prefix(pdbs: str) -> str:
    """build_index.rs:75-88 / controller/io.rs:460-470: DIR_folddisco; a database named X_foldcomp gives X_folddisco"""
    p = pdbs.rstrip("/")
    return p.replace("_foldcomp", "_folddisco") if p.endswith("_foldcomp") else p + "_folddisco"


# The reference's skip threshold is NOT its -n flag: Folddisco::new hard-wires max_residue = DEFAULT_MAX_RESIDUE = 65535
# (controller/mod.rs:40,124), set_max_residue (mod.rs:189) has no caller, and the test at mod.rs:313 therefore compares with 65535 whatever
# -n says; -n/--residue (cli/main.rs:42, default 50000) only lands in PREFIX.type (build_index.rs:222).  A structure of 50,001 ... 65,535
# residues is indexed by the reference, so it is indexed here.
REF_SKIP_MAX_RESIDUE = 65535


def _ingest(a, structure, chunk, pos0):
    """native ingest of one chunk of the input (files, or entries pos0 ... of the Foldcomp database) + the reference's warnings"""
    if a.fc is not None:
        ps, nres_c, plddt_c, raw, ok = structure.read_packed(a.fc_keys[pos0:pos0 + len(chunk)], threads=a.threads, max_residue=REF_SKIP_MAX_RESIDUE, foldcomp=a.fc)
    else:
        ps, nres_c, plddt_c, raw, ok = structure.read_packed(chunk, threads=a.threads, max_residue=REF_SKIP_MAX_RESIDUE)
    for k in np.nonzero(ok == 0)[0]:
        print(f"[WARN] {chunk[k]} could not be read. Skipping", file=sys.stderr)
    for k in np.nonzero(raw > REF_SKIP_MAX_RESIDUE)[0]:
        print(f"[WARN] {chunk[k]} has too many residues. Skipping", file=sys.stderr)
    return ps, nres_c, plddt_c, raw, ok


def _build_chunks(a, fd, structure, ctx, paths, first_id, resident=False):
    """chunked GPU builds of paths (ids first_id ...), double-buffered: a host thread ingests chunk k + 1 while the GPU builds chunk k
    -> (sub-indices: resident FolddiscoIndex objects, or exported (value, hashes, offsets) triples; nres; plddt)"""
    import time
    from concurrent.futures import ThreadPoolExecutor
    T = getattr(a, "timings", None) or {}
    nres_all, plddt_all, parts = [], [], []
    starts = list(range(0, len(paths), a.chunk))

    def ingest(c0):
        t0 = time.perf_counter()
        r = _ingest(a, structure, paths[c0:c0 + a.chunk], first_id + c0)      # native, multi-threaded; releases the GIL
        return r, time.perf_counter() - t0
    with ThreadPoolExecutor(1) as pool:
        fut = pool.submit(ingest, starts[0]) if starts else None
        if resident:
            ctx.L.fdgpu_reserve_staging(ctx.h)      # the save's page-locked staging slots, made while the first chunk is parsed (this thread only waits otherwise)
        for k, c0 in enumerate(starts):
            (ps, nres_c, plddt_c, raw, ok), t_ing = fut.result()
            fut = pool.submit(ingest, starts[k + 1]) if k + 1 < len(starts) else None      # the next chunk is parsed while this one is built
            T["ingest_s"] = T.get("ingest_s", 0.0) + t_ing
            nres_all.append(nres_c)
            plddt_all.append(np.where(nres_c > 0, plddt_c, np.float32(0.0)).astype(np.float32))
            t0 = time.perf_counter()
            batch = ctx.upload(ps)
            ix = fd.FolddiscoIndex.build(ctx, batch, first_id=first_id + c0, nbin_dist=a.distance, nbin_angle=a.angle, dist_cutoff=a.grid, hash_type=a.hash_type, multiple_bins=a.multi)
            if resident:
                parts.append(ix)
            else:
                parts.append(ix.export())
                del ix
            del batch
            ctx.synchronize()
            T["gpu_build_s"] = T.get("gpu_build_s", 0.0) + time.perf_counter() - t0
            T["chunks"] = T.get("chunks", 0) + 1
    z = lambda dt: np.zeros(0, dt)
    if not parts and resident:
        parts = [fd.FolddiscoIndex.build(ctx, ctx.upload(fd.PackedStructures.concat([])), first_id=first_id)]
    return parts, (np.concatenate(nres_all) if nres_all else z(np.uint64)), (np.concatenate(plddt_all) if plddt_all else z(np.float32))


def _cmd_index_sharded(a, fd, indexio, structure, paths, prefix, rank, world):
    """Index build sharded by structure: rank r indexes the contiguous id range shard_range(r), no data-path collective;
    the shards stay on disk (PREFIX.shard<r>of<W>, used by the sharded query) and rank 0 concatenates them per hash into the
    reference's single PREFIX / PREFIX.offset."""
    import torch.distributed as dist
    from folddisco_amd import dist as fdist
    lo, hi = fdist.shard_range(rank, world, len(paths))
    ctx = fd.Context(a.device)
    a.timings = {}
    parts, nres, plddt = _build_chunks(a, fd, structure, ctx, paths[lo:hi], lo, resident=True)
    local = _merge_resident(fd, parts)              # the rank's chunks merged on the device
    local.save(_shard_prefix(prefix, rank, world))  # the shard stays on disk for the sharded query
    # ONE index for the database without the host (Option A; csrc/fd_shard_index.hip): hash ranges of equal posting bytes, piece j of
    # every rank's sub-index to rank j — ncclSend / ncclRecv inside the library when the ranks have a GPU each (backend nccl), torch.distributed objects
    # under gloo —, per-hash concatenation of the pieces on the device, every rank writes its regions of PREFIX / PREFIX.offset
    # (FD_INDEX_EXCHANGE=objects keeps the transport-free form reachable under nccl as well: a fall-back should the N-rank ncclSend / ncclRecv group —
    # executed so far with a world of one only — misbehave on a node)
    if dist.get_backend() == "nccl" and os.environ.get("FD_INDEX_EXCHANGE", "rccl") != "objects":
        comm = fdist.Comm(ctx, rank, world)
        rng, hb, vb, ht, vt = comm.single_index(local)
    else:
        comm = None
        rng, hb, vb, ht, vt = fdist.single_index_over_process_group(ctx, local)
    if rank == 0:
        for ext in ("", ".offset"):                 # regions are written in place: a stale file of another size must not survive beside them
            if os.path.exists(prefix + ext):
                os.remove(prefix + ext)
    dist.barrier()
    rng.save_part(prefix, hb, vb, ht, vt, write_header=(rank == 0), is_last=(rank == world - 1))
    del rng, comm
    box = [None] * world
    dist.all_gather_object(box, (nres, plddt))
    dist.barrier()
    if rank == 0:
        indexio.save_lookup(prefix + ".lookup", [indexio.parse_path_by_id_type(x, a.id) for x in paths], np.concatenate([b[0] for b in box]), np.concatenate([b[1] for b in box]), db_keys=a.fc_keys)
        indexio.save_type(prefix + ".type", len(paths), grid_width=a.grid, max_residue=a.max_residue, nbin_angle=a.angle, nbin_dist=a.distance, hash_type=HASH_TYPE_NAMES[a.hash_type], multiple_bins=a.multi,
                          **(dict(input_format="FCZDB", foldcomp_db=a.pdbs) if a.fc is not None else {}))
        if a.verbose:
            print(f"[DONE] {len(paths)} structures over {world} ranks, {ht} hashes, {vt} value bytes -> {prefix}", file=sys.stderr)
    dist.barrier()
    dist.destroy_process_group()


def cmd_query(a):
    import folddisco_amd as fd
    from folddisco_amd import indexio, query, structure
    if not a.index:
        sys.exit("[FAIL] -i/--index is required")
    rank, world, dev = _init_dist(a)
    ctx = fd.Context(a.device)
    tids, nres, plddt, db_keys = indexio.load_lookup(a.index + ".lookup")
    cfg = indexio.load_type(a.index + ".type")
    shard = None
    lo, hi = 0, len(tids)
    if world > 1:
        # query sharded by structure id: this rank loads its shard of the index (written by the multi-rank index build) and the
        # coordinates of its own structures; ids in the shard are absolute, so it is loaded over the whole id range
        from folddisco_amd import dist as fdist
        sp = _shard_prefix(a.index, rank, world)
        if not os.path.exists(sp + ".offset"):
            sys.exit(f"[FAIL] {sp}.offset not found: build the index with the same number of ranks")
        lo, hi = fdist.shard_range(rank, world, len(tids))
        v, h, o = indexio.read_index_files(sp)
        ix = fd.FolddiscoIndex.load(ctx, h, o, v, len(tids))
        shard = dict(lo=lo, n_local=hi - lo, device=dev if os.environ.get("FD_BENCH_BACKEND", "nccl") == "nccl" else None, absolute=True)
    else:
        v, h, o = indexio.read_index_files(a.index)
        ix = fd.FolddiscoIndex.load(ctx, h, o, v, len(tids))
    if a.query.endswith((".txt", ".tsv")):
        queries = []
        for line in open(a.query):
            p = line.rstrip("\n").split("\t")
            queries.append((p[0], p[1] if len(p) > 1 else "", p[2] if len(p) > 2 else ""))
    else:
        queries = [(a.pdb, a.query, a.output)]
    # candidate coordinates: resolve tids like resolve_tid_path_from_index_prefix (controller/io.rs:488-528)
    def resolve(t):
        if os.path.isfile(t):
            return t
        cand = os.path.join(os.path.dirname(os.path.abspath(a.index)), t)
        return cand if os.path.isfile(cand) else t
    db_structs, batch = None, None
    # an index built from a Foldcomp database reads the hit coordinates back from it by db_key (query_pdb.rs:321-343,
    # retrieve.rs:166-178); the database is the one the index names, else INDEX-PREFIX's X_foldcomp sibling (controller/io.rs:422-448)
    fc = None
    if cfg.get("input_format") == "FCZDB" and not a.skip_match:
        cands = [cfg.get("foldcomp_db", "")]
        pfx = a.index[:-len("_folddisco")] if a.index.endswith("_folddisco") else a.index
        cands += [pfx, pfx + "_foldcomp"]
        dbp = next((c for c in cands if c and structure.is_foldcomp_db(c)), None)
        if dbp is None:
            sys.exit(f"[FAIL] Foldcomp database of index {a.index} not found (tried {', '.join(c for c in cands if c)})")
        fc = structure.FoldcompDb(dbp)
    if not a.skip_match and fc is not None:
        db_structs, _ = structure.read_compact_structures(db_keys[lo:hi], threads=a.threads, foldcomp=fc)
        batch = ctx.upload(fd.PackedStructures.concat([s.as_item() for s in db_structs]))
    elif not a.skip_match:
        db_structs, _ = structure.read_compact_structures([resolve(t) for t in tids[lo:hi]], threads=a.threads)
        batch = ctx.upload(fd.PackedStructures.concat([s.as_item() for s in db_structs]))
    dthr = [float(x) for x in a.distance.replace(" ", "").split(",") if x]
    athr = [float(x) for x in a.angle.replace(" ", "").split(",") if x]
    for pdb, qstr, outp in queries:
        if ":" in pdb and not os.path.isfile(pdb):       # DB:NAME = an entry of a Foldcomp database as the query (controller/io.rs:303-333)
            qdb = structure.FoldcompDb(pdb.split(":")[0])
            q = structure.read_compact_structures([qdb.key_of(pdb.split(":")[1])], threads=1, foldcomp=qdb)[0][0]
        else:
            q = structure.read_compact_structures([pdb], threads=1)[0][0]
        rows, matches = query.query_pdb(ctx, ix, batch, db_structs, tids, nres, plddt, q, qstr, dist_thr=dthr, angle_thr=athr,
                                        ca_distance=a.ca_distance, top_n=a.top, skip_match=a.skip_match, serial_query=a.serial_index,
                                        freq_filter=a.freq_filter, length_penalty_power=0.5 if a.length_penalty is None else a.length_penalty,
                                        dist_cutoff=float(cfg.get("grid_width", 20.0)), nbin_dist=int(cfg.get("num_bin_dist", 0)),
                                        nbin_angle=int(cfg.get("num_bin_angle", 0)), hash_type=hash_type_index(cfg.get("hash_type", "PDBTrRosetta")), multiple_bins=cfg.get("multiple_bin"), sampling_ratio=a.sampling_ratio,
                                        sampling_count=a.sampling_count, sort_by=a.sort_by, partial_fit=a.partial_fit, skip_ca_match=a.skip_ca_match, match_top_n=1000 if a.web else "same", db_keys=db_keys,
                                        filters=dict(total_match=a.total_match, covered_node=a.covered_node, covered_node_ratio=a.covered_node_ratio,
                                                     max_node=a.max_node, max_node_ratio=a.max_node_ratio, score=a.score,
                                                     connected_node=a.connected_node, connected_node_ratio=a.connected_node_ratio,
                                                     num_residue=a.num_residue, plddt=a.plddt, rmsd=a.rmsd, tm_score=a.tm_score,
                                                     gdt_ts=a.gdt_ts, gdt_ha=a.gdt_ha, chamfer=a.chamfer, hausdorff=a.hausdorff),
                                        shard=shard)
        if rank != 0:
            continue                      # every rank holds the same result; rank 0 prints
        fh = open(outp, "w") if outp else sys.stdout
        if (a.skip_match or a.per_structure) and not a.web:      # QueryMode::from_flags (controller/mode.rs:231-247): web first
            if a.header:
                fh.write("tid\tidf\ttotal_match_count\tnode_count\tedge_count\tmax_node_cov\tmin_rmsd\tnres\tplddt\tmatching_residues\tdb_key\tquery_residues\n")
            query.sort_rows(rows, query.parse_sort_by(a.sort_by, True))   # StructureSortStrategy (sort.rs:400-458)
            for r in rows:
                fh.write(query.format_structure_row(r, r.get("query_residues", qstr)) + "\n")
        else:
            cols = [c.strip() for c in a.format_output.split(",") if c.strip()] or \
                (query.MATCH_SUPERPOSE_COLUMNS if (a.superpose or a.web) else ["tid", "node_count", "idf", "rmsd", "matching_residues", "query_residues"])
            for c in cols:
                if c not in query.MATCH_COLUMNS:
                    sys.exit(f"[FAIL] unknown --format-output column '{c}' (per-match: {', '.join(query.MATCH_COLUMNS)})")
            if a.header:
                fh.write("\t".join(cols) + "\n")
            for m in matches:
                fh.write(query.format_match_columns(m, cols) + "\n")
        if outp:
            fh.close()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def main(argv=None):
    ap = argparse.ArgumentParser(prog="folddisco_amd")
    sub = ap.add_subparsers(dest="cmd", required=True)
    pi = sub.add_parser("index")
    pi.add_argument("-p", "--pdbs", required=True)
    pi.add_argument("-i", "--index", default="")
    pi.add_argument("-t", "--threads", type=int, default=1)            # host threads of the structure ingest (the hashing and the posting build run on the GPU)
    pi.add_argument("-y", "--type", default="default")
    pi.add_argument("-d", "--distance", type=int, default=0)           # number of distance bins (0 -> 16), main.rs:38
    pi.add_argument("-a", "--angle", type=int, default=0)              # number of angle bins (0 -> 4)
    pi.add_argument("--multiple-bins", default=None)                   # d1-a1,d2-a2 e.g. 16-4,8-3 (build_index.rs:45)
    pi.add_argument("-g", "--grid", type=float, default=20.0)          # CA cutoff
    pi.add_argument("-n", "--residue", "--max-residue", dest="max_residue", type=int, default=50000)   # cli/main.rs:42: written to PREFIX.type only; the skip threshold is the reference's hard-wired 65535 (REF_SKIP_MAX_RESIDUE)
    pi.add_argument("-r", "--recursive", action="store_true")
    pi.add_argument("--id", default="relpath")                          # pdb | uniprot | afdb | relpath | abspath | basename ... (build_index.rs:40)
    pi.add_argument("-v", "--verbose", action="store_true")
    pi.add_argument("--device", type=int, default=0)
    pi.add_argument("--chunk", type=int, default=16384, help="structures per ingest step and GPU build call (the next chunk is parsed while this one is built; the sub-indices are merged on the device)")
    pi.add_argument("--mmap-on-disk", action="store_true", help="accepted for the reference's command lines (indextable.rs:215-226,247: there the posting array is filled in a file-backed "
                    "mapping of PREFIX instead of anonymous memory, the files are the same): here the array is filled in HBM and streamed to PREFIX either way")
    pq = sub.add_parser("query")
    pq.add_argument("-p", "--pdb", default="")
    pq.add_argument("-q", "--query", default="")
    pq.add_argument("-i", "--index", default="")
    pq.add_argument("-t", "--threads", type=int, default=1)
    pq.add_argument("-d", "--distance", default="0.5")
    pq.add_argument("-a", "--angle", default="5")
    pq.add_argument("--ca-distance", type=float, default=1.0)
    pq.add_argument("--top", type=int, default=None)
    pq.add_argument("--skip-match", action="store_true")
    pq.add_argument("--skip-ca-match", action="store_true")          # per-match rows before the C-alpha distance check (result.rs:54-69)
    pq.add_argument("--web", action="store_true")                    # per-match rows with superposition columns, at most 1000 lines (query_pdb.rs:142, 481-493)
    pq.add_argument("--partial-fit", action="store_true")            # LMS superposition for matches of > 3 residues (cli/main.rs:92)
    pq.add_argument("--per-structure", action="store_true")
    pq.add_argument("--per-match", action="store_true")
    pq.add_argument("--header", action="store_true")
    pq.add_argument("--serial-index", action="store_true")
    pq.add_argument("--freq-filter", type=float, default=None)
    pq.add_argument("--sampling-count", type=int, default=None)
    pq.add_argument("--sampling-ratio", type=float, default=None)
    pq.add_argument("--total-match", type=int, default=0)
    pq.add_argument("--covered-node", type=int, default=0)
    pq.add_argument("--covered-node-ratio", type=float, default=0.0)
    pq.add_argument("--max-node", type=int, default=0)
    pq.add_argument("--max-node-ratio", type=float, default=0.0)
    pq.add_argument("--score", type=float, default=0.0)
    pq.add_argument("--connected-node", type=int, default=0)
    pq.add_argument("--connected-node-ratio", type=float, default=0.0)
    pq.add_argument("--num-residue", type=int, default=50000)
    pq.add_argument("--plddt", type=float, default=0.0)
    pq.add_argument("--rmsd", type=float, default=0.0)
    pq.add_argument("--tm-score", type=float, default=0.0)
    pq.add_argument("--gdt-ts", type=float, default=0.0)
    pq.add_argument("--gdt-ha", type=float, default=0.0)
    pq.add_argument("--chamfer", type=float, default=0.0)
    pq.add_argument("--hausdorff", type=float, default=0.0)
    pq.add_argument("--format-output", default="")
    pq.add_argument("--superpose", action="store_true")              # print U, T and the matching C-alpha coordinates
    pq.add_argument("--sort-by", default="")          # cli/main.rs:84: empty -> MatchSortStrategy / StructureSortStrategy default (idf desc, rmsd asc)
    pq.add_argument("--length-penalty", type=float, default=None)
    pq.add_argument("-o", "--output", default="")
    pq.add_argument("-v", "--verbose", action="store_true")
    pq.add_argument("--device", type=int, default=0)
    pa = sub.add_parser("analyze")                                   # src/cli/workflows/analyze.rs:19-40 (summary branch)
    pa.add_argument("-i", "--index", required=True)
    pa.add_argument("-p", "--pdbs", default=None)
    pa.add_argument("-o", "--output", default=None)
    pa.add_argument("--top", type=int, default=10)
    pa.add_argument("--p-value", type=float, default=0.0001)
    pa.add_argument("--min-support", type=int, default=4)
    pa.add_argument("--max-pos", type=int, default=32)
    pa.add_argument("-t", "--threads", type=int, default=1)
    pa.add_argument("-v", "--verbose", action="store_true")
    pa.add_argument("--device", type=int, default=0)
    a = ap.parse_args(argv)
    if a.cmd == "analyze":
        from folddisco_amd import analyze
        if a.pdbs is not None:                                       # enrichment branch (analyze.rs:111-132)
            import folddisco_amd as fd
            paths = _load_paths(a.pdbs, False)
            if not paths:
                sys.exit(f"[FAIL] no structures under {a.pdbs}")
            out = a.output or f"{a.pdbs}_vs_{a.index.split('/')[-1]}"    # analyze.rs:74-79
            ctx = fd.Context(getattr(a, "device", 0))
            en = analyze.enrichment(ctx, a.index, paths, a.p_value, a.threads)
            analyze.save_enrichment(en, paths, out, a.min_support, a.max_pos)
            return
        out = a.output or f"{a.index}_summary"                       # analyze.rs:71-84
        analyze.save_summary(analyze.summarize(a.index), out, a.top)
        return
    if a.cmd == "index":
        if a.mmap_on_disk and a.verbose:
            # the reference's switch chooses WHERE the posting array lives while it is filled (a mapping of PREFIX on disk, indextable.rs:215-226, or
            # anonymous memory copied to PREFIX afterwards, :247-270); PREFIX, PREFIX.offset, .lookup and .type are byte for byte the same in both modes
            print("[INFO] --mmap-on-disk: the posting array is filled in HBM and streamed to PREFIX (same files as without the flag)", file=sys.stderr)
        try:
            a.hash_type = hash_type_index(a.type)
        except ValueError:
            sys.exit(f"[FAIL] unknown hash type {a.type}")
        a.multi = parse_multiple_bins(a.multiple_bins) if a.multiple_bins else None
        if a.multi is not None and a.hash_type in (2, 4, 5, 6):
            sys.exit("[FAIL] --multiple-bins is implemented for the encodings over the PDBTrRosetta descriptor only")
        if a.multi is not None and (not a.multi or len(a.multi) > 8 or any(d == 0 or x == 0 for d, x in a.multi)):
            sys.exit("[FAIL] --multiple-bins: one to eight dist-angle pairs with non-zero counts, e.g. 16-4,8-3")
        cmd_index(a)
    else:
        if a.per_structure and a.per_match:
            sys.exit("[FAIL] --per-structure and --per-match cannot be combined")     # ContradictoryPrintError
        cmd_query(a)


if __name__ == "__main__":
    main()
