/* fdgpu.h — C ABI of libfdgpu.so, the MI355X (gfx950) implementation of Folddisco's
 * geometric-hash index-build and motif-query hot path.
 *
 * The reference (steineggerlab/folddisco, Rust) has no FFI for this path; its call seams
 * are plain Rust functions.  Each entry point below replaces one seam and
 * is what a `#[link(name = "fdgpu")] extern "C"` block in the reference would bind
 * (INTEGRATION.md shows the Rust side).  Conventions follow the one FFI precedent in the
 * reference, Foldcomp (lib/foldcomp/foldcompffi.h:8-21, src/structure/io/fcz.rs:82-93):
 * opaque handles, borrowed inputs, malloc'd outputs owned by the caller and released with
 * fdgpu_free().  Unlike Foldcomp nothing aborts: every call returns 0 or a negative
 * FDGPU_E* code and fdgpu_last_error() gives the text.
 *
 * All pointers are plain host pointers unless a parameter says "device".  No torch types.
 */
#ifndef FDGPU_H
#define FDGPU_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FDGPU_OK 0
#define FDGPU_EINVAL (-1)   /* bad argument */
#define FDGPU_EHIP (-2)     /* HIP runtime error (no device, launch failure, OOM) */
#define FDGPU_ENOMEM (-3)   /* host allocation failed */
#define FDGPU_ERANGE (-4)   /* size limit exceeded (e.g. structure > 65535 residues) */

typedef struct fdgpu_ctx fdgpu_ctx;       /* one per GPU / HIP stream */
typedef struct fdgpu_batch fdgpu_batch;   /* packed structures resident in HBM */
typedef struct fdgpu_index fdgpu_index;   /* inverted index resident in HBM */

/* ---- context ------------------------------------------------------------------------- */
/* fdgpu_create runs a start-up self-check (FDGPU_SELFCHECK=0 skips it): the six 4CHA triad hashes the reference's own test holds
 * (src/controller/graph.rs:71-79) through every evaluation path of the pair kernel — a mismatch fails with FDGPU_EHIP — and a probe of
 * the HOST's libm against the glibc generation the device arithmetic restates (fdgpu_host_libm_matches). */
int fdgpu_create(int device, fdgpu_ctx **out);
/* 1: this host's sinf/cosf/acosf/atan2f agree with the restated generation (a reference build here hashes like this library),
 * 0: they differ (warning on stderr), -1: self-check skipped */
int fdgpu_host_libm_matches(const fdgpu_ctx *ctx);
void fdgpu_destroy(fdgpu_ctx *ctx);
/* use an existing hipStream_t (e.g. torch's current stream); NULL = the context's own stream */
int fdgpu_set_stream(fdgpu_ctx *ctx, void *hip_stream);
int fdgpu_synchronize(fdgpu_ctx *ctx);
/* Returns the context's workspaces (an index build keeps 12 bytes per key of its largest call: 80 GB after a 203,250-structure call, 213 GB
 * after Swiss-Prot in one call) and the cached device blocks of destroyed indices to the device.  Indices, batches and query maps stay valid.
 * (No reference counterpart: the reference's builder frees its DashMap / Vec buffers when they go out of scope, src/cli/workflows/build_index.rs.) */
int fdgpu_release_workspaces(fdgpu_ctx *ctx);
const char *fdgpu_last_error(const fdgpu_ctx *ctx);
void fdgpu_free(void *host_ptr);
const char *fdgpu_version(void);

/* ---- packed structures --------------------------------------------------------------------
 * One batch = one rayon chunk of the reference (src/controller/mod.rs:282-348).  Residue k of
 * structure s lives at r = res_off[s] + k.  This is CompactStructure
 * (src/structure/core.rs:56-67) flattened: coordinates interleaved x,y,z as f32. */
typedef struct fd_batch_desc {
    uint64_t n_struct;
    const uint64_t *res_off;   /* [n_struct+1], res_off[0] = 0 */
    const float *n_xyz;        /* [3*R] backbone N  */
    const float *ca_xyz;       /* [3*R] C-alpha     */
    const float *cb_xyz;       /* [3*R] C-beta (real or approx_cb virtual); ignored where !cb_valid */
    const uint8_t *aa;         /* [R] map_aa_to_u8 (src/utils/convert.rs:53-81): 0..19, 255 = unknown */
    const uint8_t *cb_valid;   /* [R] 1 if CB is Some (src/structure/core.rs:147-155); NULL = all 1 */
} fd_batch_desc;

/* Parameters of the encoding.  hash_type = the reference's HashType index (src/geometry/core.rs:26-40), all nine built:
 * 3 PDBTrRosetta (the default, pdb_tr.rs:21-75; the fast table / speculative path) and, over the same
 * (d_CA, d_CB, theta, tau1, tau2) descriptor, 0 PDBMotif (pdb_motif.rs), 1 PDBMotifSinCos (pdb_motif_sincos.rs),
 * 7 FolddiscoAngle (folddisco_angle.rs), 8 FolddiscoDist (folddisco_dist.rs); with their own descriptors (one ordered pair at a
 * time, exact libm, 32-bit hashes) 2 TrRosetta (trrosetta.rs), 4 PointPairFeature (ppf.rs), 5 TertiaryInteraction
 * (tertiary_interaction.rs), 6 Hybrid (hybrid.rs) — for those four fdgpu_hash_batch serves the raw order only, and
 * fdgpu_pair_features / fdgpu_hash_features (seven-float records) are not available.  Retrieval with 5 / 6 scans every residue
 * pair (the reference's amino-acid prefilter panics there for queries of <= 200 hashes).
 * nbin_dist / nbin_angle follow the reference: if either is 0 both take the encoding's defaults
 * (controller/feature.rs:216-223), larger values clamp per encoding. */
#define FDGPU_HASH_PDBMOTIF 0u
#define FDGPU_HASH_PDBMOTIF_SINCOS 1u
#define FDGPU_HASH_TRROSETTA 2u
#define FDGPU_HASH_PDBTR 3u
#define FDGPU_HASH_PPF 4u
#define FDGPU_HASH_TERTIARY 5u
#define FDGPU_HASH_HYBRID 6u
#define FDGPU_HASH_FOLDDISCO_ANGLE 7u
#define FDGPU_HASH_FOLDDISCO_DIST 8u
#define FDGPU_MAX_MULTIPLE_BINS 8
typedef struct fd_hash_params {
    uint32_t nbin_dist;
    uint32_t nbin_angle;
    float dist_cutoff;         /* CA-CA cutoff in Angstrom (strict >, src/structure/core.rs:391) */
    uint32_t hash_type;        /* FDGPU_HASH_*; NOTE 0 is PDBMotif, as in the reference's numbering: set FDGPU_HASH_PDBTR (3) for the default encoding */
    /* --multiple-bins d1-a1,d2-a2,... (src/cli/workflows/build_index.rs:45,149-153): every residue pair is hashed once per
     * (dist, angle) bin pair and all the hashes share one index (controller/feature.rs:211-215); queries insert every
     * expansion under every bin pair (controller/query.rs:59-70) and retrieval reports a found triple per matching bin pair
     * (controller/retrieve.rs:124-131).  0 = off.  Zero bin counts inside the list are rejected (the reference treats them
     * differently at index and at query time).  nbin_dist / nbin_angle above still select the observed hash that idf is
     * looked up for (query.rs:283-288).  Honoured by fdgpu_index_build, fdgpu_make_query_map[_batch], fdgpu_match_pairs and
     * fdgpu_retrieve[_batch]; fdgpu_hash_batch returns FDGPU_EINVAL with it. */
    uint32_t n_multiple_bins;
    uint32_t multiple_bins[FDGPU_MAX_MULTIPLE_BINS][2];
} fd_hash_params;

/* copy a host batch into HBM */
int fdgpu_batch_upload(fdgpu_ctx *ctx, const fd_batch_desc *host, fdgpu_batch **out);
/* wrap arrays that already live in HBM (all pointers in `dev` are device pointers; res_off too).
 * The memory is borrowed, not owned. */
int fdgpu_batch_wrap_device(fdgpu_ctx *ctx, const fd_batch_desc *dev, uint64_t total_residues,
                            fdgpu_batch **out);
void fdgpu_batch_destroy(fdgpu_batch *b);
uint64_t fdgpu_batch_num_structures(const fdgpu_batch *b);
uint64_t fdgpu_batch_num_residues(const fdgpu_batch *b);

/* ---- S1: per-structure hashes -------------------------------------------------------------
 * Replaces get_geometric_hash_as_u32_from_structure (src/controller/feature.rs:198-231)
 * followed by the caller's sort_unstable(); dedup() (src/controller/mod.rs:343-345), for every
 * structure of the batch.  *hashes = concatenated sorted-unique lists, (*hash_off)[s] .. [s+1].
 * With sort_dedup = 0 the raw list in the reference's row-major pair order is returned. */
int fdgpu_hash_batch(fdgpu_ctx *ctx, const fdgpu_batch *b, const fd_hash_params *p, int sort_dedup,
                     uint32_t **hashes, uint64_t **hash_off);

/* The raw list WITH positions: entries [row_off[i], row_off[i + 1]) belong to residue i of the batch (row-major pair order), entry k
 * pairs it with residue partner[k] and hashes[k] is the pair's hash — the (hash, i, j) stream collect_hash_id_pos walks for
 * `analyze -p` (src/controller/summary.rs:632-690).  row_off has n_residues + 1 entries.  Release the three arrays with fdgpu_free. */
int fdgpu_hash_batch_rows(fdgpu_ctx *ctx, const fdgpu_batch *b, const fd_hash_params *p, uint32_t **hashes, uint32_t **partner, uint64_t **row_off);

/* ---- S2: inverted index build ----------------------------------------------------------------
 * Replaces Folddisco::collect_and_count + allocate_entries + add_entries +
 * wrapup_offset_and_save_entries + prune_to_sparse (src/controller/mod.rs:274-441,
 * src/index/indextable.rs:88-295) for the structures of `b`, which receive ids
 * first_id, first_id+1, ...  The result (value bytes, sparse hashes, offsets) stays in HBM.
 * One call of the default encoding holds up to 2^24 structures and 2^35 residue-pair keys and needs 12 bytes of workspace per key
 * (Swiss-Prot, 1.77e10 keys: 213 GB); FDGPU_ERANGE / FDGPU_EHIP (out of memory) beyond that: build consecutive id ranges and
 * fdgpu_index_merge them.  The other encodings and --multiple-bins: 2^32 keys per call. */
int fdgpu_index_build(fdgpu_ctx *ctx, const fdgpu_batch *b, const fd_hash_params *p, uint64_t first_id,
                      fdgpu_index **out);
/* copy the index to the host in the reference's on-disk layout:
 * value = PREFIX file, hashes/offsets = payload of PREFIX.offset. */
int fdgpu_index_export(fdgpu_ctx *ctx, const fdgpu_index *ix, uint8_t **value, uint64_t *value_len,
                       uint32_t **hashes, uint64_t **offsets, uint64_t *n_hashes);
/* upload an existing index (load_folddisco_index, src/index/indextable.rs:331-394).
 * nres[S] = per-structure residue counts from the .lookup file (length penalty). */
int fdgpu_index_load(fdgpu_ctx *ctx, const uint32_t *hashes, const uint64_t *offsets, uint64_t n_hashes,
                     const uint8_t *value, uint64_t value_len, uint64_t n_structures, fdgpu_index **out);
/* an uploaded index that is one shard of a database (structures first_id .. first_id + n_structures - 1, as written by a sharded
 * build): tells scoring and fdgpu_index_merge where the shard's ids start (0 after fdgpu_index_load) */
int fdgpu_index_set_first_id(fdgpu_index *ix, uint64_t first_id);
void fdgpu_index_destroy(fdgpu_index *ix);
uint64_t fdgpu_index_num_hashes(const fdgpu_index *ix);
uint64_t fdgpu_index_value_len(const fdgpu_index *ix);
uint64_t fdgpu_index_num_postings(const fdgpu_index *ix);
uint64_t fdgpu_index_num_structures(const fdgpu_index *ix);   /* structures the index covers (ids first_id ...); a merged index: the sum over its parts */
/* write PREFIX and PREFIX.offset byte-identically to save_offset_to_file (indextable.rs:297-326) */
int fdgpu_index_save(fdgpu_ctx *ctx, const fdgpu_index *ix, const char *prefix);
/* optional: make the page-locked staging slots of fdgpu_index_save / _save_part / _export now (256 MB, ~20 ms) — e.g. while the first chunk is still parsed */
int fdgpu_reserve_staging(fdgpu_ctx *ctx);

/* ---- S3: posting-list scoring -----------------------------------------------------------------
 * posting-list lengths of query hashes: FolddiscoIndex::get_entries(h).len()
 * (indextable.rs:83-86) — what calculate_idf_for_hash (src/controller/query.rs:17-32) and
 * count_query (src/controller/count_query.rs:121-130) need for log2(S / len). */
int fdgpu_posting_lengths(fdgpu_ctx *ctx, const fdgpu_index *ix, const uint32_t *q_hash, uint64_t n_q,
                          uint64_t *lengths);

/* byte length of the posting lists (get_raw_entries(h).len(), indextable.rs:53-81): what a scoring pass reads — the
 * algorithmic-bytes figure of the query roofline; 0 for an absent hash */
int fdgpu_posting_bytes(fdgpu_ctx *ctx, const fdgpu_index *ix, const uint32_t *q_hash, uint64_t n_q, uint64_t *bytes);

typedef struct fd_count_rec {   /* one touched structure (StructureResult prefilter fields) */
    uint32_t nid;
    uint32_t total_match_count;
    uint32_t node_count;
    uint32_t edge_count;
    float idf;                  /* sum_hits log2(S/len) * nres^(-length_penalty) */
} fd_count_rec;

/* Replaces count_query (src/controller/count_query.rs:82-220), no sampling.
 * q_hash[k] belongs to query edge (q_node[k], q_edge_j[k]) (= first / second query residue).
 * q_idf[k] = log2(S / len(posting(q_hash[k]))) as computed by the caller (glibc log2f keeps the
 * reference's bits); hashes failing freq_filter must be dropped by the caller.
 * penalty[nid] = (nres[nid] as f32).powf(-lp) (count_query.rs:200), n_structures entries.
 * Results: touched structures in ascending nid. */
int fdgpu_count_query(fdgpu_ctx *ctx, const fdgpu_index *ix, const uint32_t *q_hash, const uint32_t *q_node,
                      const uint32_t *q_edge_j, const float *q_idf, uint64_t n_q, const float *penalty,
                      fd_count_rec **out, uint64_t *n_out);

/* get_entries (src/index/indextable.rs:83-86; varint decode :421-463) for n_q hashes at once: the structure ids of hash k are
 * (*ids)[(*ids_off)[k] .. (*ids_off)[k+1]) in ascending order (empty for an absent hash).  Release both with fdgpu_free. */
int fdgpu_get_entries(fdgpu_ctx *ctx, const fdgpu_index *ix, const uint32_t *q_hash, uint64_t n_q, uint32_t **ids, uint64_t **ids_off);

/* Batched form: n_queries queries scored in one set of launches.  Query t owns entries [q_off[t], q_off[t+1]) of the
 * concatenated q_* arrays; its results are (*out)[(*out_off)[t] .. (*out_off)[t+1]) (ascending nid). */
int fdgpu_count_query_batch(fdgpu_ctx *ctx, const fdgpu_index *ix, uint64_t n_queries, const uint64_t *q_off,
                            const uint32_t *q_hash, const uint32_t *q_node, const uint32_t *q_edge_j, const float *q_idf,
                            const float *penalty, fd_count_rec **out, uint64_t **out_off);

/* As above with the candidate selection of query_pdb.rs:404-411 done on the device: per query only the top_n records come back,
 * ranked as the reference ranks them (idf descending, ties by ascending nid) — radix selection of the cut-off, then a bitonic sort in
 * LDS, so top_n records per query cross the bus.  top_n = 0: everything, in ascending nid. */
int fdgpu_count_query_batch_top(fdgpu_ctx *ctx, const fdgpu_index *ix, uint64_t n_queries, const uint64_t *q_off,
                                const uint32_t *q_hash, const uint32_t *q_node, const uint32_t *q_edge_j, const float *q_idf,
                                const float *penalty, uint32_t top_n, fd_count_rec **out, uint64_t **out_off);
/* The length penalty nres^(-lp) of the index's structures (count_query.rs:200) kept on the device: fdgpu_count_query* may then be
 * called with penalty = NULL instead of uploading n_structures floats per call.  penalty = NULL drops the resident copy. */
int fdgpu_index_set_penalty(fdgpu_ctx *ctx, fdgpu_index *ix, const float *penalty);
/* count_query for the query maps fdgpu_make_query_map[_batch] returned, without a round trip through the caller: every entry
 * (hash, (qi, qj)) of every map is scored with idf = log2f(total_structures / posting length of the hash itself)
 * (count_query.rs:181-200; the idf stored in the map belongs to the pair's observed hash and feeds the subgraph idf of the retrieval),
 * hashes the index does not hold are dropped.  Output as fdgpu_count_query_batch_top (top_n = 0: everything, ascending nid).
 * fd_query_map is declared further down. */
struct fd_query_map;
int fdgpu_count_query_maps_top(fdgpu_ctx *ctx, const fdgpu_index *ix, uint64_t n_queries, const struct fd_query_map *const *qms,
                               const float *penalty, float total_structures, uint32_t top_n, fd_count_rec **out, uint64_t **out_off);


/* ---- S4: candidate matching + RMSD --------------------------------------------------------------
 * Pair scan of retrieve_with_prefilter (src/controller/retrieve.rs:52-156) over candidate
 * structures of a resident batch.  For candidate c = cand[k] every ordered residue pair (i,j)
 * that passes the CA cutoff, the amino-acid/CA-distance window test against the query's
 * observed (aa_i, aa_j, dist, qi) list and has a feature is a "candidate pair"; those whose
 * hash is in the (sorted) query hash set are "found" triples. */
typedef struct fd_pair_rec { uint32_t cand; uint32_t i; uint32_t j; uint32_t hash; } fd_pair_rec;
typedef struct fd_cand_rec { uint32_t cand; uint32_t qi; uint32_t i; uint32_t j; } fd_cand_rec;
typedef struct fd_match_query {
    const uint32_t *hashes;     /* sorted unique query hashes */
    uint64_t n_hashes;
    const uint8_t *aad_aa1, *aad_aa2;   /* observed_distance_map flattened (query.rs:271-281) */
    const float *aad_dist;
    const uint32_t *aad_qi;
    uint64_t n_aad;
    float ca_distance_cutoff;   /* --ca-distance, default 1.0 */
    int use_aa_prefilter;       /* prefilter_amino_acid active (<= 200 query hashes, retrieve.rs:569) */
} fd_match_query;
/* resname_std[r] = 1 if the residue's 3-letter name equals map_u8_to_aa(aa) (retrieve.rs:580-586);
 * needed only when use_aa_prefilter. Pass NULL to treat every residue with aa < 20 as standard. */
int fdgpu_match_pairs(fdgpu_ctx *ctx, const fdgpu_batch *db, const uint8_t *resname_std,
                      const uint32_t *cand, uint64_t n_cand, const fd_match_query *q, const fd_hash_params *p,
                      fd_pair_rec **found, uint64_t *n_found, fd_cand_rec **cands, uint64_t *n_cands);

/* Batched Kabsch (src/structure/kabsch.rs:157-554, mode 2): problem k superposes
 * x[off[k]..off[k+1]) (moving = target points) onto y[...] (fixed = query points);
 * points are xyz f32.  rmsd[k] f32, rot[9k..], tran[3k..]. */
int fdgpu_kabsch_batch(fdgpu_ctx *ctx, const float *x, const float *y, const uint64_t *off, uint64_t n_problems,
                       float *rmsd, float *rot, float *tran);
/* Similarity metrics of n superpositions on the device (src/structure/metrics.rs:62-251 applied to KabschSuperimposer's reference /
 * transformed coordinates, kabsch.rs:86-95,145-154): problem k compares the fixed points ref[off[k] .. off[k+1]) with
 * rot[9k ..] * mov[...] + tran[3k ..] (f32, as matrix_vector_multiply + add_vec do).  metrics[5k ..] = tm_score, gdt_ts, gdt_ha,
 * chamfer_distance, hausdorff_distance — what --tm-score / --gdt-ts / --gdt-ha / --chamfer / --hausdorff filter and sort on. */
int fdgpu_metrics_batch(fdgpu_ctx *ctx, const float *ref, const float *mov, const uint64_t *off, uint64_t n_problems, const float *rot,
                        const float *tran, float *metrics);
/* Batched partial fit = LmsQcpSuperimposer::new() + set_atoms(fixed = y, moving = x) + run() with the default parameters
 * (src/structure/lms_qcp.rs:29-41, 91-249; what --partial-fit selects for matches of more than 3 residues,
 * src/controller/retrieve.rs:733-746): rmsd[k] = rms over the final core, rot/tran map x onto y.  Every problem needs
 * >= 3 pairs.  Optional: core_len[k] = size of the core, core[off[k] .. off[k] + core_len[k]) = its pair indices in
 * joining order. */
int fdgpu_lms_qcp_batch(fdgpu_ctx *ctx, const float *x, const float *y, const uint64_t *off, uint64_t n_problems,
                        float *rmsd, float *rot, float *tran, uint32_t *core_len, uint32_t *core);

/* ---- query map and retrieval (host glue in C++, numerics on the GPU) ------------------------------------
 * get_single_feature (src/controller/feature.rs:11-24, 84-99) for explicit residue pairs (i, j) of
 * structure s of a batch: features[k] = {aa_i, aa_j, d_CA, d_CB, theta, tau1, tau2}, valid[k] = has a feature. */
int fdgpu_pair_features(fdgpu_ctx *ctx, const fdgpu_batch *b, uint64_t s, const uint32_t *pair_i, const uint32_t *pair_j,
                        uint64_t n, const fd_hash_params *p, float *features /*[n][7]*/, uint8_t *valid);
/* GeometricHash::perfect_hash (src/geometry/pdb_tr.rs:21-75) on explicit 7-float feature vectors */
int fdgpu_hash_features(fdgpu_ctx *ctx, const float *features, uint64_t n, const fd_hash_params *p, uint32_t *hashes);

typedef struct fd_query_map {   /* make_query_map output (src/controller/query.rs:208-329), first-insertion order */
    uint64_t n;
    uint32_t *hash; uint32_t *qi; uint32_t *qj; uint8_t *is_primary; float *idf;
    uint64_t n_indices; uint32_t *indices;                 /* all_query_indices */
    uint64_t n_aad; uint8_t *aad_aa1; uint8_t *aad_aa2; float *aad_dist; uint32_t *aad_qi;   /* observed_distance_map */
    uint32_t *primary_hash;     /* [n] observed hash of the residue pair entry k belongs to: idf[k] = log2(S / posting length of it)
                                 * — lets a caller that shards the index recompute idf from GLOBAL posting lengths */
    /* set by the library when the map was made against an index (private to it, may be NULL): the posting length and the number of
     * 2 KB scoring segments of every entry's OWN hash in that index — fdgpu_count_query_maps_top then scores without a second
     * posting-length pass and without asking the device for its work count */
    uint64_t *post_len; uint32_t *post_seg; uint64_t post_index_uid;
    long long *post_kidx;       /* with post_len: position of every entry's hash in that index's hash array (-1: absent) — scoring skips its own search */
    uint64_t arena_bytes;       /* private: non-zero when the struct and every array above are ONE allocation of that size (release only through fdgpu_query_map_free) */
} fd_query_map;
/* qb = batch holding the query structure as structure 0; q_index[k] = residue index of the k-th query
 * residue (after parse_query_string + get_index / --serial-index resolution on the caller's side);
 * subs[k] = amino-acid substitution list of residue k or NULL; thresholds as given to -d / -a (angles in
 * degrees); index may be NULL (idf 0). */
int fdgpu_make_query_map(fdgpu_ctx *ctx, const fdgpu_batch *qb, const uint32_t *q_index, uint64_t n_q,
                         const uint8_t *const *subs, const uint32_t *n_subs, const float *dist_thr, uint64_t n_dist,
                         const float *angle_thr_deg, uint64_t n_angle, const fd_hash_params *p, const fdgpu_index *index,
                         float total_structures, fd_query_map **out);
/* Many queries, three launches in total: query t = structure q_struct[t] of qb with residues q_index[q_off[t] .. q_off[t+1]);
 * subs / n_subs run parallel to q_index (may be NULL); out[t] per query, each released with fdgpu_query_map_free. */
int fdgpu_make_query_map_batch(fdgpu_ctx *ctx, const fdgpu_batch *qb, uint64_t n_queries, const uint32_t *q_struct, const uint64_t *q_off,
                               const uint32_t *q_index, const uint8_t *const *subs, const uint32_t *n_subs, const float *dist_thr,
                               uint64_t n_dist, const float *angle_thr_deg, uint64_t n_angle, const fd_hash_params *p,
                               const fdgpu_index *index, float total_structures, fd_query_map **out);
void fdgpu_query_map_free(fd_query_map *m);

typedef struct fd_match_rec {   /* one connected component of one candidate (retrieval_wrapper, retrieve.rs:364-552) */
    uint32_t cand;              /* slot in the candidate list */
    uint32_t same;              /* processed mapping == from-hash mapping */
    float idf;                  /* subgraph idf */
    float rmsd;                 /* of the processed mapping (what the per-match output prints) */
    float rmsd_from_hash;
    float rot[9], tran[3];      /* target -> query superposition of the processed mapping */
    float metrics[5];           /* tm_score, gdt_ts, gdt_ha, chamfer, hausdorff of that superposition (structure/metrics.rs) */
    float rot_from_hash[9], tran_from_hash[3], metrics_from_hash[5];   /* the same for the from-hash mapping (--skip-ca-match prints it) */
} fd_match_rec;
/* residues: 2 * n_indices int32 per match — target residue index (relative to its structure, -1 = "_") for
 * every query residue, first the from-hash mapping then the processed (rescued) one. Release both with
 * fdgpu_matches_free. Matches are ordered by candidate slot, then by component as in graph.rs:43-45.
 * partial_fit != 0 = --partial-fit: mappings of more than 3 residues are superposed by the least-median-of-squares
 * fit (fdgpu_lms_qcp_batch) and report the rms over its core (retrieve.rs:733-746, 787-812). */
int fdgpu_retrieve(fdgpu_ctx *ctx, const fdgpu_batch *db, const uint8_t *resname_std, const uint32_t *cand, uint64_t n_cand,
                   const fd_query_map *qm, const fdgpu_batch *qb, const fd_hash_params *p, float ca_distance_cutoff,
                   uint32_t node_count, uint32_t partial_fit, fd_match_rec **matches, uint64_t *n_matches, int32_t **residues);
/* Many queries with one pair scan, one coordinate gather and one Kabsch launch in total: query t = structure q_struct[t] of qb
 * with the query map qms[t]; its candidates are cand[cand_off[t] .. cand_off[t+1]).  Matches of query t =
 * (*matches)[(*match_off)[t] .. (*match_off)[t+1]) (cand = slot inside the query's own list), its residues start at
 * (*residues)[(*res_off)[t]], 2 * qms[t]->n_indices per match.  Release matches/residues with fdgpu_matches_free, the two
 * offset arrays with fdgpu_free. */
int fdgpu_retrieve_batch(fdgpu_ctx *ctx, const fdgpu_batch *db, const uint8_t *resname_std, uint64_t n_queries, const uint32_t *cand,
                         const uint64_t *cand_off, const fd_query_map *const *qms, const fdgpu_batch *qb, const uint32_t *q_struct,
                         const fd_hash_params *p, float ca_distance_cutoff, uint32_t node_count, uint32_t partial_fit,
                         fd_match_rec **matches, uint64_t **match_off, int32_t **residues, uint64_t **res_off);
void fdgpu_matches_free(fd_match_rec *m, int32_t *residues);
/* The per-query body of the query workflow (src/cli/workflows/query_pdb.rs:376-452: make_query_map -> count_query -> sort / truncate ->
 * retrieval_wrapper over the first candidates) for a rayon chunk of queries in ONE call: arguments as fdgpu_make_query_map_batch (queries),
 * fdgpu_count_query_maps_top (penalty NULL = the index's resident one, top_n ranked records per query) and fdgpu_retrieve_batch over the first
 * match_top records of every query's ranking (candidates = nid - the index's first id: db holds the index's structures in order).  Outputs
 * are exactly those of the three calls made one after the other: maps[t] (caller's array of n_queries pointers, each released with
 * fdgpu_query_map_free), recs / rec_off, matches / match_off / residues / res_off (fdgpu_free).  What the single call adds is overlap: the
 * retrieval's tables are built while the scoring kernels run and the ranked records cross the bus while the retrieval runs. */
int fdgpu_query_batch(fdgpu_ctx *ctx, const fdgpu_index *index, const fdgpu_batch *db, const uint8_t *resname_std, const fdgpu_batch *qb, uint64_t n_queries,
                      const uint32_t *q_struct, const uint64_t *q_off, const uint32_t *q_index, const uint8_t *const *subs, const uint32_t *n_subs,
                      const float *dist_thr, uint64_t n_dist, const float *angle_thr_deg, uint64_t n_angle, const fd_hash_params *p, float total_structures,
                      const float *penalty, uint32_t top_n, uint32_t match_top, float ca_distance_cutoff, uint32_t node_count, fd_query_map **maps,
                      fd_count_rec **recs, uint64_t **rec_off, fd_match_rec **matches, uint64_t **match_off, int32_t **residues, uint64_t **res_off);
/* The same call, NON-BLOCKING: submit hands the batch to one of the context's query LANES and returns, wait collects it.  The reference keeps
 * many queries in flight from its rayon workers (src/cli/workflows/query_pdb.rs:348 `queries.into_par_iter()`, :415 the retrieval's
 * par_iter_mut); a host with ONE thread per GPU gets that overlap here: submit batches k, k + 1, k + 2, then wait for k, submit k + 3, ... — a
 * lane is a private sibling context (own HIP stream, workspaces, landing blocks) driven by a library thread through fdgpu_query_batch itself, so
 * one batch's host-side steps (table building, waits for counts, result copies) are covered by the other lanes' kernels, and the results are bit
 * for bit those of the blocking call.  Lanes are made on the first submit (default 6, env FDGPU_QUERY_LANES, or fdgpu_query_lanes beforehand;
 * each lane holds its own query scratch in HBM — ~1 GB for batches of 128 queries at 542,000 structures) and are torn down by fdgpu_destroy.
 * Inputs: the per-query arrays (q_struct, q_off, q_index, subs, n_subs, thresholds, *p) are COPIED at submit; index, db, qb, resname_std and
 * penalty are borrowed until the wait returns.  Every submitted job must be waited for exactly once (that releases it); passing NULL output
 * pointers to wait discards the results.  Errors of the batch are returned by wait (message in fdgpu_last_error(ctx)).  Submit and wait may
 * be called from different threads; jobs complete in any order. */
typedef struct fdgpu_query_job fdgpu_query_job;
int fdgpu_query_batch_submit(fdgpu_ctx *ctx, const fdgpu_index *index, const fdgpu_batch *db, const uint8_t *resname_std, const fdgpu_batch *qb, uint64_t n_queries,
                             const uint32_t *q_struct, const uint64_t *q_off, const uint32_t *q_index, const uint8_t *const *subs, const uint32_t *n_subs,
                             const float *dist_thr, uint64_t n_dist, const float *angle_thr_deg, uint64_t n_angle, const fd_hash_params *p, float total_structures,
                             const float *penalty, uint32_t top_n, uint32_t match_top, float ca_distance_cutoff, uint32_t node_count, fdgpu_query_job **job);
int fdgpu_query_batch_wait(fdgpu_ctx *ctx, fdgpu_query_job *job, fd_query_map **maps, fd_count_rec **recs, uint64_t **rec_off, fd_match_rec **matches,
                           uint64_t **match_off, int32_t **residues, uint64_t **res_off);
/* make sure the context has at least n_lanes lanes (1..8) -> the number it has, or a negative error code; n_lanes = 0 only reports */
int fdgpu_query_lanes(fdgpu_ctx *ctx, uint32_t n_lanes);
/* Result arrays of the query entry points come from a recycling pool (fdgpu_free puts them back; at most 256 MB of idle blocks are kept,
 * env FDGPU_OUT_POOL_MB overrides, 0 = keep nothing).  fdgpu_trim releases the idle blocks — page-locked ones included — e.g. after a burst of
 * large batches; the last fdgpu_destroy of the process does it too. */
void fdgpu_trim(void);

/* ---- multi-GPU query path (one process per GPU, RCCL over xGMI) ----------------------------------------------------
 * The index and the coordinates are sharded by structure id (every rank holds the postings and coordinates of its own id range,
 * fdgpu_index_build with first_id or fdgpu_index_load + fdgpu_index_set_first_id).  What the reference's query workflow
 * (src/cli/workflows/query_pdb.rs:376-452) would call instead of the single-index count_query / retrieval: rank 0 creates a unique id and
 * hands it to the other ranks by any means (file, MPI, env), every rank calls fdgpu_comm_init on its own context / GPU, then per batch of
 * queries fdgpu_sharded_count_query[_maps] and fdgpu_sharded_retrieve.  RCCL is bound at run time (dlopen); without it these calls return
 * FDGPU_EHIP.  The collectives are issued for every world size, one included.  A rank whose LOCAL STEP fails (posting lengths, scoring,
 * retrieval) still takes part in the call's collectives (zeros in the sum, its status in the gathered message) and every rank returns an
 * error: no rank is left waiting.  The exchange buffers of ordinary calls are reserved at fdgpu_comm_init.  A rank that cannot grow them
 * beyond that reserve, or whose HIP runtime fails between two collectives of a call, aborts the communicator (ncclCommAbort) and returns
 * FDGPU_EHIP: the communicator is then unusable on every rank (later calls fail at once) and must be created anew. */
#define FDGPU_COMM_ID_BYTES 128
typedef struct fdgpu_comm fdgpu_comm;
int fdgpu_comm_unique_id(uint8_t id[FDGPU_COMM_ID_BYTES]);
int fdgpu_comm_init(fdgpu_ctx *ctx, const uint8_t id[FDGPU_COMM_ID_BYTES], int rank, int world, fdgpu_comm **out);
void fdgpu_comm_destroy(fdgpu_comm *comm);
int fdgpu_comm_rank(const fdgpu_comm *comm);
int fdgpu_comm_world(const fdgpu_comm *comm);
/* collectives this communicator has issued so far (ncclAllReduce / ncclAllGather calls) */
int fdgpu_comm_stats(const fdgpu_comm *comm, uint64_t *n_allreduce, uint64_t *n_allgather);

/* ---- ONE on-disk index from N ranks, without the host (Option A) -----------------------------------------------
 * After a build sharded by structure every rank holds the resident sub-index of its id range; the reference's output contract is one
 * PREFIX / PREFIX.offset in ascending hash order (src/index/indextable.rs:239-326).  The hash space is cut into N ranges of about equal
 * posting bytes, every rank slices its sub-index at those bounds, piece j travels to rank j device to device, rank j concatenates the N
 * pieces of its range per hash on the device (fdgpu_index_merge: id ranges ascend with the source rank) and writes its regions of the two
 * files — byte-identical to the single-GPU build, because lists are (hash ascending, id ascending) either way.
 *   fdgpu_index_range_bounds   n_ranges - 1 ascending hash values cutting `index` into ranges of about equal posting bytes (range j holds
 *                              bounds[j - 1] <= hash < bounds[j]; one rank computes them, all ranks must use the same)
 *   fdgpu_index_slice          the lists with hash_lo <= hash < hash_hi (hash_hi up to 2^32) as a resident index of their own (same id range)
 *   fdgpu_comm_single_index    the whole exchange over RCCL (bounds from rank 0, slices, ncclSend / ncclRecv of piece j to rank j, merge):
 *                              -> this rank's hash range of the database's index and its place in it
 *   fdgpu_index_save_part      writes a range's regions of PREFIX / PREFIX.offset (see fd_shard_index.hip); ranks of one node may call it
 *                              concurrently; write_header != 0 on the rank of the first range, is_last != 0 on the rank of the last one
 * Hosts with another transport (MPI, gloo) use range_bounds + slice + fdgpu_index_export / _load + fdgpu_index_merge + save_part. */
int fdgpu_index_range_bounds(fdgpu_ctx *ctx, const fdgpu_index *index, uint32_t n_ranges, uint32_t *bounds);
int fdgpu_index_slice(fdgpu_ctx *ctx, const fdgpu_index *index, uint64_t hash_lo, uint64_t hash_hi, fdgpu_index **out);
int fdgpu_comm_single_index(fdgpu_ctx *ctx, fdgpu_comm *comm, const fdgpu_index *local, fdgpu_index **range_index, uint64_t *hashes_before,
                            uint64_t *value_before, uint64_t *total_hashes, uint64_t *total_value);
int fdgpu_index_save_part(fdgpu_ctx *ctx, const fdgpu_index *part, const char *prefix, uint64_t hashes_before, uint64_t value_before,
                          uint64_t total_hashes, uint64_t total_value, int write_header, int is_last);
/* lengths[k] <- sum over the ranks (ncclAllReduce): posting lengths of a shard -> posting lengths over the whole database, the
 * denominator of idf = log2(S / len) (src/controller/query.rs:17-32, count_query.rs:130) */
int fdgpu_allreduce_lengths(fdgpu_ctx *ctx, fdgpu_comm *comm, uint64_t *lengths, uint64_t n);
/* count_query of a batch of queries against the sharded index: every rank passes the same queries (layout of
 * fdgpu_count_query_batch, no idf: it is computed here from the posting lengths all-reduced on the device, with log2f) and its own shard +
 * penalty (one entry per structure of the shard; NULL = the resident copy of fdgpu_index_set_penalty).  The ranks score locally, select
 * their top_n on the device, all-gather selection state + ranked records in one device-to-device ncclAllGather and select / rank the union
 * on the device: per query (*out)[(*out_off)[t] .. (*out_off)[t+1]) = the global ranking (idf descending, nid ascending;
 * query_pdb.rs:404-411) truncated to top_n, identical on every rank.  nid = global structure id.  top_n = 0 (every touched structure)
 * and top_n > 3072 exchange variable-length lists instead (counts, then one padded payload) and rank on the host. */
int fdgpu_sharded_count_query(fdgpu_ctx *ctx, fdgpu_comm *comm, const fdgpu_index *ix, uint64_t n_queries, const uint64_t *q_off,
                              const uint32_t *q_hash, const uint32_t *q_node, const uint32_t *q_edge_j, const float *penalty,
                              uint64_t total_structures, uint32_t top_n, fd_count_rec **out, uint64_t **out_off);
/* The same for the query maps of fdgpu_make_query_map[_batch] (made with index = NULL): the sharded sibling of
 * fdgpu_count_query_maps_top.  One all-reduce carries the posting lengths of every map's hash[] (scoring idf) and primary_hash[]; the
 * maps' idf[] — the retrieval's subgraph idf (query.rs:283-288) — is REWRITTEN in place from the global lengths, so the maps can go on to
 * fdgpu_sharded_retrieve. */
int fdgpu_sharded_count_query_maps(fdgpu_ctx *ctx, fdgpu_comm *comm, const fdgpu_index *ix, uint64_t n_queries, fd_query_map *const *qms,
                                   const float *penalty, uint64_t total_structures, uint32_t top_n, fd_count_rec **out, uint64_t **out_off);
/* Retrieval against sharded coordinates (the candidate loop of query_pdb.rs:415-452): query t's candidates
 * cand_nid[cand_off[t] .. cand_off[t+1]) are GLOBAL structure ids in ranking order, the same on every rank; `db` holds the structures
 * first_id .. first_id + n - 1 of this rank.  Every candidate is matched on the rank that owns it, the match records and residue lists
 * are all-gathered (counts, then one padded payload) and merged by candidate slot: the output equals fdgpu_retrieve_batch over the whole
 * database with cand = the global lists (fd_match_rec.cand = slot in the query's list), identical on every rank.  Release like
 * fdgpu_retrieve_batch's outputs. */
int fdgpu_sharded_retrieve(fdgpu_ctx *ctx, fdgpu_comm *comm, const fdgpu_batch *db, uint64_t first_id, const uint8_t *resname_std,
                           uint64_t n_queries, const uint32_t *cand_nid, const uint64_t *cand_off, const fd_query_map *const *qms,
                           const fdgpu_batch *qb, const uint32_t *q_struct, const fd_hash_params *p, float ca_distance_cutoff,
                           uint32_t node_count, uint32_t partial_fit, fd_match_rec **matches, uint64_t **match_off, int32_t **residues,
                           uint64_t **res_off);
/* For hosts with their own transport (MPI, gloo): the two local halves of fdgpu_sharded_count_query_maps.  fdgpu_query_maps_lengths
 * writes the LOCAL posting lengths of all maps' hash[] (sum(n) values, maps in order) followed by all maps' primary_hash[] (sum(n) more);
 * the caller sums the 2 * sum(n) values over the ranks and hands them to fdgpu_count_query_maps_top_global, which rewrites the maps' idf[]
 * and scores the local shard (output as fdgpu_count_query_maps_top: this rank's top_n, ranked). */
int fdgpu_query_maps_lengths(fdgpu_ctx *ctx, const fdgpu_index *ix, uint64_t n_queries, const fd_query_map *const *qms, uint64_t *lengths);
int fdgpu_count_query_maps_top_global(fdgpu_ctx *ctx, const fdgpu_index *ix, uint64_t n_queries, fd_query_map *const *qms,
                                      const uint64_t *global_lengths, const float *penalty, float total_structures, uint32_t top_n,
                                      fd_count_rec **out, uint64_t **out_off);
/* The per-rank message of the device exchange: {u32 status, n_queries, top_n, cap} | {u32 x3, u32 count}[n_queries] |
 * fd_count_rec[n_queries][top_n], padded to 16 bytes (include/fdgpu_debug.h: the post-gather merges on hand-made messages).  0 < top_n <= 3072. */
uint64_t fdgpu_comm_message_bytes(uint64_t n_queries, uint32_t top_n);

/* ---- structure ingest (host, multi-threaded) ---------------------------------------------------------------
 * PDB / mmCIF text (optionally gzip) -> the packed arrays of fd_batch_desc plus what the .lookup file and the result
 * printer need.  Replaces read_structure_from_path + CompactStructure::build inside the index / query workflows
 * (src/controller/io.rs, src/structure/io/pdb.rs:37-77, src/structure/io/cif.rs:102-296, src/structure/core.rs:70-214,
 * quirks kept).  A structure with more than max_residue residues (0 = no limit) keeps its slot but has no residues and
 * pLDDT 0 (src/controller/mod.rs:313-318); an unreadable file has ok = 0 and no residues.  Format by file name:
 * *.cif / *.mmcif (+ .gz) = mmCIF, everything else = PDB.  Release with fdgpu_parsed_free. */
typedef struct fd_parsed {
    uint64_t n_struct, n_res;
    uint64_t *res_off;               /* [n_struct + 1] */
    float *n_xyz, *ca_xyz, *cb_xyz;  /* [3 * n_res] */
    uint8_t *aa, *cb_valid;          /* [n_res] 0..19 / 255; 1 = CB present (measured or virtual) */
    uint8_t *chain;                  /* [n_res] chain id (of the next residue's first atom, core.rs:116) */
    uint8_t *resname_std;            /* [n_res] 1 = the residue name is the standard 3-letter code of `aa` */
    uint64_t *serial;                /* [n_res] residue number */
    float *bfac;                     /* [n_res] */
    char *resname;                   /* [3 * n_res] */
    uint64_t *nres_raw;              /* [n_struct] residue count of the raw atom list (what max_residue is compared with) */
    float *plddt;                    /* [n_struct] mean b-factor, f32 sequential sum (core.rs:450-456) */
    uint8_t *ok;                     /* [n_struct] */
    uint8_t *first_chain;            /* [n_struct] chain id of the first atom (default chain of a query string) */
} fd_parsed;
int fdgpu_parse_structures(const char *const *paths, uint64_t n, uint32_t n_threads, uint64_t max_residue, fd_parsed **out);
void fdgpu_parsed_free(fd_parsed *p);
/* thread-seconds the ingest spent so far (summed over the threads of all fdgpu_parse_structures calls of the process): out[0] read + inflate,
 * out[1] text -> atom records, out[2] CompactStructure::build, out[3] files, out[4] inflated bytes; reset != 0 clears the counters */
void fdgpu_ingest_stats(double out[5], int reset);

/* Foldcomp input (reference: src/structure/io/fcz.rs — FoldcompDbReader::new :41-74, read_single_structure_by_id :203-230,
 * and the vendored decoder behind foldcomp_process, lib/foldcomp/foldcompffi.cpp).  fdgpu_foldcomp_decode turns one database
 * entry into the atom records the reference's Structure::update receives (coordinates bit-identical to the vendored decoder's:
 * backbone rebuilt from the stored torsion/bond angles forward and backward between anchors, O / CB / side chain placed from
 * them; atoms past CB carry their names but no coordinates — nothing on this path reads them).  fdgpu_foldcomp_db_list gives the
 * entries of DB.index in ascending key order (keys = the db_key column of the index's .lookup output; names from DB.lookup, joined
 * by '\n'; both released with fdgpu_free).  fdgpu_parse_foldcomp_db decodes the entries with the given keys (n_keys = 0: all, in
 * key order) and returns the same packed arrays as fdgpu_parse_structures. */
typedef struct fd_foldcomp_atom {
    float x, y, z, b;                /* coordinate, temperature factor (per-residue in Foldcomp) */
    char name[4];                    /* PDB-style padded atom name (" CA ", " OXT") */
    char res[3];                     /* residue name */
    uint8_t chain;
    uint64_t rser;                   /* residue serial: header.idx_residue + position */
} fd_foldcomp_atom;
int fdgpu_foldcomp_decode(const uint8_t *entry, uint64_t len, fd_foldcomp_atom **atoms, uint64_t *n_atoms);
int fdgpu_foldcomp_db_list(const char *db_path, uint64_t **keys, char **names, uint64_t *n_entries);
int fdgpu_parse_foldcomp_db(const char *db_path, const uint64_t *keys, uint64_t n_keys, uint32_t n_threads, uint64_t max_residue,
                            fd_parsed **out);

/* ---- merging per-GPU / per-batch sub-indices into the reference's single index ----------------------------
 * Parts must cover ascending, disjoint id ranges in the order given (index build shards by structure).  Output is
 * the on-disk layout (value bytes, sparse hashes, offsets[H+1]); host-side, buffers released with fdgpu_free. */
int fdgpu_merge_subindices(uint64_t n_parts, const uint8_t *const *values, const uint32_t *const *hashes,
                           const uint64_t *const *offsets, const uint64_t *n_hashes, uint8_t **out_value,
                           uint64_t *out_value_len, uint32_t **out_hashes, uint64_t **out_offsets, uint64_t *out_n_hashes);

/* The same merge on the device, for parts that are resident: a shard whose sort workspace does not fit one call is built as several
 * fdgpu_index_build calls over consecutive id ranges (the reference walks its input in chunks the same way,
 * src/controller/mod.rs:282-348) and the chunks' posting lists are concatenated per hash — first varint of every continuation
 * re-based to a delta — into ONE resident index, byte-identical to the index a single build over all the structures produces
 * (indextable.rs:171-202 appends ids in ascending order).  At most 64 parts per call; the parts stay valid. */
int fdgpu_index_merge(fdgpu_ctx *ctx, const fdgpu_index *const *parts, uint64_t n_parts, fdgpu_index **out);

/* `analyze -p`: right-tail hypergeometric test of every encoding (src/controller/summary.rs:543-628, get_enriched_hashes /
 * hypergeometric_test with the reference's log-factorial): p_value[k] = P(X >= query_count[k]) for a sample of total_query draws from a
 * population of total_bg + total_query holding bg_count[k] + query_count[k] successes.  Host threads; no device needed. */
int fdgpu_hypergeom_enrichment(const uint64_t *query_count, const uint64_t *bg_count, uint64_t n_encodings, uint64_t total_query, uint64_t total_bg,
                               uint32_t n_threads, double *p_value);

/* Pairs that the speculative torsion evaluation of the index build (fd_geom.h, fd_pair_both_spec) re-evaluated with the
 * exact routine since the previous call; FDGPU_EXACT=1 in the environment disables the speculative path altogether. */
int fdgpu_spec_fallbacks(fdgpu_ctx *ctx, uint64_t *out);

/* ---- profiling hooks ------------------------------------------------------------------------------
 * Per-kernel timing of the last fdgpu_index_build / fdgpu_count_query call, measured with
 * HIP events on the context's stream. names[i] is a static string. Returns the number of
 * entries written (<= cap). */
int fdgpu_last_timings(const fdgpu_ctx *ctx, const char **names, float *ms, uint64_t *bytes, int cap);
int fdgpu_enable_timing(fdgpu_ctx *ctx, int on);

/* ---- PREFIX.lookup (src/index/lookup.rs:35-56; written by build_index.rs:204-215) -------------------------------------------------
 * One line per structure: id \t tid \t nres \t plddt \t db_key \n, plddt printed like Rust's `{}` of an f32 (shortest digits that
 * round-trip, never exponent form, integral values without a fraction, NaN / inf spelled so).  tids = the n ids joined by '\n';
 * db_keys NULL = the id.  fdgpu_format_f32_display writes the n strings NUL-terminated at out + 64 * k. */
int fdgpu_write_lookup(const char *path, const char *tids, uint64_t n, const uint64_t *nres, const float *plddt, const uint64_t *db_keys);
int fdgpu_format_f32_display(const float *v, uint64_t n, char *out);

#ifdef __cplusplus
}
#endif
#endif
