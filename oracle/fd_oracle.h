/* fd_oracle.h — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * CPU restatement (plain C, single-threaded, strict IEEE f32, glibc libm) of the
 * Folddisco hot path that folddisco_amd implements in HIP.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library;
 * the product (folddisco_amd/) never links or imports it.
 *
 * Every function cites the reference file:line (relative to /root/reference) it
 * restates.  Parity status: PINNED by the reference's own literals —
 *   G1  six PDBTrRosetta hashes of the 4CHA triad  (src/controller/graph.rs:71-79)
 *   G2  README.md:216-241 end-to-end rows (index data/serine_peptidases, query 4CHA)
 *   G3  varint/offset known answer from src/index/indextable.rs:471-499
 *   G4  pair order / AA map / query grammar / Kabsch triads (reference unit tests)
 * see tests/test_oracle_golden.py.  The Rust reference itself cannot be built in
 * this image (no cargo/rustc); oracle/_ref holds only the reference's vendored C++
 * Foldcomp decoder (oracle/Makefile target `ref`), which pins the Foldcomp ingest.
 * Parity UNPINNED against reference output (the reference's tests print, they do not
 * assert): the eight non-default encodings, LMS-QCP, the similarity metrics.
 *
 * Float rule: compile with -O2 -ffp-contract=off -fno-fast-math; libm = glibc
 * sinf/cosf/acosf/atan2f (what Rust's f32::sin etc. lower to on linux-gnu).
 */
#ifndef FD_ORACLE_H
#define FD_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- structure (src/structure/core.rs:56-67 CompactStructure) ------------- */
typedef struct fdo_structure {
    int32_t n;                 /* compact residues kept (num_residues) */
    int32_t num_residues_raw;  /* Structure.num_residues (serial changes, core.rs:36-39) */
    int32_t num_atoms;
    float *n_xyz;              /* [3n] interleaved x,y,z */
    float *ca_xyz;             /* [3n] */
    float *cb_xyz;             /* [3n] (undefined where cb_ok == 0) */
    uint8_t *cb_ok;            /* [n]  CB present (real or virtual) */
    uint8_t *resname;          /* [3n] */
    uint8_t *aa;               /* [n]  map_aa_to_u8 (convert.rs:53-81), 255 = unknown */
    uint8_t *chain;            /* [n]  chain_per_residue (quirk: first atom of NEXT residue) */
    uint64_t *serial;          /* [n] */
    float *bfac;               /* [n]  (same quirk) */
    int32_t num_chains;
    uint8_t chains[256];       /* chain ids in order of appearance (Structure.chains) */
} fdo_structure;

fdo_structure *fdo_read_pdb(const char *path);
/* build from an atom table (used by tests to exercise CompactStructure::build quirks) */
fdo_structure *fdo_structure_from_atoms(int32_t natoms, const float *xyz, const uint8_t *atom_name4,
                                        const uint8_t *res_name3, const uint64_t *res_serial,
                                        const uint8_t *chain, const float *bfac);
/* build directly from packed per-residue arrays (synthetic inputs) */
fdo_structure *fdo_structure_from_packed(int32_t n, const float *n_xyz, const float *ca_xyz,
                                         const float *cb_xyz, const uint8_t *cb_ok, const uint8_t *aa,
                                         const float *bfac);
void fdo_structure_free(fdo_structure *s);
float fdo_avg_plddt(const fdo_structure *s);
int64_t fdo_get_index(const fdo_structure *s, uint8_t chain, uint64_t serial);
uint8_t fdo_map_aa_to_u8(const uint8_t aa[3]);
const char *fdo_map_u8_to_aa(uint8_t aa);

/* ---- geometry / hash ----------------------------------------------------------- */
/* feature = [aa_i, aa_j, d_ca, d_cb, theta, tau1, tau2]; returns 1 if the pair has a feature */
int fdo_pair_feature(const fdo_structure *s, int64_t i, int64_t j, float dist_cutoff, float feature[9]);
uint32_t fdo_discretize(float val, float min, float max, float num_bin);
uint32_t fdo_hash_pdbtr(const float feature[9], uint64_t nbin_dist, uint64_t nbin_angle);
/* process-wide encoding switch for the tests (default 3 = PDBTrRosetta; 0, 1, 7, 8 = the other encodings over the same descriptor,
 * geometry/{pdb_motif,pdb_motif_sincos,folddisco_angle,folddisco_dist}.rs); every function below encodes through fdo_hash_any */
int fdo_set_hash_type(uint32_t hash_type);
uint32_t fdo_get_hash_type(void);
uint32_t fdo_hash_any(const float feature[9], uint64_t nbin_dist, uint64_t nbin_angle);
/* --multiple-bins (process-wide test switch; n = 0 turns it off): pairs = {dist, angle} x n.  fdo_hash_structure / fdo_build_index hash
 * every residue pair once per bin pair (controller/feature.rs:211-215), fdo_make_query_map inserts every expansion under every
 * bin pair (query.rs:59-70), fdo_retrieve reports a found triple per matching bin pair (retrieve.rs:124-131). */
int fdo_set_multiple_bins(uint64_t n, const uint64_t *pairs);
uint64_t fdo_multiple_bins(uint64_t out[16]);
uint32_t fdo_hash_cfg(const float feature[9], uint64_t nbin_dist, uint64_t nbin_angle);
void fdo_reverse_hash_pdbtr(uint32_t hash, float out[7]);
int fdo_hash_is_symmetric(uint32_t hash);
/* all ordered pairs row-major (combination.rs:23-44) -> malloc'd list, caller frees with fdo_free */
int fdo_hash_structure(const fdo_structure *s, uint64_t nbin_dist, uint64_t nbin_angle, float dist_cutoff,
                       uint32_t **out, uint64_t *n_out);
uint64_t fdo_sort_dedup_u32(uint32_t *v, uint64_t n);
void fdo_free(void *p);

/* ---- index (src/index/indextable.rs) --------------------------------------------- */
typedef struct fdo_index fdo_index;
fdo_index *fdo_index_new(uint32_t hash_bits);
void fdo_index_count_single_entry(fdo_index *ix, uint32_t hash, uint64_t id);
void fdo_index_allocate_entries(fdo_index *ix);
void fdo_index_add_single_entry(fdo_index *ix, uint32_t hash, uint64_t id);
void fdo_index_finish(fdo_index *ix); /* wrapup_offset + prune_to_sparse */
int fdo_index_save(const fdo_index *ix, const char *prefix); /* PREFIX and PREFIX.offset */
fdo_index *fdo_index_load(const char *prefix);
void fdo_index_free(fdo_index *ix);
uint64_t fdo_index_num_hashes(const fdo_index *ix);
const uint32_t *fdo_index_hashes(const fdo_index *ix);
const uint64_t *fdo_index_offsets(const fdo_index *ix);
const uint8_t *fdo_index_values(const fdo_index *ix);
uint64_t fdo_index_value_len(const fdo_index *ix);
/* get_entries: decoded ids; returns count, *ids malloc'd (fdo_free) */
uint64_t fdo_index_get_entries(const fdo_index *ix, uint32_t hash, uint64_t **ids);
uint64_t fdo_split_by_seven_bits(uint64_t id, uint8_t out[10]);
/* two-pass build over structures exactly like Folddisco::collect_and_count/add_entries
 * (controller/mod.rs:274-441), single-threaded. nres/plddt arrays (len S) are filled. */
fdo_index *fdo_build_index(const fdo_structure *const *structs, uint64_t S, uint64_t nbin_dist,
                           uint64_t nbin_angle, float dist_cutoff, uint64_t max_residue, uint64_t *nres,
                           float *plddt);
/* same, but from already sorted-unique per-structure hash lists (CSR) */
fdo_index *fdo_build_index_from_lists(const uint32_t *hashes, const uint64_t *off, uint64_t S);
/* OpenMP fan-out of the per-structure hash+sort+dedup stage (cpu_baseline only): CSR out, fdo_free both */
int fdo_hash_batch(const fdo_structure *const *structs, uint64_t S, uint64_t nbin_dist, uint64_t nbin_angle,
                   float dist_cutoff, uint32_t **out_hashes, uint64_t **out_off);
fdo_index *fdo_build_index_from_lists_mt(const uint32_t *hashes, const uint64_t *off, uint64_t S, int n_threads);
/* index over borrowed arrays (bench.py's query cpu_baseline) */
fdo_index *fdo_index_borrow(const uint32_t *hashes, const uint64_t *offsets, uint64_t H, const uint8_t *values, uint64_t vlen);
void fdo_index_free_borrowed(fdo_index *ix);
int fdo_save_lookup(const char *path, const char *const *tids, const uint64_t *nres, const float *plddt,
                    const uint64_t *db_key, uint64_t S);
int fdo_save_type(const char *path, uint64_t chunk_size, float grid_width, uint64_t max_residue,
                  uint64_t nbin_angle, uint64_t nbin_dist);
int fdo_format_f32_display(float v, char *buf, size_t cap); /* Rust `{}` for f32 */

/* ---- query (src/controller/query.rs, count_query.rs) ---------------------------- */
typedef struct fdo_query_map {
    uint64_t n;          /* entries in insertion order (first-insert-wins) */
    uint32_t *hash;
    uint64_t *qi, *qj;   /* residue indices in the query structure */
    uint8_t *is_primary;
    float *idf;
    uint64_t n_indices;  /* resolved query residue indices (all_query_indices) */
    uint64_t *indices;
    /* observed_distance_map: flattened ((aa_i,aa_j) -> [(dist, qi)]) in insertion order */
    uint64_t n_aad;
    uint8_t *aad_aa1, *aad_aa2;
    float *aad_dist;
    uint64_t *aad_qi;
} fdo_query_map;

/* parse_query_string (query.rs:331-384). chains/serials/… malloc'd. subs[i] = list of aa or NULL */
typedef struct fdo_query_spec {
    uint64_t n;
    uint8_t *chain;
    uint64_t *serial;
    uint8_t **subs;     /* per residue: NULL or array */
    uint64_t *n_subs;   /* 0 with subs != NULL means Some(empty) */
} fdo_query_spec;
fdo_query_spec *fdo_parse_query_string(const char *q, uint8_t default_chain);
void fdo_query_spec_free(fdo_query_spec *q);

fdo_query_map *fdo_make_query_map(const fdo_structure *qs, const fdo_query_spec *spec, uint64_t nbin_dist,
                                  uint64_t nbin_angle, const float *dist_thr, uint64_t n_dist_thr,
                                  const float *angle_thr, uint64_t n_angle_thr, float dist_cutoff,
                                  int serial_query, const fdo_index *index, float total_structures);
void fdo_query_map_free(fdo_query_map *m);

typedef struct fdo_count_result {
    uint64_t nid;
    uint64_t total_match_count, node_count, edge_count;
    float idf;
} fdo_count_result;
/* count_query (count_query.rs:82-220), no sampling. freq_filter < 0 => None. results nid-ascending. */
uint64_t fdo_count_query(const fdo_query_map *m, const fdo_index *index, const uint64_t *nres, uint64_t S,
                         float freq_filter, float length_penalty, fdo_count_result **out);

/* query half of bench.py's cpu_baseline (fdo_bench.c): OpenMP over queries like query_pdb.rs:348 */
double fdo_query_bench(const fdo_index *ix, const uint64_t *nres, uint64_t S, const uint64_t *res_off, const float *n_xyz, const float *ca_xyz,
                       const float *cb_xyz, const uint8_t *aa, uint64_t n_queries, const uint64_t *q_struct, const uint64_t *q_off,
                       const uint64_t *q_res, uint64_t top_n, uint64_t match_top, int n_threads, uint64_t *n_hits, uint64_t *n_matches,
                       double stage_s[3]);

/* ---- retrieval + RMSD (src/controller/retrieve.rs, graph.rs, structure/kabsch.rs) -- */
typedef struct fdo_match {
    uint64_t n;            /* = number of query residues */
    uint8_t *has;          /* [n] */
    uint8_t *chain;        /* [n] */
    uint64_t *serial;      /* [n] */
    int64_t *tindex;       /* [n] target residue index or -1 */
    float rmsd;
    float idf;             /* subgraph idf */
    float rot[9], tran[3];
} fdo_match;
typedef struct fdo_retrieval {
    uint64_t n_matches;
    fdo_match *from_hash;   /* matching_residues */
    fdo_match *processed;   /* matching_residues_processed */
    uint64_t max_matching_node_count;
    float min_rmsd_with_max_match;
    /* intermediate products exposed for kernel parity tests */
    uint64_t n_found;       /* indices_found (i, j, hash) in scan order */
    uint64_t *found_i, *found_j;
    uint32_t *found_hash;
    uint64_t n_cand;        /* candidate_pairs (qi, (i, j)) */
    uint64_t *cand_qi, *cand_i, *cand_j;
} fdo_retrieval;
fdo_retrieval *fdo_retrieve(const fdo_structure *target, const fdo_structure *query, const fdo_query_map *m,
                            uint64_t node_count, uint64_t nbin_dist, uint64_t nbin_angle, float dist_cutoff,
                            float ca_distance_cutoff);
void fdo_retrieval_free(fdo_retrieval *r);
/* kabsch(x = coords, y = reference, mode) (kabsch.rs:157-554). returns rmsd as f32 */
float fdo_kabsch(const float *x, const float *y, uint64_t n, int mode, float rot[9], float tran[3]);
void fdo_metrics(const float *ref, const float *mov, uint64_t n, const float rot[9], const float tran[3], float out[5]);
/* LmsQcpSuperimposer (structure/lms_qcp.rs:91-249, default parameters): x = coords (moving), y = reference; returns the rms
 * over the final core; core (n slots, may be NULL) receives the core indices in insertion order. */
float fdo_lms_qcp(const float *x, const float *y, uint64_t n, float rot[9], float tran[3], uint64_t *core, uint64_t *n_core);

#ifdef __cplusplus
}
#endif
#endif
