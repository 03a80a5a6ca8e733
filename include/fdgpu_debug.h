/* fdgpu_debug.h — test-only entry points of libfdgpu.so.  Not part of the drop-in boundary (include/fdgpu.h): a host binding does not need
 * them; the parity tests do (tests/ include both).  Same conventions: error codes, library-allocated outputs released by fdgpu_free. */
#ifndef FDGPU_DEBUG_H
#define FDGPU_DEBUG_H
#include "fdgpu.h"
#ifdef __cplusplus
extern "C" {
#endif

/* What every rank runs after the all-gather of fdgpu_sharded_count_query_maps — unpack, global selection, ranking, all on the device — on `world`
 * messages (format: fdgpu_comm_message_bytes in fdgpu.h) given as one host array: tests drive the multi-rank code with it on a single GPU.
 * 0 < top_n <= 3072. */
int fdgpu_debug_merge_gathered(fdgpu_ctx *ctx, uint32_t world, uint64_t n_queries, uint32_t top_n, const uint8_t *messages,
                               fd_count_rec **out, uint64_t **out_off);
/* The unpack + merge step of fdgpu_sharded_retrieve alone, on hand-made contributions of `world` ranks (host arrays): counts[r * (n_queries + 1) + t]
 * = matches rank r found for query t, counts[r * (n_queries + 1) + n_queries] = its status (0 = fine); rank_matches[r] / rank_residues[r] = its
 * records (cand = slot in the query's GLOBAL candidate list) and residue ints in (query, slot, component) order; nres_per[t] = residue ints per
 * match of query t (2 * n_indices).  Output as fdgpu_retrieve_batch.  Lets a single-GPU test drive the multi-rank merge with ragged, empty and
 * failing ranks. */
int fdgpu_debug_merge_retrieved(fdgpu_ctx *ctx, uint32_t world, uint64_t n_queries, const uint64_t *counts, const fd_match_rec *const *rank_matches,
                                const int32_t *const *rank_residues, const uint64_t *nres_per, fd_match_rec **matches, uint64_t **match_off,
                                int32_t **residues, uint64_t **res_off);
/* The ingest's own gzip decoder (csrc/fd_inflate.cpp; the reference reads .gz through the flate2 crate, src/structure/io/pdb.rs:79-124) on a
 * buffer: every member of in[0 .. n) concatenated into *out (fdgpu_free).  FDGPU_EINVAL = the decoder declines the input (damaged, or a code it
 * does not handle); fdgpu_parse_structures then reads that file through zlib.  FDGPU_ZLIB=1 in the environment sends every file there. */
int fdgpu_debug_gunzip(const uint8_t *in, uint64_t n, uint8_t **out, uint64_t *n_out);
/* Evaluates the device restatements of glibc's sinf / cosf / acosf / atanf / atan2f (csrc/fd_libm.h) on arrays, so that tests can compare the
 * gfx950 arithmetic bit for bit with the host.  op: 0 sinf, 1 cosf, 2 acosf, 3 atanf, 4 atan2f(a, b). */
int fdgpu_debug_libm(fdgpu_ctx *ctx, int op, const float *a, const float *b, float *out, uint64_t n);

/* Host-only pieces of the retrieval glue for graphs of more than 64 nodes (csrc/fd_host_query.hip), callable without a GPU.
 * fdgpu_debug_host_components: the graph of the found residue pairs (edge_i[e] -> edge_j[e], nodes numbered by first appearance,
 * src/controller/graph.rs:16-26) and its strongly + weakly connected components of at least node_count nodes, each sorted by node, the list sorted
 * and without duplicates (graph.rs:29-50).  Component k holds (*residues)[(*comp_off)[k] .. (*comp_off)[k + 1]) — the RESIDUES of its nodes in node
 * order; both arrays are released with libc free().
 * fdgpu_debug_hash_is_symmetric: out[k] = the symmetry flag of hashes[k] for the encoding (geometry/pdb_tr.rs:158-162 and the other encodings'
 * is_symmetric). */
int fdgpu_debug_host_components(const uint32_t *edge_i, const uint32_t *edge_j, uint64_t n_edges, uint32_t node_count, uint32_t **residues,
                                uint64_t **comp_off, uint64_t *n_comps);
int fdgpu_debug_hash_is_symmetric(uint32_t hash_type, const uint32_t *hashes, uint64_t n, uint8_t *out);

#ifdef __cplusplus
}
#endif
#endif
