// fd_api_common.h — what the files of the C ABI share (fdgpu_api.hip: context, batches, S1, S2; fd_api_count.hip: S3; fd_api_match.hip: S4): the
// error macros (every entry point returns a code, the message goes to the context) and a few helpers defined in fdgpu_api.hip.
#pragma once
#include "fdgpu_internal.h"

#define HIPCHK(ctx, expr)                                                                                   \
    do {                                                                                                    \
        hipError_t _e = (expr);                                                                             \
        if (_e != hipSuccess) {                                                                             \
            char _b[512];                                                                                   \
            snprintf(_b, sizeof _b, "%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e));  \
            (ctx)->err = _b;                                                                                \
            return FDGPU_EHIP;                                                                              \
        }                                                                                                   \
    } while (0)

#define FAIL(ctx, code, msg) do { (ctx)->err = (msg); return (code); } while (0)

static inline void reset_timings(fdgpu_ctx *c) { c->timings.clear(); c->event_used = 0; }
int d2h_u64(fdgpu_ctx *c, const uint64_t *dev, uint64_t *host);
fd_hash_consts make_consts(const fd_hash_params *p);
fd_hash_consts make_consts_bins(const fd_hash_params *p, uint32_t nbd_req, uint32_t nba_req, bool either_zero_defaults);
bool fd_hash_type_supported(uint32_t t);
uint32_t fd_num_bin_configs(const fd_hash_params *p);
fd_hash_consts fd_make_consts_cfg(const fd_hash_params *p, uint32_t k);
bool fd_multiple_bins_valid(const fd_hash_params *p);
int sort_pairs(fdgpu_ctx *c, uint32_t *ka, uint32_t *va, uint32_t *kb, uint32_t *vb, uint64_t n, int key_bits);      // stable pair sort in the context's sort workspaces
