#!/bin/bash
cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out /tmp/mpt
FDGPU_MP_TIMES=/tmp/mpt/t python tools/profile_whole_query.py --structures 67750 2>&1 | grep "traced run\|scan done"
python - <<'PY'
import numpy as np, glob
for f in sorted(glob.glob("/tmp/mpt/t.*.bin"))[-2:]:
    a = np.fromfile(f, np.uint64).reshape(-1, 3)
    t0, t1, hw = a[:, 0].astype(np.int64), a[:, 1].astype(np.int64), a[:, 2]
    ok = (t0 > 0) & (t1 > 0)
    t0, t1, hw = t0[ok], t1[ok], hw[ok]
    base = t0.min(); dur = (t1 - t0) * 0.01  # us (100 MHz)
    span = (t1.max() - base) * 0.01
    print(f, "work items", len(t0), "kernel span %.1f us" % span, "sum of durations %.1f ms" % (dur.sum() / 1e3), "avg concurrency %.0f" % (dur.sum() / span))
    print("  duration us: min %.1f p50 %.1f p90 %.1f p99 %.1f max %.1f" % (dur.min(), np.percentile(dur, 50), np.percentile(dur, 90), np.percentile(dur, 99), dur.max()))
    # concurrency over time in 20 bins
    edges = np.linspace(0, span, 21)
    s0, s1 = (t0 - base) * 0.01, (t1 - base) * 0.01
    conc = [float(np.sum(np.clip(np.minimum(s1, edges[k + 1]) - np.maximum(s0, edges[k]), 0, None)) / (edges[k + 1] - edges[k])) for k in range(20)]
    print("  concurrency per 5% of the span:", [int(x) for x in conc])
    # start times of work items by index (dispatch order)
    idx = np.arange(len(s0))
    print("  start time of work item quantiles (us):", [round(float(np.percentile(s0, q)), 1) for q in (1, 25, 50, 75, 99)])
    xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xf
    cu = ((hw & np.uint64(0xffffffff)) >> np.uint64(8)).astype(np.int64) & 0xf
    se = ((hw & np.uint64(0xffffffff)) >> np.uint64(13)).astype(np.int64) & 0x7
    print("  work items per XCC:", np.bincount(xcc, minlength=8).tolist())
    key = xcc * 1000 + se * 16 + cu
    u, cnt = np.unique(key, return_counts=True)
    print("  distinct (xcc, se, cu):", len(u), "items per CU min/max", cnt.min(), cnt.max())
    # per-CU busy time
    busy = np.array([dur[key == k].sum() for k in u])
    print("  per-CU summed durations us: min %.0f p50 %.0f max %.0f" % (busy.min(), np.median(busy), busy.max()))
PY
