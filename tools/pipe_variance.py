"""Run-to-run spread of the pipelined query leg (6 lanes x 10 batches of 128 in flight, one host thread) inside ONE process, and what the host
threads' placement does to it: the same timed loop N times with the process free to run anywhere, then confined to the CPUs of each NUMA node
(lane worker threads are created after the affinity is set, so they inherit it).

    python tools/pipe_variance.py [--structures 542000] [--runs 10] [--reps 48]
"""
import argparse
import glob
import os
import sys
import time
from collections import deque

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def cpulist(s):
    out = []
    for part in s.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--structures", type=int, default=542000)
    ap.add_argument("--runs", type=int, default=10)
    ap.add_argument("--reps", type=int, default=48)
    ap.add_argument("--gc-ab", action="store_true", help="the runs twice: Python's cyclic collector on, then off (is the long run the harness?)")
    ap.add_argument("--node", type=int, default=-2, help="-2: the unconfined process only; -1: every NUMA node in turn (one fresh context each); k: node k")
    a = ap.parse_args()
    import numpy as np
    import torch
    from _resident import build_resident
    from folddisco_amd.api import PackedStructures, length_penalty
    from folddisco_amd.query import query_batch_submit
    from folddisco_amd.querybench import _pick_queries
    nodes = {}
    for d in sorted(glob.glob("/sys/devices/system/node/node[0-9]*")):
        nodes[int(d.rsplit("node", 1)[1])] = cpulist(open(d + "/cpulist").read())
    allowed = sorted(os.sched_getaffinity(0))
    print("NUMA nodes: %s; allowed CPUs %d" % ({k: "%d cpus (%d..%d)" % (len(v), v[0], v[-1]) for k, v in nodes.items()}, len(allowed)), flush=True)
    for f in glob.glob("/sys/class/drm/card*/device/numa_node"):
        print(f, open(f).read().strip(), flush=True)
    S = a.structures

    def measure(tag):
        ctx, batch, ix, d, ro = build_resident(S)
        qs = _pick_queries(d, S, 128, 4242)
        qall = ctx.upload(PackedStructures.concat([it for _, _, it in qs]))
        chunk = [(t, qs[t][1]) for t in range(128)]
        ix.set_penalty(length_penalty(np.diff(ro.cpu().numpy()).astype(np.uint64), 0.5))
        assert ctx.L.fdgpu_query_lanes(ctx.h, 6) >= 6

        done_at = []          # when every wait() returned (the gaps between them: is a slow run ONE stall or many small ones?)

        def piped(reps, depth):
            pend, tot = deque(), 0
            for _ in range(reps):
                pend.append(query_batch_submit(ctx, ix, batch, qall, chunk, float(S), 1000, 32))
                if len(pend) >= depth:
                    tot += len(pend.popleft().wait()[2][0]); done_at.append(time.perf_counter())
            while pend:
                tot += len(pend.popleft().wait()[2][0]); done_at.append(time.perf_counter())
            return tot
        piped(12, 6)
        piped(12, 10)
        import gc
        for mode in (("python gc on", True), ("python gc OFF", False)) if a.gc_ab else (("", True),):
            if not mode[1]:
                gc.collect(); gc.disable()
            vals, gaps = [], []
            for _ in range(a.runs):
                torch.cuda.synchronize()
                del done_at[:]
                t0 = time.perf_counter()
                piped(a.reps, 10)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                vals.append(128 * a.reps / dt)
                g = np.diff(np.array([t0] + done_at))
                gaps.append((float(g.max()) * 1e3, float(np.median(g)) * 1e3, int(g.argmax())))
            gc.enable()
            print("%-28s %s  (min %.0f median %.0f max %.0f queries/s) %s" % (tag, " ".join("%.0f" % (v / 1e3) for v in vals), min(vals), sorted(vals)[len(vals) // 2], max(vals), mode[0]), flush=True)
            print("%-28s longest gap between two completed batches per run, ms (median gap, position): %s" % ("", " ".join("%.2f(%.2f,%d)" % x for x in gaps)), flush=True)
        ctx.L.fdgpu_query_lanes(ctx.h, 0)
        del ix, batch, qall
        ctx.close()

    measure("anywhere (%d cpus)" % len(allowed))
    todo = [] if a.node == -2 else (sorted(nodes) if a.node == -1 else [a.node])
    for k in todo:
        cpus = [c for c in nodes[k] if c in allowed]
        if not cpus:
            continue
        os.sched_setaffinity(0, cpus)
        measure("node %d (%d cpus)" % (k, len(cpus)))
        os.sched_setaffinity(0, allowed)


if __name__ == "__main__":
    main()
