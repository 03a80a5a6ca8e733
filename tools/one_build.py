"""One index build call (or --calls N) over S synthetic structures, nothing else: the process rocprofv3 --pmc wraps when a counter has to be read
per LAUNCH (bench.py runs the build in several legs: its counters sum over all of them).

    python tools/one_build.py [--structures 67750] [--calls 1]
"""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--structures", type=int, default=67750)
    ap.add_argument("--calls", type=int, default=1)
    a = ap.parse_args()
    import torch
    import folddisco_amd as fd
    from folddisco_amd import synth
    dev = torch.device("cuda", 0)
    ctx = fd.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    db = synth.generate(a.structures, seed=7, device=dev)
    rb = db["res_off"].contiguous()
    wb = ctx.wrap_device(a.structures, int(rb[-1].item()), rb.data_ptr(), db["n_xyz"].data_ptr(), db["ca_xyz"].data_ptr(), db["cb_xyz"].data_ptr(),
                         db["aa"].data_ptr(), None, keepalive=(rb, db))
    for k in range(a.calls):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ix = fd.FolddiscoIndex.build(ctx, wb)
        ctx.synchronize()
        dt = time.perf_counter() - t0
        print("BUILD call %d: %d structures, %d residues, %d postings, %d hashes in %.2f ms" % (k, a.structures, int(rb[-1].item()), ix.num_postings, ix.num_hashes, dt * 1e3), flush=True)
        del ix


if __name__ == "__main__":
    main()
