"""Index build time of every encoding on a synthetic shard.  usage: other_enc_bench.py [S]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import folddisco_amd as fd
from folddisco_amd import synth
from folddisco_amd._lib import HASH_TYPE_NAMES
S = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
dev = torch.device("cuda", 0)
d = synth.generate(S, seed=7, device=dev)
ro = d["res_off"].contiguous(); R = int(ro[-1].item())
ctx = fd.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
b = ctx.wrap_device(S, R, ro.data_ptr(), d["n_xyz"].data_ptr(), d["ca_xyz"].data_ptr(), d["cb_xyz"].data_ptr(), d["aa"].data_ptr(), None, keepalive=d)
for t in (3, 0, 1, 7, 8, 2, 4, 5, 6):
    for rep in range(2):
        ctx.synchronize(); t0 = time.perf_counter()
        ix = fd.FolddiscoIndex.build(ctx, b, hash_type=t)
        ctx.synchronize(); dt = time.perf_counter() - t0
    print(f"{HASH_TYPE_NAMES[t]:20s} {1e3 * dt:9.1f} ms  {S / dt:10.0f} structures/s  postings {ix.num_postings}  hashes {ix.num_hashes}")
    del ix
