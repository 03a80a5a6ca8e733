"""One build call over S structures == two chunks merged per hash (fdgpu_merge_subindices), byte for byte, at bench scale."""
import hashlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import folddisco_amd as fd
from folddisco_amd import synth, indexio
S = int(sys.argv[1]) if len(sys.argv) > 1 else 67750
dev = torch.device("cuda", 0)
d = synth.generate(S, seed=11, device=dev)
ro = d["res_off"].contiguous(); R = int(ro[-1].item())
ctx = fd.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
def wrap(a, b):
    off = ro.cpu()
    r0, r1 = int(off[a]), int(off[b])
    sub = (ro[a:b + 1] - ro[a]).contiguous()
    keep = (sub, d["n_xyz"][r0:r1], d["ca_xyz"][r0:r1], d["cb_xyz"][r0:r1], d["aa"][r0:r1])
    return ctx.wrap_device(b - a, r1 - r0, sub.data_ptr(), keep[1].data_ptr(), keep[2].data_ptr(), keep[3].data_ptr(), keep[4].data_ptr(), None, keepalive=keep)
whole = fd.FolddiscoIndex.build(ctx, wrap(0, S))
v, h, o = whole.export()
print("whole:", whole.num_postings, whole.num_hashes, whole.value_len)
del whole
mid = S // 2
parts = []
for a, b in ((0, mid), (mid, S)):
    ix = fd.FolddiscoIndex.build(ctx, wrap(a, b), first_id=a)
    parts.append(ix.export()); del ix
mv, mh, mo = indexio.merge_subindices(parts)
ok = np.array_equal(mv, v) and np.array_equal(mh, h) and np.array_equal(mo, o)
# posting lists decode to ascending ids below S
assert int(o[-1]) == len(v)
print("IDENTICAL" if ok else "MISMATCH", hashlib.sha256(v.tobytes()).hexdigest()[:16])
sys.exit(0 if ok else 1)
