"""Speculative vs exact evaluation of the index build: identical index bytes?  usage: spec_check.py [S] [seed]
Runs itself three times — the default (speculative torsions + squared-distance table), FDGPU_DTAB=0 (speculative torsions, sqrt + quantiser for the
distances) and FDGPU_EXACT=1 (exact table form) — and compares sha256 of (value, hashes, offsets)."""
import hashlib, os, subprocess, sys, json, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
S = int(sys.argv[1]) if len(sys.argv) > 1 else 67750
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 20260927
if os.environ.get("FD_SPEC_CHILD"):
    import torch
    import folddisco_amd as fd
    from folddisco_amd import synth
    dev = torch.device("cuda", 0)
    d = synth.generate(S, seed=seed, device=dev)
    ro = d["res_off"].contiguous(); R = int(ro[-1].item())
    ctx = fd.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    b = ctx.wrap_device(S, R, ro.data_ptr(), d["n_xyz"].data_ptr(), d["ca_xyz"].data_ptr(), d["cb_xyz"].data_ptr(), d["aa"].data_ptr(), None, keepalive=d)
    ctx.spec_fallbacks()
    t0 = time.perf_counter()
    ix = fd.FolddiscoIndex.build(ctx, b)
    ctx.synchronize()
    dt = time.perf_counter() - t0
    miss = ctx.spec_fallbacks()
    v, h, o = ix.export()
    dig = hashlib.sha256(v.tobytes()); dig.update(h.tobytes()); dig.update(o.tobytes())
    print(json.dumps(dict(exact=os.environ.get("FDGPU_EXACT", "0"), dtab=os.environ.get("FDGPU_DTAB", "1"), sha256=dig.hexdigest(), postings=ix.num_postings, hashes=ix.num_hashes,
                          value_len=ix.value_len, fallback_pairs=miss, build_s=dt)))
    sys.exit(0)
res = []
for ex, dt in (("0", "1"), ("0", "0"), ("1", "1")):
    env = dict(os.environ, FD_SPEC_CHILD="1", FDGPU_EXACT=ex, FDGPU_DTAB=dt)
    out = subprocess.run([sys.executable, os.path.abspath(__file__), str(S), str(seed)], env=env, capture_output=True, text=True, timeout=900)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(out.stdout[-2000:], out.stderr[-2000:]); sys.exit(2)
    res.append(json.loads(line[-1])); print(res[-1])
same = res[0]["sha256"] == res[1]["sha256"] == res[2]["sha256"]
print("IDENTICAL" if same else "MISMATCH", "fallback rate %.3g" % (res[0]["fallback_pairs"] / max(res[0]["postings"] / 2, 1)))
sys.exit(0 if same else 1)
