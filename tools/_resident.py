"""A resident synthetic database for the profiling tools: blocks of <= 67,750 structures (one build call each), merged on the device into
one index, and one batch over all the coordinates — like bench.py's generator, so that 542,000 structures fit beside their index."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build_resident(S, seed0=7):
    import torch
    import folddisco_amd as fd
    from folddisco_amd import synth
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    ctx = fd.Context(0, stream=torch.cuda.current_stream(dev).cuda_stream)
    BLK = 67750
    ds, parts, wraps = [], [], []
    for b0 in range(0, S, BLK):
        n = min(BLK, S - b0)
        db = synth.generate(n, seed=seed0 + 1000 * (b0 // BLK), device=dev)
        rb = db["res_off"].contiguous()
        wb = ctx.wrap_device(n, int(rb[-1].item()), rb.data_ptr(), db["n_xyz"].data_ptr(), db["ca_xyz"].data_ptr(), db["cb_xyz"].data_ptr(),
                             db["aa"].data_ptr(), None, keepalive=(rb, db))
        parts.append(fd.FolddiscoIndex.build(ctx, wb, first_id=b0))
        ds.append(db); wraps.append(wb)
    ix = parts[0] if len(parts) == 1 else fd.FolddiscoIndexSet(parts).merge()
    del parts
    offs = [0]
    for db in ds:
        offs.append(offs[-1] + int(db["res_off"][-1].item()))
    d = dict(res_off=torch.cat([db["res_off"][(1 if k else 0):] + offs[k] for k, db in enumerate(ds)]),
             n_xyz=torch.cat([db["n_xyz"] for db in ds]), ca_xyz=torch.cat([db["ca_xyz"] for db in ds]), cb_xyz=torch.cat([db["cb_xyz"] for db in ds]),
             aa=torch.cat([db["aa"] for db in ds]))
    del ds, wraps
    ro = d["res_off"].contiguous()
    batch = ctx.wrap_device(S, int(ro[-1].item()), ro.data_ptr(), d["n_xyz"].data_ptr(), d["ca_xyz"].data_ptr(), d["cb_xyz"].data_ptr(),
                            d["aa"].data_ptr(), None, keepalive=(ro, d))
    ctx.release_workspaces()
    return ctx, batch, ix, d, ro
