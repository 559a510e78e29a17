"""Per-stage hipEvent times of one multi-view pass (GPU box). usage: tools/r5_ab.py [n] [W] [H] [views] [iters]
(Round 5 used this tool for the A/B of the segmented binning front end against the batch-wide one of rounds 1-4 — bit-identical lists,
images and gradients on three shapes, profiles/r05_frontend_ab_*.txt; the old front end has since left the tree, what remains is the
timing half. DVS_FE_NO_FUSE_A6=1 times the separate tile-range kernel.)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import divshot_amd as dv
from divshot_amd.raster import Rasterizer, params_to_device, shn_rows_to_tiled_np

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 1920
H = int(sys.argv[3]) if len(sys.argv) > 3 else 1080
V = int(sys.argv[4]) if len(sys.argv) > 4 else 8
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 10
spec = dv.make_spec(n, W, H, sh_degree=3, n_cams=max(V, 2))
P = dv.synth_splats(spec)
cams = [dv.synth_camera(spec, v) for v in range(V)]
r = Rasterizer(0, max_splats=n, max_w=W, max_h=H, max_views=V)
Pd = params_to_device(P, r.tdev)
Pd["shN"] = torch.from_numpy(shn_rows_to_tiled_np(P["shN"])).to(r.tdev)
tg = torch.rand((V, 3, H, W), device=r.tdev, generator=torch.Generator(device=r.tdev).manual_seed(1))
r.set_async(True)
r.enable_timing(True)
acc = {}
for it in range(iters + 3):
    img = r.forward_views(Pd, cams, sh_degree=3, absgrad=True, shn_tiled=True)
    fw = r.stage_timing()
    g = r.backward_views(((img - tg) / (W * H)).contiguous())
    bw = r.stage_timing()
    if it >= 3:
        for k, v in {**fw, **bw}.items():
            acc.setdefault(k, []).append(v)
torch.cuda.synchronize()
T = r.get_num_rendered()
tab = {k: round(float(np.mean(v)), 4) for k, v in acc.items()}
fe_ms = sum(tab.get(k, 0) for k in ("preprocess_fwd", "depth_sort", "tile_scan", "duplicate", "tile_sort", "tile_ranges"))
print(f"n={n} {W}x{H} V={V} T={T}  front end {fe_ms:.4f} ms  total {sum(tab.values()):.4f} ms\n    {tab}", flush=True)
r.close()
