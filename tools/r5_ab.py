"""Round-5 A/B of the binning front end (GPU box): DVS_FRONTEND=legacy (batch-wide sort, binning.hip) vs the segmented one (frontend.hip).
usage: tools/r5_ab.py [n] [W] [H] [views] [iters]   — prints per-stage hipEvent times of the multi-view pass for both, and checks that every
saved array, the images and the gradients are bit-identical between the two."""
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
out = {}
for fe in ("legacy", "seg"):
    os.environ["DVS_FRONTEND"] = fe
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
    print(f"[{fe}] n={n} {W}x{H} V={V} T={T}  front end {fe_ms:.4f} ms  total {sum(tab.values()):.4f} ms\n    {tab}", flush=True)
    s = r.state
    tiles = s.tiles_x * s.tiles_y
    keep = {"img": img.cpu().numpy(), "ranges": r._d2h(s.ranges, (V * tiles, 2), np.uint32), "tile": r._sorted_tile(s, T, V * tiles),
            "splat": r._d2h(s.sorted_splat, (T,), np.uint32), "n_contrib": r._d2h(s.n_contrib, (V, H, W), np.uint32)}
    for k in ("pos", "opacity", "scale", "rot", "sh0"):
        keep["g_" + k] = g[k].cpu().numpy()
    out[fe] = keep
    r.close()
ok = True
for k in out["legacy"]:
    a, b = out["legacy"][k], out["seg"][k]
    same = a.shape == b.shape and np.array_equal(a, b)
    if not same and k.startswith("g_"):        # gradients: atomics order differs run to run; compare to roundoff
        d = np.abs(a - b).max() / (np.abs(a).max() + 1e-30)
        same = d < 1e-5
        print(f"  {k}: max rel diff {d:.2e}")
    if not same:
        ok = False
        print(f"  MISMATCH in {k}: shapes {a.shape} {b.shape}", (a != b).sum() if a.shape == b.shape else "")
print("AB_EQUAL" if ok else "AB_DIFFERENT")
