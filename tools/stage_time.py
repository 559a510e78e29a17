"""Per-stage hipEvent timing for an arbitrary config: tools/stage_time.py N W H DEG [iters]  (development aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import divshot_amd as dv
from divshot_amd.raster import Rasterizer, params_to_device
n, W, H, deg = [int(x) for x in sys.argv[1:5]]
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 10
spec = dv.make_spec(n, W, H, sh_degree=3, n_cams=8)
P = dv.synth_splats(spec); cam = dv.synth_camera(spec, 0)
r = Rasterizer(0, max_splats=n, max_w=W, max_h=H)
Pd = params_to_device(P, r.tdev)
tgt = torch.from_numpy(dv.synth_target(spec, 0)).to(r.tdev)
r.enable_timing(True)
acc = {}
for it in range(iters + 2):
    img = r.forward(Pd, cam, sh_degree=deg, absgrad=True)
    r.backward(((img - tgt) / (W * H)).contiguous())
    if it >= 2:
        for k, v in r.stage_timing().items(): acc.setdefault(k, []).append(v)
print("n", n, "deg", deg, "T", r.num_rendered, {k: round(float(np.mean(v)), 4) for k, v in acc.items()}, "total", round(sum(float(np.mean(v)) for v in acc.values()), 4))
