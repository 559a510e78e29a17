"""Experiment: software-pipeline independent views over two contexts / two HIP streams (development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import divshot_amd as dv
from divshot_amd.raster import Rasterizer, params_to_device
from divshot_amd.parallel import GradBuffer
n, W, H, deg = 1_000_000, 1920, 1080, 3
nctx = int(sys.argv[1]) if len(sys.argv) > 1 else 2
sep = len(sys.argv) > 2 and sys.argv[2] == "sep"
spec = dv.make_spec(n, W, H, sh_degree=3, n_cams=8)
P = dv.synth_splats(spec)
dev = torch.device("cuda", 0)
params = params_to_device(P, dev)
r0 = Rasterizer(0, max_splats=n, max_w=W, max_h=H)
params["shN"] = r0.shn_relayout(params["shN"], n, True)
rasts = [r0] + [Rasterizer(0, max_splats=n, max_w=W, max_h=H) for _ in range(nctx - 1)]
streams = [torch.cuda.Stream() for _ in range(nctx)]
cams = [dv.synth_camera(spec, i) for i in range(8)]
tgts = [torch.from_numpy(dv.synth_target(spec, i)).to(dev) for i in range(8)]
outs = [torch.empty((3, H, W), device=dev) for _ in range(nctx)]
gb = GradBuffer(n, dev, shn_tiled=True)
grads = dict(gb.views); grads["absgrad2d"] = torch.zeros((n, 2), device=dev)
gabs = [torch.zeros((n, 2), device=dev) for _ in range(nctx)]
gbs = [GradBuffer(n, dev, shn_tiled=True) for _ in range(nctx)]
bwd_done = [torch.cuda.Event() for _ in range(nctx)]
inv_P = 1.0 / (W * H)
def run(views):
    for v in range(views):
        c = v % nctx
        with torch.cuda.stream(streams[c]):
            img = rasts[c].forward(params, cams[v % 8], sh_degree=deg, absgrad=True, out=outs[c], shn_tiled=True)
            dL = (img - tgts[v % 8]) * inv_P
            if sep:
                g = dict(gbs[c].views); g["absgrad2d"] = gabs[c]
                rasts[c].backward(dL, grads=g, accumulate=False)
            else:
                if v > 0:
                    streams[c].wait_event(bwd_done[(v - 1) % nctx])     # gradient rows are accumulated in view order
                rasts[c].backward(dL, grads=grads, accumulate=(v > 0))
            bwd_done[c].record(streams[c])
    torch.cuda.synchronize()
run(8)
t0 = time.perf_counter(); V = 64; run(V); el = time.perf_counter() - t0
print(f"contexts {nctx}: {V / el:.1f} views/s, {el / V * 1e3:.3f} ms/view")
