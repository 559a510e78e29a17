#!/usr/bin/env python3
"""Time A9 (dvs_raster_backward_project) alone for one view on the bench scene: full rows vs factorised (GPU box)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import divshot_amd as dv
from divshot_amd.raster import Rasterizer, params_to_device
from bench import WORKLOADS
n, W, H, deg, soff = WORKLOADS["C3"]
spec = dv.make_spec(n, W, H, sh_degree=deg, n_cams=8, scale_log_offset=soff)
dev = torch.device("cuda", 0)
P = params_to_device(dv.synth_splats(spec), dev)
cams = [dv.synth_camera(spec, i) for i in range(2)]
tgt = torch.from_numpy(dv.synth_target(spec, 0)).to(dev)
for V in (1, 2):
    r = Rasterizer(0, max_splats=n, max_w=W, max_h=H, max_views=V)
    Pt = dict(P); Pt["shN"] = r.shn_relayout(P["shN"], n, to_tiled=True)
    imgs = r.forward_views(Pt, cams[:V], sh_degree=deg, absgrad=True, shn_tiled=True)
    dL = ((imgs - tgt) / (W * H)).contiguous()
    for fact in (False, True):
        ms = []
        g = None
        for i in range(12):
            r.backward_composite(dL)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); g = r.backward_project(grads=g, factorised_sh=fact); e1.record()
            torch.cuda.synchronize(); ms.append(e0.elapsed_time(e1))
        print(json.dumps({"views": V, "factorised": fact, "a9_ms": float(np.mean(ms[2:]))}), flush=True)
    r.close()
