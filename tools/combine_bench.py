#!/usr/bin/env python3
"""Time dvs_sh_grad_combine (the local rebuild of the SH gradient rows in the factorised multi-GPU exchange) for the view counts an
8-GPU step produces. usage: python tools/combine_bench.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from divshot_amd.raster import Rasterizer, tiled_floats

n = 1_000_000
dev = torch.device("cuda:0")
rast = Rasterizer(0, max_splats=n, max_w=64, max_h=64)
pos = torch.randn(n, 3, device=dev)
sh0 = torch.zeros(n, 3, device=dev); shn = torch.zeros(tiled_floats(n), device=dev)
for V in (1, 8, 16, 32, 64):
    dcol = torch.randn(V, n, 3, device=dev)
    campos = np.random.default_rng(0).standard_normal((V, 3)).astype(np.float32)
    for _ in range(3):
        rast.sh_grad_combine(pos, campos, dcol, sh0, shn, 3, shn_tiled=True)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        rast.sh_grad_combine(pos, campos, dcol, sh0, shn, 3, shn_tiled=True)
    b.record(); torch.cuda.synchronize()
    print(f"views {V:3d}: {a.elapsed_time(b) / 10:.3f} ms")
