"""Round-5 sort microbenchmark (GPU box): dvs_sort_pairs_u32 through the segmented sort (frontend.hip) and, with DVS_FRONTEND=legacy, the
batch-wide sort of rounds 1-4, on depth-like keys. usage: tools/r5_sortbench.py [n ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from divshot_amd.raster import Rasterizer

sizes = [int(x) for x in sys.argv[1:]] or [1000000, 8000000, 24000000]
for fe in ("legacy", "seg"):
    os.environ["DVS_FRONTEND"] = fe
    r = Rasterizer(0, max_splats=1 << 16, max_w=64, max_h=64)
    for n in sizes:
        g = torch.Generator(device="cuda").manual_seed(n)
        depth = 2.0 + 10.0 * torch.rand(n, device="cuda", generator=g)
        keys0 = depth.view(torch.int32).clone()
        for bits in (32, 13):
            k0 = keys0 if bits == 32 else (torch.randint(0, 8160, (n,), device="cuda", generator=g, dtype=torch.int32))
            ref = torch.sort(k0.to(torch.int64) & ((1 << bits) - 1), stable=True)
            ts = []
            for it in range(6):
                k = k0.clone(); v = torch.arange(n, device="cuda", dtype=torch.int32)
                torch.cuda.synchronize(); t0 = time.perf_counter()
                r.sort_pairs(k, v, 0, bits)
                torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
            ok = bool(torch.equal(v.to(torch.int64), ref.indices)) and bool(torch.equal(k, k0[ref.indices]))
            print(f"[{fe}] n={n} bits={bits}: {min(ts):.3f} ms (min of 6, host-timed)  correct={ok}", flush=True)
    r.close()
