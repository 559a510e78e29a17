#!/usr/bin/env python3
"""Time the trainer-side kernels of include/dvs_train.h on the GPU (SSIM forward/backward at 1080p, Adam over the six
parameter groups of a 1M-splat model) and print achieved HBM rates. usage: python tools/train_ops_bench.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from divshot_amd.train_ops import Ssim, adam_step, adam_step_groups


def timed(fn, iters=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / iters * 1e3     # us


def main():
    dev = "cuda:0"
    W, H = 1920, 1080
    x = torch.rand(3, H, W, device=dev); y = torch.rand(3, H, W, device=dev); dL = torch.zeros_like(x)
    s = Ssim(W, H, dev)
    t = timed(lambda: s.forward(x, y)); print(f"ssim_fwd  {t:8.1f} us  {5 * x.numel() * 4 / t / 1e6:7.2f} TB/s (5 planes)")
    t = timed(lambda: s.backward(x, y, dL, -0.2, True)); print(f"ssim_bwd  {t:8.1f} us  {7 * x.numel() * 4 / t / 1e6:7.2f} TB/s (7 planes)")
    n = 1_000_000
    widths = [3, 3, 48, 1, 3, 4]
    arrs = [[torch.rand(n * w, device=dev) for _ in range(4)] for w in widths]
    def old():
        for a in arrs:
            adam_step(a[0], a[1], a[2], a[3], 1e-3, 5)
    total = sum(widths) * n * 4 * 7
    t = timed(old); print(f"adam x6   {t:8.1f} us  {total / t / 1e6:7.2f} TB/s")
    n64 = n // 64 * 64
    arrs[2] = [torch.rand(n64 * 48, device=dev) for _ in range(4)]
    def grp(active=0, vis=None):
        g = [dict(param=a[0], grad=a[1], m=a[2], v=a[3], lr=1e-3, width=(45 if w == 48 else w), tiled=(w == 48),
                  active_chunks=(active if w == 48 else 0)) for a, w in zip(arrs, widths)]
        return lambda: adam_step_groups(g, 5, visible=vis)
    t = timed(grp()); print(f"adam grp  {t:8.1f} us  {total / t / 1e6:7.2f} TB/s")
    t = timed(grp(3)); print(f"adam grp active=3 {t:8.1f} us")
    vis = (torch.rand(n, device=dev) < 0.85).int()
    t = timed(grp(0, vis)); print(f"adam grp visible 85% {t:8.1f} us")
    vis = (torch.arange(n, device=dev) % 1000 < 150).int()
    t = timed(grp(0, vis)); print(f"adam grp visible 15% clustered {t:8.1f} us")


if __name__ == "__main__":
    main()
