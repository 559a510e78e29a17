import sys, os, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from divshot_amd.train_ops import Ssim
W, H = 1920, 1080
dev = torch.device("cuda", 0)
img = torch.rand((3, H, W), device=dev); tgt = torch.rand((3, H, W), device=dev)
s = Ssim(W, H, dev)
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
print("ssim fwd ms", t(lambda: s.forward(img, tgt)))
print("loss bwd ms", t(lambda: s.loss_backward(img, tgt, 0.2)))
dL = torch.zeros_like(img)
print("ssim bwd (no l1 sum) ms", t(lambda: s.backward(img, tgt, dL, 1.0, accumulate=False)))
