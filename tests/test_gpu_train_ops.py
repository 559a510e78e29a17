"""SURVEY.md §8(f) rows 2-3 on the GPU: L1 loss gradient, fused SSIM forward/backward, fused Adam — each against a numpy
restatement (and, for SSIM, fp64 finite differences of that restatement)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def gauss_window():
    x = np.arange(11) - 5.0
    g = np.exp(-x * x / (2 * 1.5 ** 2))
    return g / g.sum()


def conv_same(img, g):
    """separable 11-tap convolution with zero padding, img [H,W] float64"""
    H, W = img.shape
    p = np.pad(img, 5)
    t = sum(g[k] * p[:, k:k + W] for k in range(11))
    return sum(g[k] * t[k:k + H, :] for k in range(11))


def ssim_np(x, y):
    g = gauss_window()
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    tot = 0.0
    for c in range(3):
        mu1, mu2 = conv_same(x[c], g), conv_same(y[c], g)
        s1 = conv_same(x[c] * x[c], g) - mu1 * mu1
        s2 = conv_same(y[c] * y[c], g) - mu2 * mu2
        s12 = conv_same(x[c] * y[c], g) - mu1 * mu2
        tot += (((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s1 + s2 + C2))).sum()
    return tot / x.size


def test_ssim_forward_backward(gpu_device):
    import torch
    from divshot_amd.train_ops import Ssim
    rng = np.random.default_rng(0)
    H, W = 37, 53                                       # not multiples of the 16x16 tile
    y = rng.uniform(0, 1, (3, H, W))
    x = np.clip(y + 0.15 * rng.standard_normal((3, H, W)), 0, 1)
    xd = torch.tensor(x, dtype=torch.float32, device=gpu_device)
    yd = torch.tensor(y, dtype=torch.float32, device=gpu_device)
    s = Ssim(W, H, gpu_device)
    val = float(s.forward(xd, yd).item())
    ref = ssim_np(x.astype(np.float32).astype(np.float64), y.astype(np.float32).astype(np.float64))
    assert abs(val - ref) < 2e-5, (val, ref)
    # identical images -> 1
    assert abs(float(s.forward(yd, yd).item()) - 1.0) < 1e-5
    # gradient of the mean SSIM against fp64 central differences on sampled pixels
    s.forward(xd, yd)
    g = s.backward(xd, yd, torch.zeros_like(xd), 1.0, accumulate=False).cpu().numpy()
    x64 = x.astype(np.float32).astype(np.float64); y64 = y.astype(np.float32).astype(np.float64)
    for _ in range(25):
        c, i, j = rng.integers(0, 3), rng.integers(0, H), rng.integers(0, W)
        e = 1e-5
        xp = x64.copy(); xp[c, i, j] += e
        xm = x64.copy(); xm[c, i, j] -= e
        fd = (ssim_np(xp, y64) - ssim_np(xm, y64)) / (2 * e)
        assert abs(g[c, i, j] - fd) <= 2e-3 * abs(fd) + 2e-7, (c, i, j, g[c, i, j], fd)
    # accumulate adds on top
    base = torch.full_like(xd, 0.25)
    g2 = s.backward(xd, yd, base, -0.2, accumulate=True).cpu().numpy()
    np.testing.assert_allclose(g2, 0.25 - 0.2 * g, rtol=1e-5, atol=1e-8)


def test_l1_and_adam(gpu_device):
    import torch
    from divshot_amd.train_ops import l1_loss_grad, adam_step
    rng = np.random.default_rng(1)
    a = rng.standard_normal(100_003).astype(np.float32); b = rng.standard_normal(100_003).astype(np.float32)
    dL, loss = l1_loss_grad(torch.tensor(a, device=gpu_device), torch.tensor(b, device=gpu_device))
    assert abs(float(loss.item()) - np.abs(a - b).mean()) < 1e-5
    np.testing.assert_allclose(dL.cpu().numpy(), np.sign(a - b) / a.size, rtol=1e-6, atol=1e-12)
    # Adam: three steps against the textbook recurrences
    p = rng.standard_normal(4099).astype(np.float32); m = np.zeros_like(p); v = np.zeros_like(p)
    pd = torch.tensor(p, device=gpu_device); md = torch.zeros_like(pd); vd = torch.zeros_like(pd)
    for t in (1, 2, 3):
        g = rng.standard_normal(p.size).astype(np.float32)
        adam_step(pd, torch.tensor(g, device=gpu_device), md, vd, 1e-2, t, eps=1e-8)
        m = 0.9 * m + 0.1 * g; v = 0.999 * v + 0.001 * g * g
        p = p - 1e-2 * (m / (1 - 0.9 ** t)) / (np.sqrt(v / (1 - 0.999 ** t)) + 1e-8)
    np.testing.assert_allclose(pd.cpu().numpy(), p, rtol=2e-5, atol=1e-6)
