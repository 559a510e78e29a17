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


def test_fused_l1_ssim_backward(gpu_device):
    """dvs_loss_l1_ssim_backward == dvs_l1_loss_grad_w(1-w) followed by dvs_ssim_backward(-w, accumulate), odd image size."""
    import ctypes as C
    import torch
    from divshot_amd._lib import lib, check
    from divshot_amd.train_ops import Ssim
    rng = np.random.default_rng(8)
    H, W, w = 45, 83, 0.2
    x = torch.tensor(rng.random((3, H, W)).astype(np.float32), device=gpu_device)
    y = torch.tensor(rng.random((3, H, W)).astype(np.float32), device=gpu_device)
    y[0, :5] = x[0, :5]                                     # exact ties: sign(0) = 0 in both paths
    s = Ssim(W, H, gpu_device)
    s.forward(x, y)
    dL_f, l1_f = s.loss_backward(x, y, w)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    dL_r = torch.empty_like(x); acc = torch.zeros(1, device=gpu_device)
    check(lib.dvs_l1_loss_grad_w(st, x.data_ptr(), y.data_ptr(), x.numel(), 1.0 - w, dL_r.data_ptr(), acc.data_ptr()))
    s.backward(x, y, dL_r, -w, accumulate=True)
    torch.cuda.synchronize()
    np.testing.assert_allclose(dL_f.cpu().numpy(), dL_r.cpu().numpy(), rtol=1e-6, atol=1e-10)
    want = (1 - w) * float((x - y).abs().double().mean())
    assert abs(float(l1_f.item()) - want) < 1e-6 * max(want, 1) and abs(float(acc.item()) - want) < 1e-6


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


def test_adam_groups(gpu_device):
    """dvs_adam_step_groups: one launch == the per-group kernel (up to FMA contraction); active-chunk skipping is the identity on
    never-touched coefficients; `visible` freezes invisible splats (the reference's visibleAdam, gs_train.cpp:87)."""
    import torch
    from divshot_amd.train_ops import adam_step, adam_step_groups
    from divshot_amd.raster import shn_rows_to_tiled_np
    rng = np.random.default_rng(5)
    n = 1000                                                   # not a multiple of 64: the last shN tile is padded
    widths = {"pos": 3, "sh0": 3, "shN": 45, "opacity": 1, "scale": 3, "rot": 4}
    dev = lambda a: torch.tensor(a, device=gpu_device)

    def make(seed, deg=3):
        r = np.random.default_rng(seed)
        out = {}
        for k, w in widths.items():
            a = [r.standard_normal((n, w)).astype(np.float32) for _ in range(2)] + [np.abs(r.standard_normal((n, w))).astype(np.float32) * 0.1
                                                                                     for _ in range(2)]
            if k == "shN" and deg < 3:                          # coefficients above the active degree never saw a gradient
                lo = 3 * ((deg + 1) ** 2 - 1)
                for x in a[1:]:
                    x[:, lo:] = 0
            if k == "shN":
                a = [shn_rows_to_tiled_np(x) for x in a]
            out[k] = [dev(x.reshape(-1)) for x in a]            # param, grad, m, v
        return out

    def groups(t, active=0):
        return [dict(param=t[k][0], grad=t[k][1], m=t[k][2], v=t[k][3], lr=1e-2 * (i + 1), width=w, tiled=(k == "shN"),
                     active_chunks=(active if k == "shN" else 0)) for i, (k, w) in enumerate(widths.items())]

    # 1. dense: same as six dvs_adam_step calls (the two kernels may contract FMAs differently: 1-ulp slack)
    a, b = make(1), make(1)
    adam_step_groups(groups(a), 3, eps=1e-8)
    for i, k in enumerate(widths):
        adam_step(b[k][0], b[k][1], b[k][2], b[k][3], 1e-2 * (i + 1), 3, eps=1e-8)
        for x, y in zip(a[k], b[k]):
            assert torch.allclose(x, y, rtol=2e-6, atol=1e-7), k
    # 2. active chunks at degree 1 (9 floats -> 3 chunks): same result as the dense update
    a, b = make(2, deg=1), make(2, deg=1)
    adam_step_groups(groups(a, active=3), 2, eps=1e-15)
    adam_step_groups(groups(b), 2, eps=1e-15)
    for k in widths:
        for x, y in zip(a[k], b[k]):
            assert torch.equal(x, y), k
    # 3. visible mask
    a, b = make(3), make(3)
    before = {k: [x.clone() for x in a[k]] for k in widths}
    radii = (rng.random(n) < 0.6).astype(np.int32) * 7
    adam_step_groups(groups(a), 4, eps=1e-8, visible=dev(radii))
    adam_step_groups(groups(b), 4, eps=1e-8)
    vis = radii > 0
    for k, w in widths.items():
        if k == "shN":
            e = np.arange(a[k][0].numel()); splat = (e // 3072) * 64 + ((e % 3072) // 4) % 64
        else:
            splat = np.arange(n * w) // w
        mask = dev((splat < n) & vis[np.minimum(splat, n - 1)])
        for j in (0, 2, 3):
            assert torch.equal(a[k][j][mask], b[k][j][mask]), (k, j)
            assert torch.equal(a[k][j][~mask], before[k][j][~mask]), (k, j)
    assert (~vis).any() and vis.any()


@pytest.mark.parametrize("tiled", [False, True])
def test_densify_plan_and_apply(gpu_device, tiled):
    """ADC clone / split / prune against a numpy restatement of the rule."""
    import ctypes as C
    import torch
    from divshot_amd._lib import lib, DensifyParams, check
    from divshot_amd.raster import shn_rows_to_tiled_np, shn_tiled_to_rows_np, tiled_floats
    rng = np.random.default_rng(2)
    n = 10_007
    A = {"pos": rng.normal(size=(n, 3)), "sh0": rng.normal(size=(n, 3)), "shN": rng.normal(size=(n, 15, 3)),
         "opacity": rng.normal(0, 3, size=(n,)), "scale": rng.normal(-3, 1, size=(n, 3)), "rot": rng.normal(size=(n, 4))}
    A = {k: v.astype(np.float32) for k, v in A.items()}
    radii = rng.integers(0, 40, n).astype(np.int32)
    absg = np.abs(rng.normal(0, 3e-7, (n, 2))).astype(np.float32)
    W, H = 640, 360
    dev = gpu_device
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    t = lambda a: torch.tensor(a, device=dev)
    ga, de, mr = torch.zeros(n, device=dev), torch.zeros(n, device=dev), torch.zeros(n, dtype=torch.int32, device=dev)
    radii_d, absg_d = t(radii), t(absg)          # keep the device tensors alive while their raw pointers are in use
    for _ in range(3):
        check(lib.dvs_densify_accumulate(st, n, radii_d.data_ptr(), absg_d.data_ptr(), W, H, ga.data_ptr(), de.data_ptr(), mr.data_ptr()))
    vis = radii > 0
    g_ref = np.where(vis, 3 * np.hypot(absg[:, 0] * W / 2, absg[:, 1] * H / 2), 0)
    np.testing.assert_allclose(ga.cpu().numpy(), g_ref, rtol=1e-5)
    assert np.array_equal(de.cpu().numpy(), np.where(vis, 3.0, 0.0)) and np.array_equal(mr.cpu().numpy(), np.where(vis, radii, 0))
    prm = DensifyParams(grad_threshold=2e-4, scale_threshold=0.05, min_opacity=0.005, max_world_scale=0.0, max_screen_radius=0,
                        cap_max=10 ** 9, seed=77, shn_layout=1 if tiled else 0)
    action = torch.zeros(n, dtype=torch.uint8, device=dev); offs = torch.zeros(n, dtype=torch.int32, device=dev)
    scratch = torch.zeros(n // 256 + 4, dtype=torch.int32, device=dev); total = torch.zeros(1, dtype=torch.int64, device=dev)
    op_d, sc_d = t(A["opacity"]), t(A["scale"])
    check(lib.dvs_densify_plan(st, n, op_d.data_ptr(), sc_d.data_ptr(), ga.data_ptr(), de.data_ptr(), mr.data_ptr(), C.byref(prm),
                               action.data_ptr(), offs.data_ptr(), scratch.data_ptr(), total.data_ptr()))
    torch.cuda.synchronize()
    sig = 1 / (1 + np.exp(-A["opacity"].astype(np.float64)))
    smax = np.exp(A["scale"].max(1).astype(np.float64))
    avg = np.where(vis, g_ref / 3.0, 0.0)
    want = np.where(sig < 0.005, 3, np.where(avg >= 2e-4, np.where(smax > 0.05, 2, 1), 0))
    got = action.cpu().numpy()
    borderline = (np.abs(sig - 0.005) < 1e-6) | (np.abs(avg - 2e-4) < 1e-9) | (np.abs(smax - 0.05) < 1e-6)
    assert np.array_equal(got[~borderline], want[~borderline])
    cnt = np.where(got == 3, 0, np.where(got == 0, 1, 2))
    assert np.array_equal(offs.cpu().numpy().view(np.uint32), np.concatenate([[0], np.cumsum(cnt)[:-1]]).astype(np.uint32))
    new_n = int(total.item())
    assert new_n == cnt.sum() and {0, 1, 2, 3} <= set(got.tolist())
    shn_np = shn_rows_to_tiled_np(A["shN"]) if tiled else A["shN"].reshape(-1)
    src = [t(A["pos"]), t(A["sh0"]), t(shn_np), op_d, sc_d, t(A["rot"])]
    shn_new = tiled_floats(new_n) if tiled else new_n * 45
    for mode in (0, 1):
        dst = [torch.full((new_n * 3,), 9.0, device=dev), torch.full((new_n * 3,), 9.0, device=dev), torch.zeros(shn_new, device=dev),
               torch.full((new_n,), 9.0, device=dev), torch.full((new_n * 3,), 9.0, device=dev), torch.full((new_n * 4,), 9.0, device=dev)]
        sp = (C.c_void_p * 6)(*[x.data_ptr() for x in src]); dp = (C.c_void_p * 6)(*[x.data_ptr() for x in dst])
        check(lib.dvs_densify_apply(st, n, action.data_ptr(), offs.data_ptr(), C.byref(prm), mode, sp, dp, new_n))
        torch.cuda.synchronize()
        pos = dst[0].cpu().numpy().reshape(new_n, 3); sc = dst[4].cpu().numpy().reshape(new_n, 3)
        op = dst[3].cpu().numpy(); rot = dst[5].cpu().numpy().reshape(new_n, 4)
        shn = shn_tiled_to_rows_np(dst[2].cpu().numpy(), new_n) if tiled else dst[2].cpu().numpy().reshape(new_n, 15, 3)
        o = offs.cpu().numpy().view(np.uint32).astype(np.int64)
        keep, clone, split = np.where(got == 0)[0], np.where(got == 1)[0], np.where(got == 2)[0]
        assert not (pos == 9.0).any()                                         # every output row was written
        if mode == 0:
            for idx, slots in ((keep, (0,)), (clone, (0, 1))):
                for c in slots:
                    assert np.array_equal(pos[o[idx] + c], A["pos"][idx]) and np.array_equal(shn[o[idx] + c], A["shN"][idx])
                    assert np.array_equal(op[o[idx] + c], A["opacity"][idx]) and np.array_equal(sc[o[idx] + c], A["scale"][idx])
            for c in (0, 1):
                np.testing.assert_allclose(sc[o[split] + c], A["scale"][split] - np.log(1.6), rtol=1e-6, atol=1e-6)
                assert np.array_equal(rot[o[split] + c], A["rot"][split]) and np.array_equal(shn[o[split] + c], A["shN"][split])
                d = np.abs(pos[o[split] + c] - A["pos"][split]).max(1)
                assert (d <= 6.0 * np.exp(A["scale"][split].max(1)) + 1e-6).all() and (d > 0).mean() > 0.99
            assert not np.array_equal(pos[o[split]], pos[o[split] + 1])       # the two children are different samples
        else:
            assert np.array_equal(pos[o[keep]], A["pos"][keep]) and np.array_equal(pos[o[clone]], A["pos"][clone])
            assert not pos[o[clone] + 1].any() and not pos[o[split]].any() and not pos[o[split] + 1].any() and not shn[o[split]].any()
    # config `revisedOpacity`: both results of a clone / split take 1 - sqrt(1 - o); kept splats are untouched
    prm.revised_opacity = 1
    dst = [torch.zeros(new_n * 3, device=dev), torch.zeros(new_n * 3, device=dev), torch.zeros(shn_new, device=dev),
           torch.zeros(new_n, device=dev), torch.zeros(new_n * 3, device=dev), torch.zeros(new_n * 4, device=dev)]
    dp = (C.c_void_p * 6)(*[x.data_ptr() for x in dst])
    check(lib.dvs_densify_apply(st, n, action.data_ptr(), offs.data_ptr(), C.byref(prm), 0, sp, dp, new_n))
    torch.cuda.synchronize()
    op = dst[3].cpu().numpy().astype(np.float64)
    assert np.array_equal(dst[3].cpu().numpy()[o[keep]], A["opacity"][keep])
    for idx in (clone, split):
        want_o = 1.0 - np.sqrt(1.0 - sig[idx])
        for c in (0, 1):
            got_o = 1 / (1 + np.exp(-op[o[idx] + c]))
            np.testing.assert_allclose(got_o, np.clip(want_o, 1e-6, 1 - 1e-6), rtol=2e-4, atol=1e-7)
            # the pair composites to the opacity it replaces: 1 - (1 - o')^2 = o
        np.testing.assert_allclose(1 - (1 - 1 / (1 + np.exp(-op[o[idx]]))) ** 2, sig[idx], rtol=5e-4, atol=2e-6)
    # opacity reset
    m, v = torch.ones(n, device=dev), torch.ones(n, device=dev)
    check(lib.dvs_reset_opacity(st, n, op_d.data_ptr(), 0.01, m.data_ptr(), v.data_ptr()))
    torch.cuda.synchronize()
    np.testing.assert_allclose(op_d.cpu().numpy(), np.minimum(A["opacity"], np.log(0.01 / 0.99)), rtol=1e-6)
    assert not m.any() and not v.any()


def _relocation_np(o, ratio, min_opacity):
    """numpy restatement of the MCMC relocation rule (opacity / scale of the c+1 copies of a splat drawn c times)."""
    from math import comb, sqrt
    no = 1.0 - (1.0 - o) ** (1.0 / ratio)
    denom = sum(comb(i - 1, k) * (-1) ** k * no ** (k + 1) / sqrt(k + 1) for i in range(1, ratio + 1) for k in range(i))
    return min(max(no, min_opacity), 1.0 - 1.1920929e-7), o / denom


@pytest.mark.parametrize("tiled", [False, True])
def test_mcmc_relocate_grow(gpu_device, tiled):
    """dvs_mcmc_relocate / dvs_mcmc_grow (densifyStrategy 1): dead splats become copies of live ones, growth appends copies, every
    drawn splat and its copies carry the relocated opacity/scale of the published rule, moments of everything touched are zero,
    untouched splats are bit-identical, and sources are drawn roughly in proportion to opacity."""
    import ctypes as C
    import torch
    from divshot_amd._lib import lib, check, McmcSets
    from divshot_amd.raster import shn_rows_to_tiled_np, shn_tiled_to_rows_np
    rng = np.random.default_rng(11)
    n, cap, n_new, min_op = 5000, 6000, 700, 0.005
    widths = [3, 3, 45, 1, 3, 4]
    P = [rng.standard_normal((n, w)).astype(np.float32) for w in widths]
    P[3][:, 0] = rng.normal(0, 2.0, n)                        # logits
    dead = rng.random(n) < 0.1
    P[3][dead, 0] = -8.0                                      # sigmoid = 3e-4 < min_opacity
    dead = 1.0 / (1.0 + np.exp(-P[3][:, 0].astype(np.float64))) <= min_op      # plus the few random logits below the threshold
    P[4] = rng.normal(-3, 0.3, (n, 3)).astype(np.float32)

    def to_dev(a, g):
        full = np.zeros((cap, widths[g]), np.float32); full[:n] = a
        if g == 2 and tiled:
            return torch.tensor(shn_rows_to_tiled_np(full), device=gpu_device)
        return torch.tensor(full.reshape(-1), device=gpu_device)

    def to_host(t, g):
        a = t.cpu().numpy()
        return shn_tiled_to_rows_np(a, cap).reshape(cap, 45) if (g == 2 and tiled) else a.reshape(cap, widths[g])

    par = [to_dev(P[g], g) for g in range(6)]
    mom = [[torch.ones_like(par[g]) for g in range(6)] for _ in range(2)]
    sets = McmcSets()
    for g in range(6):
        sets.param[g], sets.m[g], sets.v[g] = par[g].data_ptr(), mom[0][g].data_ptr(), mom[1][g].data_ptr()
    scratch = torch.empty(int(lib.dvs_mcmc_scratch_bytes(cap)), dtype=torch.uint8, device=gpu_device)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    check(lib.dvs_mcmc_init_scratch(st, scratch.data_ptr(), cap))
    n_dead = torch.zeros(1, dtype=torch.int32).pin_memory()
    check(lib.dvs_mcmc_relocate(st, n, C.byref(sets), min_op, 7, int(tiled), scratch.data_ptr(), cap, n_dead.data_ptr()), "relocate")
    torch.cuda.synchronize()
    assert int(n_dead[0]) == int(dead.sum())
    A = [to_host(par[g], g) for g in range(6)]
    sig = lambda x: 1.0 / (1.0 + np.exp(-x.astype(np.float64)))
    o0 = sig(P[3][:, 0])

    def check_copies(A, dst_rows, n_src, o_src, P_src):
        """each destination row equals a live source row except opacity/scale; returns the source of each destination"""
        key = {tuple(P_src[0][i]): i for i in range(n_src)}                   # positions are unique
        srcs = np.array([key[tuple(A[0][d])] for d in dst_rows])
        assert not (o_src[srcs] <= min_op).any()                              # only live splats are drawn
        for g in (1, 2, 5):
            np.testing.assert_array_equal(A[g][dst_rows], P_src[g][srcs])
        return srcs

    dst = np.nonzero(dead)[0]
    srcs = check_copies(A, dst, n, o0, P)
    cnt = np.bincount(srcs, minlength=n)
    touched = np.zeros(n, bool); touched[dst] = True; touched[cnt > 0] = True
    for i in np.nonzero(cnt > 0)[0][:200]:                                    # relocated opacity / scale of sources and of their copies
        no, coeff = _relocation_np(float(o0[i]), int(cnt[i]) + 1, min_op)
        rows = [i] + list(dst[srcs == i])
        for r in rows:
            assert abs(sig(A[3][r, 0]) - no) <= 2e-5 * max(no, 1e-3), (i, r)
            np.testing.assert_allclose(A[4][r], P[4][i] + np.log(coeff), rtol=0, atol=3e-5)
    for g in range(6):                                                        # untouched splats are bit-identical, moments of touched ones cleared
        np.testing.assert_array_equal(A[g][:n][~touched], P[g][~touched])
        for mm in mom:
            M = to_host(mm[g], g)[:n]
            assert (M[touched] == 0).all() and (M[~touched] == 1).all()
    # sources ~ opacity: the mean opacity of the drawn splats is well above the mean opacity of the live ones
    live = ~dead
    assert o0[srcs].mean() > 1.15 * o0[live].mean()
    # grow: appended copies at [n, n+n_new)
    P1 = [a[:n].copy() for a in A]
    o1 = sig(P1[3][:, 0])
    for mm in mom:
        for g in range(6):
            mm[g].fill_(1.0)
    check(lib.dvs_mcmc_grow(st, n, n_new, C.byref(sets), min_op, 8, int(tiled), scratch.data_ptr(), cap), "grow")
    torch.cuda.synchronize()
    B = [to_host(par[g], g) for g in range(6)]
    new_rows = np.arange(n, n + n_new)
    key = {}
    for i in range(n):                                                        # relocated duplicates share a position: any of them is a valid source
        key.setdefault(tuple(P1[0][i]), i)
    srcs2 = np.array([key[tuple(B[0][d])] for d in new_rows])
    for g in (1, 2, 5):
        np.testing.assert_array_equal(B[g][new_rows], P1[g][srcs2])
    assert (B[0][n + n_new:] == 0).all()
    for mm in mom:
        for g in range(6):
            assert (to_host(mm[g], g)[new_rows] == 0).all()
    # opacity mass is conserved in the sense of the rule: 1 - prod(1 - o') over the copies of a source == its old opacity
    for i in np.unique(srcs2)[:100]:
        same = [r for r in list(np.nonzero((P1[0] == P1[0][i]).all(1))[0])]
        if len(same) > 1:
            continue                                                          # ambiguous source (already a duplicate): skip
        rows = [i] + list(new_rows[srcs2 == i])
        o_after = sig(np.array([B[3][r, 0] for r in rows]))
        assert abs((1 - np.prod(1 - o_after)) - o1[i]) <= 1e-4 * max(o1[i], 0.05) + (len(rows) * min_op if o_after.min() <= min_op * 1.01 else 0)


def test_mcmc_noise_and_regularizer(gpu_device):
    """dvs_mcmc_add_noise: displacement = Sigma z gate(o) lr (zero for solid splats, covariance-shaped for faint ones);
    dvs_mcmc_regularize: gradients of opacity_reg mean(sigmoid) + scale_reg mean(exp)."""
    import ctypes as C
    import torch
    from divshot_amd._lib import lib, check
    n = 40000
    rng = np.random.default_rng(2)
    q = np.array([0.3, -0.5, 0.7, 0.2], np.float32)
    logs = np.log(np.array([0.05, 0.2, 0.1], np.float32))
    pos = np.zeros((n, 3), np.float32)
    scale = np.tile(logs, (n, 1)); rot = np.tile(q, (n, 1))
    opa = np.full(n, -9.0, np.float32); opa[n // 2:] = 2.0          # first half nearly dead (gate ~ 1), second half solid (gate ~ 0)
    d = lambda a: torch.tensor(a, device=gpu_device)
    dp, ds, dr, do = d(pos), d(scale), d(rot), d(opa)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    lr = 0.5
    check(lib.dvs_mcmc_add_noise(st, n, dp.data_ptr(), ds.data_ptr(), dr.data_ptr(), do.data_ptr(), lr, 3))
    torch.cuda.synchronize()
    disp = dp.cpu().numpy().astype(np.float64)
    assert np.abs(disp[n // 2:]).max() < 1e-30                      # sigmoid(-100 (0.88 - 0.005)) underflows
    qn = q / np.linalg.norm(q); w, x, y, z = qn
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)],
                  [2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)],
                  [2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)]])
    Sigma = R @ np.diag(np.exp(2 * logs.astype(np.float64))) @ R.T
    o = 1 / (1 + np.exp(9.0)); gate = 1 / (1 + np.exp(100 * (o - 0.005)))
    want = (gate * lr) ** 2 * Sigma @ Sigma.T
    got = np.cov(disp[: n // 2].T, bias=True)
    assert np.abs(disp[: n // 2].mean(0)).max() < 4 * np.sqrt(np.diag(want).max() / (n // 2))
    np.testing.assert_allclose(got, want, rtol=0.08, atol=0.03 * np.abs(want).max())
    # regulariser
    go = torch.full((n,), 0.25, device=gpu_device); gs = torch.full((n, 3), -0.5, device=gpu_device)
    check(lib.dvs_mcmc_regularize(st, n, do.data_ptr(), ds.data_ptr(), go.data_ptr(), gs.data_ptr(), 0.01, 0.02))
    torch.cuda.synchronize()
    so = 1 / (1 + np.exp(-opa.astype(np.float64)))
    np.testing.assert_allclose(go.cpu().numpy(), 0.25 + 0.01 / n * so * (1 - so), rtol=1e-5, atol=1e-12)
    np.testing.assert_allclose(gs.cpu().numpy(), -0.5 + 0.02 / (3 * n) * np.exp(scale.astype(np.float64)), rtol=1e-5, atol=1e-12)


def test_sharded_adam_device_path(gpu_device):
    """ShardedAdam's HIP path (one dvs_adam_step_groups launch over the sub-ranges of the slice) against the textbook recurrences;
    world 1 here — the two-rank reduce-scatter / all-gather plumbing is covered by tests/test_parallel.py on gloo."""
    import torch
    from divshot_amd.parallel import ShardedAdam
    n = 1001
    sizes = [3 * n, 3 * n, 4 * n, n, 3 * n, 45 * n]
    lrs = [1e-2, 5e-3, 1e-3, 5e-2, 2.5e-3, 1.25e-4]
    total = sum(sizes)
    rng = np.random.default_rng(4)
    p0 = rng.standard_normal(total).astype(np.float32)
    params = torch.tensor(p0, device=gpu_device)
    opt = ShardedAdam(params, sizes, lrs, 1, 0, eps=1e-8)
    ref = p0.astype(np.float64); m = np.zeros(total); v = np.zeros(total)
    lr_vec = np.concatenate([np.full(s_, lr) for s_, lr in zip(sizes, lrs)])
    for t in (1, 2, 3):
        g = rng.standard_normal(total).astype(np.float32)
        opt.step(torch.tensor(g, device=gpu_device))
        m = 0.9 * m + 0.1 * g; v = 0.999 * v + 0.001 * g.astype(np.float64) ** 2
        ref = ref - lr_vec * (m / (1 - 0.9 ** t)) / (np.sqrt(v / (1 - 0.999 ** t)) + 1e-8)
    np.testing.assert_allclose(params.cpu().numpy(), ref, rtol=2e-5, atol=2e-6)


def test_mcmc_edge_cases(gpu_device):
    """dvs_mcmc_relocate / dvs_mcmc_grow at the corners: no dead splat (no-op), every splat dead (nothing to draw from: no-op),
    zero growth, a single live splat drawn for every new slot (the ratio saturates at 51)."""
    import ctypes as C
    import torch
    from divshot_amd._lib import lib, check, McmcSets
    n, cap, min_op = 300, 400, 0.005
    widths = [3, 3, 45, 1, 3, 4]
    rng = np.random.default_rng(3)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def setup(logits):
        par = []
        for g, w in enumerate(widths):
            a = np.zeros((cap, w), np.float32); a[:n] = rng.standard_normal((n, w))
            if g == 3:
                a[:n, 0] = logits
            if g == 4:
                a[:n] = rng.normal(-3, 0.2, (n, 3))
            par.append(torch.tensor(a.reshape(-1), device=gpu_device))
        sets = McmcSets()
        for g in range(6):
            sets.param[g] = par[g].data_ptr()
        scratch = torch.empty(int(lib.dvs_mcmc_scratch_bytes(cap)), dtype=torch.uint8, device=gpu_device)
        check(lib.dvs_mcmc_init_scratch(st, scratch.data_ptr(), cap))
        return par, sets, scratch

    # no dead splat, and zero growth: nothing changes
    par, sets, scratch = setup(np.full(n, 1.0, np.float32))
    before = [p.clone() for p in par]
    nd = torch.full((1,), 77, dtype=torch.int32).pin_memory()
    check(lib.dvs_mcmc_relocate(st, n, C.byref(sets), min_op, 1, 0, scratch.data_ptr(), cap, nd.data_ptr()))
    check(lib.dvs_mcmc_grow(st, n, 0, C.byref(sets), min_op, 2, 0, scratch.data_ptr(), cap))
    torch.cuda.synchronize()
    assert int(nd[0]) == 0 and all(torch.equal(a, b) for a, b in zip(par, before))
    # every splat dead: nothing alive to draw from -> untouched
    par, sets, scratch = setup(np.full(n, -9.0, np.float32))
    before = [p.clone() for p in par]
    check(lib.dvs_mcmc_relocate(st, n, C.byref(sets), min_op, 1, 0, scratch.data_ptr(), cap, nd.data_ptr()))
    check(lib.dvs_mcmc_grow(st, n, 50, C.byref(sets), min_op, 2, 0, scratch.data_ptr(), cap))
    torch.cuda.synchronize()
    assert int(nd[0]) == n and all(torch.equal(a, b) for a, b in zip(par, before))
    # one live splat: it is the source of all 100 new slots; the relocation ratio saturates at 51
    logits = np.full(n, -9.0, np.float32); logits[17] = 2.0
    par, sets, scratch = setup(logits)
    src_pos = par[0].view(cap, 3)[17].clone()
    check(lib.dvs_mcmc_grow(st, n, 100, C.byref(sets), min_op, 5, 0, scratch.data_ptr(), cap))
    torch.cuda.synchronize()
    assert torch.equal(par[0].view(cap, 3)[n:n + 100], src_pos.expand(100, 3))
    o = 1 / (1 + np.exp(-2.0)); want, _ = _relocation_np(o, 51, min_op)
    got = 1 / (1 + np.exp(-par[3].cpu().numpy()[[17, n, n + 99]].astype(np.float64)))
    np.testing.assert_allclose(got, want, rtol=1e-4)
    # capacity is enforced
    assert lib.dvs_mcmc_grow(st, n, cap - n + 1, C.byref(sets), min_op, 5, 0, scratch.data_ptr(), cap) != 0
