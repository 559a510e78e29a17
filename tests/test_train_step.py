"""A10 — `train_step` (gs_train.cpp:156) at step level: K iterations of libgstrain.so, driven through the host's own call sequence by
`gaussian_train`, against tests/train_step_ref.py (CPU oracle gradients in DVS_GRAD_LINEAGE + numpy Adam on the same cameras), and one
ADC refinement against the restated decision rule. CPU part: the restatement itself (loss falls, float32 and float64 agree)."""
import ctypes as C
import os
import re
import subprocess
import numpy as np
import pytest
import divshot_amd as dv
from oracle.oracle import Oracle
from train_step_ref import TrainStepRef, KEYS, camera_stream, ssim_and_grad

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "divshot_amd", "lib")
DRIVER = os.path.join(LIB, "gaussian_train")
PLUGIN = os.path.join(LIB, "libgstrain.so")
WIDTH = {"pos": 3, "sh0": 3, "shN": 45, "opacity": 1, "scale": 3, "rot": 4}


def _read_ply(path):
    head = open(path, "rb").read(4096).split(b"end_header\n")[0].decode()
    n = int(re.search(r"element vertex (\d+)", head).group(1))
    lib = C.CDLL(PLUGIN)
    lib.gstrain_read_ply.restype = C.c_int64
    lib.gstrain_read_ply.argtypes = [C.c_char_p] + [C.c_void_p] * 6 + [C.c_uint64]
    A = {k: np.zeros((n, WIDTH[k]), np.float32) for k in KEYS}
    assert lib.gstrain_read_ply(path.encode(), *[A[k].ctypes.data for k in KEYS], n) == n
    A["shN"] = A["shN"].reshape(n, 15, 3)
    A["opacity"] = A["opacity"].reshape(n)
    return A


def _write_ply(path, A):
    lib = C.CDLL(PLUGIN)
    lib.gstrain_write_ply.restype = C.c_int
    lib.gstrain_write_ply.argtypes = [C.c_char_p, C.c_uint64] + [C.c_void_p] * 6 + [C.c_int]
    arrs = [np.ascontiguousarray(A[k], np.float32).reshape(A["pos"].shape[0], -1) for k in KEYS]
    assert lib.gstrain_write_ply(path.encode(), A["pos"].shape[0], *[a.ctypes.data for a in arrs], 0) == 0


def _scene(n, W, H, cams, sh, seed):
    spec = dv.make_spec(n, W, H, sh_degree=sh, n_cams=cams, seed=seed)
    return spec, [dv.synth_camera(spec, i) for i in range(cams)]


def test_camera_stream_is_the_plugins_xorshift():
    # first draws of xorshift64 (13, 7, 17) from 88172645463325252, modulo 4 cameras — fixed by the algorithm, checked against a C run
    s = camera_stream(4, 6)
    assert len(s) == 6 and all(0 <= c < 4 for c in s)
    assert camera_stream(4, 6, single_camera=True) == [0] * 6
    x = 88172645463325252
    x ^= (x << 13) & (2 ** 64 - 1); x ^= x >> 7; x ^= (x << 17) & (2 ** 64 - 1)
    assert s[0] == x % 4


def test_restated_train_step_learns_and_is_well_conditioned():
    """The restatement on CPU: targets = oracle renders of the generating scene, start = a perturbed copy; the L1 loss falls and the
    float32 trajectory follows the float64 one on the bulk of the elements (the rest is what Adam's eps = 1e-15 does to rounding noise)."""
    spec, cams = _scene(600, 48, 48, 3, 1, 4)
    gt = dv.synth_splats(spec)
    o = Oracle(np.float32)
    targets = [o.forward(gt, c, sh_degree=1).copy() for c in cams]
    rng = np.random.default_rng(0)
    init = {k: v.copy() for k, v in gt.items()}
    init["sh0"] = init["sh0"] + 0.5 * rng.uniform(-1, 1, init["sh0"].shape).astype(np.float32)
    init["opacity"] = init["opacity"] - 1.0
    init["shN"] = np.zeros_like(init["shN"])
    r32 = TrainStepRef(Oracle, cams, targets, init, 1, 30, np.float32)
    r64 = TrainStepRef(Oracle, cams, targets, init, 1, 30, np.float64)
    for _ in range(12):
        r32.train_step(); r64.train_step()
    assert np.mean(r32.losses[-3:]) < 0.9 * np.mean(r32.losses[:3])
    np.testing.assert_allclose(r32.losses, r64.losses, rtol=2e-4)
    for k in ("sh0", "opacity"):
        moved = np.abs(r64.P[k] - init[k]) > 0
        close = np.abs(r32.P[k] - r64.P[k]) <= 1e-4 * np.maximum(np.abs(r64.P[k]), 1e-2)
        assert close[moved].mean() > 0.9, (k, close[moved].mean())


def test_restated_ssim_gradient_against_finite_differences():
    """The analytic d(mean SSIM)/dx of tests/train_step_ref.py (what the SSIM term of the restated train_step uses) against fp64 central
    differences of its own forward value, and float32 against float64."""
    rng = np.random.default_rng(0)
    H, W = 23, 31
    y = rng.uniform(0, 1, (3, H, W))
    x = np.clip(y + 0.15 * rng.standard_normal((3, H, W)), 0, 1)
    v, g = ssim_and_grad(x, y)
    assert 0.5 < v < 1.0 and abs(ssim_and_grad(y, y)[0] - 1.0) < 1e-12
    for _ in range(40):
        c, i, j = rng.integers(0, 3), rng.integers(0, H), rng.integers(0, W)
        e = 1e-6
        xp = x.copy(); xp[c, i, j] += e
        xm = x.copy(); xm[c, i, j] -= e
        fd = (ssim_and_grad(xp, y)[0] - ssim_and_grad(xm, y)[0]) / (2 * e)
        assert abs(g[c, i, j] - fd) <= 1e-4 * abs(fd) + 1e-10, (c, i, j, g[c, i, j], fd)
    v32, g32 = ssim_and_grad(x.astype(np.float32), y.astype(np.float32))
    assert abs(v32 - v) < 1e-6 and np.abs(g32 - g).max() < 1e-5 * np.abs(g).max()


def _run(args, timeout=600, env=None):
    p = subprocess.run([DRIVER] + args, capture_output=True, text=True, timeout=timeout, env=env)
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-3000:]
    return p


def _hip_targets(spec, cams, sh):
    """The plugin's training views: the generating scene rendered by the HIP forward (same call, same options as load_synthetic)."""
    import torch
    from divshot_amd.raster import Rasterizer, params_to_device
    gt = dv.synth_splats(spec)
    r = Rasterizer(0, max_splats=spec.n, max_w=spec.width, max_h=spec.height)
    P = params_to_device(gt, torch.device("cuda", 0))
    out = [r.forward(P, c, sh_degree=sh).detach().cpu().numpy().copy() for c in cams]
    r.close()
    return out


COMMON = ["--ssim", "0", "--packLevel", "0", "--densifyStrategy", "0", "--progressTrain", "0", "--absgrad", "1"]


def _compare(got, r32, r64, init, rtol, min_fraction, report, worst_rtol=None, quantile=1.0):
    """every pinned element within worst_rtol (default rtol), the `quantile` of them within rtol"""
    worst_rtol = rtol if worst_rtol is None else worst_rtol
    for k in KEYS:
        ref = r64.P[k].reshape(got[k].shape)
        a32 = r32.P[k].reshape(got[k].shape).astype(np.float64)
        den = np.maximum(np.abs(ref), 1e-2)
        comparable = np.abs(a32 - ref) <= 0.25 * rtol * den           # the float32 restatement itself holds the bar with margin
        err = np.abs(got[k].astype(np.float64) - ref) / den
        frac = comparable.mean()
        moved = np.abs(ref - init[k].reshape(ref.shape)) > 0
        report[k] = dict(comparable=float(frac), worst=float(err[comparable].max()), moved=float(moved.mean()),
                         rel_l2_of_update=float(np.linalg.norm((got[k] - ref)[moved]) / max(np.linalg.norm((ref - init[k].reshape(ref.shape))[moved]), 1e-30)))
        report[k]["quantile_%g" % quantile] = float(np.quantile(err[comparable], quantile))
        assert frac >= min_fraction, (k, report[k])
        assert err[comparable].max() <= worst_rtol and report[k]["quantile_%g" % quantile] <= rtol, (k, report[k])


@pytest.mark.gpu
def test_plugin_trajectory_matches_oracle_plus_numpy_adam(tmp_path):
    """20 iterations of the product (L1 only, no refinement, fixed camera stream) vs oracle gradients + numpy Adam: every parameter the
    float32 restatement can pin is within 1e-4 (relative, floor 1e-2) of the float64 trajectory, >= 90 % of each group is pinned, and the
    UPDATE (final - initial) agrees to 1 % in relative L2 over everything that moved."""
    n, W, H, ncam, sh, seed, K = 2000, 64, 64, 4, 1, 11, 20
    src = f"synthetic:N={n},W={W},H={H},cams={ncam},sh={sh},seed={seed}"
    out = str(tmp_path / "m" / "it")
    flags = COMMON + ["--warmupLength", "100000"]
    _run(["--inputPath", src, "--maxIteration", "0", "--outputPath", out] + flags)
    init = _read_ply(out + "_0.ply")
    p = _run(["--inputPath", src, "--maxIteration", str(K), "--outputPath", out] + flags)
    got = _read_ply(out + f"_{K}.ply")
    assert got["pos"].shape[0] == n and "densify @" not in p.stderr
    spec, cams = _scene(n, W, H, ncam, sh, seed)
    targets = _hip_targets(spec, cams, sh)
    r32 = TrainStepRef(Oracle, cams, targets, init, sh, K, np.float32)
    r64 = TrainStepRef(Oracle, cams, targets, init, sh, K, np.float64)
    for _ in range(K):
        r32.train_step(); r64.train_step()
    report = {}
    _compare(got, r32, r64, init, 1e-4, 0.90, report)
    for k, r in report.items():
        assert r["rel_l2_of_update"] < 1e-2, (k, r)
    # the loss line the host prints (editor.cpp:1554 wording) is the mean L1 of the step's views
    m = re.search(r"Iteraions 0, loss : ([0-9.eE+-]+)", p.stderr)
    assert m and abs(float(m.group(1)) - r64.losses[0]) < 2e-4 * max(r64.losses[0], 1e-3)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    import json
    json.dump(report, open(os.path.join(ROOT, "gpurun_out", "train_step_parity.json"), "w"), indent=1)


@pytest.mark.gpu
def test_plugin_trajectory_on_the_cli_defaults_ssim_sh3_eight_views(tmp_path):
    """the loss the reference CLI defaults to — `--ssim 0.2` (main.cpp:24) — with SH degree 3 and BASELINE config
    C4's eight views per iteration (`--viewsPerIter 8`: ONE multi-view pass per train_step), 12 iterations (= 96 rendered views) of the
    product against oracle gradients + the restated SSIM / L1 gradient + numpy Adam (float32 and float64): the bar of the L1-only
    trajectory test on the elements the float32 restatement itself pins (>= 85 % of each group): 99.5 % of them within 1e-4, every one
    within 1e-3, and the 12-step update within 1e-2 in relative L2 (positions: 2e-3 — their update is a few float32 ulps of the coordinate; the other groups 2e-5)."""
    n, W, H, ncam, sh, seed, K, V, w = 2000, 64, 64, 8, 3, 21, 12, 8, 0.2
    src = f"synthetic:N={n},W={W},H={H},cams={ncam},sh={sh},seed={seed}"
    out = str(tmp_path / "m" / "it")
    flags = ["--ssim", str(w), "--packLevel", "0", "--densifyStrategy", "0", "--progressTrain", "0", "--absgrad", "1", "--warmupLength", "100000",
             "--viewsPerIter", str(V)]
    _run(["--inputPath", src, "--maxIteration", "0", "--outputPath", out] + flags)
    init = _read_ply(out + "_0.ply")
    p = _run(["--inputPath", src, "--maxIteration", str(K), "--outputPath", out] + flags)
    got = _read_ply(out + f"_{K}.ply")
    assert got["pos"].shape[0] == n and "densify @" not in p.stderr
    spec, cams = _scene(n, W, H, ncam, sh, seed)
    targets = _hip_targets(spec, cams, sh)
    r32 = TrainStepRef(Oracle, cams, targets, init, sh, K, np.float32, views_per_step=V, ssim_weight=w)
    r64 = TrainStepRef(Oracle, cams, targets, init, sh, K, np.float64, views_per_step=V, ssim_weight=w)
    for _ in range(K):
        r32.train_step(); r64.train_step()
    assert np.abs(r64.P["shN"][:, 8:] - init["shN"][:, 8:]).max() > 0            # the degree-3 band trains
    report = {}
    # 99.5 % of the pinned elements within 1e-4, every one within 1e-3: the SSIM term's convolutions give the photometric gradient a
    # float32 noise floor that Adam's normalisation (eps 1e-15) turns into a few outliers the float32 restatement does not share — and
    # which differ from run to run with the order of the composite backward's atomics (five runs on two boxes: the worst element — one opacity — 1.9e-4 in one run, 4.1e-4 in four)
    _compare(got, r32, r64, init, 1e-4, 0.85, report, worst_rtol=1e-3, quantile=0.995)
    for k, r in report.items():
        assert r["rel_l2_of_update"] < 1e-2, (k, r)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    import json
    json.dump(report, open(os.path.join(ROOT, "gpurun_out", "train_step_parity_ssim_sh3_8views.json"), "w"), indent=1)


@pytest.mark.gpu
def test_plugin_mcmc_refinement_against_the_restated_rule(tmp_path):
    """the strategy the reference CLI defaults to — `--densifyStrategy 1` (MCMC, main.cpp:20,29). Ten iterations
    (L1 + the strategy's opacity / scale regularisers, exploration noise off so that the trajectory can be restated) end in ONE
    relocation + 5 % growth step; its effect is decoded from the model the plugin saved and checked against the published rule
    (restated in tests/test_gpu_train_ops.py::_relocation_np and by its invariant):
      * every splat the restated trajectory calls dead (sigmoid(opacity) <= 0.005, 5 % margin; a tenth of the model is started there
        through --load_itr) has been moved onto a live splat,
        floor(1.05 n) - n copies have been appended, every copy carries its source's sh0 / shN / rotation;
      * a source and its copies share one position; over each such group 1 - prod(1 - o') equals the source's opacity BEFORE the step
        (each application of the rule replaces a factor (1 - o) by r factors (1 - o)^(1/r)), and its members' scales have shrunk;
      * sources are drawn in proportion to opacity; splats that are neither dead, nor drawn, nor copies carry the trajectory's
        parameters."""
    n, W, H, ncam, sh, seed, K, min_op = 3000, 96, 96, 4, 1, 13, 10, 0.005
    src = f"synthetic:N={n},W={W},H={H},cams={ncam},sh={sh},seed={seed}"
    out = str(tmp_path / "m" / "it")
    flags = ["--ssim", "0", "--packLevel", "0", "--densifyStrategy", "1", "--progressTrain", "0", "--absgrad", "1", "--noiselr", "0",
             "--warmupLength", "5", "--refineEvery", "10", "--refineStopIter", "1000"]
    _run(["--inputPath", src, "--maxIteration", "0", "--outputPath", out] + flags)
    init = _read_ply(out + "_0.ply")
    # the host's --minOpacity never reaches the trainer (gs_train.cpp:65 is commented out), so the rule's threshold is the default 0.005:
    # a tenth of the splats start below it (logit -8), through the host's own resume path (--load_itr, gs_train.cpp:113)
    rng = np.random.default_rng(3)
    kill = rng.random(n) < 0.1
    init["opacity"][kill] = -8.0
    _write_ply(out + "_0.ply", init)
    p = _run(["--inputPath", src, "--maxIteration", str(K), "--outputPath", out, "--load_itr", "0"] + flags)
    n_new = int(1.05 * n) - n
    m = re.search(r"mcmc @10: (\d+) -> (\d+) splats", p.stderr)
    assert m and int(m.group(1)) == n and int(m.group(2)) == n + n_new, p.stderr[-1500:]
    got = _read_ply(out + f"_{K}.ply")
    assert got["pos"].shape[0] == n + n_new
    spec, cams = _scene(n, W, H, ncam, sh, seed)
    targets = _hip_targets(spec, cams, sh)
    r64 = TrainStepRef(Oracle, cams, targets, init, sh, K, np.float64, mcmc_reg=(0.01, 0.01))
    for _ in range(K):
        r64.train_step()
    sig = lambda a: 1.0 / (1.0 + np.exp(-np.asarray(a, np.float64)))
    o_pre = sig(r64.P["opacity"])
    dead = o_pre <= min_op
    firm = np.abs(o_pre - min_op) > 0.05 * min_op
    assert 0.03 * n < dead.sum() < 0.6 * n
    # groups of identical positions in the saved model: a source and its copies
    key = {}
    for row in range(n + n_new):
        key.setdefault(got["pos"][row].tobytes(), []).append(row)
    where_pre = np.abs(got["pos"][:n].astype(np.float64) - r64.P["pos"]).max(1) < 1e-3 * np.maximum(np.abs(r64.P["pos"]).max(1), 1e-2)
    # (row i < n still sits where the trajectory has it <=> it was not relocated)
    assert not where_pre[dead & firm].any(), "a dead splat was left in place"
    assert where_pre[~dead & firm].all(), "a live splat was moved"
    checked, src_rows = 0, []
    for rows in key.values():
        srcs = [r_ for r_ in rows if r_ < n and where_pre[r_] and not dead[r_]]
        if len(rows) == 1:
            continue
        assert len(srcs) == 1, (rows, srcs)                        # exactly one member is the live original
        i = srcs[0]
        src_rows.append(i)
        for r_ in rows:                                            # copies carry the source's colour and rotation
            for k in ("sh0", "shN", "rot"):
                a, b = got[k][r_].astype(np.float64).ravel(), r64.P[k][i].ravel()
                assert np.abs(a - b).max() <= 1e-3 * max(np.abs(b).max(), 1e-2), (k, i, r_)
            assert (got["scale"][r_] <= r64.P["scale"][i] + 1e-3).all()
        o_after = sig(got["opacity"][rows])
        if firm[i] and o_after.min() > 1.02 * min_op:              # (an opacity clamped at the floor of the rule, minOpacity, breaks the product)
            mass = 1.0 - np.prod(1.0 - o_after)
            assert abs(mass - o_pre[i]) <= 2e-3 * max(o_pre[i], 0.05), (i, rows, mass, o_pre[i])
            checked += 1
    assert checked > 0.3 * len(src_rows) > 0, (checked, len(src_rows))
    # every relocated or appended row belongs to a group with a source
    grouped = {r_ for rows in key.values() if len(rows) > 1 for r_ in rows}
    assert all((r_ in grouped) for r_ in list(np.flatnonzero(dead & firm)) + list(range(n, n + n_new)))
    live = ~dead
    assert o_pre[src_rows].mean() > 1.1 * o_pre[live].mean()        # drawn ~ opacity
    untouched = np.array([r_ for r_ in range(n) if r_ not in grouped and firm[r_]])
    assert untouched.size > 0.2 * n
    for k in ("pos", "sh0", "scale", "rot", "opacity"):
        a, b = got[k][untouched].astype(np.float64), r64.P[k][untouched]
        bad = np.abs(a - b) > 1e-3 * np.maximum(np.abs(b), 1e-2)
        assert bad.mean() < 0.02, (k, bad.mean())


def _decode_actions(src, dst):
    """Which ADC action each source splat took, read off the compacted output: the rotation row is copied unchanged by keep / clone /
    split, the scale tells clone from split (densify.hip k_densify_apply)."""
    n, o, act = src["rot"].shape[0], 0, []
    nd = dst["rot"].shape[0]
    same = lambda a, b: np.abs(a - b).max() < 2e-3
    for i in range(n):
        if o < nd and same(dst["rot"][o], src["rot"][i]):
            if o + 1 < nd and same(dst["rot"][o + 1], src["rot"][i]) and not (i + 1 < n and same(src["rot"][i + 1], src["rot"][i])):
                act.append(2 if np.abs(dst["scale"][o] - (src["scale"][i] - np.log(1.6))).max() < 2e-3 else 1)
                o += 2
            else:
                act.append(0); o += 1
        else:
            act.append(3)
    assert o == nd, (o, nd)
    return np.array(act)


@pytest.mark.gpu
def test_plugin_adc_refinement_matches_the_restated_rule(tmp_path):
    """Ten iterations ending in ONE refinement (densifyStrategy 0, abs-grad statistics over the ten views): the actions the plugin took,
    decoded from the compacted model it saved, equal the restated rule on the oracle trajectory wherever that decision has a 5 % margin;
    kept splats carry the trajectory's parameters, split children the parent's scale - log 1.6, clones / splits the revised opacity."""
    n, W, H, ncam, sh, seed, K = 3000, 96, 96, 4, 1, 12, 10
    grow = 8.0e-4             # the median of the ten-view statistic on this scene: about half the splats split (none is small enough to clone:
                              # exp(scale) > 0.01 x extent everywhere; the clone branch is covered by tests/test_gpu_train_ops.py)
    src = f"synthetic:N={n},W={W},H={H},cams={ncam},sh={sh},seed={seed}"
    out = str(tmp_path / "m" / "it")
    flags = COMMON + ["--warmupLength", "5", "--refineEvery", "10", "--refineStopIter", "1000", "--growGrad2d", str(grow)]
    _run(["--inputPath", src, "--maxIteration", "0", "--outputPath", out] + flags)
    init = _read_ply(out + "_0.ply")
    p = _run(["--inputPath", src, "--maxIteration", str(K), "--outputPath", out] + flags)
    m = re.search(r"densify @10: (\d+) -> (\d+) splats", p.stderr)
    assert m and int(m.group(1)) == n, p.stderr[-1500:]
    got = _read_ply(out + f"_{K}.ply")
    assert got["pos"].shape[0] == int(m.group(2))
    spec, cams = _scene(n, W, H, ncam, sh, seed)
    targets = _hip_targets(spec, cams, sh)
    r64 = TrainStepRef(Oracle, cams, targets, init, sh, K, np.float64)
    for _ in range(K):
        r64.train_step()
    want, margin = r64.adc_actions(grow)
    pre = {k: r64.P[k].astype(np.float32) for k in KEYS}
    act = _decode_actions(pre, got)
    firm = margin > 0.05
    assert firm.mean() > 0.9 and {0, 2} <= set(want[firm].tolist()) and min(np.bincount(want, minlength=4)[[0, 2]]) > 0.2 * n, (firm.mean(), np.bincount(want, minlength=4))
    assert np.array_equal(act[firm], want[firm]), (np.flatnonzero(act[firm] != want[firm])[:10], np.bincount(want, minlength=4), np.bincount(act, minlength=4))
    cnt = np.where(act == 3, 0, np.where(act == 0, 1, 2))
    off = np.concatenate([[0], np.cumsum(cnt)[:-1]])
    keep, clone, split = [np.flatnonzero(act == a) for a in (0, 1, 2)]
    for k in ("pos", "sh0", "scale", "rot", "opacity"):
        a, b = got[k][off[keep]].astype(np.float64), r64.P[k][keep]
        bad = np.abs(a - b) > 1e-3 * np.maximum(np.abs(b), 1e-2)
        assert bad.mean() < 0.02, (k, bad.mean())                     # (the few elements Adam's eps makes incomparable)
    sig = 1.0 / (1.0 + np.exp(-r64.P["opacity"]))
    for idx in (clone, split):                                       # revisedOpacity (CLI default on): both copies take 1 - sqrt(1 - o)
        for c in (0, 1):
            o_new = 1.0 / (1.0 + np.exp(-got["opacity"][off[idx] + c].astype(np.float64)))
            np.testing.assert_allclose(o_new, np.clip(1.0 - np.sqrt(1.0 - sig[idx]), 1e-6, 1 - 1e-6), rtol=5e-3, atol=1e-6)
    for c in (0, 1):
        np.testing.assert_allclose(got["scale"][off[split] + c], r64.P["scale"][split] - np.log(1.6), rtol=2e-3, atol=2e-3)
        d = np.abs(got["pos"][off[split] + c] - r64.P["pos"][split]).max(1)
        assert (d <= 6.0 * np.exp(r64.P["scale"][split].max(1)) + 1e-4).all()
    if clone.size:
        assert np.abs(got["pos"][off[clone] + 1] - r64.P["pos"][clone]).max() < 1e-3
