"""dvs_opts.grad_mode = DVS_GRAD_LINEAGE (include/dvs_raster.h): the backward of the rasterizer lineage the reference credits
(README.md:95) at the two points where the forward is not smooth — the 0.99 alpha cap and the clamped branch of the EWA Jacobian.
It is NOT the derivative of the forward there, so finite differences cannot pin it; it is pinned against the independent dense
PyTorch formulation (tests/golden/dense_ref.py) with a straight-through cap and a constant clamped coordinate, on a scene built so
that both points are exercised; and the size of the difference to DVS_GRAD_TRUE is measured."""
import os
import sys
import numpy as np
import pytest
import divshot_amd as dv
from util import KEYS

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))


def _scene():
    spec = dv.make_spec(400, 64, 48, sh_degree=2, seed=5, scale_log_offset=0.9)
    P = dv.synth_splats(spec)
    cam = dv.synth_camera(spec, 0)
    tgt = dv.synth_target(spec, 0)
    P = {k: v.copy() for k, v in P.items()}
    P["opacity"][:150] = 8.0                      # sigmoid -> 0.9997: the cap is hit within ~0.14 sigma of these centres ...
    P["scale"][:150] += 1.6                       # ... so make them wide enough for that disc to hold pixel centres
    P["pos"][150:220, 0] *= 1.9                   # outside the 1.3 tan_fov guard band: clamped Jacobian branch
    P["scale"][150:220] += 1.2                    # ... but large enough to reach into the image
    return P, cam, tgt


def test_oracle_lineage_matches_dense_autograd(oracle_mod):
    import torch
    import dense_ref
    P, cam, tgt = _scene()
    o = oracle_mod.Oracle(np.float64)
    img = o.forward(P, cam, sh_degree=2)
    assert ((o.get("flags") & 24) != 0)[o.get("radii") > 0].sum() >= 5, "no visible splat on the clamped Jacobian branch"
    dL = (img - tgt) / tgt[0].size
    g = {m: {k: v.copy() for k, v in o.backward(dL, grad_mode=m).items()} for m in (0, 1)}
    diff = {k: float(np.linalg.norm(g[1][k] - g[0][k]) / np.linalg.norm(g[0][k])) for k in KEYS}
    assert diff["opacity"] > 1e-4 and diff["pos"] > 1e-3, diff          # the modes really differ on this scene
    assert diff["sh0"] == 0.0 and diff["shN"] == 0.0                    # colour gradients do not depend on the mode
    for mode in (0, 1):
        img_t, leaves = dense_ref.render(P, cam, sh_degree=2, grad_mode=mode)
        ok = ~o.get("fragile").astype(bool)
        assert np.abs(img_t.detach().numpy() - img)[:, ok].max() < 1e-9
        (0.5 * ((img_t - torch.tensor(tgt, dtype=torch.float64)) ** 2).sum() / tgt[0].size).backward()
        for k in KEYS:
            gt = leaves[k].grad.numpy().reshape(g[mode][k].shape)
            rel = np.abs(gt - g[mode][k]).max() / (np.abs(g[mode][k]).max() + 1e-300)
            assert rel < 1e-7, (mode, k, rel)


def test_lineage_fp32_oracle_close_to_fp64(oracle_mod):
    P, cam, tgt = _scene()
    o32, o64 = oracle_mod.Oracle(np.float32), oracle_mod.Oracle(np.float64)
    img = o64.forward(P, cam, sh_degree=2)
    o32.forward(P, cam, sh_degree=2)
    dL = (img - tgt) / tgt[0].size
    g64, g32 = o64.backward(dL, grad_mode=1), o32.backward(dL, grad_mode=1)
    for k in KEYS:
        l2 = np.linalg.norm((g32[k] - g64[k]).ravel()) / np.linalg.norm(g64[k].ravel())
        assert l2 < 1e-3, (k, l2)       # includes the few threshold-fragile pixels


@pytest.mark.gpu
@pytest.mark.parametrize("variant", ["reduce", "blocks", "mm", "tr"] if os.environ.get("DVS_TEST_ALL_VARIANTS") == "1" else ["blocks", "tr"])
def test_hip_lineage_mode_on_saturating_scene(gpu_device, oracle_mod, variant):
    """The HIP path in both gradient modes on the scene that really exercises them (pixels on the 0.99 cap, splats on the clamped
    Jacobian branch — the seeded configurations of test_gpu_parity.py hardly do: their measured lineage-vs-true difference is 0 to 1.6e-4),
    every A8 kernel variant, against the fp64 oracle of the same mode; and the two modes differ on the device as they do in the oracle."""
    import torch
    from divshot_amd.raster import Rasterizer, params_to_device
    P, cam, tgt = _scene()
    n = P["pos"].shape[0]
    o = oracle_mod.Oracle(np.float64)
    img64 = o.forward(P, cam, sh_degree=2)
    r = Rasterizer(0, max_splats=n, max_w=cam.width, max_h=cam.height)
    r.set_backward_variant(variant)
    Pd = params_to_device(P, r.tdev)
    got = {}
    for mode in (0, 1):
        img = r.forward(Pd, cam, sh_degree=2, absgrad=True, grad_mode=mode)
        dL = (img64 - tgt) / tgt[0].size
        g = r.backward(torch.from_numpy(dL.astype(np.float32)).to(r.tdev))
        torch.cuda.synchronize()
        ref = o.backward(dL, grad_mode=mode)
        for k in KEYS:
            a, b = g[k].double().cpu().numpy(), ref[k]
            l2 = np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-300)
            assert l2 < 2e-4, (variant, mode, k, l2)        # (this scene's huge opaque splats put many pixels near thresholds)
        got[mode] = {k: g[k].double().cpu().numpy() for k in KEYS}
    d_pos = np.linalg.norm(got[1]["pos"] - got[0]["pos"]) / np.linalg.norm(got[0]["pos"])
    d_opa = np.linalg.norm(got[1]["opacity"] - got[0]["opacity"]) / np.linalg.norm(got[0]["opacity"])
    assert d_pos > 1e-3 and d_opa > 1e-4, (d_pos, d_opa)
    r.close()


def test_decision_replay_reproduces_the_recorded_run(oracle_mod):
    """The machinery the GPU parity suite uses to hold every splat to the strict bar, on CPU: the fp32 oracle
    records which pixel took which list entry; the fp64 oracle, replaying exactly those decisions, reproduces n_contrib exactly and
    the image / gradients to fp32 accuracy — also on a scene of huge splats where its own decisions differ at fragile pixels."""
    from util import scene
    Oracle = oracle_mod.Oracle
    for (n, W, H, deg, seed, soff) in [(3000, 96, 96, 1, 5, 1.5), (2000, 64, 64, 3, 1, 0.0)]:
        spec, P, cam, tgt = scene(n, W, H, deg, seed, scale_offset=soff)
        o32 = Oracle(np.float32); o32.record_masks(True)
        img32 = o32.forward(P, cam, sh_degree=deg).copy()
        masks = o32.get("take_masks")
        assert masks.shape == (o32.get("vals").size, 4) and masks.any()
        o64 = Oracle(np.float64)
        own = o64.forward(P, cam, sh_degree=deg).copy()
        if not np.array_equal(o64.get("vals"), o32.get("vals")):
            continue                                   # fp64 bins differently: the GPU test replays in fp32 then
        o64.set_replay(masks)
        img64 = o64.forward(P, cam, sh_degree=deg).copy()
        assert np.array_equal(o64.get("n_contrib"), o32.get("n_contrib"))
        np.testing.assert_allclose(img64, img32, rtol=1e-4, atol=2e-6)
        dL = np.random.default_rng(3).normal(size=img32.shape).astype(np.float32)
        g32, g64 = o32.backward(dL), o64.backward(dL.astype(np.float64))
        for k in g32:
            ref = g64[k]
            assert np.abs(g32[k] - ref).max() <= 1e-4 * np.abs(ref).max(), k
        o64.set_replay(None)
        again = o64.forward(P, cam, sh_degree=deg)
        assert np.array_equal(again, own)              # replay off: the oracle's own decisions again


def test_set_lists_composites_over_foreign_tile_lists(oracle_mod):
    """Oracle.set_lists (round 6; what the GPU parity suite uses at C3 / C5 size, where a radius on an integer boundary makes float64 bin one
    splat differently from float32): the float64 oracle composites over the FLOAT32 run's tile lists and replays its decisions. Forced here
    with a splat whose float32 radius is one more than its float64 radius would be hard to construct, so the lists are made foreign another
    way that must not change a single pixel: every tile's list gets the same entries as the oracle's own, handed over explicitly — image,
    n_contrib and gradients must be bit-identical to the run that binned by itself; then a list with one entry REMOVED changes exactly the
    pixels of that tile; a listed splat that the run culls, and ranges of another image size, are refused."""
    from util import scene
    Oracle = oracle_mod.Oracle
    spec, P, cam, tgt = scene(1500, 80, 48, 2, 9)
    o = Oracle(np.float64)
    img = o.forward(P, cam, sh_degree=2).copy()
    vals, ranges, nc = o.get("vals").copy(), o.get("ranges").copy(), o.get("n_contrib").copy()
    dL = np.random.default_rng(1).normal(size=img.shape)
    g = {k: v.copy() for k, v in o.backward(dL).items()}
    f = Oracle(np.float64)
    f.set_lists(vals, ranges)
    img_f = f.forward(P, cam, sh_degree=2)
    assert np.array_equal(img_f, img) and np.array_equal(f.get("n_contrib"), nc) and f.get("keys").size == 0
    gf = f.backward(dL)
    for k in g:
        assert np.array_equal(gf[k], g[k]), k
    # the float32 oracle's lists under the float64 composite: same lists here (small scene), so again the same image
    o32 = Oracle(np.float32); o32.forward(P, cam, sh_degree=2)
    if np.array_equal(o32.get("vals"), vals):
        f.set_lists(o32.get("vals"), o32.get("ranges"))
        assert np.array_equal(f.forward(P, cam, sh_degree=2), img)
    # drop the first entry of the fullest tile: only that tile's pixels may change
    t = int(np.argmax(ranges[:, 1] - ranges[:, 0]))
    r2 = ranges.copy(); r2[t, 0] += 1
    f.set_lists(vals, r2)
    img2 = f.forward(P, cam, sh_degree=2)
    tx, ty = t % 5, t // 5
    changed = np.abs(img2 - img).max(axis=0) > 0
    assert changed.any() and not changed[:ty * 16].any() and not changed[(ty + 1) * 16:].any() and not changed[:, :tx * 16].any() and not changed[:, (tx + 1) * 16:].any()
    # refusals
    culled = np.where(o.get("radii") == 0)[0]
    if culled.size:
        bad = vals.copy(); bad[0] = culled[0]
        f.set_lists(bad, ranges)
        with pytest.raises(AssertionError, match="culled in this precision"):
            f.forward(P, cam, sh_degree=2)
    f.set_lists(vals, ranges[:-1])
    with pytest.raises(AssertionError, match="not this image"):
        f.forward(P, cam, sh_degree=2)
    f.set_lists(None, None)
    assert np.array_equal(f.forward(P, cam, sh_degree=2), img) and f.get("keys").size == vals.size
