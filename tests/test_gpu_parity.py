"""GPU parity: every stage of the HIP path against the CPU oracle, through the C-ABI.

Bars: radii / tiles_touched / sorted (key,value) / tile ranges bit-exact; preprocess floats bit-exact
(same IEEE op order, -ffp-contract=off); RGB, final_T within 1e-4 rel on pixels the oracle does not
flag as threshold-fragile, n_contrib exact there; the five gradient groups (and the A8 intermediates)
against the fp64 oracle: every element within 1e-4 rel + 1e-5 of the group's max |value| (absolute floor
for sums that cancel: a per-splat gradient is a sum of hundreds of signed per-pixel terms accumulated in
fp32), and the whole group within 1e-5 relative L2 error.
"""
import ctypes as C
import json
import os
import numpy as np
import pytest
import divshot_amd as dv
from util import scene, rel_close, KEYS

pytestmark = pytest.mark.gpu

CONFIGS = {
    # name: (n, W, H, deg, seed, scale_offset, antialias, bg)
    "C1_10k_256_deg0": (10_000, 256, 256, 0, 1, 0.0, False, (0, 0, 0)),
    "G2_2k_64_deg3": (2_000, 64, 64, 3, 1, 0.0, False, (0.3, 0.1, 0.2)),
    "ragged_5k_250x130_deg2_aa": (5_000, 250, 130, 2, 7, 0.5, True, (0, 0, 0)),
    "dense_3k_96_big": (3_000, 96, 96, 1, 5, 1.5, False, (1, 1, 1)),
    "C2_100k_800_deg3": (100_000, 800, 800, 3, 1, 0.0, False, (0, 0, 0)),
}


def _random_configs(k=10):
    """Seeded sweep over the corners a fixed list misses: splat counts around the wave / tile-of-64 sizes, image sizes that are not
    multiples of 16, every SH degree, anti-aliasing, backgrounds, footprint scales from sub-pixel to several tiles."""
    rng = np.random.default_rng(20240907)
    out = {}
    counts = [1, 63, 64, 65, 257, 1000, 2049, 3500, 4096, 6000]
    for i in range(k):
        n = counts[i % len(counts)]
        W, H = int(rng.integers(17, 230)), int(rng.integers(17, 150))
        deg, aa = int(rng.integers(0, 4)), bool(rng.integers(0, 2))
        soff = float(rng.uniform(-0.7, 1.2))
        bg = tuple(float(x) for x in np.round(rng.uniform(0, 1, 3), 2))
        out[f"rnd{i}_{n}_{W}x{H}_deg{deg}{'_aa' if aa else ''}"] = (n, W, H, deg, 100 + i, soff, aa, bg)
    return out


CONFIGS.update(_random_configs())


@pytest.fixture(scope="module")
def rast(gpu_device):
    from divshot_amd.raster import Rasterizer
    r = Rasterizer(0, max_splats=1 << 20, max_w=1920, max_h=1080)
    r.keep_intermediates(True)          # stage-level parity of A8 reads the gradient rows after the backward
    yield r
    r.close()


def _run_gpu(rast, P, cam, tgt, deg, aa, absgrad=True, variants=None):
    """forward once; backward for every (grad_mode, A8 kernel variant) combination on the same upstream gradient"""
    import torch
    from divshot_amd.raster import params_to_device
    Pd = params_to_device(P, rast.tdev)
    # experiment libraries (DVS_RASTER_LIB=tools/xlib/..., DVS_TEST_ALL_VARIANTS=1) also carry the retired per-block A7 kernel: bit-identical
    # (same per-pixel sequence of contributing splats, same expressions). The release library has ONE forward.
    img_q = saved_q = None
    if ALL_VARIANTS:
        rast.set_forward_variant("blocks")
        img_q = rast.forward(Pd, cam, sh_degree=deg, antialias=aa, absgrad=absgrad).clone()
        torch.cuda.synchronize()
        saved_q = rast.saved()
        rast.set_forward_variant("quadrant")
    img = rast.forward(Pd, cam, sh_degree=deg, antialias=aa, absgrad=absgrad)
    torch.cuda.synchronize()
    img_h = img.cpu().numpy()
    saved = rast.saved()
    if img_q is not None:
        assert torch.equal(img, img_q), "A7 variants differ in the image"
        assert np.array_equal(saved["final_T"].view(np.uint32), saved_q["final_T"].view(np.uint32)) and np.array_equal(saved["n_contrib"], saved_q["n_contrib"])
    keys = rast.sorted_keys()
    dL = torch.from_numpy((img_h - tgt) / tgt[0].size).to(rast.tdev)
    runs = {}
    for mode in (0, 1):
        for variant in (variants or BWD_VARIANTS):
            rast.set_backward_variant(variant)
            rast._opts.grad_mode = mode
            grads = rast.backward(dL, want_mean2d=True)
            torch.cuda.synchronize()
            runs[(mode, variant)] = ({k: v.cpu().numpy() for k, v in grads.items()}, rast.bwd_intermediates())
    rast.set_backward_variant("tr")
    rast._opts.grad_mode = 0
    return img_h, saved, keys, runs, (img_h - tgt) / tgt[0].size


# The matrix checks the shipped A8 kernel ("tr") and its round-2 predecessor ("blocks": the cross-check with a different summation order,
# the only other A8 kernel in the release library). The retired experiments ("reduce", round 1; "mm", the MFMA contraction; the per-block
# A7) exist in experiment builds only (tools/xbuild.sh): DVS_TEST_ALL_VARIANTS=1 together with DVS_RASTER_LIB=<such a library> brings
# them back into the matrix.
ALL_VARIANTS = os.environ.get("DVS_TEST_ALL_VARIANTS") == "1"
BWD_VARIANTS = ("reduce", "blocks", "mm", "tr") if ALL_VARIANTS else ("blocks", "tr")
REPORT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "parity_report.jsonl")


@pytest.mark.parametrize("name", list(CONFIGS))
def test_pipeline_parity(rast, oracle_mod, name):
    check_pipeline_parity(rast, oracle_mod, name, CONFIGS[name])


def test_wide_image_takes_the_16_bit_rectangles(gpu_device, oracle_mod):
    """A tile grid of more than 255 columns (here 4208 x 48 pixels = 263 x 3 tiles) cannot pack a splat's tile rectangle into four
    bytes: A2 writes the 4 x u16 form and A3 / A4 run their DVS_FE_RECT_U16 instantiations (frontend.hip). Bins bit-exact against the
    oracle, preprocess floats bit-identical, image within 1e-4 outside the fragile pixels, gradients within 1e-3 relative L2 per group of
    the fp32 oracle (at this shape — a focal length of 3644 pixels — the oracle's own fp32-vs-fp64 replay agreement is 1.3e-4 on the
    positions, just over the 1e-4 test_pipeline_parity asks of it, so that test's decision-replay bars are not applied here)."""
    import torch
    from divshot_amd.raster import Rasterizer, params_to_device
    n, W, H, deg, seed, soff, bg = 4000, 4208, 48, 1, 41, 0.3, (0.1, 0.2, 0.3)
    spec, P, cam, tgt = scene(n, W, H, deg, seed, scale_offset=soff, bg=bg)
    r = Rasterizer(0, max_splats=8192, max_w=W, max_h=H)
    try:
        Pd = params_to_device(P, r.tdev)
        img = r.forward(Pd, cam, sh_degree=deg, absgrad=True)
        torch.cuda.synchronize()
        saved, keys = r.saved(), r.sorted_keys()
        o = oracle_mod.Oracle(np.float32)
        ref_img = o.forward(P, cam, sh_degree=deg)
        for k in ("radii", "tiles_touched", "flags"):
            np.testing.assert_array_equal(saved[k], o.get(k))
        for k in ("mean2d", "depth", "conic_opacity", "rgb"):
            assert np.array_equal(saved[k].view(np.uint32), o.get(k).reshape(saved[k].shape).view(np.uint32)), k
        np.testing.assert_array_equal(keys, o.get("keys"))
        np.testing.assert_array_equal(saved["vals"], o.get("vals"))
        np.testing.assert_array_equal(saved["ranges"], o.get("ranges"))
        assert r.num_rendered == o.get("vals").size > n and int((keys >> 32).max()) % 263 > 255
        ok = ~o.get("fragile").astype(bool)
        assert np.abs(img.cpu().numpy() - ref_img)[:, ok].max() < 1e-4
        dL = ((img.cpu().numpy() - tgt) / tgt[0].size).astype(np.float32)
        g = r.backward(torch.from_numpy(dL).to(r.tdev))
        torch.cuda.synchronize()
        ref = o.backward(dL)
        for k in KEYS:
            a, b = g[k].cpu().numpy().astype(np.float64).ravel(), ref[k].astype(np.float64).ravel()
            assert np.linalg.norm(a - b) <= 1e-3 * max(np.linalg.norm(b), 1e-30), k
    finally:
        r.close()


def check_pipeline_parity(rast, oracle_mod, name, cfg, variants=None, n_cams=1, cam_index=0):
    """Every stage of one view against the oracle (also used at full size by tests/test_gpu_large.py)."""
    n, W, H, deg, seed, soff, aa, bg = cfg
    variants = variants or BWD_VARIANTS
    spec, P, cam, tgt = scene(n, W, H, deg, seed, n_cams=n_cams, cam_index=cam_index, scale_offset=soff, bg=bg)
    img, saved, keys, runs, dL = _run_gpu(rast, P, cam, tgt, deg, aa, variants=variants)

    o = oracle_mod.Oracle(np.float32)
    ref_img = o.forward(P, cam, sh_degree=deg, antialias=aa)

    # --- A2 preprocess: integers bit-exact, floats bit-exact -------------------------------------
    np.testing.assert_array_equal(saved["radii"], o.get("radii"))
    np.testing.assert_array_equal(saved["tiles_touched"], o.get("tiles_touched"))
    np.testing.assert_array_equal(saved["flags"], o.get("flags"))
    for k in ("mean2d", "depth", "conic_opacity", "rgb"):
        a, b = saved[k], o.get(k).reshape(saved[k].shape)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), f"{k}: {(a != b).sum()} of {a.size} floats differ"
    if not name.startswith("rnd"):
        assert (saved["radii"] > 0).sum() > 0.5 * n

    # --- A3-A6 binning: bit-exact --------------------------------------------------------------------
    assert keys.shape == o.get("keys").shape
    np.testing.assert_array_equal(keys, o.get("keys"))
    np.testing.assert_array_equal(saved["vals"], o.get("vals"))
    np.testing.assert_array_equal(saved["ranges"], o.get("ranges"))
    assert rast.num_rendered == o.get("vals").size

    # --- A7 composite forward ----------------------------------------------------------------------------
    frag = o.get("fragile").astype(bool)
    assert frag.mean() < (5e-3 if name.startswith("rnd") else 1e-3)
    ok = ~frag
    nc_ref = o.get("n_contrib")
    assert np.array_equal(saved["n_contrib"][ok], nc_ref[ok]), f"{(saved['n_contrib'][ok] != nc_ref[ok]).sum()} n_contrib mismatches"
    m, worst = rel_close(img[:, ok], ref_img[:, ok], 1e-4, 1e-6)
    assert m.all(), f"rgb worst {worst}"
    m, worst = rel_close(saved["final_T"][ok], o.get("final_T")[ok], 1e-4, 1e-6)
    assert m.all(), f"final_T worst {worst}"

    # --- decision replay (round 4) -------------------------------------------------------------------------------------------
    # A threshold decision (alpha < 1/255, T (1 - alpha) < 1e-4, power > 0) that falls within rounding distance of its threshold can go
    # differently in fp32 and fp64. Rounds 1-3 carved the affected ("tainted") splats out of the strict comparison; now the HIP forward
    # RECORDS its decisions (dvs_debug_record_decisions: per list entry and 8x8 quadrant, the pixels that took the entry) and the fp64
    # oracle replays exactly those, so every pixel and every splat is held to the strict bar. What is left outside: splats that reach a
    # pixel whose alpha sits within 1e-5 of the 0.99 cap (a decision of the DVS_GRAD_TRUE backward that the forward does not take).
    Tn = int(rast.num_rendered)
    rec = rast.record_decisions(max(Tn, 1))
    from divshot_amd.raster import params_to_device as _p2d
    img_again = rast.forward(_p2d(P, rast.tdev), cam, sh_degree=deg, antialias=aa, absgrad=True)
    import torch as _torch
    _torch.cuda.synchronize()
    assert np.array_equal(img_again.cpu().numpy().view(np.uint32), img.view(np.uint32)), "recording the decisions changed the image"
    masks = rec[:Tn].cpu().numpy().view(np.uint64)
    rast.record_decisions(0)
    o64 = oracle_mod.Oracle(np.float64)
    o64.forward(P, cam, sh_degree=deg, antialias=aa)
    own_lists = np.array_equal(o64.get("vals"), o.get("vals")) and np.array_equal(o64.get("ranges"), o.get("ranges"))
    # fp64 can bin a splat differently when a radius sits on an integer boundary (C3 and the C5 shape at full size). Rounds 4-5 fell back
    # to the fp32 replay alone there, so the HIP-vs-fp64 figures were missing for exactly the two configurations that matter most. Since
    # round 6 the fp64 instantiation then composites over the fp32 / HIP LISTS (Oracle.set_lists) and replays the recorded decisions on them:
    # what differs between the two is again the arithmetic alone. (A splat fp64 culls but fp32 lists cannot be composited: fp32 replay then.)
    orp = oracle_mod.Oracle(np.float64)
    same_lists = True
    if not own_lists:
        orp.set_lists(o.get("vals"), o.get("ranges"))
    orp.set_replay(masks)
    try:
        img_r = orp.forward(P, cam, sh_degree=deg, antialias=aa)
    except AssertionError as e:
        if own_lists or "culled in this precision" not in str(e):
            raise
        same_lists = False
        orp = oracle_mod.Oracle(np.float32)
        orp.set_replay(masks)
        img_r = orp.forward(P, cam, sh_degree=deg, antialias=aa)
    assert np.array_equal(orp.get("n_contrib"), saved["n_contrib"]), "replayed decisions do not reproduce n_contrib"
    # A7 on EVERY pixel, fragile or not: against the fp32 oracle replaying the same decisions — the specification of the composite
    # arithmetic on bit-identical projected splats (an fp64 projection moves a sharp splat's alpha by up to 1e-4 through the rounding of
    # its fp32 mean alone, so the fp64 replay is the reference of the gradients below, not of a transmittance that is a product of
    # hundreds of (1 - alpha))
    orp32 = None
    if same_lists:
        orp32 = oracle_mod.Oracle(np.float32)
        orp32.set_replay(masks)
        img_r32, fT_r32 = orp32.forward(P, cam, sh_degree=deg, antialias=aa), None
        fT_r32 = orp32.get("final_T")
        assert np.array_equal(orp32.get("n_contrib"), saved["n_contrib"])
    else:
        img_r32, fT_r32 = img_r, orp.get("final_T")
    m, worst = rel_close(img, img_r32, 1e-4, 1e-6)
    assert m.all(), f"rgb (replay, all pixels) worst {worst}"
    m, worst = rel_close(saved["final_T"], fT_r32, 1e-4, 1e-6)
    assert m.all(), f"final_T (replay, all pixels) worst {worst}"
    m, worst = rel_close(img, img_r, 1e-3, 1e-5)                       # and the fp64 replay at the looser bar that input rounding allows
    # (full size: millions of pixels behind lists of hundreds of sharp splats — at most 1e-5 of the values may miss that bar, none by more
    # than 10x; the strict comparison of every pixel is the fp32 replay above)
    n_px_out, px_allowed = int((~m).sum()), (int(1e-5 * m.size) if n >= 500000 else 0)
    assert n_px_out <= px_allowed and worst <= (10.0 if px_allowed else 1.0), f"rgb (fp64 replay, all pixels): {n_px_out} of {m.size} values beyond the bar (allowed {px_allowed}), worst {worst}"
    frag_any = frag | o64.get("fragile").astype(bool)
    capf = orp.get("cap_fragile").astype(bool)
    tainted = np.zeros(n, bool)
    fy, fx = np.where(capf)
    m2, rad, co = saved["mean2d"].astype(np.float64), saved["radii"], saved["conic_opacity"].astype(np.float64)
    tiles_x = (W + 15) // 16
    vals_l, ranges_l = saved["vals"], saved["ranges"].astype(np.int64)
    for x, y in zip(fx, fy):
        t_ = (y // 16) * tiles_x + x // 16
        cand = np.unique(vals_l[ranges_l[t_, 0]:ranges_l[t_, 1]])
        ddx, ddy = m2[cand, 0] - x, m2[cand, 1] - y
        power = -0.5 * (co[cand, 0] * ddx * ddx + co[cand, 2] * ddy * ddy) - co[cand, 1] * ddx * ddy
        alpha = np.minimum(0.99, co[cand, 3] * np.exp(np.minimum(power, 0.0)))
        hit = (np.abs(ddx) <= rad[cand]) & (np.abs(ddy) <= rad[cand]) & (rad[cand] > 0) & (power <= 1e-6) & (alpha >= (1.0 / 255.0) * (1 - 1e-3))
        tainted[cand[hit]] = True
    # what the replay cannot pin is bounded for EVERY configuration (no configuration looser than 1e-4 on > 10 %)
    assert tainted.mean() < 0.10, tainted.mean()
    clean = ~tainted
    clean_early = clean
    report = {"config": name, "n": n, "visible": int((saved["radii"] > 0).sum()), "T": int(rast.num_rendered),
              "fragile_pixel_fraction": float(frag_any.mean()), "tainted_splat_fraction": float(tainted.mean()),
              "decision_replay": ("fp32 strict + fp64 wide" + ("" if own_lists else " (fp64 composites over the fp32 lists: it bins a splat differently)")) if same_lists
                                 else "fp32 (a listed splat is culled in fp64)", "cap_fragile_pixel_fraction": float(capf.mean()),
              "rgb_vs_fp64_replay": {"values_beyond_1e-3": n_px_out, "of": int(m.size), "worst_err_over_tol": worst}, "runs": {}}

    # Untainted splats: strict, against fp64. Tainted splats (a fragile pixel in the footprint, where the HIP path may
    # legitimately take the other branch of a threshold than either oracle): the whole tainted set within 1e-3 relative L2
    # of the fp32 oracle (one flipped 1/255-alpha contribution moves a gradient by ~1e-3 of its magnitude; measured over all
    # configurations, modes and kernels: at most 2.1e-4, profiles/r02d_parity_report.jsonl).
    def check_group(tag, rec, got, want64, want32):
        got = np.asarray(got, np.float64); want64 = np.asarray(want64, np.float64); want32 = np.asarray(want32, np.float64)
        scale = np.abs(want64).max()
        out = {}
        if clean.any():
            g, w = got[clean], want64[clean]
            err = np.abs(g - w)
            tol = 1e-4 * np.abs(w) + 1e-5 * scale
            l2 = np.linalg.norm((g - w).ravel()) / max(np.linalg.norm(w.ravel()), 1e-300)
            out["clean_worst_err_over_tol"] = float((err / np.maximum(tol, 1e-300)).max()); out["clean_rel_l2"] = float(l2)
            rec[tag] = out
            assert (err <= tol).all(), f"{tag}: worst {(err / tol).max()}, frac ok {(err <= tol).mean()}"
            # scenes of a few dozen splats have no averaging over rows: their L2 bound is the plain fp32 one (still 2x inside 1e-4)
            assert l2 < (1e-5 if n >= 1000 else 5e-5), f"{tag}: relative L2 error {l2}"
        if tainted.any():
            g, w = got[tainted], want32[tainted]
            l2 = np.linalg.norm((g - w).ravel()) / max(np.linalg.norm(w.ravel()), 1e-300)
            worst = float((np.abs(g - w) / (1e-4 * np.abs(w) + 1e-5 * max(np.abs(want32).max(), 1e-300))).max())
            out["tainted_rel_l2_vs_fp32_oracle"] = float(l2); out["tainted_worst_err_over_tol"] = worst
            rec[tag] = out
            assert l2 < 1e-3, f"{tag} (tainted set): relative L2 error {l2}"

    culled = saved["radii"] == 0
    oracle_grads = {}
    o_l2, o_out = {}, {}                       # (mode, group) -> relative L2 / (outliers, worst) between the fp32 and the fp64 oracle replay
    for mode in (0, 1):
        # strict reference: the fp32 oracle replaying the HIP path's own decisions — the specification evaluated on bit-identical
        # projected splats with identical contributor sets (what is compared is the arithmetic alone). The fp64 replay is checked
        # beside it at the bar the fp32 rounding of the INPUTS allows: a splat behind hundreds of contributors inherits the relative
        # error those accumulate in T when each alpha moves by 1e-4 with the rounding of its mean (relative L2 < 1e-4 per group,
        # every element within 10x the strict tolerance). The plain fp32 oracle stays the reference of the few cap-tainted splats.
        strict = orp32 if same_lists else orp
        ref64 = {k: v.astype(np.float64) for k, v in strict.backward(dL.astype(strict.dtype), grad_mode=mode).items()}
        inter64 = {k: strict.get(k).astype(np.float64) for k in ("dL_dmean2d", "dL_dconic_opacity", "dL_drgb", "absgrad")}
        wide = None
        if same_lists:
            wide = {k: v.copy() for k, v in orp.backward(dL, grad_mode=mode).items()}
            for k in KEYS:
                a, b = ref64[k][clean_early], wide[k][clean_early]
                l2 = np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-300)
                # relative L2 < 1e-4 per group between the two ORACLE precisions; at full size the position group is input-rounding-bound
                # (C5 shape: 2.1e-4 — splats half as large at twice the resolution, the fp32 rounding of a mean moves alpha by more): there the
                # figure is recorded, bounded by 1e-3, and becomes the yardstick of the HIP-vs-fp64 assertion below
                o_l2[(mode, k)] = float(l2)
                assert l2 < (1e-4 if n < 500000 else 1e-3), f"fp32 replay vs fp64 replay, mode {mode}, {k}: relative L2 {l2}"
                ratio = np.abs(a - b) / (10 * (1e-4 * np.abs(b) + 1e-5 * np.abs(wide[k]).max()) + 1e-300)
                # Every element within 10x the strict tolerance — at full size (3 M elements per group) with a stated allowance: the two
                # ORACLE precisions themselves differ by up to 12.8x on 6 of 2 999 862 position elements of C3 (sharp splats behind hundreds of
                # contributors: the fp32 rounding of the inputs), and by up to 53x on 626 of 15 M at the C5 shape, so there at most 1e-4 of a group's elements may lie
                # between 10x and 100x, none beyond; the count and the worst ratio go into the report and bound what the HIP path may show below.
                n_out, allowed = int((ratio > 1).sum()), (int(1e-4 * ratio.size) if n >= 500000 else 0)
                o_out[(mode, k)] = (n_out, float(ratio.max()) * 10)
                report.setdefault("fp32_vs_fp64_replay", {})[f"mode{mode}/{k}"] = {"rel_l2": float(l2), "worst_err_over_strict_tol": float(ratio.max()) * 10,
                                                                                   "elements_beyond_10x": n_out, "of": int(ratio.size)}
                assert n_out <= allowed and (ratio <= 10.0).all(), (f"fp32 replay vs fp64 replay, mode {mode}, {k}: {n_out} of {ratio.size} elements beyond 10x the "
                                                                    f"strict tolerance (allowed {allowed}), worst {float(ratio.max()) * 10:.1f}x")
        ref32 = {k: v.copy() for k, v in o.backward(dL, grad_mode=mode).items()}
        inter32 = {k: o.get(k).copy() for k in ("dL_dmean2d", "dL_dconic_opacity", "dL_drgb", "absgrad")}
        oracle_grads[mode] = ref64
        for variant in variants:
            grads, inter = runs[(mode, variant)]
            rec = report["runs"].setdefault(f"grad_mode{mode}/{variant}", {})
            tag = f"[mode {mode}, {variant}] "
            for k in ("dL_dmean2d", "dL_dconic_opacity", "dL_drgb"):
                check_group(tag + k, rec, inter[k], inter64[k], inter32[k])
            check_group(tag + "absgrad", rec, grads["absgrad2d"], inter64["absgrad"], inter32["absgrad"])
            check_group(tag + "mean2d", rec, grads["mean2d"], inter64["dL_dmean2d"], inter32["dL_dmean2d"])
            for k in KEYS:
                check_group(tag + "grad " + k, rec, grads[k], ref64[k], ref32[k])
            if wide is not None:
                # the HIP gradients DIRECTLY against the fp64 oracle replaying the same decisions (not only through the fp32
                # replay): relative L2 < 1e-4 per group over the clean splats, every element within 10x the strict tolerance — the bar
                # the fp32 rounding of the INPUTS allows (an fp64 projection moves a sharp splat's alpha by up to 1e-4). Written to the report.
                for k in KEYS:
                    a, b = np.asarray(grads[k], np.float64)[clean_early], wide[k][clean_early]
                    l2 = np.linalg.norm((a - b).ravel()) / max(np.linalg.norm(b.ravel()), 1e-300)
                    worst = float((np.abs(a - b) / (1e-4 * np.abs(b) + 1e-5 * np.abs(wide[k]).max() + 1e-300)).max()) if a.size else 0.0      # (all-zero group: SH bands above the degree)
                    r_ = np.abs(a - b) / (1e-4 * np.abs(b) + 1e-5 * np.abs(wide[k]).max() + 1e-300) if a.size else np.zeros(1)
                    oo = o_out.get((mode, k), (0, 0.0))       # full size: no more outliers than the fp32 specification itself has against fp64 (+ 25 %)
                    n_out, allowed = int((r_ > 10.0).sum()), (int(1.25 * oo[0]) + 10 if n >= 500000 else 0)
                    rec[tag + "grad " + k + " vs fp64 replay"] = {"rel_l2": float(l2), "worst_err_over_strict_tol": worst, "elements_beyond_10x": n_out, "of": int(r_.size)}
                    # the HIP path must be as close to fp64 as the fp32 SPECIFICATION is: 1e-4, or — where input rounding alone puts the
                    # fp32 oracle further out (full-size position gradients) — within 25 % of the fp32 oracle's own distance
                    bar = max(1e-4, 1.25 * o_l2.get((mode, k), 0.0))
                    rec[tag + "grad " + k + " vs fp64 replay"]["fp32_oracle_vs_fp64_rel_l2"] = o_l2.get((mode, k))
                    assert l2 < bar, f"{tag}{k}: HIP vs fp64 replay relative L2 {l2} (bar {bar})"
                    assert n_out <= allowed and worst <= max(10.0, 1.25 * oo[1]), (f"{tag}{k}: HIP vs fp64 replay: {n_out} elements beyond 10x (allowed {allowed}), worst {worst} x "
                                                                                     f"the strict tolerance (fp32 oracle vs fp64: {oo[0]} elements, worst {oo[1]})")
            for k in KEYS:       # culled splats get exactly zero rows
                assert not np.any(grads[k][culled]), k
    # the size of the ambiguity between the two backward definitions on this scene (fp64 oracle): relative L2 per group
    report["lineage_vs_true_rel_l2"] = {k: float(np.linalg.norm((oracle_grads[1][k] - oracle_grads[0][k]).ravel()) /
                                                max(np.linalg.norm(oracle_grads[0][k].ravel()), 1e-300)) for k in KEYS}
    try:
        os.makedirs(os.path.dirname(REPORT), exist_ok=True)
        with open(REPORT, "a") as f:
            f.write(json.dumps(report) + "\n")
    except OSError:
        pass
    print("parity report:", json.dumps({k: report[k] for k in ("config", "tainted_splat_fraction", "fragile_pixel_fraction", "lineage_vs_true_rel_l2")}))
    return report


def test_accumulate_two_views(rast, oracle_mod):
    """opts.accumulate adds the second view's rows into the first's (multi-view batches)."""
    import torch
    from divshot_amd.raster import params_to_device
    spec = dv.make_spec(4000, 128, 96, sh_degree=3, n_cams=3)
    P = dv.synth_splats(spec)
    Pd = params_to_device(P, rast.tdev)
    total = None
    acc = None
    for ci in (0, 2):
        cam = dv.synth_camera(spec, ci)
        tgt = dv.synth_target(spec, ci)
        img = rast.forward(Pd, cam, sh_degree=3)
        dL = (img - torch.from_numpy(tgt).to(rast.tdev)) / tgt[0].size
        single = rast.backward(dL.contiguous())
        if total is None:
            total = {k: single[k].clone() for k in KEYS}
            acc = {k: single[k].clone() for k in KEYS}
        else:
            for k in KEYS:
                total[k] += single[k]
            rast.backward(dL.contiguous(), grads=acc, accumulate=True)
    torch.cuda.synchronize()
    for k in KEYS:
        m, worst = rel_close(acc[k].cpu().numpy(), total[k].cpu().numpy(), 1e-4, 2e-6)
        assert m.all(), (k, worst)


def test_split_backward_two_contexts(gpu_device):
    """dvs_raster_backward_composite + _project == dvs_raster_backward (A8 and A9 as separate calls); two views on
    two contexts and two streams, composites unordered, projects ordered by an event, accumulate into shared rows; and the
    state errors of the split API."""
    import torch
    from divshot_amd import DvsError
    from divshot_amd.raster import Rasterizer, params_to_device
    dev = torch.device(gpu_device)
    spec = dv.make_spec(6000, 160, 96, sh_degree=3, n_cams=3)
    P = dv.synth_splats(spec)
    Pd = params_to_device(P, dev)
    rasts = [Rasterizer(dev.index or 0, max_splats=6000, max_w=160, max_h=96) for _ in range(2)]
    cams = [dv.synth_camera(spec, ci) for ci in (0, 2)]
    tgts = [torch.from_numpy(dv.synth_target(spec, ci)).to(dev) for ci in (0, 2)]
    # reference: fused calls, sequential accumulation on one context
    ref = None
    for cam, tgt in zip(cams, tgts):
        img = rasts[0].forward(Pd, cam, sh_degree=3, absgrad=True)
        dL = ((img - tgt) / tgt[0].numel()).contiguous()
        ref = rasts[0].backward(dL) if ref is None else rasts[0].backward(dL, grads=ref, accumulate=True)
    ref = {k: v.clone() for k, v in ref.items()}
    torch.cuda.synchronize()
    # split: each view on its own context and stream
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]
    done = [torch.cuda.Event() for _ in range(2)]
    grads = {k: torch.empty_like(Pd[k]) for k in KEYS}
    grads["absgrad2d"] = torch.empty((6000, 2), dtype=torch.float32, device=dev)
    keep = []
    for v in range(2):
        with torch.cuda.stream(streams[v]):
            img = rasts[v].forward(Pd, cams[v], sh_degree=3, absgrad=True)
            dL = ((img - tgts[v]) / tgts[v][0].numel()).contiguous()
            keep.append(dL)
            rasts[v].backward_composite(dL)
            if v > 0:
                streams[v].wait_event(done[v - 1])
            rasts[v].backward_project(grads=grads, accumulate=(v > 0))
            done[v].record(streams[v])
    torch.cuda.synchronize()
    for k in list(KEYS) + ["absgrad2d"]:
        m, worst = rel_close(grads[k].cpu().numpy(), ref[k].cpu().numpy(), 1e-4, 2e-6)      # fp32 atomics: not bit-reproducible
        assert m.all(), (k, worst)
    # state errors: project without composite, composite without forward
    with pytest.raises(DvsError):
        rasts[0].backward_project(grads=grads)
    fresh = Rasterizer(dev.index or 0, max_splats=16, max_w=32, max_h=32)
    with pytest.raises((DvsError, AssertionError)):
        fresh.backward_composite(keep[0])


def test_factorised_sh_gradient(rast):
    """dvs_sh_grad_combine over V views == sum over views of the sh0/shN rows dvs_raster_backward writes:
    the multi-GPU exchange ships dcolor (12 B/splat/view) instead of the 192-B SH rows."""
    import torch
    from divshot_amd.raster import params_to_device
    from divshot_amd.parallel import GradBuffer, FactorisedExchange
    spec = dv.make_spec(6000, 160, 120, sh_degree=3, n_cams=4, seed=9)
    P = dv.synth_splats(spec)
    Pd = params_to_device(P, rast.tdev)
    n, V = 6000, 3
    ref = None
    gb = GradBuffer(n, rast.tdev)
    fx = FactorisedExchange(n, rast.tdev, 1)
    dcol_all = torch.zeros((V, n, 3), dtype=torch.float32, device=rast.tdev)
    geom_sum = {k: torch.zeros_like(Pd[k]) for k in ("pos", "scale", "rot", "opacity")}
    campos = []
    for v in range(V):
        cam = dv.synth_camera(spec, v + 1)
        campos.append(list(cam.campos))
        tgt = torch.from_numpy(dv.synth_target(spec, v + 1)).to(rast.tdev)
        img = rast.forward(Pd, cam, sh_degree=3)
        dL = ((img - tgt) / tgt[0].numel()).contiguous()
        full = rast.backward(dL)
        if ref is None:
            ref = {k: full[k].clone() for k in ("sh0", "shN")}
        else:
            for k in ref:
                ref[k] += full[k]
        fact = rast.backward(dL, grads=dict(gb.views), factorised_sh=True)
        for k in geom_sum:
            # geometry rows do not depend on the mode (two backward runs: fp32 atomics reorder sums, so not bit-equal)
            assert torch.allclose(fact[k], full[k], rtol=1e-4, atol=1e-5 * float(full[k].abs().max())), k
            geom_sum[k] += full[k]
        dcol_all[v].copy_(fact["dcolor"])
    # a single-view cross-check of dcolor itself: sh0 row = SH_C0 * dcolor
    assert torch.allclose(full["sh0"], 0.28209479177387814 * dcol_all[V - 1], rtol=1e-4, atol=1e-5 * float(full["sh0"].abs().max()))
    rast.sh_grad_combine(Pd["pos"], np.array(campos, np.float32), dcol_all, gb.views["sh0"], gb.views["shN"], 3)
    torch.cuda.synchronize()
    for k in ("sh0", "shN"):
        m, worst = rel_close(gb.views[k].cpu().numpy(), ref[k].cpu().numpy(), 1e-4, 1e-5)
        assert m.all(), (k, worst)
    # accumulate mode adds a second copy
    rast.sh_grad_combine(Pd["pos"], np.array(campos, np.float32), dcol_all, gb.views["sh0"], gb.views["shN"], 3, accumulate=True)
    torch.cuda.synchronize()
    m, worst = rel_close(gb.views["shN"].cpu().numpy(), 2 * ref["shN"].cpu().numpy(), 1e-4, 1e-5)
    assert m.all(), worst
    assert fx.dcolor_all.shape == (1, n, 3) and fx.dcolor_local.shape == (1, n, 3)


def test_tiled_shn_layout(rast):
    """DVS_SHN_TILED ([ceil(n/64)][45][64]) is a pure relayout: same image bit for bit, same gradients, for n not a multiple
    of 64 and for every SH degree; dvs_shn_relayout matches the numpy definition both ways; accumulate and the factorised
    combine work in the tiled layout too."""
    import torch
    from divshot_amd.raster import params_to_device, shn_rows_to_tiled_np, shn_tiled_to_rows_np, tiled_floats
    n = 5003
    spec = dv.make_spec(n, 200, 120, sh_degree=3, n_cams=3, seed=21)
    P = dv.synth_splats(spec)
    Pd = params_to_device(P, rast.tdev)
    t_dev = rast.shn_relayout(Pd["shN"], n, to_tiled=True)
    torch.cuda.synchronize()
    assert t_dev.numel() == tiled_floats(n)
    np.testing.assert_array_equal(t_dev.cpu().numpy(), shn_rows_to_tiled_np(P["shN"]))
    back = rast.shn_relayout(t_dev, n, to_tiled=False)
    np.testing.assert_array_equal(back.cpu().numpy(), P["shN"])
    Pt = dict(Pd); Pt["shN"] = t_dev
    cam = dv.synth_camera(spec, 1)
    tgt = torch.from_numpy(dv.synth_target(spec, 1)).to(rast.tdev)
    for deg in (0, 1, 2, 3):
        img_r = rast.forward(Pd, cam, sh_degree=deg).clone()
        dL = ((img_r - tgt) / tgt[0].numel()).contiguous()
        g_r = {k: v.clone() for k, v in rast.backward(dL).items()}
        img_t = rast.forward(Pt, cam, sh_degree=deg, shn_tiled=True)
        assert torch.equal(img_r, img_t), deg
        g_t = rast.backward(dL)
        torch.cuda.synchronize()
        assert g_t["shN"].numel() == tiled_floats(n)
        got = shn_tiled_to_rows_np(g_t["shN"].cpu().numpy(), n)
        m, worst = rel_close(got, g_r["shN"].cpu().numpy(), 1e-4, 1e-5)
        assert m.all(), (deg, worst)
        for k in ("pos", "sh0", "opacity", "scale", "rot"):
            m, worst = rel_close(g_t[k].cpu().numpy(), g_r[k].cpu().numpy(), 1e-4, 1e-5)
            assert m.all(), (deg, k, worst)
    # accumulate in the tiled layout: second backward adds the same rows again
    acc = {k: v.clone() for k, v in g_t.items() if k in KEYS}
    rast.backward(dL, grads=acc, accumulate=True)
    torch.cuda.synchronize()
    m, worst = rel_close(shn_tiled_to_rows_np(acc["shN"].cpu().numpy(), n), 2 * shn_tiled_to_rows_np(g_t["shN"].cpu().numpy(), n), 1e-4, 1e-5)
    assert m.all(), worst
    # factorised combine into tiled rows
    fact = rast.backward(dL, factorised_sh=True)
    sh0 = torch.zeros((n, 3), device=rast.tdev); shn = torch.zeros(tiled_floats(n), device=rast.tdev)
    rast.sh_grad_combine(Pd["pos"], np.array([list(cam.campos)], np.float32), fact["dcolor"][None].contiguous(), sh0, shn, 3, shn_tiled=True)
    torch.cuda.synchronize()
    m, worst = rel_close(shn.cpu().numpy(), g_t["shN"].cpu().numpy(), 1e-4, 1e-5)
    assert m.all(), worst


def test_edge_cases(rast, oracle_mod):
    """empty scene, everything culled, one splat on a tile corner, a splat covering the whole image
    (wave-cooperative duplication), a pixel stack that saturates (T < 1e-4), tile lists > 256 entries."""
    import torch
    from divshot_amd.raster import params_to_device
    spec = dv.make_spec(0, 80, 48, sh_degree=1)
    cam = dv.synth_camera(spec, 0)
    cam.bg[0], cam.bg[1], cam.bg[2] = 0.25, 0.5, 0.75

    def run(P, deg=1):
        Pd = params_to_device(P, rast.tdev)
        img = rast.forward(Pd, cam, sh_degree=deg, absgrad=True)
        torch.cuda.synchronize()
        o = oracle_mod.Oracle(np.float32)
        ref = o.forward(P, cam, sh_degree=deg)
        saved = rast.saved()
        np.testing.assert_array_equal(saved["radii"], o.get("radii"))
        np.testing.assert_array_equal(saved["vals"], o.get("vals"))
        np.testing.assert_array_equal(rast.sorted_keys(), o.get("keys"))
        np.testing.assert_array_equal(saved["ranges"], o.get("ranges"))
        ok = ~o.get("fragile").astype(bool)
        m, worst = rel_close(img.cpu().numpy()[:, ok], ref[:, ok], 1e-4, 1e-6)
        assert m.all(), worst
        assert np.array_equal(saved["n_contrib"][ok], o.get("n_contrib")[ok])
        dLn = np.random.default_rng(0).standard_normal(ref.shape).astype(np.float32)
        g = rast.backward(torch.from_numpy(dLn).to(rast.tdev))
        torch.cuda.synchronize()
        gref = o.backward(dLn)
        for k in KEYS:
            m, worst = rel_close(g[k].cpu().numpy(), gref[k], 2e-4, 1e-5)
            assert m.all(), (k, worst)
        return img.cpu().numpy(), saved, o

    def mk(n):
        return {"pos": np.zeros((n, 3), np.float32), "sh0": np.zeros((n, 3), np.float32), "shN": np.zeros((n, 15, 3), np.float32),
                "opacity": np.zeros((n,), np.float32), "scale": np.full((n, 3), -3.0, np.float32),
                "rot": np.tile(np.array([1, 0, 0, 0], np.float32), (n, 1))}

    # empty scene -> background
    img, saved, _ = run(mk(0))
    assert np.allclose(img[0], 0.25) and np.allclose(img[2], 0.75) and rast.num_rendered == 0
    # all behind the camera / too near / zero quaternion / transparent
    P = mk(6); P["pos"][:, 2] = [-5, 0.1, 0.2, 5, 5, 5]; P["rot"][3] = 0; P["opacity"][4] = -20.0; P["pos"][5, 0] = 1e4
    img, saved, _ = run(P)
    assert (saved["radii"] == 0).all() and rast.num_rendered == 0
    # splat centred exactly on a tile corner (pixel centre 15.5 -> between tiles), and one covering everything
    P = mk(3); P["pos"][:, 2] = 5.0
    fx = cam.focal_x
    P["pos"][0, 0] = (15.5 - (80 - 1) / 2) * 5.0 / fx; P["pos"][0, 1] = (15.5 - (48 - 1) / 2) * 5.0 / fx
    P["scale"][1] = 1.5; P["opacity"][1] = 1.0; P["sh0"][1] = [1.0, -4.0, 0.5]   # huge, green clamps to 0
    P["pos"][2] = [0.3, -0.2, 3.0]; P["scale"][2] = [-2.0, -1.0, -4.0]; P["rot"][2] = [0.3, -0.8, 0.1, 0.5]
    img, saved, o = run(P)
    assert saved["tiles_touched"][1] == 5 * 3 and saved["tiles_touched"][0] == 4
    assert (o.get("flags")[1] & 2) != 0
    # saturation + long tile lists: 1500 opaque splats stacked on a few pixels
    rng = np.random.default_rng(3)
    P = mk(1500); P["pos"][:, 2] = np.linspace(3, 9, 1500); P["pos"][:, :2] = rng.normal(0, 0.05, (1500, 2))
    P["opacity"][:] = 3.0; P["scale"][:] = -2.5; P["sh0"][:] = rng.normal(0, 1, (1500, 3))
    img, saved, o = run(P)
    r = saved["ranges"]
    assert (r[:, 1] - r[:, 0]).max() > 256
    assert (saved["final_T"] < 2e-4).any()
    # depth keys that span 17 binades (depths from 0.25 to 30 000): every byte of the 32-bit depth key carries order information;
    # the order is still the oracle's stable sort, bit for bit
    n = 3000
    P = mk(n); z = np.exp(rng.uniform(np.log(0.25), np.log(3.0e4), n)).astype(np.float32); P["pos"][:, 2] = z
    P["pos"][:, 0] = (rng.uniform(-0.3, 0.3, n) * z).astype(np.float32); P["pos"][:, 1] = (rng.uniform(-0.15, 0.15, n) * z).astype(np.float32)
    P["scale"][:] = np.log(1.5 * z / fx)[:, None].astype(np.float32); P["opacity"][:] = 0.0; P["sh0"][:] = rng.normal(0, 1, (n, 3))
    img, saved, o = run(P)
    dk = saved["depth"][saved["radii"] > 0].view(np.uint32)
    assert int(dk.max()) - int(np.float32(0.2).view(np.uint32)) >= 1 << 27 and (saved["radii"] > 0).sum() > 2000
    # a tile list with more than 65 536 entries (SURVEY.md §8(c) edge fixture): 70k faint 1-px splats over one tile; alpha only
    # just clears 1/255 near each centre, so no pixel saturates and every wave walks the whole list (274 LDS batches)
    n = 70_000
    P = mk(n); P["pos"][:, 2] = rng.uniform(4.0, 6.0, n).astype(np.float32)
    px = rng.uniform(32.0, 48.0, n); py = rng.uniform(16.0, 32.0, n)               # tile (2, 1) of the 80x48 image
    P["pos"][:, 0] = ((px - (80 - 1) / 2) * P["pos"][:, 2] / fx).astype(np.float32)
    P["pos"][:, 1] = ((py - (48 - 1) / 2) * P["pos"][:, 2] / fx).astype(np.float32)
    P["scale"][:] = np.log(0.8 * P["pos"][:, 2] / fx)[:, None].astype(np.float32)   # sigma ~ 0.8 px (+0.3 low-pass)
    P["opacity"][:] = np.float32(np.log(0.0047 / (1 - 0.0047))); P["sh0"][:] = rng.normal(0, 1, (n, 3))
    img, saved, o = run(P)
    r = saved["ranges"]
    assert (r[:, 1] - r[:, 0]).max() > 65536
    assert saved["final_T"].min() > 1e-3 and saved["n_contrib"].max() > 65536


def test_sort_pairs(rast):
    """A5 radix sort alone: stable, LSD, arbitrary sizes and bit ranges (vs numpy stable argsort)."""
    import torch
    rng = np.random.default_rng(11)
    for n, lo, hi in [(0, 0, 32), (1, 0, 32), (63, 0, 32), (2048, 0, 32), (2049, 0, 16), (100_003, 0, 32),
                      (1_000_000, 0, 32), (300_000, 8, 24), (70_000, 0, 13)]:
        keys = rng.integers(0, 2**32, size=n, dtype=np.uint64).astype(np.uint32)
        if n > 10:
            keys[rng.integers(0, n, n // 3)] = keys[0]        # many duplicates -> stability matters
        vals = np.arange(n, dtype=np.uint32)
        k_d = torch.from_numpy(keys.view(np.int32).copy()).to(rast.tdev)
        v_d = torch.from_numpy(vals.view(np.int32).copy()).to(rast.tdev)
        rast.sort_pairs(k_d, v_d, lo, hi)
        torch.cuda.synchronize()
        mask = np.uint32(((1 << (hi - lo)) - 1) << lo) if hi - lo < 32 else np.uint32(0xFFFFFFFF)
        order = np.argsort(keys & mask, kind="stable")
        np.testing.assert_array_equal(k_d.cpu().numpy().view(np.uint32), keys[order])
        np.testing.assert_array_equal(v_d.cpu().numpy().view(np.uint32), vals[order])
    # all keys equal, and already-sorted / reverse-sorted inputs
    for keys in (np.full(5000, 7, np.uint32), np.arange(5000, dtype=np.uint32), np.arange(5000, dtype=np.uint32)[::-1].copy()):
        vals = np.arange(keys.size, dtype=np.uint32)
        k_d = torch.from_numpy(keys.view(np.int32).copy()).to(rast.tdev)
        v_d = torch.from_numpy(vals.view(np.int32).copy()).to(rast.tdev)
        rast.sort_pairs(k_d, v_d)
        torch.cuda.synchronize()
        order = np.argsort(keys, kind="stable")
        np.testing.assert_array_equal(v_d.cpu().numpy().view(np.uint32), vals[order])


def test_depth_sort_widest_digits_through_the_test_hook(gpu_device):
    """the scatter's wide path with 11-bit digits (key ranges of 31 bits) cannot be reached through the projection — its squares
    overflow beyond z ~ 1e19 — so dvs_debug_sort_depth_keys drives the forward's three-pass range-adaptive depth sort directly: synthetic
    keys spanning 8 ... 31 bits (digit widths 3 ... 11, i.e. the LDS-reordering path up to 9 bits and the straight-from-registers path
    above), with culled keys (0xFFFFFFFF), many exact ties and both partition sizes, in both rank modes, against numpy's stable argsort."""
    import torch
    from divshot_amd.raster import Rasterizer
    rng = np.random.default_rng(5)
    old = os.environ.get("DVS_FE_RANK")
    try:
        for mode in ("default", "ballot"):
            if mode == "ballot": os.environ["DVS_FE_RANK"] = "ballot"
            else: os.environ.pop("DVS_FE_RANK", None)
            r = Rasterizer(0, max_splats=1 << 21, max_w=64, max_h=64)
            for n in (70_001, 1_700_000):
                for span_bits in (8, 20, 27, 28, 30, 31):
                    lo = np.uint32(rng.integers(1, 1 << 20)) if span_bits < 31 else np.uint32(1)
                    keys = (lo + rng.integers(0, (1 << span_bits) - int(lo if span_bits == 31 else 0), n, dtype=np.uint64)).astype(np.uint32)
                    keys[rng.integers(0, n, n // 4)] = keys[1]                    # exact ties: stability matters
                    keys[0], keys[-1] = lo, np.uint32(int(lo) + (1 << span_bits) - 1 - int(lo if span_bits == 31 else 0))      # the extremes are present
                    keys[rng.integers(0, n, n // 7)] = 0xFFFFFFFF               # culled
                    k_d = torch.from_numpy(keys.view(np.int32).copy()).to(r.tdev)
                    out = torch.empty(n, dtype=torch.int32, device=r.tdev)
                    cnt, bits = C.c_uint32(0), C.c_uint32(0)
                    rc = dv.lib.dvs_debug_sort_depth_keys(r.ctx, None, k_d.data_ptr(), n, out.data_ptr(), C.byref(cnt), C.byref(bits))
                    assert rc == 0, dv.lib.dvs_last_error()
                    vis = np.where(keys != 0xFFFFFFFF)[0]
                    want = vis[np.argsort(keys[vis], kind="stable")]
                    assert cnt.value == vis.size
                    rb = int(keys[vis].max() - keys[vis].min()).bit_length()
                    assert bits.value == max(1, (rb + 2) // 3), (span_bits, bits.value)
                    np.testing.assert_array_equal(out.cpu().numpy()[:cnt.value].view(np.uint32), want.astype(np.uint32), err_msg=f"{mode} n={n} span={span_bits}")
                    if span_bits == 31: assert bits.value == 11
            r.close()
    finally:
        if old is None: os.environ.pop("DVS_FE_RANK", None)
        else: os.environ["DVS_FE_RANK"] = old


def test_sort_rank_modes_agree(gpu_device):
    """Round 6: the scatters of both radix sorts rank inside a wave by returning LDS adds when the device serves the lanes of one LDS
    address in lane order (probed on the device at dvs_create: dvs_get_sort_rank_mode == 1), else by the ballot multisplit of rounds
    2-5 (DVS_FE_RANK=ballot forces it). Both must give THE stable order: pair sorts on adversarial digit distributions (all keys equal,
    two keys alternating, long runs, few distinct keys, random with many duplicates; 8- and 16-keys-per-thread partitions) against
    numpy's stable argsort in both modes, and a 3-view forward whose lists, ranges, n_contrib and images are bit-identical between them."""
    import torch
    from divshot_amd.raster import Rasterizer, params_to_device
    from divshot_amd import lib
    rng = np.random.default_rng(23)
    cases = []
    for n in (4096 * 3 + 17, 1_600_000):                      # ITEMS = 8 and ITEMS = 16 partitions
        cases += [np.full(n, 0x1234567, np.uint32), (np.arange(n) & 1).astype(np.uint32) * 0x10001,
                  (np.arange(n) // 1000).astype(np.uint32), rng.integers(0, 5, n).astype(np.uint32) * 0x01010101,
                  rng.integers(0, 2**13, n).astype(np.uint32), rng.integers(0, 2**32, n, dtype=np.uint64).astype(np.uint32)]
    n, W, H, V = 6007, 208, 120, 3
    spec = dv.make_spec(n, W, H, sh_degree=2, n_cams=4, seed=17)
    P = dv.synth_splats(spec)
    cams = [dv.synth_camera(spec, i + 1) for i in range(V)]
    got = {}
    old = os.environ.get("DVS_FE_RANK")
    try:
        for mode in ("ballot", "default"):
            if mode == "ballot": os.environ["DVS_FE_RANK"] = "ballot"
            else: os.environ.pop("DVS_FE_RANK", None)
            r = Rasterizer(0, max_splats=max(n, 1 << 16), max_w=W, max_h=H, max_views=V)
            rm = lib.dvs_get_sort_rank_mode(r.ctx)
            assert rm == (0 if mode == "ballot" else 1), (mode, rm)      # an MI355X passes the probe: the default IS the atomic ranking
            for keys in cases:
                vals = np.arange(keys.size, dtype=np.uint32)
                k_d = torch.from_numpy(keys.view(np.int32).copy()).to(r.tdev)
                v_d = torch.from_numpy(vals.view(np.int32).copy()).to(r.tdev)
                r.sort_pairs(k_d, v_d, 0, 32)
                torch.cuda.synchronize()
                order = np.argsort(keys, kind="stable")
                np.testing.assert_array_equal(v_d.cpu().numpy().view(np.uint32), vals[order], err_msg=mode)
                np.testing.assert_array_equal(k_d.cpu().numpy().view(np.uint32), keys[order], err_msg=mode)
            Pd = params_to_device(P, r.tdev)
            imgs = r.forward_views(Pd, cams, sh_degree=2)
            torch.cuda.synchronize()
            got[mode] = (imgs.clone(), [r.view_saved(v) for v in range(V)])
            r.close()
    finally:
        if old is None: os.environ.pop("DVS_FE_RANK", None)
        else: os.environ["DVS_FE_RANK"] = old
    assert torch.equal(got["ballot"][0], got["default"][0])
    for v in range(V):
        for k in ("vals", "sorted_tile", "ranges", "n_contrib", "radii"):
            np.testing.assert_array_equal(got["ballot"][1][v][k], got["default"][1][v][k], err_msg=f"view {v} {k}")


def test_error_paths_and_nan_inputs(gpu_device):
    """Status codes instead of crashes or silent fallbacks; NaN / inf parameters cull the splat and poison nothing else."""
    import ctypes as C
    import torch
    from divshot_amd._lib import lib, Splats, Opts, SplatGrads
    from divshot_amd.raster import Rasterizer, params_to_device
    r = Rasterizer(0, max_splats=1000, max_w=128, max_h=96)
    spec = dv.make_spec(2000, 128, 96, sh_degree=3)
    P = dv.synth_splats(spec)
    cam = dv.synth_camera(spec, 0)
    Pd = params_to_device(P, r.tdev)
    with pytest.raises(dv.DvsError, match="capacity"):
        r.forward(Pd, cam)                                   # n = 2000 > max_splats = 1000
    small = {k: v[:500].contiguous() for k, v in Pd.items()}
    with pytest.raises(dv.DvsError, match="sh_degree"):
        r.forward(small, cam, sh_degree=4)
    big = dv.synth_camera(dv.make_spec(10, 4096, 96), 0)
    with pytest.raises(dv.DvsError, match="capacity"):
        r.forward(small, big)                                # image wider than max_w
    # backward without a forward on this context
    sp = Splats(*[small[k].data_ptr() for k in ("pos", "sh0", "shN", "opacity", "scale", "rot")], 500, 0)
    g = {k: torch.empty_like(v) for k, v in small.items()}
    sg = SplatGrads(*[g[k].data_ptr() for k in ("pos", "sh0", "shN", "opacity", "scale", "rot")], None, None, None)
    dL = torch.zeros((3, 96, 128), device=r.tdev)
    opts = Opts(3, 0, 0, 0, 0)
    assert lib.dvs_raster_backward(r.ctx, None, C.byref(sp), C.byref(cam), C.byref(opts), dL.data_ptr(), C.byref(sg)) == 4   # DVS_ERR_STATE
    assert b"forward" in lib.dvs_last_error()
    # NaN / inf parameters
    bad = {k: v.clone() for k, v in small.items()}
    bad["pos"][0, 0] = float("nan"); bad["scale"][1, 1] = float("inf"); bad["rot"][2] = float("nan"); bad["opacity"][3] = float("nan")
    bad["scale"][4] = -float("inf")
    img = r.forward(bad, cam, sh_degree=3)
    torch.cuda.synchronize()
    s = r.saved()
    assert (s["radii"][[0, 2, 3]] == 0).all() and s["radii"][4] >= 0 and np.isfinite(s["conic_opacity"]).all()
    assert torch.isfinite(img).all()
    # the alignment contract of dvs_raster.h: a slice that starts inside an allocation is rejected, not run at reduced rate
    with pytest.raises(dv.DvsError, match="16-byte aligned"):
        r.forward({k: v[5:] for k, v in small.items()}, cam, sh_degree=3)
    ref = r.forward({k: v[5:].clone() for k, v in small.items()}, cam, sh_degree=3).clone()
    img2 = r.forward({k: v[5:].clone() for k, v in bad.items()}, cam, sh_degree=3)
    assert torch.equal(ref, img2)
    grads = r.backward(torch.ones_like(img2))
    torch.cuda.synchronize()
    assert all(torch.isfinite(v).all() for v in grads.values())
    r.close()


def test_async_forward_no_host_sync(gpu_device):
    """dvs_set_async: T stays on the device (SURVEY.md §8(a) A3 "avoid the readback"): same image, same saved state and same
    gradients as the synchronous forward; an instance-arena overflow is a hard error reported by the next call, after which the
    enlarged arena renders the view correctly."""
    import torch
    from divshot_amd.raster import Rasterizer, params_to_device
    n, W, H = 30_000, 400, 300
    spec = dv.make_spec(n, W, H, sh_degree=2, seed=3)
    P = dv.synth_splats(spec); cam = dv.synth_camera(spec, 0); tgt = torch.from_numpy(dv.synth_target(spec, 0)).cuda()
    r = Rasterizer(0, max_splats=n, max_w=W, max_h=H)
    Pd = params_to_device(P, r.tdev)
    img_s = r.forward(Pd, cam, sh_degree=2, absgrad=True).clone()
    T_sync = r.num_rendered
    saved_s = r.saved()
    g_s = {k: v.clone() for k, v in r.backward(((img_s - tgt) / (W * H)).contiguous()).items()}
    r.set_async(True)
    img_a = r.forward(Pd, cam, sh_degree=2, absgrad=True)
    assert r.num_rendered == 2 ** 64 - 1                       # DVS_T_UNKNOWN until asked for
    g_a = r.backward(((img_a - tgt) / (W * H)).contiguous())
    assert r.get_num_rendered() == T_sync
    saved_a = r.saved()
    assert torch.equal(img_s, img_a)
    for k in ("radii", "vals", "sorted_tile", "ranges", "n_contrib"):
        np.testing.assert_array_equal(saved_s[k], saved_a[k], err_msg=k)
    for k in KEYS:
        m, worst = rel_close(g_a[k].cpu().numpy(), g_s[k].cpu().numpy(), 1e-4, 2e-6)       # fp32 atomics: not bit-reproducible
        assert m.all(), (k, worst)
    # the asynchronous forward does not materialise the sorted tile ids (dvs_fwd_state.sorted_tile NULL) unless asked to
    from divshot_amd import _lib as _l
    st_ = _l.FwdState()
    assert dv.lib.dvs_get_view_state(r.ctx, 0, C.byref(st_)) == 0 and not st_.sorted_tile
    assert dv.lib.dvs_set_export_sorted_tiles(r.ctx, 1) == 0
    img_e = r.forward(Pd, cam, sh_degree=2, absgrad=True)
    assert r.get_num_rendered() == T_sync and torch.equal(img_e, img_s)
    assert dv.lib.dvs_get_view_state(r.ctx, 0, C.byref(st_)) == 0 and st_.sorted_tile
    exported = r._d2h(st_.sorted_tile, (T_sync,), np.uint32)
    np.testing.assert_array_equal(exported, saved_s["sorted_tile"])
    assert dv.lib.dvs_set_export_sorted_tiles(r.ctx, 0) == 0
    r.close()
    # overflow: splats covering ~30 tiles each against an arena sized for 5 per splat
    spec = dv.make_spec(8000, 640, 480, sh_degree=0, seed=4, scale_log_offset=2.2)
    P = dv.synth_splats(spec); cam = dv.synth_camera(spec, 0)
    big = Rasterizer(0, max_splats=8000, max_w=640, max_h=480)
    Pd = params_to_device(P, big.tdev)
    ref = big.forward(Pd, cam, sh_degree=0).clone()
    T_true = big.num_rendered
    assert T_true > 6 * 8000, T_true
    big.close()
    small = Rasterizer(0, max_splats=8000, max_w=640, max_h=480)
    small.set_async(True)
    small.forward(Pd, cam, sh_degree=0)                         # overflows the fresh arena (nothing is written out of bounds)
    with pytest.raises(dv.DvsError, match="more tile instances"):
        small.get_num_rendered()
    img = small.forward(Pd, cam, sh_degree=0)                   # the arena was enlarged by the failed call
    assert small.get_num_rendered() == T_true
    assert torch.equal(img, ref)
    # the same overflow noticed by the next forward instead of by the getter
    small2 = Rasterizer(0, max_splats=8000, max_w=640, max_h=480)
    small2.set_async(True)
    small2.forward(Pd, cam, sh_degree=0)
    torch.cuda.synchronize()
    with pytest.raises(dv.DvsError, match="more tile instances"):
        small2.forward(Pd, cam, sh_degree=0)
    img = small2.forward(Pd, cam, sh_degree=0)
    torch.cuda.synchronize()
    assert torch.equal(img, ref)
    small.close(); small2.close()


def test_hip_graph_replay_of_the_pass(gpu_device):
    """The asynchronous multi-view pass has no host synchronisation and fixed launch shapes, so a whole training-iteration pass
    (forward of two views, loss gradient, backward) can be captured into a HIP graph and replayed: after the parameters change, a
    replay gives the same images and gradients as the eager pass on the new parameters, and the instance count is still reported."""
    import torch
    from divshot_amd.raster import Rasterizer, params_to_device
    n, W, H, V = 20_000, 320, 240, 2
    spec = dv.make_spec(n, W, H, sh_degree=3, n_cams=4, seed=9)
    P = dv.synth_splats(spec)
    cams = [dv.synth_camera(spec, i) for i in range(V)]
    tg = torch.stack([torch.from_numpy(dv.synth_target(spec, i)) for i in range(V)]).cuda()
    r = Rasterizer(0, max_splats=n, max_w=W, max_h=H, max_views=V)
    r.set_async(True)
    Pd = params_to_device(P, r.tdev)
    Pd["shN"] = r.shn_relayout(Pd["shN"], n, to_tiled=True)
    out = torch.empty((V, 3, H, W), device=r.tdev)
    grads = {k: torch.zeros_like(v) for k, v in Pd.items()}
    def one_pass():
        imgs = r.forward_views(Pd, cams, sh_degree=3, absgrad=True, out=out, shn_tiled=True)
        dL = (imgs - tg) * (1.0 / (W * H))
        r.backward_views(dL, grads=grads)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for _ in range(3):                                     # warm-up: the arena reaches its steady size, T of earlier passes is known
            one_pass()
    torch.cuda.synchronize()
    T0 = r.get_num_rendered()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        one_pass()
    Pd["pos"] += 0.01 * torch.randn_like(Pd["pos"])            # in place: the graph holds the pointers
    Pd["opacity"] += 0.1
    graph.replay()
    torch.cuda.synchronize()
    T1 = r.get_num_rendered()
    img_g = out.clone(); g_g = {k: v.clone() for k, v in grads.items()}
    one_pass()
    torch.cuda.synchronize()
    assert r.get_num_rendered() == T1 and T1 != T0
    assert torch.equal(out, img_g)
    for k in KEYS:
        m, worst = rel_close(g_g[k].cpu().numpy(), grads[k].cpu().numpy(), 1e-4, 2e-6)
        assert m.all(), (k, worst)
    r.close()


@pytest.mark.parametrize("tiled,bwd", [(True, "tr"), (False, "tr"), (True, "blocks")] + ([(True, "reduce")] if os.environ.get("DVS_TEST_ALL_VARIANTS") == "1" else []))
def test_multi_view_batch_equals_single_views(gpu_device, tiled, bwd):
    """dvs_raster_forward_views / _backward_views (BASELINE config C4: several cameras per iteration in ONE pass — parameters read once,
    one depth sort / scan / (view, tile) sort / composite launch, gradients written once) against the same views run one by one with
    opts.accumulate: images and every saved per-view array bit for bit (the per-view instance list is cut out of the batch-wide
    sorted list), gradients to fp32-atomics roundoff, per-view colour gradients (factorised exchange) likewise."""
    import torch
    from divshot_amd.raster import Rasterizer, params_to_device
    n, W, H, V = 5003, 208, 120, 3
    spec = dv.make_spec(n, W, H, sh_degree=3, n_cams=4, seed=31)
    P = dv.synth_splats(spec)
    cams = [dv.synth_camera(spec, i + 1) for i in range(V)]
    for i, c_ in enumerate(cams):
        c_.bg[0], c_.bg[1], c_.bg[2] = 0.1 * i, 0.2, 0.3 + 0.1 * i          # per-view backgrounds
    tg = [torch.from_numpy(dv.synth_target(spec, i + 1)).cuda() for i in range(V)]
    single = Rasterizer(0, max_splats=n, max_w=W, max_h=H)
    batch = Rasterizer(0, max_splats=n, max_w=W, max_h=H, max_views=V)
    # the batch's A8 kernel against the OTHER shipped kernel, one view at a time (a different summation order: "tr" vs "blocks")
    single.set_backward_variant("blocks" if bwd == "tr" else "tr"); batch.set_backward_variant(bwd)
    Pd = params_to_device(P, single.tdev)
    if tiled:
        Pd = dict(Pd); Pd["shN"] = single.shn_relayout(Pd["shN"], n, to_tiled=True)
    ref_imgs, ref_saved, ref_dcol, ref = [], [], [], None
    for v in range(V):
        img = single.forward(Pd, cams[v], sh_degree=3, absgrad=True, shn_tiled=tiled)
        ref_imgs.append(img.clone()); ref_saved.append(single.saved())
        dL = ((img - tg[v]) / (W * H)).contiguous()
        ref = single.backward(dL, grads=ref, accumulate=ref is not None, want_mean2d=True)
        ref_dcol.append(single.backward(dL, factorised_sh=True)["dcolor"].clone())
        if v == 0:
            ref = {k: t.clone() for k, t in ref.items()}
    imgs = batch.forward_views(Pd, cams, sh_degree=3, absgrad=True, shn_tiled=tiled)
    torch.cuda.synchronize()
    assert batch.num_rendered == sum(s_["vals"].size for s_ in ref_saved)
    for v in range(V):
        assert torch.equal(imgs[v], ref_imgs[v]), f"image of view {v}"
        sv = batch.view_saved(v)
        for k in ("radii", "flags", "tiles_touched", "vals", "sorted_tile", "ranges", "n_contrib"):
            np.testing.assert_array_equal(sv[k], ref_saved[v][k], err_msg=f"view {v} {k}")
        for k in ("mean2d", "depth", "conic_opacity", "rgb", "final_T"):
            assert np.array_equal(sv[k].view(np.uint32), ref_saved[v][k].view(np.uint32)), f"view {v} {k}"
    dL_all = torch.stack([(imgs[v] - tg[v]) / (W * H) for v in range(V)]).contiguous()
    g = batch.backward_views(dL_all, want_mean2d=True)
    torch.cuda.synchronize()
    for k in list(KEYS) + ["absgrad2d", "mean2d"]:
        m, worst = rel_close(g[k].cpu().numpy(), ref[k].cpu().numpy(), 1e-4, 2e-6)
        assert m.all(), (k, worst)
    # factorised: per-view colour gradients instead of SH rows; accumulate adds a second copy of the geometry rows
    g2 = {k: g[k].clone() for k in ("pos", "scale", "rot", "opacity")}
    g2["dcolor"] = torch.empty((V, n, 3), device=batch.tdev)
    batch.forward_views(Pd, cams, sh_degree=3, absgrad=True, shn_tiled=tiled)
    batch.backward_composite(dL_all)
    early = batch.backward_dcolor(torch.full((V, n, 3), 7.0, device=batch.tdev))     # dvs_raster_backward_dcolor: before A9, from the A8 rows
    batch.backward_project(grads=g2, accumulate=True, factorised_sh=True)
    torch.cuda.synchronize()
    assert torch.equal(early, g2["dcolor"]), "dvs_raster_backward_dcolor differs from the colour gradients A9 emits"
    for v in range(V):
        m, worst = rel_close(g2["dcolor"][v].cpu().numpy(), ref_dcol[v].cpu().numpy(), 1e-4, 2e-6)
        assert m.all(), (v, worst)
    for k in ("pos", "scale", "rot", "opacity"):
        m, worst = rel_close(g2[k].cpu().numpy(), 2 * ref[k].cpu().numpy(), 1e-4, 2e-6)
        assert m.all(), (k, worst)
    # asynchronous forward (device-side T) works for batches too
    batch.set_async(True)
    imgs_a = batch.forward_views(Pd, cams, sh_degree=3, absgrad=True, shn_tiled=tiled)
    assert batch.get_num_rendered() == sum(s_["vals"].size for s_ in ref_saved)
    assert torch.equal(imgs_a, imgs)
    with pytest.raises(dv.DvsError, match="max_views"):
        single.forward_views(Pd, cams, sh_degree=3, shn_tiled=tiled)
    single.close(); batch.close()


def test_fused_sh_rows_equal_the_two_kernel_form(gpu_device):
    """Round 6: on one GPU the multi-view A9 builds the SH rows in its epilogue (k_preprocess_bwd_views<.., FUSE_SH>: colour gradient and
    unit direction of every view parked in LDS) instead of writing per-view colour gradients for k_sh_grad_combine. Same expressions, same
    view order: sh0 / shN rows BIT-identical to the two-kernel form (DVS_A9_NO_FUSE_SH=1), every other group too; overwrite and
    accumulate; 1, 3 and 8 views; degree 0, 2, 3; a view that sees nothing."""
    import torch
    from divshot_amd.raster import Rasterizer, params_to_device
    n, W, H = 5003, 208, 120
    for V, deg in ((1, 3), (3, 2), (8, 3), (2, 0)):
        spec = dv.make_spec(n, W, H, sh_degree=deg, n_cams=V + 1, seed=41 + V)
        P = dv.synth_splats(spec)
        cams = [dv.synth_camera(spec, i + 1) for i in range(V)]
        if V == 3:                                                   # the middle camera looks the other way: it sees nothing
            for c_ in range(4):
                cams[1].view[c_ * 4 + 2] = -cams[1].view[c_ * 4 + 2]     # depth axis flipped: everything is behind this camera
        r = Rasterizer(0, max_splats=n, max_w=W, max_h=H, max_views=V)
        r.keep_intermediates(True)
        Pd = params_to_device(P, r.tdev)
        Pd["shN"] = r.shn_relayout(Pd["shN"], n, to_tiled=True)
        imgs = r.forward_views(Pd, cams, sh_degree=deg, absgrad=True, shn_tiled=True)
        dL = torch.rand(imgs.shape, device=r.tdev, generator=torch.Generator(device=r.tdev).manual_seed(3)) - 0.5
        r.backward_composite(dL.contiguous())
        out = {}
        old = os.environ.get("DVS_A9_NO_FUSE_SH")
        try:
            for tag in ("two", "fused"):
                if tag == "two": os.environ["DVS_A9_NO_FUSE_SH"] = "1"
                else: os.environ.pop("DVS_A9_NO_FUSE_SH", None)
                g = r.backward_project()
                g = {k: t.clone() for k, t in g.items()}
                g2 = r.backward_project(grads={k: t.clone() for k, t in g.items()}, accumulate=True)
                torch.cuda.synchronize()
                out[tag] = (g, {k: t.clone() for k, t in g2.items()})
        finally:
            if old is None: os.environ.pop("DVS_A9_NO_FUSE_SH", None)
            else: os.environ["DVS_A9_NO_FUSE_SH"] = old
        for k in KEYS:
            assert torch.equal(out["two"][0][k], out["fused"][0][k]), (V, deg, k)
            assert torch.equal(out["two"][1][k], out["fused"][1][k]), (V, deg, k, "accumulate")
        assert out["fused"][0]["sh0"].abs().max() > 0 and (deg == 0 or out["fused"][0]["shN"].abs().max() > 0)
        assert torch.allclose(out["fused"][1]["shN"], 2 * out["fused"][0]["shN"], rtol=1e-6, atol=0)
        r.close()


def test_forward_prepared_in_chunks_equals_forward(gpu_device):
    """dvs_raster_forward_views_prepare (round 6): A2 of the next forward run ahead of it, by splat range — what a data-parallel step does
    behind its optimizer chunks. A forward after a complete, matching preparation skips its own A2 and is BIT-identical (images, every
    saved array, gradients of a following backward are those of the plain forward); a preparation that is incomplete, out of order, for
    other cameras / options, or cancelled is ignored (and refused where the API can tell)."""
    import torch
    from divshot_amd.raster import Rasterizer, params_to_device
    from divshot_amd import _lib as _l
    n, W, H, V = 6007, 208, 120, 3
    spec = dv.make_spec(n, W, H, sh_degree=3, n_cams=4, seed=61)
    P = dv.synth_splats(spec)
    cams = [dv.synth_camera(spec, i + 1) for i in range(V)]
    r = Rasterizer(0, max_splats=n, max_w=W, max_h=H, max_views=V)
    Pd = params_to_device(P, r.tdev)
    Pd["shN"] = r.shn_relayout(Pd["shN"], n, to_tiled=True)
    ref = r.forward_views(Pd, cams, sh_degree=3, absgrad=True, shn_tiled=True).clone()
    ref_saved = [r.view_saved(v) for v in range(V)]
    dL = torch.rand(ref.shape, device=r.tdev, generator=torch.Generator(device=r.tdev).manual_seed(5)) - 0.5
    g_ref = {k: t.clone() for k, t in r.backward_views(dL.contiguous()).items()}
    sp = r._splats(Pd, tiled=True)
    carr = (_l.Camera * V)(*cams)
    opts = _l.Opts(); opts.sh_degree = 3; opts.absgrad = 1; opts.shn_layout = 1
    def prepare(first, count, cam_arr=carr, o=opts):
        return dv.lib.dvs_raster_forward_views_prepare(r.ctx, None, C.byref(sp), cam_arr, V, C.byref(o), first, count)
    torch.cuda.synchronize()
    # complete preparation in three ragged chunks -> the forward is the same forward
    for first, count in ((0, 2048), (2048, 2560), (4608, n - 4608)):
        assert prepare(first, count) == 0, dv.lib.dvs_last_error()
    poison = torch.full_like(Pd["pos"], float("nan"))          # if the forward did run its own A2 now, it would see these positions
    good = Pd["pos"].clone()
    torch.cuda.synchronize()
    Pd["pos"].copy_(poison); torch.cuda.synchronize()
    img = r.forward_views(Pd, cams, sh_degree=3, absgrad=True, shn_tiled=True)
    torch.cuda.synchronize()
    Pd["pos"].copy_(good); torch.cuda.synchronize()
    assert torch.equal(img, ref), "the prepared forward differs (or ran its own A2)"
    for v in range(V):
        sv = r.view_saved(v)
        for k in ("radii", "flags", "tiles_touched", "vals", "ranges", "n_contrib"):
            np.testing.assert_array_equal(sv[k], ref_saved[v][k], err_msg=f"view {v} {k}")
        for k in ("mean2d", "depth", "conic_opacity", "rgb", "final_T"):
            assert np.array_equal(sv[k].view(np.uint32), ref_saved[v][k].view(np.uint32)), (v, k)
    g = r.backward_views(dL.contiguous())
    for k in KEYS:
        m, worst = rel_close(g[k].cpu().numpy(), g_ref[k].cpu().numpy(), 1e-4, 2e-6)          # (fp32 atomics in A8: not bit-reproducible)
        assert m.all(), (k, worst)
    # the preparation is used once: the next forward projects by itself again (NaN positions now cull everything)
    Pd["pos"].copy_(poison); torch.cuda.synchronize()
    img_nan = r.forward_views(Pd, cams, sh_degree=3, absgrad=True, shn_tiled=True)
    torch.cuda.synchronize()
    Pd["pos"].copy_(good); torch.cuda.synchronize()
    assert r.num_rendered == 0 and not torch.equal(img_nan, ref)
    # incomplete / cancelled / mismatching preparations are ignored; out-of-order chunks are refused
    assert prepare(0, 2048) == 0
    assert torch.equal(r.forward_views(Pd, cams, sh_degree=3, absgrad=True, shn_tiled=True), ref)             # incomplete
    assert prepare(0, 2048) == 0 and prepare(4096, 512) != 0 and b"ascending" in dv.lib.dvs_last_error()        # a gap
    assert prepare(0, n) == 0 and dv.lib.dvs_raster_forward_cancel_prepared(r.ctx) == 0
    Pd["pos"].copy_(poison); torch.cuda.synchronize()
    assert r.forward_views(Pd, cams, sh_degree=3, absgrad=True, shn_tiled=True) is not None and r.num_rendered == 0        # cancelled: own A2 saw the NaNs
    Pd["pos"].copy_(good); torch.cuda.synchronize()
    other = (_l.Camera * V)(*[dv.synth_camera(spec, i) for i in range(V)])
    assert prepare(0, n, cam_arr=other) == 0                                                                      # prepared for OTHER cameras
    assert torch.equal(r.forward_views(Pd, cams, sh_degree=3, absgrad=True, shn_tiled=True), ref)
    assert prepare(100, 256) != 0 and prepare(0, 0) != 0                                                          # not a multiple of 256 / empty
    r.close()


def test_live_lists_give_the_same_gradients(gpu_device):
    """dvs_set_live_lists: the "tr" backward over the forward's compacted lists (entries that reach their tile) against the same
    kernel over the full lists — one view and a 3-view pass, a scene with many entries that miss their tiles (small, faint splats) and
    one with saturated stacks: equal to the roundoff of the fp32 atomics; the exported lists and n_contrib do not change."""
    import torch
    from divshot_amd.raster import Rasterizer, params_to_device
    for (n, W, H, soff, seed) in ((6001, 200, 136, -0.5, 5), (3000, 96, 64, 1.2, 6)):
        V = 3
        spec = dv.make_spec(n, W, H, sh_degree=3, n_cams=4, seed=seed, scale_log_offset=soff)
        P = dv.synth_splats(spec)
        cams = [dv.synth_camera(spec, i) for i in range(V)]
        tg = torch.stack([torch.from_numpy(dv.synth_target(spec, i)) for i in range(V)]).cuda()
        res = {}
        for on in (False, True):
            r = Rasterizer(0, max_splats=n, max_w=W, max_h=H, max_views=V)
            r.set_backward_variant("tr"); r.set_live_lists(on)
            Pd = params_to_device(P, r.tdev)
            img1 = r.forward(Pd, cams[1], sh_degree=3, absgrad=True).clone()
            s1 = r.saved()
            g1 = {k: v.clone() for k, v in r.backward(((img1 - tg[1]) / (W * H)).contiguous()).items()}
            Pt = dict(Pd); Pt["shN"] = r.shn_relayout(Pd["shN"], n, to_tiled=True)
            imgs = r.forward_views(Pt, cams, sh_degree=3, absgrad=True, shn_tiled=True)
            gv = {k: v.clone() for k, v in r.backward_views(((imgs - tg) / (W * H)).contiguous()).items()}
            torch.cuda.synchronize()
            res[on] = (img1, s1, g1, imgs.clone(), gv)
            r.close()
        a, b = res[False], res[True]
        assert torch.equal(a[0], b[0]) and torch.equal(a[3], b[3])
        for k in ("vals", "sorted_tile", "ranges", "n_contrib"):
            assert np.array_equal(a[1][k], b[1][k]), k
        for ga, gb in ((a[2], b[2]), (a[4], b[4])):
            for k in ("pos", "sh0", "shN", "opacity", "scale", "rot", "absgrad2d"):
                if k in ga:
                    x, y = ga[k].double(), gb[k].double()
                    assert float((x - y).norm()) <= 2e-6 * float(x.norm()) + 1e-30, k


def test_project_chunks_equal_project(gpu_device):
    """dvs_raster_backward_project_chunk (A9 in splat chunks, so that a data-parallel step can send each chunk's geometry gradients off
    while the next chunk computes): bit-identical to the unchunked call — single view and a 3-view pass, with and without accumulate —
    and the state errors of the chunk API."""
    import torch
    from divshot_amd import DvsError
    from divshot_amd.raster import Rasterizer, params_to_device
    n, W, H, V = 5003, 208, 120, 3
    spec = dv.make_spec(n, W, H, sh_degree=3, n_cams=4, seed=11)
    P = dv.synth_splats(spec)
    cams = [dv.synth_camera(spec, i) for i in range(V)]
    tg = [torch.from_numpy(dv.synth_target(spec, i)).cuda() for i in range(V)]
    r1 = Rasterizer(0, max_splats=n, max_w=W, max_h=H)
    rv = Rasterizer(0, max_splats=n, max_w=W, max_h=H, max_views=V)
    Pd = params_to_device(P, r1.tdev)
    Pd = dict(Pd); Pd["shN"] = r1.shn_relayout(Pd["shN"], n, to_tiled=True)
    chunks = [(0, 1536), (1536, 2048), (3584, n - 3584)]
    keys = ("pos", "scale", "rot", "opacity", "dcolor", "absgrad2d", "mean2d")

    def both(r, fwd, dL, nv):
        out = []
        fwd()
        r.keep_intermediates(True)        # the composite backward's rows stay, so that both forms of A9 read the SAME rows (two
        r.backward_composite(dL)          # composite runs differ by the roundoff of their fp32 atomics)
        for chunked in (False, True):
            for acc in (False, True):
                g = {k: torch.full_like(v, 0.25) for k, v in Pd.items()}
                g["dcolor"] = torch.zeros((nv, n, 3), device=r.tdev) if nv > 1 else torch.zeros((n, 3), device=r.tdev)
                g["absgrad2d"] = torch.full((n, 2), 0.5, device=r.tdev); g["mean2d"] = torch.full((n, 2), 0.5, device=r.tdev)
                if chunked:
                    seen = []
                    r.backward_project_chunks(g, chunks, lambda k, f, c: seen.append((k, f, c)), accumulate=acc, want_mean2d=True)
                    assert seen == [(k, f, c) for k, (f, c) in enumerate(chunks)]
                else:
                    r.backward_project(grads=g, accumulate=acc, want_mean2d=True, factorised_sh=True)
                torch.cuda.synchronize()
                out.append({k: g[k].clone() for k in keys})
        for k in keys:
            assert torch.equal(out[0][k], out[2][k]), k          # overwrite: chunked == whole
            assert torch.equal(out[1][k], out[3][k]), k          # accumulate: chunked == whole
        assert not torch.equal(out[0]["pos"], out[1]["pos"])
        r.keep_intermediates(False)

    img = r1.forward(Pd, cams[0], sh_degree=3, absgrad=True, shn_tiled=True)
    both(r1, lambda: r1.forward(Pd, cams[0], sh_degree=3, absgrad=True, shn_tiled=True), ((img - tg[0]) / (W * H)).contiguous(), 1)
    imgs = rv.forward_views(Pd, cams, sh_degree=3, absgrad=True, shn_tiled=True)
    dLv = torch.stack([(imgs[v] - tg[v]) / (W * H) for v in range(V)]).contiguous()
    both(rv, lambda: rv.forward_views(Pd, cams, sh_degree=3, absgrad=True, shn_tiled=True), dLv, V)
    # out of order / misaligned / without a pending composite
    g = {k: torch.zeros_like(v) for k, v in Pd.items()}; g["dcolor"] = torch.zeros((n, 3), device=r1.tdev)
    r1.forward(Pd, cams[0], sh_degree=3, absgrad=True, shn_tiled=True)
    with pytest.raises(DvsError):
        r1.backward_project_chunks(g, [(0, n)], None)       # no composite backward pending
    r1.backward_composite(((img - tg[0]) / (W * H)).contiguous())
    with pytest.raises(DvsError):
        r1.backward_project_chunks(g, [(256, 512)], None)
    with pytest.raises(DvsError):
        r1.backward_project_chunks(g, [(0, 100), (100, n - 100)], None)       # the second chunk does not start on a multiple of 256
    r1.close(); rv.close()


@pytest.mark.parametrize("zmax", [40.0, 3.0e4, 1.0e9, 1.0e17])
def test_depth_sort_digit_width_follows_the_key_range(gpu_device, oracle_mod, zmax):
    """The depth sort takes three passes whatever the scene: its digit is ceil(bits(max key - min key) / 3) bits wide, chosen per view
    on the device (frontend.hip). Depths spread log-uniformly from the near plane to zmax put the range at 23 + log2(zmax / 0.25) bits:
    9-bit digits (re-ordered in LDS) up to a depth ratio of 2^12 from the 0.2 near plane, then 10-bit digits scattered straight from registers (11-bit digits
    — all 31 key bits — cannot be reached through the projection: beyond z ~ 1e19 its squares overflow and the splat is culled; the
    static-digit path of the same kernel is covered by test_sort_pairs). Bins (radii, (tile | depth) keys, values, ranges) bit-exact against the oracle's std::stable_sort, with many
    exactly equal depths (ties resolve by splat id) and a fifth of the splats culled (they leave in the first pass)."""
    import torch
    from divshot_amd.raster import Rasterizer, params_to_device
    n, W, H = 20000, 200, 152
    spec = dv.make_spec(n, W, H, sh_degree=1, seed=31)
    P = dv.synth_splats(spec)
    cam = dv.synth_camera(spec, 0)
    rng = np.random.default_rng(17)
    z = np.exp(rng.uniform(np.log(0.25), np.log(zmax), n)).astype(np.float32)
    z[: n // 10] = z[n // 10: 2 * (n // 10)]                              # exact depth ties
    z[rng.random(n) < 0.2] = 0.1                                          # inside the near plane: culled
    xy = rng.uniform(-0.45, 0.45, (n, 2)).astype(np.float32)
    P["pos"][:, 0] = xy[:, 0] * z; P["pos"][:, 1] = xy[:, 1] * z * (H / W); P["pos"][:, 2] = z
    P["scale"][:] = (np.log(np.maximum(z, 0.2) * 0.004)[:, None] + rng.normal(0, 0.3, (n, 3))).astype(np.float32)
    r = Rasterizer(0, max_splats=n, max_w=W, max_h=H)
    img = r.forward(params_to_device(P, r.tdev), cam, sh_degree=1)
    torch.cuda.synchronize()
    got, keys = r.saved(), r.sorted_keys()
    o = oracle_mod.Oracle(np.float32)
    ref = o.forward(P, cam, sh_degree=1)
    vis = got["radii"] > 0
    assert 0.5 * n < vis.sum() < 0.9 * n
    rbits = int(got["depth"][vis].view(np.uint32).max() - got["depth"][vis].view(np.uint32).min()).bit_length()
    assert rbits >= (28 if zmax > 1e3 else 20) and (rbits <= 27) == (zmax < 1e3), rbits      # 9-bit digits up to 27 bits (z up to ~800), 10-bit beyond
    assert np.array_equal(got["radii"], o.get("radii")) and np.array_equal(got["tiles_touched"], o.get("tiles_touched"))
    assert np.array_equal(keys, o.get("keys")) and np.array_equal(got["vals"], o.get("vals")) and np.array_equal(got["ranges"], o.get("ranges"))
    ok = ~o.get("fragile").astype(bool)
    assert np.abs(img.cpu().numpy() - ref)[:, ok].max() < 1e-4
    r.close()


@pytest.mark.parametrize("asy", [False, True])
def test_multi_view_batch_with_a_view_that_sees_nothing(gpu_device, asy):
    """A batch in which one camera looks the other way: its segment of both sorts is EMPTY (zero visible splats, zero instances — every
    per-view descriptor of the segmented front end has count 0). The other views must come out exactly as alone, the blind view is its
    background, and the gradients are those of the two seeing views. Synchronous and asynchronous forwards."""
    import copy, torch
    from divshot_amd.raster import Rasterizer, params_to_device
    n, W, H = 3001, 176, 112
    spec = dv.make_spec(n, W, H, sh_degree=2, n_cams=3, seed=77)
    P = dv.synth_splats(spec)
    cams = [dv.synth_camera(spec, 0), dv.synth_camera(spec, 1), dv.synth_camera(spec, 2)]
    blind = cams[1]
    for c in range(4):
        blind.view[c * 4 + 2] = -blind.view[c * 4 + 2]          # depth axis flipped: everything is behind this camera
    blind.bg[0], blind.bg[1], blind.bg[2] = 0.25, 0.5, 0.75
    tg = [torch.from_numpy(dv.synth_target(spec, i)).cuda() for i in range(3)]
    single = Rasterizer(0, max_splats=n, max_w=W, max_h=H)
    batch = Rasterizer(0, max_splats=n, max_w=W, max_h=H, max_views=3)
    batch.set_async(asy)
    Pd = params_to_device(P, single.tdev)
    ref_imgs, ref_saved, ref = [], [], None
    for v in range(3):
        img = single.forward(Pd, cams[v], sh_degree=2, absgrad=True)
        ref_imgs.append(img.clone()); ref_saved.append(single.saved())
        dL = ((img - tg[v]) / (W * H)).contiguous()
        ref = single.backward(dL, grads=ref, accumulate=ref is not None)
        if v == 0:
            ref = {k: t.clone() for k, t in ref.items()}
    assert ref_saved[1]["vals"].size == 0 and (ref_saved[1]["radii"] == 0).all()
    for rep in range(2):                                          # twice: the second forward starts from the first one's arenas and counters
        imgs = batch.forward_views(Pd, cams, sh_degree=2, absgrad=True)
        torch.cuda.synchronize()
        assert batch.get_num_rendered() == ref_saved[0]["vals"].size + ref_saved[2]["vals"].size
        for v in range(3):
            assert torch.equal(imgs[v], ref_imgs[v]), f"image of view {v}"
            sv = batch.view_saved(v)
            for k in ("radii", "flags", "tiles_touched", "vals", "sorted_tile", "ranges", "n_contrib"):
                np.testing.assert_array_equal(sv[k], ref_saved[v][k], err_msg=f"view {v} {k}")
        bgc = torch.tensor([0.25, 0.5, 0.75], device=imgs.device).view(3, 1, 1)
        assert torch.equal(imgs[1], bgc.expand(3, H, W))
        dL_all = torch.stack([(imgs[v] - tg[v]) / (W * H) for v in range(3)]).contiguous()
        g = batch.backward_views(dL_all)
        torch.cuda.synchronize()
        for k in KEYS:
            m, worst = rel_close(g[k].cpu().numpy(), ref[k].cpu().numpy(), 1e-4, 2e-6)
            assert m.all(), (rep, k, worst)
    single.close(); batch.close()


def test_tile_ranges_fused_into_the_sort_equal_the_separate_kernel(gpu_device):
    """A6 rides on the tile sort's last pass (every run of equal tile ids raises (~start, end) with atomic max, k_render_fwd decodes); with
    DVS_FE_NO_FUSE_A6=1 the separate boundary kernel of rounds 1-4 runs instead. Same ranges, same lists, same image — one view and a
    batch, synchronous and asynchronous (the asynchronous fused forward does not even write the sorted tile ids). Third run:
    DVS_FE_NO_KEY16=1 — A4 writes 32-bit tile ids instead of 16-bit ones (the form views of more than 65 536 tiles take)."""
    import subprocess, sys
    code = r"""
import os, sys, numpy as np, torch
sys.path.insert(0, os.environ["DVS_ROOT"])
import divshot_amd as dv
from divshot_amd.raster import Rasterizer, params_to_device
n, W, H, V = 30000, 330, 250, 3
spec = dv.make_spec(n, W, H, sh_degree=2, n_cams=4, seed=5)
P = dv.synth_splats(spec); cams = [dv.synth_camera(spec, i + 1) for i in range(V)]
r = Rasterizer(0, max_splats=n, max_w=W, max_h=H, max_views=V)
Pd = params_to_device(P, r.tdev)
out = {}
for asy in (False, True):
    r.set_async(asy)
    img1 = r.forward(Pd, cams[0], sh_degree=2).clone(); s1 = r.saved()
    imgv = r.forward_views(Pd, cams, sh_degree=2).clone(); sv = [r.view_saved(v) for v in range(V)]
    out[asy] = (img1.cpu().numpy(), s1, imgv.cpu().numpy(), sv)
np.save(sys.argv[1], np.array([out], dtype=object), allow_pickle=True)
"""
    import tempfile
    res = {}
    for tag, env_extra in (("fused", {}), ("separate", {"DVS_FE_NO_FUSE_A6": "1"}), ("keys32", {"DVS_FE_NO_KEY16": "1"})):
        with tempfile.TemporaryDirectory() as d:
            path = os.path.join(d, "o.npy")
            env = dict(os.environ, DVS_ROOT=os.path.dirname(os.path.dirname(os.path.abspath(__file__))), **env_extra)
            for k in ("DVS_FE_NO_FUSE_A6", "DVS_FE_NO_KEY16"):
                if k not in env_extra: env.pop(k, None)
            p = subprocess.run([sys.executable, "-c", code, path], capture_output=True, text=True, timeout=600, env=env)
            assert p.returncode == 0, p.stdout[-1500:] + p.stderr[-3000:]
            res[tag] = np.load(path, allow_pickle=True)[0]
    for asy in (False, True):
      for other in ("separate", "keys32"):
        f, s_ = res["fused"][asy], res[other][asy]
        assert np.array_equal(f[0], s_[0]) and np.array_equal(f[2], s_[2])
        for a, b in [(f[1], s_[1])] + list(zip(f[3], s_[3])):
            for k in ("vals", "sorted_tile", "ranges", "n_contrib"):
                assert np.array_equal(a[k], b[k]), (other, asy, k)
        # synchronous and asynchronous forwards agree as well
        for k in ("vals", "sorted_tile", "ranges", "n_contrib"):
            assert np.array_equal(res["fused"][False][1][k], res["fused"][True][1][k]), k


@pytest.mark.parametrize("name", ["C1_10k_256_deg0", "G2_2k_64_deg3", "ragged_5k_250x130_deg2_aa", "dense_3k_96_big", "C2_100k_800_deg3",
                                  "rnd3_65", "rnd7_3500", "rnd9_6000"])
def test_tight_tiles_opt_in(rast, oracle_mod, name):
    """dvs_opts.tile_bounds = DVS_TILES_TIGHT (opt-in): the instance list the binning stage emits equals the
    oracle twin's BIT FOR BIT (tiles_touched, (tile | depth) keys, values, ranges), is a sub-list of the canonical one, and nothing
    anybody can observe downstream changes: the image and final_T are bit-identical to the canonical run's, the gradients agree to
    fp32 summation order, and n_contrib names the same splat per pixel."""
    import torch
    from divshot_amd.raster import params_to_device
    full = [k for k in CONFIGS if k.startswith(name)]
    assert full, name
    n, W, H, deg, seed, soff, aa, bg = CONFIGS[full[0]]
    spec, P, cam, tgt = scene(n, W, H, deg, seed, scale_offset=soff, bg=bg)
    Pd = params_to_device(P, rast.tdev)
    rast.set_backward_variant("tr")
    img_c = rast.forward(Pd, cam, sh_degree=deg, antialias=aa, absgrad=True).clone()
    sc, keys_c = rast.saved(), rast.sorted_keys()
    dL = ((img_c - torch.from_numpy(tgt).to(rast.tdev)) / tgt[0].size).contiguous()
    g_c = {k: v.clone() for k, v in rast.backward(dL).items()}
    img_t = rast.forward(Pd, cam, sh_degree=deg, antialias=aa, absgrad=True, tight_tiles=True).clone()
    stt, keys_t = rast.saved(), rast.sorted_keys()
    g_t = {k: v.clone() for k, v in rast.backward(dL).items()}
    torch.cuda.synchronize()
    o = oracle_mod.Oracle(np.float32)
    o.forward(P, cam, sh_degree=deg, antialias=aa, tight_tiles=True)
    np.testing.assert_array_equal(stt["tiles_touched"], o.get("tiles_touched"))
    np.testing.assert_array_equal(keys_t, o.get("keys"))
    np.testing.assert_array_equal(stt["vals"], o.get("vals"))
    np.testing.assert_array_equal(stt["ranges"], o.get("ranges"))
    np.testing.assert_array_equal(stt["radii"], sc["radii"])                       # visibility (radius > 0) is not redefined
    assert set(keys_t.tolist()) <= set(keys_c.tolist()) and keys_t.size <= keys_c.size
    if n >= 2000 and soff < 1.0:
        assert keys_t.size < 0.9 * keys_c.size, (keys_t.size, keys_c.size)         # the point of the option
    assert torch.equal(img_t, img_c), "tight tiles changed the image"
    assert np.array_equal(stt["final_T"].view(np.uint32), sc["final_T"].view(np.uint32))
    # n_contrib is a position in the (shorter) list: it must name the same last contributor
    def last_splat(s):
        nc, out = s["n_contrib"], np.full(s["n_contrib"].shape, -1, np.int64)
        tiles_x = (W + 15) // 16
        ys, xs = np.nonzero(nc)
        t = (ys // 16) * tiles_x + xs // 16
        out[ys, xs] = s["vals"][s["ranges"][t, 0].astype(np.int64) + nc[ys, xs].astype(np.int64) - 1]
        return out
    assert np.array_equal(last_splat(stt), last_splat(sc))
    for k in g_c:
        a, b = g_t[k].double(), g_c[k].double()
        assert float((a - b).norm() / max(float(b.norm()), 1e-300)) < 2e-6, k
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()) + 1e-12, k


def test_tight_tiles_multi_view_batch(gpu_device):
    """The opt-in through the multi-view pass: every view's image bit-identical to the canonical batch, fewer instances, same gradients."""
    import torch
    from divshot_amd.raster import Rasterizer, params_to_device
    n, W, H, deg = 30_000, 320, 240, 2
    spec = dv.make_spec(n, W, H, sh_degree=deg, n_cams=3, seed=9)
    Pd = params_to_device(dv.synth_splats(spec), gpu_device)
    cams = [dv.synth_camera(spec, i) for i in range(3)]
    r = Rasterizer(0, max_splats=n, max_w=W, max_h=H, max_views=3)
    img_c = r.forward_views(Pd, cams, sh_degree=deg, absgrad=True).clone()
    Tc = r.get_num_rendered() if r.num_rendered == 2 ** 64 - 1 else r.num_rendered
    dL = torch.randn_like(img_c) / (W * H)
    g_c = {k: v.clone() for k, v in r.backward_views(dL).items()}
    img_t = r.forward_views(Pd, cams, sh_degree=deg, absgrad=True, tight_tiles=True).clone()
    Tt = r.get_num_rendered() if r.num_rendered == 2 ** 64 - 1 else r.num_rendered
    g_t = r.backward_views(dL)
    torch.cuda.synchronize()
    assert torch.equal(img_t, img_c) and Tt < 0.9 * Tc, (Tt, Tc)
    for k in g_c:
        assert float((g_t[k].double() - g_c[k].double()).norm() / max(float(g_c[k].double().norm()), 1e-300)) < 2e-6, k
    r.close()
