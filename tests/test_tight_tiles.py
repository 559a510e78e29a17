"""DVS_TILES_TIGHT (include/dvs_raster.h, opt-in) on the CPU oracle: the tightened instance list is a sub-list of the canonical one and
drops only instances that never contribute — image, final_T and every gradient are bit-identical to the canonical oracle run — on the
seeded scenes, the edge scenes of tests/golden/edge_scenes.py and with anti-aliasing. (The HIP path is compared with this twin bit for
bit in tests/test_gpu_parity.py::test_tight_tiles_opt_in.)"""
import numpy as np
import pytest
import divshot_amd as dv
from oracle.oracle import Oracle
from util import scene


@pytest.mark.parametrize("cfg", [(3000, 160, 120, 1, 3, 0.0, False), (500, 64, 64, 0, 5, 0.0, False), (6000, 250, 130, 2, 7, 0.5, True),
                                 (3000, 96, 96, 1, 5, 1.5, False), (64, 40, 23, 3, 2, -0.5, True)])
def test_tight_lists_drop_only_what_never_contributes(cfg):
    n, W, H, deg, seed, soff, aa = cfg
    spec, P, cam, tgt = scene(n, W, H, deg, seed, scale_offset=soff)
    a, b = Oracle(np.float32), Oracle(np.float32)
    ia = a.forward(P, cam, sh_degree=deg, antialias=aa, absgrad=True).copy()
    ib = b.forward(P, cam, sh_degree=deg, antialias=aa, absgrad=True, tight_tiles=True).copy()
    ka, kb = a.get("keys"), b.get("keys")
    assert kb.size <= ka.size and set(kb.tolist()) <= set(ka.tolist())
    assert np.array_equal(a.get("radii"), b.get("radii"))
    assert (b.get("tiles_touched") <= a.get("tiles_touched")).all()
    assert np.array_equal(ia.view(np.uint32), ib.view(np.uint32)) and np.array_equal(a.get("final_T"), b.get("final_T"))
    dL = np.random.default_rng(1).normal(size=ia.shape).astype(np.float32)
    ga, gb = a.backward(dL), b.backward(dL)
    for k in ga:
        assert np.array_equal(ga[k], gb[k]), k
    assert np.array_equal(a.get("absgrad"), b.get("absgrad"))
    if n >= 3000 and soff < 1.0:
        assert kb.size < 0.85 * ka.size, (kb.size, ka.size)


def test_rectangles_of_more_than_64_tiles_stay_whole():
    """A splat covering the whole image (the edge scene of tests/golden/edge_scenes.py) keeps its full rectangle: the tile mask has 64 bits."""
    import os, sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
    import edge_scenes as es
    cam, deg = es.camera(), es.DEG
    for name, P in es.scenes(cam).items():
        if name.startswith("E4"):
            continue                       # (70 000 splats in one tile: covered on the GPU; slow here)
        a, b = Oracle(np.float32), Oracle(np.float32)
        ia = a.forward(P, cam, sh_degree=deg).copy()
        ib = b.forward(P, cam, sh_degree=deg, tight_tiles=True).copy()
        assert np.array_equal(ia.view(np.uint32), ib.view(np.uint32)), name
        big = a.get("tiles_touched") > 64
        assert np.array_equal(a.get("tiles_touched")[big], b.get("tiles_touched")[big]), name


def test_thin_diagonal_splats_keep_every_contributing_tile():
    """the adversarial case of the tight tile bounds — needle-thin splats lying diagonally across several tiles. The per-pixel
    form a dx^2 + 2 b dx dy + c dy^2 then cancels terms thousands of times its value, so its rounding (which decides whether a pixel at
    the alpha = 1/255 rim takes the splat) exceeds a fixed margin; the margin scaled with the size of the terms must keep every tile in
    which the canonical run lets the splat contribute: image, final_T, n_contrib and gradients stay bit-identical."""
    rng = np.random.default_rng(5)
    n, W, H = 1500, 192, 160
    spec = dv.make_spec(n, W, H, sh_degree=0, seed=9)
    P = dv.synth_splats(spec)
    cam = dv.synth_camera(spec, 0)
    # needles: one long axis (about 30-60 px on screen), two axes a hundredth of a pixel wide, rotated about the view axis by ~45 degrees
    z = P["pos"][:, 2]
    px = z / cam.focal_x
    P["scale"][:, 0] = np.log(px * rng.uniform(10.0, 20.0, n)).astype(np.float32)
    P["scale"][:, 1] = np.log(px * 0.02).astype(np.float32)
    P["scale"][:, 2] = np.log(px * 0.02).astype(np.float32)
    ang = np.deg2rad(45.0 + rng.uniform(-8, 8, n)) * rng.choice([-1.0, 1.0], n)
    P["rot"][:] = np.stack([np.cos(ang / 2), np.zeros(n), np.zeros(n), np.sin(ang / 2)], 1).astype(np.float32)      # (w, x, y, z): about z
    P["opacity"][:] = rng.uniform(1.0, 4.0, n).astype(np.float32)
    a, b = Oracle(np.float32), Oracle(np.float32)
    ia = a.forward(P, cam, sh_degree=0, absgrad=True).copy()
    ib = b.forward(P, cam, sh_degree=0, absgrad=True, tight_tiles=True).copy()
    tt_a, tt_b = a.get("tiles_touched"), b.get("tiles_touched")
    multi = (tt_a > 3) & (tt_a <= 64)
    assert multi.sum() > 0.3 * n and tt_b[multi].sum() < 0.8 * tt_a[multi].sum()          # the needles span tiles, and most of their rectangles go
    assert np.array_equal(ia.view(np.uint32), ib.view(np.uint32))
    assert np.array_equal(a.get("final_T"), b.get("final_T")) and np.array_equal(a.get("n_contrib") > 0, b.get("n_contrib") > 0)
    dL = rng.normal(size=ia.shape).astype(np.float32)
    ga, gb = a.backward(dL), b.backward(dL)
    for k in ga:
        assert np.array_equal(ga[k], gb[k]), k
