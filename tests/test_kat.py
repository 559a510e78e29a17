"""Known-answer tests derived from the reference's in-tree code: the only numeric
conventions fenghuayumo/DIVSHOT pins for this path. They run against the CPU oracle (and, marked gpu, the HIP path)."""
import math
import numpy as np
import pytest
import divshot_amd as dv
from oracle import Oracle


def test_kat1_sh_basis_constants_and_signs():
    """gsplat_sh.hlsl:42-61 constants, :65-103 basis; gaussian_model.cpp:128 SH_C0."""
    o = Oracle()
    x, y, z = 0.3, -0.5, math.sqrt(1 - 0.09 - 0.25)
    b = o.sh_basis(3, (x, y, z))
    C1 = 0.4886025119029199
    want = [0.28209479177387814, -C1 * y, C1 * z, -C1 * x,
            1.0925484305920792 * x * y, -1.0925484305920792 * y * z, 0.31539156525252005 * (2 * z * z - x * x - y * y),
            -1.0925484305920792 * x * z, 0.5462742152960396 * (x * x - y * y),
            -0.5900435899266435 * y * (3 * x * x - y * y), 2.890611442640554 * x * y * z,
            -0.4570457994644658 * y * (4 * z * z - x * x - y * y), 0.3731763325901154 * z * (2 * z * z - 3 * x * x - 3 * y * y),
            -0.4570457994644658 * x * (4 * z * z - x * x - y * y), 1.445305721320277 * z * (x * x - y * y),
            -0.5900435899266435 * x * (x * x - 3 * y * y)]
    np.testing.assert_allclose(b, want, rtol=1e-6, atol=1e-8)
    # real spherical harmonics are orthonormal: sum_k b_k^2 = (deg+1)^2 / (4 pi) for any unit direction
    for d in ([0, 0, 1], [1, 0, 0], [0.6, 0.0, 0.8], [x, y, z]):
        bb = o.sh_basis(3, d)
        assert abs((bb ** 2).sum() - 16 / (4 * math.pi)) < 1e-6
    # degree gating: bands above the active degree are zero
    assert not o.sh_basis(1, (x, y, z))[4:].any() and not o.sh_basis(0, (x, y, z))[1:].any()


def test_kat2_cov3d():
    """gsplat_vs.hlsl:171-209: identity quaternion -> diag(s^2); 90 degrees about z swaps xx and yy."""
    o = Oracle()
    s = np.array([0.5, 2.0, 3.0])
    np.testing.assert_allclose(o.cov3d(s, [1, 0, 0, 0]), [0.25, 0, 0, 4.0, 0, 9.0], atol=1e-12)
    h = math.sqrt(0.5)
    np.testing.assert_allclose(o.cov3d(s, [h, 0, 0, h]), [4.0, 0, 0, 0.25, 0, 9.0], atol=1e-12)
    # general rotation: Sigma = R diag(s^2) R^T with the matrix of gsplat_vs.hlsl:196-200
    q = np.array([0.3, -0.8, 0.1, 0.5]); q /= np.linalg.norm(q)
    r, x, y, z = q
    R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y)],
                  [2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x)],
                  [2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)]])
    S = R @ np.diag(s ** 2) @ R.T
    np.testing.assert_allclose(o.cov3d(s, q), [S[0, 0], S[0, 1], S[0, 2], S[1, 1], S[1, 2], S[2, 2]], atol=1e-12)
    np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-12)


def test_kat3_depth_bits_are_order_preserving():
    """gaussian_common.hlsl:115-120 (sortable float): for z > 0 the raw IEEE bits already sort like the values,
    which is what the (tile << 32 | depth_bits) key relies on."""
    z = np.sort(np.random.default_rng(0).uniform(0.2, 1e4, 10000).astype(np.float32))
    bits = z.view(np.uint32)
    assert (np.diff(bits.astype(np.int64)) >= 0).all()

    def sortable(f):          # the viewer's general form
        u = np.float32(f).view(np.uint32)
        mask = np.uint32(0xFFFFFFFF) if (u >> 31) else np.uint32(0x80000000)
        return u ^ mask
    assert sortable(-1.0) < sortable(-0.5) < sortable(0.25) < sortable(3.0)


def test_kat4_activations():
    """gaussian_model.cpp:14-22,150-157: sigmoid / exp / quaternion normalisation; and the deterministic exp is
    within 2 ulp of the correctly rounded one over the range the path uses."""
    o = Oracle()
    xs = np.concatenate([np.linspace(-87, 88, 20001), np.random.default_rng(1).normal(0, 3, 20000)]).astype(np.float32)
    got = np.array([o.expf(v) for v in xs], np.float32)
    want = np.exp(xs.astype(np.float64))
    ulp = np.abs(got.astype(np.float64) - want) / np.spacing(want.astype(np.float32)).astype(np.float64)
    assert ulp.max() < 2.0, ulp.max()
    for v in (-20.0, -1.0, 0.0, 0.5, 7.0):
        assert abs(o.sigmoidf(v) - 1 / (1 + math.exp(-v))) < 2e-7
    assert o.sigmoidf(0.0) == 0.5


def test_kat6_sh_layout_and_colour():
    """gaussian_model.cpp:163-167 (shs_n is coefficient-major / channel-minor [j*3+c]); :128,150 colour = SH_C0*dc + 0.5;
    gsplat_sh.hlsl:124 max(colour, 0). One splat straight ahead: only dc and the z-aligned bands contribute."""
    spec = dv.make_spec(1, 32, 32, sh_degree=3)
    cam = dv.synth_camera(spec, 0)
    P = {"pos": np.array([[0, 0, 5.0]], np.float32), "sh0": np.array([[1.0, -3.0, 0.25]], np.float32),
         "shN": np.zeros((1, 15, 3), np.float32), "opacity": np.array([2.0], np.float32),
         "scale": np.full((1, 3), -2.0, np.float32), "rot": np.array([[1, 0, 0, 0]], np.float32)}
    P["shN"][0, 1, 0] = 0.5          # coefficient j=1 (basis b2 = C1*z), channel 0 (red)
    P["shN"][0, 5, 2] = -0.25        # coefficient j=5 (basis b6 = C2_2*(2zz-xx-yy)), channel 2 (blue)
    o = Oracle(np.float64)
    o.forward(P, cam, sh_degree=3)
    rgb = o.get("rgb")[0]
    C0, C1, C22 = 0.28209479177387814, 0.4886025119029199, 0.31539156525252005
    np.testing.assert_allclose(rgb[0], C0 * 1.0 + C1 * 0.5 + 0.5, rtol=1e-6)
    assert rgb[1] == 0.0 and (o.get("flags")[0] & 2)          # C0*(-3)+0.5 < 0 -> clamped, flag bit 1
    np.testing.assert_allclose(rgb[2], C0 * 0.25 + C22 * 2.0 * (-0.25) + 0.5, rtol=1e-6)
    # ndc2Pix (gsplat_vs.hlsl:211-214): the optical axis lands between the two centre pixels
    np.testing.assert_allclose(o.get("mean2d")[0], [15.5, 15.5], atol=1e-5)
    # conic of an isotropic splat: cov2D = (f s / z)^2 + 0.3 on the diagonal
    f, s, z = cam.focal_x, math.exp(-2.0), 5.0
    np.testing.assert_allclose(o.get("conic_opacity")[0, [0, 2]], 1.0 / ((f * s / z) ** 2 + 0.3), rtol=1e-5)
    np.testing.assert_allclose(o.get("conic_opacity")[0, 3], 1 / (1 + math.exp(-2.0)), rtol=1e-6)


@pytest.mark.gpu
def test_kat_on_hip(gpu_device):
    """The same single-splat known answers through the HIP path."""
    import torch
    from divshot_amd.raster import Rasterizer, params_to_device
    spec = dv.make_spec(1, 32, 32, sh_degree=3)
    cam = dv.synth_camera(spec, 0)
    P = {"pos": np.array([[0, 0, 5.0]], np.float32), "sh0": np.array([[1.0, -3.0, 0.25]], np.float32),
         "shN": np.zeros((1, 15, 3), np.float32), "opacity": np.array([2.0], np.float32),
         "scale": np.full((1, 3), -2.0, np.float32), "rot": np.array([[1, 0, 0, 0]], np.float32)}
    P["shN"][0, 1, 0] = 0.5
    r = Rasterizer(0, max_splats=16, max_w=32, max_h=32)
    img = r.forward(params_to_device(P, r.tdev), cam, sh_degree=3)
    torch.cuda.synchronize()
    s = r.saved()
    C0, C1 = 0.28209479177387814, 0.4886025119029199
    np.testing.assert_allclose(s["rgb"][0, 0], C0 + C1 * 0.5 + 0.5, rtol=1e-6)
    assert s["rgb"][0, 1] == 0.0 and (s["flags"][0] & 2)
    np.testing.assert_allclose(s["mean2d"][0], [15.5, 15.5], atol=1e-5)
    # centre pixels: alpha = min(0.99, o * exp(-0.5 * conic * (0.5^2 + 0.5^2)))
    co = s["conic_opacity"][0]
    a = min(0.99, co[3] * math.exp(-0.5 * co[0] * 0.5))
    np.testing.assert_allclose(img[0, 15, 15].item(), a * s["rgb"][0, 0], rtol=1e-5)
    r.close()
