"""Edge-case scenes of SURVEY.md §8(c) as explicit parameter arrays (no scene generator involved): built from closed forms only, so the
same arrays come out on every machine. Used by make_golden.py (fixtures) and tests/test_golden.py (oracle on CPU, HIP path on GPU)."""
import numpy as np
import divshot_amd as dv

W, H, DEG = 80, 48, 1
BG = (0.25, 0.5, 0.75)


def camera():
    spec = dv.make_spec(0, W, H, sh_degree=DEG)
    cam = dv.synth_camera(spec, 0)
    for k in range(3):
        cam.bg[k] = BG[k]
    return cam


def _mk(n):
    return {"pos": np.zeros((n, 3), np.float32), "sh0": np.zeros((n, 3), np.float32), "shN": np.zeros((n, 15, 3), np.float32),
            "opacity": np.zeros((n,), np.float32), "scale": np.full((n, 3), -3.0, np.float32),
            "rot": np.tile(np.array([1, 0, 0, 0], np.float32), (n, 1))}


def _frac(i, k):
    """deterministic low-discrepancy value in [0, 1): fractional part of i * k (k irrational-ish), in float64"""
    return np.mod(np.arange(i, dtype=np.float64) * k, 1.0)


def scenes(cam):
    fx = float(cam.focal_x)
    out = {}
    # E1: nothing visible — behind the camera, inside the near plane (z <= 0.2), zero quaternion, opacity below 1/255, far off screen
    P = _mk(6); P["pos"][:, 2] = [-5, 0.1, 0.2, 5, 5, 5]; P["rot"][3] = 0; P["opacity"][4] = -20.0; P["pos"][5, 0] = 1e4
    out["E1_all_culled"] = P
    # E2: a splat centred exactly on a tile corner (pixel coordinate 15.5), one covering the whole image (wave-cooperative duplication,
    # a colour channel clamped at 0), one anisotropic and rotated
    P = _mk(3); P["pos"][:, 2] = 5.0
    P["pos"][0, 0] = (15.5 - (W - 1) / 2) * 5.0 / fx; P["pos"][0, 1] = (15.5 - (H - 1) / 2) * 5.0 / fx
    P["scale"][1] = 1.5; P["opacity"][1] = 1.0; P["sh0"][1] = [1.0, -4.0, 0.5]
    P["pos"][2] = [0.3, -0.2, 3.0]; P["scale"][2] = [-2.0, -1.0, -4.0]; P["rot"][2] = [0.3, -0.8, 0.1, 0.5]
    out["E2_tile_corner_and_full_cover"] = P
    # E3: 1500 nearly opaque splats stacked on a few pixels: the stack saturates (T < 1e-4) and the tile lists exceed 256 entries
    n = 1500
    P = _mk(n); P["pos"][:, 2] = np.linspace(3, 9, n)
    P["pos"][:, 0] = (0.10 * (_frac(n, 0.6180339887498949) - 0.5)).astype(np.float32)
    P["pos"][:, 1] = (0.10 * (_frac(n, 0.7548776662466927) - 0.5)).astype(np.float32)
    P["opacity"][:] = 3.0; P["scale"][:] = -2.5
    for c, k in enumerate((0.5698402909980532, 0.4142135623730951, 0.7320508075688772)):
        P["sh0"][:, c] = (3.0 * (_frac(n, k) - 0.5)).astype(np.float32)
    out["E3_saturated_stack_long_lists"] = P
    # E4: one tile with more than 65 536 entries (list offsets beyond 16 bits): 70 000 faint small splats inside tile (2, 1)
    n = 70000
    P = _mk(n)
    z = 4.0 + 4.0 * _frac(n, 0.3247179572447460)
    px = 32.0 + 15.0 * _frac(n, 0.6180339887498949); py = 16.0 + 15.0 * _frac(n, 0.7548776662466927)
    P["pos"][:, 2] = z.astype(np.float32)
    P["pos"][:, 0] = ((px - (W - 1) / 2) * z / fx).astype(np.float32)
    P["pos"][:, 1] = ((py - (H - 1) / 2) * z / fx).astype(np.float32)
    P["opacity"][:] = -4.0; P["scale"][:] = -4.2
    for c, k in enumerate((0.5698402909980532, 0.4142135623730951, 0.7320508075688772)):
        P["sh0"][:, c] = (2.0 * (_frac(n, k) - 0.5)).astype(np.float32)
        P["shN"][:, 0, c] = (0.3 * (_frac(n, k * 0.37) - 0.5)).astype(np.float32)
    out["E4_tile_with_over_65536_entries"] = P
    return out


def upstream(shape):
    """deterministic upstream gradient dL/drgb (closed form, no generator)"""
    c, h, w = np.meshgrid(np.arange(shape[0]), np.arange(shape[1]), np.arange(shape[2]), indexing="ij")
    return (np.sin(0.37 * w + 0.61 * h + 1.3 * c) * 1e-3).astype(np.float32)
