"""Shared helpers for the parity tests."""
import numpy as np
import divshot_amd as dv

KEYS = ("pos", "sh0", "shN", "opacity", "scale", "rot")


def scene(n, w, h, deg=3, seed=1, n_cams=1, cam_index=0, scale_offset=0.0, bg=(0.0, 0.0, 0.0)):
    spec = dv.make_spec(n, w, h, sh_degree=deg, seed=seed, n_cams=n_cams, scale_log_offset=scale_offset)
    P = dv.synth_splats(spec)
    cam = dv.synth_camera(spec, cam_index)
    for k in range(3):
        cam.bg[k] = bg[k]
    tgt = dv.synth_target(spec, cam_index)
    return spec, P, cam, tgt


def rel_close(a, ref, rtol, atol_frac):
    """|a-ref| <= rtol*|ref| + atol_frac*max|ref|  ->  (ok mask, worst normalised error)"""
    a = np.asarray(a, np.float64); ref = np.asarray(ref, np.float64)
    scale = np.abs(ref).max() if ref.size else 0.0
    tol = rtol * np.abs(ref) + atol_frac * scale
    err = np.abs(a - ref)
    worst = float((err / np.maximum(tol, 1e-300)).max()) if ref.size else 0.0
    return err <= tol, worst
