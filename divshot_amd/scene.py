"""Synthetic scenes through the C generator in libdvsraster.so (include/dvs_scene.h)."""
import ctypes as C
import numpy as np
from ._lib import lib, SceneSpec, Camera, check


def make_spec(n, width, height, sh_degree=3, n_cams=1, seed=1, fov_x_deg=60.0, scale_log_offset=0.0):
    return SceneSpec(n=n, width=width, height=height, sh_degree=sh_degree, n_cams=n_cams, seed=seed,
                     fov_x_deg=fov_x_deg, scale_log_offset=scale_log_offset)


def synth_splats(spec):
    """-> dict of float32 numpy arrays in the A0 layout."""
    n = spec.n
    out = {"pos": np.empty((n, 3), np.float32), "sh0": np.empty((n, 3), np.float32),
           "shN": np.empty((n, 15, 3), np.float32), "opacity": np.empty((n,), np.float32),
           "scale": np.empty((n, 3), np.float32), "rot": np.empty((n, 4), np.float32)}
    check(lib.dvs_synth_splats(C.byref(spec), *[out[k].ctypes.data for k in ("pos", "sh0", "shN", "opacity", "scale", "rot")]),
          "dvs_synth_splats")
    return out


def synth_camera(spec, index=0):
    cam = Camera()
    check(lib.dvs_synth_camera(C.byref(spec), index, C.byref(cam)), "dvs_synth_camera")
    return cam


def synth_target(spec, index=0):
    t = np.empty((3, spec.height, spec.width), np.float32)
    check(lib.dvs_synth_target(C.byref(spec), index, t.ctypes.data), "dvs_synth_target")
    return t
