"""ctypes wrapper of oracle/_build/libdvs_oracle.so (CPU ORACLE — test infrastructure, not product code).

PARITY UNPINNED: the reference's rasterizer source is absent; see dvs_oracle.hpp.
"""
import ctypes as C
import os
import subprocess
import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "libdvs_oracle.so")
_lib = None


def build(force=False):
    """Compile the oracle with g++ (seconds). Safe to call repeatedly."""
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-C", _HERE] + (["-B"] if force else []), stdout=subprocess.DEVNULL)
    return LIB_PATH


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(LIB_PATH)
        _lib.dvso_set_threads.argtypes = [C.c_int]
        _lib.dvso_create.restype = C.c_void_p
        _lib.dvso_create.argtypes = [C.c_int]
        _lib.dvso_destroy.argtypes = [C.c_void_p]
        _lib.dvso_forward.restype = C.c_int
        _lib.dvso_forward.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 6 + [C.c_void_p, C.c_void_p]
        _lib.dvso_backward.restype = C.c_int
        _lib.dvso_backward.argtypes = [C.c_void_p, C.c_void_p]
        _lib.dvso_set_grad_mode.argtypes = [C.c_void_p, C.c_int]
        _lib.dvso_record_masks.argtypes = [C.c_void_p, C.c_int]
        _lib.dvso_set_replay.restype = C.c_int
        _lib.dvso_set_replay.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64]
        _lib.dvso_set_lists.restype = C.c_int
        _lib.dvso_set_lists.argtypes = [C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_uint64]
        _lib.dvso_array.restype = C.c_void_p
        _lib.dvso_array.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_uint64), C.POINTER(C.c_int)]
        _lib.dvso_interactions.restype = C.c_uint64
        _lib.dvso_interactions.argtypes = [C.c_void_p]
        _lib.dvso_expf.restype = C.c_float
        _lib.dvso_expf.argtypes = [C.c_float]
        _lib.dvso_sigmoidf.restype = C.c_float
        _lib.dvso_sigmoidf.argtypes = [C.c_float]
        _lib.dvso_sh_basis.argtypes = [C.c_int, C.c_double, C.c_double, C.c_double, C.c_void_p]
        _lib.dvso_cov3d.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    return _lib


_INT_DTYPES = {"radii": np.int32, "rect": np.int32, "flags": np.uint32, "tiles_touched": np.uint32,
               "depth_bits": np.uint32, "offsets": np.uint32, "keys": np.uint64, "vals": np.uint32,
               "ranges": np.uint32, "n_contrib": np.uint32, "fragile": np.uint8, "cap_fragile": np.uint8, "take_masks": np.uint64}
_SHAPES = {"mean2d": (-1, 2), "conic_opacity": (-1, 4), "rgb": (-1, 3), "rect": (-1, 4), "ranges": (-1, 2),
           "dL_dmean2d": (-1, 2), "dL_dconic_opacity": (-1, 4), "dL_drgb": (-1, 3), "absgrad": (-1, 2),
           "take_masks": (-1, 4), "g_pos": (-1, 3), "g_sh0": (-1, 3), "g_shN": (-1, 15, 3), "g_scale": (-1, 3), "g_rot": (-1, 4)}


def set_threads(n):
    """OpenMP threads used by the oracle from now on (0 = all cores)."""
    _load().dvso_set_threads(int(n))


class Oracle:
    """One oracle state. dtype float32 = the specification the HIP path must match; float64 = ground truth."""

    def __init__(self, dtype=np.float32):
        self.lib = _load()
        self.dtype = np.dtype(dtype)
        assert self.dtype in (np.dtype(np.float32), np.dtype(np.float64))
        self.h = self.lib.dvso_create(1 if self.dtype == np.float64 else 0)
        self.W = self.H = 0

    def __del__(self):
        if getattr(self, "h", None):
            self.lib.dvso_destroy(self.h)
            self.h = None

    def forward(self, params, cam, sh_degree=3, antialias=False, absgrad=False, grad_mode=0, tight_tiles=False):
        """params: dict of numpy arrays (A0 layout); cam: divshot_amd.Camera (ctypes struct, same layout as dvs_camera)."""
        arrs = [np.ascontiguousarray(params[k], dtype=self.dtype) for k in ("pos", "sh0", "shN", "opacity", "scale", "rot")]
        n = arrs[0].shape[0]
        opts = (C.c_int32 * 8)(sh_degree, int(antialias), int(absgrad), 0, 0, int(grad_mode), int(bool(tight_tiles)), 0)     # dvs_opts
        # (grad_mode: 0 = DVS_GRAD_TRUE, 1 = DVS_GRAD_LINEAGE — only the backward differs)
        self.W, self.H = cam.width, cam.height
        rc = self.lib.dvso_forward(self.h, n, *[a.ctypes.data for a in arrs], C.addressof(cam), C.addressof(opts))
        assert rc == 0, {2: "the recorded decisions do not belong to these tile lists", 3: "set_lists: a listed splat is culled in this precision",
                         4: "set_lists: the ranges are not this image's"}.get(rc, rc)
        return self.get("out_color").reshape(3, self.H, self.W)

    def record_masks(self, on=True):
        """Every forward from now on records its own decisions: get("take_masks") -> uint64 [T, 4] (the layout of the HIP hook)."""
        self.lib.dvso_record_masks(self.h, int(bool(on)))

    def set_replay(self, masks):
        """Decision replay: masks = uint64 [T, 4] recorded by the HIP forward (Rasterizer.record_decisions) — bit l of masks[j, q] says pixel
        lane l of 8x8 quadrant q of the tile takes list entry j. From now on forward / backward use these decisions instead of evaluating
        the alpha / transmittance thresholds (None or an empty array: back to the oracle's own decisions)."""
        m = np.ascontiguousarray(masks if masks is not None else np.zeros((0, 4)), dtype=np.uint64)
        assert self.lib.dvso_set_replay(self.h, m.ctypes.data, m.size) == 0

    def set_lists(self, vals, ranges):
        """From now on forward() composites over these tile lists (vals: splat ids in list order; ranges: [tiles, 2]) instead of binning
        the scene itself — e.g. the float64 oracle over the float32 run's lists when a radius on an integer boundary makes float64 bin a
        splat differently. None, None: back to its own binning."""
        v = np.ascontiguousarray(vals if vals is not None else np.zeros(0), dtype=np.uint32)
        r = np.ascontiguousarray(ranges if ranges is not None else np.zeros(0), dtype=np.uint32)
        assert self.lib.dvso_set_lists(self.h, v.ctypes.data, v.size, r.ctypes.data, r.size) == 0

    def backward(self, dL_dout, grad_mode=None):
        """grad_mode: None = the mode given to forward(); 0 = DVS_GRAD_TRUE, 1 = DVS_GRAD_LINEAGE (the forward does not depend on it)."""
        if grad_mode is not None:
            self.lib.dvso_set_grad_mode(self.h, int(grad_mode))
        g = np.ascontiguousarray(dL_dout, dtype=self.dtype)
        assert g.shape == (3, self.H, self.W)
        rc = self.lib.dvso_backward(self.h, g.ctypes.data)
        assert rc == 0
        return {k: self.get("g_" + k) for k in ("pos", "sh0", "shN", "opacity", "scale", "rot")}

    def get(self, name):
        cnt, eb = C.c_uint64(0), C.c_int(0)
        p = self.lib.dvso_array(self.h, name.encode(), C.byref(cnt), C.byref(eb))
        if eb.value == 0:
            raise KeyError(name)
        dt = _INT_DTYPES.get(name, self.dtype)
        assert np.dtype(dt).itemsize == eb.value, (name, dt, eb.value)
        if cnt.value == 0:
            a = np.empty((0,), dt)
        else:
            a = np.ctypeslib.as_array(C.cast(p, C.POINTER(C.c_byte)), shape=(cnt.value * eb.value,)).view(dt).copy()
        if name in ("final_T", "n_contrib", "fragile", "cap_fragile"):
            return a.reshape(self.H, self.W)
        if name in _SHAPES:
            return a.reshape(_SHAPES[name])
        return a

    @property
    def interactions(self):
        return int(self.lib.dvso_interactions(self.h))

    # scalar helpers for the known-answer tests
    def expf(self, x):
        return float(self.lib.dvso_expf(float(x)))

    def sigmoidf(self, x):
        return float(self.lib.dvso_sigmoidf(float(x)))

    def sh_basis(self, deg, d):
        out = np.zeros(16, np.float64)
        self.lib.dvso_sh_basis(deg, float(d[0]), float(d[1]), float(d[2]), out.ctypes.data)
        return out

    def cov3d(self, scale, quat_unit):
        s = np.ascontiguousarray(scale, np.float64)
        q = np.ascontiguousarray(quat_unit, np.float64)
        out = np.zeros(6, np.float64)
        self.lib.dvso_cov3d(s.ctypes.data, q.ctypes.data, out.ctypes.data)
        return out
