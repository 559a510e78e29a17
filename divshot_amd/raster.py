"""Rasterizer — Python handle over the dvs_* C-ABI (include/dvs_raster.h).

torch supplies device buffers (`tensor.data_ptr()`) and the HIP stream; all compute runs in
libdvsraster.so. Mirrors the two-op surface of the reference's rasterizer library (forward ->
image + saved state, backward -> per-splat gradient rows).
"""
import ctypes as C
import numpy as np
import torch
from ._lib import lib, Splats, Camera, Opts, FwdState, SplatGrads, check, DvsError

PARAM_KEYS = ("pos", "sh0", "shN", "opacity", "scale", "rot")
PARAM_WIDTH = {"pos": 3, "sh0": 3, "shN": 45, "opacity": 1, "scale": 3, "rot": 4}


def tiled_floats(n):
    """Number of floats of an shN array in the DVS_SHN_TILED layout ([ceil(n/64)][12][64][4])."""
    return ((n + 63) // 64) * 64 * 48


def shn_rows_to_tiled_np(a):
    """numpy definition of the tiled layout: [n,15,3] (or [n,45]) -> flat [ceil(n/64)*64*48]."""
    n = a.shape[0]
    nt = (n + 63) // 64
    pad = np.zeros((nt * 64, 48), a.dtype)
    pad[:n, :45] = a.reshape(n, 45)
    return np.ascontiguousarray(pad.reshape(nt, 64, 12, 4).transpose(0, 2, 1, 3)).reshape(-1)


def shn_tiled_to_rows_np(t, n):
    nt = (n + 63) // 64
    rows = np.ascontiguousarray(np.asarray(t).reshape(nt, 12, 64, 4).transpose(0, 2, 1, 3)).reshape(nt * 64, 48)
    return rows[:n, :45].reshape(n, 15, 3)


def _stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Rasterizer:
    def __init__(self, device=0, max_splats=1 << 20, max_w=1920, max_h=1080, max_views=1):
        if not torch.cuda.is_available():
            raise DvsError("no HIP device visible: the rasterizer has no CPU fallback")
        self.device = device
        self.tdev = torch.device("cuda", device)
        self.max_views = max_views
        self.ctx = lib.dvs_create_views(device, max_splats, max_w, max_h, max_views)
        if not self.ctx:
            raise DvsError("dvs_create failed: " + lib.dvs_last_error().decode())
        self.state = None
        self._cam = None
        self._opts = None
        self._params = None

    def close(self):
        if self.ctx:
            lib.dvs_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers ---------------------------------------------------------------------------
    def _splats(self, params, tiled=False):
        n = params["pos"].shape[0]
        for k in PARAM_KEYS:
            t = params[k]
            assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), k
            want = tiled_floats(n) if (tiled and k == "shN") else n * PARAM_WIDTH[k]
            assert t.numel() == want, (k, t.numel(), want)
        return Splats(*[params[k].data_ptr() for k in PARAM_KEYS], n, 0)

    def shn_relayout(self, src, n, to_tiled):
        """DEVICE relayout of an shN array between the reference rows [n,45] and the tiled layout (returns a new tensor)."""
        dst = torch.zeros(tiled_floats(n) if to_tiled else n * 45, dtype=torch.float32, device=self.tdev)
        with torch.cuda.device(self.tdev):
            check(lib.dvs_shn_relayout(self.ctx, _stream_ptr(), n, src.data_ptr(), dst.data_ptr(), 1 if to_tiled else 0), "dvs_shn_relayout")
        return dst if to_tiled else dst.view(n, 15, 3)

    def keep_intermediates(self, on=True):
        check(lib.dvs_keep_bwd_intermediates(self.ctx, 1 if on else 0))

    def set_async(self, on=True):
        """dvs_set_async: forward never synchronises the stream; T stays on the device (num_rendered unknown until get_num_rendered())."""
        check(lib.dvs_set_async(self.ctx, 1 if on else 0), "dvs_set_async")

    def get_num_rendered(self):
        """Synchronises the current stream and returns T of the last forward (raises DvsError on an instance-arena overflow)."""
        T = C.c_uint64(0)
        with torch.cuda.device(self.tdev):
            check(lib.dvs_get_num_rendered(self.ctx, _stream_ptr(), C.byref(T)), "dvs_get_num_rendered")
        self.num_rendered = int(T.value)
        if self.state is not None:
            self.state.num_rendered = self.num_rendered
        return self.num_rendered

    def set_backward_variant(self, variant):
        """A8 kernel: 3 / "tr" (default, render_tr.hip), 0 / "blocks" (round 2), 1 / "reduce" (round 1), 2 / "mm" — see dvs_raster.h."""
        v = {"blocks": 0, "reduce": 1, "mm": 2, "tr": 3}.get(variant, variant)
        check(lib.dvs_set_backward_variant(self.ctx, int(v)), "dvs_set_backward_variant")

    def set_forward_variant(self, variant):
        """A7 kernel: 1 / "quadrant" (default), 0 / "blocks" (experiment); bit-identical results."""
        v = {"blocks": 0, "quadrant": 1}.get(variant, variant)
        check(lib.dvs_set_forward_variant(self.ctx, int(v)), "dvs_set_forward_variant")

    def set_live_lists(self, on=True):
        """dvs_set_live_lists: the forward's compacted per-tile lists for the "tr" backward (default on)."""
        check(lib.dvs_set_live_lists(self.ctx, 1 if on else 0), "dvs_set_live_lists")

    def record_decisions(self, capacity_instances):
        """dvs_debug_record_decisions (test hook): single-view synchronous forwards from now on record which pixel took which list entry;
        returns the device tensor [capacity, 4] (uint64 as int64) they are written to. capacity 0 / None switches it off."""
        if not capacity_instances:
            check(lib.dvs_debug_record_decisions(self.ctx, None, 0), "dvs_debug_record_decisions")
            self._rec = None
            return None
        self._rec = torch.zeros((int(capacity_instances), 4), dtype=torch.int64, device=self.tdev)
        check(lib.dvs_debug_record_decisions(self.ctx, self._rec.data_ptr(), int(capacity_instances)), "dvs_debug_record_decisions")
        return self._rec

    def enable_timing(self, on=True):
        check(lib.dvs_enable_stage_timing(self.ctx, 1 if on else 0))

    def kernel_probe(self, on=True):
        """hipEvent pairs around k_render_fwd / k_render_bwd on the caller's stream, no synchronisation (dvs_enable_kernel_probe)."""
        check(lib.dvs_enable_kernel_probe(self.ctx, 1 if on else 0))

    def read_kernel_probe(self):
        """-> {"render_fwd": (mean_ms, launches), "render_bwd": (mean_ms, launches)} since the probe was enabled / last read."""
        ms, cnt = (C.c_float * 2)(), (C.c_int * 2)()
        check(lib.dvs_read_kernel_probe(self.ctx, ms, cnt), "dvs_read_kernel_probe")
        return {"render_fwd": (float(ms[0]), int(cnt[0])), "render_bwd": (float(ms[1]), int(cnt[1]))}

    def stage_timing(self):
        names = C.POINTER(C.c_char_p)()
        ms = C.POINTER(C.c_float)()
        k = lib.dvs_get_stage_timing(self.ctx, C.byref(names), C.byref(ms))
        return {names[i].decode(): float(ms[i]) for i in range(k)}

    # -- the two ops -------------------------------------------------------------------------
    def forward(self, params, cam, sh_degree=3, antialias=False, absgrad=False, out=None, shn_tiled=False, grad_mode=0, tight_tiles=False):
        """params: dict of CUDA float32 tensors (A0 layout; params["shN"] in the tiled layout when shn_tiled).
        grad_mode: 0 = DVS_GRAD_TRUE, 1 = DVS_GRAD_LINEAGE (dvs_raster.h) — used by the backward calls that follow.
        tight_tiles: dvs_opts.tile_bounds = DVS_TILES_TIGHT (opt-in; the canonical 3-sigma rectangles are the default).
        Returns out_rgb [3,H,W] (CUDA)."""
        sp = self._splats(params, shn_tiled)
        opts = Opts(sh_degree, int(antialias), int(absgrad), 0, int(shn_tiled), int(grad_mode), int(bool(tight_tiles)))
        self._tiled = bool(shn_tiled)
        if out is None:
            out = torch.empty((3, cam.height, cam.width), dtype=torch.float32, device=self.tdev)
        st = FwdState()
        T = C.c_uint64(0)
        with torch.cuda.device(self.tdev):
            check(lib.dvs_raster_forward(self.ctx, _stream_ptr(), C.byref(sp), C.byref(cam), C.byref(opts),
                                         out.data_ptr(), C.byref(st), C.byref(T)), "dvs_raster_forward")
        self.state, self._cam, self._opts, self._params = st, cam, opts, params
        self._cams = None                              # (single view: the split backward calls take the one camera)
        self.num_rendered = int(T.value)
        return out

    # -- multi-view batches (dvs_raster_forward_views / dvs_raster_backward_views) -----------------------------------
    def forward_views(self, params, cams, sh_degree=3, antialias=False, absgrad=False, out=None, shn_tiled=False, grad_mode=0, tight_tiles=False):
        """cams: list of Camera (same image size). Returns out_rgb [V,3,H,W] (CUDA). One depth sort / scan / (view, tile) sort /
        composite launch for the whole batch; the parameters are read once."""
        V = len(cams)
        sp = self._splats(params, shn_tiled)
        opts = Opts(sh_degree, int(antialias), int(absgrad), 0, int(shn_tiled), int(grad_mode), int(bool(tight_tiles)))
        self._tiled = bool(shn_tiled)
        if out is None:
            out = torch.empty((V, 3, cams[0].height, cams[0].width), dtype=torch.float32, device=self.tdev)
        carr = (Camera * V)(*cams)
        with torch.cuda.device(self.tdev):
            check(lib.dvs_raster_forward_views(self.ctx, _stream_ptr(), C.byref(sp), carr, V, C.byref(opts), out.data_ptr()), "dvs_raster_forward_views")
        st = FwdState()
        check(lib.dvs_get_view_state(self.ctx, 0, C.byref(st)), "dvs_get_view_state")
        self.state, self._cam, self._cams, self._opts, self._params = st, cams[0], carr, opts, params
        self.num_rendered = int(st.num_rendered)
        return out

    def backward_views(self, dL_drgb, grads=None, accumulate=False, want_mean2d=False, factorised_sh=False):
        """dL_drgb: CUDA [V,3,H,W]. Gradient rows = the sum over the views of the last forward_views; grads["dcolor"] (factorised_sh or
        given) is [V,n,3]."""
        V = len(self._cams)
        n = self._params["pos"].shape[0]
        if factorised_sh and (grads is None or "dcolor" not in grads):
            grads = dict(grads or {}); grads["dcolor"] = torch.empty((V, n, 3), dtype=torch.float32, device=self.tdev)
        had = grads is not None and any(k in grads for k in PARAM_KEYS)
        if not had:
            base = {k: (torch.zeros_like(self._params[k]) if (k == "shN" and self._tiled) else torch.empty_like(self._params[k])) for k in PARAM_KEYS}
            base.update(grads or {}); grads = base; accumulate = False
        grads, opts, g = self._grad_args(grads, accumulate, want_mean2d, factorised_sh)
        sp = self._splats(self._params, self._tiled)
        assert dL_drgb.is_cuda and dL_drgb.dtype == torch.float32 and dL_drgb.is_contiguous() and dL_drgb.shape[0] == V
        with torch.cuda.device(self.tdev):
            check(lib.dvs_raster_backward_views(self.ctx, _stream_ptr(), C.byref(sp), self._cams, V, C.byref(opts), dL_drgb.data_ptr(), C.byref(g)),
                  "dvs_raster_backward_views")
        return grads

    def view_saved(self, v):
        """Host copies of view v's saved arrays of the last forward_views; the instance list of the view is cut out of the batch-wide
        sorted lists with its tile ranges and its values are made view-local splat ids again (comparable with a single-view run)."""
        st = FwdState()
        check(lib.dvs_get_view_state(self.ctx, v, C.byref(st)), "dvs_get_view_state")
        if self.state.num_rendered == 2 ** 64 - 1:
            self.get_num_rendered()
        n, W, H, tiles = st.n, st.width, st.height, st.tiles_x * st.tiles_y
        T = self.state.num_rendered
        rec = self._d2h(st.splat2d, (n, 16), np.float32)
        ranges = self._d2h(st.ranges, (tiles, 2), np.uint32).astype(np.int64)
        st0 = FwdState()
        check(lib.dvs_get_view_state(self.ctx, 0, C.byref(st0)), "dvs_get_view_state")          # (the batch-wide ranges start at view 0)
        all_tile = self._sorted_tile(st0, T, tiles * len(self._cams)); all_splat = self._d2h(st.sorted_splat, (T,), np.uint32)
        nz = ranges[:, 1] > ranges[:, 0]
        lo = int(ranges[nz, 0].min()) if nz.any() else 0
        hi = int(ranges[nz, 1].max()) if nz.any() else 0
        loc = ranges.copy(); loc[nz] -= lo; loc[~nz] = 0
        return {"radii": self._d2h(st.radii, (n,), np.int32), "mean2d": rec[:, 0:2].copy(), "depth": self._d2h(st.depth, (n,), np.float32),
                "conic_opacity": rec[:, 2:6].copy(), "rgb": rec[:, 6:9].copy(), "flags": self._d2h(st.flags, (n,), np.uint32),
                "tiles_touched": self._d2h(st.tiles_touched, (n,), np.uint32),
                "sorted_tile": all_tile[lo:hi] - np.uint32(v * tiles), "vals": all_splat[lo:hi] - np.uint32(v * n),
                "ranges": loc.astype(np.uint32), "final_T": self._d2h(st.final_T, (H, W), np.float32),
                "n_contrib": self._d2h(st.n_contrib, (H, W), np.uint32)}

    def _grad_args(self, grads, accumulate, want_mean2d, factorised_sh):
        params = self._params
        n = params["pos"].shape[0]
        if grads is None:
            # the tiled shN array has pad lanes (splats >= n of the last tile) that the kernels never write: keep them defined
            grads = {k: (torch.zeros_like(params[k]) if (k == "shN" and self._tiled) else torch.empty_like(params[k])) for k in PARAM_KEYS}
            accumulate = False
        opts = Opts(self._opts.sh_degree, self._opts.antialias, self._opts.absgrad, int(accumulate), self._opts.shn_layout, self._opts.grad_mode,
                    self._opts.tile_bounds)
        if self._opts.absgrad and "absgrad2d" not in grads:
            grads["absgrad2d"] = torch.empty((n, 2), dtype=torch.float32, device=self.tdev)
        if want_mean2d and "mean2d" not in grads:
            grads["mean2d"] = torch.empty((n, 2), dtype=torch.float32, device=self.tdev)
        if factorised_sh and "dcolor" not in grads:
            grads["dcolor"] = torch.empty((n, 3), dtype=torch.float32, device=self.tdev)

        def ptr(k):
            if factorised_sh and k in ("sh0", "shN"):
                return None
            return grads[k].data_ptr()
        g = SplatGrads(*[ptr(k) for k in PARAM_KEYS],
                       grads["absgrad2d"].data_ptr() if "absgrad2d" in grads else None,
                       grads["mean2d"].data_ptr() if "mean2d" in grads else None,
                       grads["dcolor"].data_ptr() if factorised_sh else None)
        return grads, opts, g

    def backward(self, dL_drgb, grads=None, accumulate=False, want_mean2d=False, factorised_sh=False):
        """dL_drgb: CUDA [3,H,W]. Returns dict of gradient tensors shaped like params (+absgrad2d / mean2d).
        factorised_sh=True: the sh0/shN rows are NOT written; grads["dcolor"] [n,3] is (see sh_grad_combine)."""
        assert self.state is not None, "forward first"
        grads, opts, g = self._grad_args(grads, accumulate, want_mean2d, factorised_sh)
        sp = self._splats(self._params, self._tiled)
        assert dL_drgb.is_cuda and dL_drgb.dtype == torch.float32 and dL_drgb.is_contiguous()
        with torch.cuda.device(self.tdev):
            check(lib.dvs_raster_backward(self.ctx, _stream_ptr(), C.byref(sp), C.byref(self._cam), C.byref(opts),
                                          dL_drgb.data_ptr(), C.byref(g)), "dvs_raster_backward")
        return grads

    def backward_composite(self, dL_drgb):
        """A8 alone (dvs_raster_backward_composite): touches only this context's intermediate rows."""
        assert self.state is not None, "forward first"
        assert dL_drgb.is_cuda and dL_drgb.dtype == torch.float32 and dL_drgb.is_contiguous()
        cam_arg = self._cams if getattr(self, "_cams", None) is not None else C.byref(self._cam)      # after forward_views: all its cameras
        with torch.cuda.device(self.tdev):
            check(lib.dvs_raster_backward_composite(self.ctx, _stream_ptr(), cam_arg, C.byref(self._opts), dL_drgb.data_ptr()),
                  "dvs_raster_backward_composite")

    def backward_dcolor(self, dcolor):
        """Between backward_composite and backward_project: the per-view colour gradients [V,n,3] (or [n,3]) straight from the A8
        rows (dvs_raster_backward_dcolor) — what backward_project(factorised_sh=True) writes to grads["dcolor"], bit for bit."""
        assert dcolor.is_cuda and dcolor.dtype == torch.float32 and dcolor.is_contiguous()
        with torch.cuda.device(self.tdev):
            check(lib.dvs_raster_backward_dcolor(self.ctx, _stream_ptr(), dcolor.data_ptr()), "dvs_raster_backward_dcolor")
        return dcolor

    def backward_project(self, grads=None, accumulate=False, want_mean2d=False, factorised_sh=False):
        """A9 alone (dvs_raster_backward_project) after backward_composite; same arguments and result as backward()."""
        grads, opts, g = self._grad_args(grads, accumulate, want_mean2d, factorised_sh)
        sp = self._splats(self._params, self._tiled)
        cam_arg = self._cams if getattr(self, "_cams", None) is not None else C.byref(self._cam)
        with torch.cuda.device(self.tdev):
            check(lib.dvs_raster_backward_project(self.ctx, _stream_ptr(), C.byref(sp), cam_arg, C.byref(opts), C.byref(g)),
                  "dvs_raster_backward_project")
        return grads

    def backward_project_chunks(self, grads, chunks, after_chunk, accumulate=False, want_mean2d=False):
        """A9 in splat chunks (dvs_raster_backward_project_chunk; factorised form: grads["dcolor"] given, SH rows rebuilt later).
        chunks: [(first, count)] covering [0, n) in order; after_chunk(k, first, count) runs right behind chunk k's launch on the
        current stream — a data-parallel step records an event there and starts that chunk's gradient all-reduce on its side stream."""
        grads, opts, g = self._grad_args(grads, accumulate, want_mean2d, True)
        sp = self._splats(self._params, self._tiled)
        cam_arg = self._cams if getattr(self, "_cams", None) is not None else C.byref(self._cam)
        with torch.cuda.device(self.tdev):
            for k, (first, count) in enumerate(chunks):
                check(lib.dvs_raster_backward_project_chunk(self.ctx, _stream_ptr(), C.byref(sp), cam_arg, C.byref(opts), C.byref(g),
                                                            int(first), int(count)), "dvs_raster_backward_project_chunk")
                if after_chunk is not None:
                    after_chunk(k, first, count)
        return grads

    def sh_grad_combine(self, pos, campos, dcolor_all, g_sh0, g_shN, sh_degree, accumulate=False, shn_tiled=False):
        """g_sh0/g_shN (+)= sum over views of the SH rows implied by dcolor_all [V,n,3]; campos: [V,3] host floats."""
        campos = np.ascontiguousarray(campos, np.float32)
        V, n = dcolor_all.shape[0], pos.shape[0]
        assert campos.shape == (V, 3) and dcolor_all.is_contiguous() and dcolor_all.shape == (V, n, 3)
        with torch.cuda.device(self.tdev):
            check(lib.dvs_sh_grad_combine(self.ctx, _stream_ptr(), n, pos.data_ptr(), sh_degree, V, campos.ctypes.data,
                                          dcolor_all.data_ptr(), g_sh0.data_ptr(), g_shN.data_ptr(), int(accumulate), int(shn_tiled)),
                  "dvs_sh_grad_combine")

    # -- stage-level access for the parity tests --------------------------------------------------
    def _sorted_tile(self, st, T, total_tiles):
        """dvs_fwd_state.sorted_tile — NULL after an asynchronous forward (the tile ids are not materialised then: A6 rides on the last
        sort pass and nothing on the device reads them); the sorted list is grouped by tile, so it follows from the ranges."""
        if st.sorted_tile:
            return self._d2h(st.sorted_tile, (T,), np.uint32)
        rg = self._d2h(st.ranges, (total_tiles, 2), np.uint32).astype(np.int64)
        out = np.repeat(np.arange(total_tiles, dtype=np.uint32), (rg[:, 1] - rg[:, 0]))
        assert out.size == T, (out.size, T)
        return out

    def _d2h(self, ptr, shape, dtype):
        a = np.empty(shape, dtype)
        if a.nbytes:
            check(lib.dvs_memcpy_d2h(self.ctx, a.ctypes.data, C.c_void_p(ptr), a.nbytes), "dvs_memcpy_d2h")
        return a

    def saved(self):
        """Host copies of every saved forward array (dict of numpy arrays)."""
        s = self.state
        if s.num_rendered == 2 ** 64 - 1:          # DVS_T_UNKNOWN (asynchronous forward): ask for it
            self.get_num_rendered()
        n, T, W, H = s.n, s.num_rendered, s.width, s.height
        tiles = s.tiles_x * s.tiles_y
        rec = self._d2h(s.splat2d, (n, 16), np.float32)          # the packed 64-B record (DVS_S2D_* offsets)
        return {
            "radii": self._d2h(s.radii, (n,), np.int32), "mean2d": rec[:, 0:2].copy(),
            "depth": self._d2h(s.depth, (n,), np.float32), "conic_opacity": rec[:, 2:6].copy(),
            "rgb": rec[:, 6:9].copy(), "splat2d": rec, "flags": self._d2h(s.flags, (n,), np.uint32),
            "tiles_touched": self._d2h(s.tiles_touched, (n,), np.uint32),
            "sorted_tile": self._sorted_tile(s, T, tiles), "vals": self._d2h(s.sorted_splat, (T,), np.uint32),
            "ranges": self._d2h(s.ranges, (tiles, 2), np.uint32), "final_T": self._d2h(s.final_T, (H, W), np.float32),
            "n_contrib": self._d2h(s.n_contrib, (H, W), np.uint32),
        }

    def sorted_keys(self):
        if self.state.num_rendered == 2 ** 64 - 1:
            self.get_num_rendered()
        T = self.state.num_rendered
        buf = torch.empty((max(T, 1),), dtype=torch.int64, device=self.tdev)
        check(lib.dvs_export_sorted_keys(self.ctx, _stream_ptr(), buf.data_ptr()), "dvs_export_sorted_keys")
        torch.cuda.synchronize(self.tdev)
        return buf[:T].cpu().numpy().view(np.uint64)

    def bwd_intermediates(self):
        rows, width = C.c_void_p(), C.c_int(0)
        check(lib.dvs_get_bwd_intermediates(self.ctx, C.byref(rows), C.byref(width)))
        n = self.state.n
        r = self._d2h(rows.value, (n, width.value), np.float32).astype(np.float64)
        # A8 publishes moments about the mean; dvs_raster.h (dvs_get_bwd_intermediates) gives the conversion A9 applies
        rec = self._d2h(self.state.splat2d, (n, 16), np.float32).astype(np.float64)
        a, b, c = rec[:, 2], rec[:, 3], rec[:, 4]
        dmean = np.stack([-(a * r[:, 0] + b * r[:, 1]), -(c * r[:, 1] + b * r[:, 0])], 1)
        dconic = np.stack([-0.5 * r[:, 2], -r[:, 3], -0.5 * r[:, 4], r[:, 5]], 1)
        return {"dL_dmean2d": dmean.astype(np.float32), "dL_dconic_opacity": dconic.astype(np.float32),
                "dL_drgb": r[:, 6:9].astype(np.float32), "absgrad": r[:, 9:11].astype(np.float32)}

    def sort_pairs(self, keys, vals, bit_lo=0, bit_hi=32):
        """In-place stable LSD radix sort of CUDA uint32-as-int32 tensors."""
        assert keys.is_cuda and vals.is_cuda and keys.numel() == vals.numel()
        check(lib.dvs_sort_pairs_u32(self.ctx, _stream_ptr(), keys.data_ptr(), vals.data_ptr(), keys.numel(), bit_lo, bit_hi),
              "dvs_sort_pairs_u32")


def params_to_device(params_np, device):
    return {k: torch.from_numpy(np.ascontiguousarray(v)).to(device) for k, v in params_np.items()}
