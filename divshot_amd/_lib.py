"""ctypes loader for libdvsraster.so — mirrors include/dvs_raster.h and include/dvs_scene.h 1:1."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DVS_RASTER_LIB") or os.path.join(_HERE, "lib", "libdvsraster.so")     # override: experiment builds (tools/)


class DvsError(RuntimeError):
    pass


if not os.path.exists(LIB_PATH):
    raise ImportError(
        f"{LIB_PATH} is missing: build it with `make -C divshot_amd/csrc` (or __graft_entry__.build()). "
        "divshot_amd has no CPU fallback.")

lib = C.CDLL(LIB_PATH, mode=C.RTLD_GLOBAL)


class Splats(C.Structure):
    _fields_ = [("pos", C.c_void_p), ("sh0", C.c_void_p), ("shN", C.c_void_p), ("opacity", C.c_void_p),
                ("scale", C.c_void_p), ("rot", C.c_void_p), ("n", C.c_int32), ("_pad", C.c_int32)]


class Camera(C.Structure):
    _fields_ = [("view", C.c_float * 16), ("proj", C.c_float * 16), ("tan_fovx", C.c_float), ("tan_fovy", C.c_float),
                ("focal_x", C.c_float), ("focal_y", C.c_float), ("campos", C.c_float * 3), ("width", C.c_int32),
                ("height", C.c_int32), ("bg", C.c_float * 3)]


class Opts(C.Structure):
    _fields_ = [("sh_degree", C.c_int32), ("antialias", C.c_int32), ("absgrad", C.c_int32), ("accumulate", C.c_int32),
                ("shn_layout", C.c_int32), ("grad_mode", C.c_int32), ("tile_bounds", C.c_int32), ("_reserved", C.c_int32 * 1)]


class FwdState(C.Structure):
    _fields_ = [("radii", C.c_void_p), ("splat2d", C.c_void_p), ("depth", C.c_void_p), ("flags", C.c_void_p), ("tiles_touched", C.c_void_p), ("sorted_tile", C.c_void_p),
                ("sorted_splat", C.c_void_p), ("ranges", C.c_void_p), ("final_T", C.c_void_p), ("n_contrib", C.c_void_p),
                ("num_rendered", C.c_uint64), ("n", C.c_int32), ("width", C.c_int32), ("height", C.c_int32),
                ("tiles_x", C.c_int32), ("tiles_y", C.c_int32), ("_pad", C.c_int32)]


class SplatGrads(C.Structure):
    _fields_ = [("pos", C.c_void_p), ("sh0", C.c_void_p), ("shN", C.c_void_p), ("opacity", C.c_void_p),
                ("scale", C.c_void_p), ("rot", C.c_void_p), ("absgrad2d", C.c_void_p), ("mean2d", C.c_void_p), ("dcolor", C.c_void_p)]


class DensifyParams(C.Structure):
    _fields_ = [("grad_threshold", C.c_float), ("scale_threshold", C.c_float), ("min_opacity", C.c_float), ("max_world_scale", C.c_float),
                ("max_screen_radius", C.c_int32), ("cap_max", C.c_int32), ("seed", C.c_uint32), ("shn_layout", C.c_int32), ("revised_opacity", C.c_int32)]


class AdamGroup(C.Structure):
    _fields_ = [("param", C.c_void_p), ("grad", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("count", C.c_uint64),
                ("lr", C.c_float), ("width", C.c_int32), ("layout", C.c_int32), ("active_chunks", C.c_int32)]


class McmcSets(C.Structure):
    _fields_ = [("param", C.c_void_p * 6), ("m", C.c_void_p * 6), ("v", C.c_void_p * 6)]


class SceneSpec(C.Structure):
    _fields_ = [("n", C.c_int32), ("width", C.c_int32), ("height", C.c_int32), ("sh_degree", C.c_int32),
                ("n_cams", C.c_int32), ("seed", C.c_uint64), ("fov_x_deg", C.c_float), ("scale_log_offset", C.c_float)]


# every symbol include/*.h declares (tests/test_abi.py checks this list against the headers)
_PROTOS = {
    "dvs_create": (C.c_void_p, [C.c_int, C.c_size_t, C.c_int, C.c_int]),
    "dvs_create_views": (C.c_void_p, [C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_int]),
    "dvs_destroy": (None, [C.c_void_p]),
    "dvs_raster_forward_views": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(Splats), C.POINTER(Camera), C.c_int, C.POINTER(Opts), C.c_void_p]),
    "dvs_raster_forward_cancel_prepared": (C.c_int, [C.c_void_p]),
    "dvs_raster_forward_views_prepare": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(Splats), C.POINTER(Camera), C.c_int, C.POINTER(Opts), C.c_int64, C.c_int64]),
    "dvs_raster_backward_views": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(Splats), C.POINTER(Camera), C.c_int, C.POINTER(Opts),
                                            C.c_void_p, C.POINTER(SplatGrads)]),
    "dvs_get_view_state": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(FwdState)]),
    "dvs_raster_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(Splats), C.POINTER(Camera), C.POINTER(Opts),
                                     C.c_void_p, C.POINTER(FwdState), C.POINTER(C.c_uint64)]),
    "dvs_raster_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(Splats), C.POINTER(Camera), C.POINTER(Opts),
                                      C.c_void_p, C.POINTER(SplatGrads)]),
    "dvs_raster_backward_composite": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(Opts), C.c_void_p]),
    "dvs_raster_backward_project": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(Splats), C.c_void_p, C.POINTER(Opts),
                                              C.POINTER(SplatGrads)]),
    "dvs_raster_backward_project_chunk": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(Splats), C.c_void_p, C.POINTER(Opts),
                                                    C.POINTER(SplatGrads), C.c_int64, C.c_int64]),
    "dvs_raster_backward_dcolor": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "dvs_sh_grad_combine": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_int, C.c_int]),
    "dvs_shn_relayout": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int]),
    "dvs_sort_pairs_u32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_int, C.c_int]),
    "dvs_export_sorted_keys": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "dvs_get_bwd_intermediates": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int)]),
    "dvs_keep_bwd_intermediates": (C.c_int, [C.c_void_p, C.c_int]),
    "dvs_set_async": (C.c_int, [C.c_void_p, C.c_int]),
    "dvs_get_num_rendered": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64)]),
    "dvs_get_sort_rank_mode": (C.c_int, [C.c_void_p]),
    "dvs_debug_sort_depth_keys": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "dvs_set_export_sorted_tiles": (C.c_int, [C.c_void_p, C.c_int]),
    "dvs_get_arena_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "dvs_set_backward_variant": (C.c_int, [C.c_void_p, C.c_int]),
    "dvs_debug_record_decisions": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint64]),
    "dvs_set_forward_variant": (C.c_int, [C.c_void_p, C.c_int]),
    "dvs_set_live_lists": (C.c_int, [C.c_void_p, C.c_int]),
    "dvs_enable_stage_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "dvs_enable_kernel_probe": (C.c_int, [C.c_void_p, C.c_int]),
    "dvs_read_kernel_probe": (C.c_int, [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_int)]),
    "dvs_get_stage_timing": (C.c_int, [C.c_void_p, C.POINTER(C.POINTER(C.c_char_p)), C.POINTER(C.POINTER(C.c_float))]),
    "dvs_memcpy_d2h": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "dvs_memcpy_h2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "dvs_device_malloc": (C.c_void_p, [C.c_void_p, C.c_size_t]),
    "dvs_device_free": (None, [C.c_void_p, C.c_void_p]),
    "dvs_comm_create": (C.c_void_p, [C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int]),
    "dvs_comm_bootstrap": (C.c_int, [C.c_int, C.c_int, C.c_char_p, C.c_int, C.c_void_p]),
    "dvs_comm_destroy": (None, [C.c_void_p]),
    "dvs_comm_rank": (C.c_int, [C.c_void_p]),
    "dvs_comm_world": (C.c_int, [C.c_void_p]),
    "dvs_comm_backend_ranks": (C.c_int, [C.c_void_p]),
    "dvs_comm_backend_name": (C.c_char_p, [C.c_void_p]),
    "dvs_comm_all_reduce_sum_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "dvs_comm_all_reduce_max_i32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "dvs_comm_reduce_scatter_sum_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "dvs_comm_all_gather_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]),
    "dvs_comm_broadcast": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]),
    "dvs_comm_group_start": (C.c_int, [C.c_void_p]),
    "dvs_comm_group_end": (C.c_int, [C.c_void_p]),
    "dvs_last_error": (C.c_char_p, []),
    "dvs_version": (C.c_char_p, []),
    "dvs_synth_splats": (C.c_int, [C.POINTER(SceneSpec)] + [C.c_void_p] * 6),
    "dvs_synth_camera": (C.c_int, [C.POINTER(SceneSpec), C.c_int, C.POINTER(Camera)]),
    "dvs_synth_target": (C.c_int, [C.POINTER(SceneSpec), C.c_int, C.c_void_p]),
    "dvs_make_camera": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.c_int, C.POINTER(Camera)]),
    "dvs_l1_loss_grad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]),
    "dvs_l1_loss_grad_w": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_float, C.c_void_p, C.c_void_p]),
    "dvs_l2_loss_grad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_float, C.c_void_p, C.c_void_p]),
    "dvs_ssim_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dvs_ssim_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                    C.c_void_p, C.c_int]),
    "dvs_loss_l1_ssim_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float,
                                            C.c_void_p, C.c_void_p]),
    "dvs_densify_accumulate": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dvs_densify_accumulate_rows": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dvs_any_view_radius": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "dvs_densify_plan": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(DensifyParams),
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "dvs_densify_apply": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(DensifyParams), C.c_int, C.c_void_p, C.c_void_p, C.c_int]),
    "dvs_reset_opacity": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p]),
    "dvs_adam_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_float, C.c_float,
                                C.c_float, C.c_float, C.c_int]),
    "dvs_mcmc_scratch_bytes": (C.c_size_t, [C.c_int]),
    "dvs_mcmc_init_scratch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int]),
    "dvs_mcmc_relocate": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(McmcSets), C.c_float, C.c_uint32, C.c_int, C.c_void_p, C.c_int, C.c_void_p]),
    "dvs_mcmc_grow": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(McmcSets), C.c_float, C.c_uint32, C.c_int, C.c_void_p, C.c_int]),
    "dvs_mcmc_add_noise": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_uint32]),
    "dvs_mcmc_regularize": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float]),
    "dvs_mcmc_add_noise_range": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_uint32]),
    "dvs_mcmc_regularize_range": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float]),
    "dvs_adam_step_groups": (C.c_int, [C.c_void_p, C.POINTER(AdamGroup), C.c_int, C.c_float, C.c_float, C.c_float, C.c_int, C.c_void_p,
                                       C.c_int32]),
}
for _name, (_res, _args) in _PROTOS.items():
    _f = getattr(lib, _name)          # AttributeError here = the .so does not export a declared symbol
    _f.restype = _res
    _f.argtypes = _args


def check(status, what="dvs call"):
    if status != 0:
        raise DvsError(f"{what} failed with status {status}: {lib.dvs_last_error().decode()}")
