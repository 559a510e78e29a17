"""divshot_amd — MI355X (gfx950) differentiable 3DGS rasterizer for DIVSHOT's training loop.

The product is the HIP library `lib/libdvsraster.so` (C-ABI in include/dvs_raster.h) and the
`libgstrain.so` plugin built on it. This Python package is a thin ctypes binding used by the
tests and bench.py; torch is only used for device memory, streams and torch.distributed.
There is no CPU fallback: everything here fails loudly when the HIP library is missing.
"""
from ._lib import lib, Camera, Opts, Splats, FwdState, SplatGrads, SceneSpec, DvsError, LIB_PATH  # noqa: F401
from .scene import synth_splats, synth_camera, synth_target, make_spec  # noqa: F401

__all__ = ["lib", "Camera", "Opts", "Splats", "FwdState", "SplatGrads", "SceneSpec", "DvsError",
           "synth_splats", "synth_camera", "synth_target", "make_spec", "LIB_PATH"]
