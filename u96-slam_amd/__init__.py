"""MI355X-native stereo block-matching disparity engine (drop-in for the cv::StereoBM call of
sdoira/U96-SLAM, src/slam/src/core/main.cpp:197-217).

Layout
  csrc/         hand-written HIP kernels (gfx950) + the C-ABI (include/sbm.h)  -> lib/libsbm_hip.so
  stereobm.py   host-side mirror of the cv::StereoBM interface over that C-ABI (ctypes)
  synth.py      deterministic synthetic stereo frames
  shard.py      one-process-per-GPU sharding of pair batches (torch.distributed; RCCL on GPU, gloo on CPU)

There is no CPU fallback in this package: constructing a StereoBM without the built HIP library or without a
GPU raises. The CPU oracle lives in /oracle and is only used by the tests and by bench.py's cpu_baseline leg.
"""
from .stereobm import (StereoBM, StereoBMError, SbmParams, StereoModel, library_path, load_library, PREFILTER_XSOBEL,  # noqa: F401
                       PREFILTER_NORMALIZED_RESPONSE, RectCam, make_rect_cam, PREFILTER_FLAVOUR_CV, PREFILTER_FLAVOUR_RTL, trim,
                       FpgaParams, fpga_params, fpga_params_from_regs, fpga_sad_size_reg, fpga_validate, compute_multi)

__all__ = ["StereoBM", "StereoBMError", "SbmParams", "StereoModel", "library_path", "load_library", "PREFILTER_XSOBEL",
           "PREFILTER_NORMALIZED_RESPONSE", "RectCam", "make_rect_cam", "PREFILTER_FLAVOUR_CV", "PREFILTER_FLAVOUR_RTL", "trim",
           "FpgaParams", "fpga_params", "fpga_params_from_regs", "fpga_sad_size_reg", "fpga_validate", "compute_multi"]
