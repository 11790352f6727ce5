"""Host-side mirror of cv::StereoBM over the C-ABI of lib/libsbm_hip.so (include/sbm.h).

Same names, argument meaning and error behaviour as the interface used at
src/slam/src/core/main.cpp:201-215 of the reference:

    bm = StereoBM.create(16, 9); bm.setPreFilterCap(31); bm.setBlockSize(21); ...; disp = bm.compute(left, right)

`compute` accepts numpy uint8 images (host path, sbm_compute / sbm_compute_batch) or torch CUDA uint8 tensors
(device path, sbm_compute_device; torch is used for device memory only). Parameter errors raise StereoBMError
with the status code and OpenCV's message, where cv::StereoBM::compute would throw cv::Error.
"""
import collections
import ctypes
import os
import pathlib

import numpy as np

PREFILTER_NORMALIZED_RESPONSE = 0
PREFILTER_XSOBEL = 1

_HERE = pathlib.Path(__file__).resolve().parent
_LIB = None


class SbmParams(ctypes.Structure):
    """`sbm_params` of include/sbm.h."""

    _fields_ = [
        ("prefilter_type", ctypes.c_int32), ("prefilter_size", ctypes.c_int32), ("prefilter_cap", ctypes.c_int32),
        ("block_size", ctypes.c_int32), ("min_disparity", ctypes.c_int32), ("num_disparities", ctypes.c_int32),
        ("texture_threshold", ctypes.c_int32), ("uniqueness_ratio", ctypes.c_int32),
        ("speckle_window_size", ctypes.c_int32), ("speckle_range", ctypes.c_int32), ("disp12_max_diff", ctypes.c_int32),
        ("roi1", ctypes.c_int32 * 4), ("roi2", ctypes.c_int32 * 4),
    ]


class StereoModel(ctypes.Structure):
    """`sbm_stereo_model` of include/sbm.h: the StereoCameraModel entries the reference's reprojection reads
    (include/core/StereoCameraModel.h:25-34) plus the optional local transform."""

    _fields_ = [(k, ctypes.c_double) for k in ("fx_l", "fy_l", "cx_l", "cy_l", "Tx_l", "fx_r", "fy_r", "cx_r", "Tx_r")] + [
        ("local", ctypes.c_float * 12), ("has_local", ctypes.c_int32)]


class RectCam(ctypes.Structure):
    """`sbm_rect_cam` of include/sbm.h = struct RECT_PARAM_CH (src/StereoBM/src/fpga.h:250-256), one camera."""

    _fields_ = [("f", ctypes.c_int32 * 2), ("c", ctypes.c_int32 * 2), ("f2inv", ctypes.c_int32 * 2),
                ("c2_f2", ctypes.c_int32 * 2), ("rot", (ctypes.c_int32 * 3) * 3)]


def make_rect_cam(f, c, f2inv, c2_f2, rot):
    cam = RectCam()
    cam.f[:] = [int(v) for v in f]
    cam.c[:] = [int(v) for v in c]
    cam.f2inv[:] = [int(v) for v in f2inv]
    cam.c2_f2[:] = [int(v) for v in c2_f2]
    for r in range(3):
        for k in range(3):
            cam.rot[r][k] = int(rot[r][k])
    return cam


PREFILTER_FLAVOUR_CV = 0
PREFILTER_FLAVOUR_RTL = 1


class FpgaParams(ctypes.Structure):
    """`sbm_fpga_params` of include/sbm.h: the fields of the BM register block (struct FPGA_REG_BM,
    src/StereoBM/src/fpga.h:154-169) as decoded by src/dvp/rtl/bm.v:172-193."""

    _fields_ = [(k, ctypes.c_int32) for k in ("width", "height", "block_size", "num_disparities", "uni_enable", "uni_mode",
                                              "uni_threshold")]


def fpga_params(width, height, block_size=21, num_disparities=64, uni_enable=0, uni_mode=0, uni_threshold=0):
    return FpgaParams(width, height, block_size, num_disparities, uni_enable, uni_mode, uni_threshold)


def fpga_params_from_regs(image_size, bm_setting, uni_filt_ctrl=0):
    """ImageSize [1708h], BmSetting [170Ch], UniFiltCtrl [1728h] -> FpgaParams (firmware: fpga.c:155,158)."""
    q = FpgaParams()
    _check(load_library().sbm_fpga_params_from_regs(image_size, bm_setting, uni_filt_ctrl, ctypes.byref(q)))
    return q


def fpga_sad_size_reg(params):
    """Read-back value of SAD_Size [1724h] (bm.v:208)."""
    return int(load_library().sbm_fpga_sad_size_reg(ctypes.byref(params)))


def fpga_validate(params):
    return int(load_library().sbm_fpga_params_validate(ctypes.byref(params)))


class StereoBMError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"sbm status {code}: {message}")
        self.code = code


def library_path():
    """lib/libsbm_hip.so, or -- SBM_LIB_AB=<file name> -- another build of the same HIP engine inside lib/ for kernel A/B
    runs (tools/exp). Only a bare libsbm_hip*.so name is accepted and the file must exist: never a fallback, never a path."""
    name = os.environ.get("SBM_LIB_AB", "libsbm_hip.so")
    if name != os.path.basename(name) or not (name.startswith("libsbm_hip") and name.endswith(".so")):
        raise ImportError(f"SBM_LIB_AB={name!r}: expected the bare name of a libsbm_hip*.so inside {_HERE / 'lib'}")
    return _HERE / "lib" / name


def loaded_library_name():
    """File name of the engine library this process uses (bench.py prints it)."""
    return library_path().name


def load_library():
    """Load lib/libsbm_hip.so. Fails loudly when it has not been built (no fallback of any kind)."""
    global _LIB
    if _LIB is not None:
        return _LIB
    # torch ships its own HIP runtime; when both live in one process it must be the first one loaded so that
    # libsbm_hip.so binds to the same runtime (device memory and streams are shared with torch).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    path = library_path()
    if not path.exists():
        raise ImportError(f"{path} is missing: build it with `make` (or __graft_entry__.build()); "
                          "this package has no CPU fallback")
    L = ctypes.CDLL(str(path))
    vp, ci, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
    pp = ctypes.POINTER(SbmParams)
    L.sbm_params_default.argtypes = [pp, ci, ci]
    L.sbm_params_default.restype = None
    L.sbm_params_validate.argtypes = [pp, ci, ci]
    L.sbm_create.argtypes = [ctypes.POINTER(vp), pp, ci]
    L.sbm_destroy.argtypes = [vp]
    L.sbm_destroy.restype = None
    L.sbm_set_params.argtypes = [vp, pp]
    L.sbm_get_params.argtypes = [vp, pp]
    L.sbm_compute.argtypes = [vp, vp, sz, vp, sz, ci, ci, vp, sz]
    L.sbm_compute_batch.argtypes = [vp, ci, ctypes.POINTER(vp), sz, ctypes.POINTER(vp), sz, ci, ci, ctypes.POINTER(vp), sz]
    L.sbm_compute_device.argtypes = [vp, ci, vp, vp, ci, ci, vp, ci]
    L.sbm_synchronize.argtypes = [vp]
    L.sbm_submit_dense.argtypes = [vp, ci, vp, vp, ci, ci, vp]
    L.sbm_wait_oldest.argtypes = [vp]
    L.sbm_compute_batch_multi.argtypes = [ctypes.POINTER(vp), ci, ci, vp, vp, ci, ci, vp]
    L.sbm_debug_fetch.argtypes = [vp, ci, vp, sz]
    L.sbm_set_profiling.argtypes = [vp, ci]
    L.sbm_get_profile.argtypes = [vp, ctypes.c_char_p, ctypes.POINTER(ctypes.c_float)]
    L.sbm_last_kernel_name.argtypes = [vp, ctypes.c_char_p, sz]
    mp = ctypes.POINTER(StereoModel)
    L.sbm_disparity_to_float_device.argtypes = [vp, ci, vp, ci, ci, vp, ci]
    L.sbm_decimate_device.argtypes = [vp, ci, vp, ci, ci, ci, vp, ci]
    L.sbm_reproject_device.argtypes = [vp, ci, vp, ci, ci, ci, mp, ci, vp, ci]
    L.sbm_keypoints3d_device.argtypes = [vp, vp, ci, ci, vp, ci, mp, ctypes.c_float, ctypes.c_float, vp, ci]
    L.sbm_rect_map_device.argtypes = [vp, ctypes.POINTER(RectCam), ci, ci, vp, ci]
    L.sbm_rect_remap_device.argtypes = [vp, ci, vp, vp, ci, ci, vp, ci]
    L.sbm_prefilter_device.argtypes = [vp, ci, vp, ci, ci, ci, ci, vp, ci]
    fp = ctypes.POINTER(FpgaParams)
    u32 = ctypes.c_uint32
    L.sbm_fpga_params_from_regs.argtypes = [u32, u32, u32, fp]
    L.sbm_fpga_sad_size_reg.argtypes = [fp]
    L.sbm_fpga_sad_size_reg.restype = u32
    L.sbm_fpga_params_validate.argtypes = [fp]
    L.sbm_fpga_bm_device.argtypes = [vp, ci, vp, vp, fp, vp, ci]
    L.sbm_fpga_compute_device.argtypes = [vp, ci, vp, vp, fp, vp, ci]
    L.sbm_gftt_eig_device.argtypes = [vp, ci, vp, ci, ci, vp, vp, ci]
    L.sbm_fpga_compute.argtypes = [vp, vp, sz, vp, sz, fp, vp, sz]
    L.sbm_gftt_eig.argtypes = [vp, vp, sz, ci, ci, vp, sz, ctypes.POINTER(u32)]
    L.sbm_stream.argtypes = [vp]
    L.sbm_stream.restype = vp
    L.sbm_strerror.argtypes = [ci]
    L.sbm_strerror.restype = ctypes.c_char_p
    L.sbm_last_hip_error.argtypes = [vp]
    L.sbm_version.restype = ci
    _LIB = L
    return L


def _check(code, handle=None):
    if code != 0:
        L = load_library()
        msg = L.sbm_strerror(code).decode()
        if code == -21 and handle:
            msg += f" (hipError {L.sbm_last_hip_error(handle)})"
        raise StereoBMError(code, msg)


class StereoBM:
    """cv::StereoBM look-alike. One instance owns one device handle (stream + scratch); not thread-safe."""

    def __init__(self, numDisparities=0, blockSize=21, device=0):
        L = load_library()
        self._L = L
        self._p = SbmParams()
        L.sbm_params_default(ctypes.byref(self._p), numDisparities, blockSize)
        self._h = ctypes.c_void_p()
        self._device = device
        self._inflight = []                   # buffers of asynchronous compute_device calls, until synchronize()
        self._host_inflight = collections.deque()   # (left, right, disparity) of submit_host, oldest first
        _check(L.sbm_create(ctypes.byref(self._h), ctypes.byref(self._p), device))

    @staticmethod
    def create(numDisparities=0, blockSize=21, device=0):
        return StereoBM(numDisparities, blockSize, device)

    def close(self):
        """Release the engine. Outstanding submit_host() submissions are drained first (sbm_synchronize) while their arrays are
        still referenced here, so every submitted `disparity` array is filled -- sbm_destroy on its own would let the queued
        copies finish and DROP the maps of the newest submission (include/sbm.h, "sbm_destroy() and the asynchronous feed")."""
        h = getattr(self, "_h", None)
        if h:
            if getattr(self, "_host_inflight", None):
                self._L.sbm_synchronize(h)
                self._host_inflight.clear()
            self._L.sbm_destroy(h)
            self._h = None

    def __del__(self):
        self.close()

    # ---- the cv::StereoBM / cv::StereoMatcher setters and getters -------------------------------------------
    def _set(self, name, v):
        setattr(self._p, name, int(v))
        _check(self._L.sbm_set_params(self._h, ctypes.byref(self._p)), self._h)

    def setPreFilterType(self, v): self._set("prefilter_type", v)
    def setPreFilterSize(self, v): self._set("prefilter_size", v)
    def setPreFilterCap(self, v): self._set("prefilter_cap", v)
    def setBlockSize(self, v): self._set("block_size", v)
    def setMinDisparity(self, v): self._set("min_disparity", v)
    def setNumDisparities(self, v): self._set("num_disparities", v)
    def setTextureThreshold(self, v): self._set("texture_threshold", v)
    def setUniquenessRatio(self, v): self._set("uniqueness_ratio", v)
    def setSpeckleWindowSize(self, v): self._set("speckle_window_size", v)
    def setSpeckleRange(self, v): self._set("speckle_range", v)
    def setDisp12MaxDiff(self, v): self._set("disp12_max_diff", v)

    def setROI1(self, rect):
        self._p.roi1[:] = [int(v) for v in rect]
        _check(self._L.sbm_set_params(self._h, ctypes.byref(self._p)), self._h)

    def setROI2(self, rect):
        self._p.roi2[:] = [int(v) for v in rect]
        _check(self._L.sbm_set_params(self._h, ctypes.byref(self._p)), self._h)

    def getPreFilterType(self): return self._p.prefilter_type
    def getPreFilterSize(self): return self._p.prefilter_size
    def getPreFilterCap(self): return self._p.prefilter_cap
    def getBlockSize(self): return self._p.block_size
    def getMinDisparity(self): return self._p.min_disparity
    def getNumDisparities(self): return self._p.num_disparities
    def getTextureThreshold(self): return self._p.texture_threshold
    def getUniquenessRatio(self): return self._p.uniqueness_ratio
    def getSpeckleWindowSize(self): return self._p.speckle_window_size
    def getSpeckleRange(self): return self._p.speckle_range
    def getDisp12MaxDiff(self): return self._p.disp12_max_diff
    def getROI1(self): return tuple(self._p.roi1)
    def getROI2(self): return tuple(self._p.roi2)

    def params(self):
        q = SbmParams()
        ctypes.memmove(ctypes.byref(q), ctypes.byref(self._p), ctypes.sizeof(SbmParams))
        return q

    # ---- compute ----------------------------------------------------------------------------------------------
    def compute(self, left, right, disparity=None):
        """cv::StereoBM::compute. numpy (H,W) or (n,H,W) uint8 -> numpy int16; torch CUDA uint8 -> torch CUDA int16."""
        if isinstance(left, np.ndarray):
            return self._compute_host(left, right, disparity)
        return self.compute_device(left, right, disparity)

    def _compute_host(self, left, right, disparity):
        if left.shape != right.shape:
            raise StereoBMError(-2, "All the images must have the same size")
        if left.dtype != np.uint8 or right.dtype != np.uint8:
            raise StereoBMError(-2, "Both input images must have CV_8UC1")
        single = left.ndim == 2
        L3 = left[None] if single else left
        R3 = right[None] if single else right
        if L3.ndim != 3:
            raise StereoBMError(-2, "expected (H,W) or (n,H,W) images")
        n, h, w = L3.shape
        # honour arbitrary row strides like cv::Mat::step, but rows themselves must be dense
        def rows(a):
            if a.strides[-1] != 1:
                a = np.ascontiguousarray(a)
            return a
        L3, R3 = rows(L3), rows(R3)
        out = disparity if disparity is not None else np.empty(L3.shape, np.int16)
        if not isinstance(out, np.ndarray) or out.dtype != np.int16:
            raise StereoBMError(-2, "disparity must be a numpy int16 array")
        out3 = out[None] if out.ndim == 2 else out
        if out3.shape != L3.shape or not out.flags.writeable:
            raise StereoBMError(-2, f"disparity has shape {out.shape}, the images {left.shape}")
        # strides travel to C as size_t: rows must be dense and every row stride positive and at least one row long
        for a, item in ((L3, 1), (R3, 1), (out3, 2)):
            if a.strides[-1] != item or a.strides[-2] < a.shape[-1] * item or (a.ndim == 3 and a.shape[0] > 1 and a.strides[0] <= 0):
                raise StereoBMError(-2, "rows must be dense with a positive row stride (negative or overlapping strides are not supported)")
        vp = ctypes.c_void_p
        lp = (vp * n)(*[L3[i].ctypes.data for i in range(n)])
        rp = (vp * n)(*[R3[i].ctypes.data for i in range(n)])
        dp = (vp * n)(*[out3[i].ctypes.data for i in range(n)])
        _check(self._L.sbm_compute_batch(self._h, n, lp, L3.strides[-2], rp, R3.strides[-2], w, h, dp, out3.strides[-2]),
               self._h)
        return out[0] if (single and disparity is None) else out

    def submit_host(self, left, right, disparity):
        """sbm_submit_dense: queue one dense (n,H,W) uint8 batch in (pinned) host memory; `disparity` (n,H,W) int16 is filled
        when the matching wait_host() returns. At most three submissions are in flight. Dropping the engine with submissions
        outstanding drains them first (close()); the bare C call sbm_destroy() would drop the newest submission's maps."""
        for a, dt in ((left, np.uint8), (right, np.uint8), (disparity, np.int16)):
            if not isinstance(a, np.ndarray) or a.dtype != dt or a.ndim != 3 or not a.flags.c_contiguous:
                raise StereoBMError(-2, "submit_host takes C-contiguous (n,H,W) arrays: uint8 images, int16 disparity")
        if left.shape != right.shape or left.shape != disparity.shape:
            raise StereoBMError(-2, "All the images must have the same size")
        n, h, w = left.shape
        # The engine drains to at most two outstanding submissions INSIDE this call, before it queues the new one: only once
        # it has returned are the oldest submissions' copies known to be complete, so their arrays are released afterwards.
        _check(self._L.sbm_submit_dense(self._h, n, left.ctypes.data, right.ctypes.data, w, h, disparity.ctypes.data), self._h)
        while len(self._host_inflight) > 2:
            self._host_inflight.popleft()
        self._host_inflight.append((left, right, disparity))

    def wait_host(self):
        """sbm_wait_oldest: block until the oldest outstanding submit_host() has delivered its maps."""
        _check(self._L.sbm_wait_oldest(self._h), self._h)
        if self._host_inflight:
            self._host_inflight.popleft()     # its arrays are the caller's again

    def compute_device(self, left, right, disparity=None, sync=True):
        """Device-resident batch: torch CUDA uint8 tensors (n,H,W) or (H,W), contiguous. Returns a torch int16 tensor."""
        import torch

        if left.shape != right.shape:
            raise StereoBMError(-2, "All the images must have the same size")
        if left.dtype != torch.uint8 or right.dtype != torch.uint8 or not left.is_cuda or not right.is_cuda:
            raise StereoBMError(-2, "Both input images must be CUDA uint8 tensors")
        if left.device.index != self._device:
            raise StereoBMError(-20, f"tensor on cuda:{left.device.index}, engine on device {self._device}")
        if left.dim() not in (2, 3):
            raise StereoBMError(-2, "expected (H,W) or (n,H,W) images")
        left, right = left.contiguous(), right.contiguous()
        shape = left.shape
        n = 1 if left.dim() == 2 else shape[0]
        h, w = shape[-2], shape[-1]
        if disparity is None:
            disparity = torch.empty(shape, dtype=torch.int16, device=left.device)
        elif (not isinstance(disparity, torch.Tensor) or disparity.dtype != torch.int16 or not disparity.is_cuda
              or disparity.device != left.device or tuple(disparity.shape) != tuple(shape) or not disparity.is_contiguous()):
            # the C-ABI writes n*h*w int16 through the raw pointer: anything else would be an out-of-bounds / strided-wrong write
            raise StereoBMError(-2, f"disparity must be a contiguous CUDA int16 tensor of shape {tuple(shape)} on {left.device}")
        # the engine runs on its own (non-blocking) stream: order it behind whatever produced the inputs
        torch.cuda.current_stream(left.device).synchronize()
        _check(self._L.sbm_compute_device(self._h, n, left.data_ptr(), right.data_ptr(), w, h, disparity.data_ptr(),
                                          1 if sync else 0), self._h)
        if not sync:
            # torch's caching allocator only knows its own streams: without this the .contiguous() temporaries and a
            # freshly allocated output could be handed out again while the engine's kernels still use them
            # (a list: back-to-back asynchronous calls each keep their buffers until the next synchronize())
            self._inflight.append((left, right, disparity))
        else:
            # a synchronous call drains the engine's compute stream: earlier asynchronous DEVICE calls are done too
            # (host submissions keep their arrays: their maps may still be on the way home on the copy stream)
            self._inflight.clear()
        return disparity

    def _check_device_images(self, *tensors):
        """Every image handed to the engine as a raw pointer: CUDA uint8, on the handle's device, (H,W) or (n,H,W)."""
        import torch

        for t in tensors:
            if not isinstance(t, torch.Tensor) or t.dtype != torch.uint8 or not t.is_cuda:
                raise StereoBMError(-2, "images must be CUDA uint8 tensors")
            if t.device.index != self._device:
                raise StereoBMError(-20, f"tensor on cuda:{t.device.index}, engine on device {self._device}")
            if t.dim() not in (2, 3):
                raise StereoBMError(-2, "expected (H,W) or (n,H,W) images")

    def launch_raw(self, n, d_left, d_right, w, h, d_disp, sync=False):
        """Thin call of sbm_compute_device on raw device addresses (used by bench.py's timed loop)."""
        _check(self._L.sbm_compute_device(self._h, n, d_left, d_right, w, h, d_disp, 1 if sync else 0), self._h)

    # ---- consumers of the map (SensorData.cpp:50-58, Stereo.cpp:53-117,157-199, main.cpp:522-553) ----------------
    def to_float(self, disp):
        """CV_32F form of a torch CUDA int16 disparity tensor: disp / 16 as float32 (cv convertTo(CV_32F, 1/16))."""
        import torch

        disp = disp.contiguous()
        h, w = disp.shape[-2], disp.shape[-1]
        n = 1 if disp.dim() == 2 else disp.shape[0]
        out = torch.empty(disp.shape, dtype=torch.float32, device=disp.device)
        torch.cuda.current_stream(disp.device).synchronize()
        _check(self._L.sbm_disparity_to_float_device(self._h, n, disp.data_ptr(), w, h, out.data_ptr(), 1), self._h)
        return out

    def decimate(self, disp, scale=4):
        """torch CUDA int16 (n,H,W) or (H,W) -> every scale-th pixel, on the device."""
        import torch

        d3 = disp if disp.dim() == 3 else disp[None]
        d3 = d3.contiguous()
        n, h, w = d3.shape
        out = torch.empty((n, h // scale, w // scale), dtype=torch.int16, device=d3.device)
        torch.cuda.current_stream(d3.device).synchronize()
        _check(self._L.sbm_decimate_device(self._h, n, d3.data_ptr(), w, h, scale, out.data_ptr(), 1), self._h)
        return out if disp.dim() == 3 else out[0]

    def reproject(self, disp, model, scale=1, apply_local=True):
        """torch CUDA int16 map(s) -> float32 (..., H, W, 3) points, NaN where invalid."""
        import torch

        d3 = disp if disp.dim() == 3 else disp[None]
        d3 = d3.contiguous()
        n, h, w = d3.shape
        xyz = torch.empty((n, h, w, 3), dtype=torch.float32, device=d3.device)
        torch.cuda.current_stream(d3.device).synchronize()
        _check(self._L.sbm_reproject_device(self._h, n, d3.data_ptr(), w, h, scale, ctypes.byref(model),
                                            1 if apply_local else 0, xyz.data_ptr(), 1), self._h)
        return xyz if disp.dim() == 3 else xyz[0]

    def keypoints3d(self, disp, kpts, model, min_depth=0.0, max_depth=0.0):
        """One full-resolution torch CUDA int16 map + float32 (nk,2) keypoints (x,y) -> float32 (nk,3)."""
        import torch

        disp = disp.contiguous()
        kpts = kpts.contiguous()
        h, w = disp.shape
        xyz = torch.empty((kpts.shape[0], 3), dtype=torch.float32, device=disp.device)
        torch.cuda.current_stream(disp.device).synchronize()
        _check(self._L.sbm_keypoints3d_device(self._h, disp.data_ptr(), w, h, kpts.data_ptr(), kpts.shape[0],
                                              ctypes.byref(model), min_depth, max_depth, xyz.data_ptr(), 1), self._h)
        return xyz

    # ---- producers in front of the path (fpga.c:303-366, rect_intp.v:285-404, xsbl2.v:661-874) --------------------
    def rect_map(self, cam, width, height):
        """Inverse rectification map of one camera: torch CUDA int16 (H, W, 2), (x, y) in 1/32 source pixels."""
        import torch

        m = torch.empty((height, width, 2), dtype=torch.int16, device=f"cuda:{self._device}")
        _check(self._L.sbm_rect_map_device(self._h, ctypes.byref(cam), width, height, m.data_ptr(), 1), self._h)
        return m

    def rect_remap(self, src, rmap):
        """torch CUDA uint8 (n,H,W) or (H,W) raw frames + a map from rect_map -> rectified frames, on the device."""
        import torch

        src, rmap = src.contiguous(), rmap.contiguous()
        h, w = src.shape[-2], src.shape[-1]
        if tuple(rmap.shape) != (h, w, 2) or rmap.dtype != torch.int16 or src.dtype != torch.uint8:
            raise StereoBMError(-2, "map must be int16 (H,W,2) and frames uint8 (..,H,W)")
        n = 1 if src.dim() == 2 else src.shape[0]
        out = torch.empty_like(src)
        torch.cuda.current_stream(src.device).synchronize()
        _check(self._L.sbm_rect_remap_device(self._h, n, src.data_ptr(), rmap.data_ptr(), w, h, out.data_ptr(), 1), self._h)
        return out

    def prefilter(self, src, flavour=PREFILTER_FLAVOUR_CV, cap=None):
        """Stand-alone x-Sobel prefilter of torch CUDA uint8 (n,H,W) or (H,W) frames, cv or RTL flavour."""
        import torch

        self._check_device_images(src)
        src = src.contiguous()
        h, w = src.shape[-2], src.shape[-1]
        n = 1 if src.dim() == 2 else src.shape[0]
        out = torch.empty_like(src)
        torch.cuda.current_stream(src.device).synchronize()
        _check(self._L.sbm_prefilter_device(self._h, n, src.data_ptr(), w, h, flavour,
                                            self._p.prefilter_cap if cap is None else cap, out.data_ptr(), 1), self._h)
        return out

    # ---- the reference's own matcher: FPGA flavour (src/dvp/rtl/bm*.v; FPGA.cpp:270-279 consumers) ---------------------
    def _fpga(self, fn, a, b, params):
        import torch

        self._check_device_images(a, b)
        if a.shape != b.shape:
            raise StereoBMError(-2, "both inputs must be CUDA uint8 tensors of the same shape")
        a, b = a.contiguous(), b.contiguous()
        h, w = a.shape[-2], a.shape[-1]
        if (w, h) != (params.width, params.height):
            raise StereoBMError(-2, f"images are {w}x{h}, ImageSize says {params.width}x{params.height}")
        n = 1 if a.dim() == 2 else a.shape[0]
        out = torch.empty(a.shape, dtype=torch.int16, device=a.device)
        torch.cuda.current_stream(a.device).synchronize()
        _check(fn(self._h, n, a.data_ptr(), b.data_ptr(), ctypes.byref(params), out.data_ptr(), 1), self._h)
        return out

    def fpga_bm(self, xsbl_l, xsbl_r, params):
        """RTL block matcher on x-Sobel planes (torch CUDA uint8, (n,H,W) or (H,W)) -> int16 s11.4, -1 = none."""
        return self._fpga(self._L.sbm_fpga_bm_device, xsbl_l, xsbl_r, params)

    def fpga_compute(self, left, right, params):
        """xsbl2.v prefilter + RTL block matcher on rectified frames: the PL pipeline behind Fpga::receiveDepthMap."""
        return self._fpga(self._L.sbm_fpga_compute_device, left, right, params)

    def fpga_compute_host(self, left, right, params):
        """numpy uint8 (H,W) rectified pair -> numpy int16 (H,W): the frame Fpga::receiveDepthMap would hand out."""
        if left.shape != right.shape or left.dtype != np.uint8 or right.dtype != np.uint8 or left.ndim != 2:
            raise StereoBMError(-2, "both inputs must be (H,W) uint8 arrays of the same shape")
        if left.strides[1] != 1 or right.strides[1] != 1 or left.strides[0] < left.shape[1] or right.strides[0] < right.shape[1]:
            raise StereoBMError(-2, "rows must be dense with a positive row stride")
        out = np.empty(left.shape, np.int16)
        _check(self._L.sbm_fpga_compute(self._h, left.ctypes.data, left.strides[0], right.ctypes.data, right.strides[0],
                                        ctypes.byref(params), out.ctypes.data, out.strides[0]), self._h)
        return out

    def gftt_eig_host(self, img):
        """numpy uint8 (H,W) -> (numpy uint16 map, Max register value), as FPGA.cpp:283-291 assembles them."""
        if img.dtype != np.uint8 or img.ndim != 2 or img.strides[1] != 1 or img.strides[0] < img.shape[1]:
            raise StereoBMError(-2, "image must be an (H,W) uint8 array with dense rows")
        out = np.empty(img.shape, np.uint16)
        mx = ctypes.c_uint32()
        _check(self._L.sbm_gftt_eig(self._h, img.ctypes.data, img.strides[0], img.shape[1], img.shape[0], out.ctypes.data,
                                    out.strides[0], ctypes.byref(mx)), self._h)
        return out, int(mx.value)

    def gftt_eig(self, img):
        """PL GFTT min-eigenvalue map of torch CUDA uint8 frames (n,H,W) or (H,W): (int16-viewed uint16 map as torch.int32,
        per-image maximum) -- the inputs of generateKeypoints2 (src/slam/src/core/GFTT.cpp:41)."""
        import torch

        self._check_device_images(img)
        img = img.contiguous()
        h, w = img.shape[-2], img.shape[-1]
        n = 1 if img.dim() == 2 else img.shape[0]
        eig = torch.empty(img.shape, dtype=torch.int16, device=img.device)     # uint16 payload (torch has no uint16 math)
        mx = torch.empty((n,), dtype=torch.int32, device=img.device)
        torch.cuda.current_stream(img.device).synchronize()
        _check(self._L.sbm_gftt_eig_device(self._h, n, img.data_ptr(), w, h, eig.data_ptr(), mx.data_ptr(), 1), self._h)
        return eig.to(torch.int32) & 0xffff, mx

    def synchronize(self):
        _check(self._L.sbm_synchronize(self._h), self._h)
        self._inflight.clear()   # buffers of asynchronous compute_device calls may be released now
        self._host_inflight.clear()

    def stream(self):
        return self._L.sbm_stream(self._h)

    def set_profiling(self, on):
        # 0 = off, 1 = sync after every call, 2 = stage events only (no host sync; up to 64 calls per profile() read),
        # 3 = as 2 on every 4th call only
        _check(self._L.sbm_set_profiling(self._h, int(on)), self._h)

    def last_kernel(self):
        """Template instantiation of the SAD kernel the last compute call launched (sbm_last_kernel_name)."""
        buf = ctypes.create_string_buffer(128)
        _check(self._L.sbm_last_kernel_name(self._h, buf, 128), self._h)
        return buf.value.decode()

    def profile(self):
        out = {}
        for k in ("prefilter", "sad", "lrcheck", "speckle", "total"):
            v = ctypes.c_float()
            _check(self._L.sbm_get_profile(self._h, k.encode(), ctypes.byref(v)), self._h)
            out[k] = v.value
        return out

    def debug_fetch(self, which, n, h, w):
        dt = {0: np.uint8, 1: np.uint8, 2: np.int32, 3: np.int16}[which]
        a = np.empty((n, h, w), dt)
        _check(self._L.sbm_debug_fetch(self._h, which, a.ctypes.data, a.nbytes), self._h)
        return a


def compute_multi(engines, left, right, disparity):
    """sbm_compute_batch_multi: one dense (n,H,W) uint8 host batch over several StereoBM engines (normally one per GPU),
    contiguous pair blocks, maps delivered into `disparity` (n,H,W) int16 in place. Host memory should be pinned."""
    for a, dt in ((left, np.uint8), (right, np.uint8), (disparity, np.int16)):
        if not isinstance(a, np.ndarray) or a.dtype != dt or a.ndim != 3 or not a.flags.c_contiguous:
            raise StereoBMError(-2, "compute_multi takes C-contiguous (n,H,W) arrays: uint8 images, int16 disparity")
    if left.shape != right.shape or left.shape != disparity.shape:
        raise StereoBMError(-2, "All the images must have the same size")
    if not engines:
        raise StereoBMError(-24, "compute_multi needs at least one engine")
    n, h, w = left.shape
    L = load_library()
    hs = (ctypes.c_void_p * len(engines))(*[e._h for e in engines])
    _check(L.sbm_compute_batch_multi(hs, len(engines), n, left.ctypes.data, right.ctypes.data, w, h, disparity.ctypes.data), engines[0]._h)
    return disparity


def trim():
    """Free the handles parked by destroyed matchers (see sbm_trim in include/sbm.h)."""
    load_library().sbm_trim()


def validate(params, width, height):
    """Status code of cv::StereoBM::compute's parameter checks (0 = ok)."""
    return load_library().sbm_params_validate(ctypes.byref(params), width, height)
