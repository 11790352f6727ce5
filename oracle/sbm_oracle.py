"""ctypes binding of the CPU oracle (oracle/libsbm_oracle.so). TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product
package (u96-slam_amd/) never does. See oracle/sbm_oracle.h for what is restated and its pinning status
(prefilter pinned by the reference's data/ref_xsbl_* vectors; block-matching output: parity unpinned).
"""
import ctypes
import os
import pathlib
import subprocess

import numpy as np

_HERE = pathlib.Path(__file__).resolve().parent
_LIB = None


class SbmParams(ctypes.Structure):
    """Mirror of `sbm_params` (include/sbm.h)."""

    _fields_ = [
        ("prefilter_type", ctypes.c_int32),
        ("prefilter_size", ctypes.c_int32),
        ("prefilter_cap", ctypes.c_int32),
        ("block_size", ctypes.c_int32),
        ("min_disparity", ctypes.c_int32),
        ("num_disparities", ctypes.c_int32),
        ("texture_threshold", ctypes.c_int32),
        ("uniqueness_ratio", ctypes.c_int32),
        ("speckle_window_size", ctypes.c_int32),
        ("speckle_range", ctypes.c_int32),
        ("disp12_max_diff", ctypes.c_int32),
        ("roi1", ctypes.c_int32 * 4),
        ("roi2", ctypes.c_int32 * 4),
    ]


class StereoModel(ctypes.Structure):
    """Mirror of `sbm_stereo_model` (include/sbm.h)."""

    _fields_ = [(k, ctypes.c_double) for k in ("fx_l", "fy_l", "cx_l", "cy_l", "Tx_l", "fx_r", "fy_r", "cx_r", "Tx_r")] + [
        ("local", ctypes.c_float * 12), ("has_local", ctypes.c_int32)]


class RectCam(ctypes.Structure):
    """Mirror of `sbm_rect_cam` (include/sbm.h)."""

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


def make_model(fx=718.856, fy=718.856, cx=607.1928, cy=185.2157, baseline=0.537, cx_r=None, local=None):
    """KITTI-like rectified pair: P0 = [f 0 cx 0; ...], P1 = [f 0 cx -f*b; ...]."""
    m = StereoModel()
    m.fx_l, m.fy_l, m.cx_l, m.cy_l, m.Tx_l = fx, fy, cx, cy, 0.0
    m.fx_r, m.fy_r, m.cx_r, m.Tx_r = fx, fy, cx if cx_r is None else cx_r, -fx * baseline
    if local is not None:
        m.local[:] = [float(v) for v in local]
        m.has_local = 1
    return m


def make_params(num_disparities=64, block_size=21, prefilter_cap=31, min_disparity=0, texture_threshold=10,
                uniqueness_ratio=15, speckle_window_size=0, speckle_range=0, disp12_max_diff=-1,
                prefilter_type=1, prefilter_size=9, roi1=(0, 0, 0, 0), roi2=(0, 0, 0, 0)):
    p = SbmParams()
    p.prefilter_type, p.prefilter_size, p.prefilter_cap = prefilter_type, prefilter_size, prefilter_cap
    p.block_size, p.min_disparity, p.num_disparities = block_size, min_disparity, num_disparities
    p.texture_threshold, p.uniqueness_ratio = texture_threshold, uniqueness_ratio
    p.speckle_window_size, p.speckle_range, p.disp12_max_diff = speckle_window_size, speckle_range, disp12_max_diff
    p.roi1[:] = list(roi1)
    p.roi2[:] = list(roi2)
    return p


def build(force=False):
    so = _HERE / "libsbm_oracle.so"
    srcs = [_HERE / "sbm_oracle.c", _HERE / "sbm_oracle_fpga.c", _HERE / "sbm_oracle_simd.c", _HERE / "sbm_oracle.h"]
    if force or not so.exists() or so.stat().st_mtime < max(f.stat().st_mtime for f in srcs):
        subprocess.run(["make", "-C", str(_HERE), "libsbm_oracle.so"], check=True, capture_output=True)
    return so


def lib():
    global _LIB
    if _LIB is None:
        override = os.environ.get("SBM_ORACLE_LIB")   # e.g. the sanitizer build, see oracle/Makefile
        so = pathlib.Path(override).resolve() if override else build()
        try:
            L = ctypes.CDLL(str(so))
        except OSError:
            so = build(force=True)
            L = ctypes.CDLL(str(so))
        u8p, i16p, i32p = (ctypes.POINTER(ctypes.c_uint8), ctypes.POINTER(ctypes.c_int16), ctypes.POINTER(ctypes.c_int32))
        sz, ci = ctypes.c_size_t, ctypes.c_int
        pp = ctypes.POINTER(SbmParams)
        L.sbmo_prefilter_xsobel.argtypes = [u8p, sz, u8p, sz, ci, ci, ci]
        L.sbmo_prefilter_xsobel.restype = None
        L.sbmo_prefilter_xsobel_fpga.argtypes = [u8p, sz, u8p, sz, ci, ci, ctypes.c_uint8]
        L.sbmo_prefilter_xsobel_fpga.restype = None
        L.sbmo_valid_roi.argtypes = [i32p, i32p, ci, ci, ci, i32p]
        L.sbmo_valid_roi.restype = None
        for name in ("sbmo_find_correspondence", "sbmo_find_correspondence_bruteforce"):
            f = getattr(L, name)
            f.argtypes = [u8p, u8p, sz, ci, ci, ci, ci, pp, i16p, sz, i32p, sz]
            f.restype = None
        L.sbmo_validate_disparity.argtypes = [i16p, sz, i32p, sz, ci, ci, ci, ci, ci]
        L.sbmo_validate_disparity.restype = None
        L.sbmo_filter_speckles.argtypes = [i16p, sz, ci, ci, ci, ci, ci]
        L.sbmo_filter_speckles.restype = None
        L.sbmo_compute.argtypes = [pp, u8p, sz, u8p, sz, ci, ci, i16p, sz, u8p, u8p, i16p, i32p]
        L.sbmo_compute.restype = ci
        L.sbmo_compute_batch.argtypes = [pp, ci, u8p, u8p, ci, ci, i16p, ci]
        L.sbmo_compute_batch.restype = ci
        L.sbmo_max_threads.restype = ci
        f32p = ctypes.POINTER(ctypes.c_float)
        mp = ctypes.POINTER(StereoModel)
        L.sbmo_decimate.argtypes = [i16p, ci, ci, ci, i16p]
        L.sbmo_decimate.restype = None
        L.sbmo_reproject.argtypes = [i16p, ci, ci, ci, mp, ci, f32p]
        L.sbmo_reproject.restype = None
        L.sbmo_keypoints3d.argtypes = [i16p, ci, ci, f32p, ci, mp, ctypes.c_float, ctypes.c_float, f32p]
        L.sbmo_keypoints3d.restype = None
        L.sbmo_rect_map.argtypes = [ctypes.POINTER(RectCam), ci, ci, i16p]
        L.sbmo_rect_map.restype = None
        L.sbmo_rect_remap.argtypes = [u8p, i16p, ci, ci, u8p]
        L.sbmo_rect_remap.restype = None
        u16p, u32 = ctypes.POINTER(ctypes.c_uint16), ctypes.c_uint32
        L.sbmo_rtl_diven.argtypes = [ci, ci, ci, ci, u32, u32]
        L.sbmo_rtl_diven.restype = u32
        L.sbmo_rtl_det.argtypes = [u16p, u16p, u16p, ctypes.POINTER(ci), ctypes.POINTER(ci), u16p, u16p]
        L.sbmo_rtl_det.restype = None
        L.sbmo_rtl_frac.argtypes = [ctypes.c_uint16] * 3
        L.sbmo_rtl_frac.restype = ctypes.c_uint8
        L.sbmo_rtl_pack_disparity.argtypes = [ctypes.c_uint8, ctypes.c_uint8]
        L.sbmo_rtl_pack_disparity.restype = ctypes.c_int16
        L.sbmo_fpga_regs_decode.argtypes = [u32, u32, u32, i32p]
        L.sbmo_fpga_regs_decode.restype = ci
        L.sbmo_fpga_check.argtypes = [ci, ci, ci, ci]
        L.sbmo_fpga_check.restype = ci
        L.sbmo_fpga_bm.argtypes = [u8p, u8p, ci, ci, ci, ci, ci, ci, ci, i16p]
        L.sbmo_fpga_bm.restype = ci
        L.sbmo_fpga_compute.argtypes = [u8p, u8p, ci, ci, ci, ci, ci, ci, ci, i16p]
        L.sbmo_fpga_compute.restype = ci
        L.sbmo_gftt_eig.argtypes = [u8p, ci, ci, u16p, ctypes.POINTER(u32)]
        L.sbmo_gftt_eig.restype = ci
        _LIB = L
    return _LIB


READ_ROI_MINUS_MIND, READ_COST_SHORT, READ_SPECKLE_X16, READ_ODD_ROW_COMPUTED, READ_LR_TIE_LATER = 1, 2, 4, 8, 16


class reading:
    """with sbm_oracle.reading(mask): ... -- the oracle under an alternative reading of the cv::StereoBM behaviours nothing in the
    reference pins (sbm_oracle.h SBMO_READ_*; process-global, so not for concurrent use)."""

    def __init__(self, mask):
        self.mask = int(mask)

    def __enter__(self):
        L = lib()
        self.prev = L.sbmo_get_reading()
        L.sbmo_set_reading(self.mask)
        return self

    def __exit__(self, *exc):
        lib().sbmo_set_reading(self.prev)
        return False


class simd:
    """with sbm_oracle.simd(): compute / compute_batch take the u16-vectorised correspondence stage (sbm_oracle_simd.c) where every
    window sum fits 16 bits -- a TIMING variant for bench.py's cpu_baseline, bit-identical to the scalar restatement."""

    def __init__(self, on=True):
        self.on = 1 if on else 0

    def __enter__(self):
        L = lib()
        self.prev = L.sbmo_get_simd()
        L.sbmo_set_simd(self.on)
        return self

    def __exit__(self, *exc):
        lib().sbmo_set_simd(self.prev)
        return False


def simd_ok(params):
    return bool(lib().sbmo_simd_ok(ctypes.byref(params)))


def _p(a, t):
    return a.ctypes.data_as(ctypes.POINTER(t))


def _u8(a):
    a = np.ascontiguousarray(a, dtype=np.uint8)
    assert a.ndim == 2
    return a


def prefilter_xsobel(img, cap):
    img = _u8(img)
    out = np.empty_like(img)
    h, w = img.shape
    lib().sbmo_prefilter_xsobel(_p(img, ctypes.c_uint8), w, _p(out, ctypes.c_uint8), w, w, h, cap)
    return out


def prefilter_norm(img, winsize, cap):
    img = _u8(img)
    h, w = img.shape
    out = np.empty((h, w), np.uint8)
    L = lib()
    L.sbmo_prefilter_norm.argtypes = [ctypes.POINTER(ctypes.c_uint8), ctypes.c_size_t, ctypes.POINTER(ctypes.c_uint8),
                                      ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    L.sbmo_prefilter_norm.restype = None
    L.sbmo_prefilter_norm(_p(img, ctypes.c_uint8), w, _p(out, ctypes.c_uint8), w, w, h, winsize, cap)
    return out


def prefilter_xsobel_fpga(img, fill=0):
    img = _u8(img)
    out = np.empty_like(img)
    h, w = img.shape
    lib().sbmo_prefilter_xsobel_fpga(_p(img, ctypes.c_uint8), w, _p(out, ctypes.c_uint8), w, w, h, fill)
    return out


def valid_roi(roi1, roi2, min_disparity, num_disparities, block_size):
    a = (ctypes.c_int32 * 4)(*roi1)
    b = (ctypes.c_int32 * 4)(*roi2)
    o = (ctypes.c_int32 * 4)()
    lib().sbmo_valid_roi(a, b, min_disparity, num_disparities, block_size, o)
    return tuple(o)


def find_correspondence(pl, pr, params, row0, row1, brute=False, want_cost=True):
    pl, pr = _u8(pl), _u8(pr)
    h, w = pl.shape
    fill = (params.min_disparity - 1) * 16
    disp = np.full((h, w), fill, dtype=np.int16)
    cost = np.zeros((h, w), dtype=np.int32)
    fn = lib().sbmo_find_correspondence_bruteforce if brute else lib().sbmo_find_correspondence
    fn(_p(pl, ctypes.c_uint8), _p(pr, ctypes.c_uint8), w, w, h, row0, row1, ctypes.byref(params),
       _p(disp, ctypes.c_int16), w, _p(cost, ctypes.c_int32) if want_cost else None, w)
    return disp, cost


def validate_disparity(disp, cost, min_disparity, num_disparities, disp12_max_diff):
    disp = np.ascontiguousarray(disp, dtype=np.int16).copy()
    cost = np.ascontiguousarray(cost, dtype=np.int32)
    h, w = disp.shape
    lib().sbmo_validate_disparity(_p(disp, ctypes.c_int16), w, _p(cost, ctypes.c_int32), w, w, h, min_disparity,
                                  num_disparities, disp12_max_diff)
    return disp


def filter_speckles(img, new_val, max_speckle_size, max_diff):
    img = np.ascontiguousarray(img, dtype=np.int16).copy()
    h, w = img.shape
    lib().sbmo_filter_speckles(_p(img, ctypes.c_int16), w, w, h, new_val, max_speckle_size, max_diff)
    return img


def compute(params, left, right, stages=False):
    """cv::StereoBM::compute restated. Returns disp (int16) or (status, dict of stages) when stages=True."""
    left, right = _u8(left), _u8(right)
    h, w = left.shape
    disp = np.empty((h, w), dtype=np.int16)
    if stages:
        pf_l = np.empty((h, w), np.uint8)
        pf_r = np.empty((h, w), np.uint8)
        pre = np.empty((h, w), np.int16)
        cost = np.empty((h, w), np.int32)
        st = lib().sbmo_compute(ctypes.byref(params), _p(left, ctypes.c_uint8), w, _p(right, ctypes.c_uint8), w, w, h,
                                _p(disp, ctypes.c_int16), 2 * w, _p(pf_l, ctypes.c_uint8), _p(pf_r, ctypes.c_uint8),
                                _p(pre, ctypes.c_int16), _p(cost, ctypes.c_int32))
        return st, dict(disp=disp, pf_l=pf_l, pf_r=pf_r, pre_lr=pre, cost=cost)
    st = lib().sbmo_compute(ctypes.byref(params), _p(left, ctypes.c_uint8), w, _p(right, ctypes.c_uint8), w, w, h,
                            _p(disp, ctypes.c_int16), 2 * w, None, None, None, None)
    if st != 0:
        raise ValueError(f"oracle status {st}")
    return disp


def compute_status(params, width, height):
    """Status code only (parameter validation), on dummy images."""
    z = np.zeros((max(height, 1), max(width, 1)), np.uint8)
    d = np.zeros((max(height, 1), max(width, 1)), np.int16)
    return lib().sbmo_compute(ctypes.byref(params), _p(z, ctypes.c_uint8), max(width, 1), _p(z, ctypes.c_uint8),
                              max(width, 1), width, height, _p(d, ctypes.c_int16), 2 * max(width, 1), None, None, None, None)


def compute_batch(params, left, right, threads=None):
    left = np.ascontiguousarray(left, dtype=np.uint8)
    right = np.ascontiguousarray(right, dtype=np.uint8)
    n, h, w = left.shape
    disp = np.empty((n, h, w), np.int16)
    if threads is None:
        threads = max_threads()
    st = lib().sbmo_compute_batch(ctypes.byref(params), n, _p(left, ctypes.c_uint8), _p(right, ctypes.c_uint8), w, h,
                                  _p(disp, ctypes.c_int16), threads)
    if st != 0:
        raise ValueError(f"oracle status {st}")
    return disp


def max_threads():
    return min(lib().sbmo_max_threads(), os.cpu_count() or 1)


def decimate(disp, scale):
    disp = np.ascontiguousarray(disp, dtype=np.int16)
    h, w = disp.shape
    out = np.empty((h // scale, w // scale), np.int16)
    lib().sbmo_decimate(_p(disp, ctypes.c_int16), w, h, scale, _p(out, ctypes.c_int16))
    return out


def reproject(disp, scale, model, apply_local=True):
    disp = np.ascontiguousarray(disp, dtype=np.int16)
    h, w = disp.shape
    xyz = np.empty((h, w, 3), np.float32)
    lib().sbmo_reproject(_p(disp, ctypes.c_int16), w, h, scale, ctypes.byref(model), 1 if apply_local else 0,
                         _p(xyz, ctypes.c_float))
    return xyz


def keypoints3d(disp, kpts, model, min_depth=0.0, max_depth=0.0):
    disp = np.ascontiguousarray(disp, dtype=np.int16)
    kpts = np.ascontiguousarray(kpts, dtype=np.float32)
    h, w = disp.shape
    xyz = np.empty((len(kpts), 3), np.float32)
    lib().sbmo_keypoints3d(_p(disp, ctypes.c_int16), w, h, _p(kpts, ctypes.c_float), len(kpts), ctypes.byref(model),
                           min_depth, max_depth, _p(xyz, ctypes.c_float))
    return xyz


def rect_map(cam, width, height):
    """rect_remap() of the reference firmware restated: int16 (H, W, 2), (x, y) in 1/32 source pixels."""
    m = np.empty((height, width, 2), np.int16)
    lib().sbmo_rect_map(ctypes.byref(cam), width, height, _p(m, ctypes.c_int16))
    return m


def rect_remap(src, rmap):
    src = _u8(src)
    rmap = np.ascontiguousarray(rmap, dtype=np.int16)
    h, w = src.shape
    out = np.empty((h, w), np.uint8)
    lib().sbmo_rect_remap(_p(src, ctypes.c_uint8), _p(rmap, ctypes.c_int16), w, h, _p(out, ctypes.c_uint8))
    return out


# ---- FPGA flavour of the matcher (sbm_oracle_fpga.c; parity unpinned, restated from src/dvp/rtl/bm*.v) ----------------
def rtl_diven(DW, VW, QW, MSB_INV, dividend, divisor):
    return int(lib().sbmo_rtl_diven(DW, VW, QW, MSB_INV, dividend & 0xffffffff, divisor & 0xffffffff))


def rtl_det(sad34):
    a = np.ascontiguousarray(sad34, dtype=np.uint16)
    assert a.shape == (34,)
    m1, m2, l, r = (ctypes.c_uint16() for _ in range(4))
    i1, i2 = ctypes.c_int(), ctypes.c_int()
    lib().sbmo_rtl_det(_p(a, ctypes.c_uint16), ctypes.byref(m1), ctypes.byref(m2), ctypes.byref(i1), ctypes.byref(i2),
                       ctypes.byref(l), ctypes.byref(r))
    return dict(min1=m1.value, min2=m2.value, idx1=i1.value, idx2=i2.value, l=l.value, r=r.value)


def rtl_frac(c, l, r):
    return int(lib().sbmo_rtl_frac(c, l, r))


def rtl_pack_disparity(disp, frac):
    return int(lib().sbmo_rtl_pack_disparity(disp & 0xff, frac & 0xff))


def fpga_regs_decode(image_size, bm_setting, uni_filt_ctrl=0):
    o = (ctypes.c_int32 * 7)()
    lib().sbmo_fpga_regs_decode(image_size, bm_setting, uni_filt_ctrl, o)
    return dict(zip(("width", "height", "block_size", "num_disparities", "uni_enable", "uni_mode", "uni_threshold"), o))


def fpga_check(width, height, wsz, ndisp):
    return int(lib().sbmo_fpga_check(width, height, wsz, ndisp))


def fpga_bm(xl, xr, wsz, ndisp, uni_enb=0, uni_mode=0, uni_thr=0):
    """The RTL matcher on x-Sobel planes (uint8, 6 significant bits). int16 (H,W), -1 where there is no disparity."""
    xl, xr = _u8(xl), _u8(xr)
    h, w = xl.shape
    d = np.empty((h, w), np.int16)
    st = lib().sbmo_fpga_bm(_p(xl, ctypes.c_uint8), _p(xr, ctypes.c_uint8), w, h, wsz, ndisp, uni_enb, uni_mode, uni_thr,
                            _p(d, ctypes.c_int16))
    if st != 0:
        raise ValueError(f"oracle status {st}")
    return d


def fpga_compute(left, right, wsz, ndisp, uni_enb=0, uni_mode=0, uni_thr=0):
    left, right = _u8(left), _u8(right)
    h, w = left.shape
    d = np.empty((h, w), np.int16)
    st = lib().sbmo_fpga_compute(_p(left, ctypes.c_uint8), _p(right, ctypes.c_uint8), w, h, wsz, ndisp, uni_enb, uni_mode,
                                 uni_thr, _p(d, ctypes.c_int16))
    if st != 0:
        raise ValueError(f"oracle status {st}")
    return d


def gftt_eig(img):
    """PL min-eigenvalue map (uint16 HxW) and the `Max` register value; exact-floor square root."""
    img = _u8(img)
    h, w = img.shape
    e = np.empty((h, w), np.uint16)
    m = ctypes.c_uint32()
    st = lib().sbmo_gftt_eig(_p(img, ctypes.c_uint8), w, h, _p(e, ctypes.c_uint16), ctypes.byref(m))
    if st != 0:
        raise ValueError(f"oracle status {st}")
    return e, int(m.value)
