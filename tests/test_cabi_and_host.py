"""CPU-side checks of the boundary: the C-ABI library loads and exports every symbol include/sbm.h declares, the
parameter checks agree with the oracle's restatement of cv::StereoBM's, the host mirror behaves like cv::StereoBM
without a GPU (fails loudly), and the C++ adaptor header compiles against the library. No compute calls here."""
import ctypes
import pathlib
import re
import subprocess

import numpy as np
import pytest

ROOT = pathlib.Path(__file__).resolve().parents[1]


def declared_symbols():
    text = (ROOT / "include" / "sbm.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sbm_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol(pkg):
    lib = pkg.load_library()
    names = declared_symbols()
    assert len(names) >= 17
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/sbm.h but not exported"
    assert lib.sbm_version() >= 1


def test_param_validation_matches_oracle(pkg, oracle):
    from u96_slam_amd import stereobm

    rng = np.random.default_rng(1)
    for _ in range(300):
        kw = dict(num_disparities=int(rng.choice([0, 8, 16, 20, 48, 64, 128, 4112])), block_size=int(rng.integers(1, 40)),
                  prefilter_cap=int(rng.integers(-1, 70)), texture_threshold=int(rng.integers(-2, 20)),
                  uniqueness_ratio=int(rng.integers(-2, 20)), prefilter_size=int(rng.choice([3, 4, 5, 9, 255, 257])),
                  prefilter_type=int(rng.choice([0, 1, 1, 1, 2])))
        w, h = int(rng.integers(1, 80)), int(rng.integers(1, 60))
        po = oracle.make_params(**kw)
        pe = stereobm.SbmParams()
        ctypes.memmove(ctypes.byref(pe), ctypes.byref(po), ctypes.sizeof(pe))
        assert stereobm.validate(pe, w, h) == oracle.compute_status(po, w, h), (kw, w, h)


def test_defaults_match_cv_create(pkg):
    from u96_slam_amd import stereobm

    p = stereobm.SbmParams()
    pkg.load_library().sbm_params_default(ctypes.byref(p), 0, 0)
    assert (p.num_disparities, p.block_size, p.prefilter_type, p.prefilter_size, p.prefilter_cap) == (64, 21, 1, 9, 31)
    assert (p.min_disparity, p.texture_threshold, p.uniqueness_ratio, p.speckle_window_size, p.speckle_range,
            p.disp12_max_diff) == (0, 10, 15, 0, 0, -1)
    pkg.load_library().sbm_params_default(ctypes.byref(p), 16, 9)       # main.cpp:201
    assert (p.num_disparities, p.block_size) == (16, 9)


def test_no_gpu_fails_loudly(pkg):
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pkg.StereoBMError) as e:
        pkg.StereoBM.create(64, 21)
    assert e.value.code == -20 and "no CPU backend" in str(e.value)


def test_strerror_covers_all_codes(pkg):
    lib = pkg.load_library()
    for code in (0, -1, -2, -3, -4, -5, -6, -7, -8, -9, -20, -21, -22, -23, -24):
        assert lib.sbm_strerror(code).decode() != "unknown status"
    assert lib.sbm_strerror(-99).decode() == "unknown status"


def test_product_never_imports_oracle():
    for f in (ROOT / "u96-slam_amd").rglob("*"):
        if f.suffix in (".py", ".hip", ".h", ".hpp", ".cpp"):
            assert "oracle" not in f.read_text().replace("the CPU oracle lives in /oracle", "").replace("CPU oracle", "").lower() \
                or f.name == "__init__.py", f


def test_cpp_adaptor_compiles_and_links(tmp_path, pkg):
    src = tmp_path / "t.cpp"
    src.write_text(r'''
#include "sbm_stereobm.hpp"
#include <cstdio>
int main() {
  sbm_params p; sbm_params_default(&p, 16, 9);
  if (p.num_disparities != 16 || p.block_size != 9) return 1;
  if (sbm_params_validate(&p, 640, 480) != SBM_OK) return 2;
  p.num_disparities = 20;
  if (sbm_params_validate(&p, 640, 480) != SBM_ERR_NUM_DISPARITIES) return 3;
  try {
    auto bm = sbm::StereoBM::create(16, 9);   // main.cpp:201 spelling
    bm->setPreFilterCap(31); bm->setBlockSize(21); bm->setMinDisparity(0); bm->setNumDisparities(64);
    bm->setTextureThreshold(10); bm->setUniquenessRatio(10); bm->setSpeckleWindowSize(50); bm->setSpeckleRange(32);
    bm->setDisp12MaxDiff(1);
    std::printf("created\n");
  } catch (const sbm::Error& e) {
    std::printf("error %d\n", e.code);
    return e.code == SBM_ERR_NO_DEVICE ? 0 : 4;
  }
  return 0;
}
''')
    exe = tmp_path / "t"
    lib = pkg.library_path()
    r = subprocess.run(["g++", "-std=c++17", "-I", str(ROOT / "include"), str(src), "-o", str(exe), str(lib),
                        f"-Wl,-rpath,{lib.parent}", "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True)
    assert r.returncode == 0, (r.stdout, r.stderr)


def test_fpga_register_decode_matches_oracle_and_firmware(pkg, oracle):
    """Host side of the FPGA flavour (no GPU needed): sbm_fpga_params_from_regs / _validate / _sad_size_reg against the
    oracle's restatement of bm.v:172-193,208,249-255 and against the words the firmware writes (fpga.c:155,158)."""
    p = pkg.fpga_params_from_regs((480 << 16) + 640, 0x00150040, 0)
    assert (p.width, p.height, p.block_size, p.num_disparities, p.uni_enable, p.uni_mode, p.uni_threshold) == (640, 480, 21, 64, 0, 0, 0)
    # bm.v:235-245 documents sad_wdt 491 / sad_hgt 460 for ndisp 128; with the firmware's ndisp 64: 640-64-1-20 = 555
    assert pkg.fpga_sad_size_reg(p) == (460 << 16) | 555
    rng = np.random.default_rng(0)
    for _ in range(300):
        regs = [int(v) for v in rng.integers(0, 2**32, 3, dtype=np.uint64)]
        q = pkg.fpga_params_from_regs(*regs)
        o = oracle.fpga_regs_decode(*regs)
        assert (q.width, q.height, q.block_size, q.num_disparities, q.uni_enable, q.uni_mode, q.uni_threshold) == tuple(o.values())
        assert pkg.fpga_validate(q) == oracle.fpga_check(q.width, q.height, q.block_size, q.num_disparities)
