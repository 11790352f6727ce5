#!/usr/bin/env python3
"""Pin kit for the block-matching output.

The reference calls cv::StereoBM (src/slam/src/core/main.cpp:201-215); OpenCV exists in neither image, so the engine and its
oracle are checked against an in-repo restatement only ("parity unpinned"). This script makes settling that a one-command event
for anyone who has OpenCV:

  on the GPU box:    python tools/pin_kit.py --out tests/golden/pin_kit.npz       (writes the kit + pin_kit.sha256)
  anywhere with cv2: python tools/verify_with_opencv.py tests/golden/pin_kit.npz  (numpy + cv2 only, no repo import)

One case per risk item of SURVEY.md A.7 plus the one VERDICT r04 added (getValidDisparityROI's "- minDisparity"): for every case
the inputs, the parameter block and the ENGINE's outputs (HIP path, through the C-ABI) at four stages -- filters switched on one
at a time, so that a difference names the first stage that disagrees:
  s0_wta   uniqueness 0, texture 0, no LR check, no speckle filter (winner search + sub-pixel + valid-ROI fill)
  s1_uniq  + texture threshold and uniqueness ratio of the case
  s2_lr    + disp12MaxDiff of the case
  s3_full  + speckle filter of the case
Every engine output is also compared with the oracle here; a kit is only written when they agree everywhere.

Kit v2 (round 6): every risk also has its ALTERNATIVE reading in the engine (environment SBM_CV_READING, sbm_common.h kRead*) and in
the oracle (sbmo_set_reading); for the cases that can tell the two readings apart the kit carries the engine's outputs under the
alternative as `<case>/alt<bit>/<stage>`, and `risks` lists (bit, cases, description). tools/verify_with_opencv.py then NAMES the
reading a given OpenCV implements; adopting it is a default flip of one bit, not a rewrite."""
import argparse
import hashlib
import pathlib
import sys

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parents[1]
FIELDS = ("prefilter_type", "prefilter_size", "prefilter_cap", "block_size", "min_disparity", "num_disparities", "texture_threshold",
          "uniqueness_ratio", "speckle_window_size", "speckle_range", "disp12_max_diff", "roi1_x", "roi1_y", "roi1_w", "roi1_h",
          "roi2_x", "roi2_y", "roi2_w", "roi2_h")
STAGES = ("s0_wta", "s1_uniq", "s2_lr", "s3_full")
# bit of SBM_CV_READING / sbmo_set_reading -> (cases that can tell the readings apart, what the alternative reading is)
RISKS = {
    1: (("mind_neg8_rois", "mind_pos4_rois"), "getValidDisparityROI: xmax = min(roi1 right edge, roi2 right edge - minDisparity) - w/2 (OpenCV 2.4 lineage)"),
    2: (("cost_wrap_ramps",), "validateDisparity sees the block-matching cost plane as `short`: sums beyond 32 767 wrap"),
    4: (("speckle_range1", "speckle_range16"), "filterSpeckles receives speckleRange * 16 (StereoSGBM's convention)"),
    8: (("odd_height",), "prefilterXSobel computes the last row of an odd-height image (reflect-101) instead of filling it with preFilterCap"),
    16: (("lr_ties_d12_0", "lr_ties_d12_1", "lr_ties_d12_2"), "validateDisparity: on equal cost the LATER x takes the right-view slot ('>=' instead of '>')"),
}


def params(nd=64, w=21, cap=31, mind=0, tex=10, uniq=10, spw=50, spr=32, d12=1, ptype=1, psize=9, roi1=(0, 0, 0, 0), roi2=(0, 0, 0, 0)):
    return dict(prefilter_type=ptype, prefilter_size=psize, prefilter_cap=cap, block_size=w, min_disparity=mind, num_disparities=nd,
                texture_threshold=tex, uniqueness_ratio=uniq, speckle_window_size=spw, speckle_range=spr, disp12_max_diff=d12,
                roi1=tuple(roi1), roi2=tuple(roi2))


def stage_params(p, stage):
    q = dict(p)
    if stage == "s0_wta":
        q.update(uniqueness_ratio=0, texture_threshold=0, disp12_max_diff=-1, speckle_window_size=0, speckle_range=0)
    elif stage == "s1_uniq":
        q.update(disp12_max_diff=-1, speckle_window_size=0, speckle_range=0)
    elif stage == "s2_lr":
        q.update(speckle_window_size=0, speckle_range=0)
    return q


def flat(p):
    return np.array([p["prefilter_type"], p["prefilter_size"], p["prefilter_cap"], p["block_size"], p["min_disparity"], p["num_disparities"],
                     p["texture_threshold"], p["uniqueness_ratio"], p["speckle_window_size"], p["speckle_range"], p["disp12_max_diff"],
                     *p["roi1"], *p["roi2"]], np.int32)


def tie_pair(W, H, nd, period=8):
    """Horizontally periodic texture: equal sums at disparities one period apart (LR-check and winner tie rules, A.5)."""
    rng = np.random.default_rng(77)
    tile = rng.integers(40, 216, (H, period), dtype=np.uint8)
    L = np.tile(tile, (1, W // period + 2))[:, :W].copy()
    L[:, W // 2:] = np.tile(rng.integers(0, 256, (H, 5), dtype=np.uint8), (1, W // 5 + 2))[:, :W - W // 2]
    R = np.roll(L, -9, axis=1)
    return np.ascontiguousarray(L), np.ascontiguousarray(R)


def slant_pair(W, H, nd):
    """Slanted surface (disparity 6 + 0.16 x px, bilinear resampling of a box-filtered texture) with a few displaced
    blobs: horizontal neighbours differ by 2-3 sixteenths of a pixel, so speckleRange 1 and 16 cut the components differently (A.6:
    the range is compared UNSCALED, in 1/16 px), and the blobs make real speckles."""
    rng = np.random.default_rng(78)
    T = rng.integers(0, 256, (H, 2 * W + 2 * nd + 8)).astype(np.float64)
    T = (T + np.roll(T, 1, 0) + np.roll(T, 1, 1) + np.roll(np.roll(T, 1, 0), 1, 1)) / 4
    x = np.arange(W)
    L = T[:, nd:nd + W]
    pos = x + nd + 6 + 0.16 * x
    i0 = np.floor(pos).astype(int)
    f = pos - i0
    R = T[:, i0] * (1 - f) + T[:, i0 + 1] * f
    for _ in range(18):
        y, xx, h, w = int(rng.integers(4, H - 12)), int(rng.integers(nd + 12, W - 16)), int(rng.integers(3, 8)), int(rng.integers(3, 8))
        R[y:y + h, xx - 9:xx - 9 + w] = L[y:y + h, xx:xx + w]          # blob at disparity 9, off the surface
    return np.clip(L, 0, 255).astype(np.uint8), np.clip(np.rint(R), 0, 255).astype(np.uint8)


def ramp_pair(W, H, blk=24):
    """Opposed sawtooth ramps whose slope changes every `blk` columns and every 16 rows: the left image rises along x (x-Sobel clips
    at +cap nearly everywhere), the right one falls (-cap), so EVERY disparity's window sum is close to its maximum w^2 * 2 * cap --
    winners with costs on both sides of 32 767 and many different disparities competing for the same right-view columns, which is
    what a `short` cost plane would decide differently (A.7 item 2; ordinary frames never get there: their minima are far below)."""
    rng = np.random.default_rng(81)

    def saw(sign):
        out = np.zeros((H, W), np.int64)
        for r0 in range(0, H, 16):
            slopes = rng.integers(5, 14, (W + blk - 1) // blk)
            acc = np.cumsum(np.repeat(slopes, blk)[:W]) + rng.integers(0, 256)
            out[r0:r0 + 16] = acc[None, :] + (np.arange(min(16, H - r0))[:, None] * int(rng.integers(1, 7)))
        v = (out + rng.integers(0, 4, (H, W))) % 256
        return v if sign > 0 else 255 - v

    return saw(1).astype(np.uint8), saw(-1).astype(np.uint8)


def patch_pair(synth, W, H, nd):
    """Textured pair with constant patches (texture sum 0 there) and faint-texture patches around the threshold."""
    L, R = synth.make_batch(905, 1, W, H, nd)
    L, R = L[0].copy(), R[0].copy()
    rng = np.random.default_rng(79)
    for k in range(6):
        y, x, h, w = int(rng.integers(0, H - 30)), int(rng.integers(nd, W - 60)), int(rng.integers(18, 30)), int(rng.integers(30, 60))
        if k % 2 == 0:
            L[y:y + h, x:x + w] = 120; R[y:y + h, x - 12:x - 12 + w] = 120
        else:
            faint = 120 + rng.integers(-1, 2, (h, w))
            L[y:y + h, x:x + w] = faint; R[y:y + h, x - 12:x - 12 + w] = faint
    return L, R


def cases(golden, synth):
    g_l, g_r = golden["rect_l"][140:340], golden["rect_r"][140:340]          # 200 rows of the reference's bundled pair
    out = []

    def synth_pair(seed, W, H, nd):
        L, R = synth.make_batch(seed, 1, W, H, nd)
        return L[0], R[0]

    out.append(("ref_pair_w21_callsite", g_l, g_r, params(64, 21), "the reference's call-site parameters (main.cpp:204-212) on data/ref_rect_*"))
    out.append(("ref_pair_w9", g_l, g_r, params(64, 9), "BASELINE configs[0]: 9x9"))
    s64 = synth_pair(900, 320, 96, 64)
    out.append(("mind_neg8", *s64, params(64, 15, mind=-8), "A.7 + VERDICT r04: minDisparity < 0 (valid ROI, FILTERED value, border fill)"))
    out.append(("mind_pos4", *s64, params(64, 15, mind=4), "minDisparity > 0"))
    out.append(("mind_neg8_rois", *s64, params(64, 15, mind=-8, roi1=(20, 10, 260, 70), roi2=(30, 12, 250, 72)),
                "getValidDisparityROI with non-empty ROIs and minDisparity < 0: decides the '- minDisparity' in xmax (2.4 lineage)"))
    out.append(("mind_pos4_rois", *s64, params(64, 15, mind=4, roi1=(20, 10, 260, 70), roi2=(30, 12, 250, 72)), "same, minDisparity > 0"))
    out.append(("rois_only", *s64, params(64, 15, roi1=(16, 8, 280, 80), roi2=(8, 4, 300, 88)), "A.7 item 6: non-empty ROI1 / ROI2, minDisparity 0"))
    out.append(("cost_w15_cap63", *s64, params(64, 15, cap=63), "control: w^2 * 2 * cap = 28 350 fits a short cost plane"))
    out.append(("cost_w17_cap63", *s64, params(64, 17, cap=63), "A.7 item 2: 36 414 > 32 767 -- a short cost plane wraps, the LR check sees it"))
    out.append(("cost_w23_cap31", *s64, params(64, 23), "32 798 > 32 767 by a hair"))
    out.append(("cost_w27_cap31", *s64, params(64, 27), "45 198"))
    out.append(("cost_wrap_ramps", *ramp_pair(320, 96), params(32, 27, tex=0, uniq=0, spw=0, spr=0, d12=1),
                "opposed ramps, 27 x 27 at cap 31 (45 198): winning costs on both sides of 32 767 -- the case that tells a `short` cost plane from an exact one"))
    out.append(("odd_height", *synth_pair(901, 320, 97, 64), params(64, 15), "A.7 item 3: odd H, last prefilter row = cap"))
    sl = slant_pair(320, 96, 64)
    out.append(("speckle_range1", *sl, params(64, 9, tex=0, uniq=5, spw=150, spr=1), "A.7 item 1: speckleRange 1 (x16 if the range were scaled)"))
    out.append(("speckle_range16", *sl, params(64, 9, tex=0, uniq=5, spw=150, spr=16), "speckleRange 16"))
    pp = patch_pair(synth, 320, 96, 32)
    out.append(("texture_10", *pp, params(32, 9, tex=10, uniq=5, spw=0, spr=0), "texture threshold 10 on constant and faint patches"))
    out.append(("texture_200", *pp, params(32, 9, tex=200, uniq=5, spw=0, spr=0), "texture threshold 200"))
    tp = tie_pair(320, 64, 32)
    for d12 in (0, 1, 2):
        out.append((f"lr_ties_d12_{d12}", *tp, params(32, 9, tex=0, uniq=0, spw=0, spr=0, d12=d12), "A.7 item 4: periodic texture, equal costs -- LR claim / tie rules"))
    out.append(("uniq_0", *s64, params(64, 15, uniq=0), "uniquenessRatio 0"))
    out.append(("uniq_25", *s64, params(64, 15, uniq=25), "uniquenessRatio 25"))
    out.append(("prefilter_norm_9", *s64, params(64, 15, ptype=0, psize=9), "PREFILTER_NORMALIZED_RESPONSE, preFilterSize 9"))
    out.append(("prefilter_norm_5", *s64, params(64, 15, ptype=0, psize=5), "PREFILTER_NORMALIZED_RESPONSE, preFilterSize 5"))
    out.append(("nd16_w5", *synth_pair(902, 200, 64, 16), params(16, 5), "smallest window / disparity count"))
    out.append(("nd128_w11", *synth_pair(903, 400, 80, 128), params(128, 11), "128 disparities"))
    return out


def digest(arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=str(ROOT / "tests" / "golden" / "pin_kit.npz"))
    a = ap.parse_args()
    sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle"))
    import _pkg
    import sbm_oracle

    pkg = _pkg.load()
    from u96_slam_amd import synth

    golden = np.load(ROOT / "tests" / "golden" / "ref_pair_640x480.npz")
    import os

    blob, lines = {}, []
    names = []
    told_apart = {b: 0 for b in RISKS}

    def run_stages(L, R, p, reading):
        outs = []
        for st in STAGES:
            q = stage_params(p, st)
            bm = pkg.StereoBM.create(q["num_disparities"], q["block_size"])
            bm.setPreFilterType(q["prefilter_type"]); bm.setPreFilterSize(q["prefilter_size"]); bm.setPreFilterCap(q["prefilter_cap"])
            bm.setMinDisparity(q["min_disparity"]); bm.setTextureThreshold(q["texture_threshold"]); bm.setUniquenessRatio(q["uniqueness_ratio"])
            bm.setSpeckleWindowSize(q["speckle_window_size"]); bm.setSpeckleRange(q["speckle_range"]); bm.setDisp12MaxDiff(q["disp12_max_diff"])
            bm.setROI1(q["roi1"]); bm.setROI2(q["roi2"])
            os.environ["SBM_CV_READING"] = str(reading)        # (read per call by the engine)
            try:
                got = bm.compute(L, R)
            finally:
                os.environ.pop("SBM_CV_READING", None)
            po = sbm_oracle.make_params(q["num_disparities"], q["block_size"], q["prefilter_cap"], q["min_disparity"], q["texture_threshold"],
                                        q["uniqueness_ratio"], q["speckle_window_size"], q["speckle_range"], q["disp12_max_diff"],
                                        q["prefilter_type"], q["prefilter_size"], q["roi1"], q["roi2"])
            with sbm_oracle.reading(reading):
                ref = sbm_oracle.compute(po, L, R)
            if not np.array_equal(got, ref):
                raise SystemExit(f"{name} {st} reading {reading}: engine and oracle differ in {(got != ref).sum()} pixels -- no kit written")
            outs.append(got)
        return outs, bm

    for name, L, R, p, why in cases(golden, synth):
        names.append(name)
        blob[f"{name}/left"], blob[f"{name}/right"], blob[f"{name}/params"] = L, R, flat(p)
        outs, bm = run_stages(L, R, p, 0)
        for st, got in zip(STAGES, outs):
            blob[f"{name}/{st}"] = got
        hashed = [L, R, flat(p), *outs]
        alts = []
        for bit, (risk_cases, _) in RISKS.items():
            if name in risk_cases:
                aouts, _bm = run_stages(L, R, p, bit)
                for st, got in zip(STAGES, aouts):
                    blob[f"{name}/alt{bit}/{st}"] = got
                hashed += aouts
                ndiff = sum(int((a != b).sum()) for a, b in zip(aouts, outs))
                told_apart[bit] += ndiff > 0
                alts.append(f"alt{bit}:{ndiff}px")
        d = digest(hashed)
        lines.append(f"{d}  {name}  {L.shape[1]}x{L.shape[0]}  valid(s3)={float((outs[-1] > (p['min_disparity'] - 1) * 16).mean()):.3f}  {' '.join(alts)}  # {why}")
        print(lines[-1], flush=True)
    for bit, n in told_apart.items():
        if n == 0:
            raise SystemExit(f"no case of the kit tells reading bit {bit} from the default -- no kit written")
    blob["risk_bits"] = np.array(sorted(RISKS), np.int32)
    blob["risk_cases"] = np.array([",".join(RISKS[b][0]) for b in sorted(RISKS)])
    blob["risk_text"] = np.array([RISKS[b][1] for b in sorted(RISKS)])
    blob["names"] = np.array(names)
    blob["fields"] = np.array(FIELDS)
    blob["stages"] = np.array(STAGES)
    out = pathlib.Path(a.out)
    np.savez_compressed(out, **blob)
    out.with_suffix(".sha256").write_text("\n".join(lines) + "\n")
    print("wrote", out, out.stat().st_size, "bytes;", len(names), "cases, kernel of the last case:", bm.last_kernel())


if __name__ == "__main__":
    main()
