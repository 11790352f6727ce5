#!/usr/bin/env python3
"""Stand-alone check of a pin kit against the real cv::StereoBM. Needs numpy and cv2 (any OpenCV >= 3) -- nothing from this
repository.  usage: python verify_with_opencv.py pin_kit.npz

For every case of the kit it runs cv2.StereoBM with the stored parameters on the stored inputs, stage by stage (filters switched
on one at a time), and prints OK or the FIRST stage that differs from the stored output of the MI355X engine, with the number of
differing pixels and the first differing (row, column, engine value, OpenCV value). Exit code 0 = every case agrees.

Kit v2: for the behaviours the engine restates from memory the kit also holds the engine's output under the ALTERNATIVE reading
(`<case>/alt<bit>/<stage>`, one bit of SBM_CV_READING each). A case that differs from the default but equals an alternative is
reported as such, and the summary names the bits to flip (u96-slam_amd/csrc/sbm_common.h kRead*, oracle/sbm_oracle.h SBMO_READ_*)."""
import sys

import numpy as np


def make_bm(cv2, p):
    bm = cv2.StereoBM_create(numDisparities=int(p["num_disparities"]), blockSize=int(p["block_size"]))
    bm.setPreFilterType(int(p["prefilter_type"])); bm.setPreFilterSize(int(p["prefilter_size"])); bm.setPreFilterCap(int(p["prefilter_cap"]))
    bm.setMinDisparity(int(p["min_disparity"])); bm.setTextureThreshold(int(p["texture_threshold"]))
    bm.setUniquenessRatio(int(p["uniqueness_ratio"])); bm.setSpeckleWindowSize(int(p["speckle_window_size"]))
    bm.setSpeckleRange(int(p["speckle_range"])); bm.setDisp12MaxDiff(int(p["disp12_max_diff"]))
    bm.setROI1((int(p["roi1_x"]), int(p["roi1_y"]), int(p["roi1_w"]), int(p["roi1_h"])))
    bm.setROI2((int(p["roi2_x"]), int(p["roi2_y"]), int(p["roi2_w"]), int(p["roi2_h"])))
    return bm


def stage_params(p, stage):
    q = dict(p)
    if stage == "s0_wta":
        q.update(uniqueness_ratio=0, texture_threshold=0, disp12_max_diff=-1, speckle_window_size=0, speckle_range=0)
    elif stage == "s1_uniq":
        q.update(disp12_max_diff=-1, speckle_window_size=0, speckle_range=0)
    elif stage == "s2_lr":
        q.update(speckle_window_size=0, speckle_range=0)
    return q


def main():
    import cv2

    kit = np.load(sys.argv[1])
    fields = [str(f) for f in kit["fields"]]
    print("OpenCV", cv2.__version__, "--", len(kit["names"]), "cases")
    bad = 0
    risk = {}
    if "risk_bits" in kit.files:
        for b, cs, txt in zip(kit["risk_bits"].tolist(), kit["risk_cases"].tolist(), kit["risk_text"].tolist()):
            for c in str(cs).split(","):
                risk.setdefault(c, []).append((int(b), str(txt)))
    flips = {}
    for name in kit["names"]:
        name = str(name)
        p = dict(zip(fields, kit[f"{name}/params"].tolist()))
        L, R = kit[f"{name}/left"], kit[f"{name}/right"]
        got_all = {str(st): make_bm(cv2, stage_params(p, str(st))).compute(L, R) for st in kit["stages"]}
        verdict = "OK"
        for st in kit["stages"]:
            st = str(st)
            want, got = kit[f"{name}/{st}"], got_all[st]
            if not np.array_equal(got, want):
                alt = [(b, txt) for b, txt in risk.get(name, []) if all(np.array_equal(got_all[str(s2)], kit[f"{name}/alt{b}/{s2}"]) for s2 in kit["stages"])]
                if alt:
                    b, txt = alt[0]
                    verdict = f"ALTERNATIVE READING bit {b} matches at every stage: {txt}"
                    flips[b] = txt
                else:
                    ys, xs = np.nonzero(got != want)
                    verdict = (f"FIRST DIFFERENCE at stage {st}: {len(ys)} pixels, first at row {ys[0]} column {xs[0]}: "
                               f"engine {int(want[ys[0], xs[0]])}, OpenCV {int(got[ys[0], xs[0]])}; rows {ys.min()}..{ys.max()}, columns {xs.min()}..{xs.max()}")
                bad += 1
                break
        print(f"{name:24s} {verdict}")
    if bad == 0:
        print("ALL CASES AGREE: the engine's block-matching output is pinned to this OpenCV build")
    else:
        print(f"{bad} case(s) differ")
        if flips:
            print("(cases reported as FIRST DIFFERENCE may simply follow from the alternative readings below: the kit only stores the "
                  "alternatives for the cases built to tell them apart -- flip, regenerate, re-run)")
            print("this OpenCV implements the alternative reading of: " + "; ".join(f"bit {b} ({t})" for b, t in sorted(flips.items())))
            print(f"-> make SBM_CV_READING / sbmo_set_reading default to {sum(flips)} (sbm_api.hip: env_switch(\"SBM_CV_READING\", ...); oracle g_reading), regenerate the kit")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
