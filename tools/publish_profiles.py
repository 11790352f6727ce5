#!/usr/bin/env python3
"""Copy the judged summaries of a tools/profile_round.sh run from gpurun_out/ into profiles/ and refresh
profiles/hbm_traffic.json (HBM bytes per launch of the SAD kernel, corrected as profiles/hbm_calibration.json says).

usage: publish_profiles.py gpurun_out/prof_<tag> r01
"""
import json
import shutil
import sys
from pathlib import Path


def last_json(p):
    lines = [l for l in Path(p).read_text().splitlines() if l.startswith("{")]
    return json.loads(lines[-1])


def main():
    run, rnd = Path(sys.argv[1]), sys.argv[2]
    prof = Path(__file__).resolve().parents[1] / "profiles"
    shutil.copy(run / "summary_kernels.md", prof / f"{rnd}_kitti_b64_kernels.md")
    shutil.copy(run / "summary_pmc.json", prof / f"{rnd}_kitti_b64_pmc.json")
    for src, dst in (("plain", "plain"), ("trace", "trace"), ("fhd", "fhd"), ("ref640", "ref640"), ("uhd", "uhd")):
        f = run / f"bench_{src}.json"
        if f.exists():
            (prof / f"{rnd}_bench_{dst}.json").write_text(json.dumps(last_json(f)) + "\n")
    pmc = json.loads((run / "summary_pmc.json").read_text())
    bench = last_json(run / "bench_plain.json")
    B = bench["config"]["pairs_per_gpu_per_step"]
    name = next(k for k in pmc if "sad_fast_kernel" in k)
    e = pmc[name]
    fetch, write = e["FETCH_SIZE"]["mean"], e["WRITE_SIZE"]["mean"]
    valu = 4.0 * e["SQ_ACTIVE_INST_VALU"]["mean"] / (1024 * e["GRBM_GUI_ACTIVE"]["mean"] / 8.0)
    lds = 4.0 * e["SQ_ACTIVE_INST_LDS"]["mean"] / (1024 * e["GRBM_GUI_ACTIVE"]["mean"] / 8.0)
    tj = {
        f"kitti_w15_b{B}": {
            "kernel": name,
            "bytes_per_launch": int(round((2 * fetch + write) * 1024)),
            "fetch_size_kb": fetch,
            "write_size_kb": write,
            "correction": "2*FETCH_SIZE + WRITE_SIZE (profiles/hbm_calibration.json)",
            "valu_busy_frac": round(valu, 4),
            "lds_busy_frac": round(lds, 4),
            "valu_busy_formula": "4*SQ_ACTIVE_INST_VALU / (1024 SIMDs * GRBM_GUI_ACTIVE/8 XCDs)",
            "source": f"profiles/{rnd}_kitti_b64_pmc.json",
        }
    }
    (prof / "hbm_traffic.json").write_text(json.dumps(tj, indent=1) + "\n")
    print(json.dumps(tj, indent=1))


if __name__ == "__main__":
    main()
