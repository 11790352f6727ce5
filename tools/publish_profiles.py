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


def source_hash(root):
    """sha256 over the engine sources (csrc/*.hip, *.h, include/*.h): tools/profile_round.sh records it on the GPU box next to
    the counters, so that counters of one build are never published for another."""
    import hashlib

    h = hashlib.sha256()
    files = sorted((root / "u96-slam_amd" / "csrc").glob("*.hip")) + sorted((root / "u96-slam_amd" / "csrc").glob("*.h")) + sorted((root / "include").glob("*.h"))
    for f in files:
        h.update(f.name.encode())
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


def norm_kernel(name):
    """rocprofv3's 'void sbm::sad_fast_kernel<64, 2, 5, 3, true>(sbm::FastArgs)' -> the engine's 'sad_fast_kernel<64,2,5,3,true>'."""
    n = name.replace("void ", "").replace("sbm::", "").replace(" ", "")
    return n.split("(")[0]


def main():
    if len(sys.argv) == 2 and sys.argv[1] == "--source-hash":
        print(source_hash(Path(__file__).resolve().parents[1]))
        return
    run, rnd = Path(sys.argv[1]), sys.argv[2]
    prof = Path(__file__).resolve().parents[1] / "profiles"
    hf = run / "source_hash.txt"
    here = source_hash(prof.parent)
    if not hf.exists() or hf.read_text().strip() != here:
        sys.exit(f"refusing to publish {run}: its counters were taken on engine sources {hf.read_text().strip() if hf.exists() else '(unrecorded)'}, "
                 f"the working tree is {here} -- re-run tools/profile_round.sh on this build")
    shutil.copy(run / "summary_kernels.md", prof / f"{rnd}_kitti_b64_kernels.md")
    shutil.copy(run / "summary_pmc.json", prof / f"{rnd}_kitti_b64_pmc.json")
    if (run / "summary_ref640t_kernels.md").exists():
        shutil.copy(run / "summary_ref640t_kernels.md", prof / f"{rnd}_ref640_b64_kernels.md")
    for tag in ("fhd64", "uhd32", "frontend"):      # BASELINE's own batch sizes; the PL blocks / front-end kernels
        for suffix in ("kernels.md", "pmc.json"):
            if (run / f"summary_{tag}_{suffix}").exists():
                shutil.copy(run / f"summary_{tag}_{suffix}", prof / f"{rnd}_{tag}_{suffix}")
    if (run / "frontend_kernels.json").exists():
        shutil.copy(run / "frontend_kernels.json", prof / f"{rnd}_frontend_kernels.json")
    for src, dst in (("plain", "plain"), ("trace", "trace"), ("fhd", "fhd"), ("ref640", "ref640"), ("uhd", "uhd"), ("fhd64", "fhd64"), ("uhd32", "uhd32")):
        f = run / f"bench_{src}.json"
        if f.exists():
            (prof / f"{rnd}_bench_{dst}.json").write_text(json.dumps(last_json(f)) + "\n")
    import datetime
    import subprocess

    try:
        commit = subprocess.run(["git", "rev-parse", "--short", "HEAD"], capture_output=True, text=True, cwd=prof.parent).stdout.strip()
    except Exception:
        commit = "?"
    stamp = datetime.date.today().isoformat()
    shapes = {"kitti": (1242, 375, 128), "fhd": (1920, 1080, 256), "uhd": (3840, 2160, 256), "ref640": (640, 480, 64)}
    tj = {}
    for wl, pmcfile, benchfile, tag in (("kitti", run / "summary_pmc.json", run / "bench_plain.json", f"{rnd}_kitti_b64_pmc.json"),
                                        ("fhd", run / "summary_fhd_pmc.json", run / "bench_fhd.json", f"{rnd}_fhd_pmc.json"),
                                        ("uhd", run / "summary_uhd_pmc.json", run / "bench_uhd.json", f"{rnd}_uhd_pmc.json"),
                                        ("ref640", run / "summary_ref640_pmc.json", run / "bench_ref640.json", f"{rnd}_ref640_pmc.json"),
                                        ("fhd", run / "summary_fhd64_pmc.json", run / "bench_fhd64.json", f"{rnd}_fhd64_pmc.json"),
                                        ("uhd", run / "summary_uhd32_pmc.json", run / "bench_uhd32.json", f"{rnd}_uhd32_pmc.json")):
        if not pmcfile.exists() or not benchfile.exists():
            continue
        pmc = json.loads(pmcfile.read_text())
        if wl != "kitti" and pmcfile.resolve() != (prof / tag).resolve():
            shutil.copy(pmcfile, prof / tag)
        bench = last_json(benchfile)
        B = bench["config"]["pairs_per_gpu_per_step"]
        wsz = int(bench["config"]["workload"].split(" SAD")[0].split(", ")[-1].split("x")[0])
        name = next(k for k in pmc if "sad_fast_kernel" in k)
        e = pmc[name]
        W, H, nd = shapes[wl]
        # `kernel` = the engine's own name of the instantiation the un-profiled bench line of this run launched (bench.py compares it
        # with sbm_last_kernel_name() of later runs); it must be the kernel the counters were read from
        eng = bench["roofline"]["kernel"]
        if norm_kernel(name) != eng.split(" ")[0]:
            sys.exit(f"{wl}: counters are of {norm_kernel(name)}, the bench line launched {eng}")
        ent = {"kernel": eng, "rocprof_kernel": name, "source": f"profiles/{tag}", "date": stamp, "commit": commit, "source_hash": here}
        if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
            fetch, write = e["FETCH_SIZE"]["mean"], e["WRITE_SIZE"]["mean"]
            ent.update({"bytes_per_launch": int(round((2 * fetch + write) * 1024)), "fetch_size_kb": fetch, "write_size_kb": write,
                        "correction": "2*FETCH_SIZE + WRITE_SIZE (profiles/hbm_calibration.json)"})
        if "SQ_ACTIVE_INST_VALU" in e and "GRBM_GUI_ACTIVE" in e:
            ent["valu_busy_frac"] = round(4.0 * e["SQ_ACTIVE_INST_VALU"]["mean"] / (1024 * e["GRBM_GUI_ACTIVE"]["mean"] / 8.0), 4)
            ent["valu_busy_formula"] = "4*SQ_ACTIVE_INST_VALU / (1024 SIMDs * GRBM_GUI_ACTIVE/8 XCDs)"
        if "SQ_ACTIVE_INST_LDS" in e and "GRBM_GUI_ACTIVE" in e:
            ent["lds_busy_frac"] = round(4.0 * e["SQ_ACTIVE_INST_LDS"]["mean"] / (1024 * e["GRBM_GUI_ACTIVE"]["mean"] / 8.0), 4)
        if "SQ_INSTS_VALU" in e:
            ent["valu_wave_instructions_per_launch"] = e["SQ_INSTS_VALU"]["mean"]
            ent["lane_ops_per_pixel_disparity"] = round(e["SQ_INSTS_VALU"]["mean"] * 64.0 / (float(B) * W * H * nd), 3)
        tj[f"{wl}_w{wsz}_b{B}"] = ent
    (prof / "hbm_traffic.json").write_text(json.dumps(tj, indent=1) + "\n")
    print(json.dumps(tj, indent=1))


if __name__ == "__main__":
    main()
