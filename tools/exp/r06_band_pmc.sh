#!/bin/bash
# round 6: what the speckle band walk spends its time on -- instruction mix and wait counters of the kernel at 64 KITTI / VGA pairs
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r06/band_pmc; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for spec in "kitti 64" "ref640 64" "kitti 1"; do
  set -- $spec; tag=$1$2
  WB="python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --prewarm-s 0 --workload $1 --pairs $2"
  mkdir -p "$OUT/$tag"
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace -d "$OUT/$tag/pmc_sq" -o x -- $WB > "$OUT/$tag/b1.json" 2> "$OUT/$tag/sq.err"
  rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SMEM --kernel-trace -d "$OUT/$tag/pmc_sq2" -o x -- $WB > "$OUT/$tag/b2.json" 2> "$OUT/$tag/sq2.err"
  rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_SALU SQ_INSTS_BRANCH SQ_ACTIVE_INST_MISC --kernel-trace -d "$OUT/$tag/pmc_sq3" -o x -- $WB > "$OUT/$tag/b3.json" 2> "$OUT/$tag/sq3.err"
  python3 $R/tools/rocprof_summary.py "$OUT/$tag" "$OUT/summary_$tag" > "$OUT/$tag/summary.txt" 2>&1
done
cd $R
python3 - <<'PY'
import json, glob, os
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out", "r06", "band_pmc")
for f in sorted(glob.glob(out + "/summary_*_pmc.json")):
    j = json.load(open(f))
    for k in j:
        if "speckle_band" in k or "lrcheck" in k:
            c = {a: round(b["mean"]) for a, b in j[k].items()}
            print(os.path.basename(f), k[:40], json.dumps(c))
PY
