#!/bin/bash
# round 5: where the cooperating-wavefront layouts lose vector-issue time. VALU / LDS busy and wait counters of the SAD kernel
# over (wavefronts per strip 1 / 2) x (window terms 5 / 7) at frame-sized launches. usage (GPU box): bash tools/exp/r05_busy_attrib.sh
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/busy_attrib
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
i=0
for spec in "kitti --block 15" "kitti --block 21" "kitti --block 15 --ndisp 256" "kitti --block 21 --ndisp 256" "fhd --block 15" "fhd --block 21" "fhd --block 21 --pairs 64" "fhd --block 15 --ndisp 128" "fhd --block 21 --ndisp 128"; do
  i=$((i+1)); tag=c$i
  WB="python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --prewarm-s 0 --workload $spec"
  mkdir -p "$OUT/$tag"; echo "$spec" > "$OUT/$tag/spec.txt"
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace -d "$OUT/$tag/pmc_sq" -o x -- $WB > "$OUT/$tag/bench_sq.json" 2> "$OUT/$tag/sq.err"
  rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d "$OUT/$tag/pmc_sq2" -o x -- $WB > "$OUT/$tag/bench_sq2.json" 2> "$OUT/$tag/sq2.err"
  python3 $R/tools/rocprof_summary.py "$OUT/$tag" "$OUT/summary_$tag" > "$OUT/$tag/summary.txt" 2>&1
done
cd $R
python3 - <<'PY'
import json, glob, os
out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "gpurun_out", "busy_attrib")
for d in sorted(glob.glob(out + "/c*/")):
    tag = os.path.basename(d.rstrip("/"))
    spec = open(d + "spec.txt").read().strip()
    f = out + "/summary_" + tag + "_pmc.json"
    if not os.path.exists(f):
        print(tag, spec, "no summary"); continue
    j = json.load(open(f))
    k = [n for n in j if "sad_fast" in n]
    if not k:
        print(tag, spec, "no sad kernel", list(j)[:4]); continue
    c = {a: b["mean"] for a, b in j[k[0]].items()}
    g = c["GRBM_GUI_ACTIVE"] / 8
    line = dict(valu_busy=round(4 * c["SQ_ACTIVE_INST_VALU"] / (1024 * g), 4), lds_busy=round(c["SQ_ACTIVE_INST_LDS"] / (1024 * g) * 4, 4) if "SQ_ACTIVE_INST_LDS" in c else None,
                insts_valu=c.get("SQ_INSTS_VALU"), insts_lds=c.get("SQ_INSTS_LDS"))
    for n in ("SQ_WAIT_INST_LDS", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"):
        if n in c: line[n] = c[n]
    if "SQ_WAVE_CYCLES" in c:
        for n in ("SQ_WAIT_INST_LDS", "SQ_WAIT_INST_ANY", "SQ_WAIT_ANY"):
            line[n + "/wave"] = round(c[n] / c["SQ_WAVE_CYCLES"], 4)
    print(tag, spec, k[0][:60], json.dumps(line))
PY
