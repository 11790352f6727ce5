#!/bin/bash
# Collect the rocprofv3 evidence behind bench.py's roofline line on the GPU box (run through gpurun from the repo root):
#   tools/profile_round.sh <tag>      -> gpurun_out/prof_<tag>/{trace,pmc_*}/, summarised by tools/rocprof_summary.py
# One kernel-trace pass (durations) and separate --pmc passes (never combined with sys/hip/hsa tracing).
set -u
TAG=${1:-r01}
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_$TAG
mkdir -p "$OUT"
python3 "$R/tools/publish_profiles.py" --source-hash > "$OUT/source_hash.txt"   # which engine sources these counters belong to
cd /tmp && export TMPDIR=/tmp
BENCH="python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --prewarm-s 0"

rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o kitti -- $BENCH > "$OUT/bench_trace.json" 2> "$OUT/trace.err"
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/pmc_fetch" -o kitti -- $BENCH > "$OUT/bench_fetch.json" 2> "$OUT/fetch.err"
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/pmc_write" -o kitti -- $BENCH > "$OUT/bench_write.json" 2> "$OUT/write.err"
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace \
  -d "$OUT/pmc_sq1" -o kitti -- $BENCH > "$OUT/bench_sq1.json" 2> "$OUT/sq1.err"
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE \
  --kernel-trace -d "$OUT/pmc_sq2" -o kitti -- $BENCH > "$OUT/bench_sq2.json" 2> "$OUT/sq2.err"
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --kernel-trace -d "$OUT/pmc_sq3" -o kitti -- $BENCH > "$OUT/bench_sq3.json" 2> "$OUT/sq3.err"

# kernel trace of the reference's own call-site configuration (640x480, nd 64, 21x21)
mkdir -p "$OUT/ref640t"
rocprofv3 --kernel-trace --stats -d "$OUT/ref640t/trace" -o ref640 -- $BENCH --workload ref640 > "$OUT/bench_ref640_trace.json" 2> "$OUT/ref640_trace.err"

# HBM traffic of the other single-GPU workloads (ref640 = the reference's own call-site configuration)
for wl in fhd uhd ref640; do
  WB="python3 $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --prewarm-s 0 --workload $wl"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/${wl}_pmc_fetch" -o $wl -- $WB > "$OUT/bench_${wl}_fetch.json" 2> "$OUT/${wl}_fetch.err"
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/${wl}_pmc_write" -o $wl -- $WB > "$OUT/bench_${wl}_write.json" 2> "$OUT/${wl}_write.err"
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace -d "$OUT/${wl}_pmc_sq" -o $wl -- $WB > "$OUT/bench_${wl}_sq.json" 2> "$OUT/${wl}_sq.err"
done

cd "$R"
# un-profiled bench lines for every workload, with the oracle check
python3 bench.py --check > "$OUT/bench_plain.json" 2> "$OUT/plain.err"
for wl in fhd ref640 uhd; do
  python3 bench.py --workload $wl --check > "$OUT/bench_$wl.json" 2>> "$OUT/plain.err"
done
python3 tools/rocprof_summary.py "$OUT" "$OUT/summary" > "$OUT/summary.txt" 2>&1
for wl in fhd uhd ref640; do   # same summariser on the per-workload counter passes (directories named <wl>_pmc_*)
  mkdir -p "$OUT/$wl"; for d in "$OUT"/${wl}_pmc_*; do ln -sfn "$d" "$OUT/$wl/pmc_$(basename $d | sed "s/${wl}_pmc_//")"; done
  python3 tools/rocprof_summary.py "$OUT/$wl" "$OUT/summary_$wl" >> "$OUT/summary.txt" 2>&1
done
python3 tools/rocprof_summary.py "$OUT/ref640t" "$OUT/summary_ref640t" >> "$OUT/summary.txt" 2>&1
tail -n 40 "$OUT/summary.txt"
cat "$OUT"/bench_plain.json
