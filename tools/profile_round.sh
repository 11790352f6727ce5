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

# BASELINE.json's own batch sizes for configs[2] (1080p nd 256, 64 pairs on one GPU: "the HBM-roofline run") and for the per-GPU
# share of configs[4] (2160p, 256 pairs over 8 GPUs = 32): kernel trace + traffic + vector / LDS busy (VERDICT r04 missing 5)
for spec in "fhd 64" "uhd 32"; do
  set -- $spec; wl=$1; np=$2; tag=${wl}${np}
  WB="python3 $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --prewarm-s 0 --workload $wl --pairs $np"
  mkdir -p "$OUT/$tag"
  rocprofv3 --kernel-trace --stats -d "$OUT/$tag/trace" -o $tag -- $WB > "$OUT/bench_${tag}_trace.json" 2> "$OUT/${tag}_trace.err"
  rocprofv3 --pmc FETCH_SIZE --kernel-trace -d "$OUT/$tag/pmc_fetch" -o $tag -- $WB > "$OUT/bench_${tag}_fetch.json" 2> "$OUT/${tag}_fetch.err"
  rocprofv3 --pmc WRITE_SIZE --kernel-trace -d "$OUT/$tag/pmc_write" -o $tag -- $WB > "$OUT/bench_${tag}_write.json" 2> "$OUT/${tag}_write.err"
  rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace -d "$OUT/$tag/pmc_sq" -o $tag -- $WB > "$OUT/bench_${tag}_sq.json" 2> "$OUT/${tag}_sq.err"
done
# the reference's own PL blocks and the front-end kernels (tools/bench_frontend.py): kernel trace + vector / LDS busy of the
# FPGA-flavour matcher, the GFTT map, the rectifier (VERDICT r04 item 8 asks for the matcher's counters before any rework)
mkdir -p "$OUT/frontend"
FB="python3 $R/tools/bench_frontend.py --pairs 64 --reps 10"
rocprofv3 --kernel-trace --stats -d "$OUT/frontend/trace" -o fe -- $FB > "$OUT/bench_frontend_trace.json" 2> "$OUT/frontend_trace.err"
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace -d "$OUT/frontend/pmc_sq" -o fe -- $FB > "$OUT/bench_frontend_sq.json" 2> "$OUT/frontend_sq.err"
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace -d "$OUT/frontend/pmc_sq2" -o fe -- $FB > "$OUT/bench_frontend_sq2.json" 2> "$OUT/frontend_sq2.err"

cd "$R"
python3 bench.py --workload fhd --pairs 64 --check --cpu-sample 2 > "$OUT/bench_fhd64.json" 2>> "$OUT/plain.err"
python3 bench.py --workload uhd --pairs 32 --check --cpu-sample 1 > "$OUT/bench_uhd32.json" 2>> "$OUT/plain.err"
python3 tools/bench_frontend.py > "$OUT/frontend_kernels.json" 2>> "$OUT/plain.err"
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
for tag in fhd64 uhd32 frontend; do python3 tools/rocprof_summary.py "$OUT/$tag" "$OUT/summary_$tag" >> "$OUT/summary.txt" 2>&1; done
tail -n 40 "$OUT/summary.txt"
cat "$OUT"/bench_plain.json
