#!/bin/bash
# round 5: kernel durations of one-pair calls (how much of each stage is the kernel, how much the dependent-launch gap)
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r05/onepair; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for wl in ref640 kitti; do
  rocprofv3 --kernel-trace --stats -d "$OUT/$wl/trace" -o t -- python3 $R/bench.py --workload $wl --pairs 1 --no-cpu-baseline --steps 200 --warmup 5 --prewarm-s 0 > "$OUT/bench_$wl.json" 2> "$OUT/$wl.err"
  python3 $R/tools/rocprof_summary.py "$OUT/$wl" "$OUT/summary_$wl" > "$OUT/summary_$wl.txt" 2>&1
  head -14 "$OUT/summary_${wl}_kernels.md"; tail -1 "$OUT/bench_$wl.json" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl', d['ms_per_step'], d['roofline']['stage_ms'])"
done
cd $R; LIB=libsbm_hip.so bash tools/exp/r05_nseg_auto.sh 2>&1 | grep -E "n=(1|2|64) " 
