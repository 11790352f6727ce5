#!/bin/bash
# round 6: per-kernel durations (rocprofv3 --kernel-trace --stats) of builds / environment variants. usage: bash tools/exp/r06_trace.sh "<tag> <lib> <workload> <pairs> [ENV=..]" ...
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r06/trace; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for spec in "$@"; do
  set -- $spec; tag=$1; lib=$2; wl=$3; np=$4; shift 4
  for kv in "$@"; do export "$kv"; done
  export SBM_LIB_AB=$lib
  rocprofv3 --kernel-trace --stats -d "$O/$tag/trace" -o t -- python3 $R/bench.py --workload $wl --pairs $np --no-cpu-baseline --steps 40 --warmup 5 --prewarm-s 0 > "$O/bench_$tag.json" 2> "$O/$tag.err"
  for kv in "$@"; do unset "${kv%%=*}"; done
  python3 $R/tools/rocprof_summary.py "$O/$tag" "$O/summary_$tag" > "$O/summary_$tag.txt" 2>&1
  echo "== $tag ($lib $wl x$np $*)"; head -16 "$O/summary_${tag}_kernels.md" | cut -c1-150
done
