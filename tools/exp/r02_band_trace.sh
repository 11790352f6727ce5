#!/bin/bash
# full GPU suite, then per-kernel durations of the default speckle path (rocprofv3 kernel trace)
timeout 1500 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/bandt -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 20 --warmup 3 > /dev/null 2>&1
