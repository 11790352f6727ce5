#!/bin/bash
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/bandpmc -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 10 --warmup 2 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/bandtr -o t -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 20 --warmup 3 > /dev/null 2>&1
