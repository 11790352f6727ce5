#!/bin/bash
# round 5: vector / LDS busy counters and kernel trace of the FPGA-flavour matcher and the other front-end kernels
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r05/frontend; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
FB="python3 $R/tools/bench_frontend.py --pairs 64 --reps 10"
rocprofv3 --kernel-trace --stats -d "$OUT/trace" -o fe -- $FB > "$OUT/bench_frontend_trace.json" 2> "$OUT/frontend_trace.err"
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS GRBM_GUI_ACTIVE --kernel-trace -d "$OUT/pmc_sq" -o fe -- $FB > "$OUT/bench_frontend_sq.json" 2> "$OUT/frontend_sq.err"
rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_ANY SQ_BUSY_CYCLES --kernel-trace -d "$OUT/pmc_sq2" -o fe -- $FB > "$OUT/bench_frontend_sq2.json" 2> "$OUT/frontend_sq2.err"
rocprofv3 --pmc SQ_INSTS_SALU SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY --kernel-trace -d "$OUT/pmc_sq3" -o fe -- $FB > "$OUT/bench_frontend_sq3.json" 2> "$OUT/frontend_sq3.err"
cd $R
python3 tools/rocprof_summary.py "$OUT" "$OUT/summary" > "$OUT/summary.txt" 2>&1
tail -40 "$OUT/summary.txt"
timeout 600 python -m pytest tests/test_gpu_multi.py -m gpu -q 2>&1 | tail -5
