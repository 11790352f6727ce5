#!/bin/bash
# round 6: closing evidence run -- GPU test-suite, tools/profile_round.sh r06, one-pair traces, host entry points, the default bench line
O=gpurun_out/r06; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $O/gputests_final.txt; cat $O/gputests_final.txt
bash tools/profile_round.sh r06 > gpurun_out/profile_r06.log 2>&1; tail -2 gpurun_out/profile_r06.log | cut -c1-200
bash tools/exp/r06_trace.sh "one_kitti libsbm_hip.so kitti 1" "one_ref640 libsbm_hip.so ref640 1" "one_fhd libsbm_hip.so fhd 1" "kitti8 libsbm_hip.so kitti 8" > $O/one_pair_traces.txt 2>&1
rm -rf $O/trace/*/trace        # (the raw databases stay on the box)
python3 tools/bench_host.py > $O/host_entry.json 2> $O/host_entry.err; tail -c 1500 $O/host_entry.json
TAG=final LIB=libsbm_hip.so STEPS=60 bash tools/exp/r06_base.sh
python3 tools/bench_prefilter.py --cold --reps 20 > $O/pf_cold_final.json 2>/dev/null
python3 bench.py --check 2>/dev/null | tail -1 > $O/bench_attached.json; python3 -c "
import json; j=json.load(open('$O/bench_attached.json')); r=j['roofline']; print(j['ms_per_step'], r.get('traffic'), r.get('valu_busy_frac'), r.get('lane_ops_per_pixel_disparity'), r.get('traffic_reason'))"
