#!/bin/bash
O=gpurun_out/r05; mkdir -p $O   # round 5: closing evidence run (GPU tests, profile round, one bench line)
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -4 > $O/gputests_final.txt; cat $O/gputests_final.txt
bash tools/profile_round.sh r05 > gpurun_out/profile_r05.log 2>&1; tail -2 gpurun_out/profile_r05.log | cut -c1-200
python3 bench.py --check 2>/dev/null | tail -1 > $O/bench_attached.json; python3 -c "
import json; j=json.load(open('$O/bench_attached.json')); r=j['roofline']; print(j['ms_per_step'], r.get('traffic'), r.get('valu_busy_frac'), r.get('lane_ops_per_pixel_disparity'), r.get('traffic_reason'))"
