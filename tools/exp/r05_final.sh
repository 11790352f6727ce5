#!/bin/bash
# round 5: final evidence run on the product build
O=gpurun_out/r05; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -6 > $O/gputests_final.txt; cat $O/gputests_final.txt
bash tools/profile_round.sh r05 > gpurun_out/profile_r05.log 2>&1; tail -3 gpurun_out/profile_r05.log | cut -c1-300
bash tools/exp/r05_envelope.sh > $O/envelope_final.txt 2>&1; grep -c "check True" $O/envelope_final.txt
python3 tools/bench_host.py 2>/dev/null | tail -1 > $O/host_entry_final.json; cut -c1-700 $O/host_entry_final.json
LIB=libsbm_hip.so bash tools/exp/r05_nseg_auto.sh > $O/nseg_auto_final.txt 2>&1; cat $O/nseg_auto_final.txt
