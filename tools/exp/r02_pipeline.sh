#!/bin/bash
# sub-batch pipeline sweep: SBM_SUBBATCH x SBM_SAD_STREAMS, KITTI b64 (bench default)
out=gpurun_out/r02_pipe; mkdir -p $out
for sub in 1 2 3 4 6 8; do for ss in 1 2; do
  [ $sub = 1 ] && [ $ss = 2 ] && continue
  SBM_SUBBATCH=$sub SBM_SAD_STREAMS=$ss python3 bench.py --no-cpu-baseline --steps 100 --warmup 10 > $out/sub${sub}_ss${ss}.json 2>$out/sub${sub}_ss${ss}.err
  python3 - <<PY
import json
j=json.load(open("$out/sub${sub}_ss${ss}.json"))
print("sub",$sub,"ss",$ss,"ms/step",j["ms_per_step"],j["roofline"]["stage_ms"])
PY
done; done
timeout 900 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -5
