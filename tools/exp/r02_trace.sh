#!/bin/bash
# kernel timeline of the sub-batch pipeline: do the post-filter kernels overlap the next SAD launch?
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; out=$R/gpurun_out/r02_trace; mkdir -p $out
cd $R
for cfg in "2 1 4" "2 1 8" "4 1 8"; do set -- $cfg
  export SBM_SUBBATCH=$1 SBM_SAD_STREAMS=$2 GPU_MAX_HW_QUEUES=$3
  rocprofv3 --kernel-trace --output-format csv -d $out/t_$1_$2_$3 -- python3 bench.py --no-cpu-baseline --steps 3 --warmup 2 > $out/t_$1_$2_$3.json 2>$out/t_$1_$2_$3.err
  f=$(find $out/t_$1_$2_$3 -name '*kernel_trace.csv' | head -1)
  python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last step only: last 40 kernels
t0=None
for r in rows[-22:]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    if t0 is None: t0=s
    print(f'{(s-t0)/1e3:9.1f} {(e-t0)/1e3:9.1f} q{r.get("Queue_Id","?")} {r["Kernel_Name"][:50]}')
PY
  python3 -c "import json;j=json.load(open('$out/t_$1_$2_$3.json'));print('cfg $cfg ms/step',j['ms_per_step'])"
done
for q in 4 8; do
GPU_MAX_HW_QUEUES=$q SBM_SUBBATCH=2 python3 bench.py --no-cpu-baseline | python3 -c "import json,sys;j=json.loads(sys.stdin.read());print('hwq $q sub2 ms/step',j['ms_per_step'],j['roofline']['stage_ms'])"
done
