#!/bin/bash
out=gpurun_out/r02_gaps; mkdir -p $out
run() { python3 bench.py --no-cpu-baseline "$@" | python3 -c "import json,sys;j=json.loads(sys.stdin.read());print('$TAG', j['ms_per_step'],j['roofline']['stage_ms'])"; }
TAG="prof2 side1" run
TAG="noprof side1" run --no-profile
export SBM_SIDE=0
TAG="prof2 side0" run
TAG="noprof side0" run --no-profile
unset SBM_SIDE
TAG="noprof side1 b128" run --no-profile --pairs 128
TAG="prof2 side1 b128" run --pairs 128
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; cd $R
rocprofv3 --kernel-trace --output-format csv -d $out/t1 -- python3 bench.py --no-cpu-baseline --no-profile --steps 3 --warmup 2 > $out/t1.json 2>$out/t1.err
f=$(find $out/t1 -name '*kernel_trace.csv' | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
t0=None
for r in rows[-18:]:
    s,e=int(r["Start_Timestamp"]),int(r["End_Timestamp"])
    if t0 is None: t0=s
    print(f'{(s-t0)/1e3:9.1f} {(e-t0)/1e3:9.1f} q{r.get("Queue_Id","?")} {r["Kernel_Name"][:50]}')
PY
