#!/bin/bash
# round 5, gpurun call 3: plan + dword staging everywhere (dev), LDS-direct staging for every layout (dev_all)
O=gpurun_out/r05; mkdir -p $O
for L in libsbm_hip_dev.so libsbm_hip_dev_all.so; do
  echo "== $L"; SBM_LIB_AB=$L python3 tools/exp/r04_quick.py 320,96,64,21,5 640,120,256,21,2 300,70,96,21,17 1242,100,128,15,70 420,80,112,15,72 900,100,64,15,40 \
    1000,90,160,21,12 333,77,48,11,7 300,70,96,27,17 400,90,128,19,9 400,90,128,25,40 500,80,64,13,30 500,80,32,7,30 700,90,256,17,20 640,100,192,23,9 640,100,64,5,1 800,100,192,21,12 640,480,64,21,1 2>&1 | grep -v amdgpu.ids
done > $O/quick3.txt 2>&1
cat $O/quick3.txt
LIBS="libsbm_hip.so libsbm_hip_dev.so libsbm_hip_dev_all.so" bash tools/exp/r05_ab.sh > $O/ab3.txt 2>&1; cat $O/ab3.txt
for L in libsbm_hip.so libsbm_hip_dev.so libsbm_hip_dev_all.so; do
  for wl in ref640 kitti fhd; do
    SBM_LIB_AB=$L python3 bench.py --workload $wl --pairs 1 --no-cpu-baseline --steps 50 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L $wl n=1', d['ms_per_step'], d['roofline']['stage_ms']['sad'], d['roofline'].get('kernel'))"
  done
  for nd in 192 160; do
    SBM_LIB_AB=$L python3 bench.py --workload fhd --ndisp $nd --check --cpu-sample 2 --steps 20 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$L fhd nd $nd', d['ms_per_step'], d['roofline']['stage_ms']['sad'], d['cpu_baseline'].get('bit_exact_vs_gpu'), d['roofline'].get('kernel'))"
  done
done > $O/small3.txt 2>&1; cat $O/small3.txt
SBM_LIB_AB=libsbm_hip_dev.so bash tools/exp/r05_envelope.sh > $O/envelope_dev.txt 2>&1; cat $O/envelope_dev.txt
SBM_LIB_AB=libsbm_hip_devapi.so python3 tools/exp/r05_host_attrib.py 2>&1 | grep -v amdgpu.ids > $O/host_attrib.txt; cat $O/host_attrib.txt
