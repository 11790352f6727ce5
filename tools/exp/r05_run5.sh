#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
for L in libsbm_hip_p19a.so libsbm_hip_p19b.so libsbm_hip_p19c.so libsbm_hip_p19a.so libsbm_hip_p19b.so libsbm_hip_p19c.so; do
  echo "== $L"; SBM_LIB_AB=$L NDS="128 64" WS="19 23" bash tools/exp/r05_envelope.sh
done > $O/plan19.txt 2>&1; cat $O/plan19.txt
for z in 1 0 1 0; do echo "== SBM_HOST_ZEROCOPY=$z"; SBM_HOST_ZEROCOPY=$z SBM_LIB_AB=libsbm_hip_devapi.so python3 tools/exp/r05_host_attrib.py 2>&1 | grep -v amdgpu.ids; done > $O/host_attrib2.txt; cat $O/host_attrib2.txt
for sp in 1 0; do for nd in 256 224 160 144; do
  SBM_FAST_SPLIT=$sp SBM_LIB_AB=libsbm_hip_dev_all.so python3 bench.py --workload fhd --ndisp $nd --pairs 1 --check --cpu-sample 1 --steps 40 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('split=$sp fhd n=1 nd $nd', d['ms_per_step'], d['roofline']['stage_ms']['sad'], d['cpu_baseline'].get('bit_exact_vs_gpu'), d['roofline'].get('kernel'))"
done; done > $O/split5.txt 2>&1; cat $O/split5.txt
