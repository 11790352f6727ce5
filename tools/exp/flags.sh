#!/bin/bash
# try scheduler flags on the SAD kernel object only; prints VGPRs of the bench variant
cd /root/repo
for fl in "" "-mllvm -amdgpu-enable-max-ilp-scheduling-strategy" "-mllvm -amdgpu-schedule-relaxed-occupancy" "-mllvm -amdgpu-disable-unclustered-high-rp-reschedule" "-mllvm -enable-post-misched=0"; do
  out=/tmp/isa/fl.s
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 $fl -S --cuda-device-only -Iinclude -Iu96-slam_amd/csrc u96-slam_amd/csrc/sbm_sad_fast.hip -o $out 2>/tmp/isa/fl.err || { echo "flag [$fl]: compile failed: $(head -1 /tmp/isa/fl.err)"; continue; }
  python3 - "$fl" <<'PY'
import re,sys
s=open('/tmp/isa/fl.s').read()
for m in re.finditer(r'\.name:\s+(_ZN3sbm15sad_fast_kernelILi64ELi2ELi5ELi3ELb1\S+)', s):
    blk=s[m.start():m.start()+3000]
    print("flag [%s]: vgpr %s spill %s" % (sys.argv[1], re.search(r'\.vgpr_count:\s+(\d+)',blk).group(1), re.search(r'\.vgpr_spill_count:\s+(\d+)',blk).group(1)))
PY
done
