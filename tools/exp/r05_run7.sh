#!/bin/bash
O=gpurun_out/r05; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -25 > $O/gputests7.txt; cat $O/gputests7.txt
for z in 1 0 1 0; do echo "== SBM_HOST_ZEROCOPY=$z"; SBM_HOST_ZEROCOPY=$z python3 tools/bench_host.py 2>/dev/null | tail -1 | cut -c1-330; SBM_HOST_ZEROCOPY=$z python3 tools/exp/r05_host_attrib.py 2>&1 | grep -v amdgpu.ids; done > $O/host7.txt 2>&1; cat $O/host7.txt
