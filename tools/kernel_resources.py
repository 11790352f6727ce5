#!/usr/bin/env python3
"""VGPR / scratch / LDS of the kernels in a hipcc -S listing (amdhsa metadata), optionally filtered by a regex on the name.

usage: hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -Iinclude -o x.s file.hip; tools/kernel_resources.py x.s [regex]
"""
import re
import subprocess
import sys


def main():
    s = open(sys.argv[1]).read()
    pat = re.compile(sys.argv[2]) if len(sys.argv) > 2 else None
    md = s[s.index("amdhsa.kernels"):]
    for b in md.split("  - .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", b).group(1)
        try:
            dem = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-cxxfilt", name], capture_output=True, text=True).stdout.strip()
        except OSError:
            dem = name
        if pat and not pat.search(dem):
            continue
        g = lambda k: re.search(r"\.%s:\s+(\d+)" % k, b).group(1)
        print(f"{dem[:90]:90s} vgpr {g('vgpr_count'):>3s} sgpr {g('sgpr_count'):>3s} scratch {g('private_segment_fixed_size'):>4s}")


if __name__ == "__main__":
    main()
