#!/bin/bash
# round 6: regenerate the pin kit (v2: alternative readings) on the GPU box and bring it home
python3 tools/pin_kit.py --out gpurun_out/pin_kit.npz 2>&1 | tail -32
ls -la gpurun_out/pin_kit.*
