#!/bin/bash
# row-of-taps filter gradient: parity tests, then the per-layer A/B against the per-tap kernel (one box)
R=${GRAFT_REPO_ROOT:-$PWD}; cd $R; mkdir -p gpurun_out
{ hostname; /opt/rocm/bin/rocm-smi --showuniqueid 2>/dev/null | grep -i "unique"; } > gpurun_out/w3_box.txt 2>&1
true
timeout 300 python scripts/wgrad_ab.py ${1:-w3} > gpurun_out/w3_ab.log 2>&1; cat gpurun_out/w3_ab.log
