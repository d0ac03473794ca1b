#!/bin/bash
mkdir -p gpurun_out/r06_held
timeout 900 python -m pytest tests/test_gpu_fused.py -x -q -k "held" 2>&1 | tail -3
bash scripts/r06_held_ab.sh "$@" 2>&1 | grep -v amdgpu.ids
FDJAC_TEST_SWITCHES=1 FDJAC_FUSED_TRACE=1 timeout 300 python scripts/fused_trace.py 10000000 2>&1 | grep -v amdgpu.ids | grep -v "^finisher\|^eps_published"
