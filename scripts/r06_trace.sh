#!/bin/bash
# device-clock timelines of the fused step (profiles/r06_fused_trace.md): one GPU at N = 10^6, one rank of 8 / 4 at N = 10^7
mkdir -p gpurun_out/r06_trace
export FDJAC_TEST_SWITCHES=1 FDJAC_FUSED_TRACE=1
python scripts/fused_trace.py 1000000 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_trace/one_gpu_1e6.txt
python scripts/fused_trace_shard.py 10000000 8 4 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_trace/rank4of8_1e7.txt
python scripts/fused_trace_shard.py 10000000 4 1 2>&1 | grep -v amdgpu.ids > gpurun_out/r06_trace/rank1of4_1e7.txt
tail -n +1 gpurun_out/r06_trace/*.txt
