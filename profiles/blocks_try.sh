#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/six_tests.log 2>&1 || { tail -30 gpurun_out/six_tests.log; exit 1; }
tail -1 gpurun_out/six_tests.log
timeout -k 10 120 python3 profiles/prog_parts.py 2>&1 | grep -v amdgpu
PT_GRID_BLOCKS=1280 timeout -k 10 120 python3 profiles/prog_parts.py 2>&1 | grep -v amdgpu
timeout -k 10 120 python3 profiles/partition_time.py demo-1080p-64spp-d8 2>&1 | grep -v amdgpu
timeout -k 10 200 python3 profiles/partition_time.py dense100k-4k-256spp-d8 2>&1 | grep -v amdgpu
cp pathtracer_amd/libptcore.so pathtracer_amd/variant_cur.so
profiles/ab2.sh cur
