#!/bin/bash
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests -m gpu -x -q --timeout 180 --timeout-method thread > gpurun_out/six_tests.log 2>&1 || { tail -30 gpurun_out/six_tests.log; exit 1; }
tail -1 gpurun_out/six_tests.log
for round in 1 2; do
for n in cur dry; do
	STEPS=200 WARMUP=50 BENCH_ARGS="--no-secondary" profiles/ab.sh $n || exit 1
	STEPS=300 WARMUP=50 BENCH_ARGS="--no-secondary --workload demo-1080p-progressive" profiles/ab.sh $n || exit 1
	STEPS=100 WARMUP=20 BENCH_ARGS="--no-secondary --workload mixed9-1080p-64spp-d16" profiles/ab.sh $n || exit 1
done
done
profiles/ab2.sh cur dry
for n in cur dry; do
	echo $n
	PT_CORE_LIB=$PWD/pathtracer_amd/variant_$n.so timeout -k 10 120 python3 profiles/partition_time.py demo-1080p-64spp-d8 2>&1 | grep -v amdgpu
	PT_CORE_LIB=$PWD/pathtracer_amd/variant_$n.so timeout -k 10 120 python3 profiles/prog_parts.py 2>&1 | grep -v amdgpu
done
