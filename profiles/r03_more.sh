#!/bin/bash
# Round-3 figures beside the profiles: one partition's kernel time for 1 / 2 / 4 / 8 partitions (demo and 100 000-entity frames),
# progressive latency, one-shot render with host pointers, the harness's progressive mode, and the default bench line
mkdir -p gpurun_out
timeout -k 10 120 python3 profiles/partition_time.py demo-1080p-64spp-d8 > gpurun_out/r03_partition_demo.txt 2>&1 || { cat gpurun_out/r03_partition_demo.txt; exit 1; }
timeout -k 10 300 python3 profiles/partition_time.py dense100k-4k-256spp-d8 > gpurun_out/r03_partition_dense.txt 2>&1 || { cat gpurun_out/r03_partition_dense.txt; exit 1; }
timeout -k 10 120 python3 profiles/progressive_latency.py > gpurun_out/r03_progressive_latency.json 2>gpurun_out/r03_progressive_latency.err || { cat gpurun_out/r03_progressive_latency.err; exit 1; }
timeout -k 10 120 python3 profiles/single_shot.py > gpurun_out/r03_single_shot.json 2>gpurun_out/r03_single_shot.err || { cat gpurun_out/r03_single_shot.err; exit 1; }
timeout -k 10 120 harness/pt_headless --progressive 256 --gpus 0,0,0 > gpurun_out/r03_headless_progressive.json 2>&1 || { cat gpurun_out/r03_headless_progressive.json; exit 1; }
timeout -k 10 120 harness/pt_headless --bounce-histogram > gpurun_out/r03_headless_histogram.json 2>&1 || { cat gpurun_out/r03_headless_histogram.json; exit 1; }
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r03_bench.json 2>gpurun_out/r03_bench.err || { tail gpurun_out/r03_bench.err; exit 1; }
cat gpurun_out/r03_partition_demo.txt gpurun_out/r03_partition_dense.txt gpurun_out/r03_progressive_latency.json gpurun_out/r03_single_shot.json gpurun_out/r03_headless_progressive.json
