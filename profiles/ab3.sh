#!/bin/bash
# A/B of library variants on the three frames that matter: demo (brute force), glass (1080p / 16 spp) and 100 000 entities
# (1080p / 128 spp: work items as short as at full size; at 16 spp one item over the densest cluster is half the frame)
# usage: profiles/ab3.sh name ...
for n in "$@"; do
	STEPS=200 WARMUP=60 BENCH_ARGS="--no-secondary" profiles/ab.sh $n || exit 1
	STEPS=30 WARMUP=5 BENCH_ARGS="--no-secondary --workload glass-1080p-16spp-d32" profiles/ab.sh $n || exit 1
	STEPS=6 WARMUP=2 BENCH_ARGS="--no-secondary --workload dense100k-1920x1080-128spp-d8" profiles/ab.sh $n || exit 1
done
