#!/bin/bash
# A/B of library variants on the two grid frames: glass (1080p / 16 spp) and 100 000 entities (1080p / 128 spp), twice each
# usage: profiles/ab2.sh name ...
for round in 1 2; do
for n in "$@"; do
	STEPS=30 WARMUP=5 BENCH_ARGS="--no-secondary --workload glass-1080p-16spp-d32" profiles/ab.sh $n || exit 1
	STEPS=6 WARMUP=2 BENCH_ARGS="--no-secondary --workload dense100k-1920x1080-128spp-d8" profiles/ab.sh $n || exit 1
done
done
