#!/bin/bash
# A/B of library variants on the three frames that matter: demo (brute force), glass and 100 000 entities (cooperative cast, 1080p / 16 spp)
# usage: profiles/ab3.sh name ...
for n in "$@"; do
	STEPS=200 WARMUP=60 BENCH_ARGS="--no-secondary" profiles/ab.sh $n || exit 1
	STEPS=30 WARMUP=5 BENCH_ARGS="--no-secondary --workload glass-1080p-16spp-d32" profiles/ab.sh $n || exit 1
	STEPS=20 WARMUP=4 BENCH_ARGS="--no-secondary --workload dense100k-1080p-16spp-d8" profiles/ab.sh $n || exit 1
done
