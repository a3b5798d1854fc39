#!/bin/bash
# A/B of one environment switch of the library on the two grid frames (glass 1080p / 16 spp, 100 000 entities 1080p / 128 spp), twice each
# usage: profiles/ab_env.sh NAME=value
for round in 1 2; do
for setting in "" "$1"; do
	echo "== ${setting:-default}"
	env $setting STEPS=30 WARMUP=5 BENCH_ARGS="--no-secondary --workload glass-1080p-16spp-d32" profiles/ab.sh now || exit 1
	env $setting STEPS=6 WARMUP=2 BENCH_ARGS="--no-secondary --workload dense100k-1920x1080-128spp-d8" profiles/ab.sh now || exit 1
done
done
