#!/bin/bash
# events and phase split of the cooperative cast on the two grid frames (diagnostic builds variant_stats / variant_timing)
mkdir -p gpurun_out/r03_coop
for w in glass-1080p-16spp-d32 dense100k-1920x1080-128spp-d8; do
	PT_CORE_LIB=$PWD/pathtracer_amd/variant_stats.so timeout -k 10 120 python3 profiles/coop_stats.py $w > gpurun_out/r03_coop/events_$w.txt 2>&1 || { cat gpurun_out/r03_coop/events_$w.txt; exit 1; }
	PT_CORE_LIB=$PWD/pathtracer_amd/variant_timing.so timeout -k 10 120 python3 profiles/coop_stats.py $w > gpurun_out/r03_coop/phases_$w.txt 2>&1 || { cat gpurun_out/r03_coop/phases_$w.txt; exit 1; }
done
cat gpurun_out/r03_coop/*glass*
