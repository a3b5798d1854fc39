#!/bin/bash
# Round-3 evidence in one GPU session: the library compiled from source on the GPU box, then kernel traces and counter passes
# of the three frames.  usage: profiles/r03_final.sh   (from the repository root)
mkdir -p gpurun_out
( timeout -k 10 400 python3 -c "import __graft_entry__ as g; g.build_core(force=True)" && rocminfo | grep -m1 gfx9 ) > gpurun_out/r03_build_on_box.txt 2>&1 || { cat gpurun_out/r03_build_on_box.txt; exit 1; }
cat gpurun_out/r03_build_on_box.txt
profiles/run_profiles.sh r03_c2 > gpurun_out/r03_c2.log 2>&1 || { tail gpurun_out/r03_c2.log; exit 1; }
profiles/run_profiles.sh r03_glass --workload glass-1080p-128spp-d32 > gpurun_out/r03_glass.log 2>&1 || { tail gpurun_out/r03_glass.log; exit 1; }
TRACE_STEPS=20 TRACE_WARMUP=4 profiles/run_profiles.sh r03_dense100k --workload dense100k-4k-256spp-d8 > gpurun_out/r03_dense.log 2>&1 || { tail gpurun_out/r03_dense.log; exit 1; }
echo done
