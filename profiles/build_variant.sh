#!/bin/bash
# builds pathtracer_amd/variant_<name>.so with extra compiler flags (A/B timing of kernel variants)
# usage: profiles/build_variant.sh <name> [extra hipcc flags...]
name=$1; shift
# the brute-force variants of the trace kernel are a unit of their own, optimised for size (as in __graft_entry__.build_core)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize "$@" ${BRUTE_OPT:--Os} -DPT_UNIT_BRUTE \
	-Iinclude -Ipathtracer_amd/csrc -c pathtracer_amd/csrc/pt_kernels.hip -o /tmp/variant_${name}_brute.o -Rpass-analysis=kernel-resource-usage > /tmp/variant_${name}_brute.log 2>&1 || { grep -E "error" /tmp/variant_${name}_brute.log | head -5; exit 1; }
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-slp-vectorize "$@" -DPT_SPLIT_BRUTE \
	-Iinclude -Ipathtracer_amd/csrc /tmp/variant_${name}_brute.o pathtracer_amd/csrc/pt_api.cpp pathtracer_amd/csrc/pt_host.cpp pathtracer_amd/csrc/pt_scene.cpp pathtracer_amd/csrc/pt_kernels.hip pathtracer_amd/csrc/pt_gridbuild.hip \
	-o pathtracer_amd/variant_$name.so -Rpass-analysis=kernel-resource-usage > /tmp/variant_$name.log 2>&1
rc=$?
grep -E "error" /tmp/variant_$name.log | head -5
python3 - "$name" <<'PY'
import re, sys
log = open(f"/tmp/variant_{sys.argv[1]}_brute.log").read() + open(f"/tmp/variant_{sys.argv[1]}.log").read()
for m in re.finditer(r"Function Name: (\S+).*?VGPRs: (\d+).*?ScratchSize \[bytes/lane\]: (\d+).*?Occupancy \[waves/SIMD\]: (\d+).*?SGPRs Spill: (\d+).*?VGPRs Spill: (\d+)", log, re.S):
    if "trace" in m.group(1):
        print(f"  {sys.argv[1]:12s} {m.group(1)[-28:]:28s} vgpr {m.group(2)} scratch {m.group(3)} occ {m.group(4)} sgpr-spill {m.group(5)} vgpr-spill {m.group(6)}")
PY
exit $rc
