#!/bin/bash
# usage: scripts/build_variants.sh tag1:"-DA=1 -DB=2" tag2:"..."   (run from the repo root, CPU container)
# Builds tricolour_amd/variants/lib_<tag>.so with the library's flags plus the given defines,
# four at a time; load one with TRICOLOUR_AMD_LIB=<path>.
mkdir -p tricolour_amd/variants
FLAGS="--offload-arch=gfx950 -O3 -fPIC -shared -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt"
n=0
for spec in "$@"; do
  tag=${spec%%:*}; defs=${spec#*:}
  sched="-mllvm -amdgpu-sched-strategy=max-ilp"
  case "$defs" in *NOMAXILP*) sched="";; esac
  /opt/rocm/bin/hipcc $FLAGS $sched $defs -o tricolour_amd/variants/lib_$tag.so tricolour_amd/csrc/tricolour_amd.hip 2>/dev/null &
  n=$((n+1)); if [ $((n % 4)) -eq 0 ]; then wait; fi
done
wait
ls -la tricolour_amd/variants/
