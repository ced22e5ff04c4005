#!/bin/bash
# GPU box: rocprofv3 kernel times of kernels matching $1 for every tricolour_amd/variants/lib_*.so (bench args after $1)
pat=$1; shift
for f in tricolour_amd/variants/lib_*.so; do
  tag=$(basename $f .so)
  TRICOLOUR_AMD_LIB=$PWD/$f bash scripts/prof.sh v_$tag --steps 1 --warmup 0 --no-cpu-baseline --no-roofline "$@" > /dev/null 2>&1
  echo "== $tag"
  python3 scripts/kernel_summary.py gpurun_out/prof_v_$tag | grep "$pat" | cut -c1-140
done
