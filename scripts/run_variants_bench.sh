#!/bin/bash
# GPU box: whole-call bench (args after the first are passed to bench.py) for every tricolour_amd/variants/lib_*.so
for f in tricolour_amd/variants/lib_*.so; do
  echo -n "$(basename $f): "
  TRICOLOUR_AMD_LIB=$PWD/$f python bench.py --steps 1 --warmup 1 --no-roofline --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
