#!/bin/bash
# GPU box: `bench.py <args>` once per tricolour_amd/variants/lib_*.so (value, ms per step)
for f in tricolour_amd/variants/lib_*.so; do
  echo -n "$(basename $f): "
  TRICOLOUR_AMD_LIB=$PWD/$f python bench.py --no-roofline --no-cpu-baseline --no-other-params "$@" 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"
done
