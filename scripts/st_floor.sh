#!/bin/bash
# Memory floor of the SumThreshold kernel in the step's REAL launch geometry (1008 windows x 1024 x 4096):
# timing-only builds that keep every load, prefix sum and store but drop the clamps and / or the hits of
# stages 1-3 (results wrong on purpose).  Run scripts/build_variants.sh first (CPU container):
#   bash scripts/build_variants.sh full:"-DX=0" noclamp:"-DST_ABLATE_CLAMP" nohits:"-DST_ABLATE_HITS" neither:"-DST_ABLATE_CLAMP -DST_ABLATE_HITS"
for v in full noclamp nohits neither; do
  echo -n "$v: "
  TRICOLOUR_AMD_LIB=$PWD/tricolour_amd/variants/lib_$v.so python bench.py --roofline-only 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read())['roofline'][0]; print(d['ms_per_launch'], 'ms', d['achieved'], 'GB/s', d['frac'])"
done
