#!/bin/bash
# GPU box: times SumThreshold kernel variant $1 (default 3) for every tricolour_amd/variants/lib_*.so
v=${1:-3}
for f in tricolour_amd/variants/lib_*.so; do
  for blk in ${BLKS:-256}; do
  echo -n "$(basename $f) blk=$blk: "
  TRI_ST_BLK=$blk TRICOLOUR_AMD_LIB=$PWD/$f TRI_BENCH_ST_VARIANT=$v python bench.py --bl 16 --steps 1 --warmup 0 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['achieved'], d['roofline']['ms_per_launch'])"
  done
done
