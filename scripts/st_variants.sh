#!/bin/bash
# Times the SumThreshold kernel variants (2 = register cascade, 3 = lane-mask cascade)
# on the roofline geometry of bench.py (--bl 16: 64 windows x 1024 x 4096).
for v in 2 3; do
  echo -n "variant=$v: "
  TRI_BENCH_ST_VARIANT=$v python bench.py --bl 16 --steps 1 --warmup 0 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['achieved'], d['roofline']['ms_per_launch'])"
done
