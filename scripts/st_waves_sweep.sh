#!/bin/bash
# rebuilds the extension with different occupancy targets for the fused SumThreshold
# kernel and prints its measured bandwidth (run on the GPU box)
for wv in 2 3 4; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -shared -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -mllvm -amdgpu-sched-strategy=max-ilp -DST_WAVES=$wv -o tricolour_amd/libtricolour_amd.so tricolour_amd/csrc/tricolour_amd.hip 2>/dev/null
  echo -n "ST_WAVES=$wv: "
  python bench.py --bl 4 --steps 1 --warmup 0 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['achieved'], d['roofline']['ms_per_launch'])"
done
