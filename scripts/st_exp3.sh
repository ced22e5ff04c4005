#!/bin/bash
# compiler scheduling strategies for the fused SumThreshold kernel
for flags in "-mllvm -amdgpu-sched-strategy=max-ilp" "-mllvm -amdgpu-sched-strategy=max-memory-clause" "-mllvm -amdgpu-use-amdgpu-trackers=1" ""; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -shared -std=c++17 -ffp-contract=off -fno-fast-math -fhip-fp32-correctly-rounded-divide-sqrt -mllvm -amdgpu-sched-strategy=max-ilp $flags -o tricolour_amd/libtricolour_amd.so tricolour_amd/csrc/tricolour_amd.hip 2>&1 | grep -i "error\|unknown" | head -2
  echo -n "[$flags]: "
  python bench.py --bl 16 --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['roofline']['achieved'], d['roofline']['ms_per_launch'], d['value'])"
done
