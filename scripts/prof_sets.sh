# GPU box: per-kernel summaries of one step of the three recorded parameter sets (rocprofv3 --kernel-trace), all rows
set -euo pipefail
cd $GRAFT_REPO_ROOT
for p in stage1 defaults very_broad; do
  bash scripts/prof.sh c_$p --steps 1 --warmup 1 --no-cpu-baseline --no-roofline --no-other-params --no-scatter --params $p --no-parity-check --no-other-workloads
  cd $GRAFT_REPO_ROOT
  python scripts/kernel_summary.py gpurun_out/prof_c_$p 80 > gpurun_out/c_${p}_kernel_summary.txt
  rm -rf gpurun_out/prof_c_$p
  head -2 gpurun_out/c_${p}_kernel_summary.txt
done
