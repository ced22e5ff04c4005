# GPU box: every kernel of one stage-1 step (no row limit), and the span from the first start to the last end
set -euo pipefail
cd $GRAFT_REPO_ROOT
bash scripts/prof.sh full1 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-other-params --no-scatter --params stage1 --no-parity-check --no-other-workloads
cd $GRAFT_REPO_ROOT
python scripts/kernel_summary.py gpurun_out/prof_full1 200 > gpurun_out/full_stage1_kernel_summary.txt
rm -rf gpurun_out/prof_full1
tail -n +1 gpurun_out/full_stage1_kernel_summary.txt | cut -c1-150 | sed -n 1,3p
