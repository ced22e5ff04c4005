# GPU box: the wave medians with several rows of a segment per wave against one segment per wave (TRI_MEDIAN_WAVE_OLD=1) -- tests first, then one
# stage-1 kernel summary per route.  Outputs under gpurun_out/.
set -euo pipefail
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py -x -q -k "median_kernels or median_wave or golden or edge_cases or random_cases or large_windows or alternate_kernel" > gpurun_out/t_medw.log 2>&1
tail -3 gpurun_out/t_medw.log
A="--steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-other-params --no-scatter --params stage1 --no-parity-check --no-other-workloads"
bash scripts/prof.sh medw_new $A
cd $GRAFT_REPO_ROOT
python scripts/kernel_summary.py gpurun_out/prof_medw_new > gpurun_out/medw_new_kernel_summary.txt; rm -rf gpurun_out/prof_medw_new
export TRI_MEDIAN_WAVE_OLD=1
bash scripts/prof.sh medw_old $A
cd $GRAFT_REPO_ROOT
python scripts/kernel_summary.py gpurun_out/prof_medw_old > gpurun_out/medw_old_kernel_summary.txt; rm -rf gpurun_out/prof_medw_old
grep -h k_median_wave gpurun_out/medw_new_kernel_summary.txt gpurun_out/medw_old_kernel_summary.txt | cut -c1-160
