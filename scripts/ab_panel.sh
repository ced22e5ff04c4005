# GPU box: the time-axis SumThreshold on column panels against plain rows (TRI_ST_NO_PANEL=1), one stage-1 step each under rocprofv3
set -euo pipefail
cd $GRAFT_REPO_ROOT
A="--steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-other-params --no-scatter --params stage1 --no-parity-check --no-other-workloads"
bash scripts/prof.sh pan_on $A
cd $GRAFT_REPO_ROOT
python scripts/kernel_summary.py gpurun_out/prof_pan_on 60 > gpurun_out/pan_on_kernel_summary.txt; rm -rf gpurun_out/prof_pan_on
export TRI_ST_NO_PANEL=1
bash scripts/prof.sh pan_off $A
cd $GRAFT_REPO_ROOT
python scripts/kernel_summary.py gpurun_out/prof_pan_off 60 > gpurun_out/pan_off_kernel_summary.txt; rm -rf gpurun_out/prof_pan_off
(echo "# column panels (default)"; head -2 gpurun_out/pan_on_kernel_summary.txt; grep -E "k_colst_mask|k_transpose<float|k_median_wave<8|k_combine_dilate16|k_unpanel|k_panel" gpurun_out/pan_on_kernel_summary.txt; echo "# TRI_ST_NO_PANEL=1 (plain rows)"; head -2 gpurun_out/pan_off_kernel_summary.txt; grep -E "k_colst_mask|k_transpose<float|k_median_wave<8|k_combine_dilate16" gpurun_out/pan_off_kernel_summary.txt) | cut -c1-150 > gpurun_out/panel_ab.txt
cat gpurun_out/panel_ab.txt
