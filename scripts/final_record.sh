# GPU box: the closing record of a round after the last source change -- full GPU test suite, the default bench line, the PMC
# traffic files keyed on this build (profiles/r04_pmc_*.json) and the two-rank gloo rehearsal of the streamed scatter leg.
# Outputs under gpurun_out/.
set -euo pipefail
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/t_full_final.log 2>&1
tail -2 gpurun_out/t_full_final.log
timeout -k 10 600 python bench.py > gpurun_out/c_bench_slab_stage1.log 2>&1
tail -1 gpurun_out/c_bench_slab_stage1.log | cut -c1-160
bash scripts/pmc_run.sh > gpurun_out/pmc_run.log 2>&1
tail -5 gpurun_out/pmc_run.log
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
timeout -k 10 500 python bench.py --gpus 2 --backend gloo --share-gpu --bl 8 --steps 2 --warmup 1 --no-other-params --no-other-workloads --no-cpu-baseline > gpurun_out/c_gloo2.log 2>&1
tail -1 gpurun_out/c_gloo2.log | cut -c1-200
