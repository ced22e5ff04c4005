set -e
cd $GRAFT_REPO_ROOT
timeout -k 10 400 python bench.py > gpurun_out/c_bench_slab_stage1.log 2>&1
tail -1 gpurun_out/c_bench_slab_stage1.log | cut -c1-200
timeout -k 10 400 python bench.py --workload chain > gpurun_out/c_bench_chain.log 2>&1
tail -1 gpurun_out/c_bench_chain.log | cut -c1-200
timeout -k 10 400 python bench.py --workload ska > gpurun_out/c_bench_ska.log 2>&1
tail -1 gpurun_out/c_bench_ska.log | cut -c1-200
for p in stage1 defaults very_broad; do
  bash scripts/prof.sh c_$p --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-other-params --no-scatter --params $p
  cd $GRAFT_REPO_ROOT
  python scripts/kernel_summary.py gpurun_out/prof_c_$p > gpurun_out/c_${p}_kernel_summary.txt
done
