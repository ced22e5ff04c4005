# GPU box: the round's record -- bench lines of the three workloads, per-kernel summaries of the three parameter sets,
# the PMC traffic files bench.py reads, the exact row filter's timings and counters.  Outputs under gpurun_out/.
set -euo pipefail
cd $GRAFT_REPO_ROOT
if [ -z "$SKIP_SLAB" ]; then timeout -k 10 500 python bench.py > gpurun_out/c_bench_slab_stage1.log 2>&1; tail -1 gpurun_out/c_bench_slab_stage1.log | cut -c1-200; fi
timeout -k 10 400 python bench.py --workload chain > gpurun_out/c_bench_chain.log 2>&1
tail -1 gpurun_out/c_bench_chain.log | cut -c1-200
timeout -k 10 400 python bench.py --workload ska --steps 16 > gpurun_out/c_bench_ska.log 2>&1
tail -1 gpurun_out/c_bench_ska.log | cut -c1-200
for p in stage1 defaults very_broad; do
  bash scripts/prof.sh c_$p --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-other-params --no-scatter --params $p
  cd $GRAFT_REPO_ROOT
  python scripts/kernel_summary.py gpurun_out/prof_c_$p > gpurun_out/c_${p}_kernel_summary.txt
  rm -rf gpurun_out/prof_c_$p
done
bash scripts/pmc_run.sh
rm -rf gpurun_out/pmc_fetch gpurun_out/pmc_write
python scripts/boxfilter_bench.py --stage 1 --variants 4 --radii 43,110,166,221,277 --win 252 --rounds 2 > gpurun_out/c_boxx.txt 2>&1
A="scripts/boxfilter_bench.py --stage 1 --variants 4 --radii 277 --win 252 --rounds 1"
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD" "FETCH_SIZE" "WRITE_SIZE"; do
  bash scripts/pmc_any.sh bx "$c" $A >> gpurun_out/pmc_passes.log 2>&1
  python3 scripts/pmc_report.py gpurun_out/pmc_bx k_boxx >> gpurun_out/c_boxx.txt
done
rm -rf gpurun_out/pmc_bx
