set -euo pipefail
cd $GRAFT_REPO_ROOT
python scripts/boxfilter_bench.py --stage 1 --variants 4 --radii 43,110,166,221,277 --win 252 --rounds 2 2>&1 | grep -v amdgpu > gpurun_out/c_boxx.txt
python scripts/boxfilter_bench.py --stage 1 --variants 0 --radii 43,55 --win 252 --rounds 2 2>&1 | grep -v amdgpu >> gpurun_out/c_boxx.txt
A="scripts/boxfilter_bench.py --stage 1 --variants 4 --radii 277 --win 252 --rounds 1"
for c in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY" "SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD" "FETCH_SIZE" "WRITE_SIZE"; do
  bash scripts/pmc_any.sh bx "$c" $A >> gpurun_out/pmc_passes.log 2>&1
  python3 scripts/pmc_report.py gpurun_out/pmc_bx k_boxx >> gpurun_out/c_boxx.txt
done
rm -rf gpurun_out/pmc_bx
cat gpurun_out/c_boxx.txt
