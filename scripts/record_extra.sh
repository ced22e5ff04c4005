# GPU box: the rest of the round's record -- kernel summaries of the chain and SKA workloads, instructions per sample
# and HBM traffic per kernel of one stage-1 step (counters in their own passes).  Outputs under gpurun_out/.
set -euo pipefail
cd $GRAFT_REPO_ROOT
bash scripts/prof.sh chain --workload chain --steps 1 --warmup 0 --no-cpu-baseline --no-roofline
cd $GRAFT_REPO_ROOT
python scripts/kernel_summary.py gpurun_out/prof_chain > gpurun_out/x_chain_kernel_summary.txt; rm -rf gpurun_out/prof_chain
bash scripts/prof.sh ska --workload ska --steps 8 --warmup 1 --no-cpu-baseline --no-roofline
cd $GRAFT_REPO_ROOT
python scripts/kernel_summary.py gpurun_out/prof_ska > gpurun_out/x_ska_kernel_summary.txt; rm -rf gpurun_out/prof_ska
A="--steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-other-params --no-scatter --params stage1 --no-parity-check"
bash scripts/pmc.sh insts "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES" $A >> gpurun_out/pmc_passes.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/pmc_valu_all.py gpurun_out/pmc_insts 4227858432 > gpurun_out/x_instructions_stage1.txt; rm -rf gpurun_out/pmc_insts
bash scripts/pmc.sh f1 FETCH_SIZE $A >> gpurun_out/pmc_passes.log 2>&1
cd $GRAFT_REPO_ROOT
bash scripts/pmc.sh w1 WRITE_SIZE $A >> gpurun_out/pmc_passes.log 2>&1
cd $GRAFT_REPO_ROOT
python scripts/pmc_traffic_all.py gpurun_out/pmc_f1 gpurun_out/pmc_w1 4227858432 > gpurun_out/x_traffic_stage1.txt; rm -rf gpurun_out/pmc_f1 gpurun_out/pmc_w1
tail -3 gpurun_out/x_chain_kernel_summary.txt | cut -c1-120; head -5 gpurun_out/x_instructions_stage1.txt | cut -c1-150; head -5 gpurun_out/x_traffic_stage1.txt | cut -c1-150
