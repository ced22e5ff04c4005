set -e
cd $GRAFT_REPO_ROOT
bash scripts/pmc.sh fetch FETCH_SIZE --roofline-only
cd $GRAFT_REPO_ROOT
bash scripts/pmc.sh write WRITE_SIZE --roofline-only
cd $GRAFT_REPO_ROOT
grep '^{' gpurun_out/pmc_write.log | tail -1 > gpurun_out/pmc_line.json
mkdir -p gpurun_out/pmc_json
python scripts/pmc_roofline.py gpurun_out/pmc_fetch gpurun_out/pmc_write gpurun_out/pmc_line.json gpurun_out/pmc_json
