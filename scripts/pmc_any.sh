#!/bin/bash
# usage (GPU box): bash scripts/pmc_any.sh <tag> "<counters>" <python script> [args...]
# one rocprofv3 --pmc pass (counters in their own run, kernel-trace only) of any script
tag=$1; ctr=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -- python3 $GRAFT_REPO_ROOT/"$@" > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log 2>&1
python3 $GRAFT_REPO_ROOT/scripts/pmc_report.py $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag k_box
