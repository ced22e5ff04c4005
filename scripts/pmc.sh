#!/bin/bash
# usage (GPU box): bash scripts/pmc.sh <tag> <counter> <bench args...>
# one rocprofv3 --pmc pass (counters in their own run, kernel-trace only)
tag=$1; ctr=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag
rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag -- python3 $GRAFT_REPO_ROOT/bench.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag.log 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/pmc_$tag/*/ | head
