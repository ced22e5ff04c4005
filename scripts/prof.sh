#!/bin/bash
# usage (on the GPU box): bash scripts/prof.sh <tag> <bench args...>
# rocprofv3 kernel trace + stats of one bench.py run; output under gpurun_out/prof_<tag>/
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_$tag -- python3 $GRAFT_REPO_ROOT/bench.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log 2>&1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/prof_$tag.log | cut -c1-120
