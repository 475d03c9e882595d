#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r2tl
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $O/tr -- python3 $GRAFT_REPO_ROOT/bench.py --workload large_shop --no-cpu-baseline --steps 3 --warmup 2 > $O/bench.log 2>&1
cd $GRAFT_REPO_ROOT
TL_VERBOSE=1 python tools/timeline.py $O/tr 2 > $O/tl_prev.txt 2>&1
rm -rf $O/tr
