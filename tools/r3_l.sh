#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r3l
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $GRAFT_REPO_ROOT/tools/lsqr_time.py > $O/lsqr_time.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/kstats.py $(ls -t $O/stats/*/*kernel_stats.csv | head -1) "lsqr" > $O/kstats.txt 2>&1
rm -rf $O/stats/*/*kernel_trace.csv
