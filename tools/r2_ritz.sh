#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r2ritz
rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
python tools/ritz_bench.py > $O/default.log 2>&1
for t in 64 128 256; do VICAN_RITZ_THREADS=$t python tools/ritz_bench.py > $O/t$t.log 2>&1; done
