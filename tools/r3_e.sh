#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r3e
mkdir -p $O
cd $GRAFT_REPO_ROOT
hipcc --offload-arch=gfx950 -O3 tools/stride_read_bench.hip -o /tmp/srb > $O/srb_build.log 2>&1 && /tmp/srb > $O/srb.log 2>&1
