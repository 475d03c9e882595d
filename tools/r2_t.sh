#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r2t
rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -q -W ignore --tb=short 2>&1 | grep -v "amdgpu.ids" | tail -40 > $O/pytest_tiled.log
