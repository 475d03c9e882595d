#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r2t
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -m gpu -q -W ignore --tb=short 2>&1 | grep -v "amdgpu.ids" | tail -30 > $O/pytest_gpu.log
timeout 3000 python -m pytest tests -m gpu -q -W ignore --tb=short 2>&1 | grep -v "amdgpu.ids" | tail -5 > $O/pytest_gpu2.log
