#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r2ritz
rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x -k "ritz" 2>&1 | tail -25 > $O/pytest.log
echo "== fast" > $O/sum.txt
timeout 300 python tools/ritz_bench.py 2>&1 | grep -v amdgpu >> $O/sum.txt
echo "== jacobi" >> $O/sum.txt
VICAN_RITZ_FAST=0 timeout 300 python tools/ritz_bench.py 2>&1 | grep -v amdgpu >> $O/sum.txt
