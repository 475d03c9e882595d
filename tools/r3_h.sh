#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r3h
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_parity_gpu.py tests/test_random_parity_gpu.py tests/test_tiled_gpu.py -m gpu -q -W ignore --tb=short -k "lsqr or direct or tiles" 2>&1 | grep -v "amdgpu.ids\|Gloo\|^$" | tail -30 > $O/pytest_lsqr.log
timeout 600 python tools/lsqr_time.py > $O/lsqr_time.log 2>&1
timeout 1500 python -m pytest tests -m gpu -q -W ignore --tb=short --durations=12 2>&1 | grep -v "amdgpu.ids\|Gloo\|^$" | tail -60 > $O/pytest_gpu.log
