#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r2k
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_kernels_gpu.py -m gpu -q -W ignore --tb=short -x 2>&1 | grep -v "amdgpu.ids" | tail -60 > $O/pytest_k.log
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-large-shop > $O/bench_wave.log 2>&1
VICAN_LAYOUT=block timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-large-shop > $O/bench_block.log 2>&1
timeout 600 python bench.py --workload sparse --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_sparse_wave.log 2>&1
timeout 600 python bench.py --workload large_shop --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_ls_wave.log 2>&1
VICAN_LAYOUT=block timeout 600 python bench.py --workload large_shop --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_ls_block.log 2>&1
