#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r3j
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -m gpu -q -W ignore --tb=short 2>&1 | grep -v "amdgpu.ids\|Gloo\|^$" | tail -30 > $O/pytest_gpu.log
timeout 1500 python tools/random_campaign.py ${1:-300} $O/campaign > $O/campaign.log 2>&1
timeout 600 python bench.py --workload large_shop --dtype f64 --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_ls_f64.log 2>&1
