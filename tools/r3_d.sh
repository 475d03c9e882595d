#!/bin/bash
# round 3, fourth GPU pass: device merge tests, whole suite, bench with detail.sparse, campaign
O=$GRAFT_REPO_ROOT/gpurun_out/r3d
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_merge_gpu.py -m gpu -q -W ignore --tb=short -s 2>&1 | grep -v "amdgpu.ids\|Gloo\|^$" | tail -40 > $O/pytest_merge.log
timeout 3000 python -m pytest tests -m gpu -q -W ignore --tb=short 2>&1 | grep -v "amdgpu.ids\|Gloo\|^$" | tail -60 > $O/pytest_gpu.log
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err
timeout 1500 python tools/random_campaign.py ${1:-300} $O/campaign > $O/campaign.log 2>&1
