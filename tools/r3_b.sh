#!/bin/bash
# round 3, second GPU pass: whole GPU suite (double-word CG, faithful J^T J diagonal, bounded barriers, sharded g9), campaign sample, LDS microbench
O=$GRAFT_REPO_ROOT/gpurun_out/r3b
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -m gpu -q -W ignore --tb=short -x -s 2>&1 | grep -v "amdgpu.ids\|Gloo\|^$" | tail -120 > $O/pytest_gpu.log
timeout 1500 python tools/random_campaign.py ${1:-200} $O/campaign > $O/campaign.log 2>&1
hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/lds_atomic_bench.hip -o /tmp/ldsb > $O/ldsb_build.log 2>&1 && /tmp/ldsb > $O/ldsb.log 2>&1
