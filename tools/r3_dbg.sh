#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r3dbg
mkdir -p $O
cd $GRAFT_REPO_ROOT
VICAN_DIST_BACKEND=gloo timeout 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 tools/dist_probe.py > $O/probe2.log 2>&1
echo "rc $?" >> $O/probe2.log
timeout 600 python -m pytest tests/test_kernels_gpu.py -m gpu -q -W ignore --tb=short -k "lsqr" --durations=5 2>&1 | grep -v "amdgpu.ids" | tail -40 > $O/pytest_lsqr.log
