#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r2final
rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -m gpu -q -W ignore --tb=short 2>&1 | grep -v "amdgpu.ids" | tail -15 > $O/pytest_gpu.log
timeout 900 python tools/random_campaign.py 1000 gpurun_out/random_parity > $O/campaign.log 2>&1
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1
