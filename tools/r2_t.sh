#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r2t
rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q -W ignore --tb=short -s 2>&1 | grep -v "amdgpu.ids" | grep -E "passed|failed|FAILED|Error|translation stage|large_shop scale|assert|mismatch" | head -60 > $O/pytest.log
