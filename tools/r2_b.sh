#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r2b
rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q -W ignore --tb=short -s 2>&1 | grep -v "amdgpu.ids" | grep -E "passed|failed|FAILED|Error|translation stage|large_shop scale|assert|mismatch" | head -60 > $O/pytest.log
for rep in 1 2; do
VICAN_TL_NT=1 python bench.py --no-cpu-baseline --no-large-shop --steps 10 --warmup 3 > $O/bench_tlnt1_$rep.log 2>&1
VICAN_TL_NT=0 python bench.py --no-cpu-baseline --no-large-shop --steps 10 --warmup 3 > $O/bench_tlnt0_$rep.log 2>&1
done
