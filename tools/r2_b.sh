#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r2b
rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q -W ignore --tb=short -x 2>&1 | grep -v "amdgpu.ids" | tail -12 > $O/pytest.log
for rep in 1 2; do
python bench.py --workload large_shop --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_ls_$rep.log 2>&1
done
python bench.py --no-cpu-baseline --steps 10 --warmup 3 > $O/bench_wave.log 2>&1
