#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r2cgres
rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "cg_resident" 2>&1 | tail -25 > $O/pytest.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python3 $GRAFT_REPO_ROOT/bench.py --workload large_shop --no-cpu-baseline > $O/b.log 2>&1
f=$(ls $O/p/*/*kernel_stats.csv | head -1)
python3 $GRAFT_REPO_ROOT/tools/kstats.py $f "cg_res|coop|ritz|wave_sweep" > $O/sum.txt
grep -o '"ms_per_step": [0-9.]*' $O/b.log >> $O/sum.txt
rm -rf $O/p
