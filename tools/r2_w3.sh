#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r2w3
rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "fused" 2>&1 | tail -15 > $O/pytest.log
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-large-shop"
for v in ; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/$v -- $B > $O/$v.log 2>&1
  f=$(ls $O/$v/*/*kernel_stats.csv | head -1)
  echo "== $v" >> $O/sum.txt
  python3 $GRAFT_REPO_ROOT/tools/kstats.py $f "wave_sweep|cg_wsweep|trans_rhs" >> $O/sum.txt
  grep -o '"ms_per_step": [0-9.]*' $O/$v.log >> $O/sum.txt
  rm -rf $O/$v
done
