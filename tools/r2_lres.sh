#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r2lres
rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x 2>&1 | tail -3 >> $O/pytest.log
cd /tmp && export TMPDIR=/tmp
for w in large_shop stress; do
rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python3 $GRAFT_REPO_ROOT/bench.py --workload $w --no-cpu-baseline --no-large-shop > $O/b_$w.log 2>&1
f=$(ls $O/p/*/*kernel_stats.csv | head -1)
echo "== $w" >> $O/sum.txt
python3 $GRAFT_REPO_ROOT/tools/kstats.py $f "resident|coop|ritz|wave_sweep<float, (4|12), 0" >> $O/sum.txt
grep -o '"ms_per_step": [0-9.]*' $O/b_$w.log >> $O/sum.txt
rm -rf $O/p
done
cd $GRAFT_REPO_ROOT
for i in 1 2; do
python bench.py --workload large_shop --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('resident', d['ms_per_step'], d['detail']['rot_loop_ms_per_step'])" >> $O/sum.txt
VICAN_LANCZOS_RESIDENT=0 python bench.py --workload large_shop --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('pairs   ', d['ms_per_step'], d['detail']['rot_loop_ms_per_step'])" >> $O/sum.txt
done
