#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r2fence
rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -x 2>&1 | tail -2 > $O/pytest.log
for i in 1 2; do python3 bench.py --workload large_shop --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('default', d['ms_per_step'], d['detail']['cg_ms_per_step'])" >> $O/sum.txt
VICAN_LANCZOS_RESIDENT=1 python3 bench.py --workload large_shop --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('lres on', d['ms_per_step'], d['detail']['cg_ms_per_step'])" >> $O/sum.txt
done
VICAN_LANCZOS_RESIDENT=1 timeout 500 python tools/random_campaign.py 1000 gpurun_out/random_parity_lres 2>&1 | grep -A6 outcomes | head -8 >> $O/sum.txt
