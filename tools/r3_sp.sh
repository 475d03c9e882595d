#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r3sp
mkdir -p $O; rm -f $O/pair.log
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_kernels_gpu.py tests/test_parity_gpu.py -m gpu -q -W ignore --tb=short -x 2>&1 | grep -v "amdgpu.ids\|Gloo\|^$" | tail -3 > $O/pytest_pair.log
P='import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j["roofline"]; print(sys.argv[1], "ms/solve %.3f  sweep %.1f us  frac %.3f" % (j["ms_per_step"], 1e3*r["avg_launch_ms"], r["frac"]))'
for i in 1 2; do
timeout 300 python bench.py --no-cpu-baseline --no-large-shop --no-sparse 2>/dev/null | python -c "$P" "stress" >> $O/pair.log
timeout 300 python bench.py --workload sparse --no-cpu-baseline 2>/dev/null | python -c "$P" "sparse" >> $O/pair.log
done
timeout 300 python bench.py --workload sparse --cams 1000 --cams-per-t 64 --timesteps 250000 --no-cpu-baseline 2>/dev/null | python -c "$P" "C=1000 cpt=64" >> $O/pair.log
timeout 300 python bench.py --workload sparse --dtype f64 --timesteps 1000000 --no-cpu-baseline 2>/dev/null | python -c "$P" "sparse f64" >> $O/pair.log
timeout 300 python bench.py --dtype f64 --timesteps 50000 --no-cpu-baseline --no-large-shop --no-sparse 2>/dev/null | python -c "$P" "stress f64 T=50k" >> $O/pair.log
