#!/bin/bash
# slot order: where rows of a graph become long enough for the bank-aware order to win
O=$GRAFT_REPO_ROOT/gpurun_out/r3sp
mkdir -p $O; rm -f $O/sweep.log
cd $GRAFT_REPO_ROOT
P='import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j["roofline"]; print(sys.argv[1], "ms/solve %.3f  sweep %.1f us  frac %.3f" % (j["ms_per_step"], 1e3*r["avg_launch_ms"], r["frac"]))'
for cfg in "1000 16 1000000" "1000 32 500000" "1000 64 250000" "1000 128 125000" "300 24 600000"; do
set -- $cfg
for so in banks rows; do
VICAN_SLOT_ORDER=$so timeout 300 python bench.py --workload sparse --cams $1 --cams-per-t $2 --timesteps $3 --no-cpu-baseline 2>/dev/null | python -c "$P" "C=$1 cpt=$2 T=$3 f32 $so" >> $O/sweep.log
done
done
for so in banks rows; do
VICAN_SLOT_ORDER=$so timeout 300 python bench.py --workload sparse --dtype f64 --timesteps 1000000 --no-cpu-baseline 2>/dev/null | python -c "$P" "C=100 cpt=8 T=1M f64 $so" >> $O/sweep.log
done
