#!/bin/bash
# phase 2 of the operator sweep by DPP (f32 blocks) against the previous build
O=$GRAFT_REPO_ROOT/gpurun_out/r3dpp
mkdir -p $O; rm -f $O/time.log
cd $GRAFT_REPO_ROOT
OLD=$GRAFT_REPO_ROOT/vican_amd/csrc/variants/libvican_hip_old.so
P='import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j["roofline"]; print(sys.argv[1], "ms/solve %.3f  sweep %.1f us  frac %.3f" % (j["ms_per_step"], 1e3*r["avg_launch_ms"], r["frac"]))'
for rep in 1 2; do
for lib in old new; do
  if [ $lib = old ]; then export VICAN_LIB=$OLD; else unset VICAN_LIB; fi
  timeout 300 python tools/ragged_time.py 2>&1 | grep default | sed "s/^/$lib /" | cut -c1-60,150-260 >> $O/time.log
  timeout 300 python bench.py --workload sparse --no-cpu-baseline 2>/dev/null | python -c "$P" "$lib sparse" >> $O/time.log
  timeout 300 python bench.py --no-cpu-baseline --no-large-shop --no-sparse 2>/dev/null | python -c "$P" "$lib stress" >> $O/time.log
  timeout 300 python bench.py --workload large_shop --no-cpu-baseline 2>/dev/null | python -c "$P" "$lib large_shop" >> $O/time.log
done
done
unset VICAN_LIB
timeout 300 python bench.py --workload sparse --dtype f64 --timesteps 1000000 --no-cpu-baseline 2>/dev/null | python -c "$P" "new sparse f64" >> $O/time.log
VICAN_LIB=$OLD timeout 300 python bench.py --workload sparse --dtype f64 --timesteps 1000000 --no-cpu-baseline 2>/dev/null | python -c "$P" "old sparse f64" >> $O/time.log
