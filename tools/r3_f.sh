#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r3f
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -m gpu -q -W ignore --tb=short -x 2>&1 | grep -v "amdgpu.ids\|Gloo\|^$" | tail -30 > $O/pytest_gpu.log
timeout 300 python tools/cgsweep_time.py --tag "stress" > $O/cgsweep.log 2>&1
timeout 300 python tools/cgsweep_time.py --cams 100 --timesteps 2000000 --cpt 8 --tag "sparse" >> $O/cgsweep.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-large-shop > $O/bench_stats.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/kstats.py $(ls -t $O/stats/*/*kernel_stats.csv | head -1) "sweep|rhs|cg_|fold" > $O/kstats.txt 2>&1
rm -rf $O/stats/*/*kernel_trace.csv
