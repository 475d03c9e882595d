#!/bin/bash
# round 3, third GPU pass: whole GPU suite, campaign, kernel stats of the default bench (rocprofv3 --kernel-trace --stats)
O=$GRAFT_REPO_ROOT/gpurun_out/r3c
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -m gpu -q -W ignore --tb=short -s 2>&1 | grep -v "amdgpu.ids\|Gloo\|^$" | tail -150 > $O/pytest_gpu.log
timeout 1500 python tools/random_campaign.py ${1:-300} $O/campaign > $O/campaign.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-large-shop > $O/bench_stats.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/kstats.py $(ls -t $O/stats/*/*kernel_stats.csv | head -1) "sweep|rhs|cg_|fold" > $O/kstats.txt 2>&1
rm -rf $O/stats/*/*kernel_trace.csv
python bench.py --workload sparse --no-cpu-baseline --steps 5 --warmup 2 > $O/bench_sparse.log 2>&1
python bench.py --workload large_shop --no-cpu-baseline --steps 20 --warmup 3 > $O/bench_ls.log 2>&1
