#!/bin/bash
# usage (GPU box, repo root):  tools/profile_round.sh r01
# rocprofv3 evidence for bench.py's default command: kernel-trace stats + HBM traffic counters
# (separate --pmc passes, as MI355X_MICROARCH.md prescribes).  Output: gpurun_out/profile_<tag>/ ;
# copy the summaries into profiles/ with tools/collect_profiles.py.
TAG=${1:-r01}
O=$GRAFT_REPO_ROOT/gpurun_out/profile_$TAG
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-large-shop --no-sparse --no-wide --no-facade --no-sharded-schedule"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- $B > $O/bench_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B --steps 1 --warmup 0 > $O/bench_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/write -- $B --steps 1 --warmup 0 > $O/bench_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/sq1 -- $B --steps 1 --warmup 0 > $O/bench_sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_SALU --output-format csv -d $O/sq2 -- $B --steps 1 --warmup 0 > $O/bench_sq2.log 2>&1
cd $GRAFT_REPO_ROOT && python bench.py > $O/bench.json 2> $O/bench.err
# busy/idle timelines of one WARM timed solve per workload
python tools/timeline.py $O/stats 4 > $O/timeline_stress.txt 2>&1      # (the last two solves of a run are the cold and the instrumented one: bench.py)
cd /tmp
rocprofv3 --kernel-trace --output-format csv -d $O/trace_ls -- python3 $GRAFT_REPO_ROOT/bench.py --workload large_shop --no-cpu-baseline --no-facade --steps 3 --warmup 1 > $O/bench_trace_ls.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/timeline.py $O/trace_ls 4 > $O/timeline_large_shop.txt 2>&1
rm -rf $O/trace_ls $O/*/*/*kernel_trace.csv $O/*/*/*agent_info.csv
# the sparse capture (100 cameras x 2 M timesteps x 8 cameras per timestep): kernel stats + HBM traffic counters of its operator sweep
cd /tmp
BS="python3 $GRAFT_REPO_ROOT/bench.py --workload sparse --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/sparse_stats -- $BS > $O/bench_sparse_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/sparse_fetch -- $BS --steps 1 --warmup 0 > $O/bench_sparse_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/sparse_write -- $BS --steps 1 --warmup 0 > $O/bench_sparse_write.log 2>&1
cd $GRAFT_REPO_ROOT && python bench.py --workload sparse --no-cpu-baseline > $O/bench_sparse.json 2> $O/bench_sparse.err
rm -rf $O/*/*/*kernel_trace.csv $O/*/*/*agent_info.csv
# the camera-tiled path (4000 cameras x 100 000 timesteps x 250 cameras per timestep): kernel stats, HBM traffic and SQ counters of the fused launch
cd /tmp
BW="python3 $GRAFT_REPO_ROOT/bench.py --workload wide --no-cpu-baseline --steps 3 --warmup 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/wide_stats -- $BW > $O/bench_wide_stats.log 2>&1
cd $GRAFT_REPO_ROOT && python tools/timeline.py $O/wide_stats 2 > $O/timeline_wide.txt 2>&1
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/wide_fetch -- $BW --steps 1 --warmup 0 > $O/bench_wide_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/wide_write -- $BW --steps 1 --warmup 0 > $O/bench_wide_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/wide_sq1 -- $BW --steps 1 --warmup 0 > $O/bench_wide_sq1.log 2>&1
cd $GRAFT_REPO_ROOT && python bench.py --workload wide --no-cpu-baseline --steps 3 --warmup 1 > $O/bench_wide.json 2> $O/bench_wide.err
rm -rf $O/*/*/*kernel_trace.csv $O/*/*/*agent_info.csv
