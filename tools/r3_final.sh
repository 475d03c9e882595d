#!/bin/bash
# round 3: evidence run - GPU suite, rocprofv3 passes of the default bench + sparse capture, 1000-seed campaign, sharded g9, translation kernels
O=$GRAFT_REPO_ROOT/gpurun_out/r3final
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 3000 python -m pytest tests -m gpu -q -W ignore --tb=short 2>&1 | grep -v "amdgpu.ids\|Gloo\|^$" | tail -30 > $O/pytest_gpu.log
tools/profile_round.sh r03 > $O/profile_round.log 2>&1
timeout 2400 python tools/random_campaign.py 1000 $O/campaign > $O/campaign.log 2>&1
for n in 2 4; do
  VICAN_DIST_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) tools/dist_g9.py $O/dist_g9_${n}ranks.json 2>&1 | grep "^g9\|^dist" > $O/dist_g9_${n}ranks.log
done
{
  timeout 300 python tools/cgsweep_time.py --tag "stress"
  timeout 300 python tools/cgsweep_time.py --cams 100 --timesteps 2000000 --cpt 8 --tag "sparse"
  timeout 300 python tools/rhs_time.py --tag "stress"
  timeout 300 python tools/rhs_time.py --cams 100 --timesteps 2000000 --cpt 8 --tag "sparse"
  timeout 600 python tools/lsqr_time.py
} 2>&1 | grep -v amdgpu > $O/translation_kernels.txt
hipcc --offload-arch=gfx950 -O3 tools/stride_read_bench.hip -o /tmp/srb > /dev/null 2>&1 && /tmp/srb > $O/stride_read_bench.txt 2>&1
hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/lds_atomic_bench.hip -o /tmp/ldsb > /dev/null 2>&1 && /tmp/ldsb > $O/lds_atomic_bench.txt 2>&1
timeout 600 python tools/api_time.py --oracle 2>&1 | grep -v "amdgpu\|Warn" > $O/api_time.txt
python tools/cold_trace.py 24 2>&1 | grep "^call" > $O/cold_calls.txt
timeout 300 python tools/ragged_solve.py 2>&1 | grep -v "amdgpu\|Warn" > $O/ragged_solve.txt
