#!/bin/bash
# round 3: the one-message CG of sharded runs - stage tests, dist tests, campaign sample, sharded g9
O=$GRAFT_REPO_ROOT/gpurun_out/r3cg1
rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
for n in 2 4; do
  VICAN_DIST_BACKEND=gloo timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500 + n)) tools/dist_g9.py $O/dist_g9_${n}ranks.json 2>&1 | grep "^g9\|^dist\|Error\|error" > $O/dist_g9_${n}ranks.log
done
timeout 200 python bench.py --gpus 2 --workload large_shop --no-cpu-baseline > $O/bench_ls_2.json 2> $O/bench_ls_2.err
VICAN_CG_MESSAGES=2 timeout 200 python bench.py --gpus 2 --workload large_shop --no-cpu-baseline > $O/bench_ls_2_two_messages.json 2> $O/bench_ls_2_two_messages.err
timeout 300 python bench.py --gpus 2 --no-cpu-baseline --timesteps 20000 > $O/bench_stress_2.json 2> $O/bench_stress_2.err
timeout 600 python -m pytest tests/test_translation_stage.py tests/test_dist_gpu.py -m gpu -q -W ignore --tb=short -s 2>&1 | grep -v "amdgpu.ids\|Gloo\|^$" | tail -60 > $O/pytest.log
timeout 600 python tools/random_campaign.py 300 $O/campaign > $O/campaign.log 2>&1
