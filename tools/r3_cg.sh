#!/bin/bash
# round 3, first GPU pass: double-word CG accumulators - parity tests of the translation stage, cost of the second word
O=$GRAFT_REPO_ROOT/gpurun_out/r3a
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_translation_stage.py tests/test_kernels_gpu.py -m gpu -q -W ignore --tb=short -s 2>&1 | grep -v "amdgpu.ids" | tail -60 > $O/pytest_trans.log
for v in "" hionly loads; do
  lib=""; [ -n "$v" ] && lib=$GRAFT_REPO_ROOT/vican_amd/csrc/variants/libvican_hip_$v.so
  VICAN_LIB=$lib timeout 300 python tools/cgsweep_time.py --tag "stress ${v:-full}" >> $O/cgsweep.log 2>&1
  VICAN_LIB=$lib timeout 300 python tools/cgsweep_time.py --cams 100 --timesteps 2000000 --cpt 8 --tag "sparse ${v:-full}" >> $O/cgsweep.log 2>&1
done
timeout 1200 python -m pytest tests/test_parity_gpu.py tests/test_large_shop_scale.py tests/test_random_parity_gpu.py tests/test_dist_gpu.py tests/test_tiled_gpu.py -m gpu -q -W ignore --tb=short -s 2>&1 | grep -v "amdgpu.ids" | tail -60 > $O/pytest_parity.log
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $O/bench.log 2>&1
