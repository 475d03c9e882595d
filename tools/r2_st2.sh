#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r2st
rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
V=vican_amd/csrc/variants
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -W ignore --tb=short -x 2>&1 | grep -v "amdgpu.ids" | tail -15 > $O/pytest_k.log
for rep in 1 2; do
timeout 300 python tools/wsweep_time.py wave:12:8 block > $O/v_default_$rep.log 2>&1
for v in st; do VICAN_LIB=$V/libvican_hip_$v.so timeout 300 python tools/wsweep_time.py wave:12:8 > $O/v_${v}_$rep.log 2>&1; done
done
VICAN_LIB=$V/libvican_hip_stamp.so timeout 300 python tools/wsweep_time.py --stamp wave:12:8 > $O/stamp.log 2>&1
timeout 300 python tools/wsweep_time.py --cams 100 --timesteps 2000000 --cpt 8 wave block > $O/time_sparse.log 2>&1
