#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r2st
mkdir -p $O
cd $GRAFT_REPO_ROOT
V=vican_amd/csrc/variants
timeout 1500 python -m pytest tests/test_kernels_gpu.py -m gpu -q -W ignore --tb=short -x 2>&1 | grep -v "amdgpu.ids" | tail -30 > $O/pytest_k.log
python tools/wsweep_time.py wave:12:8 wave:12:16 block > $O/time.log 2>&1
VICAN_LIB=$V/libvican_hip_stamp.so python tools/wsweep_time.py --stamp wave:12:8 > $O/stamp.log 2>&1
python tools/wsweep_time.py --cams 100 --timesteps 2000000 --cpt 8 wave block > $O/time_sparse.log 2>&1
