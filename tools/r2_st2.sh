#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r2st
rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
V=vican_amd/csrc/variants
timeout 900 python -m pytest tests/test_kernels_gpu.py -m gpu -q -W ignore --tb=short -x 2>&1 | grep -v "amdgpu.ids" | tail -15 > $O/pytest_k.log
for rep in 1 2 3; do
timeout 300 python tools/wsweep_time.py wave:12:8 wave:12:4 block > $O/v_nt_$rep.log 2>&1
VICAN_LIB=$V/libvican_hip_nont.so timeout 300 python tools/wsweep_time.py wave:12:8 block > $O/v_nont_$rep.log 2>&1
done
