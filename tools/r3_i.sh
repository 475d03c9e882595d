#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r3i
mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_kernels_gpu.py tests/test_translation_stage.py tests/test_parity_gpu.py -m gpu -q -W ignore --tb=short -x 2>&1 | grep -v "amdgpu.ids\|Gloo\|^$" | tail -15 > $O/pytest.log
timeout 300 python tools/cgsweep_time.py --tag "stress" > $O/cgsweep.log 2>&1
VICAN_LIB=$GRAFT_REPO_ROOT/vican_amd/csrc/variants/libvican_hip_cgwstamp.so timeout 300 python tools/cgsweep_time.py --stamp --tag "stress stamp" >> $O/cgsweep.log 2>&1
timeout 300 python tools/rhs_time.py --tag "stress" >> $O/cgsweep.log 2>&1
timeout 600 python bench.py --no-cpu-baseline --no-large-shop --no-sparse > $O/bench.log 2>&1
