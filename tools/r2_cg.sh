#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r2cg
mkdir -p $O
cd $GRAFT_REPO_ROOT
V=vican_amd/csrc/variants
{
timeout 300 python tools/cgsweep_time.py --tag new
VICAN_LIB=$V/libvican_hip_cgz1.so timeout 300 python tools/cgsweep_time.py --tag swz_xor
VICAN_LIB=$V/libvican_hip_cgz2.so timeout 300 python tools/cgsweep_time.py --tag swz_rand
VICAN_LIB=$V/libvican_hip_cgz1s.so timeout 300 python tools/cgsweep_time.py --tag swz_xor_stamp --stamp
} 2>&1 | grep -v amdgpu.ids > $O/cg.log
