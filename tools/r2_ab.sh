#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r2ab
mkdir -p $O
cd $GRAFT_REPO_ROOT
V=vican_amd/csrc/variants
python tools/wsweep_time.py wave wave:12:32 wave:12:8 wave:8 block > $O/base.log 2>&1
VICAN_LIB=$V/libvican_hip_stamp.so python tools/wsweep_time.py --stamp wave > $O/stamp.log 2>&1
for a in a1 a2 a3 a4 a5; do VICAN_LIB=$V/libvican_hip_$a.so python tools/wsweep_time.py wave > $O/$a.log 2>&1; done
