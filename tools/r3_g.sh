#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r3g
mkdir -p $O
cd $GRAFT_REPO_ROOT
for w in 8 12; do VICAN_WRHS_WAVES=$w timeout 300 python tools/rhs_time.py --tag "stress waves=$w" >> $O/rhs.log 2>&1; done
for w in 8 12; do VICAN_WRHS_WAVES=$w timeout 300 python tools/rhs_time.py --cams 100 --timesteps 2000000 --cpt 8 --tag "sparse waves=$w" >> $O/rhs.log 2>&1; done
VICAN_LIB=$GRAFT_REPO_ROOT/vican_amd/csrc/variants/libvican_hip_cgwstamp.so timeout 300 python tools/cgsweep_time.py --stamp --tag "stress stamp" >> $O/cgstamp.log 2>&1
VICAN_LIB=$GRAFT_REPO_ROOT/vican_amd/csrc/variants/libvican_hip_cgwstamp.so timeout 300 python tools/cgsweep_time.py --stamp --cams 100 --timesteps 2000000 --cpt 8 --tag "sparse stamp" >> $O/cgstamp.log 2>&1
