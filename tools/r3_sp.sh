#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r3sp
mkdir -p $O; rm -f $O/trans.log
cd $GRAFT_REPO_ROOT
for so in banks rows; do
  VICAN_SLOT_ORDER=$so timeout 300 python tools/cgsweep_time.py --cams 100 --timesteps 2000000 --cpt 8 --tag "sparse $so" 2>&1 | grep -v amdgpu >> $O/trans.log
  VICAN_SLOT_ORDER=$so timeout 300 python tools/rhs_time.py --cams 100 --timesteps 2000000 --cpt 8 --tag "sparse $so" 2>&1 | grep -v amdgpu >> $O/trans.log
  VICAN_SLOT_ORDER=$so timeout 300 python tools/cgsweep_time.py --tag "stress $so" 2>&1 | grep -v amdgpu >> $O/trans.log
  VICAN_SLOT_ORDER=$so timeout 300 python tools/rhs_time.py --tag "stress $so" 2>&1 | grep -v amdgpu >> $O/trans.log
done
