#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r2st
rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
timeout 300 python tools/wsweep_time.py --cams 100 --timesteps 2000000 --cpt 8 wave:12:1 wave:12:2 wave:12:4 wave:8:2 block > $O/sparse_$rep.log 2>&1
done
timeout 300 python tools/wsweep_time.py --cams 340 --timesteps 10000 --cpt 4 --reps 60 wave:4 wave:4:1 wave:8 block > $O/ls.log 2>&1
