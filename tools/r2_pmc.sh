#!/bin/bash
# PMC passes for the operator sweep (wave and block layouts) on the stress graph
O=$GRAFT_REPO_ROOT/gpurun_out/r2pmc
rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/tools/wsweep_time.py --reps 12 ${VARIANTS:-wave:12:8 block}"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/sq1 -- $B > $O/sq1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_SALU --output-format csv -d $O/sq2 -- $B > $O/sq2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE --output-format csv -d $O/sq3 -- $B > $O/sq3.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/pmc_summary.py sweep_kernel $O/sq1 $O/sq2 $O/sq3 > $O/summary.txt 2>&1
rm -rf $O/*/*/*kernel_trace.csv $O/*/*/*agent_info.csv
