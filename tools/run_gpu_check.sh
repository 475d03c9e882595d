#!/bin/bash
# usage (on the GPU box, from the repo root): tools/run_gpu_check.sh [bench args...]
# GPU tests + stress bench at two workgroup sizes + PMC pass for the sweep kernel.
O=$GRAFT_REPO_ROOT/gpurun_out
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -W ignore --tb=line 2>&1 | grep -E "Error|FAILED|passed|failed" | cut -c1-300 | head -30 > $O/pytest_gpu.log
for bt in 1024 768 512; do python bench.py --steps 2 --warmup 1 --no-cpu-baseline --block-threads $bt "$@" > $O/bench_bt$bt.log 2>&1; done
cd /tmp && export TMPDIR=/tmp
B="python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-cpu-baseline $*"
rm -rf $O/pmc1 $O/pmc2
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc1 -- $B > $O/pmc1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_SALU --output-format csv -d $O/pmc2 -- $B > $O/pmc2.log 2>&1
