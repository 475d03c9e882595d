#!/bin/bash
cat gpurun_out/pytest_gpu.log
for bt in 1024 768 512; do tail -1 gpurun_out/bench_bt$bt.log | python -c "
import sys,json
try:
    d=json.loads(sys.stdin.read()); r=d['roofline']; print('bt$bt: op %.1f us  %.0f GB/s  frac %.3f | step %.2f ms (rot %.2f cg %.2f) lanczos %s cg_it %s' % (r['avg_launch_ms']*1e3, r['achieved'], r['frac'], d['ms_per_step'], d['detail']['rot_loop_ms_per_step'], d['detail']['cg_ms_per_step'], d['detail']['lanczos_steps'], d['detail']['cg_iters']))
except Exception as e: print('bench bt$bt failed', e)"; done
python tools/pmc_summary.py "${1:-block_sweep_kernel<float, 1024, 0>}" gpurun_out/pmc1 gpurun_out/pmc2 | grep -v "^gpurun"
