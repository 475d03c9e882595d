#!/bin/bash
O=$GRAFT_REPO_ROOT/gpurun_out/r2b
rm -rf $O; mkdir -p $O
cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests/test_kernels_gpu.py tests/test_parity_gpu.py tests/test_translation_stage.py tests/test_large_shop_scale.py -m gpu -q -W ignore --tb=short -x -s 2>&1 | grep -v "amdgpu.ids" | grep -E "passed|failed|FAILED|Error|translation stage|large_shop scale|assert|mismatch" | head -60 > $O/pytest.log
for rep in 1 2; do
python bench.py --no-cpu-baseline --no-large-shop --steps 10 --warmup 3 > $O/bench_wave_$rep.log 2>&1
done
python bench.py --workload sparse --no-cpu-baseline --steps 5 --warmup 2 > $O/bench_sparse_wave.log 2>&1
python bench.py --workload large_shop --no-cpu-baseline --steps 5 --warmup 2 > $O/bench_ls_wave.log 2>&1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_wave -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-large-shop --steps 5 --warmup 2 > $O/bench_wave_prof.log 2>&1
cd $GRAFT_REPO_ROOT
python - > $O/summary.txt <<'PY'
import csv, glob
f = glob.glob("gpurun_out/r2b/stats_wave/*/*kernel_stats.csv")[0]
for r in list(csv.DictReader(open(f)))[:16]:
    print("%-70s calls %4s avg %9.1f us  total %8.2f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
rm -rf $O/*/*/*kernel_trace.csv $O/*/*/*agent_info.csv
