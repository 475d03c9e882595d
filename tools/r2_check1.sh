#!/bin/bash
# round-2 first GPU pass: can a GPU-initialised python start children?  GPU tests, 2-rank bench, default bench.
O=$GRAFT_REPO_ROOT/gpurun_out/r2c1
mkdir -p $O
cd $GRAFT_REPO_ROOT
python - > $O/child_probe.log 2>&1 <<'PY'
import subprocess, sys, torch
print("cuda", torch.cuda.is_available())
x = torch.ones(4, device="cuda"); print(float(x.sum()))
r = subprocess.run([sys.executable, "-c", "import torch; print('child ok', torch.cuda.is_available())"], capture_output=True, text=True)
print("rc", r.returncode, r.stdout[-300:], r.stderr[-600:])
PY
timeout 2400 python -m pytest tests -m gpu -x -q -W ignore --tb=short 2>&1 | tail -40 > $O/pytest_gpu.log
timeout 600 python bench.py --gpus 2 --workload large_shop --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_ls_g2.log 2>&1
timeout 600 python bench.py --gpus 1 --workload large_shop --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_ls_g1.log 2>&1
timeout 900 python bench.py --steps 10 --warmup 3 > $O/bench_default.log 2>&1
timeout 900 python bench.py --workload sparse --steps 5 --warmup 2 --no-cpu-baseline > $O/bench_sparse.log 2>&1
