#!/usr/bin/env python3
"""What ONE GPU can measure of the collective path: ncclAllReduce on a one-rank RCCL communicator held by the C library
(vican_comm_*, forced to enqueue: include/vican_hip_test.h) at the message sizes of the sharded solve - 3C x 3 doubles per
operator application, 3C + 1 doubles and one scalar per CG iteration (scipy's recurrence), C = 340 (large_shop) and 1000
(stress) - as (a) device time per call between two HIP events around a batch on the launch stream, (b) host time per
enqueue, (c) one call sandwiched between two kernels of the solve's size class (event-timed: kernel + collective + kernel
against kernel + kernel).  A FLOOR for the latencies DESIGN.md section 7 assumes for the 2/4/8-GPU curve: no xGMI hop, no
peer synchronisation, RCCL's one-rank kernel instead of its ring/tree kernels - and one full large_shop solve on the sharded
schedule with these collectives in the loop.

    python tools/rccl_onerank.py [out=gpurun_out/rccl_onerank.txt]
    rocprofv3 --kernel-trace --stats -d gpurun_out/rccl_trace -- python3 tools/rccl_onerank.py      (kernel names of the collective)"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from vican_amd import _lib                                  # noqa: E402

out_path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "gpurun_out", "rccl_onerank.txt")
os.makedirs(os.path.dirname(out_path), exist_ok=True)
lib = _lib.load()
torch.cuda.set_device(0)
lines = []


def say(s):
    print(s, flush=True)
    lines.append(s)


buf = C.create_string_buffer(128)
_lib.check(lib.vican_comm_unique_id(buf), "unique_id")
comm = C.c_void_p()
t0 = time.perf_counter()
_lib.check(lib.vican_comm_create(0, 1, buf, C.byref(comm)), "comm_create")
say("# ncclCommInitRank (1 rank): %.1f ms" % ((time.perf_counter() - t0) * 1e3))
_lib.check(lib.vican_comm_force_enqueue(comm, 1), "force")
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
say("# n doubles | device us per call (batch of 200 between two events) | host us per enqueue | min of 5 batches")
for n_cam in (340, 1000):
    for label, n in (("3C x 3 (operator application)", 9 * n_cam), ("3C + 1 (CG: q_c | p.q)", 3 * n_cam + 1), ("1 (CG: r.r)", 1)):
        x = torch.randn(n, dtype=torch.float64, device="cuda:0")
        keep = x.clone()
        p = C.c_void_p(x.data_ptr())
        for _ in range(20):
            lib.vican_comm_allreduce_sum(comm, p, n, stream)
        torch.cuda.synchronize()
        best, host = 1e9, 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            t0 = time.perf_counter()
            for _ in range(200):
                lib.vican_comm_allreduce_sum(comm, p, n, stream)
            host = min(host, (time.perf_counter() - t0) / 200 * 1e6)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 200 * 1e3)
        assert torch.equal(x, keep)
        say("C=%4d  %-32s n=%5d  device %.2f us  host %.2f us" % (n_cam, label, n, best, host))

# (c) the collective between two kernels (the slab fold in front of it, the camera-side step behind it are ~5-10 us kernels)
for n_cam in (340, 1000):
    n = 9 * n_cam
    x = torch.randn(n, dtype=torch.float64, device="cuda:0")
    p = C.c_void_p(x.data_ptr())

    def seq(with_coll):
        y = x
        for _ in range(50):
            y = x * 1.0000001
            if with_coll:
                lib.vican_comm_allreduce_sum(comm, C.c_void_p(y.data_ptr()), n, stream)
            y = y + 1.0
        return y
    res = {}
    for with_coll in (False, True):
        seq(with_coll)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            seq(with_coll)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 50 * 1e3)
        res[with_coll] = best
    say("C=%4d  kernel | all-reduce(3C x 3) | kernel: %.2f us per triple, kernel | kernel: %.2f us  => +%.2f us per collective in stream order"
        % (n_cam, res[True], res[False], res[True] - res[False]))

# one large_shop-sized solve on the sharded schedule, collectives through RCCL, against identity collectives and the plain solve
from test_comm_gpu import _large_shop_problem               # noqa: E402
from vican_amd.bipgo import solve_problem                   # noqa: E402
from vican_amd.solver import Comm                           # noqa: E402
prob = _large_shop_problem(np.float32)
for name, mk in (("sharded schedule + RCCL one-rank collectives", lambda: Comm.single(force_sharded=True, native=True)),
                 ("sharded schedule + identity collectives", lambda: Comm.single(force_sharded=True)),
                 ("plain single-rank schedule", lambda: Comm.single())):
    c = mk()
    info = {}
    solve_problem(prob, 4, "conjugate_gradient", np.float32, info=info, comm=c)
    ts = []
    for _ in range(5):
        info = {}
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        solve_problem(prob, 4, "conjugate_gradient", np.float32, info=info, comm=c)
        ts.append(time.perf_counter() - t0)
    say("large_shop (340 x 10000 x 4) drop-in solve_problem, %-46s: rot %.2f ms + trans %.2f ms (best of 5 calls %.2f ms incl. pack/gather), "
        "%d all-reduces, cg %d" % (name, info["t_rot"] * 1e3, info["t_trans"] * 1e3, min(ts) * 1e3, c.n_allreduce // 6, info["cg_iters"]))
lib.vican_comm_destroy(comm)
open(out_path, "w").write("\n".join(lines) + "\n")
