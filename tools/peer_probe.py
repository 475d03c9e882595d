#!/usr/bin/env python3
"""The peer exchange (include/vican_hip.h: vican_comm_peer_*) between PROCESSES: N ranks, each a fresh process, map each other's
mailboxes through hipIpc handles and all-reduce messages of the solver's sizes through ONE launch of the library's kernel.

  VICAN_DIST_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/peer_probe.py [out.json]

On a one-GPU box the ranks share cuda:0 (same-device hipIpc: the mailboxes of the other process in this process's address
space, written and polled by kernels of both processes running side by side - the links are the only part a node adds).
Checks, per message size: the result equals the sum of the ranks' inputs formed IN RANK ORDER bit for bit (inputs gathered
over the process group), on every rank; a gated launch that the device cancels leaves the buffer and the epoch alone;
a burst of 200 back-to-back exchanges without host synchronisation (both parities, the reuse hazard) stays exact.  Reports
microseconds per exchange.  Prints "peer probe: transport=<t> mismatches <n>" (transport=torch: this pool refused the mapping)."""
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np                                          # noqa: E402
import torch                                                # noqa: E402
import torch.distributed as dist                            # noqa: E402

from vican_amd import _lib                                  # noqa: E402
from vican_amd.solver import Comm                           # noqa: E402

backend = os.environ.get("VICAN_DIST_BACKEND", "nccl")
dist.init_process_group(backend)
rank, world = dist.get_rank(), dist.get_world_size()
ndev = torch.cuda.device_count()
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank)) % max(ndev, 1))
dev = torch.device("cuda", torch.cuda.current_device())
lib = _lib.load()
comm = Comm(transport=os.environ.get("VICAN_COMM", "peer"))
comm._setup_native(dev)
report = {"world": world, "backend": backend, "devices": min(ndev, world), "transport": comm.transport, "notes": comm.notes, "sizes": {}}
bad = 0
contended = False
if comm.transport == "peer":
    lib.vican_comm_peer_set_timeout(comm.native_handle(), 5_000_000)       # (a test: five seconds per wait, not thirty)
    gen = torch.Generator(device="cpu"); gen.manual_seed(100 + rank)
    stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    for n in (1, 7, 512, 3 * 340 + 96, 9 * 340, 9 * 1000, 3 * 1000 + 96, 40000, Comm.PEER_MAX_DOUBLES):
        x = (torch.randn(n, generator=gen, dtype=torch.float64) * 10.0 ** torch.randint(-8, 8, (n,), generator=gen).double())
        parts = [None] * world
        dist.all_gather_object(parts, x.numpy())
        want = parts[0].copy()
        for p in parts[1:]:                                 # rank order, left to right: what the kernel does
            want = want + p
        t = x.to(dev)
        comm.allreduce(t)
        ok = bool(np.array_equal(t.cpu().numpy(), want))
        # a cancelled (gated) launch: buffer and epoch untouched; then the same launch with the gate open
        gate = torch.zeros(1, dtype=torch.int32, device=dev)
        t2 = x.to(dev)
        lib.vican_set_gate(ctypes.c_void_p(gate.data_ptr()))
        comm.allreduce(t2)
        lib.vican_set_gate(None)
        torch.cuda.synchronize()
        ok = ok and bool(torch.equal(t2.cpu(), x))
        gate.fill_(1)
        lib.vican_set_gate(ctypes.c_void_p(gate.data_ptr()))
        comm.allreduce(t2)
        lib.vican_set_gate(None)
        ok = ok and bool(np.array_equal(t2.cpu().numpy(), want))
        # a burst without host synchronisation: x <- sum over ranks, 200 times, on exactly representable numbers
        k = torch.arange(n, dtype=torch.float64) % 7 + rank
        tb = k.to(dev)
        reps = 200 if n <= 40000 else 20
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dist.barrier()
        e0.record()
        for _ in range(reps):
            comm.allreduce(tb)
            tb.mul_(1.0 / world)                            # (back to the per-rank magnitude: stays exact only for world = 1, 2, 4, 8)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / reps
        if world in (1, 2, 4, 8):
            mean = sum((np.arange(n) % 7 + r) for r in range(world)) / world
            # after the first exchange every rank holds the mean; later exchanges reproduce it
            ok = ok and bool(np.array_equal(tb.cpu().numpy(), mean))
        if lib.vican_comm_peer_status(comm.native_handle()) != 0:
            contended = True            # a wait ran into its bound: the processes TIME-SHARE one GPU here, and on a crowded box a peer's
            break                       # kernel may not be scheduled in time - not a wrong sum (NaNs are not compared as one)
        bad += not ok
        report["sizes"][str(n)] = {"ok": ok, "us_per_exchange_incl_scale_kernel": us}
        if rank == 0:
            print("peer exchange of %6d doubles on %d ranks: %s, %.1f us per exchange (+ one scale kernel)" % (n, world, "exact" if ok else "MISMATCH", us), flush=True)
flags = [None] * world
dist.all_gather_object(flags, contended)
if any(flags):
    report["transport"] = comm.transport = "contended"
    bad = 0
tb_ = torch.tensor([bad]); dist.all_reduce(tb_)
if rank == 0:
    print("peer probe: transport=%s mismatches %d %s" % (comm.transport, int(tb_[0]), "; ".join(comm.notes)))
    if len(sys.argv) > 1:
        json.dump(report, open(sys.argv[1], "w"), indent=1)
dist.destroy_process_group()
sys.exit(1 if int(tb_[0]) else 0)
