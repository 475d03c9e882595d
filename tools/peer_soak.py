#!/usr/bin/env python3
"""Soak of the peer exchange between processes: N all-reduces of alternating sizes (both kernel forms, both buffer parities, the
slot-reuse order) back to back without host synchronisation, EVERY result compared on the device with its closed form (rank r
sends (r + 1)(pattern + iteration mod 97): exactly representable, the rank-ordered sum is W (W + 1) / 2 times the same).

  VICAN_DIST_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 4 --master-addr 127.0.0.1 tools/peer_soak.py [N=200000]

Prints "peer soak: transport=<t> exchanges <N> wrong <n> timed-out waits <m>" on rank 0."""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                # noqa: E402
import torch.distributed as dist                            # noqa: E402

from vican_amd import _lib                                  # noqa: E402
from vican_amd.solver import Comm                           # noqa: E402

dist.init_process_group(os.environ.get("VICAN_DIST_BACKEND", "nccl"))
rank, world = dist.get_rank(), dist.get_world_size()
ndev = torch.cuda.device_count()
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank)) % max(ndev, 1))
dev = torch.device("cuda", torch.cuda.current_device())
lib = _lib.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
comm = Comm(transport=os.environ.get("VICAN_COMM", "peer"))
comm._setup_native(dev)
wrong = torch.zeros((), dtype=torch.int64, device=dev)
status = -1
t0 = time.perf_counter()
if comm.transport == "peer":
    sizes = (1, 96, 3 * 340 + 96, 9 * 340, 512, 9 * 1000, 3 * 1000 + 96, 70000, 7)
    nmax = max(sizes)
    base = (torch.arange(nmax, dtype=torch.float64, device=dev) % 13) + 0.25
    buf = torch.empty(nmax, dtype=torch.float64, device=dev)
    scale = float(world * (world + 1) // 2)
    dist.barrier()
    for it in range(N):
        n = sizes[it % len(sizes)]
        pat = base[:n] + float(it % 97)
        b = buf[:n]
        torch.mul(pat, float(rank + 1), out=b)
        comm.allreduce(b)
        wrong += (b != pat * scale).any()
        if it % 20000 == 19999:
            torch.cuda.synchronize()
            if lib.vican_comm_peer_status(comm.native_handle()) != 0:
                break
    torch.cuda.synchronize()
    status = int(lib.vican_comm_peer_status(comm.native_handle()))
tot = torch.tensor([int(wrong), max(status, 0)])
dist.all_reduce(tot)
if rank == 0:
    print("peer soak: transport=%s ranks %d exchanges %d wrong %d timed-out waits %d (%.1f us per exchange with fill and check)" % (
        comm.transport, world, N, int(tot[0]), int(tot[1]), (time.perf_counter() - t0) / max(N, 1) * 1e6), flush=True)
dist.destroy_process_group()
sys.exit(1 if int(tot[0]) else 0)
