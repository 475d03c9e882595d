#!/usr/bin/env python3
"""The recovery path of the peer exchange: N ranks solve a sharded problem over the exchange, ONE rank then reports a timed-out wait
(injected: vican_comm_peer_inject_fault), and the next solve must notice it on EVERY rank (solver.Comm.healthy: one collective flag),
switch the whole group to the fall-back transport and return the same poses.

    VICAN_DIST_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/peer_demote_probe.py"""
import os
import sys
import warnings

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("VICAN_SHARD_MIN_EDGES", "0")
import numpy as np                                          # noqa: E402
import torch                                                # noqa: E402
import torch.distributed as dist                            # noqa: E402

from vican_amd import _lib, frontend, synth                 # noqa: E402
from vican_amd.bipgo import solve_problem                   # noqa: E402
from vican_amd.geometry import SE3                          # noqa: E402
from vican_amd.solver import Comm                           # noqa: E402

dist.init_process_group(os.environ.get("VICAN_DIST_BACKEND", "gloo"))
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank)) % max(torch.cuda.device_count(), 1))
lib = _lib.load()
scene = synth.make_scene(n_cam=12, n_time=101, n_marker=3, seed=5)
flat = synth.make_camera_edges(scene, cpt=3, mpv=2, sigma_r=1e-3, sigma_t=1e-3, seed=6)
src = synth.edges_to_dict(flat, SE3); cons = synth.constraints_from_scene(scene, SE3)
unit, keep = (lambda e: 1.0), (lambda e: True)
prob = frontend.flatten(src, cons, unit, unit, keep, np.float64)
bad = 0
info = {}
out0 = solve_problem(prob, 4, "conjugate_gradient", np.float64, group=dist.group.WORLD, info=info)
first = info.get("transport")
comm = Comm(dist.group.WORLD)
comm._setup_native(torch.device("cuda", torch.cuda.current_device()))
if first == "peer":
    comm._verified = False                                   # (ask the next solve for its collective health check again)
    if rank == world - 1:
        _lib.check(lib.vican_comm_peer_inject_fault(comm.native_handle()), "vican_comm_peer_inject_fault")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        info = {}
        out1 = solve_problem(prob, 4, "conjugate_gradient", np.float64, group=dist.group.WORLD, info=info)
    bad += info.get("transport") == "peer"                   # every rank must have left the exchange
    bad += not any("peer exchange switched off" in str(x.message) for x in w)
    bad += not all(np.allclose(a, b, rtol=0, atol=1e-9) for a, b in zip(out0, out1))
    info = {}
    out2 = solve_problem(prob, 4, "conjugate_gradient", np.float64, group=dist.group.WORLD, info=info)   # a fresh Comm of the group: still demoted
    bad += info.get("transport") == "peer"
    bad += not all(np.array_equal(a, b) for a, b in zip(out1, out2))
    if rank == 0:
        print("peer demotion: first transport %s, after the injected fault %s; poses moved by %.1e" % (
            first, info.get("transport"), max(float(np.abs(a - b).max()) for a, b in zip(out0, out1))), flush=True)
tb = torch.tensor([bad]); dist.all_reduce(tb)
if rank == 0:
    print("peer demote probe: first transport=%s mismatches %d" % (first, int(tb[0])))
dist.destroy_process_group()
sys.exit(1 if int(tb[0]) else 0)
