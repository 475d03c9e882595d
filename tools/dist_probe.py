#!/usr/bin/env python3
"""Two ranks sharing one GPU over gloo: sharded solves of tiny and uneven problems (including a rank without rows)
against the single-rank solve of the same problem.
  VICAN_DIST_BACKEND=gloo python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/dist_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist
from vican_amd import frontend, synth
from vican_amd.bipgo import solve_problem
from vican_amd.geometry import SE3, geodesic

os.environ.setdefault("VICAN_SHARD_MIN_EDGES", "0")        # (tiny graphs: the small-graph policy would replicate the solves)
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
torch.cuda.set_device(0)
bad = 0
# one-rank groups, created by EVERY rank in the same order (torch.distributed.new_group is collective)
singles = [dist.new_group([r]) for r in range(world)]
# (the last case again with camera TILES of 5 - the path graphs beyond 1024 cameras take: the tiles' rows are re-ordered inside
#  each rank's graph, device.TiledGraph.row_perm, and handed back in the caller's order)
for seed, (C, T) in enumerate([(3, 1), (2, 2), (4, 3), (5, 7), (8, 40), (12, 101), (12, 101)]):
    if seed == 6:
        os.environ["VICAN_TILE_CAMS"] = "5"
    scene = synth.make_scene(n_cam=C, n_time=T, n_marker=3, seed=seed)
    flat = synth.make_camera_edges(scene, cpt=min(C, 3), mpv=2, sigma_r=1e-3, sigma_t=1e-3, seed=seed + 1)
    src = synth.edges_to_dict(flat, SE3); cons = synth.constraints_from_scene(scene, SE3)
    unit, keep = (lambda e: 1.0), (lambda e: True)
    for dt in (np.float64, np.float32):
        prob = frontend.flatten(src, cons, unit, unit, keep, dt)
        info = {}
        Rc, Rt, pc, pt = solve_problem(prob, 4, "conjugate_gradient", dt, group=dist.group.WORLD, info=info)
        single = singles[rank]                          # every rank also solves the whole problem alone
        Rc1, Rt1, pc1, pt1 = solve_problem(prob, 4, "conjugate_gradient", dt, group=single)
        rot = max(float(geodesic(Rc, Rc1).max()), float(geodesic(Rt, Rt1).max()))
        tr = max(float(np.abs(pc - pc1).max()), float(np.abs(pt - pt1).max()))
        ok = rot < (1e-7 if dt == np.float64 else 1e-5) and tr < (1e-6 if dt == np.float64 else 1e-3)
        bad += not ok
        ok = ok and (seed != 6 or info.get("layout") == "tiled")
        if rank == 0:
            print("C=%d T=%d %s %s: sharded vs single rot %.1e trans %.1e cg %s transport %s %s" % (C, T, np.dtype(dt).name, info.get("layout"), rot, tr, info.get("cg_iters"), info.get("transport"), "" if ok else "  <-- MISMATCH"))
t = torch.tensor([bad]); dist.all_reduce(t)
if rank == 0:
    print("dist probe: mismatches", int(t[0]))
dist.destroy_process_group()
