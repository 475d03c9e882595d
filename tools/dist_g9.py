#!/usr/bin/env python3
"""BASELINE configs[3], CHECKED: the large_shop-scale golden (g9: 340 cameras x 10 000 timesteps, 80 000 source edges) through
the drop-in API with the timesteps sharded over the ranks of the process group - poses against the REAL reference's
(tests/golden/g9_large_shop.npz) with the tolerances of the single-rank test, plus the sharded answer against the
single-rank answer computed inside the same processes.

  VICAN_DIST_BACKEND=gloo python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/dist_g9.py [out.json]

One-GPU boxes: the ranks share cuda:0 and the collectives go over gloo (RCCL refuses two ranks on one device); every
kernel launch, shard and collective call site is the one an N-GPU RCCL run uses."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np                                          # noqa: E402
import torch                                                # noqa: E402
import torch.distributed as dist                            # noqa: E402

import golden_cases as gc                                   # noqa: E402
from util import cg_sensitivity, expected, load_golden, pose_errors, translation_tol      # noqa: E402
from vican_amd import synth                                 # noqa: E402
from vican_amd.bipgo import bipartite_se3sync               # noqa: E402
from vican_amd.geometry import SE3, geodesic                # noqa: E402

os.environ.setdefault("VICAN_SHARD_MIN_EDGES", "0")        # (40 000 merged edges: the small-graph policy would replicate the solve)
backend = os.environ.get("VICAN_DIST_BACKEND", "nccl")
dist.init_process_group(backend)
rank, world = dist.get_rank(), dist.get_world_size()
ndev = torch.cuda.device_count()
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", rank)) % max(ndev, 1))
g = load_golden("g9_large_shop")
scene, flat = gc.build_flat(gc.LARGE_SHOP)
src = synth.edges_to_dict(flat, SE3)
cons = synth.constraints_from_scene(scene, SE3)
nr, nt, ff = (gc.CALLABLES[gc.LARGE_SHOP[k]] for k in ("noise_r", "noise_t", "filt"))
single = [dist.new_group([r]) for r in range(world)][rank]  # every rank also solves the whole problem alone (new_group is collective: same calls everywhere)
report, bad = {"world": world, "backend": backend, "devices": min(ndev, world)}, 0
for dt in ("float64", "float32"):
    exp = expected(g, "conjugate_gradient", dt)
    info, info1 = {}, {}
    res = bipartite_se3sync(src, cons, nr, nt, ff, gc.MAXITER, "conjugate_gradient", np.dtype(dt).type, info=info, group=dist.group.WORLD)
    one = bipartite_se3sync(src, cons, nr, nt, ff, gc.MAXITER, "conjugate_gradient", np.dtype(dt).type, info=info1, group=single)
    rot, tr = pose_errors(res, exp)
    R = np.stack([np.asarray(res[k].R(), dtype=np.float64) for k in res]); R1 = np.stack([np.asarray(one[k].R(), dtype=np.float64) for k in one])
    t = np.stack([res[k].t() for k in res]); t1 = np.stack([one[k].t() for k in one])
    _, iters = cg_sensitivity("g9_large_shop", dt)
    row = dict(rot_vs_reference_rad=rot, trans_vs_reference_m=tr, cg_iters=int(info["cg_iters"]), cg_reference=int(exp["cg_iters"]),
               reference_window=[min(int(iters.min()), int(exp["cg_iters"])), max(int(iters.max()), int(exp["cg_iters"]))],
               rot_vs_single_rank_rad=float(geodesic(R, R1).max()), trans_vs_single_rank_m=float(np.linalg.norm(t - t1, axis=1).max()),
               rotations_bit_identical_to_single_rank=bool(np.array_equal(R, R1)), cg_iters_single_rank=int(info1["cg_iters"]),
               tol_rot=5e-6 if dt == "float32" else 1e-7, tol_trans=min(translation_tol("g9_large_shop", dt), 2e-3),
               transport=info.get("transport"), policy=info.get("policy"), n_allreduce=info.get("n_allreduce"))
    lo, hi = row["reference_window"]            # golden run + the nine runs of the sensitivity fixture (the reference's own spread)
    ok = rot < row["tol_rot"] and tr < row["tol_trans"] and lo - 1 <= row["cg_iters"] <= hi + 1
    row["ok"] = bool(ok)
    bad += not ok
    report[dt] = row
    if rank == 0:
        print("g9 %s on %d ranks: rot %.2e rad, trans %.2e m vs the reference; cg %d (reference %d, its window %d..%d); vs single rank: "
              "rot %.1e trans %.1e (cg %d); transport %s, %s all-reduces%s" % (dt, world, rot, tr, row["cg_iters"], row["cg_reference"], lo, hi,
                                                  row["rot_vs_single_rank_rad"], row["trans_vs_single_rank_m"], row["cg_iters_single_rank"],
                                                  row["transport"], row["n_allreduce"], "" if ok else "   <-- MISMATCH"), flush=True)
tb = torch.tensor([bad]); dist.all_reduce(tb)
if rank == 0:
    print("dist g9: mismatches", int(tb[0]))
    if len(sys.argv) > 1:
        json.dump(report, open(sys.argv[1], "w"), indent=1)
dist.destroy_process_group()
sys.exit(1 if int(tb[0]) else 0)
