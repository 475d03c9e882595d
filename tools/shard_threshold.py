#!/usr/bin/env python3
"""Where does sharding the timestep rows over N GPUs start to pay?  (vican_amd.bipgo.SHARD_MIN_EDGES)

One GPU can measure both sides of the trade: the single-rank solve of a graph of E merged edges, and the SHARDED schedule
(launch sequences, the peer exchange with the rank's own slot as its peer: tests/test_peer_gpu.py) on the E / N edges one of N
ranks would hold.  What it cannot measure is the link: `--link-us` (default 3) is ADDED per collective as the assumed extra
latency of an xGMI hop over the same-device exchange.  Prints, per graph size, the predicted speed-up of 2 / 4 / 8 ranks over one.

    python tools/shard_threshold.py [out.txt]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                                                # noqa: E402

from vican_amd import synth                                 # noqa: E402
from vican_amd.device import HipBackend, LocalGraph         # noqa: E402
from vican_amd.solver import Comm, RotationSolver, TranslationSolver      # noqa: E402

link_us = float(os.environ.get("LINK_US", 3.0))
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
lines = []


def say(s):
    print(s, flush=True)
    lines.append(s)


def solve_ms(C, T, cpt, comm, n=5):
    gr = synth.make_merged_graph_torch(C, T, cpt, dev, torch.float32, seed=0)
    g = LocalGraph(C, gr["row_ptr"], gr["col"], gr["blk"], gr["a"], gr["w"], gr["u"], gr["v"])
    del gr
    K = HipBackend(g)
    rot, tr = RotationSolver(K, comm), TranslationSolver(K, comm)

    def solve():
        rot.stats = dict(sweeps=0, lanczos_steps=[], evals=[], restarts=0, resid=[])
        tr.poll_every = 8
        rc, Rt = rot.run(4)
        tr.setup(rc, Rt)
        tr.solve(3 * (C + T))
        K.synchronize()
    for _ in range(3):
        solve()
    n0 = comm.n_allreduce
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        solve()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / n * 1e3
    return ms, (comm.n_allreduce - n0) / n, g.n_edges


say("# single-rank solve of E edges vs the sharded schedule on E/N edges (+ %.1f us per collective assumed for the link)" % link_us)
say("# C cameras, cpt cameras per timestep; ms per warm solve; predicted speed-up = t_single(E) / (t_sharded(E/N) + n_coll * link)")
shapes = ((340, 4), (500, 50), (1000, 250))
if os.environ.get("SHAPES"):
    shapes = tuple(tuple(int(v) for v in sh.split("x")) for sh in os.environ["SHAPES"].split(","))
for C, cpt in shapes:
    for E_target in (40_000, 250_000, 1_000_000, 2_000_000, 4_000_000, 8_000_000, 25_000_000):
        T = E_target // cpt
        if T < 64 or (cpt == 4 and E_target > 4_000_000) or (cpt == 250 and E_target < 1_000_000):
            continue
        if os.environ.get("VERBOSE"):
            print("... C=%d cpt=%d T=%d" % (C, cpt, T), flush=True)
        t1, _, E = solve_ms(C, T, cpt, Comm.single())
        row = "C=%4d cpt=%3d E=%9d  single %.3f ms |" % (C, cpt, E, t1)
        for N in (2, 4, 8):
            if os.environ.get("VERBOSE"):
                print("...   N=%d" % N, flush=True)
            tn, ncoll, _ = solve_ms(C, max(T // N, 8), cpt, Comm.single(force_sharded=True, peer=True))
            pred = tn + ncoll * link_us * 1e-3
            row += "  N=%d: %.3f ms (+%d coll) -> x%.2f" % (N, tn, ncoll, t1 / pred)
        say(row)
        torch.cuda.empty_cache()
if len(sys.argv) > 1:
    open(sys.argv[1], "w").write("\n".join(lines) + "\n")
