"""The schedule hints a solver object keeps between solves of the same graph (time series, benchmark loop): where the Ritz checks are
placed.  They never decide convergence - every solve ends on a passed device-side check of the same rule - only how many checks and
steps are spent getting there (vican_amd/solver.py RotationSolver.spectral: pred_steps, the probe one step below it)."""
import numpy as np
import pytest
import torch

from vican_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("dt", [torch.float32, torch.float64])
@pytest.mark.parametrize("shape", [(340, 10000, 4), (100, 3000, 5)])
def test_repeated_solves_settle_on_the_smallest_step_counts_and_the_same_answer(shape, dt):
    from vican_amd.device import HipBackend, LocalGraph
    from vican_amd.solver import Comm, RotationSolver
    dev = torch.device("cuda:0")
    C, T, cpt = shape
    gr = synth.make_merged_graph_torch(C, T, cpt, dev, dt, seed=0)
    K = HipBackend(LocalGraph(C, gr["row_ptr"], gr["col"], gr["blk"], gr["a"], gr["w"], gr["u"], gr["v"]))
    rot = RotationSolver(K, Comm())
    assert rot.small_graph
    hist, outs = [], []
    for _ in range(9):
        rot.stats = dict(sweeps=0, lanczos_steps=[], evals=[], restarts=0, resid=[])
        rc, Rt = rot.run(4)
        K.synchronize()
        hist.append(list(rot.stats["lanczos_steps"]))
        outs.append((rc.clone(), Rt.clone()))
        assert rot.stats["restarts"] == 0
    # capture-sized graphs check every four steps on their first solve; later solves walk down one step at a time and stop at the
    # first count whose check fails (that solve takes the step back): never more steps than the solve before, settled well before
    # the ninth solve, and below the first solve's count
    tot = [sum(h) for h in hist]
    assert all(b <= a for a, b in zip(tot, tot[1:])), hist
    assert hist[-1] == hist[-2] == hist[-3] and tot[-1] < tot[0], hist
    assert all(rot._probe_done.get(i, False) or rot.pred_steps[i] == 1 for i in range(4)), (rot._probe_done, rot.pred_steps)
    # the reference for "smallest": the same solver checking after EVERY step from the start (no hints)
    ref = RotationSolver(K, Comm())
    ref.min_steps = ref.warm_min_steps = ref.check_every = 1
    ref._probe_done = {i: True for i in range(64)}
    rc_ref, Rt_ref = ref.run(4)
    K.synchronize()
    assert all(a <= b for a, b in zip(hist[-1], ref.stats["lanczos_steps"])), (hist[-1], ref.stats["lanczos_steps"])
    # ... and the answer is the same answer: within the eigen tolerance (f64) / the rounding floor of f32 blocks of the first solve's
    tol = 5e-7 if dt == torch.float32 else 1e-9
    for a, b in zip(outs[0], outs[-1]):
        assert float((a - b).abs().max()) < tol
    assert float((outs[-1][0] - rc_ref).abs().max()) < tol and float((outs[-1][1] - Rt_ref).abs().max()) < tol


def test_large_graphs_are_checked_step_by_step_and_do_not_probe():
    from vican_amd.device import HipBackend, LocalGraph
    from vican_amd.solver import Comm, RotationSolver
    dev = torch.device("cuda:0")
    C, T, cpt = 500, 20000, 120
    gr = synth.make_merged_graph_torch(C, T, cpt, dev, torch.float32, seed=0)
    K = HipBackend(LocalGraph(C, gr["row_ptr"], gr["col"], gr["blk"], gr["a"], gr["w"], gr["u"], gr["v"]))
    rot = RotationSolver(K, Comm())
    assert not rot.small_graph
    hist = []
    for _ in range(4):
        rot.stats = dict(sweeps=0, lanczos_steps=[], evals=[], restarts=0, resid=[], n_check=0)
        rot.run(4)
        K.synchronize()
        hist.append((list(rot.stats["lanczos_steps"]), rot.stats["n_check"]))
    assert all(h[0] == hist[0][0] for h in hist), hist            # the first solve's counts are already the smallest
    assert all(h[1] == 4 for h in hist[1:]), hist                  # one check per eigen-solve once the counts are remembered
