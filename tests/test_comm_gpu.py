"""vican_comm_* (include/vican_hip.h, csrc/vican_comm.hip): RCCL loaded at run time by the C library, a communicator created
from a unique id, the in-place f64 sum-all-reduce enqueued on the caller's stream, and the composite vican_block_op_z_comm.
A 1-GPU box can only form a group of ONE rank (RCCL refuses two ranks on one device): that exercises the loading, the
communicator life cycle and the call path; the multi-rank arithmetic is ncclAllReduce's."""
import ctypes as C

import numpy as np
import pytest
import torch

from vican_amd import _lib

pytestmark = pytest.mark.gpu


def test_rccl_communicator_life_cycle_and_allreduce_call_path():
    lib = _lib.load()
    torch.cuda.set_device(0)
    buf = C.create_string_buffer(128)
    assert lib.vican_comm_unique_id(buf) == 0, lib.vican_last_error()
    assert any(buf.raw)                                          # an id was written
    comm = C.c_void_p()
    assert lib.vican_comm_create(0, 1, buf, C.byref(comm)) == 0, lib.vican_last_error()
    x = torch.arange(1000, dtype=torch.float64, device="cuda:0")
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert lib.vican_comm_allreduce_sum(comm, C.c_void_p(x.data_ptr()), x.numel(), stream) == 0, lib.vican_last_error()
    torch.cuda.synchronize()
    assert torch.equal(x.cpu(), torch.arange(1000, dtype=torch.float64))
    assert lib.vican_comm_allreduce_sum(comm, None, 3, stream) == _lib.ERR_ARG
    assert lib.vican_comm_create(1, 1, buf, C.byref(C.c_void_p())) == _lib.ERR_ARG          # rank out of range
    assert lib.vican_comm_destroy(comm) == 0 and lib.vican_comm_destroy(None) == 0


def test_block_op_with_the_collective_behind_one_call():
    """vican_block_op_z_comm on a one-rank communicator = vican_block_op_z (bit for bit)."""
    from test_kernels_gpu import make_backends
    lib = _lib.load()
    H, N, g = make_backends(23, 120, 2, 9, 5, np.float64, "wave")
    rng = np.random.default_rng(0)
    lamT = H.from_numpy(np.tile(np.eye(3).reshape(9), (g.n_time, 1)) * rng.uniform(0.5, 2.0, (g.n_time, 1)))
    H.set_duals(lamT)
    x = H.from_numpy(rng.standard_normal((3 * g.n_cam, 3)))
    z0, z1 = H.empty(3 * g.n_cam, 3), H.empty(3 * g.n_cam, 3)
    H.block_op(lamT, x, z0)
    buf = C.create_string_buffer(128)
    assert lib.vican_comm_unique_id(buf) == 0
    comm = C.c_void_p()
    assert lib.vican_comm_create(0, 1, buf, C.byref(comm)) == 0
    p = lambda t: C.c_void_p(t.data_ptr())
    rc = lib.vican_block_op_z_comm(C.byref(g.desc), p(lamT), p(x), p(H.zpart), p(g.fx), p(z1), comm, C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, lib.vican_last_error()
    torch.cuda.synchronize()
    assert torch.equal(z0, z1)
    lib.vican_comm_destroy(comm)


def _one_rank_comm(lib, force=True):
    buf = C.create_string_buffer(128)
    assert lib.vican_comm_unique_id(buf) == 0, lib.vican_last_error()
    comm = C.c_void_p()
    assert lib.vican_comm_create(0, 1, buf, C.byref(comm)) == 0, lib.vican_last_error()
    assert lib.vican_comm_force_enqueue(comm, int(force)) == 0
    return comm


@pytest.mark.parametrize("n_cam", [340, 1000])
def test_forced_one_rank_allreduce_executes_rccl_and_keeps_the_bits(n_cam):
    """vican_comm_force_enqueue: ncclAllReduce is REALLY called on the one-rank communicator (as ncclAvg = sum x 1/1; RCCL
    elides an in-place ncclSum on one rank) - RCCL's one-rank kernel runs on the caller's stream, in stream order behind the
    kernel that produced the message; the message sizes of the solve (3C x 3 per operator application, 3C + 1 and one scalar
    per CG iteration) keep their bits.  The timing of the same calls: tools/rccl_onerank.py -> profiles/r05_rccl_onerank.txt."""
    lib = _lib.load()
    torch.cuda.set_device(0)
    comm = _one_rank_comm(lib)
    rng = np.random.default_rng(n_cam)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for n in (9 * n_cam, 3 * n_cam + 1, 3 * n_cam + 2, 1):
        h = rng.standard_normal(n) * 10.0 ** rng.integers(-20, 20, n)
        x = torch.from_numpy(h).to("cuda:0")
        y = x * 2.0                                   # a kernel in front of the collective on the same stream
        assert lib.vican_comm_allreduce_sum(comm, C.c_void_p(y.data_ptr()), n, stream) == 0, lib.vican_last_error()
        z = y * 0.5                                   # ... and one behind it
        torch.cuda.synchronize()
        assert np.array_equal(z.cpu().numpy().view(np.int64), h.view(np.int64))
    assert lib.vican_comm_force_enqueue(None, 1) == _lib.ERR_ARG
    assert lib.vican_comm_destroy(comm) == 0


def _large_shop_problem(dt=np.float32):
    from vican_amd import frontend, synth
    scene = synth.make_scene(n_cam=340, n_time=10000, n_marker=6, seed=0)
    flat = synth.make_camera_edges(scene, cpt=4, mpv=2, sigma_r=1e-3, sigma_t=1e-3, seed=1)
    from vican_amd.geometry import SE3
    cons = synth.constraints_from_scene(scene, SE3)
    cams = flat["cam_key"].astype(str)
    tm = np.char.partition(flat["marker_key"].astype(str), "_")
    ones = np.ones(len(cams))
    return frontend.flatten_arrays(cams, tm[:, 0], tm[:, 2], flat["R"], flat["t"], ones, ones, cons, dt)


@pytest.mark.parametrize("dt", [np.float32, np.float64])
def test_sharded_schedule_with_rccl_collectives_on_one_rank(dt):
    """One full large_shop-sized solve on the SHARDED schedule (launch sequences, host-issued all-reduces, scipy's two-message
    CG) with every all-reduce a real ncclAllReduce on a one-rank RCCL communicator held by the C library
    (Comm.single(force_sharded=True, native=True)): bit-equal to the same schedule with identity collectives (RCCL did not
    touch a bit, in stream order) and - translations: the CG is the same recurrence on the same kernels' sums - to the plain
    single-rank solve within the eigen-solver's tolerance."""
    from vican_amd.bipgo import solve_problem
    from vican_amd.solver import Comm
    torch.cuda.set_device(0)
    prob = _large_shop_problem(dt)
    info_r, info_i, info_p = {}, {}, {}
    comm = Comm.single(force_sharded=True, native=True)
    out_r = solve_problem(prob, 4, "conjugate_gradient", dt, info=info_r, comm=comm)
    n_rccl = comm.n_allreduce
    ident = Comm.single(force_sharded=True)
    ident.gateable = False                 # (the schedule of collectives that cannot be gated: no speculation through them)
    out_i = solve_problem(prob, 4, "conjugate_gradient", dt, info=info_i, comm=ident)
    out_p = solve_problem(prob, 4, "conjugate_gradient", dt, info=info_p)
    print("large_shop %s: %d ncclAllReduce calls in the solve (identity run: %d), cg %d / %d / %d, sweeps %d / %d" % (
        np.dtype(dt).name, n_rccl, ident.n_allreduce, info_r["cg_iters"], info_i["cg_iters"], info_p["cg_iters"], info_r["sweeps"], info_p["sweeps"]))
    assert n_rccl == ident.n_allreduce and n_rccl >= info_r["sweeps"] - 4 + 2 * info_r["cg_iters"]
    for a, b in zip(out_r, out_i):
        assert np.array_equal(a, b)                                   # RCCL in the loop: not one bit moved
    # against the plain schedule (cooperative kernels, speculation, resident CG): the same answer to the solver's own tolerances
    from vican_amd.geometry import geodesic
    assert float(geodesic(out_r[0], out_p[0]).max()) < (1e-9 if dt == np.float64 else 2e-6)
    assert float(geodesic(out_r[1], out_p[1]).max()) < (1e-9 if dt == np.float64 else 2e-6)
    assert abs(info_r["cg_iters"] - info_p["cg_iters"]) <= 1
    assert float(np.abs(out_r[2] - out_p[2]).max()) < (1e-7 if dt == np.float64 else 1e-3)
