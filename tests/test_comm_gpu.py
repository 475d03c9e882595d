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
