"""The numeric half of the front-end on the device (vican_merge.hip): constraint products, weighting, segment sums and the
J^T J diagonal of bipgo.py:203-221, 445-468 - bit-identical to frontend.merge_host.  (Split out of device.py in round 6.)"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _lib
from ._rt import _ptr, _stream, download, upload


def merge_edges(ix, R, t, k_r, k_t, dtype=np.float32, device=None, kr_f32=None):
    """frontend.merge_host ON THE DEVICE (vican_merge.hip; reference bipgo.py:203-221, 445-469): the per-edge arrays go to
    HBM once (124 B per source edge), the merged timestep-major CSR problem never leaves it.  Same bits as merge_host
    (tests/test_merge_gpu.py).  Returns a frontend.Problem whose numeric fields are device tensors (`on_device`)."""
    from . import frontend
    lib = _lib.load()
    if not torch.cuda.is_available():
        raise _lib.VicanError("vican_amd needs a GPU (MI355X); there is no CPU fallback")
    dev = device or torch.device("cuda", torch.cuda.current_device())
    n, C_, T_ = int(ix.n), len(ix.cam_names), len(ix.time_names)
    t_h = np.asarray(t, dtype=np.float64).reshape(n, 3)
    kt_h = np.asarray(k_t, dtype=np.float64)
    f64, i32 = torch.float64, torch.int32
    cam, tim, mk, Rd, td, krd, ktd, CmT, qtau = upload(dev, [
        (ix.ci, i32), (ix.ti, i32), (ix.mi, i32), (np.asarray(R).reshape(n, 9), f64), (t_h, f64), (np.asarray(k_r).reshape(n), f64),
        (kt_h, f64), (np.asarray(ix.CmT).reshape(-1, 9), f64), (np.asarray(ix.qtau).reshape(-1, 3), f64)])
    # (kr_f32: where numpy forms k_r * R in float32 - frontend.f32_product_mask)
    flags = None if kr_f32 is None else upload(dev, [(np.ascontiguousarray(kr_f32, dtype=np.uint8).reshape(n), torch.uint8)])[0]
    wsb = int(lib.vican_merge_ws_bytes(n, C_, T_))
    if wsb < 0:
        raise _lib.VicanError("vican_merge_ws_bytes failed")
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev)
    e = lambda *shape, dt=torch.float64: torch.empty(*shape, dtype=dt, device=dev)
    nm, row_ptr, col = e(1, dt=torch.int32), e(T_ + 1, dt=torch.int32), e(n, dt=torch.int32)
    blk, a, w, u, v, deg_c, deg_t = e(n, 9), e(n), e(n), e(n, 3), e(n, 3), e(C_), e(T_)
    storage = _lib.STORE_F32 if np.dtype(dtype) == np.float32 else _lib.STORE_F64
    _lib.check(lib.vican_merge_edges(n, C_, T_, int(CmT.shape[0]), storage, _ptr(cam), _ptr(tim), _ptr(mk), _ptr(Rd), _ptr(td), _ptr(krd),
                                     _ptr(flags), _ptr(ktd), _ptr(CmT), _ptr(qtau), _ptr(ws), wsb, _ptr(nm), _ptr(row_ptr), _ptr(col), _ptr(blk), _ptr(a),
                                     _ptr(w), _ptr(u), _ptr(v), _ptr(deg_c), _ptr(deg_t), _stream()), "vican_merge_edges")
    E = int(nm.item())                                           # (the one synchronisation: sizes the outputs)
    p = frontend.Problem()
    p.on_device = True
    p.root, p.n_src = ix.root, n
    p.cam_names, p.time_names, p.tnodes = ix.cam_names, ix.time_names, ix.tnodes
    p.tnode_of_cam, p.tnode_of_time = ix.tnode_of_cam, ix.tnode_of_time
    p.row_ptr, p.col, p.blk, p.a, p.w, p.u, p.v = row_ptr, col[:E], blk[:E], a[:E], w[:E], u[:E], v[:E]
    p.deg_c, p.deg_t = deg_c, deg_t
    p.row_ptr_host, p.col_host = download([row_ptr, p.col])
    p.src_cam, p.src_time, p.src_t, p.src_qtau, p.src_kt = ix.ci, ix.ti, t_h, ix.qtau[ix.mi], kt_h
    return p
