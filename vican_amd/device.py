"""Device side of the solver: thin wrappers over the C ABI.

``HipBackend`` exposes each entry point of ``include/vican_hip.h`` on torch tensors - the kernel interface that
``vican_amd/solver.py`` drives (one instance per rank / ``layout.LocalGraph``).  PyTorch is only the allocator / stream
provider here - every numerical step is a hand-written HIP kernel.  No CPU fallback: without a GPU construction raises.

Round 6 split the former 1 580-line module: graph planning lives in ``layout.py`` (``LocalGraph``), the camera-tiled graph and
backend in ``tiled.py``, the device merge of the front-end in ``merge.py``, the run-time plumbing (staging buffer, status pool,
abort word, stream look-up) in ``_rt.py``.  Their public names are re-exported here: ``from vican_amd.device import ...`` keeps
working for every name it served before.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import math
import os

import numpy as np
import torch

from . import _lib
from ._rt import (STREAM_NT_BYTES, X_BOUND, _STATUS_POOL, _ptr, _stream, barrier_abort_word, download, n_cu, upload)      # noqa: F401
from .layout import LocalGraph, _Layout, _wave_params                                                                      # noqa: F401


class HipBackend:
    """Kernel interface used by ``solver.py`` (one instance per rank / LocalGraph)."""

    def __init__(self, graph: LocalGraph):
        self.lib = _lib.load()
        self.g = graph
        self.dev = graph.device
        self.C, self.T = graph.n_cam, graph.n_time
        self.storage_f64 = graph.storage_dtype == torch.float64
        # folding the sweep's slabs inside the (<= 32 workgroup) camera-side kernel pays off while there are few of them:
        # measured -3..-8 % of the rotation stage at 40 slabs (large_shop), +1.3 % at 256 (stress) - tools/ab_fold.py
        self.fold_in_step_ok = graph.n_wg <= 64
        self.layout = graph.layout
        self._gref = C.byref(graph.desc)
        self._gref_t = C.byref(graph.desc_t)            # layout of the translation arrays (block layout)
        self.tl = graph.tl
        nwg = max(graph.n_wg, graph.tl.n_wg)
        self.zpart = torch.empty(nwg * 9 * self.C, dtype=torch.float64, device=self.dev)   # f64 or i64 slabs
        self.pq_part = torch.empty(max(nwg, 1), dtype=torch.float64, device=self.dev)    # (nwg: the larger of the two layouts)
        self.rr_part = torch.zeros(1536, dtype=torch.float64, device=self.dev)      # 3 x CG_PARTS: r.r, max |r_t|, max |p_t|
        self.ws = torch.zeros(1024, dtype=torch.float64, device=self.dev)
        # adds into one fixed-point accumulator by one workgroup: its rows (cameras), a chunk (rows)
        self.n_add = float(max(graph.tl.rows_per_wg_max, graph.tl.slots) + 1)
        self._status_host = {}
        self.coop_cam_step = os.environ.get("VICAN_COOP", "1") != "0"   # one cooperative kernel per camera-side Lanczos step (False: launch sequence)
        self._coop_ws, self._coop_sync, self._gram_ws = None, None, None
        self.coop_failures = []
        barrier_abort_word()
        self._w_scaled, self._cg_w, self._cg_wmax = None, graph.w_cg, getattr(graph, "wmax", None)
        # layout the CG sweep runs on: the rotation layout if it is a wave layout, else the translation block layout
        cgl = graph.rot if graph.layout == "wave" else graph.tl
        self._gref_cg, self.cgl = (self._gref if graph.layout == "wave" else self._gref_t), cgl
        self.n_add_cg = float(max(cgl.rows_per_wg_max, cgl.slots) + 1)

    # -- allocation helpers -------------------------------------------------
    def empty(self, *shape, dtype=torch.float64):
        return torch.empty(*shape, dtype=dtype, device=self.dev)

    def zeros(self, *shape, dtype=torch.float64):
        return torch.zeros(*shape, dtype=dtype, device=self.dev)

    def from_numpy(self, a):
        return torch.from_numpy(np.ascontiguousarray(a)).to(self.dev)

    def pinned(self, *shape):
        """Page-locked host buffer for small asynchronous device<->host copies."""
        return torch.zeros(*shape, dtype=torch.float64, pin_memory=True)

    def __del__(self):
        try:                                        # status slots back to the pool (post_status)
            for n, slot in getattr(self, "_status_host", {}).items():
                slot[1].synchronize()
                _STATUS_POOL.setdefault((str(self.dev), n), []).append(slot)
            self._status_host = {}
        except Exception:                           # noqa: BLE001  (interpreter shutdown)
            pass

    def synchronize(self):
        torch.cuda.current_stream().synchronize()

    # -- cooperative kernels: bounded grid barriers (vican_common.h vican_grid_sync) -------------------------------------
    def barrier_aborted(self):
        """True when a grid barrier of a cooperative kernel gave up since the last cooperative_failed(): the outputs of
        that launch (and everything computed from them) are undefined."""
        return bool(barrier_abort_word()[0].item())

    def cooperative_failed(self, which):
        """Stop using the cooperative kernels on this backend (the launch-sequence paths compute the same thing) and
        re-arm the barrier words; called after an aborted barrier or a launcher's co-residency refusal."""
        import warnings
        warnings.warn("%s: the grid of a cooperative kernel could not run co-resident (device shared or partly masked); "
                      "continuing on the launch-sequence path" % which, RuntimeWarning, stacklevel=3)
        self.synchronize()
        self.coop_failures.append(which)
        self.coop_cam_step, self._cgres_ok, self.fold_in_step_ok = False, False, False
        barrier_abort_word().zero_()
        if self._coop_sync is not None:
            self._coop_sync.zero_()
        if getattr(self, "_cgres_ws", None) is not None:
            self._cgres_ws.zero_()

    def _ck(self, rc, what):
        return _lib.check(rc, what)

    def _fx_finish(self):
        self._ck(self.lib.vican_fx_finish(_ptr(self.g.fx), X_BOUND, float(self.g.rows_per_wg_sweep + 1),
                                          self.g.desc.storage, _stream()),
                 "vican_fx_finish")

    # -- rotation stage -----------------------------------------------------
    def init_duals(self, lamT_inv, cam_deg):
        """lamT_inv[t] = I/d_t from the stored row sums; cam_deg = local camera sums of a."""
        cam_deg.copy_(self.g.cam_sum_a)
        self._ck(self.lib.vican_init_duals(self.T, _ptr(self.g.row_sum_a), _ptr(self.g.rnorm), _ptr(lamT_inv),
                                           _ptr(self.g.fx), _stream()), "vican_init_duals")
        self._fx_finish()

    def set_duals(self, lamT_inv):
        """Caller-supplied duals: refresh the fixed-point bound before block_op."""
        self._ck(self.lib.vican_duals_bound(self.T, _ptr(lamT_inv), _ptr(self.g.rnorm), _ptr(self.g.fx), _stream()),
                 "vican_duals_bound")
        self._fx_finish()

    def scaled_identity(self, scale, out):
        self._ck(self.lib.vican_scaled_identity(scale.numel(), _ptr(scale), _ptr(out), _stream()), "vican_scaled_identity")

    @staticmethod
    def make_launch_timers(n):
        """n pairs of HIP events for time_next_sweep, created up front (torch creates the underlying hipEvent_t on the
        first record) so that binding one to a launch puts NO extra command into the stream."""
        pairs = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for e0, e1 in pairs:
            e0.record(); e1.record()
        torch.cuda.current_stream().synchronize()
        return pairs

    def time_next_sweep(self, pair):
        """Bind a pair of events to the NEXT edge-sweep launch (vican_set_launch_events): after the stream has passed
        it, pair[0].elapsed_time(pair[1]) is that kernel's own duration on the device."""
        self._ck(self.lib.vican_set_launch_events(C.c_void_p(pair[0].cuda_event), C.c_void_p(pair[1].cuda_event)),
                 "vican_set_launch_events")

    def block_op_raw(self, lamT_inv, x):
        """Only the sweep kernel (the benchmark binds HIP events to this launch)."""
        self._ck(self.lib.vican_block_op(self._gref, _ptr(lamT_inv), _ptr(x), _ptr(self.zpart), _ptr(self.g.fx), _stream()),
                 "vican_block_op")

    # -- composites: several launches behind one host call (the per-call Python/ctypes cost
    #    exceeded the run time of the small camera-side kernels) ----------------------------
    def block_op(self, lamT_inv, x, z_out):
        """z_out[3C,3] = local slab-reduced  P x  (caller all-reduces across ranks)."""
        self._ck(self.lib.vican_block_op_z(self._gref, _ptr(lamT_inv), _ptr(x), _ptr(self.zpart), _ptr(self.g.fx),
                                           _ptr(z_out), _stream()), "vican_block_op_z")

    def block_op_comm(self, lamT_inv, x, z_out, comm):
        """z_out = P x summed over the ranks of `comm` (solver.Comm): sweep, slab fold and the all-reduce behind ONE host call
        where the communicator lives in the C library (include/vican_hip.h: vican_block_op_z_comm); else block_op + comm.allreduce."""
        h = comm.native_handle() if hasattr(comm, "native_handle") else None
        if h is None and z_out.is_cuda and not getattr(comm, "_native_tried", True):
            comm._setup_native(z_out.device)                     # (first device message of the group: a collective set-up)
            h = comm.native_handle()
        if h is None or type(self).block_op is not HipBackend.block_op or z_out.numel() > getattr(comm, "PEER_MAX_DOUBLES", 0):
            self.block_op(lamT_inv, x, z_out)
            comm.allreduce(z_out)
            return
        self._ck(self.lib.vican_block_op_z_comm(self._gref, _ptr(lamT_inv), _ptr(x), _ptr(self.zpart), _ptr(self.g.fx),
                                                _ptr(z_out), h, _stream()), "vican_block_op_z_comm")
        comm.n_allreduce += 1

    def node_degrees(self, out):
        """Weighted degrees of all C+T nodes (cameras first) - bipgo.py:95."""
        out[: self.C].copy_(self.g.cam_sum_a)
        out[self.C:].copy_(self.g.row_sum_a[: self.T])

    def bip_scales(self):
        """Fixed-point scales for bip_apply (instead of init_duals / set_duals)."""
        self._ck(self.lib.vican_bip_scales(_ptr(self.g.fx), X_BOUND, float(self.g.rows_per_wg_sweep + 1), self.g.desc.storage,
                                           _stream()), "vican_bip_scales")

    def bip_apply(self, x, z_out):
        """z_out = R~ x on all C+T nodes (x, z_out: [3(C+T), 3], cameras first) - one pass over the blocks."""
        off = 8 * 9 * self.C
        self._ck(self.lib.vican_bip_apply(self._gref, _ptr(x), C.c_void_p(x.data_ptr() + off), _ptr(self.zpart), _ptr(self.g.fx),
                                          _ptr(z_out), C.c_void_p(z_out.data_ptr() + off), _stream()), "vican_bip_apply")

    def _gram_workspace(self, n):
        """Slice partials of vican_tall_gram for long vectors (the non-eliminated solver); None for short ones."""
        if n < 16384:
            return None
        if self._gram_ws is None:
            self._gram_ws = torch.empty(_lib.GRAM_WS_DOUBLES, dtype=torch.float64, device=self.dev)
        return self._gram_ws

    def block_op_slabs(self, lamT_inv, x):
        """The sweep of P x only: the result stays in the fixed-point slabs for lanczos_cam_step(from_slabs=True)."""
        self.block_op_raw(lamT_inv, x)

    def lanczos_step_slabs(self, lamT_inv, lamC, V, ld, j, Hcol, beta, x, pivot_floor):
        """Sweep + cooperative camera-side step behind ONE host call (x: operand in, next operand out).  False: the sweep has run
        (slabs in zpart) but the cooperative step is not available - the caller folds and takes lanczos_cam_step."""
        if not self.coop_cam_step:
            self.block_op_raw(lamT_inv, x)
            return False
        if self._coop_ws is None:
            self._coop_ws = torch.zeros(int(self.lib.vican_lanczos_coop_ws_doubles(self.C)), dtype=torch.float64, device=self.dev)
            self._coop_sync = torch.zeros(2, dtype=torch.int32, device=self.dev)
        rc = self.lib.vican_lanczos_step_slabs(self._gref, _ptr(lamT_inv), _ptr(x), _ptr(self.zpart), _ptr(self.g.fx), _ptr(lamC), _ptr(V), ld, j,
                                               _ptr(self._coop_ws), _ptr(Hcol), _ptr(beta), _ptr(x), float(pivot_floor), _ptr(self._coop_sync),
                                               int(self.g.n_wg <= 64), _stream())
        if rc == _lib.ERR_CAPACITY:
            self.cooperative_failed("vican_lanczos_cam_coop")
            return False
        self._ck(rc, "vican_lanczos_step_slabs")
        return True

    def lanczos_cam_step(self, lamC, V, ld, j, z, R, H, G, Hcol, beta, x_out, pivot_floor, from_slabs=False):
        n_nodes = lamC.numel() // 9             # C for the eliminated solver, C + T for the general one
        if self.coop_cam_step and n_nodes == self.C:
            if self._coop_ws is None:
                self._coop_ws = torch.zeros(int(self.lib.vican_lanczos_coop_ws_doubles(self.C)), dtype=torch.float64, device=self.dev)
                self._coop_sync = torch.zeros(2, dtype=torch.int32, device=self.dev)
            slabs = (None, 0, None, None)
            if from_slabs:
                fxp = self.g.fx.data_ptr()
                slabs = (_ptr(self.zpart), self.g.n_wg, C.c_void_p(fxp + 24), C.c_void_p(fxp + 56))
            # fenced barriers where the sweeps leave little dirty data in L2 (few slabs): 0.4-0.8 us each there, ~10 us with
            # the stress graph's 256 slabs (which relies on the kernel's agent-scope atomics alone)
            rc = self.lib.vican_lanczos_cam_coop(self.C, _ptr(lamC), _ptr(V), ld, j, _ptr(z), _ptr(self._coop_ws), _ptr(Hcol),
                                                 _ptr(beta), _ptr(x_out), float(pivot_floor), _ptr(self._coop_sync), *slabs,
                                                 int(getattr(self.g, "n_wg", 1 << 30) <= 64), _stream())
            if rc != _lib.ERR_CAPACITY:
                self._ck(rc, "vican_lanczos_cam_coop")
                return
            self.cooperative_failed("vican_lanczos_cam_coop")        # grid not co-resident even on an idle device
        if from_slabs:
            self.fold_z(z)
        ws = self._gram_workspace(3 * n_nodes)
        self._ck(self.lib.vican_lanczos_cam_step(n_nodes, _ptr(lamC), _ptr(V), ld, j, _ptr(z), _ptr(R), _ptr(H), _ptr(G),
                                                 _ptr(Hcol), _ptr(beta), _ptr(x_out), float(pivot_floor), _ptr(ws),
                                                 0 if ws is None else ws.numel(), _stream()),
                 "vican_lanczos_cam_step")

    def cg_iter_local(self, deg_t, r_c, p_c, r_t, p_t, q_t, qcpq, rtol, st, n_rr_part):
        part = self.zpart[: self.cgl.n_wg * 6 * self.C]            # double-word camera partials: (hi, lo) planes per workgroup
        self._ck(self.lib.vican_cg_iter_local(self._gref_cg, _ptr(self._cg_w), _ptr(deg_t), _ptr(r_c), _ptr(p_c), _ptr(r_t),
                                              _ptr(p_t), _ptr(q_t), _ptr(part), _ptr(self.pq_part), _ptr(qcpq), float(rtol),
                                              _ptr(self.rr_part), int(n_rr_part), self.n_add_cg, _ptr(st), _stream()),
                 "vican_cg_iter_local")

    def cg_iter_finish(self, deg_c, qcpq, p_c, x_c, r_c, p_t, q_t, x_t, r_t, st):
        return self._ck(self.lib.vican_cg_iter_finish(self.C, self.T, _ptr(deg_c), _ptr(qcpq), _ptr(p_c), _ptr(x_c), _ptr(r_c),
                                                      _ptr(p_t), _ptr(q_t), _ptr(x_t), _ptr(r_t), _ptr(self.rr_part),
                                                      self.rr_part.numel(), _ptr(st), _stream()), "vican_cg_iter_finish")

    def cg_iter_fused(self, deg_t, deg_c, r_c, p_c, x_c, r_t, p_t, q_t, x_t, qcpq, rtol, st, first):
        """One single-rank CG iteration behind one host call, bit-reproducible from run to run (p_t.q_t over fixed slices):
        include/vican_hip.h vican_cg_iter_fused; same recurrence as cg_iter_local + cg_iter_finish."""
        if getattr(self, "_cg_ticket", None) is None:
            self._cg_ticket = torch.zeros(256, dtype=torch.int32, device=self.dev)    # workspace of the fused iteration: byte 256..: p.q partials
        part = self.zpart[: self.cgl.n_wg * 6 * self.C]
        self._ck(self.lib.vican_cg_iter_fused(self._gref_cg, _ptr(self._cg_w), _ptr(deg_t), _ptr(deg_c), _ptr(r_c), _ptr(p_c), _ptr(x_c),
                                              _ptr(r_t), _ptr(p_t), _ptr(q_t), _ptr(x_t), _ptr(part), _ptr(self.pq_part), _ptr(qcpq),
                                              float(rtol), _ptr(self.rr_part), self.rr_part.numel(), self.n_add_cg,
                                              int(bool(first)),
                                              _ptr(st), _ptr(self._cg_ticket), _stream()), "vican_cg_iter_fused")

    def cg_iter_comm(self, deg_t, deg_c, r_c, p_c, x_c, r_t, p_t, q_t, x_t, msg, rtol, st, first, comm):
        """One CG iteration of a sharded solve behind one host call, both all-reduces enqueued from C in stream order
        (include/vican_hip.h: vican_cg_iter_comm); comm: solver.Comm whose communicator lives in the C library, or a forced
        one-rank Comm without one (identity collectives: the same launches minus the two messages).  msg: 3C + CG_PQ_SLICES."""
        part = self.zpart[: self.cgl.n_wg * 6 * self.C]
        self._ck(self.lib.vican_cg_iter_comm(self._gref_cg, _ptr(self._cg_w), _ptr(deg_t), _ptr(deg_c), _ptr(r_c), _ptr(p_c), _ptr(x_c),
                                             _ptr(r_t), _ptr(p_t), _ptr(q_t), _ptr(x_t), _ptr(part), _ptr(self.pq_part), _ptr(msg),
                                             float(rtol), _ptr(self.rr_part), self.rr_part.numel(), self.n_add_cg, int(bool(first)),
                                             _ptr(st), comm.native_handle(), _stream()), "vican_cg_iter_comm")

    # one message per CG iteration (sharded solves; include/vican_hip.h: vican_cg1_iter_local / vican_cg1_iter_finish)
    def cg1_iter_local(self, deg_t, r_c, r_t, s_t, msg, st, n_rr_part):
        if getattr(self, "_cg1_sw", None) is None:
            self._cg1_sw = torch.zeros(_lib.CG_STATE_DOUBLES, dtype=torch.float64, device=self.dev)     # the sweep's view of the state
            self._cg1_sc = torch.zeros(4, dtype=torch.float64, device=self.dev)
        part = self.zpart[: self.cgl.n_wg * 6 * self.C]
        self._ck(self.lib.vican_cg1_iter_local(self._gref_cg, _ptr(self._cg_w), _ptr(deg_t), _ptr(r_c), _ptr(r_t), _ptr(s_t), _ptr(part),
                                               _ptr(self.pq_part), _ptr(msg), _ptr(self.rr_part), int(n_rr_part), self.n_add_cg,
                                               _ptr(st), _ptr(self._cg1_sw), _stream()), "vican_cg1_iter_local")

    def cg1_iter_finish(self, k, deg_c, msg, r_c, r_c_new, p_c, q_c, x_c, r_t, s_t, p_t, q_t, x_t, rtol, st):
        return self._ck(self.lib.vican_cg1_iter_finish(self.C, self.T, int(k), float(rtol), _ptr(deg_c), _ptr(msg), _ptr(r_c), _ptr(r_c_new), _ptr(p_c),
                                                       _ptr(q_c), _ptr(x_c), _ptr(r_t), _ptr(s_t), _ptr(p_t), _ptr(q_t), _ptr(x_t),
                                                       _ptr(self.rr_part), self.rr_part.numel(), _ptr(self._cg1_sc), _ptr(st), _stream()),
                        "vican_cg1_iter_finish")

    def fold_z(self, z_out):
        """Fold the fixed-point slabs of the last block_op_raw into z_out[3C,3]."""
        fxp = self.g.fx.data_ptr()
        self._ck(self.lib.vican_slab_reduce_fx(_ptr(self.zpart), self.g.n_wg, self.C, 9, 1.0, C.c_void_p(fxp + 24),
                                               C.c_void_p(fxp + 56), _ptr(z_out), _stream()), "vican_slab_reduce_fx")

    def dual_update(self, rc, Rt, lamT_inv):
        self._ck(self.lib.vican_dual_update(self._gref, _ptr(rc), _ptr(Rt), _ptr(lamT_inv), _ptr(self.g.rnorm),
                                            _ptr(self.g.fx), _stream()), "vican_dual_update")
        self._fx_finish()

    def dual_update_op(self, rc, Rt, lamT_inv, z_raw):
        """dual_update + z_raw[3C,3] = local partial of P_new rc (the next eigen-solve's first operator application)
        in ONE pass over the blocks (include/vican_hip.h: vican_dual_update_op)."""
        self._ck(self.lib.vican_dual_update_op(self._gref, _ptr(rc), _ptr(Rt), _ptr(lamT_inv), _ptr(self.g.rnorm),
                                               _ptr(self.g.fx), _ptr(self.zpart), _ptr(z_raw), _stream()), "vican_dual_update_op")
        self._fx_finish()

    def lanczos_seed(self, x0, V, ld, beta0, xrow, zraw=None, z=None):
        """Start block in one launch (include/vican_hip.h: vican_lanczos_seed); False if the vectors are too long."""
        n = x0.numel() // 3
        if n > _lib.SEED_MAX_N:
            if self._coop_sync is not None:
                self._coop_sync.zero_()
            return False
        # (the seed kernel also re-arms the cooperative step's grid-barrier counters: a completed launch leaves them at
        #  zero, but a launch that faulted or was torn down would make the next one pass its barriers early)
        self._ck(self.lib.vican_lanczos_seed(n, _ptr(x0), _ptr(V), ld, _ptr(beta0), _ptr(xrow), _ptr(zraw), _ptr(z),
                                             _ptr(self._coop_sync), _stream()), "vican_lanczos_seed")
        return True

    def right_solve3(self, X, beta, Z):
        self._ck(self.lib.vican_right_solve3(X.numel() // 3, _ptr(X), _ptr(beta), _ptr(Z), _stream()), "vican_right_solve3")

    def polar_dual(self, mats, R_out, lam_out, mode):
        self._ck(self.lib.vican_polar_dual(mats.numel() // 9, _ptr(mats), _ptr(R_out), _ptr(lam_out), mode, _stream()),
                 "vican_polar_dual")

    def gauge_project(self, x_in, x_out):
        self._ck(self.lib.vican_gauge_project(x_in.numel() // 9, _ptr(x_in), _ptr(x_out), _stream()), "vican_gauge_project")

    # -- Lanczos helpers ----------------------------------------------------
    def lap_apply(self, lamC, V, ld, col0, z, aq):
        self._ck(self.lib.vican_lap_apply(lamC.numel() // 9, _ptr(lamC), _ptr(V), ld, col0, _ptr(z), _ptr(aq), _stream()), "vican_lap_apply")

    def tall_gram(self, n, V, ld, ka, R, H):
        ws = self._gram_workspace(n)
        self._ck(self.lib.vican_tall_gram(n, _ptr(V), ld, ka, _ptr(R), _ptr(H), _ptr(ws), 0 if ws is None else ws.numel(), _stream()),
                 "vican_tall_gram")

    def tall_update(self, n, V, ld, ka, H, R, H_out, accumulate):
        self._ck(self.lib.vican_tall_update(n, _ptr(V), ld, ka, _ptr(H), _ptr(R), _ptr(H_out), int(accumulate), _stream()),
                 "vican_tall_update")

    def chol_qr3(self, n, R, G, V, ld, col0, beta_out, x_out, pivot_floor=0.0):
        self._ck(self.lib.vican_chol_qr3(n, _ptr(R), _ptr(G), _ptr(V), ld, col0, _ptr(beta_out), _ptr(x_out),
                                         float(pivot_floor), _stream()), "vican_chol_qr3")

    def ritz(self, HB, hw, steps, flags, eig_tol, floor_tol, floor_level, Y, status, gate, stall_ratio=0.25):
        """Device Ritz step over the first `steps` rows of HB (include/vican_hip.h: vican_ritz)."""
        self._ck(self.lib.vican_ritz(_ptr(HB), HB.stride(0), hw, steps, flags, float(eig_tol), float(floor_tol),
                                     float(floor_level), float(stall_ratio), _ptr(Y), _ptr(status), _ptr(gate), _stream()),
                 "vican_ritz")

    @contextlib.contextmanager
    def gated(self, gate):
        """Launches of the gated entry points inside this block run only if gate[0] == 1 on the device."""
        self.lib.vican_set_gate(_ptr(gate))
        try:
            yield
        finally:
            self.lib.vican_set_gate(None)

    def capture(self, fn):
        """Record the launches of fn() into a HIP graph (nothing executes); returns an object with .replay().
        fn must only enqueue kernels on the current stream with arguments that stay valid (device-resident
        state, preallocated buffers) - used for launch-bound inner loops on small graphs."""
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.graph(g, stream=side):
            fn()
        return g

    def post_status(self, status):
        """Asynchronous device->host copy of a small status vector; returns a handle for wait_status."""
        host = self._status_host.get(status.numel())
        if host is None:
            # (page-locked buffer + side stream + events come from a process-wide pool and go back to it with the backend: a
            #  one-shot drop-in call builds a fresh backend, and creating these cost it 1.3 ms of idle GPU in its first check)
            free = _STATUS_POOL.setdefault((str(self.dev), status.numel()), [])
            host = self._status_host[status.numel()] = free.pop() if free else (
                self.pinned(status.numel()), torch.cuda.Event(), torch.cuda.Event(), torch.cuda.Stream())
        # the copy runs on a side stream: in the launch stream it would sit between the Ritz kernel and the
        # speculative continuation and cost ~20 us of copy-engine latency per primal-dual iteration
        buf, done, ready, side = host
        ready.record()
        with torch.cuda.stream(side):
            side.wait_event(ready)
            buf.copy_(status, non_blocking=True)
            done.record()
        return host

    def wait_status(self, handle):
        handle[1].synchronize()                     # only the copy, not the work enqueued behind it
        return handle[0].numpy()

    def tall_combine(self, n, V, ld, ka, Y, X):
        self._ck(self.lib.vican_tall_combine(n, _ptr(V), ld, ka, _ptr(Y), _ptr(X), _stream()), "vican_tall_combine")

    def rows_to_cols(self, n, X, V, ld, col0):
        self._ck(self.lib.vican_rows_to_cols(n, _ptr(X), _ptr(V), ld, col0, _stream()), "vican_rows_to_cols")

    # -- translation stage --------------------------------------------------
    def trans_degrees(self, deg_t, deg_c):
        """Degrees of the weighted Laplacian (graph constants from pack time)."""
        deg_t[: self.g.row_sum_w.numel()].copy_(self.g.row_sum_w)
        deg_c.copy_(self.g.cam_sum_w)

    def trans_rhs(self, rc, rt, rhs_t, rhs_c):
        part = self.zpart[: self.tl.n_wg * 6 * self.C]              # double-word camera slabs
        self._ck(self.lib.vican_trans_rhs(self._gref_t, _ptr(self.g.u), _ptr(self.g.v), _ptr(rc), _ptr(rt), _ptr(rhs_t), _ptr(rhs_c),
                                          _ptr(part), self.g.gmax, self.n_add, _stream()), "vican_trans_rhs")

    # Jacobi scaling (tight translation solve): the CG entry points then run on the scaled weights
    def jacobi_scale(self, deg, s_out):
        self._ck(self.lib.vican_jacobi_scale(deg.numel(), _ptr(deg), _ptr(s_out), _stream()), "vican_jacobi_scale")

    def row_scale(self, s, x):
        self._ck(self.lib.vican_row_scale(s.numel(), x.numel() // max(s.numel(), 1), _ptr(s), _ptr(x), _stream()), "vican_row_scale")

    def set_cg_scaling(self, s_c, s_t):
        """CG sweeps use w~ = w s_c s_t (<= 1) until clear_cg_scaling()."""
        if self._w_scaled is None:
            self._w_scaled = torch.empty_like(self.g.w_cg)
        self._ck(self.lib.vican_scale_weights(self._gref_cg, _ptr(self.g.w_cg), _ptr(s_c), _ptr(s_t), _ptr(self._w_scaled), _stream()),
                 "vican_scale_weights")
        self._cg_w, self._cg_wmax = self._w_scaled, 1.0

    def clear_cg_scaling(self):
        self._cg_w, self._cg_wmax = self.g.w_cg, self.g.wmax

    def cg_init(self, b_c, b_t, x_c, x_t, r_c, r_t, p_c, p_t, st):
        self._ck(self.lib.vican_cg_init(self.C, self.T, _ptr(b_c), _ptr(b_t), _ptr(x_c), _ptr(x_t), _ptr(r_c), _ptr(r_t),
                                        _ptr(p_c), _ptr(p_t), _ptr(st), _ptr(self.ws), self._cg_wmax, _stream()), "vican_cg_init")

    @property
    def cg_resident_ok(self):
        """The whole CG as one cooperative launch (vican_cgres.hip): wave-layout graphs whose workgroups are co-resident."""
        if getattr(self, "_cgres_ok", None) is None:
            l = self.cgl
            self._cgres_ok = bool(
                self.layout == "wave" and self._gref_cg is self._gref and
                l.n_chunk > 0 and l.n_wg <= min(n_cu(), 128) and        # (a grid barrier costs 1 us at 40 workgroups, 2 at 128,
                                                                         #  3.8 at 256: measured CG stage 0.35 / 0.49 ms at 40
                                                                         #  workgroups, 0.44 / 0.50 at 98, 0.72 / 0.60 at 235)
                int(self.lib.vican_cg_resident_lds_bytes(self.C, l.max_rows, l.n_copy, l.rows_per_wg_max)) <= int(self.lib.vican_lds_limit_bytes()))
        return self._cgres_ok

    def cg_resident(self, deg_t, deg_c, b_c, b_t, x_c, x_t, rtol, maxiter, st):
        if getattr(self, "_cgres_ws", None) is None:
            n = int(self.lib.vican_cg_resident_ws_doubles(self.C, self.cgl.n_wg))
            self._cgres_ws = torch.zeros(n, dtype=torch.float64, device=self.dev)       # zeroed once: holds the barrier counter
        rc = self.lib.vican_cg_resident(self._gref_cg, _ptr(self._cg_w), _ptr(deg_t), _ptr(deg_c), _ptr(b_c), _ptr(b_t), _ptr(x_c),
                                        _ptr(x_t), _ptr(self.zpart), _ptr(self._cgres_ws), float(rtol),
                                        int(min(maxiter, 2 ** 31 - 1)), self.n_add_cg, float(self._cg_wmax),
                                        int(self.cgl.rows_per_wg_max), _ptr(st), _stream())
        if rc == _lib.ERR_CAPACITY:               # grid not co-resident even on an idle device: report as an aborted launch
            st.view(torch.int32)[_lib.CG_I["done"]] = -1
            return
        self._ck(rc, "vican_cg_resident")

    def cg_begin(self, r_c, p_c, rtol, st, n_rr_part=0):
        self._ck(self.lib.vican_cg_begin(self.C, _ptr(r_c), _ptr(p_c), float(rtol), _ptr(self.rr_part), int(n_rr_part),
                                         self.n_add_cg, _ptr(st), _stream()), "vican_cg_begin")

    def cg_sweep(self, deg_t, p_c, r_t, p_t, q_t, qcpq, st):
        """qcpq[0:3C] = local sum_t w p_t (slab-reduced), qcpq[3C] = local p_t.q_t."""
        nwg = self.cgl.n_wg
        part = self.zpart[: nwg * 6 * self.C]
        self._ck(self.lib.vican_cg_sweep(self._gref_cg, _ptr(self._cg_w), _ptr(deg_t), _ptr(p_c), _ptr(r_t), _ptr(p_t), _ptr(q_t),
                                         _ptr(part), _ptr(self.pq_part), _ptr(st), _stream()), "vican_cg_sweep")
        self._ck(self.lib.vican_cg_fold(_ptr(part), nwg, self.C, _ptr(self.pq_part), _ptr(qcpq), _ptr(st), _stream()), "vican_cg_fold")

    def cg_cam_step(self, deg_c, qcpq, p_c, x_c, r_c, st):
        self._ck(self.lib.vican_cg_cam_step(self.C, _ptr(deg_c), _ptr(qcpq), C.c_void_p(qcpq.data_ptr() + 8 * 3 * self.C),
                                            _ptr(p_c), _ptr(x_c), _ptr(r_c), _ptr(st), _stream()), "vican_cg_cam_step")

    def cg_time_step(self, p_t, q_t, x_t, r_t, st):
        return self._ck(self.lib.vican_cg_time_step(self.T, _ptr(p_t), _ptr(q_t), _ptr(x_t), _ptr(r_t), _ptr(self.rr_part),
                                                    self.rr_part.numel(), _ptr(st), _stream()), "vican_cg_time_step")

    def cg_end(self, n_part, st):
        self._ck(self.lib.vican_cg_end(_ptr(self.rr_part), int(n_part), _ptr(st), _stream()), "vican_cg_end")


# -- LSQR ("direct") wrappers, attached to HipBackend ------------------------------------------
class _LsqrCtx:
    pass


def _lsqr_ctx(self):
    """Arrays of the LSQR kernels.  The device-resident path (vican_lsqr_init_u / vican_lsqr_step) runs on the graph's own
    layout, wave or block; the round-2 path with host scalars (vican_lsqr_u_step / vican_lsqr_v_step, `lsqr_host_scalars`)
    is block-only: wave-layout graphs pack a second layout for it on first use (LocalGraph.lsqr_layout)."""
    legacy = bool(getattr(self, "lsqr_host_scalars", False))
    cache = self.__dict__.setdefault("_lsqr_ctxs", {})
    if legacy not in cache:
        c = _LsqrCtx()
        if legacy or self.g.rot.kind == "block":
            c.ll, c.desc, c.w, c.u_in, c.v_in = self.g.lsqr_layout()
        else:
            c.ll, c.desc, c.w, c.u_in, c.v_in = self.g.rot, self.g.desc, self.g.w, self.g.u, self.g.v
        c.gref = C.byref(c.desc)
        c.n_add = float(max(c.ll.rows_per_wg_max, c.ll.slots) + 1)
        nslot = max(1, c.ll.n_chunk) * c.ll.slots
        c.u = torch.zeros(3 * nslot, dtype=torch.float64, device=self.dev)
        c.sw = torch.zeros(nslot, dtype=torch.float64, device=self.dev)       # sqrt(w), written by lsqr_init_u
        c.part = torch.zeros(max(c.ll.n_wg, 1024), dtype=torch.float64, device=self.dev)
        c.slab = torch.empty(max(c.ll.n_wg, 1) * 6 * self.C, dtype=torch.float64, device=self.dev)
        cache[legacy] = c
    return cache[legacy]


def _lsqr_init_u(self, rc, rt, nrm2_out):
    c = _lsqr_ctx(self)
    self._ck(self.lib.vican_lsqr_init_u(c.gref, _ptr(c.w), _ptr(c.u_in), _ptr(c.v_in), _ptr(rc), _ptr(rt),
                                        _ptr(c.u), _ptr(c.sw), _ptr(c.part), _ptr(nrm2_out), _stream()),
             "vican_lsqr_init_u")


def _lsqr_u_step(self, v_c, v_t, coef, nrm2_out):
    c = _lsqr_ctx(self)
    self._ck(self.lib.vican_lsqr_u_step(c.gref, _ptr(c.sw), _ptr(v_c), _ptr(v_t), float(coef), _ptr(c.u),
                                        _ptr(c.part), _ptr(nrm2_out), _stream()), "vican_lsqr_u_step")


def _lsqr_v_step(self, inv_beta, beta, v_t, acc_c, nrm2_t_out):
    """v_t updated in place; acc_c[3C] = this rank's camera-side sums (all-reduce, then lsqr_cam_v)."""
    c = _lsqr_ctx(self)
    inv = C.c_double(0.0)
    self._ck(self.lib.vican_lsqr_v_step(c.gref, _ptr(c.sw), _ptr(c.u), float(inv_beta), float(beta), _ptr(v_t),
                                        _ptr(c.slab), _ptr(c.part), _ptr(nrm2_t_out), math.sqrt(self.g.wmax), c.n_add,
                                        C.byref(inv), _stream()), "vican_lsqr_v_step")
    self._ck(self.lib.vican_slab_reduce_fx(_ptr(c.slab), c.ll.n_wg, self.C, 3, inv.value, None, None, _ptr(acc_c), _stream()),
             "vican_slab_reduce_fx")


def _lsqr_cam_v(self, acc_c, beta, v_c, nrm2_out):
    self._ck(self.lib.vican_lsqr_cam_v(self.C, _ptr(acc_c), float(beta), _ptr(v_c), _ptr(nrm2_out), _stream()), "vican_lsqr_cam_v")


def _lsqr_update(self, inv_alfa, t1, t2, v, w, x, nrm2_w_out):
    if not hasattr(self, "_lsqr_part"):
        self._lsqr_part = torch.zeros(1024, dtype=torch.float64, device=self.dev)
    self._ck(self.lib.vican_lsqr_update(v.numel(), float(inv_alfa), float(t1), float(t2), _ptr(v), _ptr(w), _ptr(x),
                                        _ptr(self._lsqr_part), _ptr(nrm2_w_out), _stream()), "vican_lsqr_update")


def _lsqr_step(self, v_c, v_t, z_t, acc, st):
    """One fused pass over the edges (vican_lsqr_step): u~ <- J~ v - coef u~, z_t, acc[0:3C] camera sums, acc[3C] = |u^|^2."""
    c = _lsqr_ctx(self)
    self._ck(self.lib.vican_lsqr_step(c.gref, _ptr(c.sw), _ptr(c.u), _ptr(v_c), _ptr(v_t), _ptr(z_t), _ptr(c.slab),
                                      _ptr(c.part), _ptr(acc), _ptr(st), _stream()), "vican_lsqr_step")


def _lsqr_nodes(self, z_t, acc, v_t, v_c, part2, st):
    return self._ck(self.lib.vican_lsqr_nodes(self.C, self.T, _ptr(z_t), _ptr(acc), _ptr(v_t), _ptr(v_c), _ptr(part2), _ptr(st), _stream()),
                    "vican_lsqr_nodes")


def _lsqr_scalars(self, acc, part2, n_part, tsum, wpart_t, n_wt, wpart_c, n_wc, wsum_t, st):
    self._ck(self.lib.vican_lsqr_scalars(self.C, _ptr(acc), _ptr(part2), int(n_part), _ptr(tsum), _ptr(wpart_t), int(n_wt), _ptr(wpart_c),
                                         int(n_wc), _ptr(wsum_t), _ptr(st), _stream()), "vican_lsqr_scalars")


def _lsqr_update_st(self, v, w, x, part, last, st):
    return self._ck(self.lib.vican_lsqr_update_st(v.numel(), _ptr(v), _ptr(w), _ptr(x), _ptr(part), int(last), _ptr(st), _stream()),
                    "vican_lsqr_update_st")


def _lsqr_device_params(self):
    """(smax, n_add) of the fused step's fixed-point scale."""
    return math.sqrt(self.g.wmax), _lsqr_ctx(self).n_add


HipBackend.lsqr_step = _lsqr_step
HipBackend.lsqr_nodes = _lsqr_nodes
HipBackend.lsqr_scalars = _lsqr_scalars
HipBackend.lsqr_update_st = _lsqr_update_st
HipBackend.lsqr_device_params = _lsqr_device_params
HipBackend.lsqr_init_u = _lsqr_init_u
HipBackend.lsqr_u_step = _lsqr_u_step
HipBackend.lsqr_v_step = _lsqr_v_step
HipBackend.lsqr_cam_v = _lsqr_cam_v
HipBackend.lsqr_update = _lsqr_update


def make_backend(n_cam, row_ptr, col, blk, a, w=None, u=None, v=None, deg_t=None, deg_c=None, row_ptr_host=None):
    """(graph, backend) for one rank's rows: the fused layouts up to TILE_CAMS cameras, camera tiles beyond.
    deg_t / deg_c: diagonal of the translation system for this rank's rows / this rank's share of the camera diagonal
    (default: sums of w)."""
    import os
    from .tiled import TILE_CAMS, TiledBackend, TiledGraph
    tile = int(os.environ.get("VICAN_TILE_CAMS") or TILE_CAMS)
    if n_cam > tile:
        g = TiledGraph(n_cam, row_ptr, col, blk, a, w, u, v, tile=tile, deg_t=deg_t, deg_c=deg_c,
                       permute_rows=True)
        return g, TiledBackend(g)
    g = LocalGraph(n_cam, row_ptr, col, blk, a, w, u, v, deg_t=deg_t, deg_c=deg_c, row_ptr_host=row_ptr_host)
    return g, HipBackend(g)


from .merge import merge_edges                                      # noqa: E402,F401


def __getattr__(name):
    # TILE_CAMS / TiledBackend / TiledGraph live in tiled.py (which imports THIS module): resolved on first use, whichever of the
    # two modules is imported first
    if name in ("TILE_CAMS", "TiledBackend", "TiledGraph"):
        from . import tiled
        return getattr(tiled, name)
    raise AttributeError("module %r has no attribute %r" % (__name__, name))
