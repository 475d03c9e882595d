"""Host orchestration of the primal-dual solve on top of a kernel backend.

Algorithm = reference ``large_bipartite_so3sync`` (bipgo.py:279-348) and the
normal-equation CG of ``bipartite_se3sync`` (bipgo.py:476-478), restructured
for the GPU:

  * the power graph P = R~ Lambda_T^-1 R~^T is never formed; every use is one
    fused sweep over the timestep-major edge blocks (``block_op``);
  * the shift-invert ARPACK call (bipgo.py:288) becomes a matrix-free block
    Lanczos iteration (block size 3, full re-orthogonalisation) whose only
    heavy step is that sweep; the small projected eigenproblem (<= 3m x 3m) is
    solved on the device too (``vican_ritz``: Jacobi iteration + convergence verdict),
    and the rest of the iteration is enqueued speculatively behind it under a launch
    gate, so the host never stalls the pipeline on a converged check;
  * per-node SVDs are batched device kernels;
  * CG keeps alpha/beta/residual norms in a device-resident state struct and
    follows scipy's recurrence and stopping test exactly.

Multi-GPU: timestep rows are sharded over ranks; every camera-side quantity is
replicated.  The only communication is ``comm.allreduce`` of camera-side
partials (3C x 3 doubles per operator application; 3C+1 doubles and one scalar
per CG step) - no edge data ever moves.

This file contains no numerics of its own and is
backend-agnostic so that the sharding / all-reduce logic can be exercised on
CPU with gloo by the tests (which inject a NumPy backend).
"""
from __future__ import annotations

import os

import numpy as np
import torch

from ._lib import CG_F, CG_I, CG_STATE_DOUBLES, LSQR_F, LSQR_I, LSQR_STATE_DOUBLES


_NATIVE_COMMS = {}       # (process group, device index) -> the C library's communicator of that group: made once per process


class Comm:
    """Sum-all-reduce over ranks; identity for a single rank.

    Transport of the all-reduces of DEVICE tensors in a multi-rank group (``transport=`` / ``VICAN_COMM``; default "auto"):

      "peer"   a communicator held by the C library (include/vican_hip.h: vican_comm_*), every collective enqueued from C on
               the launch stream: the peer exchange (ONE launch of the library's own kernel over hipIpc-mapped mailboxes;
               gateable, so sharded solves speculate like single-rank ones) with RCCL - where the group is an RCCL group -
               behind it for messages larger than the mailboxes;
      "rccl"   the same communicator, ncclAllReduce only;
      "torch"  torch.distributed.all_reduce, issued by this Python code (what host tensors - the NumPy stand-in backend of the
               CPU tests - always take).
      "auto"   "peer" where it can be set up AND passes its self-test on every rank (a known message through the exchange,
               bounded waits), else "rccl" on RCCL groups, else "torch".  A transport that fails is never an error.
    """

    PEER_MAX_DOUBLES = 131072            # mailbox capacity per message (9 C doubles: up to 14 000 cameras); larger messages: RCCL / torch

    def __init__(self, group=None, native=None, force_sharded=False, transport=None):
        self.group = group
        self.world = 1
        self.rank = 0
        if group is not None or (torch.distributed.is_available() and torch.distributed.is_initialized()):
            self.world = torch.distributed.get_world_size(group)
            self.rank = torch.distributed.get_rank(group)
        self.n_allreduce = 0
        # force_sharded: take the sharded code paths (launch sequences, all-reduces, two-message CG) although this
        # rank holds every row - how the multi-GPU schedule is exercised, timed and compared bit for bit on ONE GPU
        self.force_sharded = bool(force_sharded)
        if transport is None:
            transport = os.environ.get("VICAN_COMM", "auto")
        if native is not None:                                   # (round-5 spelling: native=True = RCCL from C, False = torch)
            transport = "rccl" if native else "torch"
        if transport not in ("auto", "peer", "rccl", "torch"):
            raise ValueError("Comm: unknown transport %r" % (transport,))
        self.transport_wanted = transport
        self.transport = "torch" if self.world > 1 else None      # what is in use (set by _setup_native on the first device message)
        self._native, self._native_lib, self._native_tried = None, None, False
        self.gateable = self.world == 1
        self.notes = []

    @property
    def sharded(self):
        """True where the solver must take the sharded schedule (more than one rank, or forced on one)."""
        return self.world > 1 or getattr(self, "force_sharded", False)

    # -- the C library's communicator ------------------------------------------------------------------------------------
    def native_handle(self):
        """The vican_comm_t* whose collectives the C composites (vican_block_op_z_comm, vican_cg_iter_comm) enqueue
        themselves, or None (torch transport, single rank without a forced transport)."""
        return getattr(self, "_native", None)

    def _setup_native(self, device):
        """First device message of a multi-rank group: create the C library's communicator (a collective over the group -
        every rank arrives here with the same message) and choose the transport."""
        self._native_tried = True
        want = self.transport_wanted
        if want == "torch" or self.world == 1:
            return
        # one communicator (RCCL init, mailboxes, hipIpc mappings, self-test) per group, device and wanted transport for the
        # life of the process: a drop-in call makes a fresh Comm, and every rank makes it at the same point
        key = (self.group if self.group is not None else "world", device.index, want)
        hit = _NATIVE_COMMS.get(key)
        if hit is None:
            self._create_native(device)
            hit = _NATIVE_COMMS[key] = dict(native=self._native, lib=self._native_lib, transport=self.transport, gateable=self.gateable,
                                            notes=list(self.notes), verified=False)
        else:
            self._native, self._native_lib, self.transport, self.gateable, self.notes = (hit["native"], hit["lib"], hit["transport"],
                                                                                           hit["gateable"], list(hit["notes"]))
        self._shared = hit                                       # (state of the group's communicator, shared by every Comm of the group)
        self._owns_native = False

    @property
    def _verified(self):
        return bool(getattr(self, "_shared", {}).get("verified", False))

    @_verified.setter
    def _verified(self, v):
        if getattr(self, "_shared", None) is not None:
            self._shared["verified"] = bool(v)

    def _create_native(self, device):
        want = self.transport_wanted
        import ctypes
        from . import _lib
        lib = _lib.load()
        dist = torch.distributed
        rccl_group = self._device_collectives()
        src = dist.get_global_rank(self.group, 0) if self.group is not None else 0

        def agree(ok):                                           # True iff every rank says True (host-side, any backend)
            flags = [None] * self.world
            dist.all_gather_object(flags, bool(ok), group=self.group)
            return all(flags)

        h = ctypes.c_void_p()
        have = False
        if rccl_group:
            ids = [None]
            if self.rank == 0:
                buf = ctypes.create_string_buffer(128)
                ids = [buf.raw if lib.vican_comm_unique_id(buf) == 0 else None]
            dist.broadcast_object_list(ids, src=src, group=self.group)
            ok = ids[0] is not None and lib.vican_comm_create(self.rank, self.world, ctypes.create_string_buffer(ids[0], 128), ctypes.byref(h)) == 0
            if not ok:
                self.notes.append("vican_comm_create: %s" % (lib.vican_last_error() or b"?").decode())
            # (a rank whose ncclCommInitRank failed while others succeeded cannot be repaired here: RCCL's init is itself a
            #  collective and fails or succeeds for the group)
            have = agree(ok)
            if not have and ok:
                lib.vican_comm_destroy(h)
        elif want in ("auto", "peer") and self.world <= 8:
            have = agree(lib.vican_comm_create_local(self.rank, self.world, ctypes.byref(h)) == 0)
        if not have:
            return
        self._native, self._native_lib = h, lib
        self.transport = "rccl" if rccl_group else "torch"
        if want in ("auto", "peer") and self.world <= 8:
            mine = ctypes.create_string_buffer(64)
            ok = lib.vican_comm_peer_export(h, self.PEER_MAX_DOUBLES, mine) == 0
            if not ok:
                self.notes.append("vican_comm_peer_export: %s" % (lib.vican_last_error() or b"?").decode())
            handles = [None] * self.world
            dist.all_gather_object(handles, mine.raw if ok else None, group=self.group)
            ok = all(x is not None for x in handles) and lib.vican_comm_peer_attach(h, ctypes.create_string_buffer(b"".join(handles), 64 * self.world)) == 0
            if not ok and all(x is not None for x in handles):
                self.notes.append("vican_comm_peer_attach: %s" % (lib.vican_last_error() or b"?").decode())
            if agree(ok):
                # (the self-test runs right behind a collective that aligned the ranks: a short bound ends a broken mapping quickly;
                #  solves get the long one - a rank may reach its first collective seconds after the others)
                lib.vican_comm_peer_set_timeout(h, 2_000_000)
                ok = agree(self._peer_selftest(device))
                lib.vican_comm_peer_set_timeout(h, int(os.environ.get("VICAN_PEER_TIMEOUT_US", 30_000_000)))
                if ok:
                    self.transport, self.gateable = "peer", True
                else:
                    self.notes.append("peer exchange failed its self-test: disabled")
            if self.transport != "peer":
                lib.vican_comm_peer_enable(h, 0)
        if self.transport == "torch":                            # a local communicator without a working exchange serves nothing
            lib.vican_comm_destroy(h)
            self._native = None

    def _peer_selftest(self, device):
        """Known messages through the exchange before any solve depends on it (three sizes; a synchronised exchange, then a
        burst of 16 back-to-back ones without host synchronisation: both parities, the slot-reuse order; bounded waits):
        rank r sends (r + 1)(i + 1) + r / 4, exactly representable, so the rank-ordered sum is known in closed form."""
        import ctypes
        lib, h, W = self._native_lib, self._native, self.world
        ok = True
        with torch.cuda.device(device):
            stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
            for n in (1, 700, min(self.PEER_MAX_DOUBLES, 9 * 1024 + 96)):
                i = torch.arange(1, n + 1, dtype=torch.float64, device=device)
                mine = (self.rank + 1) * i + 0.25 * self.rank
                want = (W * (W + 1) // 2) * i + 0.25 * (W * (W - 1) // 2)
                bufs = [mine.clone() for _ in range(17)]
                ok = ok and lib.vican_comm_allreduce_sum(h, ctypes.c_void_p(bufs[0].data_ptr()), n, stream) == 0
                torch.cuda.current_stream().synchronize()
                for b in bufs[1:]:
                    ok = ok and lib.vican_comm_allreduce_sum(h, ctypes.c_void_p(b.data_ptr()), n, stream) == 0
                torch.cuda.current_stream().synchronize()
                ok = ok and all(bool(torch.equal(b, want)) for b in bufs) and lib.vican_comm_peer_status(h) == 0
        return ok

    def healthy(self):
        """COLLECTIVE (every rank of the group calls it at the same point): True iff no wait of the peer exchange has timed out on
        any rank.  On False the group's communicator is demoted on EVERY rank - the exchange is switched off, the all-reduces go
        through RCCL from C (RCCL groups) or torch.distributed, speculation through collectives stops - and stays demoted for
        the life of the process: the caller re-runs what it computed since the last healthy point.  Cheap to call between solves
        (one tiny host-side all-gather); never called inside a timed region."""
        if self.world == 1 or getattr(self, "_native", None) is None or getattr(self, "transport", None) != "peer":
            return True
        bad = self._native_lib.vican_comm_peer_status(self._native) > 0
        flags = [None] * self.world
        torch.distributed.all_gather_object(flags, bool(bad), group=self.group)
        if not any(flags):
            return True
        self._demote("a wait of the peer exchange timed out on rank(s) %s" % [r for r, f in enumerate(flags) if f])
        return False

    def _demote(self, why):
        self.notes.append("peer exchange switched off: " + why)
        self._native_lib.vican_comm_peer_enable(self._native, 0)
        self.gateable = False
        if self._device_collectives():
            self.transport = "rccl"
        else:                                                   # a local communicator has no other collective: torch carries them
            self.transport, self._native = "torch", None
        if getattr(self, "_shared", None) is not None:           # later Comm objects of this group start demoted
            self._shared.update(native=self._native, transport=self.transport, gateable=False, notes=list(self.notes))

    def check(self):
        """Raise if a wait of the peer exchange timed out since the communicator was made (its messages came back as NaN)."""
        if getattr(self, "_native", None) is not None and getattr(self, "transport", None) == "peer":
            n = self._native_lib.vican_comm_peer_status(self._native)
            if n > 0:
                raise RuntimeError("peer exchange: %d element waits timed out on rank %d (a rank stopped, or left the collective "
                                   "schedule); set VICAN_COMM=rccl to run without the exchange" % (n, self.rank))

    def __del__(self):
        try:
            if getattr(self, "_native", None) is not None and getattr(self, "_owns_native", False):
                self._native_lib.vican_comm_destroy(self._native)
                self._native = None
        except Exception:                                       # noqa: BLE001  (interpreter shutdown)
            pass

    @classmethod
    def single(cls, force_sharded=False, native=False, peer=False):
        """A one-rank communicator even inside an initialised process group (replicated computations).
        force_sharded: the solver takes its multi-rank schedule on it.  What one GPU can execute of the collective path:
        native: the all-reduces of that schedule are REAL ncclAllReduce calls on a one-rank RCCL communicator held by the C
        library (RCCL's one-rank kernel on the launch stream; include/vican_hip_test.h: vican_comm_force_enqueue);
        peer: they are launches of the peer exchange with the rank's own mailbox slot as its peer (the same kernel, granules,
        waits and epoch as with eight ranks - minus the links)."""
        c = cls.__new__(cls)
        c.group, c.world, c.rank, c.n_allreduce, c._native = None, 1, 0, 0, None
        c.force_sharded = bool(force_sharded)
        c.transport_wanted, c.transport, c._native_tried, c.gateable, c.notes = "torch", None, True, True, []
        if native or peer:
            import ctypes
            from . import _lib
            lib = _lib.load()
            h = ctypes.c_void_p()
            if peer:
                _lib.check(lib.vican_comm_create_local(0, 1, ctypes.byref(h)), "vican_comm_create_local")
                _lib.check(lib.vican_comm_peer_export(h, cls.PEER_MAX_DOUBLES, ctypes.create_string_buffer(64)), "vican_comm_peer_export")
                _lib.check(lib.vican_comm_peer_attach(h, None), "vican_comm_peer_attach")
                c.transport = "peer"
            else:
                buf = ctypes.create_string_buffer(128)
                _lib.check(lib.vican_comm_unique_id(buf), "vican_comm_unique_id")
                _lib.check(lib.vican_comm_create(0, 1, buf, ctypes.byref(h)), "vican_comm_create")
                _lib.check(lib.vican_comm_force_enqueue(h, 1), "vican_comm_force_enqueue")
                c.transport, c.gateable = "rccl", False
            c._native_lib, c._native, c._owns_native = lib, h, True
        return c

    def allreduce(self, t):
        if self.sharded:
            if t.is_cuda and not getattr(self, "_native_tried", True):
                self._setup_native(t.device)
            if (getattr(self, "_native", None) is not None and t.is_cuda and t.dtype == torch.float64 and t.is_contiguous()
                    and (self.transport != "peer" or t.numel() <= self.PEER_MAX_DOUBLES or self.world == 1 or self._device_collectives())):
                import ctypes
                from . import _lib
                _lib.check(self._native_lib.vican_comm_allreduce_sum(self._native, ctypes.c_void_p(t.data_ptr()), t.numel(),
                                                                     ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)), "vican_comm_allreduce_sum")
            elif self.world > 1:
                torch.distributed.all_reduce(t, group=self.group)
            self.n_allreduce += 1
        return t

    def any_flag(self, flag, device=None):
        """Logical OR of a host flag over the ranks (one tiny MAX all-reduce; device tensor for RCCL groups)."""
        if self.world == 1:
            return bool(flag)
        t = torch.tensor([1.0 if flag else 0.0], dtype=torch.float64, device=device if self._device_collectives() else None)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX, group=self.group)
        return bool(t.item() > 0)

    def _device_collectives(self):
        """True if the group runs collectives on device tensors (RCCL) - decided from the group's backend CONFIGURATION
        ("nccl", or "cpu:gloo,cuda:nccl" of a default multi-backend group), not from the single name get_backend() reports."""
        if getattr(self, "_dev_coll", None) is None:
            name = str(torch.distributed.get_backend(self.group))
            try:
                cfg = str(torch.distributed.get_backend_config(self.group))
            except Exception:                                   # noqa: BLE001  (older torch)
                cfg = name
            self._dev_coll = "nccl" in name or "cuda:nccl" in cfg or cfg.strip() == "nccl"
        return self._dev_coll

    def gather_rows(self, loc, n_local, bounds):
        """Concatenate per-rank row blocks (rank r owns rows bounds[r] .. bounds[r + 1]) into the full [T, width] array
        on every rank: ONE all-gather of equally sized (padded) blocks - each rank sends only its own rows."""
        if self.world == 1:
            return loc[:n_local]
        width = loc.shape[1]
        nmax = max(bounds[r + 1] - bounds[r] for r in range(self.world))
        send = torch.zeros(nmax, width, dtype=loc.dtype, device=loc.device)
        send[:n_local] = loc[:n_local]
        # the path is chosen by the backend's NAME, never by catching an error (a failing RCCL collective must surface):
        # RCCL gathers device tensors in place; gloo has no all-gather for device tensors, so its blocks travel through host memory
        if self._device_collectives() or not loc.is_cuda:
            recv = torch.empty(self.world * nmax, width, dtype=loc.dtype, device=loc.device)
            torch.distributed.all_gather_into_tensor(recv, send, group=self.group)
        else:
            parts = [torch.empty(nmax, width, dtype=loc.dtype) for _ in range(self.world)]
            torch.distributed.all_gather(parts, send.cpu(), group=self.group)
            recv = torch.cat(parts, 0).to(loc.device)
        self.n_allgather = getattr(self, "n_allgather", 0) + 1
        return torch.cat([recv[r * nmax: r * nmax + bounds[r + 1] - bounds[r]] for r in range(self.world)], 0)


class RotationSolver:
    def __init__(self, K, comm=None, m_max=32, eig_tol=1e-10, floor_tol=1e-7, min_steps=4, check_every=2, warm_min_steps=2,
                 max_restarts=20, seed=1234, n_nodes=None, prop_sweeps=None):
        self.K, self.comm = K, comm or Comm()
        self.C = K.C
        self.N = K.C if n_nodes is None else n_nodes          # nodes of the eigenproblem (cameras; C+T for the general solver)
        self.n = 3 * self.N
        self.m_max = max(1, min(m_max, 32, self.n // 3))       # VICAN_RITZ_MAX_STEPS
        self.eig_tol, self.min_steps, self.check_every = eig_tol, min_steps, check_every
        self.warm_min_steps = warm_min_steps
        # the stalled-residual rule below exists for the rounding floor of f32 blocks (~6e-8); f64 blocks have none
        # above 1e-13, and a slowly converging iteration (< 4x per check) must not be mistaken for a floor
        self.floor_tol = floor_tol if not getattr(K, "storage_f64", False) else min(floor_tol, 1e-13)
        # on cache-resident graphs an edge sweep costs tens of microseconds - less than one projection
        # check (device Ritz kernel + cancelled speculative launches) - so check less often there
        n_edges = getattr(getattr(K, "g", None), "n_edges", None)
        self.small_graph = n_edges is not None and n_edges * max(self.comm.world, 1) < 2_000_000
        if self.small_graph:
            self.min_steps, self.warm_min_steps, self.check_every = max(min_steps, 8), max(warm_min_steps, 4), max(check_every, 4)
        self.max_restarts, self.seed = max_restarts, seed
        # start block of the FIRST eigen-solve: rotations propagated from the gauge camera through the power graph
        # (_propagated_start) instead of a random block; 0 = the random block of rounds 1-3
        # sweeps: as many as it takes the propagation to reach (nearly) every camera - one where a camera shares timesteps
        # with several times C others (dense captures), three on sparse co-visibility graphs
        if prop_sweeps is None and "VICAN_PROP_SWEEPS" in os.environ:
            prop_sweeps = int(os.environ["VICAN_PROP_SWEEPS"])
        # (chosen in init() from the graph's GLOBAL edge and row counts, which ride with the degree message of sharded runs:
        #  every rank must run the same number of sweeps - each carries an all-reduce)
        self.prop_sweeps = None if prop_sweeps is None else int(prop_sweeps)
        self.start_is_warm = False
        n, m = self.n, self.m_max
        self.ld = n
        self.V = K.empty((3 * (m + 1)) * n)         # column-major basis
        self.R = K.empty(3 * n)                     # work block, column-major [3][n]
        self.H = K.empty(3 * (m + 1) * 3)
        self.G = K.empty(9)
        self.beta0 = K.empty(9)
        self.pivot_floor = 0.0
        self.x0 = None
        # per step: projected column V^T A Q_j (3(m+1) x 3) followed by beta_j (3 x 3) in ONE row, so a
        # projection check is a single device->host copy into a pinned buffer
        self.hw = 3 * (m + 1) * 3
        self.HB = K.zeros(m, self.hw + 9)
        self.Hbuf, self.Bbuf = self.HB[:, : self.hw], self.HB[:, self.hw:]
        self.Yd = K.zeros(3 * (m + 1), 3)
        self.status = K.zeros(16)                    # VICAN_RITZ_STATUS_DOUBLES
        self.gate = K.zeros(1, dtype=torch.int32)
        self.pred_steps = {}                        # iteration index -> steps needed last time
        self._pred_fail = {}                        # iteration index -> consecutive solves whose first check at pred_steps failed
        self._probe_done = {}                       # iteration index -> True once the remembered count is known to be the smallest
        self.floor_level = {}                       # iteration index -> residual level at which it stalled
        self.xrow = K.empty(n, 3)                   # current Lanczos block, row-major (sweep input)
        self.z = K.empty(n, 3)
        self.X = K.empty(n, 3)
        self.Xp = K.empty(n, 3)
        self.rc = K.empty(n, 3)                     # r_c of the reference (node<-world), stacked
        self.lamC = K.empty(self.N, 9)
        self._deg_msg = K.zeros(self.N + 2)          # [weighted degrees | edges | rows of this rank]: ONE set-up all-reduce
        self.cam_deg = self._deg_msg[: self.N]
        self.lamT = K.empty(max(K.T, 1), 9)
        self.Rt = K.empty(max(K.T, 1), 9)
        self.zraw = K.empty(n, 3)                   # P_new rc from the fused dual update (see _tail)
        self.z_ready = False
        self.fuse_dual_op = bool(getattr(K, "fused_dual_ok", True))       # (camera-tiled graphs: two separate passes)
        # single rank, few slabs: the camera-side Lanczos kernel folds the sweep's slabs itself (one launch less per step)
        self.fold_in_step = bool(getattr(K, "fold_in_step_ok", True))
        self.stats = dict(sweeps=0, lanczos_steps=[], evals=[], restarts=0, resid=[])

    # -- operator ------------------------------------------------------------
    def apply_P(self, x, z):
        """z = P x for the next lanczos_cam_step.  Single rank: the sweep's slabs are folded inside that step's kernel
        (returns True: z itself is not written); sharded: fold, then all-reduce the camera-side partials."""
        self.stats["sweeps"] += 1
        if not self.comm.sharded and self.fold_in_step and hasattr(self.K, "block_op_slabs"):
            self.K.block_op_slabs(self.lamT, x)
            return True
        self._op_allreduce(self.lamT, x, z)
        return False

    def _op_allreduce(self, lamT, x, z):
        """z = P x summed over the ranks: sweep, slab fold and all-reduce behind ONE host call where the backend and the
        communicator allow it (include/vican_hip.h: vican_block_op_z_comm)."""
        fused = getattr(self.K, "block_op_comm", None)
        if fused is not None and self.comm.sharded:
            fused(lamT, x, z, self.comm)
        else:
            self.K.block_op(lamT, x, z)
            self.comm.allreduce(z)

    # -- spectral step ---------------------------------------------------------
    def _seed_block(self, x0, with_z=False):
        """Orthonormal start block from x0 (and, with_z: z = zraw beta0^-1, the operator applied to it - see _tail)."""
        K, n = self.K, self.n
        if K.lanczos_seed(x0, self.V, self.ld, self.beta0, self.xrow, self.zraw if with_z else None, self.z if with_z else None):
            return
        K.rows_to_cols(n, x0, self.R, n, 0)
        K.tall_gram(n, self.R, n, 3, self.R, self.G)
        K.chol_qr3(n, self.R, self.G, self.V, self.ld, 0, self.beta0, self.xrow, 0.0)
        if with_z:
            K.right_solve3(self.zraw, self.beta0, self.z)

    def _ritz(self, steps, first, floor_level, gap=4):
        """Enqueue the device Ritz step for the first `steps` blocks and the asynchronous read-back of its
        verdict; returns the handle to wait on.  gap: Lanczos steps since the previous check of this run - a residual
        counts as stalled (rounding floor of f32 blocks) if it fell by less than 2x over one step / 4x over several."""
        flags = (1 if first else 0) | (2 if steps >= self.m_max else 0)
        self.K.ritz(self.HB, self.hw, steps, flags, self.eig_tol, self.floor_tol, floor_level, self.Yd, self.status, self.gate,
                    stall_ratio=0.5 if gap <= 1 else 0.25)
        self.stats["n_check"] = self.stats.get("n_check", 0) + 1
        return self.K.post_status(self.status)

    def spectral(self, x0, warm=False, it=None, tail=None):
        """3 algebraically smallest eigenvectors of L = Lambda_C - P (up to a 3x3 mixing,
        which the gauge fix removes) -> self.X.  Returns eigenvalue estimates (host array).

        The Ritz step (projected eigenproblem, residuals, stop/converged decision) runs on the device and
        raises a gate flag; `tail` (the rest of the primal-dual iteration) is enqueued right behind it under
        that gate BEFORE the host learns the verdict, so a converged check costs no pipeline bubble and a
        failed one cancels the speculative work on the device."""
        K, n, ld = self.K, self.n, self.ld
        total_steps, floor_at = 0, 0
        th4_last = float("nan")                     # fourth smallest Ritz value of the most recent check that had one
        for restart in range(self.max_restarts + 1):
            have_z = restart == 0 and self.z_ready
            self._seed_block(x0, with_z=have_z)
            self.z_ready = False
            steps = 0
            # a failed projection check costs a pipeline bubble (host round trip) against >= one edge sweep
            # per extra step, so warm-started iterations (previous R_c: already ~1e-3 from the answer) are
            # checked early and then every `check_every` steps
            next_check = min(self.warm_min_steps if (warm and restart == 0) else self.min_steps, self.m_max)
            probing = False
            if restart == 0 and it in self.pred_steps:
                # the same graph was solved before (time series, benchmark loop): go straight to the step
                # count that sufficed last time instead of paying for checks that are known to fail
                next_check = min(max(self.pred_steps[it], 1), self.m_max)
                # Capture-sized graphs are checked every FOUR steps (a Ritz call costs two Lanczos steps there), so the remembered
                # count is the smallest multiple of the check spacing that sufficed, not the smallest count: later solves of the
                # same object try ONE STEP FEWER each until a first check fails - that solve takes the step back and checks again
                # (one Ritz call more, once), and the count stays.  Every solve still ends on a passed check of the same rule.
                # (large_shop [8, 5, 4, 4] -> [8, 5, 3, 2], small_room-sized graphs 21 -> 12 steps per solve.)
                if self.small_graph and next_check > 1 and not self._probe_done.get(it, False):
                    next_check -= 1
                    probing = True
            # residual level of the rounding floor: remembered per iteration index, else the one met by the previous
            # iteration (a property of the f32 products, not of the iterate)
            level = -1.0
            if restart == 0 and it is not None:
                level = self.floor_level.get(it, self.floor_level.get(it - 1, -1.0))
            prev_res, floor_hit, prev_steps, first = None, False, 0, True
            while True:
                j = steps
                in_slabs = False
                one_call = (not (j == 0 and have_z) and not self.comm.sharded and self.fold_in_step and self.N == self.C
                            and hasattr(K, "lanczos_step_slabs"))
                if one_call:
                    # single rank, few slabs: sweep and cooperative camera-side step behind ONE host call (capture-sized graphs
                    # are bound by the launch rate of the host once their kernels take 10-15 us each)
                    self.stats["sweeps"] += 1
                    if not K.lanczos_step_slabs(self.lamT, self.lamC, self.V, ld, j, self.HB[j, : self.hw], self.HB[j, self.hw:],
                                                self.xrow, self.pivot_floor):
                        K.fold_z(self.z)                       # (cooperative grid refused: the sweep has run - fold, launch sequence)
                        K.lanczos_cam_step(self.lamC, self.V, ld, j, self.z, self.R, self.H, self.G, self.HB[j, : self.hw],
                                           self.HB[j, self.hw:], self.xrow, self.pivot_floor)
                elif not (j == 0 and have_z):
                    # (j == 0 with a warm start from rc: P rc was formed by the fused dual update of the previous
                    #  primal-dual iteration - one pass over the blocks instead of two - and normalised by the seed)
                    in_slabs = bool(self.apply_P(self.xrow, self.z))
                # camera side of the step: A Q_j = Lambda_C Q_j - z, two Gram-Schmidt passes against the
                # whole basis (coefficients -> Hbuf[j]), Cholesky-QR -> block j+1, beta_j, next sweep input
                if not one_call:
                    K.lanczos_cam_step(self.lamC, self.V, ld, j, self.z, self.R, self.H, self.G, self.HB[j, : self.hw],
                                       self.HB[j, self.hw:], self.xrow, self.pivot_floor, **({"from_slabs": True} if in_slabs else {}))
                steps += 1
                total_steps += 1
                if steps >= next_check or steps >= self.m_max:
                    handle = self._ritz(steps, first, level, gap=steps - prev_steps)
                    first = False
                    # Sharded runs whose all-reduces cannot be gated (RCCL, torch.distributed: the collective runs whatever the
                    # device decided) speculate only where a check is expected to pass (a step count remembered from an earlier
                    # solve of this graph): every failed check would pay for two collectives on cancelled data.  The peer
                    # exchange IS gated (a cancelled launch advances nothing on any rank): those runs speculate like one rank.
                    speculate = (not self.comm.sharded or getattr(self.comm, "gateable", False)
                                 or (restart == 0 and it in self.pred_steps and steps >= self.pred_steps[it]))
                    if speculate:
                        with K.gated(self.gate):               # speculative: runs iff the device says converged
                            K.tall_combine(n, self.V, ld, 3 * steps, self.Yd, self.X)
                            if tail is not None:
                                tail()
                    st = K.wait_status(handle)
                    if not speculate and st[2] != 0 and st[3] != 0:       # stop and converged: the tail, now that it is known
                        K.tall_combine(n, self.V, ld, 3 * steps, self.Yd, self.X)
                        if tail is not None:
                            tail()
                    r, stop, conv, floor_hit, eff, breakdown = st[0], st[2] != 0, st[3] != 0, st[4] != 0, int(st[5]), st[6] != 0
                    if st[15] == st[15]:
                        th4_last = float(st[15])
                    # counterpart of the five values the reference's eigs(k=5, sigma=-1e-6) returns (bipgo.py:288-292)
                    self.small5 = np.array([st[7], st[8], st[9], st[15], st[13]])
                    # noise floor: with f32 blocks the products carry ~6e-8 relative rounding, so the Ritz
                    # residual stalls somewhere below `floor_tol`; a stalled residual there is converged
                    # (the rule itself is evaluated by vican_ritz; here only the bookkeeping for later solves)
                    if floor_hit and restart == 0 and it is not None and not (it in self.pred_steps and steps > self.pred_steps[it]):
                        # remember where the floor was first reached (the earlier of the two checks) and its level
                        # (not from a solve that ran past the remembered step count: see pred_steps below)
                        self.floor_level[it] = max(r, prev_res) if prev_res is not None else max(r, self.floor_level.get(it, r))
                        floor_at = prev_steps if (prev_res is not None and prev_res <= 2.0 * self.floor_level[it]) else steps
                    prev_res, prev_steps = r, steps
                    if probing:                                  # (the first check of this run was the probe)
                        probing = False
                        if not (stop and conv):
                            self._probe_done[it] = True          # one step fewer does not do: the remembered count is the smallest
                            if not stop:
                                next_check = min(steps + 1, self.m_max)
                                continue
                    if stop:
                        break
                    # a failed check costs less than one edge sweep on large graphs (device Ritz step, cancelled
                    # speculative launches), so the first few steps are checked one by one
                    # ... and so is the step after a check that failed inside the band where an f32 rounding floor can
                    # sit (r <= floor_tol): a stalled residual is then recognised one step later, not check_every later
                    near_floor = self.floor_tol > 1e-12 and r <= self.floor_tol
                    next_check = min(steps + (1 if (steps < 8 and not self.small_graph) or near_floor else self.check_every), self.m_max)
            th = st[7:12].copy()
            self.th4 = th4_last                         # (a restart from converged Ritz vectors may exhaust its Krylov space
                                                        #  after one block: that check has no fourth value of its own)
            if conv:
                self.tail_done = tail is not None
                break
            # finished without convergence (step budget or truncated basis): restart from the Ritz vectors
            K.tall_combine(n, self.V, ld, 3 * steps, self.Yd, self.X)
            if restart == self.max_restarts:
                if tail is not None:
                    tail()
                break
            x0 = self.X.clone()
            self.stats["restarts"] += 1
        self.stats["lanczos_steps"].append(total_steps)
        if it is not None and self.stats["restarts"] == 0:
            # The count to go straight to next time.  A solve that needed MORE steps than the remembered count is not believed at
            # once: on this hardware about one solve in a few hundred runs its eigen-iterations into a stalled residual and takes
            # 3-6 extra steps (a rare timing-dependent event in the rotation-stage kernels, present since round 5; the result is
            # still converged and checked) - adopting that count made every later solve of the same object pay 3-7 extra sweeps
            # (one benchmark run in five lost 15 % that way).  The larger count is adopted only when the first check at the
            # remembered one fails in two solves in a row (the data did change: a time series drifting); a one-off costs one solve.
            want = floor_at if (floor_hit and floor_at > 0) else total_steps
            have = self.pred_steps.get(it)
            if have is None or want <= have:
                self.pred_steps[it], self._pred_fail[it] = want, 0
            else:
                self._pred_fail[it] = self._pred_fail.get(it, 0) + 1
                if self._pred_fail[it] >= 2:
                    self.pred_steps[it], self._pred_fail[it] = want, 0
                    self._probe_done[it] = True                  # (the data moved: no more probing below a count that just grew)
        self.stats["resid"].append(float(r))
        self.stats["evals"].append(th)               # 3 smallest + 2 largest Ritz values (cf. eigs k=5)
        return th

    # -- full primal-dual loop -------------------------------------------------
    def init(self):
        K = self.K
        K.init_duals(self.lamT, self.cam_deg)
        first = self.x0 is None
        if first:
            self._deg_msg[self.N] = float(getattr(getattr(K, "g", None), "n_edges", 0) or 0)
            self._deg_msg[self.N + 1] = float(K.T)
        self.comm.allreduce(self._deg_msg if first else self.cam_deg)
        K.scaled_identity(self.cam_deg, self.lamC)
        if first:                                               # graph constants: once per solver object
            h = self._deg_msg.cpu().numpy()                     # one-off host read
            lscale = float(h[: self.N].max())                   # |L| <~ 2 max deg
            self.pivot_floor = (1e-12 * lscale) ** 2
            if self.prop_sweeps is None:
                # as many sweeps as it takes the propagation to reach every camera: MEASURED by the first solve (_propagated_start
                # counts the cameras its sweeps have not reached yet); until then an estimate - one where a camera shares
                # timesteps with several times C others (dense captures), three on sparse co-visibility graphs
                self._prop_auto = True
                n_e, T_ = float(h[self.N]), max(float(h[self.N + 1]), 1.0)
                self.prop_sweeps = 3
                if n_e > 0:
                    hops1 = (n_e / max(self.C, 1)) * max(n_e / T_ - 1.0, 0.0) / max(self.C, 1)     # expected co-visible cameras / C
                    self.prop_sweeps = 1 if hops1 >= 4.0 else (2 if hops1 >= 0.5 else 3)
            if self.prop_sweeps > 0 and self.N == self.C:
                self._x_seed = K.zeros(self.n, 3)
                self._x_seed[:3] = torch.eye(3, dtype=torch.float64, device=self._x_seed.device)
                self.x0 = K.empty(self.n, 3)
            else:
                g = torch.Generator(device="cpu"); g.manual_seed(self.seed)
                self.x0 = torch.randn(self.n, 3, generator=g, dtype=torch.float64).to(self.X.device)
        if self.prop_sweeps > 0 and self.N == self.C:
            self._propagated_start()                            # (part of every solve: three sweeps, not a cached constant)

    def _propagated_start(self):
        """Start block of the first eigen-solve (the reference's ARPACK call starts from a random vector, bipgo.py:288; the
        answer does not depend on the start, the number of operator applications does).  In the noise-free case the three
        wanted eigenvectors ARE the stacked camera rotations, P x = Lambda_C x, so rotations propagated from the gauge camera
        through the power graph - x <- polar(P x) per camera, starting from the identity at camera 0 and zero elsewhere: sweep
        k reaches the cameras k co-visibility hops away - are a start as good as the measurement noise, i.e. as good as the
        warm starts of the later iterations.  Three sweeps instead of 6-7 Lanczos steps (CPU prototype on the goldens: first
        eigen-solve 12 -> 5 steps on small_room-sized scenes, 16 -> 10 on g3; final rotations unchanged to 1e-13 rad).
        Cameras not reached keep the identity: a worse start for them, never a wrong answer."""
        K, x = self.K, self.x0
        x.copy_(self._x_seed)
        if getattr(self, "_prop_auto", False):
            # First solve of this solver object: sweep until EVERY camera has been reached (its block of P x is non-zero) - the
            # eccentricity of the gauge camera in the co-visibility graph, a property of the graph, remembered for the later solves.
            # A camera the start block knows nothing about costs the eigen-solve one Lanczos step per hop it is away (the Krylov
            # space grows by one hop per step), and a step is a sweep PLUS the camera-side step and a larger Ritz problem:
            # large_shop (340 cameras, 4 per timestep: 8 hops) 2 sweeps + 8 steps -> 8 sweeps + 4 steps, 1.40 -> 1.34 ms.
            # One host read per sweep, this once; sharded runs read the all-reduced block: the same count on every rank.
            k, prev = 0, None
            while True:
                self._op_allreduce(self.lamT, x, self.z)
                unreached = int((self.z.reshape(self.C, 9).abs().amax(dim=1) == 0).sum())
                K.polar_dual(self.z, x, None, 0)
                self.stats["sweeps"] += 1
                k += 1
                if unreached == 0 or k >= 32 or (prev is not None and unreached >= prev):      # (no progress: a disconnected graph)
                    break
                prev = unreached
            self.prop_sweeps, self._prop_auto = k, False
            self.start_is_warm = True
            return
        for _ in range(self.prop_sweeps):
            self._op_allreduce(self.lamT, x, self.z)
            K.polar_dual(self.z, x, None, 0)                    # (nearest rotation per camera, det fix: geometry.py:175-191)
            self.stats["sweeps"] += 1
        self.start_is_warm = True

    def _tail(self, fuse=False):
        """Everything of a primal-dual iteration after the eigen-solve (enqueued under the Ritz gate).
        fuse: another iteration follows - its eigen-solve starts from rc, and P_new rc comes out of the dual
        update's own pass over the blocks (lamT_new Z_t is the polar factor of Z_t)."""
        K = self.K
        K.gauge_project(self.X, self.Xp)                        # bipgo.py:295-297
        self._op_allreduce(self.lamT, self.Xp, self.z)          # bipgo.py:300
        K.polar_dual(self.z, self.rc, self.lamC, 1)             # bipgo.py:306-315
        if fuse:
            K.dual_update_op(self.rc, self.Rt, self.lamT, self.zraw)   # bipgo.py:318-332 (+ :285-288 of the next iteration)
            self.comm.allreduce(self.zraw)
        else:
            K.dual_update(self.rc, self.Rt, self.lamT)          # bipgo.py:318-332

    def iterate(self, first, it=None, last=True):
        fuse = self.fuse_dual_op and not last
        self.z_ready = self.z_ready and not first
        self.spectral(self.x0 if first else self.rc, warm=(not first) or self.start_is_warm, it=it, tail=lambda: self._tail(fuse))
        self.z_ready = fuse
        self.stats["sweeps"] += 2

    def run(self, maxiter):
        if maxiter < 1:
            # the reference leaves its loop without ever binding r_c and dies at bipgo.py:346
            raise UnboundLocalError("local variable 'r_c' referenced before assignment")
        self.init()
        tol_final = self.eig_tol
        self.small5 = None
        for it in range(maxiter):
            # bipgo.py:283: the reference ends its loop once all five returned eigenvalues are <= 1e-6 in magnitude
            # (>= 5 near-null vectors: only graphs with several components).  The five smallest Ritz values bound
            # the five smallest eigenvalues from above, so the exit taken here is one the reference takes too; the
            # converse can fail while theta_4, theta_5 are unconverged (disconnected graphs).
            if self.small5 is not None and np.all(np.isfinite(self.small5)) and np.abs(self.small5).max() <= 1e-6:
                self.stats["early_exit"] = it
                break
            # The outer primal-dual iteration contracts errors of earlier spectral steps by orders of
            # magnitude per iteration (checked against the reference goldens: the schedule below leaves the
            # final rotations within 1e-12 rad of the fully converged variant), so only the last two spectral
            # steps are solved to the full tolerance; each earlier one is relaxed by 100x, at most to 1e-4.
            relax = max(0, (maxiter - 2) - it)
            self.eig_tol = min(max(tol_final * 100.0 ** relax, tol_final), 1e-4)
            self.iterate(it == 0, it, last=(it == maxiter - 1))
        self.eig_tol = tol_final
        return self.rc, self.Rt


class GeneralRotationSolver(RotationSolver):
    """The reference's non-eliminated variant ``bipartite_so3sync`` (bipgo.py:94-133): primal-dual iteration on
    ALL C+T nodes (cameras first, then timesteps - np.unique order of 'c..'/'t..' names, bipgo.py:54), full 3x3
    dual blocks Lambda_i = U S U^T on every node, no det fix in the final polar step (bipgo.py:126-127).

    Same machinery as RotationSolver - block Lanczos on L = Lambda - R~ with the device Ritz step and the gated
    speculative tail - with the operator  R~ [x_c; x_t] = [sum_t M_ct x_t ; sum_c M_ct^T x_c]  evaluated in one pass
    over the blocks (``bip_apply``).  Single rank: the timestep-side vectors are as large as the camera side here,
    so there is no cheap sharding."""

    def __init__(self, K, comm=None, **kw):
        comm = comm or Comm()
        if comm.world != 1:
            raise ValueError("bipartite_so3sync runs on a single rank (no timestep sharding in the non-eliminated variant)")
        super().__init__(K, comm, n_nodes=K.C + K.T, **kw)

    def apply_P(self, x, z):
        self.K.bip_apply(x, z)
        self.stats["sweeps"] += 1

    def init(self):
        K = self.K
        K.node_degrees(self.cam_deg)                            # bipgo.py:95-99
        K.scaled_identity(self.cam_deg, self.lamC)
        K.bip_scales()
        if self.x0 is None:
            lscale = float(self.cam_deg.max())
            self.pivot_floor = (1e-12 * lscale) ** 2
            g = torch.Generator(device="cpu"); g.manual_seed(self.seed)
            self.x0 = torch.randn(self.n, 3, generator=g, dtype=torch.float64).to(self.X.device)

    def _tail(self):
        K = self.K
        K.gauge_project(self.X, self.Xp)                        # bipgo.py:113-116
        K.bip_apply(self.Xp, self.z)                            # bipgo.py:119
        K.polar_dual(self.z, self.rc, self.lamC, 5)             # bipgo.py:125-131 (U V^T without det fix, U S U^T)

    INTERIOR_MAX_N = 6144        # largest 3(C+T) the interior regime forms a dense Laplacian for (288 MB, one eigh)
    SIGMA = -1e-6                # the reference's shift (bipgo.py:106)

    class _Interior(Exception):
        pass

    def iterate(self, first, it=None):
        if getattr(self, "interior", False):
            self._interior_step()
            return
        th = self.spectral(self.x0 if first else self.rc, warm=not first, it=it, tail=self._tail)
        self.stats["sweeps"] += 1
        # The reference asks ARPACK for the eigenvalues CLOSEST TO -1e-6 (shift-invert, bipgo.py:106); that is the
        # bottom of the spectrum - what the Lanczos iteration delivers - only while no eigenvalue lies further
        # below zero than the fourth one lies above it.  Once the dual iterate makes L strongly indefinite (noisy
        # multi-marker graphs: ~100 negative eigenvalues, golden g3) the reference's three eigenvectors are INTERIOR ones and
        # the iteration is a chaotic map (an eigenvector error of 4e-8 in the FIRST step - this solver's tolerance over a gap
        # of 0.02 - shows as 1e-6 in the next step's eigenvalues and as O(1) in the answer): the run starts over with every
        # step on the dense Laplacian, decomposed to rounding (_interior_step).
        if self.th4 == self.th4 and th[0] < 0.0 and -th[0] > self.th4:
            if self.n > self.INTERIOR_MAX_N or not getattr(self.K, "storage_f64", False):
                # (float32: the reference runs ARPACK in single precision there and the chaotic iteration amplifies its 1e-7 to
                #  O(0.1) - its answer differs from run to run with ARPACK's random start vector: nothing to reproduce)
                raise ArithmeticError("bipartite_so3sync: the connection Laplacian became indefinite (smallest eigenvalue "
                                      "%.3e, fourth %.3e); the reference's eigs(sigma=-1e-6) selects interior eigenvalues here, "
                                      "which this solver reproduces on a dense float64 matrix of up to %d unknowns only (this graph: "
                                      "%d unknowns, %s blocks)" % (th[0], self.th4, self.INTERIOR_MAX_N, self.n,
                                                                   "float64" if getattr(self.K, "storage_f64", False) else "float32"))
            self.interior = True
            self.stats["interior_from"] = len(self.stats["evals"]) - 1
            raise self._Interior()

    def _interior_step(self):
        """One primal-dual iteration where L = Lambda - R~ is indefinite (bipgo.py:101-133 as they are): the reference's
        eigs(k=5, sigma=-1e-6) returns the eigenvalues CLOSEST TO sigma and its columns 0..2 span what the iteration goes on
        with - interior eigenvectors, out of reach of a Lanczos iteration on L.  The legacy variant is small where this happens
        (the reference builds its matrices in Python loops): L is formed densely - R~ column block by column block through the
        backend's own one-pass operator (bip_apply on an identity block per node), the dual blocks on the diagonal, symmetrised as
        bipgo.py:103 - and decomposed by one symmetric eigen-solve (torch.linalg.eigh: LAPACK on the stand-in backend, rocSOLVER on
        the GPU; plumbing for a regime that is dead code upstream, not the hot path).  Only the SPAN of the three vectors matters
        (r = V V_0^-1, bipgo.py:113), so any orthonormal basis of it reproduces the reference."""
        K, N, n = self.K, self.N, self.n
        dev = self.X.device
        edges = getattr(self, "dense_edges", None)
        if edges is not None:
            # the merged blocks as the caller holds them (host CSR: row_ptr [T+1], col [E], blk [E][9], float64): R~ placed entry by
            # entry - exact, where the operator's fixed-point sums carry 1e-14 that this regime amplifies to 2e-7 in the answer
            row_ptr, col, blk = (np.asarray(a) for a in edges)
            C_, T_ = K.C, N - K.C
            rows = np.repeat(np.arange(T_), np.diff(row_ptr))
            Rh = np.zeros((N, 3, N, 3))
            M = blk.reshape(-1, 3, 3).astype(np.float64)
            Rh[col, :, C_ + rows, :] = M
            Rh[C_ + rows, :, col, :] = np.swapaxes(M, 1, 2)
            Rd = torch.from_numpy(Rh.reshape(n, n)).to(dev)
        else:
            Rd = torch.zeros(n, n, dtype=torch.float64, device=dev)
            x, z = K.empty(n, 3), K.empty(n, 3)
            eye = torch.eye(3, dtype=torch.float64, device=dev)
            for i in range(N):                                    # column block i of R~ (|x_i|_F = sqrt(3): inside bip_apply's bound)
                x.zero_()
                x[3 * i: 3 * i + 3] = eye
                K.bip_apply(x, z)
                Rd[:, 3 * i: 3 * i + 3] = z
            self.stats["sweeps"] += N
        L = -Rd
        idx = torch.arange(N, device=dev)
        L.view(N, 3, N, 3)[idx, :, idx, :] += self.lamC.view(N, 3, 3)
        L = 0.5 * (L + L.T)                                       # bipgo.py:103
        w, V = torch.linalg.eigh(L)
        order = torch.sort((w - self.SIGMA).abs(), stable=True).indices[:5]
        self.X.copy_(V[:, order[:3]])
        ev = w[order].cpu().numpy()
        self.small5 = ev.copy()
        self.stats["evals"].append(ev)
        self.stats["lanczos_steps"].append(0)
        self.stats["resid"].append(0.0)
        self._tail()
        self.stats["sweeps"] += 1

    def run(self, maxiter):
        # no relaxed early tolerances here: this iteration contracts the error of an earlier spectral step only
        # ~5-10x per iteration (checked on the goldens), so every step is solved to the full tolerance
        self.init()
        try:
            for it in range(maxiter):
                self.iterate(it == 0, it)
        except self._Interior:
            # (iterate: the Laplacian turned indefinite - every step again, on the dense matrix)
            for k in ("evals", "lanczos_steps", "resid"):
                del self.stats[k][:]
            self.init()
            for it in range(maxiter):
                self._interior_step()
        return self.rc


class TranslationSolver:
    """CG on the normal equations (weighted bipartite Laplacian (x) I3), scipy semantics."""

    def __init__(self, K, comm=None, rtol=1e-5, poll_every=8):
        self.K, self.comm = K, comm or Comm()
        self.rtol, self.poll_every = rtol, poll_every
        C, T = K.C, max(K.T, 1)
        # the camera-side set-up quantities share one buffer: ONE all-reduce of [deg_c | b_c] in sharded runs
        self._setup_msg = K.empty(4 * C)
        self.deg_t, self.deg_c = K.empty(T), self._setup_msg[:C]
        self.b_c, self.b_t = self._setup_msg[C:].view(C, 3), K.empty(T, 3)
        self.x_c, self.r_c, self.p_c = K.empty(C, 3), K.empty(C, 3), K.empty(C, 3)
        self.x_t, self.r_t, self.p_t, self.q_t = K.empty(T, 3), K.empty(T, 3), K.empty(T, 3), K.empty(T, 3)
        self.qcpq = K.zeros(3 * C + 1)
        self.st = K.zeros(CG_STATE_DOUBLES)
        self.info = {}
        n_edges = getattr(getattr(K, "g", None), "n_edges", None)
        self.small_graph = n_edges is not None and n_edges < 2_000_000
        self._graphs, self._n_solves, self._last_iters = {}, 0, None
        # Sharded runs keep scipy's own recurrence (bipgo.py:477) - two all-reduces per iteration, [q_c | p.q] and r.r - as
        # SURVEY.md 8(e) prescribes for parity mode.  VICAN_CG_MESSAGES=1 opts into the Chronopoulos-Gear arrangement (ONE
        # message per iteration, other roundings: measurably further from scipy's iterates, profiles/r04_random_parity_3000_summary.json)
        self.one_message = os.environ.get("VICAN_CG_MESSAGES", "2") == "1"
        self._cg1 = None

    def _state(self):
        h = self.st.cpu()
        hi = h.view(torch.int32)
        return {**{k: float(h[i]) for k, i in CG_F.items()}, **{k: int(hi[i]) for k, i in CG_I.items()}}

    def setup(self, rc, rt):
        K = self.K
        K.trans_degrees(self.deg_t, self.deg_c)
        K.trans_rhs(rc, rt, self.b_t, self.b_c)
        self.comm.allreduce(self._setup_msg)

    def solve(self, n_unknowns_total, maxiter=None, stop_at=None):
        """stop_at: run exactly this many iterations, whatever the residual (the state reports done = 0) - what an
        iteration-matched comparison with scipy's iterates needs (tests/test_cg_iterates.py)."""
        K, comm, st = self.K, self.comm, self.st
        multi = comm.sharded
        maxiter = 10 * n_unknowns_total if maxiter is None else maxiter       # scipy default
        rtol = self.rtol
        if stop_at is not None:
            return self._solve_fixed(int(stop_at))
        if not multi and self.small_graph and getattr(K, "cg_resident_ok", False) and not K.barrier_aborted():
            # capture-sized graphs: the whole solve as one cooperative launch (an iteration of the multi-kernel path is
            # four dependent launches of a few microseconds each - launch latency only)
            K.cg_resident(self.deg_t, self.deg_c, self.b_c, self.b_t, self.x_c, self.x_t, self.rtol, maxiter, st)
            s = self._state()
            if s["done"] != -1 and not K.barrier_aborted():
                self._n_solves += 1
                self.info = dict(cg_iters=s["iter"], converged=s["done"] == 1, resident=True,
                                 relres=float(np.sqrt(s["rho"] / s["bnorm2"])) if s["bnorm2"] > 0 else 0.0)
                return self.x_c, self.x_t
            # done == -1: a grid barrier of the cooperative kernel was not passed within its time limit (the device is shared
            # with something that kept part of the grid out): nothing was lost, the launch sequence below solves the system
            K.cooperative_failed("vican_cg_resident")
        K.cg_init(self.b_c, self.b_t, self.x_c, self.x_t, self.r_c, self.r_t, self.p_c, self.p_t, st)
        if multi and self.one_message and getattr(K, "cg1_iter_local", None) is not None:
            return self._solve_one_message(maxiter)
        if multi:
            comm.allreduce(st[CG_F["rr_time"]:CG_F["rr_time"] + 1])
        n_part, it_launched, s = None, 0, None          # (None: the first iteration)

        # single rank: one host call per iteration with p_t.q_t formed over fixed slices - bit-reproducible from run to run,
        # which the sequence below is not (its p.q partial depends on the sweep's ticket order).  use_fused = False: the sequence
        # (what backends without the fused call - camera tiles - take; the tests compare the two)
        fused = (not multi) and getattr(K, "cg_iter_fused", None) is not None and getattr(self, "use_fused", True)
        comm_iter = multi and self._comm_iter_ok()

        def one_iteration(n_part):
            if fused:
                K.cg_iter_fused(self.deg_t, self.deg_c, self.r_c, self.p_c, self.x_c, self.r_t, self.p_t, self.q_t, self.x_t, self.qcpq,
                                self.rtol, st, first=(n_part is None))
                return 0
            if comm_iter:
                # sharded: the same launches with the two sums that cross ranks all-reduced from C in stream order, partials over
                # fixed slices (bit-reproducible, bit-identical on every rank) - include/vican_hip.h: vican_cg_iter_comm
                K.cg_iter_comm(self.deg_t, self.deg_c, self.r_c, self.p_c, self.x_c, self.r_t, self.p_t, self.q_t, self.x_t, self._msg,
                               self.rtol, st, first=(n_part is None), comm=comm)
                if not getattr(K, "comm_iter_host", False):     # (the C library's all-reduces are not seen by comm.allreduce)
                    comm.n_allreduce += 2
                return 0
            n_part = n_part or 0
            K.cg_iter_local(self.deg_t, self.r_c, self.p_c, self.r_t, self.p_t, self.q_t, self.qcpq, self.rtol, st, n_part)
            if multi:
                comm.allreduce(self.qcpq)
            n_part = K.cg_iter_finish(self.deg_c, self.qcpq, self.p_c, self.x_c, self.r_c, self.p_t, self.q_t,
                                      self.x_t, self.r_t, st)
            if multi:
                K.cg_end(n_part, st)
                comm.allreduce(st[CG_F["rr_time"]:CG_F["rr_time"] + 1])
                n_part = 0
            return n_part

        # On small graphs an iteration is five kernels of a few microseconds each - launch-bound from Python.
        # All per-iteration quantities (alpha, beta, norms, the done flag) live in the device state struct, so
        # every iteration after the first is the SAME sequence of launches: recorded once into a HIP graph and
        # replayed (single rank only; the graph is kept while the buffers it names stay alive).  Recording costs a
        # few milliseconds, so only a solver object that is used again (time series, benchmark loop) does it.
        use_graph = self.small_graph and not multi and hasattr(K, "capture") and self._n_solves >= 1
        self._n_solves += 1
        first_burst = True
        while True:
            # (scipy: `for iteration in range(maxiter)` - at most maxiter updates of x, no test behind the last one)
            burst = min(self.poll_every, maxiter - it_launched)
            if first_burst and self._last_iters is not None:
                # the same system was solved before (time series, benchmark loop): launch exactly as many iterations
                # as it took then, plus the one that detects convergence, before the first poll
                # (the fused iteration detects convergence in the launch that made the last update)
                burst = min(max(self._last_iters + 1, 1), 64, maxiter - it_launched)
            first_burst = False
            left = burst
            if it_launched == 0:                                  # the first iteration passes other arguments
                n_part = one_iteration(n_part)
                it_launched += 1
                left -= 1
            if use_graph and left > 0:
                # the rest of the burst as ONE graph launch (iterations past convergence cancel themselves)
                key = (n_part, left, float(self.rtol), self.deg_t.data_ptr(), self.deg_c.data_ptr())
                if key not in self._graphs:
                    npart = n_part

                    def burst_fn(npart=npart, left=left):
                        for _ in range(left):
                            one_iteration(npart)
                    self._graphs[key] = K.capture(burst_fn)
                self._graphs[key].replay()
                it_launched += left
            else:
                for _ in range(left):
                    n_part = one_iteration(n_part)
                    it_launched += 1
            s = self._state()
            if s["done"] or it_launched >= maxiter:
                break
            self.poll_every = min(self.poll_every * 2, 64)
        self._last_iters = int(s["iter"]) if s["done"] == 1 else None
        self.info = dict(cg_iters=s["iter"] if s["done"] else it_launched, converged=s["done"] == 1,
                         relres=float(np.sqrt(s["rho"] / s["bnorm2"])) if s["bnorm2"] > 0 else 0.0)
        return self.x_c, self.x_t


    def _comm_iter_ok(self):
        """Sharded solve whose all-reduces the C library enqueues itself (solver.Comm with a native communicator, or a forced
        one-rank Comm: identity) on a backend with vican_cg_iter_comm: one host call per iteration."""
        K, comm = self.K, self.comm
        if getattr(K, "cg_iter_comm", None) is None or not hasattr(comm, "native_handle") or not getattr(self, "use_comm_iter", True):
            return False                                         # (use_comm_iter = False: the launch sequence with host-issued all-reduces)
        host_side = bool(getattr(K, "comm_iter_host", False))   # (the backend issues the two all-reduces through comm.allreduce itself)
        if comm.world > 1 and not getattr(comm, "_native_tried", True) and not host_side:
            comm._setup_native(self.st.device)
        if comm.native_handle() is None and comm.world > 1 and not host_side:
            return False
        if getattr(self, "_msg", None) is None:
            from ._lib import CG_PQ_SLICES
            self._msg = K.zeros(3 * K.C + CG_PQ_SLICES)
        return True

    def _solve_fixed(self, n_iter):
        """Exactly n_iter iterations of the recurrence this solver would run (resident kernel, launch sequence or the
        sharded runs' one-message arrangement), stopping test disabled (rtol = 0)."""
        K, comm, st = self.K, self.comm, self.st
        multi = comm.sharded
        if not multi and self.small_graph and getattr(K, "cg_resident_ok", False):
            K.cg_resident(self.deg_t, self.deg_c, self.b_c, self.b_t, self.x_c, self.x_t, 0.0, n_iter, st)
            s = self._state()
            if s["done"] == -1:
                raise RuntimeError("vican_cg_resident: grid barrier timed out")
            self.info = dict(cg_iters=s["iter"], converged=False, resident=True, fixed=True,
                             relres=float(np.sqrt(s["rho"] / s["bnorm2"])) if s["bnorm2"] > 0 else 0.0)
            return self.x_c, self.x_t
        K.cg_init(self.b_c, self.b_t, self.x_c, self.x_t, self.r_c, self.r_t, self.p_c, self.p_t, st)
        if multi and self.one_message and getattr(K, "cg1_iter_local", None) is not None:
            rtol, last = self.rtol, self._last_iters
            self.rtol, self._last_iters = 0.0, None
            try:
                # (bursts of the polling loop overshoot: bound them to what is asked for)
                self.poll_every, keep = n_iter, self.poll_every
                out = self._solve_one_message(n_iter)
                self.poll_every = keep
            finally:
                self.rtol, self._last_iters = rtol, last
            self.info["fixed"] = True
            return out
        if multi:
            comm.allreduce(st[CG_F["rr_time"]:CG_F["rr_time"] + 1])
        if multi and self._comm_iter_ok():
            for k in range(n_iter):
                K.cg_iter_comm(self.deg_t, self.deg_c, self.r_c, self.p_c, self.x_c, self.r_t, self.p_t, self.q_t, self.x_t, self._msg,
                               0.0, st, first=(k == 0), comm=comm)
                if not getattr(K, "comm_iter_host", False):
                    comm.n_allreduce += 2
            s = self._state()
            self.info = dict(cg_iters=n_iter, converged=False, fixed=True,
                             relres=float(np.sqrt(s["rho"] / s["bnorm2"])) if s["bnorm2"] > 0 else 0.0)
            return self.x_c, self.x_t
        n_part = 0
        for _ in range(n_iter):
            K.cg_iter_local(self.deg_t, self.r_c, self.p_c, self.r_t, self.p_t, self.q_t, self.qcpq, 0.0, st, n_part)
            if multi:
                comm.allreduce(self.qcpq)
            n_part = K.cg_iter_finish(self.deg_c, self.qcpq, self.p_c, self.x_c, self.r_c, self.p_t, self.q_t, self.x_t, self.r_t, st)
            if multi:
                K.cg_end(n_part, st)
                comm.allreduce(st[CG_F["rr_time"]:CG_F["rr_time"] + 1])
                n_part = 0
        s = self._state()
        self.info = dict(cg_iters=n_iter, converged=False, fixed=True,
                         relres=float(np.sqrt(s["rho"] / s["bnorm2"])) if s["bnorm2"] > 0 else 0.0)
        return self.x_c, self.x_t

    def _solve_one_message(self, maxiter):
        """Sharded solves: ONE all-reduce per iteration, [sum_t w r_t | r.s | r.r] (3C + 2 doubles) - the Chronopoulos-Gear
        arrangement of the same CG (include/vican_hip.h: vican_cg1_iter_local / vican_cg1_iter_finish; scipy's recurrence
        needs two reductions per iteration because its beta depends on the r.r of the update that precedes the product).
        Same iterates in exact arithmetic, scipy's stopping test on the directly formed r.r; other roundings.
        Opt-in (VICAN_CG_MESSAGES=1): sharded runs default to scipy's recurrence, two messages per iteration."""
        K, comm, st = self.K, self.comm, self.st
        if self._cg1 is None:
            C, T = K.C, max(K.T, 1)
            self._cg1 = dict(msg=K.zeros(3 * C + 2), s_t=K.empty(T, 3), q_c=K.empty(C, 3), r_c2=K.empty(C, 3))
        msg, s_t, q_c = self._cg1["msg"], self._cg1["s_t"], self._cg1["q_c"]
        r_c, r_c_new = self.r_c, self._cg1["r_c2"]          # the camera residual alternates between two buffers (vican_cg1_iter_finish)
        n_part, k, s, poll = 0, 0, None, self.poll_every
        burst = poll if self._last_iters is None else min(max(self._last_iters + 1, 1), 64)
        while True:
            for _ in range(min(burst, maxiter - k)):
                K.cg1_iter_local(self.deg_t, r_c, self.r_t, s_t, msg, st, n_part)
                comm.allreduce(msg)
                n_part = K.cg1_iter_finish(k, self.deg_c, msg, r_c, r_c_new, self.p_c, q_c, self.x_c, self.r_t, s_t, self.p_t, self.q_t,
                                           self.x_t, self.rtol, st)
                r_c, r_c_new = r_c_new, r_c
                k += 1
            s = self._state()
            if s["done"] or k >= maxiter:
                break
            poll = min(poll * 2, 64)
            burst = poll
        self._n_solves += 1
        self._last_iters = int(s["iter"]) if s["done"] == 1 else None
        self.info = dict(cg_iters=s["iter"], converged=s["done"] == 1, one_message=True,
                         relres=float(np.sqrt(s["rho"] / s["bnorm2"])) if s["bnorm2"] > 0 else 0.0)
        return self.x_c, self.x_t


class TightTranslationSolver(TranslationSolver):
    """Translations converged to `rtol` (default 1e-10) instead of scipy's 1e-5 (NOT the
    reference's behaviour - its loosely converged answer is up to metres from this one on heavy-tailed
    weights).  Jacobi-preconditioned CG, realised as plain CG on the symmetrically scaled
    system S A S (S = D^-1/2): that system is again a weighted bipartite Laplacian with unit degrees and weights
    w s_c s_t, so the same device kernels run it.  Gauge as the reference: translations of all nodes sum to 0."""

    def __init__(self, K, comm=None, rtol=1e-10, poll_every=16):
        super().__init__(K, comm, rtol=rtol, poll_every=poll_every)
        self.s_c, self.s_t = K.empty(K.C), K.empty(max(K.T, 1))

    def solve(self, n_unknowns_total, maxiter=None):
        K, comm = self.K, self.comm
        K.jacobi_scale(self.deg_c, self.s_c)
        K.jacobi_scale(self.deg_t, self.s_t)
        K.row_scale(self.s_c, self.b_c)
        K.row_scale(self.s_t, self.b_t)
        deg_c, deg_t = self.deg_c, self.deg_t
        self.deg_c, self.deg_t = torch.ones_like(deg_c), torch.ones_like(deg_t)       # unit diagonal of S A S
        K.set_cg_scaling(self.s_c, self.s_t)
        try:
            super().solve(n_unknowns_total, maxiter)
        finally:
            K.clear_cg_scaling()
            self.deg_c, self.deg_t = deg_c, deg_t
        K.row_scale(self.s_c, self.x_c)
        K.row_scale(self.s_t, self.x_t)
        # CG from 0 on the scaled system leaves the null-space component (a common shift) D-weighted;
        # move it to the reference's gauge: sum over all nodes = 0
        tot = self.x_t[: K.T].sum(0)
        comm.allreduce(tot)
        shift = (tot + self.x_c.sum(0)) / (n_unknowns_total // 3)
        self.x_c -= shift
        self.x_t -= shift
        return self.x_c, self.x_t


def with_cooperative_fallback(K, comm, fn):
    """Run ``fn()`` (stages that may launch cooperative kernels) and wait for it; if a grid barrier gave up meanwhile
    (bounded spins, include/vican_hip.h: vican_set_barrier_abort - the device is shared with something that kept part of a
    cooperative grid out) the results are undefined: the backend stops using the cooperative kernels and ``fn()`` runs
    again on the launch-sequence paths.  Sharded runs cannot re-run one rank alone (the other ranks sit in collectives):
    there the abort is an error."""
    aborted = getattr(K, "barrier_aborted", None)
    if aborted is None:                                      # stand-in backends (tests/numpy_backend.py)
        return fn()
    try:
        out = fn()
        K.synchronize()
    except Exception:
        if not aborted():
            raise
        out = None
    flag = bool(aborted())
    if comm is not None and comm.world > 1:
        # every rank must take the same branch: the rank whose barrier timed out would raise while its peers return and block
        # in the next collective - the abort flag is all-reduced (MAX) so that all of them raise together
        any_flag = getattr(comm, "any_flag", None)
        if any_flag is not None:
            flag = any_flag(flag, getattr(K, "dev", None))
    if flag:
        if comm is not None and comm.world > 1:
            raise RuntimeError("a grid barrier of a cooperative kernel timed out on rank %d (device shared with another resident "
                               "kernel?); set VICAN_COOP=0 to run sharded solves without cooperative kernels" % comm.rank)
        K.cooperative_failed("grid barrier timeout")
        out = fn()
        K.synchronize()
    return out


def _lsqr_solve_device(self, beta, c2, bnorm, ctol, iter_lim, multi):
    """The iterations with every scalar on the device (vican_lsqr_state_t) and ONE fused pass over the edges per iteration
    (vican_lsqr_step): the host enqueues iterations in bursts and polls the state; nothing on the critical path is read back.
    Entered after vican_lsqr_init_u (u~_1 = b~ stored unnormalised, |b~| = beta known, v = w = x = 0).  The first
    bidiagonalisation step alfa_1 v_1 = J~^T u_1 is the same fused pass with coef = -1 on v = 0 (u^ = u~, z = J~^T u~)."""
    K, comm = self.K, self.comm
    C3 = 3 * K.C
    smax, n_add = K.lsqr_device_params()
    h = np.zeros(LSQR_STATE_DOUBLES)
    hi = h.view(np.int32)
    lo_bits = int(min(48, max(8, 62 - int(np.ceil(np.log2(max(n_add, 1.0)))))))

    def scale_for(bound):                                            # fix_scale(c, n_add, bits = 49) of vican_sweep_common.h
        cb = max(bound, 1e-300)
        e = min(49 - int(np.ceil(np.log2(cb))), 61 - int(np.ceil(np.log2(cb * max(n_add, 1.0)))))
        e = max(-1000, min(1000, e))
        return np.ldexp(1.0, e), np.ldexp(1.0, -e)
    z_t, acc = K.zeros(max(K.T, 1), 3), K.zeros(C3 + 2)          # acc: [camera sums | |u^|^2 | |w_t|^2 of the direction (sharded runs)]
    part2, wp_c, wp_t = K.zeros(1025), K.zeros(1024), K.zeros(1024)
    self.info = dict(lsqr_iters=0, istop=0, converged=True, device_scalars=True)
    if beta == 0.0:
        return self.x_c, self.x_t
    # ---- first step: |s u^| <= smax |u~| <= smax beta
    h[LSQR_F["coef"]], h[LSQR_F["smax"]], h[LSQR_F["n_add"]] = -1.0, smax, n_add
    h[LSQR_F["qscale"]], h[LSQR_F["qinv"]] = scale_for(smax * beta)
    hi[LSQR_I["lo_bits"]] = lo_bits
    st = K.from_numpy(h)
    K.lsqr_step(self.v_c, self.v_t, z_t, acc, st)
    if multi:
        comm.allreduce(acc[: C3 + 1])
    nb = K.lsqr_nodes(z_t, acc, self.v_t, self.v_c, part2, st)      # v~ = J~^T u~ / beta
    nv2 = part2[:nb].sum().reshape(1)
    if multi:
        comm.allreduce(nv2)
    alfa = float(np.sqrt(float(nv2.cpu()[0]) + float(part2[1024].cpu())))
    if alfa == 0.0:
        return self.x_c, self.x_t
    # v1 = v/alfa, w1 = v1, x = 0
    K.lsqr_update(1.0 / alfa, 0.0, 0.0, self.v_c, self.w_c, self.x_c, self.s2[3:4])
    K.lsqr_update(1.0 / alfa, 0.0, 0.0, self.v_t, self.w_t, self.x_t, self.s2[1:2])
    # ---- the state of scipy's loop
    h[:] = 0.0
    for k, v in dict(alfa=alfa, beta=beta, rhobar=alfa, phibar=beta, cs2=-1.0, c2=c2, bnorm=bnorm, atol=self.atol, btol=self.btol, ctol=ctol,
                     coef=alfa / beta, smax=smax, n_add=n_add).items():
        h[LSQR_F[k]] = v
    h[LSQR_F["qscale"]], h[LSQR_F["qinv"]] = scale_for(smax * (2.0 * smax + alfa))
    hi[LSQR_I["iter_lim"]] = int(min(iter_lim, 2 ** 31 - 1))
    hi[LSQR_I["lo_bits"]] = lo_bits
    st.copy_(K.from_numpy(h))
    # |w_1|^2 from the initial update (s2[3] cameras, s2[1] timesteps)
    wpart_t, n_wt, wpart_c, n_wc = self.s2[1:2], 1, self.s2[3:4], 1
    if multi:
        acc[C3 + 1:C3 + 2].copy_(self.s2[1:2])
    burst, launched, state = 8, 0, None
    if getattr(self, "_last_iters", None):                      # the same system was solved before: launch exactly that many first
        burst = max(1, min(self._last_iters, 256))
    while True:
        for _ in range(min(burst, max(iter_lim - launched, 1))):
            K.lsqr_step(self.v_c, self.v_t, z_t, acc, st)
            if multi:
                comm.allreduce(acc)                              # one message: camera sums, |u^|^2 and the pending |w_t|^2
            nb = K.lsqr_nodes(z_t, acc, self.v_t, self.v_c, part2, st)
            tsum = None
            if multi:
                tsum = part2[:nb].sum().reshape(1)
                comm.allreduce(tsum)
            K.lsqr_scalars(acc, part2, nb, tsum, None if multi else wpart_t, n_wt, wpart_c, n_wc, acc[C3 + 1:] if multi else None, st)
            n_wc = K.lsqr_update_st(self.v_c, self.w_c, self.x_c, wp_c, 0, st)
            n_wt = K.lsqr_update_st(self.v_t, self.w_t, self.x_t, wp_t, 1, st)
            wpart_t, wpart_c = wp_t, wp_c
            if multi:
                acc[C3 + 1:C3 + 2].copy_(wp_t[:n_wt].sum().reshape(1))
            launched += 1
        hs = st.cpu().numpy()
        state = {**{k: float(hs[i]) for k, i in LSQR_F.items()}, **{k: int(hs.view(np.int32)[i]) for k, i in LSQR_I.items()}}
        if state["done"] or launched >= iter_lim:
            break
        burst = min(2 * burst, 64)
    self._last_iters = state["itn"] if state["done"] else None
    self.info = dict(lsqr_iters=state["itn"], istop=state["istop"], converged=True, rnorm=state["rnorm"], arnorm=state["arnorm"],
                     anorm=state["anorm"], acond=state["acond"], xnorm=state["xnorm"], device_scalars=True)
    return self.x_c, self.x_t



def solve_on_backend(K, comm, maxiter, n_unknowns_total, eig_tol=1e-10, rtol=1e-5, lsqr_solver="conjugate_gradient",
                     bnorm2_fn=None, tight=False):
    """Rotation stage then translation stage on one rank's backend ``K``.
    Returns (rc [3C,3] node<-world stacked, Rt_local [T,9], x_c [C,3], x_t [T,3], stats).
    ``bnorm2_fn(rc, Rt_local) -> |b|^2`` of the reference's un-merged system (LSQR stopping tests)."""
    return with_cooperative_fallback(K, comm, lambda: _solve_on_backend(K, comm, maxiter, n_unknowns_total, eig_tol, rtol, lsqr_solver,
                                                                        bnorm2_fn, tight))


def _solve_on_backend(K, comm, maxiter, n_unknowns_total, eig_tol, rtol, lsqr_solver, bnorm2_fn, tight):
    rot = RotationSolver(K, comm, eig_tol=eig_tol)
    rc, Rt_loc = rot.run(maxiter)
    K.synchronize()
    if tight:
        tr = TightTranslationSolver(K, comm)
        tr.setup(rc, Rt_loc)
        x_c, x_t = tr.solve(n_unknowns_total)
    elif lsqr_solver == "direct":
        tr = LsqrTranslationSolver(K, comm)
        x_c, x_t = tr.solve(rc, Rt_loc, n_unknowns_total, None if bnorm2_fn is None else bnorm2_fn(rc, Rt_loc))
    else:
        tr = TranslationSolver(K, comm, rtol=rtol)
        tr.setup(rc, Rt_loc)
        x_c, x_t = tr.solve(n_unknowns_total)
    K.synchronize()
    stats = dict(rot.stats)
    stats.update(tr.info)
    return rc, Rt_loc, x_c, x_t, stats


def _sym_ortho(a, b):
    """Stable Givens rotation (c, s, r) with c*a + s*b = r (Paige & Saunders' SymOrtho)."""
    if b == 0:
        return np.sign(a), 0.0, abs(a)
    if a == 0:
        return 0.0, np.sign(b), abs(b)
    if abs(b) > abs(a):
        tau = a / b
        s = np.sign(b) / np.sqrt(1 + tau * tau)
        return s * tau, s, b / s
    tau = b / a
    c = np.sign(a) / np.sqrt(1 + tau * tau)
    return c, c * tau, a / c


class LsqrTranslationSolver:
    """LSQR (Paige & Saunders 1982) with scipy.sparse.linalg.lsqr's defaults and stopping tests
    (atol = btol = 1e-6, conlim = 1e8, iter_lim = 2n, damp = 0) - the reference's
    ``lsqr_solver="direct"`` path (bipgo.py:479-480).

    The O(E) vector work runs on the device on the MERGED system J~ p = b~ (see
    csrc/vican_lsqr.hip); it has the same normal equations as the reference's 3E' x 3N system,
    hence the same iterates, while every residual norm differs by the constant
    c^2 = |b|^2 - |b~|^2, which ``bnorm2_true`` lets this driver add back so that the stopping
    iteration is the reference's.  The Golub-Kahan / QR scalars are host doubles (two tiny
    device->host reads per iteration)."""

    def __init__(self, K, comm=None, atol=1e-6, btol=1e-6, conlim=1e8):
        self.K, self.comm = K, comm or Comm()
        self.atol, self.btol, self.conlim = atol, btol, conlim
        C, T = K.C, max(K.T, 1)
        self.v_c, self.w_c, self.x_c = K.zeros(C, 3), K.zeros(C, 3), K.zeros(C, 3)
        self.v_t, self.w_t, self.x_t = K.zeros(T, 3), K.zeros(T, 3), K.zeros(T, 3)
        self.acc = K.zeros(3 * C + 1)          # [camera sums | partial |v_t|^2]  (one all-reduce)
        self.s2 = K.zeros(4)                   # scalar slots: |u|^2, |w_t|^2, |v_c|^2, |w_c|^2
        self.info = {}

    def _get(self, t):
        return t.cpu().numpy().copy()

    def solve(self, rc, rt, n_unknowns_total, bnorm2_true=None, iter_lim=None):
        K, comm = self.K, self.comm
        multi = comm.sharded
        eps = np.finfo(np.float64).eps
        for t in (self.v_c, self.w_c, self.x_c, self.v_t, self.w_t, self.x_t):
            t.zero_()
        iter_lim = 2 * n_unknowns_total if iter_lim is None else iter_lim
        ctol = 1.0 / self.conlim if self.conlim > 0 else 0.0
        # u1 = b~ / beta1
        K.lsqr_init_u(rc, rt, self.s2[0:1])
        if multi:
            comm.allreduce(self.s2[0:1])
        beta = float(np.sqrt(self._get(self.s2)[0]))
        c2 = 0.0 if bnorm2_true is None else max(bnorm2_true - beta * beta, 0.0)
        bnorm = np.sqrt(beta * beta + c2)
        if hasattr(K, "lsqr_step") and not getattr(K, "lsqr_host_scalars", False):
            return self._solve_device(beta, c2, bnorm, ctol, iter_lim, multi)
        alfa = 0.0
        if beta > 0:
            K.lsqr_v_step(1.0 / beta, 0.0, self.v_t, self.acc[: 3 * K.C], self.acc[3 * K.C:])
            if multi:
                comm.allreduce(self.acc)
            K.lsqr_cam_v(self.acc[: 3 * K.C], 0.0, self.v_c, self.s2[2:3])
            alfa = float(np.sqrt(self._get(self.s2)[2] + self._get(self.acc)[3 * K.C]))
        self.info = dict(lsqr_iters=0, istop=0, converged=True)
        if alfa * beta == 0.0:
            return self.x_c, self.x_t
        # v1 = v/alfa, w1 = v1, x = 0
        K.lsqr_update(1.0 / alfa, 0.0, 0.0, self.v_c, self.w_c, self.x_c, self.s2[3:4])
        K.lsqr_update(1.0 / alfa, 0.0, 0.0, self.v_t, self.w_t, self.x_t, self.s2[1:2])
        rhobar, phibar = alfa, beta
        anorm = acond = ddnorm = xnorm = xxnorm = z = 0.0
        cs2, sn2 = -1.0, 0.0
        beta_prev = beta
        itn, istop = 0, 0
        while itn < iter_lim:
            itn += 1
            # bidiagonalisation: beta u = J~ v - alfa u ; alfa v = J~^T u - beta v
            K.lsqr_u_step(self.v_c, self.v_t, alfa / beta_prev, self.s2[0:1])
            if multi:
                comm.allreduce(self.s2[0:2])                   # |u|^2 and the pending |w_t|^2
            h = self._get(self.s2)
            beta = float(np.sqrt(h[0]))
            wnorm2 = float(h[1] + h[3])                        # |w_k|^2 of the CURRENT direction
            if beta > 0:
                anorm = np.sqrt(anorm * anorm + alfa * alfa + beta * beta)
                K.lsqr_v_step(1.0 / beta, beta, self.v_t, self.acc[: 3 * K.C], self.acc[3 * K.C:])
                if multi:
                    comm.allreduce(self.acc)
                K.lsqr_cam_v(self.acc[: 3 * K.C], beta, self.v_c, self.s2[2:3])
                alfa = float(np.sqrt(self._get(self.s2)[2] + self._get(self.acc)[3 * K.C]))
            # QR of the bidiagonal matrix, solution update coefficients
            cs, sn, rho = _sym_ortho(rhobar, beta)
            theta = sn * alfa
            rhobar = -cs * alfa
            phi = cs * phibar
            phibar = sn * phibar
            tau = sn * phi
            t1, t2 = phi / rho, -theta / rho
            ddnorm += wnorm2 / (rho * rho)
            inv_alfa = 1.0 / alfa if alfa > 0 else 1.0
            K.lsqr_update(inv_alfa, t1, t2, self.v_c, self.w_c, self.x_c, self.s2[3:4])
            K.lsqr_update(inv_alfa, t1, t2, self.v_t, self.w_t, self.x_t, self.s2[1:2])
            beta_prev = beta if beta > 0 else beta_prev
            # norm estimates and scipy's stopping tests (residuals corrected by c2)
            delta = sn2 * rho
            gambar = -cs2 * rho
            rhs = phi - delta * z
            zbar = rhs / gambar
            xnorm = np.sqrt(xxnorm + zbar * zbar)
            gamma = np.sqrt(gambar * gambar + theta * theta)
            cs2, sn2 = gambar / gamma, theta / gamma
            z = rhs / gamma
            xxnorm += z * z
            acond = anorm * np.sqrt(ddnorm)
            rnorm = np.sqrt(phibar * phibar + c2)
            arnorm = alfa * abs(tau)
            test1 = rnorm / bnorm
            test2 = arnorm / (anorm * rnorm + eps)
            test3 = 1.0 / (acond + eps)
            tt1 = test1 / (1 + anorm * xnorm / bnorm)
            rtol = self.btol + self.atol * anorm * xnorm / bnorm
            if itn >= iter_lim: istop = 7
            if 1 + test3 <= 1: istop = 6
            if 1 + test2 <= 1: istop = 5
            if 1 + tt1 <= 1: istop = 4
            if test3 <= ctol: istop = 3
            if test2 <= self.atol: istop = 2
            if test1 <= rtol: istop = 1
            if istop:
                break
        self.info = dict(lsqr_iters=itn, istop=istop, converged=True, rnorm=float(rnorm), arnorm=float(arnorm),
                         anorm=float(anorm), acond=float(acond), xnorm=float(xnorm))
        return self.x_c, self.x_t


LsqrTranslationSolver._solve_device = _lsqr_solve_device
