"""Camera tiling: graphs with more cameras than the LDS-resident sweeps hold (C > 1024).

``TiledGraph`` cuts the edge set by camera range into tiles that share one chunking of the timestep rows; ``TiledBackend`` runs
the solver's kernel interface tile by tile - the operator as ONE launch that reads every block once (vican_tiled_op) where the
grid is co-resident.  (Split out of device.py in round 6.)
"""
from __future__ import annotations

import ctypes as C
import math
import os

import numpy as np
import torch

from . import _lib
from ._rt import X_BOUND, _ptr, _stream, download, n_cu, upload
from .device import HipBackend, _LsqrCtx
from .layout import LocalGraph, _wave_params

# ---------------------------------------------------------------------------------------------------------------
# Camera tiling: graphs with more cameras than the LDS-resident sweeps hold (C > 1024)
# ---------------------------------------------------------------------------------------------------------------
TILE_CAMS = 1024


class TiledGraph:
    """The edge set cut by camera range into tiles of at most `tile` cameras, each a ``LocalGraph`` over ALL timestep rows
    (camera indices local to the tile) in the layout the planner picks for it - the wave layout wherever the tile's rows fit
    a 64-lane chunk (VICAN_TILE_LAYOUT=block|wave forces one); the translation CG runs tile by tile as well, so there is no
    limit on the number of cameras (the reference has none, bipgo.py:225-232).

    The reference has no camera limit (bipgo.py:225-232); the fused sweeps keep the camera tables in LDS, which caps
    them at 1024 cameras.  Beyond that the operator z = sum_t M_.t Lambda_t^-1 (sum_c M_ct^T x_c) is evaluated tile by
    tile (TiledBackend): every edge block is read twice per application instead of once."""

    def __init__(self, n_cam, row_ptr, col, blk, a, w=None, u=None, v=None, tile=None, deg_t=None, deg_c=None, permute_rows=False):
        import os
        lib = _lib.load()
        tile = int(tile or os.environ.get("VICAN_TILE_CAMS") or TILE_CAMS)
        dev = blk.device
        self.device, self.n_cam, self.n_time, self.n_edges = dev, int(n_cam), int(row_ptr.numel() - 1), int(col.numel())
        self.storage_dtype = blk.dtype
        self.layout = "tiled"
        have_t = w is not None
        T = self.n_time
        row_ptr = row_ptr.to(dev, torch.int64)
        col = col.to(dev, torch.int64)
        rows = torch.repeat_interleave(torch.arange(T, device=dev), row_ptr[1:] - row_ptr[:-1])
        # tiles of EQUAL width (4000 cameras: 4 x 1000, not 3 x 1024 + 928): a row's edges then split evenly over the tiles, and
        # the shared chunking fills every tile's slots at the same pace (with 1024-wide tiles the three full ones average 64 edges
        # per row - four rows = the 256 slots of a chunk exactly, so a fourth row fitted one chunk in two)
        n_tiles = max(1, -(-self.n_cam // tile))
        tile = min(tile, -(-(-(-self.n_cam // n_tiles)) // 8) * 8)
        self.bounds = list(range(0, self.n_cam, tile)) + [self.n_cam]
        self.tiles = []
        want = os.environ.get("VICAN_TILE_LAYOUT") or None
        storage = _lib.STORE_F32 if blk.dtype == torch.float32 else _lib.STORE_F64
        # row_perm[new] = old row (None: the rows in their own order), row_inv[old] = new: set by the packing below
        self.row_perm = self.row_inv = None

        def cut(rows_of_edges):
            out = []
            for k in range(len(self.bounds) - 1):
                c0, c1 = self.bounds[k], self.bounds[k + 1]
                sel = ((col >= c0) & (col < c1)).nonzero().squeeze(1)
                if self.row_perm is not None:                   # edges of the tile in the NEW row order (stable: cameras stay ascending)
                    sel = sel[torch.argsort(rows_of_edges[sel], stable=True)]
                rp = torch.zeros(T + 1, dtype=torch.int64, device=dev)
                rp[1:] = torch.cumsum(torch.bincount(rows_of_edges[sel], minlength=T), 0)
                out.append((c0, c1, sel, rp.to(torch.int32)))
            return out
        parts = cut(rows)
        rps_host = download([p_[3] for p_ in parts]) if T else [np.zeros(1, np.int32) for _ in parts]
        rps_host = [np.ascontiguousarray(r, dtype=np.int32) for r in rps_host]
        packed_chunks = None
        if permute_rows and want != "block" and os.environ.get("VICAN_TILE_SHARED", "1") != "0" and T > 0 and self.n_edges > 0 and len(parts) > 1:
            # Rows in a better ORDER for the shared chunking (vican_plan_rows_multi: consecutive rows pad it to 1.33 slots per edge on
            # the wide benchmark graph, packed ones to 1.03): everything per row inside this graph and its backend lives
            # in the new order; `unpermute_rows` / `permute_rows` translate at the boundary (bipgo.solve_problem)
            try:
                cap_rows, slots = 64, 0
                for (c0, c1, sel, _), rph in zip(parts, rps_host):
                    n_e = int(rph[-1])
                    if n_e == 0:
                        raise _lib.VicanError("a tile without edges")
                    slots, rows_t, n_copy_k, _ = _wave_params(lib, c1 - c0, max(1.0, n_e / T), n_e, storage)
                    while rows_t > 1 and int(lib.vican_tiled_op_lds_bytes(c1 - c0, rows_t, storage, n_copy_k)) > int(lib.vican_lds_limit_bytes()):
                        rows_t -= 1
                    cap_rows = min(cap_rows, rows_t)
                ptrs = (C.c_void_p * len(parts))(*[r.ctypes.data for r in rps_host])
                perm, c0s = np.empty(T, dtype=np.int32), np.empty(T + 2, dtype=np.int32)
                # the packing costs (rows x pool x tiles) comparisons on the host and pays where few rows fill a chunk (the
                # integer effect: 3 or 4 rows); chunks of many short rows fill well in any order: a smaller pool there, none
                # beyond 32 rows per chunk
                rows_est = max(1.0, slots / max(1.0, max(float(r[-1]) for r in rps_host) / T))
                window = 512 if rows_est <= 8 else 128 if rows_est <= 32 else 1
                nch = _lib.check(lib.vican_plan_rows_multi(T, len(parts), C.cast(ptrs, C.c_void_p), slots, cap_rows, window,
                                                            C.c_void_p(perm.ctypes.data), C.c_void_p(c0s.ctypes.data), T + 2),
                                 "vican_plan_rows_multi")
                packed_chunks = c0s[: nch + 1].copy()
                if not np.array_equal(perm, np.arange(T, dtype=np.int32)):     # (else: the rows stay where they are)
                    inv = np.empty(T, dtype=np.int64)
                    inv[perm] = np.arange(T)
                    self.row_perm = torch.from_numpy(perm.astype(np.int64)).to(dev)
                    self.row_inv = torch.from_numpy(inv).to(dev)
                    parts = cut(self.row_inv[rows])
                    rps_host = download([p_[3] for p_ in parts])
                    rps_host = [np.ascontiguousarray(r, dtype=np.int32) for r in rps_host]
                    if deg_t is not None:
                        deg_t = deg_t.to(dev)[self.row_perm]
            except _lib.VicanError:
                self.row_perm = self.row_inv = packed_chunks = None
                parts = cut(rows)
                rps_host = download([p_[3] for p_ in parts])
                rps_host = [np.ascontiguousarray(r, dtype=np.int32) for r in rps_host]
        # A chunking SHARED by all tiles (chunk k = the same timestep rows in every tile) lets the operator run as ONE launch
        # that reads every block once (vican_tiled_op, csrc/vican_tsweep.hip); it pads a little more than per-tile chunkings
        # (a row joins a chunk only while EVERY tile's edges still fit).  VICAN_TILE_SHARED=0: per-tile chunkings (two passes).
        self.shared_chunks = packed_chunks
        if packed_chunks is None and want != "block" and os.environ.get("VICAN_TILE_SHARED", "1") != "0" and T > 0 and self.n_edges > 0:
            try:
                cap_rows, slots = 64, 0
                for (c0, c1, sel, _), rph in zip(parts, rps_host):
                    n_e = int(rph[-1])
                    if n_e == 0:
                        raise _lib.VicanError("a tile without edges")
                    slots, rows_t, n_copy_k, _ = _wave_params(lib, c1 - c0, max(1.0, n_e / T), n_e, storage)
                    # (the fused launch runs 8 wavefronts per workgroup whatever the tile's own plan says: its LDS must fit too)
                    while rows_t > 1 and int(lib.vican_tiled_op_lds_bytes(c1 - c0, rows_t, storage, n_copy_k)) > int(lib.vican_lds_limit_bytes()):
                        rows_t -= 1
                    cap_rows = min(cap_rows, rows_t)
                ptrs = (C.c_void_p * len(parts))(*[r.ctypes.data for r in rps_host])
                out = np.empty(T + 2, dtype=np.int32)
                nch = _lib.check(lib.vican_plan_chunks_multi(T, len(parts), C.cast(ptrs, C.c_void_p), slots, cap_rows,
                                                             C.c_void_p(out.ctypes.data), T + 2), "vican_plan_chunks_multi")
                self.shared_chunks = out[: nch + 1].copy()
            except _lib.VicanError:
                self.shared_chunks = None
        def build(shared):
            tiles = []
            for (c0, c1, sel, rp), rph in zip(parts, rps_host):
                pick = lambda x: None if x is None else x[sel].contiguous()
                # wave layout wherever the tile's rows fit a 64-lane chunk, whatever it pads (measured on 4000 cameras x 250 per
                # timestep: 15 % padding, and still 52 + 53 us per tile for the rows and camera passes against 2 x 78 us for the
                # block layout's two-sided sweep)
                args = (c1 - c0, rp, (col[sel] - c0).to(torch.int32), blk[sel].contiguous(), a[sel].contiguous(), pick(w), pick(u), pick(v))
                if shared is not None:
                    t = LocalGraph(*args, layout="wave", row_ptr_host=rph, forced_chunks=shared)
                else:
                    try:
                        t = LocalGraph(*args, layout=want or "wave", row_ptr_host=rph)
                    except _lib.VicanError:
                        if want == "wave":
                            raise
                        t = LocalGraph(*args, layout="block", row_ptr_host=rph)
                tiles.append(t)
            return tiles
        try:
            self.tiles = build(self.shared_chunks)
        except _lib.VicanError:
            if self.shared_chunks is None:
                raise
            self.shared_chunks = None
            self.tiles = build(None)
        # global graph constants
        self.row_sum_a = torch.stack([t.row_sum_a for t in self.tiles]).sum(0)
        self.rnorm = torch.stack([t.rnorm for t in self.tiles]).sum(0)
        self.cam_sum_a = torch.cat([t.cam_sum_a for t in self.tiles])
        if have_t:
            self.row_sum_w = torch.stack([t.row_sum_w for t in self.tiles]).sum(0)
            self.cam_sum_w = torch.cat([t.cam_sum_w for t in self.tiles])
            if deg_t is not None:                           # the caller's diagonal (LocalGraph)
                self.row_sum_w[: T].copy_(deg_t.to(dev, torch.float64))
            if deg_c is not None:
                self.cam_sum_w.copy_(deg_c.to(dev, torch.float64))
            self.wmax = max(t.wmax for t in self.tiles)
            # (the CG product runs tile by tile on the tiles' own weight arrays - TiledBackend.cg_iter_local: no camera limit)

    def unpermute_rows(self, x):
        """A per-row array of this graph (first dimension = rows in the graph's own order) in the CALLER's row order."""
        return x if self.row_inv is None else x[self.row_inv]

    def permute_rows(self, x):
        """A per-row array in the caller's row order -> the graph's own order."""
        return x if self.row_perm is None else x[self.row_perm]

    def op_bytes(self, ncols=3):
        return sum(t.op_bytes(ncols) for t in self.tiles) * 2

    def padded_slots(self):
        return sum(t.padded_slots() for t in self.tiles)


class TiledBackend(HipBackend):
    """``HipBackend`` interface on a ``TiledGraph``: the edge sweeps run tile by tile - a rows pass (y_t = the tile's share
    of sum_c M_ct^T x_c) and a camera pass (z_c = sum_t M_ct w_t for the tile's cameras) per tile: wave-layout tiles through
    vican_tile_rows / vican_tile_cams (sweep MODEs 1 and 4 of the wave kernel), block-layout tiles through the one-pass
    bipartite operator (sweep MODE 2 with a zero operand on the unused side); the
    per-row partials of the tiles are summed in tile order by ``vican_sum_apply3``; everything camera-sided (Lanczos
    step, Ritz, gauge, polar) is the launch-sequence path of the untiled backend, which has no camera limit; the CG
    product and the LSQR steps run tile by tile too (vican_cg_sweep_partial + vican_cg_combine_rows; vican_lsqr_step per tile):
    no limit on the number of cameras.  No fused dual update (a performance feature of the untiled sweeps)."""
    fused_dual_ok = False
    cg_iter_fused = None           # (the tiled CG product is several launches: no fused iteration)
    cg_iter_comm = None            # ... nor the sharded iteration behind one host call
    cg1_iter_local = None          # sharded tiled solves keep the two-message CG (the one-message product is an untiled sweep)

    def __init__(self, graph: TiledGraph):
        self.lib, self.g, self.dev = _lib.load(), graph, graph.device
        self.C, self.T = graph.n_cam, graph.n_time
        self.storage_f64 = graph.storage_dtype == torch.float64
        self.tiles = [HipBackend(t) for t in graph.tiles]
        # (the camera-side Lanczos step as one cooperative launch: 125 workgroups for 4000 cameras - against lap_apply + 3 Gram
        #  products + 2 updates + the QR, 7 launches and ~75 us per step on the wide benchmark; the slabs of a tiled sweep are
        #  folded by vican_tiled_op_z, so never `from_slabs`)
        self.fold_in_step_ok, self.layout = False, "tiled"
        self.coop_cam_step = os.environ.get("VICAN_COOP", "1") != "0" and self.C <= 8192
        self._status_host, self._coop_ws, self._coop_sync, self._gram_ws = {}, None, None, None
        self.coop_failures = []
        self.rr_part = torch.zeros(1536, dtype=torch.float64, device=self.dev)
        self.ws = torch.zeros(1024, dtype=torch.float64, device=self.dev)
        T1, nt = max(self.T, 1), len(self.tiles)
        self.ypart = torch.zeros(nt, T1, 9, dtype=torch.float64, device=self.dev)      # per-tile row partials
        self.wrow = torch.zeros(T1, 9, dtype=torch.float64, device=self.dev)           # phase-3 operand of the second pass
        self.zero_rows = torch.zeros(T1, 9, dtype=torch.float64, device=self.dev)
        self.scratch_c = [torch.zeros(3 * (b1 - b0), 3, dtype=torch.float64, device=self.dev) for b0, b1 in zip(graph.bounds[:-1], graph.bounds[1:])]
        self._fused = None
        if getattr(graph, "shared_chunks", None) is not None:
            self._setup_fused()
        if graph.tiles[0].w is not None:
            self.n_add_cg = float(max(max(t.tl.rows_per_wg_max, t.tl.slots) for t in graph.tiles) + 1)
            self._cg_wmax = graph.wmax
            self._cg_w = [t.w for t in graph.tiles]                                     # per tile, in the tile's slot order
            self._w_scaled = None
            self.pq_part = torch.empty(1024, dtype=torch.float64, device=self.dev)
            self.acc_t = torch.zeros(nt, T1, 3, dtype=torch.float64, device=self.dev)   # per-tile row sums of the CG product
            self.rhs_part = torch.zeros(nt, T1, 3, dtype=torch.float64, device=self.dev)
            # the tiles' CG products in ONE launch (vican_cg_sweep_tiles: 2..4 wave-layout tiles of one launch shape; the
            # launcher refuses anything else and the per-tile launches take over)
            self._tcg = None
            if 2 <= nt <= 4 and all(K.layout == "wave" for K in self.tiles):
                t = _LsqrCtx()
                t.nwgt = max(1, n_cu() // nt)
                t.host = (_lib.CgTile * nt)()
                t.parts = [torch.empty(t.nwgt * 6 * K.C, dtype=torch.float64, device=self.dev) for K in self.tiles]
                for k, K in enumerate(self.tiles):
                    t.host[k].g = K.g.desc
                    t.host[k].acc_t, t.host[k].qc_part = self.acc_t[k].data_ptr(), t.parts[k].data_ptr()
                    self.n_add_cg = max(self.n_add_cg, float(-(-K.g.n_chunk // t.nwgt) * K.g.max_rows + 1))
                self._tcg = t

    def _tile_rows(self, k):
        b = self.g.bounds
        return 3 * b[k], 3 * b[k + 1]

    # -- the operator as ONE launch that reads every block once (vican_tiled_op; tiles with a shared chunking) ----------
    def _setup_fused(self):
        nt, T1 = len(self.tiles), max(self.T, 1)
        nwgt = n_cu() // nt
        if nwgt < 1 or nt > 64:
            return
        tl = self.g.tiles
        if int(self.lib.vican_tiled_op_lds_bytes(max(t.n_cam for t in tl), max(t.max_rows for t in tl), tl[0].desc.storage,
                                                 max(t.n_copy for t in tl))) > int(self.lib.vican_lds_limit_bytes()):
            return
        f = _LsqrCtx()
        f.nwgt, f.parity = nwgt, 0
        # adds into one camera accumulator by one workgroup of the fused launch = rows it handles (<= its chunks x rows per chunk):
        # the tiles' fixed-point scales are finished for at least that many
        # (tiled_sweep_kernel hands chunks out per WAVEFRONT with stride nwgt * 8: a workgroup takes up to 8 * ceil(n / (8 nwgt)))
        n_chunk = self.g.tiles[0].n_chunk
        for t in self.g.tiles:
            t.rows_per_wg_sweep = max(t.rows_per_wg_sweep, min(self.T, 8 * -(-n_chunk // (8 * nwgt)) * t.max_rows))
        f.x = torch.zeros(3 * self.C, 3, dtype=torch.float64, device=self.dev)           # the operand, at a fixed address
        f.yp = torch.empty(2, nt, T1, 9, dtype=torch.float64, device=self.dev)           # share buffers of alternate launches
        self._ck(self.lib.vican_tiled_op_sentinel(_ptr(f.yp), f.yp.numel(), _stream()), "vican_tiled_op_sentinel")
        f.host = (_lib.Tile * nt)()
        b = self.g.bounds
        # (slabs of the fused launch: n_wg_tile per tile - a tile's own zpart is sized for ITS plan's workgroups)
        f.zpart = [torch.empty(nwgt * 9 * K.C, dtype=torch.float64, device=self.dev) for K in self.tiles]
        for k, K in enumerate(self.tiles):
            e = f.host[k]
            e.g = K.g.desc
            e.x = f.x.data_ptr() + 8 * 9 * b[k]
            e.zpart, e.fx = f.zpart[k].data_ptr(), K.g.fx.data_ptr()
            e.ypart[0], e.ypart[1] = f.yp[0, k].data_ptr(), f.yp[1, k].data_ptr()
        raw = np.frombuffer(bytes(f.host), dtype=np.uint8).copy()
        f.dev = upload(self.dev, [(raw, torch.uint8)])[0]
        self._fused = f

    def _fused_op(self, lamT_inv, x, z_out):
        """False: the fused launch is not available (grid not co-resident) - the caller takes the two-pass path."""
        f = self._fused
        if torch.cuda.is_current_stream_capturing():
            return False                # (the share buffer's parity is a launch ARGUMENT: a replayed graph would reuse one buffer)
        # operand and result in the caller's arrays, the tiles' slab folds in one launch (vican_tiled_op_z; round 4 copied x to a
        # fixed address and folded tile by tile: five launches more per application)
        if not x.is_contiguous():
            f.x.copy_(x)
            x = f.x
        z = z_out if z_out.is_contiguous() else torch.empty_like(f.x)
        rc = self.lib.vican_tiled_op_z(C.cast(f.host, C.c_void_p), _ptr(f.dev), len(self.tiles), f.nwgt, _ptr(lamT_inv), _ptr(x), _ptr(z),
                                       f.parity, _stream())
        if rc == _lib.ERR_CAPACITY:
            self._fused = None
            self.coop_failures.append("vican_tiled_op_z: " + self.lib.vican_last_error().decode())
            return False
        self._ck(rc, "vican_tiled_op_z")
        f.parity ^= 1
        if z is not z_out:
            z_out.copy_(z)
        return True

    def cooperative_failed(self, which):
        """As HipBackend.cooperative_failed; the fused tiled operator spins on other workgroups too and is dropped with the rest."""
        self._fused = None
        super().cooperative_failed(which)

    def _sum_apply(self, A, B, n_b, out, width=9):
        self._ck(self.lib.vican_sum_apply3(self.T, width, _ptr(A), _ptr(B), n_b, B.stride(0), _ptr(out), _stream()), "vican_sum_apply3")

    def _refresh_scales(self, lamT_inv):
        """omega = max_t |Lambda_t^-1|_F * rnorm[t] with the row norms of ALL tiles, into every tile's scale buffer."""
        # (the bound once, into the first tile's buffer; one launch finishes all tiles' scales: 3 launches instead of 3 per tile)
        tl = self.tiles
        self._ck(self.lib.vican_duals_bound(self.T, _ptr(lamT_inv), _ptr(self.g.rnorm), _ptr(tl[0].g.fx), _stream()), "vican_duals_bound")
        fxs = (C.c_void_p * len(tl))(*[K.g.fx.data_ptr() for K in tl])
        nadd = (C.c_double * len(tl))(*[float(K.g.rows_per_wg_sweep + 1) for K in tl])
        self._ck(self.lib.vican_fx_finish_multi(C.cast(fxs, C.c_void_p), C.cast(nadd, C.c_void_p), len(tl), X_BOUND, tl[0].g.desc.storage, _stream()),
                 "vican_fx_finish_multi")

    def _rows_T(self, x):
        """ypart[k] = sum_{c in tile k} M_ct^T x_c for every tile (first pass of the one-pass operator; its camera-side
        half runs on a zero operand and is discarded)."""
        for k, K in enumerate(self.tiles):
            r0, r1 = self._tile_rows(k)
            if K.layout == "wave":
                self._ck(self.lib.vican_tile_rows(K._gref, _ptr(x[r0:r1]), _ptr(self.ypart[k]), _ptr(K.g.fx), _stream()), "vican_tile_rows")
            else:
                self._ck(self.lib.vican_bip_apply(K._gref, _ptr(x[r0:r1]), _ptr(self.zero_rows), _ptr(K.zpart), _ptr(K.g.fx),
                                                  _ptr(self.scratch_c[k]), _ptr(self.ypart[k]), _stream()), "vican_bip_apply")

    # -- rotation stage ---------------------------------------------------------------------------------------
    def init_duals(self, lamT_inv, cam_deg):
        cam_deg.copy_(self.g.cam_sum_a)
        K0 = self.tiles[0]
        self._ck(self.lib.vican_init_duals(self.T, _ptr(self.g.row_sum_a), _ptr(self.g.rnorm), _ptr(lamT_inv), _ptr(K0.g.fx), _stream()),
                 "vican_init_duals")
        self._refresh_scales(lamT_inv)

    def set_duals(self, lamT_inv):
        self._refresh_scales(lamT_inv)

    def block_op(self, lamT_inv, x, z_out):
        """z_out = P x: one fused launch where the tiles share their chunking (every block read once); else a rows pass over all
        tiles, w_t = Lambda_t^-1 (sum of the tiles' row partials) and a camera pass per tile."""
        if self._fused is not None and self._fused_op(lamT_inv, x, z_out):
            return
        self._rows_T(x)
        self._sum_apply(lamT_inv, self.ypart, len(self.tiles), self.wrow)
        for k, K in enumerate(self.tiles):
            r0, r1 = self._tile_rows(k)
            if K.layout == "wave":
                self._ck(self.lib.vican_tile_cams(K._gref, _ptr(self.wrow), _ptr(K.zpart), _ptr(K.g.fx), _ptr(z_out[r0:r1]), _stream()),
                         "vican_tile_cams")
            else:
                self._ck(self.lib.vican_bip_apply(K._gref, _ptr(x[r0:r1]), _ptr(self.wrow), _ptr(K.zpart), _ptr(K.g.fx), _ptr(z_out[r0:r1]),
                                                  _ptr(self.ypart[k]), _stream()), "vican_bip_apply")

    def dual_update(self, rc, Rt, lamT_inv):
        """Z_t = sum_c M_ct^T R_c over all tiles, then R_t, Lambda_t^-1 = U S^-1 U^T per row (bipgo.py:318-332)."""
        self._rows_T(rc)
        self._sum_apply(None, self.ypart, len(self.tiles), self.wrow)
        self.polar_dual(self.wrow, Rt, lamT_inv, 2)
        self._refresh_scales(lamT_inv)

    # -- translation stage ------------------------------------------------------------------------------------
    def trans_degrees(self, deg_t, deg_c):
        deg_t[: self.g.row_sum_w.numel()].copy_(self.g.row_sum_w)
        deg_c.copy_(self.g.cam_sum_w)

    def trans_rhs(self, rc, rt, rhs_t, rhs_c):
        b = self.g.bounds
        for k, K in enumerate(self.tiles):
            K.trans_rhs(rc[3 * b[k]: 3 * b[k + 1]], rt, self.rhs_part[k], rhs_c[b[k]: b[k + 1]])
        self._ck(self.lib.vican_sum_apply3(self.T, 3, None, _ptr(self.rhs_part), len(self.tiles), self.rhs_part.stride(0), _ptr(rhs_t), _stream()),
                 "vican_sum_apply3")

    # The CG product q = A p one camera tile at a time (vican_cg_sweep_partial): a tile's sweep yields its row sums
    # sum_{c in tile} w p_c and the complete camera sums of its own cameras; the rows are combined in tile order.
    cg_resident_ok = False

    def cg_iter_local(self, deg_t, r_c, p_c, r_t, p_t, q_t, qcpq, rtol, st, n_rr_part):
        b = self.g.bounds
        self._ck(self.lib.vican_cg_begin(self.C, _ptr(r_c), _ptr(p_c), float(rtol), _ptr(self.rr_part), int(n_rr_part), self.n_add_cg,
                                         _ptr(st), _stream()), "vican_cg_begin")
        self._ck(self.lib.vican_cg_update_pt(self.T, _ptr(r_t), _ptr(p_t), _ptr(st), _stream()), "vican_cg_update_pt")
        t = self._tcg
        if t is not None:
            for k in range(len(self.tiles)):
                t.host[k].w, t.host[k].p_c = self._cg_w[k].data_ptr(), p_c.data_ptr() + 8 * 3 * b[k]
            rc = self.lib.vican_cg_sweep_tiles(C.cast(t.host, C.c_void_p), len(self.tiles), t.nwgt, _ptr(p_t), _ptr(st), _stream())
            if rc == _lib.ERR_CAPACITY:
                self._tcg = t = None                         # (launch shapes differ / small graphs: per-tile launches)
            else:
                self._ck(rc, "vican_cg_sweep_tiles")
                nt = len(self.tiles)
                parts = (C.c_void_p * nt)(*[t.parts[k].data_ptr() for k in range(nt)])
                ncams = (C.c_int32 * nt)(*[K.C for K in self.tiles])
                self._ck(self.lib.vican_cg_fold_tiles(C.cast(parts, C.c_void_p), C.cast(ncams, C.c_void_p), nt, t.nwgt, _ptr(qcpq), _ptr(st),
                                                      _stream()), "vican_cg_fold_tiles")        # (one launch for all tiles)
        for k, K in enumerate(self.tiles if t is None else ()):
            part = K.zpart[: K.tl.n_wg * 6 * K.C]
            self._ck(self.lib.vican_cg_sweep_partial(K._gref_t, _ptr(self._cg_w[k]), _ptr(p_c[b[k]: b[k + 1]]), _ptr(p_t), _ptr(self.acc_t[k]),
                                                     _ptr(part), _ptr(st), _stream()), "vican_cg_sweep_partial")
            self._ck(self.lib.vican_cg_fold(_ptr(part), K.tl.n_wg, K.C, None, C.c_void_p(qcpq.data_ptr() + 8 * 3 * b[k]), _ptr(st), _stream()),
                     "vican_cg_fold")
        nb = self._ck(self.lib.vican_cg_combine_rows(self.T, len(self.tiles), self.acc_t.stride(0), _ptr(deg_t), _ptr(p_t), _ptr(self.acc_t),
                                                     _ptr(q_t), _ptr(self.pq_part), self.pq_part.numel(), _ptr(st), _stream()),
                      "vican_cg_combine_rows")
        self._ck(self.lib.vican_cg_reduce_pq(_ptr(self.pq_part), nb, C.c_void_p(qcpq.data_ptr() + 8 * 3 * self.C), _ptr(st), _stream()),
                 "vican_cg_reduce_pq")

    def set_cg_scaling(self, s_c, s_t):
        """CG sweeps use w~ = w s_c s_t (<= 1) until clear_cg_scaling() - per tile."""
        b = self.g.bounds
        if self._w_scaled is None:
            self._w_scaled = [torch.empty_like(t.w) for t in self.g.tiles]
        for k, K in enumerate(self.tiles):
            self._ck(self.lib.vican_scale_weights(K._gref_t, _ptr(self.g.tiles[k].w), _ptr(s_c[b[k]: b[k + 1]]), _ptr(s_t), _ptr(self._w_scaled[k]),
                                                  _stream()), "vican_scale_weights")
        self._cg_w, self._cg_wmax = self._w_scaled, 1.0

    def clear_cg_scaling(self):
        self._cg_w, self._cg_wmax = [t.w for t in self.g.tiles], self.g.wmax

    # LSQR (lsqr_solver="direct") tile by tile: every tile keeps its own edge vector u~; row sums of the tiles are added in
    # tile order, camera sums are complete per tile, |u^|^2 is the sum of the tiles' parts
    def _lsqr_tiles_alloc(self):
        if not hasattr(self, "_ls_tmp"):
            nt, T1 = len(self.tiles), max(self.T, 1)
            self._ls_tmp = torch.zeros(nt + 1, dtype=torch.float64, device=self.dev)
            self._ls_rows = torch.zeros(nt, T1, 3, dtype=torch.float64, device=self.dev)
            self._lsqr_part = torch.zeros(1024, dtype=torch.float64, device=self.dev)

    def lsqr_init_u(self, rc, rt, nrm2_out):
        self._lsqr_tiles_alloc()
        b = self.g.bounds
        for k, K in enumerate(self.tiles):
            K.lsqr_init_u(rc[3 * b[k]: 3 * b[k + 1]], rt, self._ls_tmp[k: k + 1])
        nrm2_out.copy_(self._ls_tmp[: len(self.tiles)].sum().reshape(1))

    def lsqr_v_step(self, inv_beta, beta, v_t, acc_c, nrm2_t_out):
        b = self.g.bounds
        for k, K in enumerate(self.tiles):
            self._ls_rows[k].zero_()
            K.lsqr_v_step(inv_beta, 0.0, self._ls_rows[k], acc_c[3 * b[k]: 3 * b[k + 1]], self._ls_tmp[k: k + 1])
        v_t.copy_(self._ls_rows.sum(0) - beta * v_t)
        nrm2_t_out.copy_((v_t[: self.T] ** 2).sum().reshape(1))

    def lsqr_device_params(self):
        self._lsqr_tiles_alloc()
        n_add = max(K.lsqr_device_params()[1] for K in self.tiles)
        return math.sqrt(self.g.wmax), n_add

    def lsqr_step(self, v_c, v_t, z_t, acc, st):
        b, C3 = self.g.bounds, 3 * self.C
        nt = len(self.tiles)
        for k, K in enumerate(self.tiles):
            # the tile writes its camera sums into its slice of acc and its part of |u^|^2 right behind the slice
            K.lsqr_step(v_c[b[k]: b[k + 1]], v_t, self._ls_rows[k], acc[3 * b[k]:], st)
            self._ls_tmp[k: k + 1].copy_(acc[3 * b[k + 1]: 3 * b[k + 1] + 1])
        acc[C3: C3 + 1].copy_(self._ls_tmp[:nt].sum().reshape(1))
        self._ck(self.lib.vican_sum_apply3(self.T, 3, None, _ptr(self._ls_rows), nt, self._ls_rows.stride(0), _ptr(z_t), _stream()),
                 "vican_sum_apply3")

    def _unsupported(self, *a, **k):
        raise _lib.VicanError("not available on camera-tiled graphs (more than %d cameras)" % TILE_CAMS)

    dual_update_op = block_op_raw = fold_z = bip_apply = bip_scales = node_degrees = cg_sweep = cg_resident = _unsupported
    lsqr_u_step = _unsupported
