"""Graph planning: the chunked edge layout of one rank's timestep rows.

``LocalGraph`` owns (as PyTorch-ROCm tensors) the timestep-major chunked CSR-of-3x3-blocks of ONE rank's timestep rows plus
the graph constants computed once at pack time (row / camera weight sums, block-norm bounds) - reference bipgo.py:244-276.
``_Layout`` sizes a layout (wave: one wavefront per chunk; block: one workgroup per chunk) for the LDS of a compute unit.
(Split out of device.py in round 6.)  No CPU fallback: without a GPU construction raises.
"""
from __future__ import annotations

import ctypes as C
import math
import os

import numpy as np
import torch

from . import _lib
from ._rt import STREAM_NT_BYTES, _ptr, _stream, n_cu

def _wave_params(lib, n_cam, deg_avg, n_edges, storage, n_copy=None, wg_waves=None):
    """(slots, rows_target, n_copy, wg_waves) of a wave layout: what the LDS of a compute unit allows for this graph."""
    epl = 4 if storage == _lib.STORE_F32 else 2
    lim = int(lib.vican_lds_limit_bytes())
    # one wavefront per chunk (vican_wsweep.hip): 64 lanes x EPL slots, whole rows, <= 64 rows per chunk
    slots = 64 * epl
    # (<= 64 rows per chunk; 63 with 1024 cameras: camera 1023 of row 63 would read as the padding word of the 2-byte index)
    rows_target = max(1, min(63 if n_cam >= 1024 else 64, int(math.ceil(1.25 * slots / deg_avg)) + 1))
    if n_copy is None:        # lanes of a wavefront that share a row = deg / EPL
        n_copy = 1        # (measured on the stress graph, 62 lanes per row: 8 copies 187 us, 16 copies 195 us - the fold grows)
        while n_copy < 8 and n_copy * epl < deg_avg:
            n_copy *= 2
    if wg_waves is None:
        wg_waves = 12
        if n_edges < 12 * slots * n_cu():          # small graphs: fewer wavefronts per workgroup, more workgroups
            wg_waves = 8 if n_edges >= 8 * slots * n_cu() else 4
    fits = lambda rows, nc, nw: int(lib.vican_wsweep_lds_bytes(n_cam, rows, storage, nc, nw)) <= lim
    while not fits(rows_target, n_copy, wg_waves) and n_copy > 1:
        n_copy //= 2
    if not fits(rows_target, n_copy, wg_waves):
        # before giving up wavefronts (occupancy): a row limit without the 25 % margin, if the chunks still fill their
        # slots with it (ragged rows of 2-8 edges: 62 rows instead of 64 keep 12 wavefronts resident instead of 8;
        # worth 1.5 % there - that sweep is bound by the LDS work per row, not by occupancy: tools/ragged_time.py)
        r = rows_target
        while r > 1 and not fits(r, n_copy, wg_waves):
            r -= 1
        if fits(r, n_copy, wg_waves) and r * deg_avg >= 1.05 * slots:
            rows_target = r
    while not fits(rows_target, n_copy, wg_waves) and wg_waves > 4:
        wg_waves -= 4
    while not fits(rows_target, n_copy, wg_waves) and rows_target > 1:
        rows_target -= 1
    if not fits(rows_target, n_copy, wg_waves):
        raise _lib.VicanError("camera tables (C=%d) do not fit in LDS" % n_cam)
    return slots, rows_target, n_copy, wg_waves


class _Layout:
    """One chunked edge layout of a rank's rows (device arrays + the vican_graph_t view of them)."""

    def __init__(self, lib, kind, n_cam, n_time, rp_host, deg_max, deg_avg, n_edges, storage, dev, block_threads=None, n_wg=None,
                 n_copy=None, wg_waves=None, forced_chunks=None):
        epl = 4 if storage == _lib.STORE_F32 else 2
        self.kind, self.n_time = kind, n_time
        lim = int(lib.vican_lds_limit_bytes())
        if kind == "wave":
            slots, rows_target, n_copy, wg_waves = _wave_params(lib, n_cam, deg_avg, n_edges, storage, n_copy, wg_waves)
            max_rows, block_threads = rows_target, 64 * wg_waves
        else:
            wg_waves = 0
            if block_threads is None:
                # 768 threads (12 wavefronts, <= 168 VGPRs) holds two register sets of a chunk without
                # spilling and measured fastest on the HBM-bound stress graph; small graphs use 256 so
                # that there are enough chunks to occupy the chip
                block_threads = 768 if n_edges >= 768 * epl * n_cu() else 256
                if deg_max > 256 * epl:
                    block_threads = 768
                if deg_max > 768 * epl:
                    block_threads = 1024
            slots = block_threads * epl
            # lane-striped copies of the row accumulators: as many as LDS allows while a chunk can
            # still hold its natural number of rows (slots / average degree)
            rows_target = min(65535, int(math.ceil(1.25 * slots / deg_avg)) + 1)
            if n_copy is None:
                n_copy = 8        # measured: 4..32 copies are within 3 % on the stress graph; 8 leaves LDS for rows
                while n_copy > 1 and lib.vican_max_rows_for(n_cam, storage, n_copy) < rows_target:
                    n_copy //= 2
            max_rows = int(lib.vican_max_rows_for(n_cam, storage, n_copy))
            if max_rows < 1:
                raise _lib.VicanError("camera tables (C=%d) do not fit in LDS" % n_cam)
            max_rows = min(max_rows, max(rows_target, 1))
        cap = n_time + 2
        if forced_chunks is not None:
            # a chunking shared with other graphs over the same rows (camera tiles, vican_tiled_op): taken as given if it fits
            c0 = np.ascontiguousarray(forced_chunks, dtype=np.int32)
            nchunk = len(c0) - 1
            rp_np = rp_host.numpy()
            if nchunk < 1 or c0[0] != 0 or c0[-1] != n_time or (np.diff(c0) < 1).any() or int(np.diff(c0).max()) > max_rows \
                    or int((rp_np[c0[1:]] - rp_np[c0[:-1]]).max()) > slots:
                raise _lib.VicanError("the forced chunking does not fit this layout (rows per chunk <= %d, edges <= %d)" % (max_rows, slots))
        else:
            c0 = np.empty(cap, dtype=np.int32)
            nchunk = _lib.check(lib.vican_plan_chunks(n_time, C.c_void_p(rp_host.data_ptr()), slots, max_rows,
                                                       C.c_void_p(c0.ctypes.data), cap), "vican_plan_chunks")
        self.chunk_row0_host = c0[: nchunk + 1].copy()
        rows_per_chunk = np.diff(self.chunk_row0_host) if nchunk else np.zeros(0, np.int32)
        self.max_rows = int(rows_per_chunk.max()) if nchunk else 1
        self.n_chunk, self.slots, self.block_threads, self.n_copy, self.wg_waves = int(nchunk), slots, block_threads, int(n_copy), wg_waves
        if kind == "wave":
            lds = int(lib.vican_wsweep_lds_bytes(n_cam, self.max_rows, storage, n_copy, wg_waves))
            per_wg = wg_waves
        else:
            lds = int(lib.vican_sweep_lds_bytes(n_cam, self.max_rows, storage, n_copy))
            per_wg = 1
        occ = max(1, min(lim // lds, 2048 // block_threads))
        if n_wg is None:
            n_wg = max(1, min(-(-self.n_chunk // per_wg), n_cu() * occ))
        self.n_wg = int(n_wg)
        # max timestep rows one workgroup handles (bounds the adds into one z accumulator)
        bounds = (np.arange(self.n_wg + 1, dtype=np.int64) * self.n_chunk) // self.n_wg
        self.rows_per_wg_max = int(np.diff(self.chunk_row0_host[bounds]).max()) if nchunk else 1
        if kind == "wave":
            # ranges of NW / 2 NW chunks are handed to the workgroups by a device counter (vican_wsweep.hip); a workgroup
            # takes at most `cap` chunks, which bounds the adds into one of its z accumulators
            per = -(-self.n_chunk // self.n_wg) if nchunk else 1
            self.wg_chunk_cap = (-(-13 * per // (10 * wg_waves)) + 3) * wg_waves
            self.rows_per_wg_sweep = max(min(n_time, self.wg_chunk_cap * self.max_rows), 1)
        else:
            # the block sweeps hand chunks out dynamically (tickets); a workgroup takes at most `cap` of them, which
            # bounds the adds into one of its z accumulators
            per = -(-self.n_chunk // self.n_wg) if nchunk else 1
            self.wg_chunk_cap = per + max(2, -(-per // 8))
            self.rows_per_wg_sweep = max(self.rows_per_wg_max, min(n_time, self.wg_chunk_cap * self.max_rows), 1)
        self.chunk_row0 = torch.from_numpy(self.chunk_row0_host).to(dev)
        # order of the edges inside a chunk (include/vican_hip.h: vican_graph_t.slot_order)
        # bank-aware (conflict-free camera-side LDS accesses; needs >= 32 * epl edges per row before a lane holds a whole
        # run of one row) or row-major (few row flushes).  Measured on MI355X (operator sweep, f32, ps per edge): rows of 8 / 16 /
        # 32 / 64 / 128 / 250 edges - bank-aware 9.9 / 11.5 / 10.5 / 8.1 / 8.2 / 6.8, row-major 8.4 / 8.4 / 8.3 / 7.5 / 7.6 / 7.2
        import os
        so = os.environ.get("VICAN_SLOT_ORDER")
        self.slot_order = {"banks": 0, "rows": 1}[so] if so else int(deg_avg < 48 * epl)
        self.nslot = max(1, self.n_chunk) * slots
        self.idx = torch.empty(self.nslot, dtype=torch.int32, device=dev)

    def describe(self, n_cam, storage, blk):
        # an edge stream that cannot stay in the 256 MB Infinity Cache between two sweeps is read with non-temporal loads
        stream_bytes = self.nslot * (9 * (4 if storage == _lib.STORE_F32 else 8) + 4)
        if blk is None:                                      # translation layout: index + weight words of the CG sweep
            stream_bytes = self.nslot * 12
        self.stream_nt = int(stream_bytes > STREAM_NT_BYTES)
        self.desc = _lib.Graph(n_cam, self.n_time, self.n_chunk, self.slots, self.max_rows, storage, self.block_threads,
                               self.n_wg, self.n_copy, self.wg_chunk_cap, _lib.LAYOUT_WAVE if self.kind == "wave" else _lib.LAYOUT_BLOCK,
                               self.wg_waves, self.stream_nt, self.slot_order, None if blk is None else blk.data_ptr(), self.idx.data_ptr(),
                               self.chunk_row0.data_ptr())
        return self.desc


class LocalGraph:
    """Chunked layout(s) of this rank's timestep rows.

    Parameters are device tensors in timestep-major CSR order:
    row_ptr (T+1,) int32, col (E,) int32 (ascending camera index inside a row),
    blk (E,9) / a (E,) in the storage dtype (float32 or float64), and optionally
    the translation-stage arrays w (E,), u (E,3), v (E,3) in float64.

    layout: "wave" = one wavefront per chunk of <= 256 (f32) / 128 (f64) slots (vican_wsweep.hip; rows must fit a
    chunk), "block" = one workgroup per chunk of 1024..4096 slots (vican_sweep.hip), None = wave where the rows allow
    it without more padding than the block layout needs (VICAN_LAYOUT overrides).  The translation arrays always live
    in a block layout (`desc_t`; the same object as `desc` when the rotation layout is a block layout).
    """

    def __init__(self, n_cam, row_ptr, col, blk, a, w=None, u=None, v=None, block_threads=None,
                 n_wg=None, n_copy=None, layout=None, wg_waves=None, deg_t=None, deg_c=None, row_ptr_host=None, keep_csr=None,
                 forced_chunks=None):
        import os
        lib = _lib.load()
        if not torch.cuda.is_available():
            raise _lib.VicanError("vican_amd needs a GPU (MI355X); there is no CPU fallback")
        dev = blk.device
        self.device = dev
        self.n_cam = int(n_cam)
        self.n_time = int(row_ptr.numel() - 1)
        self.n_edges = int(col.numel())
        self.storage_dtype = blk.dtype
        storage = _lib.STORE_F32 if blk.dtype == torch.float32 else _lib.STORE_F64
        epl = 4 if storage == _lib.STORE_F32 else 2
        if self.n_cam > 65535:
            raise _lib.VicanError("more than 65535 cameras are not supported by the packed edge index")
        # (row_ptr_host: the caller's host copy of row_ptr, if it has one - saves a device->host read)
        rp_host = (torch.from_numpy(np.ascontiguousarray(row_ptr_host, dtype=np.int32)) if row_ptr_host is not None
                   else row_ptr.to("cpu", torch.int32).contiguous())
        deg = (rp_host[1:] - rp_host[:-1]) if self.n_time else torch.zeros(1, dtype=torch.int32)
        deg_max, deg_avg = int(deg.max()), max(1.0, float(deg.float().mean()))
        layout = layout or os.environ.get("VICAN_LAYOUT") or None
        if layout not in (None, "wave", "block"):
            raise ValueError("layout must be 'wave', 'block' or None")
        mk = lambda kind, **kw: _Layout(lib, kind, self.n_cam, self.n_time, rp_host, deg_max, deg_avg, self.n_edges, storage, dev, **kw)
        rot = None
        if layout != "block" and block_threads is None and deg_max <= 64 * epl and self.n_cam <= 1024 and self.n_edges > 0:
            try:
                rot = mk("wave", n_wg=n_wg, n_copy=n_copy, wg_waves=wg_waves, forced_chunks=forced_chunks)
            except _lib.VicanError:
                if forced_chunks is not None:
                    raise
                rot = None
            # (capture-sized graphs are latency-bound, padding costs them nothing, and only the wave layout has the resident
            #  CG kernel: ragged rows of 2-5 edges - what real captures look like - pad a 64-row chunk by 15 % and stay here)
            if rot is not None and layout is None and self.n_edges >= 2_000_000 and rot.nslot > 1.40 * max(self.n_edges, 1) + 64 * epl * 8:
                # rows pack badly into 64-lane chunks (very short rows: a chunk holds at most 64 of them; rows of ~150 edges: one
                # per chunk): the block layout is taken where it pads LESS - with rows of 1-4 edges its chunks are limited by
                # their row count too and it pads more (measured: 2.07x against 1.6x).  Padding up to 1.4 slots per edge stays
                # in the wave layout: measured whole solves on 25 M edges of 1000 cameras (ms, wave / block) - 30 edges per row
                # (1.07 slots per edge in the wave layout) 6.39 / 7.26, 60 (1.07) 5.50 / 6.55, 100 (1.28) 5.79 / 6.42, 120 (1.07)
                # 5.18 / 6.40, 200 (1.28, one row per chunk) 4.92 / 5.27; only at 150 (1.71) the block layout wins, 5.74 / 6.28 -
                # the sweeps run alike per slot, the translation kernels of the wave layout are the faster ones (round 4; the
                # rule was 1.06 before and sent all of these to the block layout)
                alt = mk("block", block_threads=block_threads, n_wg=n_wg, n_copy=n_copy)
                if alt.nslot < rot.nslot:
                    rot = alt
        if rot is None:
            if layout == "wave":
                raise _lib.VicanError("the wave layout needs rows of at most %d edges and C <= 1024" % (64 * epl))
            rot = mk("block", block_threads=block_threads, n_wg=n_wg, n_copy=n_copy)
        have_t = w is not None
        # the translation arrays (w, u, v) live in the rotation layout's slot order, whichever layout that is: right-hand side
        # and CG sweep have a kernel for each (vican_trans.hip / vican_wtrans.hip).  Only the LSQR kernels are block-layout
        # only; a wave-layout graph builds that second layout on first use (lsqr_layout()).
        tl = rot
        self.rot, self.tl = rot, tl
        self.layout = rot.kind
        self._mk_block, self._storage = (lambda: mk("block")), storage
        # the rotation layout's numbers under the historical attribute names
        self.chunk_row0_host, self.max_rows, self.n_chunk, self.slots = rot.chunk_row0_host, rot.max_rows, rot.n_chunk, rot.slots
        self.block_threads, self.n_copy, self.n_wg, self.wg_waves = rot.block_threads, rot.n_copy, rot.n_wg, rot.wg_waves
        self.rows_per_wg_max, self.wg_chunk_cap, self.rows_per_wg_sweep = rot.rows_per_wg_max, rot.wg_chunk_cap, rot.rows_per_wg_sweep
        self.chunk_row0, self.idx = rot.chunk_row0, rot.idx
        self.blk = torch.empty(9 * rot.nslot, dtype=blk.dtype, device=dev)
        self.a = torch.empty(rot.nslot, dtype=blk.dtype, device=dev)
        self.w = torch.empty(tl.nslot, dtype=torch.float64, device=dev) if have_t else None
        self.u = torch.empty(3 * tl.nslot, dtype=torch.float64, device=dev) if have_t else None
        self.v = torch.empty(3 * tl.nslot, dtype=torch.float64, device=dev) if have_t else None
        self.w_cg = self.w
        self.desc = rot.describe(self.n_cam, storage, self.blk)
        self.desc_t = self.desc
        gref = C.byref(self.desc)
        row_ptr = row_ptr.to(dev, torch.int32).contiguous()
        col = col.to(dev, torch.int32).contiguous()
        blk = blk.contiguous(); a = a.to(blk.dtype).contiguous()
        if have_t:
            w, u, v = (t.to(dev, torch.float64).contiguous() for t in (w, u, v))
        if self.n_edges == 0:
            # a rank without rows (more ranks than timesteps): empty tensors have NULL data pointers, which the C entry
            # points reject - hand them one zero element each (nothing is read: there are no chunks)
            col = torch.zeros(1, dtype=torch.int32, device=dev)
            blk, a = torch.zeros(9, dtype=blk.dtype, device=dev), torch.zeros(1, dtype=blk.dtype, device=dev)
            if have_t:
                w, u, v = (torch.zeros(k, dtype=torch.float64, device=dev) for k in (1, 3, 3))
        st = _stream()
        perm_ws = torch.empty(rot.nslot, dtype=torch.int32, device=dev)
        _lib.check(lib.vican_pack_edges(gref, _ptr(row_ptr), _ptr(col), _ptr(blk), _ptr(a), _ptr(w) if have_t else None,
                                        _ptr(u) if have_t else None, _ptr(v) if have_t else None, _ptr(self.a),
                                        _ptr(self.w) if have_t else None, _ptr(self.u) if have_t else None,
                                        _ptr(self.v) if have_t else None, _ptr(perm_ws), st), "vican_pack_edges")
        # The legacy host-scalar LSQR path (cross-checks only: HipBackend.lsqr_host_scalars) needs a second, block layout of the
        # translation arrays, packed from the CSR-order inputs on first use.  They are retained - references, ~60 B per edge of
        # HBM, and through views the caller's whole upload - ONLY when asked for (keep_csr=True / VICAN_KEEP_CSR=1): the
        # device-resident LSQR and everything else run on the packed arrays, and "inputs may be freed by the caller" holds.
        # wave layout: the 2-byte index the edge sweeps stream (vican_graph_t.idx16: camera | row << 10)
        self.idx16 = None
        if rot.kind == "wave":
            self.idx16 = torch.empty(rot.nslot, dtype=torch.int16, device=dev)
            _lib.check(lib.vican_pack_idx16(gref, _ptr(self.idx16), st), "vican_pack_idx16")
            self.desc.idx16 = self.idx16.data_ptr()
        # float32 copy of the CG weights where every one of them is exactly a float32 (dtype=float32 problems: the reference's
        # J^T J is accumulated in float32) - the one-row CG product then streams 6 instead of 10 bytes per edge, same bits
        # (vican_graph_t.w32; plain slot order, the float64 array is in slot_pos8 order: [chunk][half][lane][2])
        self.w32 = None
        if (have_t and rot.kind == "wave" and epl == 4 and rot.n_chunk == self.n_time and rot.n_chunk > 0
                and os.environ.get("VICAN_CG_W32", "1") != "0"):
            w32, flag = torch.empty(rot.nslot, dtype=torch.float32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev)
            _lib.check(lib.vican_pack_w32(gref, _ptr(self.w), _ptr(w32), _ptr(flag), st), "vican_pack_w32")
            if int(flag.item()) == 0:
                self.w32 = w32
                self.desc.w32, self.desc.w32_src = self.w32.data_ptr(), self.w.data_ptr()
        keep = keep_csr if keep_csr is not None else os.environ.get("VICAN_KEEP_CSR") == "1"
        self._csr_t = (row_ptr, col, w, u, v) if (keep and have_t and rot.kind == "wave") else None
        self._lsqr_layout = None
        gref_t = gref
        # graph constants
        T1 = max(self.n_time, 1)
        f64 = dict(dtype=torch.float64, device=dev)
        self.row_sum_a, self.cam_sum_a = torch.zeros(T1, **f64), torch.zeros(self.n_cam, **f64)
        self.rnorm, self.fx = torch.zeros(T1, **f64), torch.zeros(_lib.FX_DOUBLES, **f64)
        cam_ws = torch.empty(self.n_cam, dtype=torch.int64, device=dev)
        # the three host scalars of the graph constants in ONE device->host read (max |a|; max w and max (|u| + |v|): the bounds
        # that size the fixed-point scales of the translation stage)
        amax_a, self.wmax, self.gmax = 1.0, 1.0, 1.0
        if self.n_edges:
            sc = [a.abs().max().to(torch.float64)]
            if have_t:
                sc += [w.max(), (u.norm(dim=1) + v.norm(dim=1)).max()]
            sc = torch.stack(sc).tolist()
            amax_a = float(sc[0])
            if have_t:
                self.wmax, self.gmax = float(sc[1]), float(sc[2])
        _lib.check(lib.vican_edge_sums(gref, _ptr(self.a), int(storage == _lib.STORE_F64), amax_a, _ptr(self.row_sum_a),
                                       _ptr(self.cam_sum_a), _ptr(cam_ws), st), "vican_edge_sums")
        _lib.check(lib.vican_block_norms(gref, _ptr(self.rnorm), _ptr(self.fx), st), "vican_block_norms")
        if have_t:
            self.row_sum_w, self.cam_sum_w = torch.zeros(T1, **f64), torch.zeros(self.n_cam, **f64)
            _lib.check(lib.vican_edge_sums(gref_t, _ptr(self.w), 1, self.wmax, _ptr(self.row_sum_w), _ptr(self.cam_sum_w),
                                           _ptr(cam_ws), st), "vican_edge_sums")
            # diagonal of the translation system when the caller knows it better than "sum of the weights": the front-end
            # passes the reference's own J^T J diagonal (float32-accumulated for dtype=float32, frontend.flatten_arrays);
            # deg_c is this RANK's share (the solver all-reduces it: rank 0 carries the vector, the others zeros)
            if deg_t is not None:
                self.row_sum_w[: self.n_time].copy_(deg_t.to(dev, torch.float64))
            if deg_c is not None:
                self.cam_sum_w.copy_(deg_c.to(dev, torch.float64))
        torch.cuda.current_stream().synchronize()      # inputs may be freed by the caller

    def lsqr_layout(self):
        """(layout, desc, w, u, v) in a BLOCK layout for the LSQR kernels (vican_lsqr.hip): the graph's own arrays when the
        rotation layout is a block layout, else a second layout packed on first use from the retained CSR-order inputs."""
        if self.rot.kind == "block":
            return self.rot, self.desc, self.w, self.u, self.v
        if self._lsqr_layout is None:
            lib = _lib.load()
            if self._csr_t is None:
                raise _lib.VicanError("the host-scalar LSQR path needs the CSR-order inputs: build the LocalGraph with keep_csr=True")
            row_ptr, col, w, u, v = self._csr_t
            bl = self._mk_block()
            desc = bl.describe(self.n_cam, self._storage, None)
            wb = torch.empty(bl.nslot, dtype=torch.float64, device=self.device)
            ub, vb = torch.empty(3 * bl.nslot, dtype=torch.float64, device=self.device), torch.empty(3 * bl.nslot, dtype=torch.float64, device=self.device)
            perm_ws = torch.empty(bl.nslot, dtype=torch.int32, device=self.device)
            _lib.check(lib.vican_pack_edges(C.byref(desc), _ptr(row_ptr), _ptr(col), None, None, _ptr(w), _ptr(u), _ptr(v), None,
                                            _ptr(wb), _ptr(ub), _ptr(vb), _ptr(perm_ws), _stream()), "vican_pack_edges")
            torch.cuda.current_stream().synchronize()
            self._lsqr_layout = (bl, desc, wb, ub, vb)
        return self._lsqr_layout

    # algorithmic HBM bytes of one operator sweep (B_op)
    def op_bytes(self, ncols=3):
        s = 4 if self.storage_dtype == torch.float32 else 8
        return self.n_edges * (9 * s + 4) + (self.n_time + 1) * 4 + self.n_time * 9 * 8 + 2 * 3 * self.n_cam * ncols * 8

    def padded_slots(self):
        return self.n_chunk * self.slots
