"""Device side of the solver: chunked edge layout + thin wrappers over the C ABI.

``LocalGraph`` owns (as PyTorch-ROCm tensors) the timestep-major chunked
CSR-of-3x3-blocks of ONE rank's timestep rows plus the graph constants computed
once at pack time (row / camera weight sums, block-norm bounds); ``HipBackend``
exposes each entry point of ``include/vican_hip.h`` on torch tensors.  PyTorch is
only the allocator / stream provider here - every numerical step is a
hand-written HIP kernel.  No CPU fallback: without a GPU construction raises.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import math
import os

import numpy as np
import torch

from . import _lib

_N_CU = None


def n_cu():
    """Compute units of the device the plans are sized for (256 on an MI355X; 256 also when planning without a GPU)."""
    global _N_CU
    if _N_CU is None:
        _N_CU = int(torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count) \
            if torch.cuda.is_available() else 256
    return _N_CU


# -- grid barriers of the cooperative kernels (include/vican_hip.h: vican_set_barrier_abort) ------------------------------
# One abort word per process in PINNED HOST memory: a workgroup whose barrier spin exceeds its time limit writes it, the
# host polls it for free (no copy, no synchronisation).
_ABORT = None


_NP_OF = {torch.float64: np.float64, torch.float32: np.float32, torch.int32: np.int32, torch.int64: np.int64, torch.uint8: np.uint8}
_STAGE = {"buf": None, "event": None, "lock": __import__("threading").Lock()}       # one staging buffer per process: serialised


def upload(dev, items):
    """Host arrays -> device tensors through ONE page-locked staging buffer and ONE copy: ``items`` = [(array, torch dtype)],
    returns the tensors (typed views of one device allocation, 256-byte aligned).  Conversions happen on the host in NumPy.
    Why not ``torch.from_numpy(a).to(dev)`` per array: a pageable host-to-device copy of ~1 MB now and then takes 70-100 ms
    on this platform (about one cold drop-in call in three: tools/dbg/upload2.py), a copy from page-locked memory never; the
    staging buffer is allocated once per process and reused (an event guards it against the copy still in flight)."""
    with _STAGE["lock"]:
        return _upload_locked(dev, items)


def _upload_locked(dev, items):
    offs, total = [], 0
    arrs = []
    for a, dt in items:
        a = np.asarray(a)
        arrs.append(a)
        offs.append(total)
        total += (a.size * np.dtype(_NP_OF[dt]).itemsize + 255) // 256 * 256
    total = max(total, 256)
    st = _STAGE
    if st["buf"] is None or st["buf"].numel() < total:
        st["buf"] = torch.empty(int(total * 1.5) + 4096, dtype=torch.uint8).pin_memory()
        st["event"] = None
    if st["event"] is not None:
        st["event"].synchronize()
    host = st["buf"].numpy()
    for a, (_, dt), o in zip(arrs, items, offs):
        if a.size:
            np.copyto(host[o:o + a.size * np.dtype(_NP_OF[dt]).itemsize].view(_NP_OF[dt]).reshape(a.shape), a, casting="unsafe")
    d = torch.empty(total, dtype=torch.uint8, device=dev)
    d.copy_(st["buf"][:total], non_blocking=True)
    ev = torch.cuda.Event()
    ev.record()
    st["event"] = ev
    out = []
    for a, (_, dt), o in zip(arrs, items, offs):
        nb = a.size * np.dtype(_NP_OF[dt]).itemsize
        out.append(d[o:o + nb].view(dt).view(a.shape) if a.size else torch.empty(a.shape, dtype=dt, device=dev))
    return out


def download(tensors):
    """Device tensors -> NumPy arrays through the page-locked staging buffer of ``upload`` (one synchronisation for all of
    them; copies into pageable memory show the same occasional 10-25 ms stalls as pageable uploads)."""
    with _STAGE["lock"]:
        return _download_locked(tensors)


def _download_locked(tensors):
    ts = [t.contiguous() for t in tensors]
    offs, total = [], 0
    for t in ts:
        offs.append(total)
        total += (t.numel() * t.element_size() + 255) // 256 * 256
    total = max(total, 256)
    st = _STAGE
    if st["buf"] is None or st["buf"].numel() < total:
        st["buf"] = torch.empty(int(total * 1.5) + 4096, dtype=torch.uint8).pin_memory()
        st["event"] = None
    if st["event"] is not None:
        st["event"].synchronize()
        st["event"] = None
    for t, o in zip(ts, offs):
        nb = t.numel() * t.element_size()
        if nb:
            st["buf"][o:o + nb].view(t.dtype).view(t.shape).copy_(t, non_blocking=True)
    torch.cuda.current_stream().synchronize()
    host = st["buf"].numpy()
    return [host[o:o + t.numel() * t.element_size()].view(_NP_OF[t.dtype]).reshape(tuple(t.shape)).copy() for t, o in zip(ts, offs)]


_STATUS_POOL = {}          # (device, doubles) -> free (pinned buffer, copy-done event, ready event, side stream) slots, see post_status


_ABORT_TLS = __import__("threading").local()


def barrier_abort_word(timeout_us=None):
    """The process-wide abort word (a pinned int32 tensor).  The library keeps the registration PER HOST THREAD
    (vican_set_barrier_abort: thread_local, like the gate and the launch timer), so every thread that builds or uses a
    backend registers the word once - a backend used on a second thread would otherwise launch its cooperative kernels
    with unbounded spins.  timeout_us: spin limit of every grid barrier (default VICAN_BARRIER_TIMEOUT_US or 2 s)."""
    global _ABORT
    if _ABORT is None:
        _ABORT = torch.zeros(4, dtype=torch.int32).pin_memory()
    if timeout_us is not None or not getattr(_ABORT_TLS, "registered", False):
        us = int(timeout_us if timeout_us is not None else getattr(_ABORT_TLS, "us", os.environ.get("VICAN_BARRIER_TIMEOUT_US", 0)))
        _lib.check(_lib.load().vican_set_barrier_abort(C.c_void_p(_ABORT.data_ptr()), us), "vican_set_barrier_abort")
        _ABORT_TLS.registered, _ABORT_TLS.us = True, us
    return _ABORT


STREAM_NT_BYTES = 192 << 20       # edge streams above this are read with non-temporal loads (LocalGraph, vican_sweep_common.h)
X_BOUND = math.sqrt(3.0)      # |x_c|_F of every sweep input: orthonormal columns / stacked rotations


def _ptr(t):
    """Device address of a tensor as a plain int (every prototype declares its pointers c_void_p: ctypes converts an int at half
    the cost of a c_void_p object built per argument - ten pointers per launch, a hundred launches per capture-sized solve)."""
    return None if t is None else t.data_ptr()


_stream_cache = {}


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """hipStream_t of torch's current stream.  The lookup is on the host critical path of every launch (a capture-sized solve is
    ~100 launches in 1.5 ms): torch's raw-stream getter where this build has it (0.2 us), else torch.cuda.current_stream()
    cached per stream object (1.5-2 us)."""
    if _raw_stream is not None and _raw_device is not None:
        h = _raw_stream(_raw_device())
        c = _stream_cache.get(h)
        if c is None:
            c = _stream_cache[h] = C.c_void_p(h)
        return c
    s = torch.cuda.current_stream()
    h = _stream_cache.get(s)
    if h is None:
        h = _stream_cache[s] = C.c_void_p(s.cuda_stream)
    return h


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


def make_backend(n_cam, row_ptr, col, blk, a, w=None, u=None, v=None, deg_t=None, deg_c=None, row_ptr_host=None):
    """(graph, backend) for one rank's rows: the fused layouts up to TILE_CAMS cameras, camera tiles beyond.
    deg_t / deg_c: diagonal of the translation system for this rank's rows / this rank's share of the camera diagonal
    (default: sums of w)."""
    import os
    tile = int(os.environ.get("VICAN_TILE_CAMS") or TILE_CAMS)
    if n_cam > tile:
        g = TiledGraph(n_cam, row_ptr, col, blk, a, w, u, v, tile=tile, deg_t=deg_t, deg_c=deg_c,
                       permute_rows=True)
        return g, TiledBackend(g)
    g = LocalGraph(n_cam, row_ptr, col, blk, a, w, u, v, deg_t=deg_t, deg_c=deg_c, row_ptr_host=row_ptr_host)
    return g, HipBackend(g)
