"""Run-time plumbing shared by the device-side modules: compute-unit count, the page-locked staging buffer of the per-call
transfers, the status-slot pool, the abort word of the bounded grid barriers, pointer / stream look-ups for ctypes.
(Split out of device.py in round 6; PyTorch is the allocator / stream provider only.)"""
from __future__ import annotations

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
