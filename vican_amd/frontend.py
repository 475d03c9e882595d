"""Host front-end: reference edge dict -> flat arrays -> merged timestep-major CSR.

Covers the host-side rows of the hot path:
the user callables (``edge_filter``, ``noise_model_r``, ``noise_model_t``) are
arbitrary Python and are evaluated once per edge (reference bipgo.py:204,212,
423,449 - the reference calls the filter twice; callables are treated as pure);
everything else is vectorised NumPy.  Output feeds ``device.LocalGraph``.
"""
from __future__ import annotations

import numpy as np


class Problem:
    """Flattened, merged problem (all float64 on the host)."""
    __slots__ = ("cam_names", "time_names", "tnodes", "tnode_of_cam", "tnode_of_time", "root",
                 "n_src", "row_ptr", "col", "blk", "a", "w", "u", "v", "deg_c", "deg_t",
                 "src_cam", "src_time", "src_t", "src_qtau", "src_kt", "on_device", "row_ptr_host", "col_host")

    def host_csr(self):
        """(row_ptr, col) as NumPy arrays - the problem's own arrays, or copies of the device arrays of a device merge."""
        if getattr(self, "on_device", False):
            return self.row_ptr_host, self.col_host
        return self.row_ptr, self.col

    @property
    def n_cam(self):
        return len(self.cam_names)

    @property
    def n_time(self):
        return len(self.time_names)

    @property
    def n_edges(self):
        return len(self.col)


def vectorized(columns_fn, python_scalars=True):
    """Decorator: give a per-edge callable (``edge_filter`` / ``noise_model_r`` / ``noise_model_t``: value dict -> bool / float,
    bipgo.py:204,212,449) a COLUMN form that the front-end calls once instead of once per edge:

        @vican_amd.frontend.vectorized(lambda cols: np.exp(-cols["reprojected_err"]))
        def noise_t(edge): return math.exp(-edge["reprojected_err"])

    ``columns_fn(cols)`` receives an ``EdgeColumns`` (``cols["reprojected_err"]`` [n], ``cols["corners"]`` [n,4,2],
    ``cols["R"]`` [n,3,3] / ``cols["t"]`` [n,3] of ``pose``, any other field as an object array; gathered on first use) and
    returns one value per edge (or a scalar).  The scalar form stays the definition - it is what the reference calls and what
    runs when the attribute is absent; the two must agree (tests/test_frontend_cpu.py checks the shipped ones bit for bit).

    One thing the column form cannot show is the TYPE the scalar form returns, and numpy's product `k_r * R` (bipgo.py:213)
    depends on it when R is a float32 array (every pose built from a 4x4 matrix): a Python float is a weak scalar and gives a
    float32 product, an np.float64 scalar a float64 one (f32_product_mask).  ``python_scalars`` says which: True (default) = the
    scalar form returns Python floats (``math.exp(...)``, ``float(...)``, arithmetic on the dict's Python values) - a float64
    column then counts as Python floats; False = it returns NumPy scalars (``np.exp(...)`` of a NumPy value) and the column's
    own dtype decides, as it does for the arrays of ``flatten_arrays``."""
    def deco(fn):
        fn.vectorized = columns_fn
        fn.vectorized_python_scalars = bool(python_scalars)
        return fn
    return deco


class EdgeColumns:
    """Lazy column view of a list of edge value dicts (the argument of a ``.vectorized`` callable)."""
    __slots__ = ("_vals", "_cache")

    def __init__(self, vals, cache=None):
        self._vals, self._cache = vals, ({} if cache is None else cache)

    def __len__(self):
        return len(self._vals)

    def poses(self):
        c = self._cache.get("pose")
        if c is None:
            c = self._cache["pose"] = [v["pose"] for v in self._vals]
        return c

    def _pose_field(self, attr, method):
        """[p.R() for p in poses] / [p.t() ...] - for poses that are exactly this package's SE3 (whose R() / t() return the
        attribute) read as the attribute in a C-level loop: 2.6 x faster than 80 000 Python method calls."""
        import operator
        from .geometry import SE3
        poses = self.poses()
        if self._cache.get("plain_se3") is None:
            self._cache["plain_se3"] = set(map(type, poses)) == {SE3}
        if self._cache["plain_se3"]:
            return list(map(operator.attrgetter(attr), poses))
        return [getattr(p, method)() for p in poses]

    def __getitem__(self, name):
        c = self._cache.get(name)
        if c is None:
            n = len(self._vals)
            if name == "pose":
                return self.poses()
            if name in ("R", "t"):
                if n and self._gather_poses_c():
                    return self._cache[name]
                if name == "R":
                    rl = self._pose_field("_R", "R")
                    # (per-edge flags, whatever the first pose holds: a mixed list must not lose its float32 products)
                    self._cache["r_is_f32"] = np.fromiter((getattr(r, "dtype", None) == np.float32 for r in rl), dtype=bool, count=n)
                    c = _stack_f64(rl, (3, 3)) if n else np.zeros((0, 3, 3))
                else:
                    c = _stack_f64(self._pose_field("_t", "t"), (3,)) if n else np.zeros((0, 3))
            elif name == "reprojected_err":
                c = self._gather_item_c(name, None)
                if c is None:
                    c = np.fromiter((v["reprojected_err"] for v in self._vals), dtype=np.float64, count=n)
            elif name == "corners":
                c = self._gather_item_c(name, np.shape(self._vals[0]["corners"])) if n else None
                if c is None:
                    c = np.array([v["corners"] for v in self._vals], dtype=np.float64)
            else:
                c = np.empty(n, dtype=object)
                c[:] = [v[name] for v in self._vals]
            self._cache[name] = c
        return c

    def _gather_poses_c(self):
        """R, t and the float32 flags of R in ONE pass in C (csrc/vican_fastpath.c); False: not available / not recognised."""
        from . import _lib
        from .geometry import SE3
        fp = _lib.fastpath()
        if fp is None or self._cache.get("no_c_poses"):
            return False
        out = fp.gather_poses(self._vals, SE3)
        if out is None:
            self._cache["no_c_poses"] = True
            return False
        self._cache["R"], self._cache["t"] = out[0], out[1]
        self._cache["r_is_f32"] = out[2].view(bool)
        return True

    def _gather_item_c(self, key, shape):
        """[n, *shape] float64 of v[key] in one C pass (shape None: scalars -> [n]); None: not available / not recognised."""
        from . import _lib
        fp = _lib.fastpath()
        if fp is None:
            return None
        out = fp.gather_item(self._vals, key, int(np.prod(shape)) if shape else 1)
        if out is None:
            return None
        return out.reshape((len(self._vals),) + tuple(shape)) if shape else out[:, 0]

    def r_is_f32(self):
        """bool [n]: which edges' pose.R() is a float32 array (f32_product_mask)."""
        self["R"]
        return self._cache.get("r_is_f32")

    def select(self, keep):
        """The columns of the edges where `keep` (bool [n]) holds (already gathered columns are sliced, not re-gathered)."""
        import itertools
        vals = list(itertools.compress(self._vals, keep))
        cache = {k: (list(itertools.compress(v, keep)) if isinstance(v, list) else v if isinstance(v, bool) else v[keep])
                 for k, v in self._cache.items() if v is not None}           # (bool entries: flags of the whole list)
        return EdgeColumns(vals, cache)


def _call_columns(fn, cols, dtype):
    """fn.vectorized(cols) as an [n] array of `dtype` (scalars broadcast), or None when fn has no column form."""
    vec = getattr(fn, "vectorized", None)
    if vec is None:
        return None
    out = np.asarray(vec(cols), dtype=dtype)
    if out.shape != (len(cols),):
        out = np.broadcast_to(out, (len(cols),)).copy()
    return out


# the last kept key list and its index codes (time series: same keys, new values) as ONE immutable pair, read and replaced in one
# step each: concurrent flatten() calls can then at worst miss the cache, never pair one call's keys with another's codes.
# clear_codes_cache() drops it (it pins the previous call's key list - ~80 000 tuples at large_shop size).
_CODES_CACHE = [(None, None)]


def clear_codes_cache():
    _CODES_CACHE[0] = (None, None)


def flatten(src_edges, constraints, noise_model_r, noise_model_t, edge_filter, dtype=np.float32, merge=None) -> Problem:
    """Filter, weight, apply marker constraints and merge multi-marker edges.  `merge`: the numeric half - None =
    merge_host (NumPy), or a callable (ix, R, t, k_r, k_t, dtype) -> Problem such as device.merge_edges.

    Per kept source edge e = (c, "t_m") (bipgo.py:203-221, 445-468):
        M_ct += k_r * R~_e R_m^T R_root,   a_ct += k_r
        w_ct += kf^2,  u_ct += kf k_t t~_e,  v_ct += kf k_t (R_root^T R_m) tau_m
    with tau_m = trans(S_m^-1 S_root) and kf = k_t rounded to ``dtype`` (the reference
    stores the incidence matrix in ``dtype`` but the measurements in float64,
    bipgo.py:434-439).

    The user callables are the only per-edge Python left: each is evaluated once per (kept) edge in its own pass over the
    value dicts (callables are treated as pure: the reference itself calls the filter twice) - or ONCE for all edges when it
    carries a column form (``vectorized``).  Ids are split and indexed as arrays; the indices of a key list seen in the
    previous call are reused (time series of captures with the same detections)."""
    import itertools
    keys, vals = list(src_edges.keys()), list(src_edges.values())
    n_all = len(vals)
    cols = EdgeColumns(vals)
    keep = _call_columns(edge_filter, cols, bool)
    if keep is None:
        keep = np.array([bool(edge_filter(v)) for v in vals], dtype=bool) if n_all else np.zeros(0, dtype=bool)
    if not keep.all():
        keys, cols = list(itertools.compress(keys, keep)), cols.select(keep)
    vals = cols._vals
    n = len(vals)
    if n == 0:
        raise ValueError("no edge passes edge_filter")
    kr = kr_raw = _call_columns(noise_model_r, cols, np.float64)
    if kr is None:
        kr_raw = [noise_model_r(v) for v in vals]
        kr = np.array(kr_raw, dtype=np.float64)
    elif getattr(noise_model_r, "vectorized_python_scalars", True):
        kr_raw = _PythonFloats(n)          # (the scalar form returns Python floats: weak scalars in numpy's k_r * R - `vectorized`)
    kt = _call_columns(noise_model_t, cols, np.float64)
    if kt is None:
        kt = np.array([noise_model_t(v) for v in vals], dtype=np.float64)
    R, t = cols["R"], cols["t"]
    ck, cc = _CODES_CACHE[0]
    codes = cc if ck is not None and len(ck) == n and ck == keys else None
    if codes is None:
        cams = [k[0] for k in keys]
        tm = np.array([k[1] for k in keys])
        times = marks = None
        if tm.dtype.kind == "U" and tm.ndim == 1:
            parts = np.char.partition(tm, "_")
            if (parts[:, 1] == "_").all() and not (np.char.find(parts[:, 2], "_") >= 0).any():
                times, marks = parts[:, 0], parts[:, 2]
        if times is None:                                          # not exactly one '_' somewhere: the reference's own unpacking (and its error)
            tsm = [k[1].split("_") for k in keys]
            times, marks = [a for a, _ in tsm], [b for _, b in tsm]
        codes = index_codes(cams, times, marks)
        _CODES_CACHE[0] = (keys, codes)
    # (float32 rotations - poses built from a 4x4 matrix: numpy weights them in float32, f32_product_mask; one dtype look-up for
    #  the usual float64 ones)
    kw = {}
    r32 = cols.r_is_f32()
    if r32 is not None and n and r32.any():
        mask = f32_product_mask(kr_raw, r32)
        if mask is not None:
            kw["kr_f32"] = mask
    return (merge or merge_host)(index_edges(None, None, None, constraints, codes=codes), R, t, kr, kt, dtype, **kw)


def _stack_f64(items, shape):
    """[n, *shape] float64 from a list of small arrays: one C-level conversion (np.stack of per-item np.asarray calls took
    six times as long for 80 000 rotations); lists of mixed shapes take the per-item path."""
    # float64 ndarrays of the right size (what SE3 holds): their bytes joined and viewed - 9.5 against 16 ms for 80 000
    # rotations; anything else (float32 poses, lists, other shapes) shows as a wrong total length and takes the paths below
    size = int(np.prod(shape))
    try:
        if len(items) and type(items[0]) is np.ndarray and items[0].dtype == np.float64:
            try:
                raw = b"".join(items)                              # (C-contiguous arrays through the buffer protocol: 2.4 x faster
            except (BufferError, TypeError, ValueError):           #  than a tobytes() per item; views of a 4x4 are not contiguous)
                raw = b"".join([x.tobytes() for x in items])
            if len(raw) == 8 * size * len(items) and all(type(x) is np.ndarray and x.dtype == np.float64 for x in items[:: max(1, len(items) // 64)]):
                return np.frombuffer(raw, dtype=np.float64).reshape((len(items),) + shape).copy()
    except (AttributeError, TypeError):
        pass
    try:
        return np.array(items, dtype=np.float64).reshape((len(items),) + shape)
    except (ValueError, TypeError):
        return np.stack([np.asarray(x, dtype=np.float64).reshape(shape) for x in items])


def _packed_keys(a):
    """ASCII strings of at most 8 characters as big-endian 64-bit integers (zero-padded: numeric order = string order, a
    prefix sorts first), or None."""
    width = a.dtype.itemsize // 4
    if a.dtype.kind != "U" or not (0 < width <= 8) or a.ndim != 1 or not len(a):
        return None
    cp = np.ascontiguousarray(a).view(np.uint32).reshape(len(a), width)          # UCS-4 code points, zero-padded
    if int(cp.max()) >= 128:
        return None
    buf = np.zeros((len(a), 8), dtype=np.uint8)
    buf[:, :width] = cp                                         # one byte per character
    return buf.view(">u8").ravel().astype(np.uint64)


def _ascii_keys(ids):
    """Packed keys (see _packed_keys) straight from a list / array of ids: ONE C-level conversion to fixed 9-byte strings -
    np.asarray of a list of 80 000 Python strings alone took 8 ms, this takes 3.  None if an id is longer than 8 bytes, not
    ASCII, or the input is not one-dimensional (the caller then takes the general path)."""
    try:
        b = np.array(ids, dtype="S9")
    except (UnicodeEncodeError, ValueError, TypeError):
        return None
    if b.ndim != 1 or not len(b):
        return None
    raw = b.view(np.uint8).reshape(len(b), 9)
    if raw[:, 8].any():
        return None
    return np.ascontiguousarray(raw[:, :8]).view(">u8").ravel().astype(np.uint64)


def _key_names(uk):
    """The strings of packed keys, without a Python loop: the big-endian bytes of a key ARE its characters ('S8' drops the padding)."""
    return uk.astype(">u8").view("S8").astype("U8")


def sorted_codes(ids, prefix="", with_keys=False):
    """(sorted unique strings, index of every entry into them) - what ``np.unique(prefix + ids, return_inverse=True)``
    returns (bipgo.py:225-229: node order = string sort order), without sorting 80 000 UCS-4 strings by comparison: ASCII
    ids of at most 8 characters are packed into big-endian 64-bit integers, whose numeric order IS their string order, and
    sorted as integers.  A common prefix does not change the order of the strings, only their spelling.  Anything else falls
    back to ``np.unique``.  with_keys: also return the packed keys of the unique strings (None on the fallback path)."""
    # (a NumPy unicode array: its UCS-4 code points ARE the input of _packed_keys - no conversion; a Python list / tuple of
    #  strings: one C-level conversion to fixed-width bytes, _ascii_keys; anything else, or ids those refuse: np.asarray)
    a = keys = None
    if isinstance(ids, np.ndarray) and ids.dtype.kind == "U" and ids.ndim == 1:
        a = ids
        keys = _packed_keys(a)
    elif not isinstance(ids, np.ndarray):
        keys = _ascii_keys(ids)
    if keys is None:
        if a is None:
            a = np.asarray(ids)
            if a.dtype.kind != "U":
                a = a.astype(str)
            keys = _packed_keys(a)
    if keys is not None:
        uk, inv = np.unique(keys, return_inverse=True)
        names = _key_names(uk)
        if prefix:
            names = np.char.add(prefix, names)
    else:
        uk = None
        names, inv = np.unique(np.char.add(prefix, a) if prefix else a, return_inverse=True)
    inv = inv.astype(np.int64)
    return (names, inv, uk) if with_keys else (names, inv)


class EdgeIndex:
    """Host half of the front-end: string ids -> node / marker indices, node orderings, the
    per-marker constraint tables.  Everything per-edge and numeric happens in ``merge_host`` (NumPy) or on the device
    (``device.merge_edges``, vican_merge.hip) from these indices."""
    __slots__ = ("n", "root", "cam_names", "time_names", "tnodes", "tnode_of_cam", "tnode_of_time", "ci", "ti", "mi", "CmT", "qtau")


class _Codes:
    """String ids -> indices (the part of EdgeIndex that depends on the keys alone, not on the constraints or the values)."""
    __slots__ = ("n", "mk_names", "mi", "cam_names", "ci", "time_names", "ti", "tnodes", "tnode_of_cam", "tnode_of_time")


def index_codes(cam_ids, time_ids, marker_ids) -> _Codes:
    cd = _Codes()
    cd.n = len(cam_ids)
    if cd.n == 0:
        raise ValueError("no edge passes edge_filter")
    cd.mk_names, cd.mi = sorted_codes(marker_ids)
    # bipgo.py:225-229 sorts 'c' + id / 't' + id: a common prefix changes the spelling, not the order
    cd.cam_names, cd.ci, kc = sorted_codes(cam_ids, with_keys=True)
    cd.time_names, cd.ti, kt = sorted_codes(time_ids, with_keys=True)
    # translation unknowns: cameras and "<t>_0" nodes in ONE string-sorted list (bipgo.py:426-430)
    t0 = np.char.add(cd.time_names, "_0")
    if kc is not None and kt is not None:
        # on the packed keys: "<t>_0" = the key of <t> with '_' and '0' in its next two bytes (needs len(t) <= 6)
        ln = (kt.astype(">u8").view(np.uint8).reshape(-1, 8) != 0).sum(1).astype(np.uint64)
        if int(ln.max()) <= 6:
            k0 = kt | (np.uint64(ord("_")) << (np.uint64(8) * (np.uint64(7) - ln))) | (np.uint64(ord("0")) << (np.uint64(8) * (np.uint64(6) - ln)))
            allk = np.concatenate([kc, k0])
            order = np.argsort(allk, kind="stable")
            sk = allk[order]
            if not (sk[1:] == sk[:-1]).any():                       # (a camera called "<t>_0": the general path merges them)
                pos = np.empty(len(allk), dtype=np.int64)
                pos[order] = np.arange(len(allk), dtype=np.int64)
                cd.tnodes = np.concatenate([cd.cam_names, t0])[order]
                cd.tnode_of_cam, cd.tnode_of_time = pos[: len(kc)], pos[len(kc):]
                return cd
    cd.tnodes = np.unique(np.concatenate([cd.cam_names, t0]))
    cd.tnode_of_cam = np.searchsorted(cd.tnodes, cd.cam_names).astype(np.int64)
    cd.tnode_of_time = np.searchsorted(cd.tnodes, t0).astype(np.int64)
    return cd


def index_edges(cam_ids, time_ids, marker_ids, constraints, codes=None) -> EdgeIndex:
    """codes: the ids' indices if the caller has them already (index_codes of the same ids)."""
    cd = codes if codes is not None else index_codes(cam_ids, time_ids, marker_ids)
    ix = EdgeIndex()
    ix.n = cd.n
    ix.root = str(min(list(constraints.keys())))                   # bipgo.py:196,411 (string min)
    r_root = np.asarray(constraints[ix.root].R(), dtype=np.float64)
    # per-marker constraint tables (KeyError for an unknown marker id, as bipgo.py:209)
    mk_names, ix.mi = cd.mk_names, cd.mi
    M = len(mk_names)
    ix.CmT = np.empty((M, 3, 3)); Q = np.empty((M, 3, 3)); tau = np.empty((M, 3))
    for i, m in enumerate(mk_names):
        cm = constraints[str(m)]
        r_m = np.asarray(cm.R(), dtype=np.float64)
        ix.CmT[i] = r_m.T @ r_root                                 # bipgo.py:213
        Q[i] = r_root.T @ r_m                                      # bipgo.py:451
        tau[i] = np.asarray((cm.inv() @ constraints[ix.root]).t(), dtype=np.float64)   # bipgo.py:452
    ix.qtau = np.einsum("mij,mj->mi", Q, tau)
    ix.cam_names, ix.ci, ix.time_names, ix.ti = cd.cam_names, cd.ci, cd.time_names, cd.ti
    ix.tnodes, ix.tnode_of_cam, ix.tnode_of_time = cd.tnodes, cd.tnode_of_cam, cd.tnode_of_time
    return ix


class _PythonFloats:
    """Stands for "n Python floats" in f32_product_mask without materialising them (a column-form weight whose scalar form
    returns Python floats: frontend.vectorized)."""
    def __init__(self, n):
        self.n = n

    def __len__(self):
        return self.n


def f32_product_mask(weights, rotations):
    """Per kept source edge: does numpy form the reference's `k_r * v['pose'].R()` (bipgo.py:213) in float32?  It does when the
    rotation is a float32 array - every pose of object mode (SE3.inv() assembles a float32 4x4, geometry.py:239-243), poses
    built from a 4x4 matrix - and the weight is a Python scalar (NEP 50: weak) or a float32 one: the weight is then rounded to
    float32 and so is every product, 6e-8 relative - which showed as 1e-9 (median) ... 6e-8 rad between the product's float64
    object-mode rotations and the reference's while this was formed in float64 (tools/object_offset_probe.py; camera-mode scenes
    with float64 rotations agree to 1e-12).  The rule is asked of the installed numpy, one example per scalar type.
    weights: what the callable returned (a list / tuple of scalars, or an array); rotations: the list of R() arrays, or a bool
    array (True: that edge's R() is float32).
    Returns a bool array [n], or None when no product is a float32 one."""
    n = len(weights)
    if n == 0 or len(rotations) != n:
        return None
    if isinstance(rotations, np.ndarray) and rotations.dtype == bool:
        r32 = rotations
    else:
        r32 = np.array([getattr(r, "dtype", None) == np.float32 for r in rotations], dtype=bool)
    if not r32.any():
        return None
    probe = np.zeros(1, dtype=np.float32)
    if isinstance(weights, _PythonFloats):
        k32 = np.full(n, (1.0 * probe).dtype == np.float32)
    elif isinstance(weights, np.ndarray):
        k32 = np.full(n, (weights[:1] * probe).dtype == np.float32)
    else:
        types = list(map(type, weights))
        rule = {}
        for tp in set(types):
            try:
                rule[tp] = (weights[types.index(tp)] * probe).dtype == np.float32
            except Exception:
                rule[tp] = False
        k32 = np.full(n, next(iter(rule.values()))) if len(rule) == 1 else np.array([rule[tp] for tp in types])
    mask = r32 & k32
    return mask if mask.any() else None


def weighted_rotations(kr, R, CmT_e, kr_f32=None):
    """(k_r R~_e) (R_m^T R_root) per edge with every rounding spelled out - products and sums in index order, no fused
    multiply-add, no BLAS - so that the device kernel (vican_merge.hip, same order, contraction off) reproduces it to the bit
    on any machine: out[e,i,j] = ((a_i0 b_0j + a_i1 b_1j) + a_i2 b_2j), a = k_r R (in float32 where `kr_f32`: f32_product_mask)."""
    A = kr[:, None, None] * R
    if kr_f32 is not None:
        m = np.asarray(kr_f32, dtype=bool)
        A[m] = (kr[m].astype(np.float32)[:, None, None] * R[m].astype(np.float32)).astype(np.float64)
    out = np.empty_like(A)
    for i in range(3):
        for j in range(3):
            out[:, i, j] = (A[:, i, 0] * CmT_e[:, 0, j] + A[:, i, 1] * CmT_e[:, 1, j]) + A[:, i, 2] * CmT_e[:, 2, j]
    return out


def merge_host(ix: EdgeIndex, R, t, k_r, k_t, dtype=np.float32, kr_f32=None) -> Problem:
    """Numeric half of the front-end on the host (bipgo.py:203-221, 445-469): per kept source edge
        M_ct += k_r R~_e R_m^T R_root,   a_ct += k_r,   w_ct += kf^2,   u_ct += kf k_t t~_e,   v_ct += kf k_t (R_root^T R_m) tau_m
    summed SEQUENTIALLY in source-edge order (np.add.at: unbuffered, in index order) - the order scipy's csr_matmat uses for
    J^T J and the order the device merge uses, so both produce the same bits."""
    n = ix.n
    R = np.asarray(R, dtype=np.float64).reshape(n, 3, 3)
    t = np.asarray(t, dtype=np.float64).reshape(n, 3)
    kr = np.asarray(k_r, dtype=np.float64); kt = np.asarray(k_t, dtype=np.float64)
    kf = kt.astype(dtype).astype(np.float64)
    ci, ti = ix.ci, ix.ti
    C, T = len(ix.cam_names), len(ix.time_names)
    key = ti * C + ci                                               # timestep-major merged-edge key
    ukey, inv = np.unique(key, return_inverse=True)
    E = len(ukey)

    def seg(x):
        out = np.zeros((E,) + x.shape[1:], dtype=x.dtype)
        np.add.at(out, inv, x)
        return out

    p = Problem()
    p.on_device = False
    p.root, p.n_src = ix.root, n
    p.cam_names, p.time_names, p.tnodes = ix.cam_names, ix.time_names, ix.tnodes
    p.tnode_of_cam, p.tnode_of_time = ix.tnode_of_cam, ix.tnode_of_time
    p.col = (ukey % C).astype(np.int32)
    rows = (ukey // C).astype(np.int64)
    p.row_ptr = np.zeros(T + 1, dtype=np.int32)
    np.cumsum(np.bincount(rows, minlength=T), out=p.row_ptr[1:])
    p.blk = seg(weighted_rotations(kr, R, ix.CmT[ix.mi], kr_f32)).reshape(E, 9)
    p.a = seg(kr)
    # Entries of the reference's normal matrix J^T J (bipgo.py:477) AS SCIPY FORMS THEM: csr_matmat accumulates the products
    # k k of the incidence entries in the matrix dtype, sequentially in source-edge order - for dtype=float32 that is float32
    # products summed in float32, for the merged off-diagonal entries w_ct and for the DIAGONAL deg_n = sum over ALL source
    # edges at node n alike.  In float32 the diagonal therefore differs from the sum of the (rounded) off-diagonal entries
    # by ~1e-7 relative - the reference's matrix is not an exact Laplacian, and a CG that used sum_c w_ct instead would run on
    # a matrix perturbed eight orders above rounding (rounds 1-2: 109 of 975 f32 campaign scenes beyond 1e-4 m).
    # np.add.at adds unbuffered, in index order, in the array's dtype: exactly that accumulation.
    kd = kt.astype(dtype)
    prod = kd * kd
    w = np.zeros(E, dtype=dtype); np.add.at(w, inv, prod)
    deg_c = np.zeros(C, dtype=dtype); np.add.at(deg_c, ci, prod)
    deg_t = np.zeros(T, dtype=dtype); np.add.at(deg_t, ti, prod)
    p.w, p.deg_c, p.deg_t = w.astype(np.float64), deg_c.astype(np.float64), deg_t.astype(np.float64)
    kk = (kf * kt)[:, None]
    p.u = seg(kk * t)
    p.v = seg(kk * ix.qtau[ix.mi])
    # per-source-edge data of the un-merged right-hand side b (bipgo.py:451-461); only its norm is
    # needed (LSQR stopping tests), see bnorm2()
    p.src_cam, p.src_time, p.src_t, p.src_qtau, p.src_kt = ci, ti, t, ix.qtau[ix.mi], kt
    return p


def flatten_arrays(cam_ids, time_ids, marker_ids, R, t, k_r, k_t, constraints, dtype=np.float32, merge=None, kr_f32=None) -> Problem:
    """The array form of ``flatten`` (everything after the per-edge Python loop): one entry per KEPT source edge - camera
    id, timestamp and marker id (strings, as in the reference's keys ``(cam, "<t>_<marker>")``), measured rotation R
    [n,3,3] and translation t [n,3] of the marker in the camera frame, and the two weights the reference obtains from
    ``noise_model_r`` / ``noise_model_t``.  Callers that already hold their detections as arrays (or evaluate their
    weight functions vectorised) skip the edge dict and its per-edge callables altogether.  = index_edges + merge_host;
    the drop-in API on a GPU box runs index_edges on the host and the merge on the device (device.merge_edges)."""
    kw = {} if kr_f32 is None else {"kr_f32": kr_f32}          # (f32_product_mask; array callers: float64 products)
    return (merge or merge_host)(index_edges(cam_ids, time_ids, marker_ids, constraints), R, t, k_r, k_t, dtype, **kw)


def flatten_so3(src_edges, constraints, noise_model, edge_filter) -> Problem:
    """Rotation-only flattening of the non-eliminated variant ``bipartite_so3sync`` (bipgo.py:33-52):
        M_ct += k_r * R~_e R_m R_root^T,   a_ct += k_r
    (note R_m R_root^T here against R_m^T R_root in ``flatten`` - the reference's two variants use different
    constraint conventions, bipgo.py:45 vs :213).  Only the rotation-stage fields of Problem are filled."""
    root = str(min(list(constraints.keys())))                      # bipgo.py:33
    r_root = np.asarray(constraints[root].R(), dtype=np.float64)
    cams, times, marks, poses, kr = [], [], [], [], []
    for key, val in src_edges.items():
        if not edge_filter(val):
            continue
        ts, mid = key[1].split("_")[:2]
        cams.append(key[0]); times.append(ts); marks.append(mid)
        poses.append(val["pose"]); kr.append(noise_model(val))
    n = len(cams)
    if n == 0:
        raise ValueError("no edge passes edge_filter")
    mk_names, mk_idx = np.unique(np.array(marks, dtype=str), return_inverse=True)
    Cm = np.stack([np.asarray(constraints[str(m)].R(), dtype=np.float64) @ r_root.T for m in mk_names])   # KeyError as bipgo.py:41
    Rl = [p.R() for p in poses]
    R = _stack_f64(Rl, (3, 3))
    m32 = f32_product_mask(kr, Rl)          # (float32 rotations - per edge, mixed lists included: numpy's float32 product)
    kr = np.asarray(kr, dtype=np.float64)
    A = kr[:, None, None] * R
    if m32 is not None:
        A[m32] = (kr[m32].astype(np.float32)[:, None, None] * R[m32].astype(np.float32)).astype(np.float64)
    wR = A @ Cm[mk_idx]
    cam_nodes, ci = np.unique(np.char.add("c", np.array(cams, dtype=str)), return_inverse=True)           # bipgo.py:54
    time_nodes, ti = np.unique(np.char.add("t", np.array(times, dtype=str)), return_inverse=True)
    C, T = len(cam_nodes), len(time_nodes)
    key = ti.astype(np.int64) * C + ci
    ukey, inv = np.unique(key, return_inverse=True)
    order = np.argsort(inv, kind="stable")
    starts = np.searchsorted(inv[order], np.arange(len(ukey)))
    p = Problem()
    p.root, p.n_src = root, n
    p.cam_names = np.array([c[1:] for c in cam_nodes])
    p.time_names = np.array([s[1:] for s in time_nodes])
    p.col = (ukey % C).astype(np.int32)
    p.row_ptr = np.zeros(T + 1, dtype=np.int32)
    np.cumsum(np.bincount((ukey // C).astype(np.int64), minlength=T), out=p.row_ptr[1:])
    p.blk = np.add.reduceat(wR[order], starts, axis=0).reshape(len(ukey), 9)
    p.a = np.add.reduceat(kr[order], starts, axis=0)
    return p


def count_components(prob: Problem) -> int:
    """Connected components of the merged camera x timestep graph (vectorised min-label propagation with
    pointer jumping).  More than one means more than three near-null eigenvectors: the reference then returns an
    arbitrary mixture of the components' gauges without any diagnostic ('disconnected graph
    silently yields garbage'); the drop-in API warns."""
    row_ptr, col = prob.host_csr()
    C, col, row_ptr = prob.n_cam, col.astype(np.int64), row_ptr.astype(np.int64)
    rows = np.repeat(np.arange(prob.n_time), np.diff(row_ptr))
    order = np.argsort(col, kind="stable")                          # edges grouped by camera
    rows_by_cam = rows[order]
    cam_start = np.searchsorted(col[order], np.arange(C))
    lab = np.arange(C)
    while True:
        lab_t = np.minimum.reduceat(lab[col], row_ptr[:-1])         # every timestep row has >= 1 edge
        new = np.minimum(lab, np.minimum.reduceat(lab_t[rows_by_cam], cam_start))
        new = new[new]
        if np.array_equal(new, lab):
            break
        lab = new
    return int(len(np.unique(lab)))


def bnorm2(prob: Problem, Rc: np.ndarray, Rt: np.ndarray) -> float:
    """|b|^2 of the reference's stacked measurement vector (bipgo.py:454-461) for world<-node
    rotations Rc [C,3,3], Rt [T,3,3]:  b_e = k_t (R_c t~_e + R_t R_root^T R_m tau_m)."""
    be = np.einsum("eij,ej->ei", Rc[prob.src_cam], prob.src_t) + np.einsum("eij,ej->ei", Rt[prob.src_time], prob.src_qtau)
    return float(np.sum((prob.src_kt[:, None] * be) ** 2))


class _InvertedEdge(dict):
    """Value dict of a re-keyed object-mode edge as the reference builds it (bipgo.py:526-531): ``corners``, ``reprojected_err``
    and ``im_filename`` of the source edge and ``pose`` = the INVERTED pose.  The inverse is formed only if a callable asks for
    it (the notebook's callables look at corners and reprojection error, main.ipynb:75-77): the solver's own use of the
    inverted poses is batched (flatten_object)."""
    __slots__ = ("_src_pose",)

    def __missing__(self, key):
        if key != "pose":
            raise KeyError(key)
        p = self._src_pose.inv()
        dict.__setitem__(self, "pose", p)
        return p

    def get(self, key, default=None):
        try:
            return self[key]
        except KeyError:
            return default

    def __contains__(self, key):
        return key == "pose" or dict.__contains__(self, key)

    def _whole(self):
        self["pose"]
        return self

    def keys(self):
        return dict.keys(self._whole())

    def values(self):
        return dict.values(self._whole())

    def items(self):
        return dict.items(self._whole())

    def __iter__(self):
        return dict.__iter__(self._whole())

    def __len__(self):
        return dict.__len__(self._whole())


_F32 = np.zeros(0, dtype=np.float32)            # stands for "a float32 rotation" in f32_product_mask


def inverted_poses(Rs, ts):
    """The rotations / translations of ``pose.inv()`` for a list of poses (reference geometry.py:235-243, as
    vican_amd.geometry.SE3.inv), all at once and to the bit: the inverse is assembled in a float32 4x4 -
    R' = float32(R^T), t' = float32((-R^T) t) with the product formed in the dtype of R and t - and handed on as float32
    views.  np.matmul over the stack takes the same per-item path as the 3x3 @ 3 product of a single pose
    (checked bit for bit, float64 and float32 inputs).  Returns float64 arrays ([n,3,3], [n,3]), or None when the inputs are
    not uniformly typed 3x3 / 3 arrays (the caller then inverts pose by pose)."""
    import operator
    try:
        dt = operator.attrgetter("dtype")
        if len(set(map(dt, Rs))) != 1 or len(set(map(dt, ts))) != 1:
            return None
        R, t = np.array(Rs), np.array(ts)                          # (ragged shapes raise)
    except (AttributeError, ValueError, TypeError):
        return None
    if R.shape[1:] != (3, 3) or t.shape[1:] != (3,) or R.dtype.kind != "f" or t.dtype.kind != "f":
        return None
    Rt = np.swapaxes(R, 1, 2)
    ti = np.matmul(-Rt, t[:, :, None])[:, :, 0]
    return Rt.astype(np.float32).astype(np.float64), ti.astype(np.float32).astype(np.float64)


def flatten_object(src_edges, noise_model_r, noise_model_t, edge_filter, dtype=np.float32, merge=None):
    """Front-end of ``object_bipartite_se3sync`` (bipgo.py:523-541) in ONE pass over the edge dict: markers take the camera
    role, every pose is inverted, the numerically smallest marker id is the root and its constraint the identity.  Same
    callables on the same value dicts (`_InvertedEdge`), same float32 rounding of the inverted poses - but the inversion is
    one batched NumPy expression instead of a Python call and a dict copy per edge (28.7 of the 31.6 ms of a cube_calib-sized
    call, profiles/r03_api_time.txt).  Returns (root, Problem)."""
    from .geometry import SE3
    rows, seen = [], set()
    add, see, new = rows.append, seen.add, _InvertedEdge
    for key, val in src_edges.items():                             # the only per-edge Python loop
        ts, mid = key[1].split("_")
        see(mid)
        pose = val["pose"]
        e = new(corners=val["corners"], reprojected_err=val["reprojected_err"], im_filename=val["im_filename"])
        e._src_pose = pose
        if edge_filter(e):
            add((mid, ts, pose, noise_model_r(e), noise_model_t(e)))
    root = str(min(map(int, seen)))                                 # bipgo.py:524 (numeric min over ALL source edges)
    if not rows:
        raise ValueError("no edge passes edge_filter")
    cams, times, poses, kr, kt = zip(*rows)
    Rs, ts_ = [p.R() for p in poses], [p.t() for p in poses]
    inv = inverted_poses(Rs, ts_)
    if inv is None:                                                 # mixed dtypes / foreign pose types: pose by pose
        inv_p = [p.inv() for p in poses]
        inv = _stack_f64([p.R() for p in inv_p], (3, 3)), _stack_f64([p.t() for p in inv_p], (3,))
    # (inverted poses are float32 whatever the source pose held: the weights' types decide the dtype of k_r * R)
    mask = f32_product_mask(kr, [_F32] * len(kr))
    return root, flatten_arrays(cams, times, np.full(len(cams), root), inv[0], inv[1], kr, kt, {root: SE3(pose=np.eye(4))}, dtype, merge,
                                kr_f32=mask)


def invert_object_edges(src_edges):
    """Re-key a moving-camera / static-object edge dict so that markers play the
    camera role (bipgo.py:523-531): (marker, "<t>_<root>") -> inverted pose.  The reference-shaped form (a Python call and
    a dict copy per edge); the drop-in call uses flatten_object, which is checked against this one to the bit."""
    root = str(min(int(k[1].split("_")[1]) for k in src_edges.keys()))           # numeric min
    edges = {}
    for k, v in src_edges.items():
        ts, mid = k[1].split("_")
        e = dict(v)
        e["pose"] = v["pose"].inv()
        edges[(mid, ts + "_" + root)] = e
    return root, edges
