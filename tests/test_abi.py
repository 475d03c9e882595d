"""The C-ABI library loads on a CPU-only box and exports every symbol the header declares;
host-side planning (no GPU needed) behaves."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from vican_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    _lib.build_library()
    return _lib.load()


def header_symbols(name="vican_hip.h"):
    txt = open(os.path.join(ROOT, "include", name)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(vican_[a-z0-9_]+)\s*\(", txt)))


def test_every_header_symbol_is_exported_and_bound(lib):
    syms = header_symbols()
    assert len(syms) >= 25
    for s in syms:
        assert hasattr(lib, s), "library does not export %s" % s
        assert s in _lib.PROTOTYPES, "ctypes table lacks %s" % s
    assert sorted(_lib.PROTOTYPES) == syms, "ctypes table and header disagree"
    assert lib.vican_abi_version() == _lib.ABI_VERSION
    # diagnostics and superseded cross-check entry points live in their own header, outside the boundary
    tsyms = header_symbols("vican_hip_test.h")
    assert sorted(_lib.TEST_PROTOTYPES) == tsyms and not set(tsyms) & set(syms)
    assert all(hasattr(lib, s) for s in tsyms)


def test_struct_sizes():
    assert C.sizeof(_lib.Graph) == 14 * 4 + 6 * 8      # 14 int32 + 6 pointers (blk, idx, chunk_row0, idx16, w32, w32_src)
    assert C.sizeof(_lib.Tile) == C.sizeof(_lib.Graph) + 5 * 8   # vican_tile_t: the graph + x, zpart, fx, ypart[2]
    assert _lib.CG_STATE_DOUBLES * 8 == 17 * 8 + 4 * 4


def header_struct(name, header="vican_hip.h"):
    """[(field, C type string)] of `typedef struct ... { ... } <name>;` in the header, comments stripped, in order."""
    txt = open(os.path.join(ROOT, "include", header)).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    body = re.search(r"typedef\s+struct\s*\w*\s*\{([^}]*)\}\s*%s\s*;" % re.escape(name), txt, flags=re.S).group(1)
    out = []
    for decl in body.split(";"):
        decl = " ".join(decl.split())
        if not decl:
            continue
        m = re.match(r"(.*?[\s\*])(\w+(?:\s*,\s*\w+)*)$", decl)
        ctype, names = m.group(1).strip(), [n.strip() for n in m.group(2).split(",")]
        out += [(n, ctype) for n in names]
    return out


def test_graph_struct_in_the_header_the_ctypes_mirror_and_the_documented_stub_agree():
    """vican_graph_t three times: the header (the truth), vican_amd/_lib.Graph, and the `class Graph` a maintainer copies out
    of INTEGRATION.md - rounds 3 and 4 each shipped that stub one field short (a caller's struct 8 bytes smaller than the one
    the library reads).  The stub's class statement is EXECUTED here and compared field by field."""
    fields = header_struct("vican_graph_t")
    kind = lambda ctype: C.c_void_p if "*" in ctype else {"int32_t": C.c_int32, "int64_t": C.c_int64, "double": C.c_double}[ctype.replace("const ", "")]
    want = [(n, kind(t)) for n, t in fields]
    assert [(n, t) for n, t in _lib.Graph._fields_] == want
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", md, flags=re.S)
    src = [b for b in blocks if "class Graph(C.Structure)" in b]
    assert len(src) == 1
    stmt = re.search(r"(class Graph\(C\.Structure\):.*?)\n(?=\S)", src[0], flags=re.S).group(1)
    ns = {"C": C}
    exec(stmt, ns)                                             # noqa: S102  (our own documentation)
    stub = ns["Graph"]
    assert [(n, t) for n, t in stub._fields_] == want, "INTEGRATION.md's Graph stub drifted from include/vican_hip.h"
    assert C.sizeof(stub) == C.sizeof(_lib.Graph)
    # every entry point the document names exists in one of the two headers; every argtypes list it shows has the bound arity
    named = set(re.findall(r"\bvican_[a-z0-9_]+\b", md))
    known = set(header_symbols()) | set(header_symbols("vican_hip_test.h"))
    types = {"vican_graph_t", "vican_lsqr_state_t", "vican_amd", "vican_hip", "vican_facade", "vican_facade_tiles", "vican_plan_t", "vican_solve_info_t", "vican_comm_t", "vican_cg_state_t"}
    patterns = {n for n in named if n.endswith("_")}          # prefixes like vican_comm_ in prose
    unknown = {n for n in named - known - types - patterns if not any(k.startswith(n) for k in known)}
    assert not unknown, "INTEGRATION.md names entry points the headers do not declare: %s" % sorted(unknown)
    for name, args in re.findall(r"lib\.(vican_\w+)\.argtypes = \[(.*?)\]", md, flags=re.S):
        assert len([a for a in args.split(",") if a.strip()]) == len(_lib.PROTOTYPES[name][1]), name


def plan(lib, rp, slots, max_rows):
    rp = np.asarray(rp, dtype=np.int32)
    out = np.empty(len(rp) + 1, dtype=np.int32)
    n = lib.vican_plan_chunks(len(rp) - 1, C.c_void_p(rp.ctypes.data), slots, max_rows,
                              C.c_void_p(out.ctypes.data), len(out))
    return n, out[: n + 1] if n >= 0 else None


def test_plan_chunks_whole_rows_and_caps(lib):
    rng = np.random.default_rng(0)
    deg = rng.integers(0, 40, 500)
    rp = np.concatenate([[0], np.cumsum(deg)])
    n, c0 = plan(lib, rp, 64, 7)
    assert n > 0 and c0[0] == 0 and c0[-1] == 500 and np.all(np.diff(c0) > 0)
    for k in range(n):
        assert rp[c0[k + 1]] - rp[c0[k]] <= 64 and c0[k + 1] - c0[k] <= 7
        # greedy: the next row would not have fitted
        if c0[k + 1] < 500 and c0[k + 1] - c0[k] < 7:
            assert rp[c0[k + 1] + 1] - rp[c0[k]] > 64


def test_plan_chunks_edge_cases(lib):
    n, c0 = plan(lib, [0], 64, 8)                      # no rows
    assert n == 0 and list(c0) == [0]
    n, c0 = plan(lib, [0, 0, 0, 0], 64, 8)             # empty rows are legal
    assert n == 1 and list(c0) == [0, 3]
    n, _ = plan(lib, [0, 100], 64, 8)                  # a row longer than a chunk
    assert n == -3 and b"more edges" in lib.vican_last_error()
    assert lib.vican_plan_chunks(3, None, 64, 8, None, 0) == -1


def test_plan_chunks_multi_is_one_chunking_that_fits_every_tile(lib):
    """vican_plan_chunks_multi (the shared chunking of camera tiles, vican_tiled_op): whole rows, every tile's edges of a chunk
    within `slots`, greedy (the next row would not have fitted in some tile), one tile = vican_plan_chunks."""
    rng = np.random.default_rng(1)
    T, n_tile, slots, max_rows = 800, 3, 64, 9
    rps = [np.concatenate([[0], np.cumsum(rng.integers(0, 30, T))]).astype(np.int32) for _ in range(n_tile)]
    ptrs = (C.c_void_p * n_tile)(*[r.ctypes.data for r in rps])
    out = np.empty(T + 2, dtype=np.int32)
    n = lib.vican_plan_chunks_multi(T, n_tile, C.cast(ptrs, C.c_void_p), slots, max_rows, C.c_void_p(out.ctypes.data), len(out))
    assert n > 0
    c0 = out[: n + 1]
    assert c0[0] == 0 and c0[-1] == T and np.all(np.diff(c0) > 0) and np.diff(c0).max() <= max_rows
    for k in range(n):
        assert all(r[c0[k + 1]] - r[c0[k]] <= slots for r in rps)
        if c0[k + 1] < T and c0[k + 1] - c0[k] < max_rows:
            assert any(r[c0[k + 1] + 1] - r[c0[k]] > slots for r in rps)
    one = (C.c_void_p * 1)(rps[0].ctypes.data)
    out1 = np.empty(T + 2, dtype=np.int32)
    n1 = lib.vican_plan_chunks_multi(T, 1, C.cast(one, C.c_void_p), slots, max_rows, C.c_void_p(out1.ctypes.data), len(out1))
    n2, c2 = plan(lib, rps[0], slots, max_rows)
    assert n1 == n2 and list(out1[: n1 + 1]) == list(c2)
    big = np.array([0, 100], dtype=np.int32)                       # a row that does not fit one of the tiles
    small = np.array([0, 3], dtype=np.int32)
    two = (C.c_void_p * 2)(small.ctypes.data, big.ctypes.data)
    assert lib.vican_plan_chunks_multi(1, 2, C.cast(two, C.c_void_p), 64, 8, C.c_void_p(out.ctypes.data), len(out)) == -3


def test_plan_rows_multi_packs_rows_for_a_shared_chunking(lib):
    """vican_plan_rows_multi (rows in a better ORDER for the shared chunking of camera tiles): a permutation, rows of a chunk
    ascending, every tile's edges of a chunk within `slots`, at most max_rows rows - and fewer chunks than consecutive rows give
    where every row splits evenly over the tiles (the wide benchmark's shape: 4 tiles x 62.5 +- 7 edges per row)."""
    rng = np.random.default_rng(2)
    T, n_tile, slots, max_rows = 3000, 4, 256, 64
    cams = np.stack([rng.choice(4000, 250, replace=False) for _ in range(T)])
    rps = [np.concatenate([[0], np.cumsum(((cams >= 1000 * k) & (cams < 1000 * (k + 1))).sum(1))]).astype(np.int32) for k in range(n_tile)]
    ptrs = (C.c_void_p * n_tile)(*[r.ctypes.data for r in rps])
    perm, c0 = np.empty(T, dtype=np.int32), np.empty(T + 2, dtype=np.int32)
    n = lib.vican_plan_rows_multi(T, n_tile, C.cast(ptrs, C.c_void_p), slots, max_rows, 128, C.c_void_p(perm.ctypes.data), C.c_void_p(c0.ctypes.data), T + 2)
    assert n > 0, lib.vican_last_error()
    c0 = c0[: n + 1]
    assert sorted(perm.tolist()) == list(range(T)) and c0[0] == 0 and c0[-1] == T and np.all(np.diff(c0) > 0) and np.diff(c0).max() <= max_rows
    cnt = np.stack([np.diff(r) for r in rps], 1)[perm]                  # per-tile edges of the rows in the new order
    for k in range(n):
        assert np.all(np.diff(perm[c0[k]:c0[k + 1]]) > 0)
        assert cnt[c0[k]:c0[k + 1]].sum(0).max() <= slots
    out = np.empty(T + 2, dtype=np.int32)
    n_consecutive = lib.vican_plan_chunks_multi(T, n_tile, C.cast(ptrs, C.c_void_p), slots, max_rows, C.c_void_p(out.ctypes.data), len(out))
    assert n < 0.85 * n_consecutive, (n, n_consecutive)                 # (1.06-1.09 against 1.33 slots per edge)
    # window 1 = consecutive rows; one tile, rows that do not fit, bad arguments
    n1 = lib.vican_plan_rows_multi(T, n_tile, C.cast(ptrs, C.c_void_p), slots, max_rows, 1, C.c_void_p(perm.ctypes.data), C.c_void_p(out.ctypes.data), T + 2)
    assert n1 == n_consecutive and list(perm) == list(range(T))
    big = np.array([0, 300], dtype=np.int32)
    one = (C.c_void_p * 1)(big.ctypes.data)
    assert lib.vican_plan_rows_multi(1, 1, C.cast(one, C.c_void_p), 256, 8, 4, C.c_void_p(perm.ctypes.data), C.c_void_p(out.ctypes.data), 3) == -3
    assert lib.vican_plan_rows_multi(1, 1, None, 256, 8, 4, C.c_void_p(perm.ctypes.data), C.c_void_p(out.ctypes.data), 3) == -1


def test_lds_budget(lib):
    for storage in (_lib.STORE_F32, _lib.STORE_F64):
        for c in (3, 24, 340, 1000):
            for ncopy in (1, 8, 32):
                m = lib.vican_max_rows_for(c, storage, ncopy)
                assert m >= 1
                assert lib.vican_sweep_lds_bytes(c, m, storage, ncopy) <= lib.vican_lds_limit_bytes()
    assert lib.vican_max_rows_for(1025, _lib.STORE_F32, 1) <= 0       # LDS-resident camera planes: C <= 1024
    assert lib.vican_max_rows_for(1024, _lib.STORE_F32, 32) >= 8
    assert lib.vican_max_rows_for(1024, _lib.STORE_F64, 1) >= 1


def test_compute_calls_fail_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from vican_amd.bipgo import solve_problem
    from vican_amd.frontend import Problem
    with pytest.raises(_lib.VicanError):
        solve_problem(Problem(), 4, "conjugate_gradient")


def test_graft_entry_build_hook_runs():
    """The driver's ``build()`` hook: a full from-source compile for gfx950, the ABI check against the header's
    VICAN_ABI_VERSION, and the package imports (round 2 shipped a stale ``== 9`` here)."""
    import __graft_entry__ as entry
    entry.build()
    txt = open(os.path.join(ROOT, "include", "vican_hip.h")).read()
    assert int(re.search(r"#define\s+VICAN_ABI_VERSION\s+(\d+)", txt).group(1)) == _lib.ABI_VERSION
    assert _lib.load().vican_abi_version() == _lib.ABI_VERSION
