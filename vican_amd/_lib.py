"""ctypes binding of the C ABI in ``include/vican_hip.h`` (``libvican_hip.so``).

The library is built IN-TREE by ``build_library()`` (``hipcc --offload-arch=gfx950``)
so it travels with the repository snapshot.  There is no CPU fallback: if the
shared object is missing, ``load()`` raises.
"""
from __future__ import annotations

import ctypes as C
import os
import re
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB_PATH = os.environ.get("VICAN_LIB") or os.path.join(CSRC, "libvican_hip.so")      # VICAN_LIB: diagnostic builds (tools/)
SOURCES = [os.path.join(CSRC, "vican_sweep.hip"), os.path.join(CSRC, "vican_kernels.hip"),
           os.path.join(CSRC, "vican_trans.hip"), os.path.join(CSRC, "vican_lsqr.hip"), os.path.join(CSRC, "vican_wtrans.hip"),
           os.path.join(CSRC, "vican_cgres.hip"), os.path.join(CSRC, "vican_merge.hip"),
           os.path.join(CSRC, "vican_facade.hip"), os.path.join(CSRC, "vican_facade_tiles.hip"), os.path.join(CSRC, "vican_comm.hip"), os.path.join(CSRC, "vican_tsweep.hip"), os.path.join(CSRC, "vican_tcg.hip")]
WSWEEP = os.path.join(CSRC, "vican_wsweep.hip")
HEADERS = [os.path.join(CSRC, "vican_common.h"), os.path.join(CSRC, "vican_sweep_common.h"), os.path.join(CSRC, "vican_cgw_impl.h"), os.path.join(CSRC, "vican_facade_impl.h"), WSWEEP]
FX_DOUBLES = 20
GRAM_WS_DOUBLES = 128 * 192 * 3     # VICAN_GRAM_WS_DOUBLES
SEED_MAX_N = 16384                  # VICAN_SEED_MAX_N
INCLUDE = os.path.join(ROOT, "include")
# VICAN_ABI_VERSION of include/vican_hip.h - the one place the number lives; load() rejects a library built from other sources
def _header_abi_version():
    """None when the package is deployed without the repository's include/ directory: load() then trusts the library's own
    vican_abi_version() (a stale library cannot be told from a current one there, but importing the package must not fail)."""
    try:
        return int(re.search(r"#define\s+VICAN_ABI_VERSION\s+(\d+)", open(os.path.join(INCLUDE, "vican_hip.h")).read()).group(1))
    except (OSError, AttributeError):
        return None


ABI_VERSION = _header_abi_version()

ERR_ARG, ERR_LAUNCH, ERR_CAPACITY = -1, -2, -3          # VICAN_ERR_*
STORE_F32, STORE_F64 = 0, 1
LAYOUT_BLOCK, LAYOUT_WAVE = 0, 1
PAD_SLOT = 0xFFFFFFFF


class Graph(C.Structure):
    """Mirror of ``vican_graph_t``."""
    _fields_ = [
        ("n_cam", C.c_int32), ("n_time", C.c_int32), ("n_chunk", C.c_int32), ("slots", C.c_int32),
        ("max_rows", C.c_int32), ("storage", C.c_int32), ("block_threads", C.c_int32), ("n_wg", C.c_int32),
        ("n_copy", C.c_int32), ("wg_chunk_cap", C.c_int32), ("layout", C.c_int32), ("wg_waves", C.c_int32),
        ("stream_nt", C.c_int32), ("slot_order", C.c_int32),
        ("blk", C.c_void_p), ("idx", C.c_void_p), ("chunk_row0", C.c_void_p), ("idx16", C.c_void_p),
        ("w32", C.c_void_p), ("w32_src", C.c_void_p),
    ]


class Tile(C.Structure):
    """Mirror of ``vican_tile_t`` (vican_tiled_op)."""
    _fields_ = [("g", Graph), ("x", C.c_void_p), ("zpart", C.c_void_p), ("fx", C.c_void_p), ("ypart", C.c_void_p * 2)]


class CgTile(C.Structure):
    """Mirror of ``vican_cg_tile_t`` (vican_cg_sweep_tiles)."""
    _fields_ = [("g", Graph), ("w", C.c_void_p), ("p_c", C.c_void_p), ("acc_t", C.c_void_p), ("qc_part", C.c_void_p)]


# doubles first (17), then 4 int32: 152 bytes == 19 doubles
CG_STATE_DOUBLES = 19
CG_F = dict(rho=0, rho_prev=1, pq=2, alpha=3, beta=4, bnorm2=5, atol2=6, rr_cam=7, pq_time=8, rr_time=9,
            rmax_cam=10, rmax_time=11, pmax=12, qscale=13, qinv=14, wmax=15, pmax_time=16)
CG_I = dict(iter=34, done=35, first=36, lo_bits=37)     # int32 index into the same buffer viewed as int32
CG_PQ_SLICES, CG_RR_SLICES = 96, 512                   # VICAN_CG_PQ_SLICES / VICAN_CG_RR_SLICES (vican_cg_iter_comm)

# vican_lsqr_state_t: 28 doubles then 8 int32 (256 bytes = 32 doubles)
LSQR_STATE_DOUBLES = 32
LSQR_F = dict(alfa=0, beta=1, rhobar=2, phibar=3, anorm=4, ddnorm=5, xxnorm=6, z=7, cs2=8, sn2=9, c2=10, bnorm=11, atol=12, btol=13,
              ctol=14, coef=15, inv_alfa=16, t1=17, t2=18, rnorm=19, arnorm=20, acond=21, xnorm=22, qscale=23, qinv=24, smax=25, n_add=26)
LSQR_I = dict(itn=56, istop=57, done=58, iter_lim=59, lo_bits=60, update=61)     # int32 index into the same buffer viewed as int32

_vp, _i32, _i64, _f64 = C.c_void_p, C.c_int32, C.c_int64, C.c_double
_G = C.POINTER(Graph)

# name -> (restype, argtypes).  Must list EVERY symbol declared in include/vican_hip.h
# (tests/test_abi.py checks the header against this table and the built library).
PROTOTYPES = {
    "vican_last_error": (C.c_char_p, []),
    "vican_abi_version": (C.c_int, []),
    "vican_set_gate": (C.c_int, [_vp]),
    "vican_set_barrier_abort": (C.c_int, [_vp, _i64]),
    "vican_set_launch_events": (C.c_int, [_vp, _vp]),
    "vican_lanczos_coop_ws_doubles": (_i64, [_i32]),
    "vican_lanczos_cam_coop": (C.c_int, [_i32, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _f64, _vp, _vp, _i32, _vp, _vp, _i32, _vp]),
    "vican_lanczos_step_slabs": (C.c_int, [_G, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _f64, _vp, _i32, _vp]),
    "vican_jacobi_scale": (C.c_int, [_i32, _vp, _vp, _vp]),
    "vican_row_scale": (C.c_int, [_i32, _i32, _vp, _vp, _vp]),
    "vican_scale_weights": (C.c_int, [_G, _vp, _vp, _vp, _vp, _vp]),
    "vican_ritz": (C.c_int, [_vp, _i32, _i32, _i32, _i32, _f64, _f64, _f64, _f64, _vp, _vp, _vp, _vp]),
    "vican_plan_chunks": (C.c_int, [_i32, _vp, _i32, _i32, _vp, _i32]),
    "vican_sweep_lds_bytes": (_i64, [_i32, _i32, _i32, _i32]),
    "vican_lds_limit_bytes": (_i64, []),
    "vican_wsweep_lds_bytes": (_i64, [_i32, _i32, _i32, _i32, _i32]),
    "vican_cg_wsweep_lds_bytes": (_i64, [_i32, _i32, _i32, _i32]),
    "vican_max_rows_for": (_i32, [_i32, _i32, _i32]),
    "vican_edge_sums": (C.c_int, [_G, _vp, _i32, _f64, _vp, _vp, _vp, _vp]),
    "vican_block_norms": (C.c_int, [_G, _vp, _vp, _vp]),
    "vican_fx_finish": (C.c_int, [_vp, _f64, _f64, _i32, _vp]),
    "vican_fx_finish_multi": (C.c_int, [_vp, _vp, _i32, _f64, _i32, _vp]),
    "vican_duals_bound": (C.c_int, [_i32, _vp, _vp, _vp, _vp]),
    "vican_slab_reduce_fx": (C.c_int, [_vp, _i32, _i32, _i32, _f64, _vp, _vp, _vp, _vp]),
    "vican_pack_edges": (C.c_int, [_G, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vican_init_duals": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp]),
    "vican_scaled_identity": (C.c_int, [_i32, _vp, _vp, _vp]),
    "vican_block_op": (C.c_int, [_G, _vp, _vp, _vp, _vp, _vp]),
    "vican_dual_update": (C.c_int, [_G, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vican_slab_reduce": (C.c_int, [_vp, _i32, _i64, _vp, _vp]),
    "vican_polar_dual": (C.c_int, [_i32, _vp, _vp, _vp, _i32, _vp]),
    "vican_gauge_project": (C.c_int, [_i32, _vp, _vp, _vp]),
    "vican_lap_apply": (C.c_int, [_i32, _vp, _vp, _i32, _i32, _vp, _vp, _vp]),
    "vican_tall_gram": (C.c_int, [_i32, _vp, _i32, _i32, _vp, _vp, _vp, C.c_int64, _vp]),
    "vican_tall_update": (C.c_int, [_i32, _vp, _i32, _i32, _vp, _vp, _vp, _i32, _vp]),
    "vican_chol_qr3": (C.c_int, [_i32, _vp, _vp, _vp, _i32, _i32, _vp, _vp, _f64, _vp]),
    "vican_tall_combine": (C.c_int, [_i32, _vp, _i32, _i32, _vp, _vp, _vp]),
    "vican_rows_to_cols": (C.c_int, [_i32, _vp, _vp, _i32, _i32, _vp]),
    "vican_lanczos_cam_step": (C.c_int, [_i32, _vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f64, _vp, C.c_int64, _vp]),
    "vican_block_op_z": (C.c_int, [_G, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vican_dual_update_op": (C.c_int, [_G, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vican_lanczos_seed": (C.c_int, [_i32, _vp, _vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vican_right_solve3": (C.c_int, [_i32, _vp, _vp, _vp, _vp]),
    "vican_sum_apply3": (C.c_int, [_i64, _i32, _vp, _vp, _i32, _i64, _vp, _vp]),
    "vican_bip_scales": (C.c_int, [_vp, _f64, _f64, _i32, _vp]),
    "vican_bip_apply": (C.c_int, [_G, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vican_tile_rows": (C.c_int, [_G, _vp, _vp, _vp, _vp]),
    "vican_cg_sweep_tiles": (C.c_int, [_vp, C.c_int32, C.c_int32, _vp, _vp, _vp]),
    "vican_pack_idx16": (C.c_int, [_G, _vp, _vp]),
    "vican_pack_w32": (C.c_int, [_G, _vp, _vp, _vp, _vp]),
    "vican_plan_chunks_multi": (C.c_int, [C.c_int32, C.c_int32, _vp, C.c_int32, C.c_int32, _vp, C.c_int32]),
    "vican_plan_rows_multi": (C.c_int, [C.c_int32, C.c_int32, _vp, C.c_int32, C.c_int32, C.c_int32, _vp, _vp, C.c_int32]),
    "vican_tiled_op_lds_bytes": (C.c_int64, [C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "vican_tiled_op_sentinel": (C.c_int, [_vp, C.c_int64, _vp]),
    "vican_tiled_op": (C.c_int, [_vp, _vp, C.c_int32, C.c_int32, _vp, C.c_int32, _vp]),
    "vican_tiled_op_z": (C.c_int, [_vp, _vp, C.c_int32, C.c_int32, _vp, _vp, _vp, C.c_int32, _vp]),
    "vican_tile_cams": (C.c_int, [_G, _vp, _vp, _vp, _vp, _vp]),
    "vican_cg_iter_local": (C.c_int, [_G, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f64, _vp, _i32, _f64, _vp, _vp]),
    "vican_cg_iter_finish": (C.c_int, [_i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp]),
    "vican_cg_iter_fused": (C.c_int, [_G, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f64, _vp, _i32, _f64, _i32, _vp, _vp, _vp]),
    "vican_cg1_iter_local": (C.c_int, [_G, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _f64, _vp, _vp, _vp]),
    "vican_cg1_iter_finish": (C.c_int, [_i32, _i32, _i32, _f64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _vp,
                                        _vp, _vp]),
    "vican_trans_rhs": (C.c_int, [_G, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f64, _f64, _vp]),
    "vican_cg_init": (C.c_int, [_i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f64, _vp]),
    "vican_cg_begin": (C.c_int, [_i32, _vp, _vp, _f64, _vp, _i32, _f64, _vp, _vp]),
    "vican_cg_sweep": (C.c_int, [_G, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vican_cg_fold": (C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp, _vp]),
    "vican_cg_fold_tiles": (C.c_int, [_vp, _vp, _i32, _i32, _vp, _vp, _vp]),
    "vican_cg_update_pt": (C.c_int, [_i32, _vp, _vp, _vp, _vp]),
    "vican_cg_sweep_partial": (C.c_int, [_G, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vican_cg_combine_rows": (C.c_int, [_i32, _i32, _i64, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp]),
    "vican_cg_reduce_pq": (C.c_int, [_vp, _i32, _vp, _vp, _vp]),
    "vican_cg_cam_step": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vican_cg_time_step": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _vp, _i32, _vp, _vp]),
    "vican_cg_end": (C.c_int, [_vp, _i32, _vp, _vp]),
    "vican_cg_resident_lds_bytes": (_i64, [_i32, _i32, _i32, _i32]),
    "vican_cg_resident_ws_doubles": (_i64, [_i32, _i32]),
    "vican_cg_resident": (C.c_int, [_G, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f64, _i32, _f64, _f64, _i32, _vp, _vp]),
    "vican_merge_ws_bytes": (_i64, [_i64, _i32, _i32]),
    "vican_merge_edges": (C.c_int, [_i64, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp,
                                    _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vican_lsqr_init_u": (C.c_int, [_G, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vican_lsqr_update": (C.c_int, [_i64, _f64, _f64, _f64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vican_lsqr_step": (C.c_int, [_G, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vican_lsqr_nodes": (C.c_int, [_i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vican_lsqr_scalars": (C.c_int, [_i32, _vp, _vp, _i32, _vp, _vp, _i32, _vp, _i32, _vp, _vp, _vp]),
    "vican_lsqr_update_st": (C.c_int, [_i64, _vp, _vp, _vp, _vp, _i32, _vp, _vp]),
    # collectives behind the C ABI (csrc/vican_comm.hip: RCCL through dlopen)
    "vican_comm_unique_id": (C.c_int, [_vp]),
    "vican_comm_create": (C.c_int, [_i32, _i32, _vp, C.POINTER(_vp)]),
    "vican_comm_create_local": (C.c_int, [_i32, _i32, C.POINTER(_vp)]),
    "vican_comm_peer_bytes": (_i64, [_i32, _i64]),
    "vican_comm_peer_export": (C.c_int, [_vp, _i64, _vp]),
    "vican_comm_peer_attach": (C.c_int, [_vp, _vp]),
    "vican_comm_peer_enable": (C.c_int, [_vp, _i32]),
    "vican_comm_peer_set_timeout": (C.c_int, [_vp, _i64]),
    "vican_comm_peer_status": (C.c_int, [_vp]),
    "vican_comm_allreduce_sum": (C.c_int, [_vp, _vp, _i64, _vp]),
    "vican_comm_destroy": (C.c_int, [_vp]),
    "vican_block_op_z_comm": (C.c_int, [_G, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "vican_cg_iter_comm": (C.c_int, [_G, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _f64, _vp, _i32, _f64, _i32, _vp, _vp, _vp]),
    # the four-call boundary (csrc/vican_facade.hip)
    "vican_plan_create": (C.c_int, [_i32, _i32, _i64, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, C.POINTER(_vp)]),
    "vican_plan_describe": (C.c_int, [_vp, _G]),
    "vican_plan_set_comm": (C.c_int, [_vp, _vp, _vp]),
    "vican_solve_rot": (C.c_int, [_vp, _i32, _f64, _vp, _vp, _vp, _vp]),
    "vican_solve_trans": (C.c_int, [_vp, _vp, _vp, _f64, _i64, _vp, _vp, _vp, _vp]),
    "vican_solve_trans_lsqr": (C.c_int, [_vp, _vp, _vp, _f64, _f64, _f64, _f64, _i64, _vp, _vp, _vp, _vp]),
    "vican_plan_destroy": (C.c_int, [_vp]),
}


class SolveInfo(C.Structure):
    """Mirror of ``vican_solve_info_t``."""
    _fields_ = [("iterations", C.c_int32), ("sweeps", C.c_int32), ("lanczos_steps", C.c_int32), ("restarts", C.c_int32),
                ("evals", C.c_double * 5), ("eig_resid", C.c_double), ("cg_iters", C.c_int32), ("cg_converged", C.c_int32),
                ("cg_relres", C.c_double)]

class LsqrInfo(C.Structure):
    """Mirror of ``vican_lsqr_info_t``."""
    _fields_ = [("itn", C.c_int32), ("istop", C.c_int32), ("rnorm", C.c_double), ("arnorm", C.c_double), ("anorm", C.c_double),
                ("acond", C.c_double), ("xnorm", C.c_double)]


# include/vican_hip_test.h: diagnostics / cross-check entry points, not part of the boundary
TEST_PROTOTYPES = {
    "vican_comm_force_enqueue": (C.c_int, [_vp, _i32]),
    "vican_comm_peer_inject_fault": (C.c_int, [_vp]),
    "vican_facade_set_tile_cams": (C.c_int, [_i32]),
    "vican_test_occupy": (C.c_int, [_i32, _i32, _i32, _i64, _vp]),
    "vican_lsqr_u_step": (C.c_int, [_G, _vp, _vp, _vp, _f64, _vp, _vp, _vp, _vp]),
    "vican_lsqr_v_step": (C.c_int, [_G, _vp, _vp, _f64, _f64, _vp, _vp, _vp, _vp, _f64, _f64, C.POINTER(C.c_double), _vp]),
    "vican_lsqr_cam_v": (C.c_int, [_i32, _vp, _f64, _vp, _vp, _vp]),
}

_lib = None


class VicanError(RuntimeError):
    pass


def hipcc_path() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise VicanError("hipcc not found - cannot build libvican_hip.so")


def build_library(force: bool = False, verbose: bool = False, out: str = None, extra_flags=()) -> str:
    """Compile the HIP sources for gfx950 into ``vican_amd/csrc/libvican_hip.so`` (or ``out``).

    One hipcc per translation unit, in parallel, then a link step.  Objects are cached OUTSIDE the tree
    (``$VICAN_BUILD_CACHE`` or ``<tmp>/vican_amd_build_cache_<uid>/<sha1 of flags + source + headers>.o``, mode 0700 - only the linked
    library travels with a repository snapshot), so diagnostic variants (``extra_flags``) and rebuilds after a one-file
    edit only recompile what changed; ``force`` ignores the cache."""
    out = out or LIB_PATH
    deps = SOURCES + HEADERS + [os.path.join(INCLUDE, "vican_hip.h"), os.path.join(INCLUDE, "vican_hip_test.h")]
    if not force and not extra_flags and os.path.exists(out):
        if os.path.getmtime(out) >= max(os.path.getmtime(p) for p in deps):
            return out
    import concurrent.futures
    import hashlib
    import threading
    # --offload-compress: the ~350 kernel instantiations are 34 MB of gfx950 code uncompressed, 4.9 MB compressed (the runtime
    # inflates the bundle when the library is loaded: 0.04 s on the MI355X box) - the library travels with every snapshot
    flags_all = ["--offload-arch=gfx950", "--offload-compress", "-O3", "-munsafe-fp-atomics", "-fPIC", "-Wno-unused-value", "-I", INCLUDE, "-I", CSRC,
                 *os.environ.get("VICAN_CFLAGS", "").split(), *extra_flags]
    # the two sweep files are compiled once per sweep mode plus once without their hot kernel (see their headers):
    # ~190 + ~160 kernel instantiations
    sweep = SOURCES[0]
    jobs = [(sweep, ["-DVICAN_SWEEP_SPLIT", "-DVICAN_SWEEP_PART=%d" % m]) for m in range(4)]
    jobs += [(sweep, ["-DVICAN_SWEEP_SPLIT"])]
    jobs += [(WSWEEP, ["-DVICAN_WSWEEP_SPLIT", "-DVICAN_WSWEEP_PART=%d" % m]) for m in (0, 1, 3, 4)]
    jobs += [(WSWEEP, ["-DVICAN_WSWEEP_SPLIT"])]
    jobs += [(src, []) for src in SOURCES[1:]]
    import tempfile
    # per-user object cache, private to its owner: the objects are linked into the library this process loads, so a
    # directory another user could have created (or can write to) must never be trusted
    cache = os.environ.get("VICAN_BUILD_CACHE") or os.path.join(tempfile.gettempdir(), "vican_amd_build_cache_%d" % os.getuid())
    os.makedirs(cache, mode=0o700, exist_ok=True)
    st_c = os.stat(cache)
    if st_c.st_uid != os.getuid() or (st_c.st_mode & 0o022):
        raise VicanError("build cache %s is not a private directory of uid %d (owner %d, mode %o): refusing to link objects "
                         "from it; set VICAN_BUILD_CACHE to a directory of your own" % (cache, os.getuid(), st_c.st_uid, st_c.st_mode & 0o777))
    hdr = b"".join(open(p, "rb").read() for p in sorted(set(HEADERS) - {WSWEEP}) + [os.path.join(INCLUDE, "vican_hip.h")])
    # the compiler is part of the key: a toolchain upgrade must not reuse objects
    tool = subprocess.run([hipcc_path(), "--version"], capture_output=True).stdout

    def compile_one(job):
        src, extra = job
        text = open(src, "rb").read()
        # a -DNAME flag that neither this file nor a header mentions cannot change its object: variants share objects
        flags = [f for f in flags_all if not (f.startswith("-D") and f[2:].split("=")[0].encode() not in text + hdr)]
        key = hashlib.sha1(" ".join(flags + extra).encode() + text + hdr + tool).hexdigest()
        obj = os.path.join(cache, key + ".o")
        if force or not os.path.exists(obj):
            # ranks started together may build the same key at once: every process writes its own temporary file
            tmp = "%s.%d.%d.tmp" % (obj, os.getpid(), threading.get_ident())
            cmd = [hipcc_path(), *flags, *extra, "-c", src, "-o", tmp]
            if verbose:
                print(" ".join(cmd))
            res = subprocess.run(cmd, capture_output=True, text=True)
            if res.returncode != 0:
                if os.path.exists(tmp):
                    os.remove(tmp)
                raise VicanError("hipcc failed:\n" + res.stdout + res.stderr)
            os.replace(tmp, obj)
        return obj

    with concurrent.futures.ThreadPoolExecutor(max_workers=min(os.cpu_count() or 4, len(jobs))) as pool:
        objs = list(pool.map(compile_one, jobs))
    link_tmp = "%s.%d.tmp" % (out, os.getpid())
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-fPIC", "-shared", *objs, "-o", link_tmp]
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        if os.path.exists(link_tmp):
            os.remove(link_tmp)
        raise VicanError("hipcc link failed:\n" + res.stdout + res.stderr)
    os.replace(link_tmp, out)
    # keep the cache bounded: drop objects not used by the last few builds
    stale = sorted((os.path.getatime(os.path.join(cache, f)), f) for f in os.listdir(cache) if f.endswith(".o"))
    for _, f in stale[:-80]:
        os.remove(os.path.join(cache, f))
    return out


FASTPATH_SRC = os.path.join(CSRC, "vican_fastpath.c")
FASTPATH_SO = os.path.join(CSRC, "_vican_fastpath.so")
_fastpath = False            # False: not tried yet; None: unavailable


def build_fastpath(force: bool = False) -> str:
    """Compile the C passes of the drop-in front-end (csrc/vican_fastpath.c: CPython + NumPy C API) with gcc, in-tree.
    Host-side plumbing: the front-end runs without it (vican_amd.frontend falls back to NumPy / list comprehensions)."""
    import sysconfig
    import numpy as np
    if not force and os.path.exists(FASTPATH_SO) and os.path.getmtime(FASTPATH_SO) >= os.path.getmtime(FASTPATH_SRC):
        return FASTPATH_SO
    cc = shutil.which("gcc") or shutil.which("cc")
    if cc is None:
        raise VicanError("no C compiler for vican_fastpath.c")
    tmp = FASTPATH_SO + ".tmp.%d" % os.getpid()
    cmd = [cc, "-O2", "-fPIC", "-shared", "-I" + sysconfig.get_paths()["include"], "-I" + np.get_include(), FASTPATH_SRC, "-o", tmp]
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise VicanError("building vican_fastpath.c failed:\n" + res.stdout + res.stderr)
    os.replace(tmp, FASTPATH_SO)
    # the module has no ABI tag in its name: record what it was built for, fastpath() refuses anything else
    with open(FASTPATH_SO + ".stamp", "w") as f:
        f.write(_fastpath_stamp())
    return FASTPATH_SO


def _fastpath_stamp():
    import sys
    import numpy as np
    return "cpython-%d.%d numpy-%s" % (sys.version_info[0], sys.version_info[1], np.__version__.split(".")[0])


def fastpath():
    """The `_vican_fastpath` extension module, or None (not built, built for another interpreter / NumPy, VICAN_FASTPATH=0)."""
    global _fastpath
    if _fastpath is False:
        _fastpath = None
        if os.environ.get("VICAN_FASTPATH", "1") != "0" and os.path.exists(FASTPATH_SO):
            try:
                # (built for another interpreter / NumPy major version - it travels with repository snapshots: not loaded)
                if open(FASTPATH_SO + ".stamp").read() != _fastpath_stamp():
                    return _fastpath
                import importlib.machinery
                import importlib.util
                loader = importlib.machinery.ExtensionFileLoader("_vican_fastpath", FASTPATH_SO)
                spec = importlib.util.spec_from_loader("_vican_fastpath", loader)
                mod = importlib.util.module_from_spec(spec)
                loader.exec_module(mod)
                _fastpath = mod
            except Exception:                 # (ImportError, an ABI mismatch of the NumPy C API ...): the Python path serves
                _fastpath = None
    return _fastpath


def load():
    """Load the shared library and attach prototypes (raises if it is missing)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise VicanError(
            "libvican_hip.so is missing (%s). Build it with `python __graft_entry__.py build` "
            "or vican_amd._lib.build_library(); there is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    lib.vican_abi_version.restype = C.c_int
    global ABI_VERSION
    if ABI_VERSION is None:
        ABI_VERSION = int(lib.vican_abi_version())
    if lib.vican_abi_version() != ABI_VERSION:
        raise VicanError("%s is stale: it reports ABI %d, this package needs %d - rebuild with "
                         "`python __graft_entry__.py build`" % (LIB_PATH, lib.vican_abi_version(), ABI_VERSION))
    for name, (res, args) in {**PROTOTYPES, **TEST_PROTOTYPES}.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> int:
    if rc < 0:
        msg = load().vican_last_error()
        raise VicanError("%s failed (%d): %s" % (what or "vican call", rc, msg.decode() if msg else "?"))
    return rc
