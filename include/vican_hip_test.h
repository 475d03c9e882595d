/* vican_hip_test.h - entry points of libvican_hip.so that are NOT part of the boundary a maintainer binds
 * (include/vican_hip.h): a diagnostic used by the tests of the bounded grid barriers, and the round-2 two-pass LSQR
 * steps with host-side scalars, superseded by the fused device-resident step (vican_lsqr_step) and kept only so that
 * tests can cross-check the two formulations (tests/test_kernels_gpu.py::test_lsqr_kernels).  Same conventions as
 * vican_hip.h: int status, raw device pointers, caller-supplied stream.                                              */
#ifndef VICAN_HIP_TEST_H
#define VICAN_HIP_TEST_H
#include "vican_hip.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Diagnostic: n_wg workgroups of `threads` threads with lds_bytes of LDS each that do nothing but stay resident for
 * `microseconds` - stands in for "something else occupies the compute units" in the tests of the bounded barriers.  */
int vican_test_occupy(int32_t n_wg, int32_t threads, int32_t lds_bytes, int64_t microseconds, void* stream);

/* A ONE-rank communicator normally enqueues nothing in vican_comm_allreduce_sum (the sum over one rank is the identity).
 * on != 0: it calls ncclAllReduce anyway - as ncclAvg, i.e. sum x 1/1, because RCCL elides an in-place ncclSum on one rank
 * without launching anything - so that a 1-GPU box executes and times RCCL's kernel on the launch stream behind the kernels
 * that produced the message (tests/test_comm_gpu.py, tools/rccl_onerank.py).  The buffer keeps its bits.                */
int vican_comm_force_enqueue(vican_comm_t* comm, int32_t on);

/* Counts one timed-out wait of the peer exchange without there having been one (vican_comm_peer_status > 0 afterwards): the tests
 * walk the recovery path - the whole group falls back to RCCL / torch.distributed - on hardware where the exchange works.       */
int vican_comm_peer_inject_fault(vican_comm_t* comm);

/* Cameras per tile of the plans vican_plan_create makes from now on (1..1024; default 1024 = one LDS table of the sweeps): the tests
 * force the camera-tiled schedule (csrc/vican_facade_tiles.hip) on golden cases of a few dozen cameras.  Process-wide.           */
int vican_facade_set_tile_cams(int32_t n_cam_per_tile);

/* ---- LSQR, two-pass form (cross-checks; reference bipgo.py:479-480) -------------------------------------------- */
/* u <- s (v_t - v_c) - coef * u ;  *nrm2_out = |u|^2 */
int vican_lsqr_u_step(const vican_graph_t* g, const double* sw, const double* v_c, const double* v_t,
                      double coef, double* u, double* part, double* nrm2_out, void* stream);
/* v_t <- sum_c s u inv_beta - beta v_t (in place), *nrm2_t_out = |v_t|^2; camera side as
 * fixed-point slabs vc_part[n_wg][3][C] of -sum_t s u inv_beta (fold with
 * vican_slab_reduce_fx, scale = *inv_out, then vican_lsqr_cam_v).  smax >= max sqrt(w_e).      */
int vican_lsqr_v_step(const vican_graph_t* g, const double* sw, const double* u, double inv_beta,
                      double beta, double* v_t, void* vc_part, double* part, double* nrm2_t_out,
                      double smax, double n_add, double* inv_out, void* stream);
/* v_c <- acc - beta v_c ; *nrm2_out = |v_c|^2 */
int vican_lsqr_cam_v(int32_t n_cam, const double* acc, double beta, double* v_c, double* nrm2_out,
                     void* stream);

#ifdef __cplusplus
}
#endif
#endif
