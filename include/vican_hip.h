/*
 * vican_hip.h - C ABI of the MI355X (gfx950) kernels behind
 *               vican.bipgo.bipartite_se3sync / object_bipartite_se3sync.
 *
 * The reference (gabmoreira/vican) is pure Python and has NO plugin / FFI
 * interface: its boundary is the Python signature (vican/bipgo.py:353-360,
 * 493-499).  This header is therefore the interface a maintainer would bind
 * with ctypes to replace the third-party native calls the reference's hot
 * path makes (SciPy CSR SpGEMM/SpMM, ARPACK eigs, LAPACK 3x3 svd, scipy cg);
 * each entry point cites the reference lines whose work it takes over.
 * INTEGRATION.md shows the ctypes stub.
 *
 * Conventions
 *   - plain pointers and sizes only; every `d_` / device pointer is memory the
 *     CALLER owns (PyTorch-ROCm tensors in the bundled host code);
 *   - every function that launches work takes a `stream` (a hipStream_t passed
 *     as void*) and is asynchronous on it; none allocates, frees or
 *     synchronises, so call sequences are graph-capturable;
 *   - return value 0 = OK, negative = error (vican_last_error() gives text);
 *   - a "3x3 block" is 9 consecutive scalars, row-major; camera-side vectors x, z are
 *     row-major [3C][3]; the Lanczos basis is column-major (see the spectral section).
 *
 * Edge layout ("CSR of 3x3 blocks", timestep-major, chunked):
 *   merged (camera,timestep) edges sorted by timestep row then camera are cut
 *   into chunks of whole rows holding at most `slots` edges and `max_rows`
 *   rows.  A chunk is padded to exactly `slots` entries and stored as
 *       blk  [n_chunk][9][slots]   block component planes (float or double)
 *       idx  [n_chunk][slots]      camera | (row - chunk_row0) << 16 ; 0xFFFFFFFF = padding
 *   so that lane l of a wavefront reads 16 contiguous bytes of every plane
 *   (fully coalesced dwordx4 loads) and the 4-byte index is the only
 *   per-edge metadata: algorithmic traffic = E*(9*s+4) bytes.
 *   Per-edge scalars/vectors of the translation stage use the same slot order.
 */
#ifndef VICAN_HIP_H
#define VICAN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VICAN_OK                 0
#define VICAN_ERR_ARG           -1
#define VICAN_ERR_LAUNCH        -2
#define VICAN_ERR_CAPACITY      -3   /* graph does not fit the LDS-resident kernel */

#define VICAN_STORE_F32 0
#define VICAN_STORE_F64 1

#define VICAN_PAD_SLOT 0xFFFFFFFFu

#define VICAN_LAYOUT_BLOCK 0
#define VICAN_LAYOUT_WAVE  1

/* Graph view: sizes + device pointers of the chunked edge layout. */
typedef struct vican_graph {
    int32_t n_cam;            /* C  (<= 65535) */
    int32_t n_time;           /* T  (local timestep rows of this rank) */
    int32_t n_chunk;
    int32_t slots;            /* edges per chunk = block_threads * edges_per_lane */
    int32_t max_rows;         /* max timestep rows in any chunk */
    int32_t storage;          /* VICAN_STORE_F32 | VICAN_STORE_F64 (type of blk) */
    int32_t block_threads;    /* 256, 512, 768 or 1024 */
    int32_t n_wg;             /* persistent workgroups per sweep (= number of partial slabs) */
    int32_t n_copy;           /* lane-striped copies of the per-row accumulators (power of 2, <= 32) */
    int32_t wg_chunk_cap;     /* most chunks one workgroup of a block sweep may take (dynamic tickets, see
                                 vican_block_op); >= ceil(n_chunk / n_wg); 0 = unlimited */
    int32_t layout;           /* VICAN_LAYOUT_BLOCK: a chunk is processed by a whole workgroup (slots = block_threads *
                                 edges_per_lane).  VICAN_LAYOUT_WAVE: a chunk is processed by ONE wavefront (slots = 64 *
                                 edges_per_lane, max_rows <= 64) and a workgroup of wg_waves wavefronts (block_threads =
                                 64 * wg_waves) shares the camera tables - the rotation sweeps (vican_block_op(_z),
                                 vican_dual_update(_op)), vican_cg_sweep / vican_cg_iter_local and vican_scale_weights; the other
                                 translation kernels (right-hand side, LSQR) and vican_bip_apply take block layouts */
    int32_t wg_waves;         /* wavefronts per workgroup of the wave layout: 4, 8 or 12 (0 in the block layout) */
    int32_t stream_nt;        /* 1: the sweeps read blk / idx with non-temporal loads (edge stream far larger than the
                                 256 MB Infinity Cache); 0: plain loads (cache-resident graphs are re-read from cache) */
    int32_t slot_order;       /* order of the edges inside a chunk, chosen at pack time (vican_pack_edges): 0 = bank-aware (lane l
                                 holds edges of cameras = l mod 32: conflict-free camera-side LDS accesses; dense rows), 1 = row-major
                                 (a lane holds consecutive edges of a row: few row flushes; short rows).  The order is part of the
                                 layout: a lane pre-sums its same-row terms in floating point, so results of the two orders agree
                                 to rounding, and each is bit-reproducible */
    const void*     blk;      /* [n_chunk][9][slots] */
    const uint32_t* idx;      /* [n_chunk][slots]    */
    const int32_t*  chunk_row0; /* [n_chunk+1] first row of each chunk */
    const uint16_t* idx16;    /* wave layout: [n_chunk][slots]  camera | (row - chunk_row0) << 10,  0xFFFF = padding (vican_pack_idx16) -
                                 a wave-layout chunk has <= 1024 cameras and <= 64 rows, so idx fits 16 bits; the edge sweeps of the
                                 wave layout (vican_block_op(_z), vican_dual_update(_op), vican_tile_rows / _cams, vican_tiled_op, the
                                 one-row CG product) stream these 2 bytes per edge instead of the 4 of idx and REQUIRE the array.
                                 (C = 1024 with 64 rows in a chunk would make camera 1023 of row 63 look like padding: such graphs
                                 are planned with <= 63 rows per chunk.)  NULL in the block layout */
    const float*    w32;      /* optional (NULL: none): the translation weights of `w32_src` as float32, [n_chunk][slots] in PLAIN slot
                                 order, for graphs whose weights are all exactly representable in float32 - every dtype=float32 problem:
                                 the reference's J^T J is accumulated in float32 there (bipgo.py:434-477).  The one-row CG product
                                 (wave layout, 4 edges per lane) then streams 6 instead of 10 bytes per edge, same bits */
    const double*   w32_src;  /* the float64 weight array (slot order of the layout) w32 mirrors: used only when the caller passes THIS
                                 array as `w` (scaled weights - vican_scale_weights - are another array and take the float64 stream) */
} vican_graph_t;

const char* vican_last_error(void);
#define VICAN_ABI_VERSION 31            /* the one place the number lives: the library returns it, vican_amd/_lib.py parses it */
int vican_abi_version(void);            /* VICAN_ABI_VERSION of the sources the library was built from */

/* Launch gate (state of the calling host thread).  While a non-NULL device pointer is set, the
 * kernels enqueued by vican_tall_combine, vican_gauge_project, vican_block_op(_z),
 * vican_slab_reduce_fx, vican_polar_dual, vican_dual_update(_op) and vican_fx_finish exit immediately
 * unless *gate == 1 WHEN THEY EXECUTE: the host enqueues the continuation of the primal-dual
 * iteration (bipgo.py:295-332) right behind vican_ritz without waiting for its verdict, and the
 * device cancels it if the eigen-solve has not converged.  All other entry points ignore the gate.
 * NULL (the default) disables gating.                                                        */
int vican_set_gate(const int32_t* gate);

/* Grid barriers of the cooperative kernels (vican_lanczos_cam_coop, vican_cg_resident, vican_tiled_op) - state of
 * the calling host thread.  Those kernels spin on a device counter, which terminates only if every workgroup of the grid
 * is resident.  (1) Their launchers refuse grids that could not be co-resident on an idle device (occupancy query x
 * compute units -> VICAN_ERR_CAPACITY; the caller uses the launch-sequence entry points instead).  (2) Every spin is
 * bounded: after timeout_us (<= 0: the default, 2 s) of waiting a workgroup writes 1 to *abort_word and leaves, every other
 * workgroup sees the word in its own spin and leaves too - a device shared with something that keeps part of the grid out
 * (another process's resident kernel, a CU mask) yields an aborted launch, not a hung queue.  abort_word: 32-bit word the
 * DEVICE can write and the host can read (pinned host memory: the host then polls it for free), zeroed by the caller;
 * NULL: unbounded spins (no way to report).  After an abort the outputs of that launch are undefined and its barrier words
 * must be zeroed before they are used again.                                                                       */
int vican_set_barrier_abort(uint32_t* abort_word, int64_t timeout_us);

/* Launch timer (state of the calling host thread).  The NEXT launch of an edge sweep (vican_block_op(_z),
 * vican_dual_update(_op), vican_bip_apply) binds the two HIP events (hipEvent_t passed as void*, created by the caller
 * with timing enabled) to its own dispatch - start / stop = begin / end of the kernel on the device - and clears them;
 * hipEventElapsedTime(start, stop) after the stream has passed the launch is the kernel's duration, the number a
 * rocprofv3 kernel trace reports.  (An event pair recorded AROUND a launch also times the 5-8 us the queue idles
 * between an event command and the next dispatch.)  NULLs cancel.                                            */
int vican_set_launch_events(void* start_event, void* stop_event);

/* ---- host-side planning (no GPU needed) ---------------------------------
 * Cut T rows (host row_ptr[T+1]) into chunks of whole rows with at most
 * `slots` edges and `max_rows` rows.  Writes first-row indices to
 * chunk_row0_out (capacity cap entries, needs n_chunk+1) and returns n_chunk,
 * or a negative error (a single row longer than `slots` -> VICAN_ERR_CAPACITY).
 * Replaces the COO->CSR assembly of bipgo.py:244-270 together with
 * vican_pack_edges.                                                         */
int vican_plan_chunks(int32_t n_time, const int32_t* row_ptr_host, int32_t slots,
                      int32_t max_rows, int32_t* chunk_row0_out, int32_t cap);

/* Bytes of dynamic LDS the operator sweep needs (x table + fixed-point z accumulators per
 * camera; duals, y sums, w and n_copy striped accumulators per row), and the LDS the device
 * offers per workgroup.                                                        */
int64_t vican_sweep_lds_bytes(int32_t n_cam, int32_t max_rows, int32_t storage, int32_t n_copy);
int64_t vican_lds_limit_bytes(void);
/* The same for the wave layout: camera tables shared by the workgroup + per wavefront the striped row accumulators,
 * the folded row sums and the phase-3 operand of its chunk.                                   */
int64_t vican_wsweep_lds_bytes(int32_t n_cam, int32_t max_rows, int32_t storage, int32_t n_copy, int32_t n_waves);
/* ... and of the CG Laplacian product on a wave-layout graph (vican_cg_sweep / vican_cg_iter_local).      */
int64_t vican_cg_wsweep_lds_bytes(int32_t n_cam, int32_t max_rows, int32_t n_copy, int32_t n_waves);
/* Largest max_rows for which every sweep kernel (operator, rhs, CG) fits in LDS;
 * <= 0 means the camera tables alone do not fit.                              */
int32_t vican_max_rows_for(int32_t n_cam, int32_t storage, int32_t n_copy);

/* ---- layout: CSR arrays -> chunked planes (device) -----------------------
 * row_ptr[T+1], col[E] int32; blk_csr [E][9], a_csr [E] in the storage type;
 * w_csr [E], u_csr [E][3], v_csr [E][3] double (may be NULL together with
 * their outputs).  Outputs: g->blk, g->idx (cast away const), a_out [n_chunk][slots],
 * w_out [n_chunk][slots], u_out / v_out [n_chunk][3][slots].  Inside a chunk the slots are
 * assigned bank-aware (lane l gets cameras = l mod 32, CSR order within a class), so the
 * order of edges in a chunk is NOT the CSR order; perm_ws is scratch for that assignment,
 * int32 [n_chunk][slots] (slot -> CSR edge or -1).
 * (bipgo.py:244-270 for the rotation arrays; 445-471 for the translation ones) */
int vican_pack_edges(const vican_graph_t* g, const int32_t* row_ptr, const int32_t* col,
                     const void* blk_csr, const void* a_csr, const double* w_csr,
                     const double* u_csr, const double* v_csr,
                     void* a_out, double* w_out, double* u_out, double* v_out,
                     int32_t* perm_ws, void* stream);

/* ---- graph constants (once per graph, at pack time) ------------------------
 * row_sum[t] = sum_c val_ct, cam_sum[c] = sum_t val_ct over THIS rank's rows (caller all-reduces
 * cam_sum across ranks).  val: [n_chunk][slots], float or double (val_is_f64); vmax >= max|val|.
 * Sums are 64-bit fixed point relative to vmax (order-independent => reproducible);
 * cam_ws: scratch of n_cam 8-byte words.  Used for d_t / camera degrees of the rotation weights
 * a (bipgo.py:271-276) and for the degrees of the translation Laplacian (weights w).        */
int vican_edge_sums(const vican_graph_t* g, const void* val, int32_t val_is_f64, double vmax,
                    double* row_sum, double* cam_sum, void* cam_ws, void* stream);

/* Fixed-point bookkeeping.  The sweeps accumulate in 64-bit fixed point (the fastest LDS
 * atomic on gfx950, and order-independent => bit-reproducible).  `fx` is a device buffer of
 * VICAN_FX_DOUBLES doubles: [0] y scale, [1] its inverse, [2] z scale, [3] its inverse,
 * [4] omega = max_t |lamT_inv[t]|_F * rnorm[t], [5] max block norm, [6] max_t rnorm[t],
 * [7] 2^-shift of the last vican_block_op (its scales are raised by 2^shift when the actual
 * max_c |x_c|_F is below x_bound), [8] x_bound, [9] z scale for a phase-3 operand bounded by x_bound itself
 * (vican_dual_update_op), [10] two 32-bit counters of the block sweeps' chunk scheduler (zero between
 * launches), [11] inverse of [9], [12..19] sixteen 32-bit counters of the wave-layout sweeps' scheduler (eight pool counters of
 * the work-stealing tail, one per group of workgroups with equal blockIdx mod 8, and the finished-workgroups count; zero
 * between launches).
 * vican_block_norms zeroes fx and fills rnorm[t] = sum_c |M_ct|_F, fx[5], fx[6];
 * vican_init_duals / vican_dual_update refresh fx[4]; vican_fx_finish turns the bounds into
 * power-of-two scales given |x_c|_F <= x_bound and n_add = max rows handled by one workgroup;
 * a contribution gets up to 47 bits; totals stay below 2^61 (f64 blocks) or 2^46 (f32 blocks, whose
 * accumulators hold raw magic-number bit patterns that are sign-extended from 48 bits at the end). */
#define VICAN_FX_DOUBLES 20
int vican_block_norms(const vican_graph_t* g, double* rnorm /*[T]*/, double* fx, void* stream);
int vican_fx_finish(double* fx, double x_bound, double n_add, int32_t storage, void* stream);
/* The same for the scale buffers of n <= 64 graphs over the SAME rows (camera tiles) in one launch: omega (fx[4], from
 * vican_duals_bound on fx[0]) is copied from the first buffer to the others; fx, n_add: HOST arrays of n entries.            */
int vican_fx_finish_multi(double* const* fx, const double* n_add, int32_t n, double x_bound, int32_t storage, void* stream);
/* fx[4] for caller-supplied duals (instead of vican_init_duals / vican_dual_update). */
int vican_duals_bound(int32_t n_time, const double* lamT_inv, const double* rnorm, double* fx,
                      void* stream);

/* ---- rotation stage ------------------------------------------------------ */

/* Initial duals (bipgo.py:271-276) from the stored row sums d_t of the rotation weights:
 * lamT_inv[t] = I/d_t; refreshes fx[4].  (lamC = cam_sum[c] * I via vican_scaled_identity.) */
int vican_init_duals(int32_t n_time, const double* row_sum_a, const double* rnorm,
                     double* lamT_inv /*[T][9]*/, double* fx, void* stream);
int vican_scaled_identity(int32_t n, const double* scale, double* out /*[n][9]*/, void* stream);

/* Fused connection-Laplacian operator  zpart[wg] = sum over the workgroup's
 * edges of  M_ct * lamT_inv[t] * (sum_c' M_c't^T x_c')   i.e. slabs of
 * P x = R~ Lambda_T^-1 R~^T x  without ever forming P (replaces the SpGEMM at
 * bipgo.py:273,334 and the SpMM at bipgo.py:300).  x: [3C][3] double with |x_c|_F <= the
 * x_bound given to vican_fx_finish; zpart: [n_wg][9][C] planes of 64-bit fixed point (scale fx[2]),
 * to be folded by vican_slab_reduce_fx.  Each block is read from HBM exactly once; products
 * are formed in the storage type (f32 for f32 blocks), sums are exact integers.
 * Chunks are handed to the workgroups through a ticket counter in fx[10] (exact integer sums: the
 * assignment cannot change the result), so launches on ONE graph must be serialised on one stream. */
int vican_block_op(const vican_graph_t* g, const double* lamT_inv, const double* x,
                   void* zpart, double* fx, void* stream);
/* Composite: vican_block_op + vican_slab_reduce_fx -> z [3C][3] (this rank's partial of P x). */
int vican_block_op_z(const vican_graph_t* g, const double* lamT_inv, const double* x,
                     void* zpart, double* fx, double* z, void* stream);

/* Non-eliminated solver (bipartite_so3sync, bipgo.py:91,119: `pairwise_r @ r` on all C+T nodes).
 * The symmetric operator R~ = [[0, M], [M^T, 0]] applied to [x_cam; x_time] in ONE pass over the blocks:
 *   y_time[t] = sum_c M_ct^T x_cam[c]   ([T][9] doubles, exact fixed-point row sums)
 *   z_cam[c]  = sum_t M_ct x_time[t]    ([C][9] doubles, slabs folded as in vican_block_op_z)
 * |x_cam[c]|_F, |x_time[t]|_F <= the x_bound given to vican_bip_scales, which prepares fx for this
 * entry point (phase-3 operand bounded by x_bound itself, i.e. omega = 1). */
int vican_bip_scales(double* fx, double x_bound, double n_add, int32_t storage, void* stream);
int vican_bip_apply(const vican_graph_t* g, const double* x_cam, const double* x_time, void* zpart,
                    double* fx, double* z_cam, double* y_time, void* stream);

/* Camera-tiled operator, one tile in the wave layout (graphs with more cameras than one LDS table holds; the reference
 * has no camera limit, bipgo.py:225-232).  rows: y_time[t] = sum_{c in tile} M_ct^T x_cam[c] ([T][9], every row written);
 * cams: z_cam[c] = sum_t M_ct w_time[t] for the tile's cameras ([C_tile][9]), w_time = Lambda_t^-1 (sum of all tiles'
 * y_time) formed by the caller (vican_sum_apply3), |w_time[t]|_F within the bound fx was finished for (vican_duals_bound
 * over ALL tiles' row norms + vican_fx_finish).  Both stream the tile's blocks once.  Block-layout tiles: vican_bip_apply. */
int vican_tile_rows(const vican_graph_t* g, const double* x_cam, double* y_time, double* fx, void* stream);
int vican_tile_cams(const vican_graph_t* g, const double* w_time, void* zpart, double* fx, double* z_cam, void* stream);

/* The tiled operator in ONE launch that reads every block once (csrc/vican_tsweep.hip).  Needs tiles that share their
 * chunking: vican_plan_chunks_multi plans it (chunk k = the same timestep rows in every tile; a row joins the open chunk
 * while every tile's edges still fit `slots`), each tile is then packed with that chunk_row0.  A workgroup is bound to a
 * tile; the wavefront that owns chunk k publishes its tile's share of the chunk's row sums, keeps the blocks in registers,
 * and an iteration later reads all tiles' shares, applies Lambda_t^-1 and forms the camera sums.  ypart[2]: the tile's
 * share buffers [T][9] of alternate launches (parity 0, 1, 0, ...), both filled by vican_tiled_op_sentinel before the first
 * launch and after an aborted one.  The grid (n_tile * n_wg_tile workgroups of 512 threads) must be co-resident
 * (VICAN_ERR_CAPACITY otherwise: use vican_tile_rows / vican_tile_cams); spins are bounded (vican_set_barrier_abort).
 * Afterwards z_cam of tile k = vican_slab_reduce_fx(zpart_k, n_wg_tile, C_k, 9, 1.0, fx_k + 3, fx_k + 7, ...).          */
typedef struct {
    vican_graph_t g;          /* the tile: wave layout, chunking shared with the other tiles */
    const double* x;          /* [C_tile][9]  the tile's slice of the operand */
    void* zpart;              /* >= n_wg_tile slabs [9][C_tile] of 64-bit words */
    double* fx;               /* the tile's scales (vican_duals_bound over ALL tiles' row norms + vican_fx_finish with
                                 n_add >= rows one workgroup of the fused launch handles) */
    double* ypart[2];         /* [T][9] each */
} vican_tile_t;
int vican_plan_chunks_multi(int32_t n_time, int32_t n_tile, const int32_t* const* row_ptrs_host, int32_t slots,
                            int32_t max_rows, int32_t* chunk_row0_out, int32_t cap);
/* The same with the rows in a BETTER ORDER (host): consecutive rows pad a shared chunking badly when every row splits evenly over
 * the tiles (a chunk takes a row only while every tile still has room: 1.33 slots per edge on 4 tiles x 62 edges per row).
 * One chunk at a time from a pool of the next `window` unassigned rows: the first of the pool opens the chunk, then the row
 * that leaves the tiles most evenly filled while two average rows still fit, then the largest row that fits (window = 1: the
 * consecutive chunking).  perm_out[new] = old row (rows of a chunk ascending), chunk_row0_out: boundaries in the new numbering
 * (cap >= chunks + 1 entries); returns the number of chunks.  The caller
 * builds the tiles from the rows in that order (the operator is a sum over rows: bipgo.py:300) and undoes it on what it returns
 * per row.                                                                                                                  */
int vican_plan_rows_multi(int32_t n_time, int32_t n_tile, const int32_t* const* row_ptrs_host, int32_t slots, int32_t max_rows,
                          int32_t window, int32_t* perm_out, int32_t* chunk_row0_out, int32_t cap);
int64_t vican_tiled_op_lds_bytes(int32_t n_cam, int32_t max_rows, int32_t storage, int32_t n_copy);
int vican_tiled_op_sentinel(double* ypart, int64_t n_doubles, void* stream);
int vican_tiled_op(const vican_tile_t* tiles_host, const vican_tile_t* tiles_dev, int32_t n_tile, int32_t n_wg_tile,
                   const double* lamT_inv, int32_t parity, void* stream);
/* The same with operand and result in the caller's arrays (x, z: [3C][3] doubles; tiles = consecutive camera ranges in descriptor
 * order, the descriptors' x fields are not read): the fused launch + ONE fold launch for all tiles.                            */
int vican_tiled_op_z(const vican_tile_t* tiles_host, const vican_tile_t* tiles_dev, int32_t n_tile, int32_t n_wg_tile,
                     const double* lamT_inv, const double* x, double* z, int32_t parity, void* stream);

/* idx16 of a packed wave-layout graph (see vican_graph_t.idx16): out [n_chunk][slots], then set g->idx16 = out. */
int vican_pack_idx16(const vican_graph_t* g, uint16_t* out, void* stream);
/* w32 of a packed wave-layout graph with 4 edges per lane (see vican_graph_t.w32): out [n_chunk][slots] floats from the packed
 * float64 weights w; *inexact (device word) = 1 if any weight is not a float32 value - then leave g->w32 NULL.  Else set
 * g->w32 = out and g->w32_src = w.                                                                                          */
int vican_pack_w32(const vican_graph_t* g, const double* w, float* out, int32_t* inexact, void* stream);

/* Timestep dual/primal update (bipgo.py:318-332): per row t,
 * Z_t = sum_c M_ct^T Rc_c, SVD -> Rt[t] = U diag(1,1,det UV^T) V^T,
 * lamT_inv[t] = U S^-1 U^T.  rc: [3C][3]; Rt, lamT_inv: [T][9].  Refreshes fx[4]
 * (call vican_fx_finish afterwards).                                            */
int vican_dual_update(const vican_graph_t* g, const double* rc, double* Rt,
                      double* lamT_inv, const double* rnorm, double* fx, void* stream);

/* out[i] = sum_s part[s][i], s < n_slab, i < n  (fixed order: bitwise reproducible). */
int vican_slab_reduce(const double* part, int32_t n_slab, int64_t n, double* out, void* stream);
/* out[c][q] = scale * (*pa) * (*pb) * sum_s part[s][q][c]: folds 64-bit fixed-point plane slabs
 * [n_slab][ncomp][n_cam] into a row-major camera vector [n_cam][ncomp] (exact integer sum).
 * pa / pb may be NULL (= 1).  vican_block_op slabs: ncomp 9, pa = fx+3, pb = fx+7;
 * (vican_trans_rhs and vican_cg_fold fold their double-word slabs themselves.) */
int vican_slab_reduce_fx(const void* part, int32_t n_slab, int32_t n_cam, int32_t ncomp, double scale,
                         const double* pa, const double* pb, double* out, void* stream);

/* vican_dual_update fused with the first operator application of the NEXT eigen-solve: the same outputs, and
 * z_raw [3C][3] = this rank's partial of  sum_t M_ct polar(Z_t)  (Z_t = sum_c M_ct^T R_c as above) - with the new
 * duals lamT_inv[t] = U S^-1 U^T the product lamT_inv[t] Z_t IS the polar factor U V^T, so z_raw = P_new R_c for the
 * warm-start block R_c of the next iteration (bipgo.py:285-292 after :318-334), obtained in the same pass over the
 * blocks.  The polar factors are formed inside the sweep by a Newton iteration (SVD fallback for singular rows).
 * vican_right_solve3 then applies the 3x3 normalisation beta^-1 of the start block (vican_chol_qr3). */
int vican_dual_update_op(const vican_graph_t* g, const double* Rc, double* Rt, double* lamT_inv,
                         const double* rnorm, double* fx, void* zpart, double* z_raw, void* stream);
/* Z[n][3] = X[n][3] * beta^-1, beta upper triangular [3][3] (zero pivot -> zero column); X may equal Z. */
int vican_right_solve3(int32_t n, const double* X, const double* beta, double* Z, void* stream);

/* Camera tiling (graphs with more cameras than the LDS-resident sweeps hold; host: device.TiledBackend).  Per-row partial
 * results of the camera tiles, B[k * b_stride + .] for k < n_b, are summed in tile order:
 *   A != NULL (width 9):  out[r] = A[r] * sum_k B_k[r]   (3x3 blocks: w_t = Lambda_T,t^-1 * sum_c M_ct^T x_c, bipgo.py:300)
 *   A == NULL:            out = sum_k B_k                 (n_rows x width doubles)
 * Honours the launch gate.                                                                                   */
int vican_sum_apply3(int64_t n_rows, int32_t width, const double* A, const double* B, int32_t n_b, int64_t b_stride,
                     double* out, void* stream);

/* Batched 3x3 polar / dual blocks (bipgo.py:306-312; geometry.py:189-190):
 * in [n][9] -> R_out [n][9] (nearest rotation, det fixed; may be NULL),
 * lam_out [n][9] (may be NULL): mode & 3 = 1: U S U^T, 2: U S^-1 U^T;
 * mode & 4: R_out = U V^T WITHOUT the det fix (bipgo.py:126-127).              */
int vican_polar_dual(int32_t n, const double* in, double* R_out, double* lam_out,
                     int32_t mode, void* stream);

/* Gauge fix + projection (bipgo.py:295-297): X <- X * inv(X[0:3,:]) then each
 * 3x3 camera block projected to SO(3).  x_in, x_out: [3C][3] (distinct buffers).    */
int vican_gauge_project(int32_t n_cam, const double* x_in, double* x_out, void* stream);

/* ---- spectral step: camera-side dense helpers for block Lanczos ---------
 * (together with vican_block_op these replace scipy eigs / ARPACK+SuperLU at
 * bipgo.py:288).  V is a column-major basis (column k at V + k*ld, ld >= n = 3C);
 * "block j" = columns 3j..3j+2; the work block R is column-major [3][n].                                                                */

/* aq = Lambda_C q - z  with q = basis columns col0..col0+2 (column-major basis V,
 * column k at V + k*ld, ld >= 3C), z row-major [3C][3] (the reduced sweep output),
 * aq column-major [3][3C].                                                    */
int vican_lap_apply(int32_t n_cam, const double* lamC, const double* V, int32_t ld,
                    int32_t col0, const double* z, double* aq, void* stream);
/* H[k][c] = V[:,k] . R[:,c]  for k < ka, c < 3  (R column-major [3][n]); fixed summation order.
 * One workgroup per basis column; with a scratch buffer ws of ws_doubles >= 128 * ka * 3 doubles
 * (VICAN_GRAM_WS_DOUBLES covers every ka) vectors of n >= 16384 rows are summed in row slices
 * by the whole chip and folded in slice order.  ws may be NULL.                          */
#define VICAN_GRAM_WS_DOUBLES (128 * 192 * 3)
int vican_tall_gram(int32_t n, const double* V, int32_t ld, int32_t ka, const double* R,
                    double* H, double* ws, int64_t ws_doubles, void* stream);
/* R -= V[:, :ka] H ; if H_out: H_out[ka][3] = H (accumulate=0) or += H (accumulate=1) */
int vican_tall_update(int32_t n, const double* V, int32_t ld, int32_t ka, const double* H,
                      double* R, double* H_out, int32_t accumulate, void* stream);
/* G = R^T R (3x3, from vican_tall_gram with V := R, ld := n): upper Cholesky
 * G = beta^T beta, Q = R beta^-1 -> basis columns col0..col0+2, beta_out[3][3],
 * and (if x_out) Q row-major [n][3] as the next sweep input.  A pivot <=
 * max(1e-28 trace(G), pivot_floor) (Krylov space exhausted) yields a zero column
 * and beta_jj = 0, which the host treats as breakdown.                           */
int vican_chol_qr3(int32_t n, const double* R, const double* G, double* V, int32_t ld,
                   int32_t col0, double* beta_out, double* x_out, double pivot_floor,
                   void* stream);
/* Start block of an eigen-solve as one launch (n <= VICAN_SEED_MAX_N rows): G = X0^T X0, upper Cholesky
 * G = beta^T beta (pivot rule of vican_chol_qr3 with pivot_floor 0), Q0 = X0 beta^-1 -> basis columns 0..2 of V and
 * x_out (row-major [n][3], must differ from X0), beta_out [3][3]; if Zraw != NULL also Z = Zraw beta^-1
 * (vican_right_solve3).  Equivalent to vican_rows_to_cols + vican_tall_gram + vican_chol_qr3 up to summation order.
 * coop_sync (may be NULL): the two barrier words of vican_lanczos_cam_coop, zeroed here - an eigen-solve starts with
 * armed barriers even if an earlier launch was torn down.                                                          */
#define VICAN_SEED_MAX_N 16384
int vican_lanczos_seed(int32_t n, const double* X0, double* V, int32_t ld, double* beta_out, double* x_out,
                       const double* Zraw, double* Z, void* coop_sync, void* stream);
/* X[n][3] (row-major) = V[:, :ka] Y[ka][3] */
int vican_tall_combine(int32_t n, const double* V, int32_t ld, int32_t ka, const double* Y,
                       double* X, void* stream);
/* basis columns col0..col0+2 = X (row-major [n][3]) */
int vican_rows_to_cols(int32_t n, const double* X, double* V, int32_t ld, int32_t col0, void* stream);
/* Composite = the camera-side half of block-Lanczos step j as one host call: vican_lap_apply,
 * two Gram-Schmidt passes (vican_tall_gram + vican_tall_update, coefficients summed into
 * Hcol[3(j+1)][3]), R^T R, vican_chol_qr3 into basis block j+1 / beta / x_out.  R [3][3C],
 * H [>= 3(j+1)*3], G [9] are scratch; ws / ws_doubles as in vican_tall_gram (may be NULL / 0). */
int vican_lanczos_cam_step(int32_t n_cam, const double* lamC, double* V, int32_t ld, int32_t j,
                           const double* z, double* R, double* H, double* G, double* Hcol,
                           double* beta, double* x_out, double pivot_floor,
                           double* ws, int64_t ws_doubles, void* stream);

/* The same step as ONE cooperative kernel: <= 32 workgroups (32 cameras each, all resident) that meet at
 * two device-side grid barriers instead of seven dependent launches.  ws: scratch of
 * vican_lanczos_coop_ws_doubles(n_cam) doubles; sync_ws: two 32-bit words, zero before the first call
 * (the kernel leaves them zero).  n_cam <= 8192 (256 workgroups of 32 cameras; VICAN_ERR_CAPACITY if that grid is not co-resident).  Results equal vican_lanczos_cam_step up to the order
 * of the (fixed-order, deterministic) partial sums.
 * zpart != NULL: z is taken straight from the fixed-point slabs [n_slab][9][n_cam] of the preceding
 * vican_block_op (folded per workgroup with the conversion of vican_slab_reduce_fx: pa = fx+3, pb = fx+7),
 * which saves the separate fold launch of every Lanczos step; the argument z is then ignored.
 * fenced != 0: the grid barriers carry agent-scope release / acquire fences (for graphs whose sweeps leave little dirty
 * data in L2 - up to ~64 slabs - where a fence costs 0.4-0.8 us; with the stress graph's 18 MB of slabs it costs ~10 us and
 * the kernel relies on its agent-scope atomics alone).  Bounded spins / co-residency check: vican_set_barrier_abort.  */
int64_t vican_lanczos_coop_ws_doubles(int32_t n_cam);
int vican_lanczos_cam_coop(int32_t n_cam, const double* lamC, double* V, int32_t ld, int32_t j,
                           const double* z, double* ws, double* Hcol, double* beta, double* x_out,
                           double pivot_floor, uint32_t* sync_ws, const void* zpart, int32_t n_slab,
                           const double* pa, const double* pb, int32_t fenced, void* stream);

/* One Lanczos step of a single rank behind one host call: vican_block_op (the sweep, slabs into zpart) + vican_lanczos_cam_coop folding
 * them (n_slab = g->n_wg, pa = fx + 3, pb = fx + 7).  x: the step's operand [3C][3] (x_out of the previous step; may equal x_out).
 * VICAN_ERR_CAPACITY: the sweep has run, the cooperative grid was refused - fold the slabs (vican_slab_reduce_fx) and call
 * vican_lanczos_cam_step.                                                                                                    */
int vican_lanczos_step_slabs(const vican_graph_t* g, const double* lamT_inv, const double* x, void* zpart, double* fx,
                             const double* lamC, double* V, int32_t ld, int32_t j, double* ws, double* Hcol, double* beta,
                             double* x_out, double pivot_floor, uint32_t* sync_ws, int32_t fenced, void* stream);

/* Ritz step on the device (replaces the shift-invert ARPACK call of bipgo.py:288 together with the
 * Lanczos steps).  HB[steps][row_stride]: row j = projected column V^T L Q_j ([hw/3][3] row-major,
 * rows < 3(j+1) used) followed at offset hw by beta_j [3][3], as written by vican_lanczos_cam_step.
 * A zero pivot in beta_j truncates the basis to eff = j+1 blocks.  The symmetric projected matrix
 * (3 eff x 3 eff) is diagonalised by a parallel cyclic Jacobi iteration in LDS; Y[3 steps][3]
 * receives the Ritz vectors of the three smallest values (zero rows beyond 3 eff).
 * flags: bit 0 = first check of this Krylov run (no previous residual), bit 1 = the step budget
 * is exhausted (forces stop).  With r = max_k |beta Y_k[last 3 rows]| / max|theta|:
 *   floor_hit = (not first and r > stall_ratio r_prev and r <= floor_tol) or (floor_level >= 0 and r <= 2 floor_level)
 *               (stall_ratio in (0,1): 1/4 for checks several steps apart, 1/2 for consecutive steps)
 *   stop      = eff < steps or breakdown or r <= eig_tol or floor_hit or bit 1
 *   converged = breakdown or floor_hit or r <= eig_tol           *gate = stop and converged
 * status[VICAN_RITZ_STATUS_DOUBLES]: [0] r, [1] max|theta|, [2] stop, [3] converged, [4] floor_hit,
 * [5] eff, [6] breakdown (all pivots of the last beta zero), [7..9] three smallest Ritz values,
 * [10..11] the two largest (NaN when 3 eff < 5), [12] r (read back as r_prev by the next call),
 * [13] 5th smallest Ritz value (NaN if none; with [7..9] and [15] the counterpart of the five values eigs(k=5, sigma=-1e-6)
 * returns, whose max|.| <= 1e-6 ends the reference's loop, bipgo.py:283), [14] unscaled residual, [15] 4th smallest Ritz value (NaN if none).  steps <= VICAN_RITZ_MAX_STEPS.    */
#define VICAN_RITZ_MAX_STEPS 32
#define VICAN_RITZ_STATUS_DOUBLES 16
int vican_ritz(const double* HB, int32_t row_stride, int32_t hw, int32_t steps, int32_t flags,
               double eig_tol, double floor_tol, double floor_level, double stall_ratio, double* Y,
               double* status, int32_t* gate, void* stream);

/* ---- front-end on the device: constraint application + multi-marker merge (bipgo.py:203-221, 445-469) --------------------
 * One entry per KEPT source edge (the host has evaluated edge_filter / noise_model_r / noise_model_t and mapped the string
 * ids to indices - frontend.index_edges): cam / tim / marker [n] int32 indices, R [n][9] and t [n][3] the measured pose of
 * the marker in the camera frame, kr / kt [n] the two weights; CmT [n_marker][9] = R_m^T R_root and qtau [n_marker][3] =
 * (R_root^T R_m) trans(S_m^-1 S_root) per marker.  Output, all device arrays sized for n (the merged count E <= n comes back
 * in *n_merged): the timestep-major CSR problem row_ptr [n_time+1], col [E] (ascending camera inside a row),
 * blk [E][9] = sum k_r R~ R_m^T R_root, a [E] = sum k_r, w [E] = sum kf^2, u [E][3] = sum kf k_t t~, v [E][3] = sum kf k_t
 * qtau_m (kf = k_t rounded to the matrix dtype `storage`), and deg_c [n_cam], deg_t [n_time] = the diagonal of the reference's
 * J^T J, accumulated in the matrix dtype in source-edge order as scipy's csr_matmat does.  Every sum runs sequentially in
 * source-edge order with unfused IEEE operations: bit-identical to frontend.merge_host.  ws: vican_merge_ws_bytes bytes.
 * kr_f32 (NULL: none): per source edge, nonzero where the reference's `k_r * R` is a float32 product - a float32 rotation
 * (every pose of object mode: SE3.inv(), geometry.py:239-243) weighted by a Python scalar: weight and product are rounded to
 * float32 there, as numpy does.                                                                                            */
int64_t vican_merge_ws_bytes(int64_t n, int32_t n_cam, int32_t n_time);
int vican_merge_edges(int64_t n, int32_t n_cam, int32_t n_time, int32_t n_marker, int32_t storage,
                      const int32_t* cam, const int32_t* tim, const int32_t* marker, const double* R, const double* t,
                      const double* kr, const uint8_t* kr_f32, const double* kt, const double* CmT, const double* qtau, void* ws, int64_t ws_bytes,
                      int32_t* n_merged, int32_t* row_ptr, int32_t* col, double* blk, double* a, double* w, double* u, double* v,
                      double* deg_c, double* deg_t, void* stream);

/* ---- translation stage ---------------------------------------------------
 * Unknowns p (cameras [C][3], timesteps [T][3], double).  Normal equations of
 * bipgo.py:463-477:  (weighted bipartite Laplacian (x) I3) p = J^T b.          */

/* Right-hand side J^T b (bipgo.py:451-461 + J^T): g_ct = Rc_c^T u_ct + Rt_t^T v_ct;
 * rhs_t[t] = sum_c g_ct, rhs_c[c] = -sum_t g_ct over this rank's rows (both written; rhs_c_part: scratch of n_wg * 6C
 * 64-bit words for the double-word fixed-point camera slabs).  u, v: per-edge 3-vectors in the slot order of g
 * [n_chunk][3][slots] - either layout (block: trans_rhs_kernel; wave: trans_wrhs_kernel, one wavefront per chunk).
 * gmax >= max_e (|u_e| + |v_e|) and n_add >= the number of contributions one workgroup adds into one accumulator
 * size the scale.  Sums are exact (double-word fixed point, to_fix2) and rounded to f64 once per output.   */
int vican_trans_rhs(const vican_graph_t* g, const double* u, const double* v,
                    const double* rc, const double* rt, double* rhs_t, double* rhs_c, void* rhs_c_part,
                    double gmax, double n_add, void* stream);

/* ---- Jacobi (diagonal) scaling for the tight translation solve ----------------------------
 * The normal equations A = [[D_c, -W], [-W^T, D_t]] (x) I3 scaled symmetrically by S = D^-1/2 are again a
 * weighted bipartite Laplacian system with unit degrees and weights w~_ct = w_ct s_c s_t <= 1, so the
 * CG entry points below run Jacobi-preconditioned CG unchanged on (w~, deg = 1, S b); x = S x~.
 * (Not in the reference - its CG stops at relres 1e-5, SURVEY.md section 7 - off by default.)       */
/* s[i] = deg[i] > 0 ? deg[i]^-1/2 : 0 */
int vican_jacobi_scale(int32_t n, const double* deg, double* s, void* stream);
/* x[i][0..ncomp) *= s[i] */
int vican_row_scale(int32_t n, int32_t ncomp, const double* s, double* x, void* stream);
/* w_out[slot] = w[slot] * s_cam[camera(slot)] * s_row[row(slot)]   (chunk layout; padding slots -> 0) */
int vican_scale_weights(const vican_graph_t* g, const double* w, const double* s_cam,
                        const double* s_row, double* w_out, void* stream);

/* Device-resident CG state (one struct in device memory, initialised by vican_cg_init).
 * Mirrors scipy.sparse.linalg.cg (x0 = 0, no preconditioner, stop when |r| < rtol*|b| tested at
 * the top of every iteration).  The sweeps accumulate q = A p in DOUBLE-WORD fixed point (two 64-bit integers per sum:
 * hi = rint(v qscale), lo = rint((v qscale - hi) 2^lo_bits), i.e. 49 + 48 bits below the bound - every term w p keeps the
 * 53 bits scipy's f64 product gives it); qscale follows a bound pmax >= max|p| that is re-derived every iteration from
 * measured maxima (|p_new| <= max|r| + beta max|p|).   */
typedef struct vican_cg_state {
    double rho;        /* r.r of the current residual */
    double rho_prev;
    double pq;         /* p.q */
    double alpha;
    double beta;
    double bnorm2;     /* |b|^2 */
    double atol2;      /* (rtol*|b|)^2 */
    double rr_cam;     /* camera part of r.r (set by cg_cam_step) */
    double pq_time;    /* timestep part of p.q */
    double rr_time;    /* timestep part of r.r (all-reduced across ranks) */
    double rmax_cam;   /* max |r_c| */
    double rmax_time;  /* max |r_t| over THIS rank's rows */
    double pmax;       /* bound on max |p| (this rank): max(exact max |p_c|, rmax_time + beta pmax_time) */
    double qscale;     /* fixed-point scale of the current sweep, and its inverse */
    double qinv;
    double wmax;       /* max edge weight (graph constant) */
    double pmax_time;  /* measured max |p_t| of the current iterate over THIS rank's rows (set by the step kernels) */
    int32_t iter;      /* completed iterations */
    int32_t done;      /* 1 once converged, -2 when r.r became NaN, -1 when vican_cg_resident's barrier spin was aborted
                          (vican_set_barrier_abort): all later kernels are no-ops */
    int32_t first;     /* 1 before the first iteration (p = r) */
    int32_t lo_bits;   /* bits of the lo word of this sweep's double-word accumulators (48 unless n_add > 2^14) */
} vican_cg_state_t;

/* x=0, r=b, p=r for both node sets; |b_t|^2 into st->rr_time (caller all-reduces it across
 * ranks before cg_begin), |b_c|^2 into st->rr_cam.  ws: >= 1024 doubles.              */
int vican_cg_init(int32_t n_cam, int32_t n_time, const double* b_c, const double* b_t,
                  double* x_c, double* x_t, double* r_c, double* r_t, double* p_c, double* p_t,
                  vican_cg_state_t* st, double* ws, double wmax, void* stream);
/* Top of an iteration.  If n_part > 0 first closes the previous iteration like
 * vican_cg_end(rr_part, n_part) (single-GPU fast path; multi-GPU callers use
 * vican_cg_end + all-reduce of st->rr_time and pass n_part = 0).  Then
 * rho = rr_cam + rr_time; the first call fixes atol2 = rtol^2 |b|^2; sets done
 * when |r| < atol (scipy's test, before the step); otherwise beta = rho/rho_prev,
 * p_c = r_c + beta p_c (p_t is updated inside the sweep) and the sweep's fixed-point scale
 * from wmax * pmax and n_add (adds into one accumulator by one workgroup).        */
int vican_cg_begin(int32_t n_cam, const double* r_c, double* p_c, double rtol,
                   const double* rr_part, int32_t n_part, double n_add, vican_cg_state_t* st,
                   void* stream);
/* Timestep-major Laplacian sweep: p_t <- r_t + beta p_t (skipped on the first
 * iteration), q_t = deg_t p_t - sum_c w_ct p_c (written), double-word fixed-point slabs
 * qc_part[wg][2][3][C] = (hi, lo) planes of sum_t w_ct p_t (folded by vican_cg_iter_local),
 * pq_part[wg] = partial p_t.q_t.                                               */
int vican_cg_sweep(const vican_graph_t* g, const double* w, const double* deg_t,
                   const double* p_c, const double* r_t, double* p_t, double* q_t,
                   void* qc_part, double* pq_part, const vican_cg_state_t* st, void* stream);
/* Fold of one sweep: qcpq[0:3C] = sum over the n_slab workgroup slabs of qc_part (double-word fixed point -> [C][3] doubles,
 * each rounded once; exact, order-independent, overflow-proof integer sums) and qcpq[3C] = sum pq_part in a fixed order
 * (pq_part = NULL: the camera part only):  the message a sharded run all-reduces per CG iteration.          */
int vican_cg_fold(const void* qc_part, int32_t n_slab, int32_t n_cam, const double* pq_part, double* qcpq,
                  const vican_cg_state_t* st, void* stream);
/* The folds of the camera tiles of one CG product (vican_cg_sweep_tiles) in one launch: qc_parts / n_cams: HOST arrays of n_tile
 * slab pointers and camera counts (tiles = consecutive camera ranges); q_c of tile k lands at qcpq[3 (C_0 + ... + C_(k-1)) ...].    */
int vican_cg_fold_tiles(const void* const* qc_parts, const int32_t* n_cams, int32_t n_tile, int32_t n_slab, double* qcpq,
                        const vican_cg_state_t* st, void* stream);
/* The CG Laplacian product of a camera-tiled graph in ONE launch (csrc/vican_tcg.hip; per tile: vican_cg_sweep_partial): tile k's
 * sweep writes its share of the row sums into acc_t and the double-word slabs of its own cameras into qc_part (n_wg_tile slabs
 * of 6 C_tile words; fold with vican_cg_fold(pq_part = NULL)); p_t already updated (vican_cg_update_pt), rows combined by
 * vican_cg_combine_rows.  2..4 wave-layout tiles of one launch shape (12 wavefronts per workgroup), VICAN_ERR_CAPACITY otherwise. */
typedef struct {
    vican_graph_t g;
    const double* w;          /* the tile's weights in its slot order */
    const double* p_c;        /* [C_tile][3] the tile's slice of the camera vector */
    double* acc_t;            /* [T][3] receives sum_{c in tile} w p_c */
    void* qc_part;            /* n_wg_tile slabs of 6 C_tile 64-bit words */
} vican_cg_tile_t;
int vican_cg_sweep_tiles(const vican_cg_tile_t* tiles, int32_t n_tile, int32_t n_wg_tile, const double* p_t,
                         const vican_cg_state_t* st, void* stream);

/* Camera-tiled graphs (more cameras than one LDS table holds; the reference has no camera limit, bipgo.py:225-232): the
 * product q = A p one camera tile at a time.  vican_cg_update_pt: p_t <- r_t + beta p_t (skipped on the first iteration) -
 * what vican_cg_sweep does while it loads its rows.  vican_cg_sweep_partial (g = ONE tile's block-layout graph, w its
 * weights, p_c the tile's slice of the camera vector): acc_t [T][3] = sum_{c in tile} w_ct p_c and the double-word slabs of
 * sum_t w_ct p_t for the tile's cameras (complete: fold them with vican_cg_fold(pq_part = NULL) into the tile's slice of
 * qcpq).  vican_cg_combine_rows: q_t = deg_t p_t - sum_tiles acc (tile order), pq_part[b] = partial p_t.q_t; returns the
 * number of partials (sum them with vican_cg_reduce_pq).  acc: [n_tile][tile_stride] doubles, tile_stride >= 3 n_time.   */
int vican_cg_update_pt(int32_t n_time, const double* r_t, double* p_t, const vican_cg_state_t* st, void* stream);
int vican_cg_sweep_partial(const vican_graph_t* g, const double* w, const double* p_c, const double* p_t, double* acc_t,
                           void* qc_part, const vican_cg_state_t* st, void* stream);
int vican_cg_combine_rows(int32_t n_time, int32_t n_tile, int64_t tile_stride, const double* deg_t, const double* p_t,
                          const double* acc, double* q_t, double* pq_part, int32_t part_cap, const vican_cg_state_t* st,
                          void* stream);
/* *out = sum pq_part (the timestep part of p.q); `out` is normally the slot right
 * behind the reduced q_c vector so that ONE all-reduce carries both.           */
int vican_cg_reduce_pq(const double* pq_part, int32_t n_part, double* out,
                       const vican_cg_state_t* st, void* stream);
/* Camera side: q_c = deg_c p_c - qc_sum; pq = *pq_time + p_c.q_c; alpha = rho/pq;
 * x_c += alpha p_c; r_c -= alpha q_c; rr_cam = r_c.r_c.  qc_sum: [C][3] (slabs
 * already reduced / all-reduced).                                            */
int vican_cg_cam_step(int32_t n_cam, const double* deg_c, const double* qc_sum,
                      const double* pq_time, const double* p_c, double* x_c, double* r_c,
                      vican_cg_state_t* st, void* stream);
/* Timestep side: x_t += alpha p_t; r_t -= alpha q_t; rr_part[blk] partial r_t.r_t and
 * rr_part[512 + blk] partial max|r_t|, rr_part[1024 + blk] partial max|p_t| (rr_part: >= 1536 doubles); returns the number of
 * partials written (>0).                                                       */
int vican_cg_time_step(int32_t n_time, const double* p_t, const double* q_t, double* x_t,
                       double* r_t, double* rr_part, int32_t part_cap, const vican_cg_state_t* st,
                       void* stream);
/* st->rr_time = sum rr_part ; st->rmax_time = max ; st->iter += 1 ; rho_prev = rho. */
int vican_cg_end(const double* rr_part, int32_t n_part, vican_cg_state_t* st, void* stream);
/* Composites: one CG iteration as two host calls.  vican_cg_iter_local = cg_begin + cg_sweep +
 * vican_cg_fold, leaving [q_c partial | p.q partial] in qcpq[3C+1] (all-reduce it when
 * sharded); vican_cg_iter_finish = cg_cam_step + cg_time_step (returns the number of rr partials). */
int vican_cg_iter_local(const vican_graph_t* g, const double* w, const double* deg_t, const double* r_c,
                        double* p_c, const double* r_t, double* p_t, double* q_t, void* qc_part,
                        double* pq_part, double* qcpq, double rtol, const double* rr_part,
                        int32_t n_part, double n_add, vican_cg_state_t* st, void* stream);
int vican_cg_iter_finish(int32_t n_cam, int32_t n_time, const double* deg_c, const double* qcpq,
                         const double* p_c, double* x_c, double* r_c, const double* p_t,
                         const double* q_t, double* x_t, double* r_t, double* rr_part,
                         int32_t part_cap, vican_cg_state_t* st, void* stream);
/* Single rank: one CG iteration behind one host call - vican_cg_begin + vican_cg_sweep + a fold that also forms p_t.q_t over
 * fixed slices (bit-reproducible from run to run: the sweep's own partial depends on the order of its chunk tickets) + the
 * step.  Same recurrence as (vican_cg_iter_local, vican_cg_iter_finish): iterates agree to the rounding of the differently
 * grouped p_t.q_t; poll st->done as there.  first != 0: the call that follows vican_cg_init.  ws: 1024 bytes owned by the solve
 * (the fold's p.q partials); rr_part >= 1536 doubles.  scipy cg, bipgo.py:477.                                              */
int vican_cg_iter_fused(const vican_graph_t* g, const double* w, const double* deg_t, const double* deg_c,
                        double* r_c, double* p_c, double* x_c, double* r_t, double* p_t, double* q_t, double* x_t,
                        void* qc_part, double* pq_part, double* qcpq, double rtol, double* rr_part, int32_t part_cap,
                        double n_add, int32_t first, vican_cg_state_t* st, uint32_t* ws, void* stream);

/* ONE message per CG iteration for timestep-sharded solves (replaces the two reductions per iteration of the pair above;
 * scipy.sparse.linalg.cg at bipgo.py:476-478 on a sharded graph): the Chronopoulos-Gear arrangement - the product is formed
 * on the RESIDUAL (s = A r), gamma = r.r and delta = r.s travel together, beta = gamma/gamma_prev,
 * alpha = gamma / (delta - beta gamma / alpha_prev), p = r + beta p, q = s + beta q, x += alpha p, r -= alpha q.  Same
 * iterates in exact arithmetic and scipy's stopping test (|r_k| < rtol |b|, r.r formed directly, before iteration k's update);
 * roundings differ from scipy's recurrence (q by recurrence, alpha from delta).  Start from vican_cg_init (x = 0, r = b).
 *   vican_cg1_iter_local : msg[0:3C] = sum_t w r_t over THIS rank's rows ([C][3]), msg[3C] = r_t.s_t, msg[3C+1] = r_t.r_t of
 *     this rank; s_t [T][3] = deg_t r_t - sum_c w r_c written.  rr_part / n_part: the partials the previous
 *     vican_cg1_iter_finish returned (n_part = 0 before the first iteration).  sw: a scratch vican_cg_state_t (the sweep's
 *     view of the state).  All-reduce msg[0:3C+2] over the ranks, then
 *   vican_cg1_iter_finish(k = 0, 1, 2, ...): the scalars, the stopping test (st->done, st->iter = completed iterations) and
 *     the vector updates; the updated camera residual goes to r_c_new (another buffer than r_c: pass it as r_c to the next
 *     vican_cg1_iter_local and swap the two); q_c [C][3]; sc: 4 doubles of scratch kept between the calls; rr_part >= 1024
 *     doubles; returns the number of partials written.  Both calls do nothing once st->done != 0.                                            */
int vican_cg1_iter_local(const vican_graph_t* g, const double* w, const double* deg_t, const double* r_c,
                         const double* r_t, double* s_t, void* qc_part, double* pq_part, double* msg,
                         const double* rr_part, int32_t n_part, double n_add, const vican_cg_state_t* st,
                         vican_cg_state_t* sw, void* stream);
int vican_cg1_iter_finish(int32_t n_cam, int32_t n_time, int32_t k, double rtol, const double* deg_c, const double* msg,
                          const double* r_c, double* r_c_new, double* p_c, double* q_c, double* x_c, double* r_t, const double* s_t,
                          double* p_t, double* q_t, double* x_t, double* rr_part, int32_t part_cap, double* sc,
                          vican_cg_state_t* st, void* stream);

/* The whole CG solve as ONE cooperative launch (vican_cgres.hip) for wave-layout graphs whose n_wg workgroups are
 * co-resident (n_wg <= compute units, <= 256) and whose per-workgroup rows fit in LDS: replaces the vican_cg_init /
 * vican_cg_iter_local / vican_cg_iter_finish sequence of scipy.sparse.linalg.cg at bipgo.py:476-478 on capture-sized
 * graphs, where that sequence is bound by launch latency.  Same recurrences, stopping test (|r| < rtol |b| at the top of
 * an iteration, at most max_iter iterations) and fixed-point accumulation; the floating-point partial sums of r.r and
 * p.q are grouped per workgroup and the scale bound uses measured maxima on both node sets, so the iterates agree with
 * the multi-kernel path to rounding.  b_c [C][3], b_t [T][3]: right-hand side; x_c, x_t: solution; slab: n_wg * 6C 64-bit
 * words; ws: vican_cg_resident_ws_doubles() doubles (its barrier counter is zeroed in-stream by every call); wmax: max edge weight; rows_per_wg: most rows in one workgroup's chunk range [n_chunk b / n_wg,
 * n_chunk (b + 1) / n_wg); st receives the final state (iter, done, rho, bnorm2, ...).  n_add as for vican_cg_begin.   */
int64_t vican_cg_resident_lds_bytes(int32_t n_cam, int32_t max_rows, int32_t n_copy, int32_t rows_per_wg);
int64_t vican_cg_resident_ws_doubles(int32_t n_cam, int32_t n_wg);
int vican_cg_resident(const vican_graph_t* g, const double* w, const double* deg_t, const double* deg_c,
                      const double* b_c, const double* b_t, double* x_c, double* x_t, void* slab, double* ws,
                      double rtol, int32_t max_iter, double n_add, double wmax, int32_t rows_per_wg,
                      vican_cg_state_t* st, void* stream);

/* ---- LSQR translation solve (lsqr_solver="direct", bipgo.py:479-480) -------------------
 * scipy.sparse.linalg.lsqr on the incidence matrix is reproduced on the MERGED system
 *   J~ p = b~ :  s_e (p_t - p_c) = g_e / s_e,  s_e = sqrt(w_e),  g_e = Rc_c^T u_e + Rt_t^T v_e
 * (same normal equations => same iterates; residual norms differ by the constant
 * |b|^2 - |b~|^2, added back by the host driver).  (The round-2 two-pass steps with host-side Golub-Kahan scalars -
 * vican_lsqr_u_step / _v_step / _cam_v - are kept for cross-checks only: include/vican_hip_test.h.)
 * u: per-edge 3-vectors in slot order [n_chunk][3][slots]; part: >= max(n_wg, 1024) doubles of
 * scratch; the *_out scalars are single device doubles (all-reduce them when sharded).        */
/* u <- b~ (unnormalised);  *nrm2_out = |u|^2 over this rank's edges;  sw[slot] = s_e = sqrt(w_e) (slot order,
 * 0 on padding) for the step kernels */
int vican_lsqr_init_u(const vican_graph_t* g, const double* w, const double* ue, const double* ve,
                      const double* rc, const double* rt, double* u, double* sw, double* part,
                      double* nrm2_out, void* stream);
/* v *= inv_alfa ; x += t1 w ; w <- v + t2 w ; *nrm2_w_out = |w_new|^2   (vectors of length n) */
int vican_lsqr_update(int64_t n, double inv_alfa, double t1, double t2, double* v, double* w,
                      double* x, double* part, double* nrm2_w_out, void* stream);

/* Device-resident LSQR iteration: every scalar of scipy's loop lives in this struct (initialised by the caller after the
 * first bidiagonalisation step), the host only polls `done`.  The edge vector is stored UNNORMALISED (u~_i = beta_i u_i). */
typedef struct vican_lsqr_state {
    double alfa, beta, rhobar, phibar, anorm, ddnorm, xxnorm, z, cs2, sn2;   /* scipy's variables of the same names */
    double c2;         /* |b|^2 - |b~|^2: added back to the residual norms of the merged system */
    double bnorm, atol, btol, ctol;
    double coef;       /* alfa_i / beta_i: factor of the stored u~ in the next step */
    double inv_alfa, t1, t2;   /* coefficients of the pending v / x / w update */
    double rnorm, arnorm, acond, xnorm;   /* scipy's estimates after the last completed iteration */
    double qscale, qinv;       /* fixed-point scale of the next step's sums */
    double smax;       /* max sqrt(w_e) */
    double n_add;      /* contributions one workgroup adds into one accumulator */
    double reserved;
    int32_t itn, istop, done, iter_lim;   /* istop: scipy's codes 1..7; 8 = NaN */
    int32_t lo_bits, update, pad0, pad1;  /* update: 1 while the x / w update of iteration itn is still to run */
} vican_lsqr_state_t;
/* One fused pass over the edges (either layout; on wave layouts u~ and sw are in the permuted 8-byte order of the layout,
 * as vican_lsqr_init_u writes them): u^ = J~ v - coef u~ written over u~, z_t [T][3] = row sums of J~^T u^, the camera
 * sums as double-word slabs (zc_part: n_wg * 6C words) folded into acc[0:3C] ([C][3]), acc[3C] = |u^|^2 = beta_{i+1}^2 - this
 * rank's part: all-reduce acc[0 : 3C+1] when sharded.  part: >= max(n_wg, 1024) doubles of scratch.                       */
int vican_lsqr_step(const vican_graph_t* g, const double* sw, double* u, const double* v_c, const double* v_t, double* z_t,
                    void* zc_part, double* part, double* acc, const vican_lsqr_state_t* st, void* stream);
/* v_t <- z_t / beta' - beta' v_t, v_c <- acc_c / beta' - beta' v_c with beta' = sqrt(acc[3C]); part2[0:ret] = partials of |v_t|^2,
 * part2[1024] = |v_c|^2 (part2: >= 1025 doubles); returns the number of partials.                                            */
int vican_lsqr_nodes(int32_t n_cam, int32_t n_time, const double* z_t, const double* acc, double* v_t, double* v_c, double* part2,
                     const vican_lsqr_state_t* st, void* stream);
/* The scalars of the iteration (Givens rotation, norm estimates, scipy's stopping tests -> istop / done, the update
 * coefficients, coef and the fixed-point scale of the next step).  |v~_t|^2 = sum part2[0:n_part], or *tsum when the caller
 * all-reduced it; |w|^2 = sum wpart_t[0:n_wt] (or *wsum_t) + sum wpart_c[0:n_wc] from the previous vican_lsqr_update_st calls. */
int vican_lsqr_scalars(int32_t n_cam, const double* acc, const double* part2, int32_t n_part, const double* tsum,
                       const double* wpart_t, int32_t n_wt, const double* wpart_c, int32_t n_wc, const double* wsum_t,
                       vican_lsqr_state_t* st, void* stream);
/* vican_lsqr_update with the coefficients of the state; runs iff state.update (set by vican_lsqr_scalars: scipy updates x
 * before it tests; cleared by the first vican_lsqr_scalars call after `done`), `last` is ignored (kept for callers of ABI 11);
 * returns the number of |w|^2 partials written to part.  */
int vican_lsqr_update_st(int64_t n, double* v, double* w, double* x, double* part, int32_t last, vican_lsqr_state_t* st,
                         void* stream);


/* ---- collectives of the sharded solve (vican_comm_*) --------------------------------------------------------
 * Timestep rows are sharded over ranks, every camera-side quantity is replicated; what crosses ranks are f64 sum-all-reduces of
 * camera-side partials (the reference has no distributed code: this is the counterpart of bipgo.py:300,318 / :477 when the rows
 * of R~ live on several GPUs).  A communicator held by the library, every collective enqueued on the caller's stream in stream
 * order - no host synchronisation, no Python hop between a kernel and the collective behind it.  Two transports:
 *   RCCL            loaded at run time (dlopen librccl.so.1; no link-time dependency): ncclAllReduce
 *   peer exchange   ONE launch of the library's own kernel over mailboxes the ranks map into each other's address space
 *                   (hipIpc handles; xGMI between the GPUs of a node): every rank pushes its partial into its slot of every
 *                   mailbox as self-validating 8-byte granules {32 data bits | epoch tag} and sums the slots of its own mailbox
 *                   in rank order - one hop, the same bits on every rank, no ordering between payload and flag relied on, bounded
 *                   waits (30 s: a peer may arrive seconds late; vican_comm_peer_set_timeout).  The launch honours vican_set_gate (a speculative tail that the device cancels advances nothing on
 *                   any rank), so sharded solves speculate like single-rank ones.  Messages up to `max_doubles`; <= 8 ranks.
 *   vican_comm_unique_id   128 bytes (ncclUniqueId) written by ONE rank, handed to the others by any means
 *   vican_comm_create      collective over the group: every rank calls it with the same id (ncclCommInitRank)
 *   vican_comm_create_local  a communicator without RCCL (ranks sharing one GPU, which RCCL refuses; callers that do not want
 *                          librccl loaded): its collectives exist as the peer exchange only
 *   vican_comm_peer_export allocate this rank's mailbox (vican_comm_peer_bytes; uncached device memory) and write its 64-byte
 *                          hipIpcMemHandle_t to handle_out; the caller gathers the handles of all ranks by any means
 *   vican_comm_peer_attach handles [world][64] in rank order: map the other ranks' mailboxes; from here on
 *                          vican_comm_allreduce_sum takes the exchange for n <= max_doubles (vican_comm_peer_enable(c, 0): back
 *                          to RCCL).  world == 1: the rank's own slot is its peer - the same kernel, the same waits.
 *   vican_comm_peer_status 0, or the number of element waits that timed out so far (host-visible word, no synchronisation):
 *                          those messages came back as NaN; the caller disables the exchange and reports
 *   vican_comm_allreduce_sum   buf[0:n] (device, f64) <- sum over ranks, in place; world == 1 without an attached exchange:
 *                          nothing is enqueued
 *   vican_block_op_z_comm  z = P x of this rank's rows summed over the ranks: vican_block_op_z + the all-reduce of z [3C][3]
 *                          behind one host call (comm NULL: single rank)
 *   vican_cg_iter_comm     one CG iteration of a sharded solve behind one host call: [cg_begin] sweep, fold (camera partials +
 *                          FIXED slices of p_t.q_t) -> msg [3C + VICAN_CG_PQ_SLICES] all-reduced -> update of x, r with alpha
 *                          from the reduced message -> rr_part [VICAN_CG_RR_SLICES] all-reduced; the next call's head closes
 *                          the iteration.  scipy's recurrence, two messages per iteration; bit-reproducible and bit-identical
 *                          on every rank (fixed slices, rank-ordered sums).  first != 0: the call after vican_cg_init and the
 *                          all-reduce of the state's rr_time.  Same buffers as vican_cg_iter_fused otherwise. */
#define VICAN_CG_PQ_SLICES 96
#define VICAN_CG_RR_SLICES 512
typedef struct vican_comm vican_comm_t;
int vican_comm_unique_id(void* id_out /* 128 bytes, host */);
int vican_comm_create(int32_t rank, int32_t world, const void* unique_id, vican_comm_t** comm_out);
int vican_comm_create_local(int32_t rank, int32_t world, vican_comm_t** comm_out);
int64_t vican_comm_peer_bytes(int32_t world, int64_t max_doubles);
int vican_comm_peer_export(vican_comm_t* comm, int64_t max_doubles, void* handle_out /* 64 bytes, host */);
int vican_comm_peer_attach(vican_comm_t* comm, const void* handles /* [world][64], host */);
int vican_comm_peer_enable(vican_comm_t* comm, int32_t on);
/* bound of every wait of the exchange (default 30 s).  A wait that ran into it poisons the communicator: results are NaN from there
 * on, vican_comm_peer_status > 0, and every later launch gives a missing granule only 1/256 of the bound (a solve in flight drains
 * in seconds; the group then falls back - vican_amd.solver.Comm.healthy).                                                          */
int vican_comm_peer_set_timeout(vican_comm_t* comm, int64_t microseconds);
int vican_comm_peer_status(vican_comm_t* comm);
int vican_comm_allreduce_sum(vican_comm_t* comm, double* buf, int64_t n, void* stream);
int vican_comm_destroy(vican_comm_t* comm);
int vican_block_op_z_comm(const vican_graph_t* g, const double* lamT_inv, const double* x, void* zpart, double* fx, double* z,
                          vican_comm_t* comm, void* stream);
int vican_cg_iter_comm(const vican_graph_t* g, const double* w, const double* deg_t, const double* deg_c,
                       double* r_c, double* p_c, double* x_c, double* r_t, double* p_t, double* q_t, double* x_t,
                       void* qc_part, double* pq_part, double* msg, double rtol, double* rr_part, int32_t part_cap,
                       double n_add, int32_t first, vican_cg_state_t* st, vican_comm_t* comm, void* stream);

/* ---- the four-call boundary ----------------------------------------------------------------------------
 * For a maintainer who wants the numerics of the reference's two stages behind ONE handle: everything above composed by host
 * code inside the library (csrc/vican_facade.hip) with the plain schedule - chunked layout planned as vican_amd/device.py does,
 * block Lanczos with a convergence check (vican_ritz + one blocking 128-byte read) every few steps, the fused dual update, CG in
 * bursts with the state polled in between.  The library owns the plan's device memory; inputs and outputs are the caller's device
 * buffers; every call synchronises `stream` before it returns.
 * More than 1024 cameras (the reference has no camera limit, bipgo.py:225-232; one LDS table of the sweeps holds 1024): the plan
 * cuts the cameras into <= 64 tiles of equal width that share one chunking of the timestep rows and runs the tiled schedule
 * (csrc/vican_facade_tiles.hip, as vican_amd/tiled.py: the operator as one launch that reads every block once, dual update / J^T b /
 * CG product tile by tile, rows summed in tile order; the rows packed for the shared chunking in an order of the plan's own, per-row
 * arguments translated at the boundary) - same calls, same outputs, vican_plan_set_comm included.  VICAN_ERR_CAPACITY
 * where a timestep row has more than 256 (f32) / 128 (f64) edges inside one tile or a tile has no edges at all (layouts only the
 * host driver vican_amd.tiled plans).
 *
 * vican_plan_create  replaces bipgo.py:244-276 (COO triplets -> CSR, degrees, power-graph constants): the merged timestep-major
 *     CSR problem (row_ptr [T+1], col [E] ascending inside a row, blk [E][9] and a [E] in the storage type; optionally the
 *     translation arrays w [E], u [E][3], v [E][3] in f64 and the diagonal of the reference's J^T J, deg_t [T] / deg_c [C],
 *     as scipy forms it - NULL: row / camera sums of w) -> chunked layout + graph constants + solver workspace.
 * vican_solve_rot     replaces the primal-dual loop bipgo.py:279-348: maxiter iterations of {3 smallest eigenvectors of
 *     Lambda_C - P, gauge fix, projections, dual updates}; rc_out [3C][3] = the stacked node<-world camera rotations (the
 *     reference's r_c), Rt_out [T][9] the timestep rotations r_t; the reference returns their transposes (bipgo.py:346,348).
 * vican_solve_trans   replaces bipgo.py:445-478: J^T b for these rotations and scipy's CG (x0 = 0, stop when |r| < rtol |b|,
 *     at most maxiter iterations; <= 0: scipy's default 10 n) -> x_c [C][3], x_t [T][3]; VICAN_ERR_LAUNCH if it does not converge
 *     (the reference asserts, bipgo.py:478).
 * info (may be NULL): counters of the stage just run.                                                                        */
typedef struct vican_plan vican_plan_t;
typedef struct vican_solve_info {
    int32_t iterations, sweeps, lanczos_steps, restarts;   /* rotation stage */
    double  evals[5];                                       /* last iteration: the five smallest Ritz values (cf. eigs k=5) */
    double  eig_resid;
    int32_t cg_iters, cg_converged;                         /* translation stage */
    double  cg_relres;
} vican_solve_info_t;
int vican_plan_create(int32_t n_cam, int32_t n_time, int64_t n_edges, int32_t storage, const int32_t* row_ptr,
                      const int32_t* col, const void* blk, const void* a, const double* w, const double* u, const double* v,
                      const double* deg_t, const double* deg_c, void* stream, vican_plan_t** plan_out);
int vican_plan_describe(const vican_plan_t* plan, vican_graph_t* g_out);   /* the layout that was planned (sizes; device pointers) */
/* One rank of a timestep-sharded solve (the north star's partitioning; the reference has no distributed code): the plan was
 * created from THIS rank's rows (its slice of the timesteps with all their edges, all C cameras; deg_c = this rank's share of the
 * camera diagonal: the whole vector on one rank and zeros on the others, or NULL).  After this call vican_solve_rot /
 * vican_solve_trans all-reduce the camera-side partials through `comm` (vican_comm_*: peer exchange or RCCL) in stream order -
 * one all-reduce per operator application, two per CG iteration - and every rank of the group must make the same calls;
 * rc_out / x_c are identical on all ranks, Rt_out / x_t are this rank's rows.  One collective inside: the graph constants that
 * must be global.  vican_solve_trans_lsqr stays single-rank.  comm = NULL: back to a single-rank plan.                        */
int vican_plan_set_comm(vican_plan_t* plan, vican_comm_t* comm, void* stream);
int vican_solve_rot(vican_plan_t* plan, int32_t maxiter, double eig_tol, double* rc_out, double* Rt_out,
                    vican_solve_info_t* info, void* stream);
int vican_solve_trans(vican_plan_t* plan, const double* rc, const double* Rt, double rtol, int64_t maxiter, double* x_c,
                      double* x_t, vican_solve_info_t* info, void* stream);
/* lsqr_solver="direct" (scipy.sparse.linalg.lsqr, bipgo.py:479-480) behind the same handle: the device-resident LSQR iteration
 * (vican_lsqr_step: one fused pass over the edges per iteration, every scalar of scipy's loop on the device) with scipy's
 * stopping tests - atol / btol / conlim / iter_lim as scipy's arguments (0 for iter_lim: 2 n).  bnorm2_true = |b|^2 of the
 * reference's un-merged right-hand side (vican_amd.frontend.bnorm2), or <= 0 where no (camera, timestep) pair carries more than
 * one detection (then it equals the merged system's own).  info: scipy's istop / itn / r1norm / arnorm / anorm / acond / xnorm. */
typedef struct vican_lsqr_info {
    int32_t itn, istop;
    double rnorm, arnorm, anorm, acond, xnorm;
} vican_lsqr_info_t;
int vican_solve_trans_lsqr(vican_plan_t* plan, const double* rc, const double* Rt, double bnorm2_true, double atol, double btol,
                           double conlim, int64_t iter_lim, double* x_c, double* x_t, vican_lsqr_info_t* info, void* stream);
int vican_plan_destroy(vican_plan_t* plan);

#ifdef __cplusplus
}
#endif
#endif /* VICAN_HIP_H */
