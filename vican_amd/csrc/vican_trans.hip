// vican_trans.hip - translation stage: right-hand side J^T b and the conjugate-gradient kernels
// (scipy.sparse.linalg.cg recurrence, reference bipgo.py:445-478) on the merged weighted
// bipartite Laplacian (x) I3.
//
// Same machinery as the rotation sweep (vican_sweep.hip): chunked timestep-major edge layout
// with bank-aware slots, camera-side vectors as LDS planes [component][camera], row
// accumulators striped over n_copy lane-indexed copies, and 64-bit FIXED-POINT accumulation
// (ds_add_u64) so that q = A p and J^T b are bit-reproducible: the reference's CG stops at a
// loose tolerance and amplifies rounding-order noise by up to 1e10, so
// order-independent sums are what makes two runs of this solver agree with each other.
// The fixed-point scale of every CG sweep follows a running bound on max|p| kept in the
// device-resident state; no host round trip.
#include "vican_sweep_common.h"

#define CG_PARTS 512


// ---------------------------------------------------------------------------
// right-hand side:  g_ct = Rc_c^T u_ct + Rt_t^T v_ct ;  rhs_t = sum_c g_ct ; rhs_c = -sum_t g_ct
// ---------------------------------------------------------------------------
// Software pipeline as in cg_sweep_kernel: the edge words of chunk k+1 (packed index + u + v: 52 B per edge)
// and the R_t blocks of its rows are in flight while chunk k is processed; two barriers per chunk.
// Sums in double-word fixed point (to_fix2): the reference forms J^T b in f64, and its loosely converged CG turns a
// relative perturbation of the right-hand side of 1e-15 into 1e-5 .. 5e-4 m (tests/golden/cg_sensitivity.npz) - one
// 64-bit word 47 bits below the global bound max (|u| + |v|) was a perturbation of that size on every small entry.
template <int EPL>
struct RhsRegs { double u[3][EPL], v[3][EPL]; uint32_t id[EPL]; };

template <int BLOCK, int EPL, int NR>
__global__ __launch_bounds__(BLOCK) void trans_rhs_kernel(vican_graph_t g, const double* __restrict__ u,
                                                          const double* __restrict__ v, const double* __restrict__ rc,
                                                          const double* __restrict__ rt, double* __restrict__ rhs_t,
                                                          u64* __restrict__ rhs_c_part, double scale, double inv, int lob) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int C = g.n_cam, ncopy = g.n_copy, cmask = ncopy - 1, mr9 = 9 * g.max_rows;
    const int lo_c = 3 * C, lo_t = 3 * g.max_rows * ncopy;
    const double lo_scale = ldexp(1.0, lob);
    u64* gc = (u64*)lds_raw;                          // [2][3][C] planes: hi words, lo words
    u64* gt = gc + 6 * C;                             // [2][max_rows*3][ncopy]
    double* rcs = (double*)(gt + (size_t)2 * lo_t);   // [9][C] planes
    double* rts = rcs + 9 * C;                        // [2][max_rows][9]
    const int tid = threadIdx.x, lane_copy = tid & cmask;
    const uint32_t pad_cam = (uint32_t)((tid & 31) < g.n_cam ? (tid & 31) : 0);   // padding slots: a valid camera, zero block
    for (int i = tid; i < 9 * C; i += BLOCK) rcs[(i % 9) * C + i / 9] = rc[i];
    for (int i = tid; i < 6 * C; i += BLOCK) gc[i] = 0ull;
    for (int i = tid; i < 2 * lo_t; i += BLOCK) gt[i] = 0ull;
    const int k0 = (int)(((long long)blockIdx.x * g.n_chunk) / gridDim.x);
    const int k1 = (int)(((long long)(blockIdx.x + 1) * g.n_chunk) / gridDim.x);

    // (52 B per edge, read once: non-temporal loads when the stream is larger than the caches - g.stream_nt)
    auto load_edges = [&](RhsRegs<EPL>& e, int k) {
        const size_t s = (size_t)k * g.slots + (size_t)tid * EPL;
        const bool nt = g.stream_nt != 0;
        if (EPL == 4) { const uint4 t = nt ? stream_load((const uint4*)(g.idx + s)) : *(const uint4*)(g.idx + s); e.id[0] = t.x; e.id[1] = t.y; e.id[2] = t.z; e.id[3] = t.w; }
        else          { const uint2 t = nt ? stream_load((const uint2*)(g.idx + s)) : *(const uint2*)(g.idx + s); e.id[0] = t.x; e.id[1] = t.y; }
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const size_t o = ((size_t)k * 3 + p) * g.slots + (size_t)tid * EPL;
#pragma unroll
            for (int j = 0; j < EPL; j += 2) {
                const double2 a = nt ? stream_load((const double2*)(u + o + j)) : *(const double2*)(u + o + j);
                const double2 b = nt ? stream_load((const double2*)(v + o + j)) : *(const double2*)(v + o + j);
                e.u[p][j] = a.x; e.u[p][j + 1] = a.y; e.v[p][j] = b.x; e.v[p][j + 1] = b.y;
            }
        }
    };
    double rv[NR];
    auto load_rows = [&](int k) {
        const int r0 = g.chunk_row0[k], n9 = 9 * (g.chunk_row0[k + 1] - r0);
#pragma unroll
        for (int m = 0; m < NR; ++m) { const int i = tid + m * BLOCK; rv[m] = i < n9 ? rt[(size_t)r0 * 9 + i] : 0.0; }
    };
    auto commit_rows = [&](int k, int buf) {
        const int n9 = 9 * (g.chunk_row0[k + 1] - g.chunk_row0[k]);
#pragma unroll
        for (int m = 0; m < NR; ++m) { const int i = tid + m * BLOCK; if (i < n9) rts[buf * mr9 + i] = rv[m]; }
    };

    RhsRegs<EPL> ea, eb;
    if (k0 < k1) { load_edges(ea, k0); load_rows(k0); commit_rows(k0, 0); }
    __syncthreads();

    auto body = [&](RhsRegs<EPL>& cur, RhsRegs<EPL>& nxt, const int k, const int buf) {
        const int r0 = g.chunk_row0[k], nrows = g.chunk_row0[k + 1] - r0;
        load_edges(nxt, k + 1 < k1 ? k + 1 : k);                               // in flight during this chunk (unconditional: exact vmcnt waits)
        if (k + 1 < k1) load_rows(k + 1);
        const double* rtb = rts + buf * mr9;
        double acc[3] = {0, 0, 0};
        uint32_t prow = 0xFFFFFFFFu;
        auto flush_row = [&](const uint32_t r) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const Fix2 f = to_fix2(acc[i], scale, lo_scale);
                u64* a = &gt[(r * 3 + i) * ncopy + lane_copy];
                lds_add_fix(a, f.hi); lds_add_fix(a + lo_t, f.lo);
            }
        };
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            const bool pad = cur.id[j] == VICAN_PAD_SLOT;
            const uint32_t cam = pad ? pad_cam : (cur.id[j] & 0xFFFFu), row = pad ? 0u : (cur.id[j] >> 16);
            if (row != prow) {
                if (prow != 0xFFFFFFFFu) flush_row(prow);
                prow = row; acc[0] = acc[1] = acc[2] = 0.0;
            }
            const double* B = rtb + row * 9;
#pragma unroll
            for (int i = 0; i < 3; ++i) {       // world<-node = transpose of the stored blocks
                const double gi = rcs[(0 * 3 + i) * C + cam] * cur.u[0][j] + rcs[(1 * 3 + i) * C + cam] * cur.u[1][j] +
                                  rcs[(2 * 3 + i) * C + cam] * cur.u[2][j] + B[0 * 3 + i] * cur.v[0][j] +
                                  B[1 * 3 + i] * cur.v[1][j] + B[2 * 3 + i] * cur.v[2][j];
                acc[i] += gi;
                const Fix2 f = to_fix2(-gi, scale, lo_scale);
                lds_add_fix(&gc[i * C + cam], f.hi); lds_add_fix(&gc[lo_c + i * C + cam], f.lo);
            }
        }
        if (prow != 0xFFFFFFFFu) flush_row(prow);
        __syncthreads();
        for (int i = tid; i < 3 * nrows; i += BLOCK) {
            long long sum = 0, slo = 0;
            for (int c = 0; c < ncopy; ++c) {
                const int a = i * ncopy + ((c + i) & cmask);
                sum += (long long)gt[a]; slo += (long long)gt[lo_t + a];
                gt[a] = 0ull; gt[lo_t + a] = 0ull;
            }
            rhs_t[(size_t)r0 * 3 + i] = fix2_value(sum, slo, lob, inv);
        }
        if (k + 1 < k1) commit_rows(k + 1, buf ^ 1);
        __syncthreads();
    };
#pragma unroll 1
    for (int k = k0; k < k1; k += 2) {
        body(ea, eb, k, 0);
        if (k + 1 < k1) body(eb, ea, k + 1, 1);
    }
    for (int i = tid; i < 6 * C; i += BLOCK) rhs_c_part[(size_t)blockIdx.x * 6 * C + i] = gc[i];
}

// fold of double-word slabs [n_slab][2][ncomp][C] -> out [C][ncomp] doubles, each rounded once (fix3_add / fix3_value)
__global__ __launch_bounds__(1024) void fold2_kernel(const long long* __restrict__ part, int n_slab, int n_cam, int ncomp, int lob,
                                                     double inv, double* __restrict__ out) {
    __shared__ long long sh[3][1024];
    const long long n = (long long)ncomp * n_cam;
    const int e = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const long long i = (long long)blockIdx.x * 64 + e;
    Fix3 a = {0, 0, 0};
    if (i < n)
        for (int k = grp; k < n_slab; k += 16) fix3_add(a, part[(size_t)k * 2 * n + i], part[(size_t)k * 2 * n + n + i], lob);
    sh[0][threadIdx.x] = a.top; sh[1][threadIdx.x] = a.bot; sh[2][threadIdx.x] = a.lo;
    __syncthreads();
    if (grp == 0 && i < n) {
        long long t = 0, b = 0, l = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) { t += sh[0][k * 64 + e]; b += sh[1][k * 64 + e]; l += sh[2][k * 64 + e]; }
        const long long q = i / n_cam, cam = i % n_cam;
        out[cam * ncomp + q] = fix3_value(t, b, l, lob, inv);
    }
}

extern "C" __attribute__((visibility("hidden"))) int vican_trans_wrhs(const vican_graph_t* g, const double* u, const double* v, const double* rc,
                                                                      const double* rt, double* rhs_t, void* rhs_c_part, double scale,
                                                                      double inv, int lob, void* stream);   // vican_wtrans.hip
extern "C" int vican_trans_rhs(const vican_graph_t* g, const double* u, const double* v, const double* rc,
                               const double* rt, double* rhs_t, double* rhs_c, void* rhs_c_part, double gmax, double n_add,
                               void* stream) {
    if (int r = vican_check_graph(g, "vican_trans_rhs")) return r;
    if (!u || !v || !rc || !rt || !rhs_t || !rhs_c || !rhs_c_part || !(gmax >= 0))
        return set_err(VICAN_ERR_ARG, "vican_trans_rhs: bad argument");
    double inv;
    const double scale = fix_scale(gmax, n_add, &inv);
    const int lob = fix2_lo_bits(n_add);
    hipStream_t st = (hipStream_t)stream;
    if (g->layout == VICAN_LAYOUT_WAVE) {
        if (int r = vican_trans_wrhs(g, u, v, rc, rt, rhs_t, rhs_c_part, scale, inv, lob, stream)) return r;
    } else {
    const size_t lds = (size_t)rhs_lds_bytes(g->n_cam, g->max_rows, g->n_copy);
    const int epl = g->slots / g->block_threads;
const int nr = (9 * g->max_rows + g->block_threads - 1) / g->block_threads;        // R_t values per thread (<= 9 EPL)
#define RHS_LAUNCH3(B, E, R)                                                                                     \
    do {                                                                                                         \
        auto kern = trans_rhs_kernel<B, E, R>;                                                                   \
        static size_t conf = 0;                                                                                  \
        if (lds > conf) { hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); conf = lds; } \
        hipLaunchKernelGGL(kern, dim3(g->n_wg), dim3(B), lds, st, *g, u, v, rc, rt, rhs_t, (u64*)rhs_c_part, scale, inv, lob); \
    } while (0)
#define RHS_LAUNCH(B, E)                                                                                         \
    do {                                                                                                         \
        if (nr <= 1) RHS_LAUNCH3(B, E, 1); else if (nr <= 4) RHS_LAUNCH3(B, E, 4); else if (nr <= 12) RHS_LAUNCH3(B, E, 12); \
        else RHS_LAUNCH3(B, E, 36);                                                                              \
    } while (0)
    if (g->block_threads == 1024)     { if (epl == 4) RHS_LAUNCH(1024, 4); else RHS_LAUNCH(1024, 2); }
    else if (g->block_threads == 768) { if (epl == 4) RHS_LAUNCH(768, 4);  else RHS_LAUNCH(768, 2); }
    else if (g->block_threads == 512) { if (epl == 4) RHS_LAUNCH(512, 4);  else RHS_LAUNCH(512, 2); }
    else                              { if (epl == 4) RHS_LAUNCH(256, 4);  else RHS_LAUNCH(256, 2); }
#undef RHS_LAUNCH3
#undef RHS_LAUNCH
    }
    const long long n = 3LL * g->n_cam;
    hipLaunchKernelGGL(fold2_kernel, dim3((unsigned)((n + 63) / 64)), dim3(1024), 0, st, (const long long*)rhs_c_part, g->n_wg,
                       g->n_cam, 3, lob, inv, rhs_c);
    LAUNCH_CHECK("vican_trans_rhs");
    return VICAN_OK;
}

// ---------------------------------------------------------------------------
// conjugate gradients
// ---------------------------------------------------------------------------
// Jacobi scaling of the normal equations (tight translation solve, see vican_hip.h)
// ---------------------------------------------------------------------------
__global__ void jacobi_scale_kernel(int n, const double* __restrict__ deg, double* __restrict__ s) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) s[i] = deg[i] > 0.0 ? 1.0 / sqrt(deg[i]) : 0.0;
}
extern "C" int vican_jacobi_scale(int32_t n, const double* deg, double* s, void* stream) {
    if (n < 0 || !deg || !s) return set_err(VICAN_ERR_ARG, "vican_jacobi_scale: bad argument");
    if (n == 0) return VICAN_OK;
    hipLaunchKernelGGL(jacobi_scale_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, deg, s);
    LAUNCH_CHECK("vican_jacobi_scale");
    return VICAN_OK;
}
__global__ void row_scale_kernel(long long n, int ncomp, const double* __restrict__ s, double* __restrict__ x) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n * ncomp) x[i] *= s[i / ncomp];
}
extern "C" int vican_row_scale(int32_t n, int32_t ncomp, const double* s, double* x, void* stream) {
    if (n < 0 || ncomp <= 0 || !s || !x) return set_err(VICAN_ERR_ARG, "vican_row_scale: bad argument");
    if (n == 0) return VICAN_OK;
    const long long tot = (long long)n * ncomp;
    hipLaunchKernelGGL(row_scale_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (long long)n,
                       ncomp, s, x);
    LAUNCH_CHECK("vican_row_scale");
    return VICAN_OK;
}
__global__ void scale_weights_kernel(vican_graph_t g, const double* __restrict__ w, const double* __restrict__ s_cam,
                                     const double* __restrict__ s_row, double* __restrict__ w_out) {
    const int k = blockIdx.y;
    const int sl = blockIdx.x * blockDim.x + threadIdx.x;
    if (sl >= g.slots) return;
    const size_t i = (size_t)k * g.slots + sl, i8 = (size_t)k * g.slots + slot_pos8(g, sl);
    const uint32_t id = g.idx[i];
    w_out[i8] = id == VICAN_PAD_SLOT ? 0.0 : w[i8] * s_cam[id & 0xFFFFu] * s_row[g.chunk_row0[k] + (int)(id >> 16)];
}
extern "C" int vican_scale_weights(const vican_graph_t* g, const double* w, const double* s_cam, const double* s_row,
                                   double* w_out, void* stream) {
    if (int r = vican_check_graph(g, "vican_scale_weights")) return r;
    if (!w || !s_cam || !s_row || !w_out) return set_err(VICAN_ERR_ARG, "vican_scale_weights: null pointer");
    if (g->n_chunk == 0) return VICAN_OK;
    hipLaunchKernelGGL(scale_weights_kernel, dim3((g->slots + 255) / 256, g->n_chunk), dim3(256), 0, (hipStream_t)stream, *g, w,
                       s_cam, s_row, w_out);
    LAUNCH_CHECK("vican_scale_weights");
    return VICAN_OK;
}

// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cg_init_kernel(int n_cam, int n_time, const double* __restrict__ b_c,
                                                      const double* __restrict__ b_t, double* x_c, double* x_t,
                                                      double* r_c, double* r_t, double* p_c, double* p_t,
                                                      double* __restrict__ part) {
    __shared__ double red[8];
    const long long n = 3LL * n_time, nc = 3LL * n_cam;
    double s = 0.0, m = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const double b = b_t[i];
        x_t[i] = 0.0; r_t[i] = b; p_t[i] = b; s += b * b; m = fmax(m, fabs(b));
    }
    if (blockIdx.x == 0)
        for (long long i = threadIdx.x; i < nc; i += 256) { const double b = b_c[i]; x_c[i] = 0.0; r_c[i] = b; p_c[i] = b; }
    const double t = block_sum(s, red);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_down(m, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        part[blockIdx.x] = t;
        part[CG_PARTS + blockIdx.x] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
    }
}
__global__ void cg_init_finish_kernel(int n_cam, const double* __restrict__ b_c, const double* __restrict__ part,
                                      int n_part, vican_cg_state_t* st, double wmax) {
    __shared__ double red[8];
    double s = 0.0, m = 0.0;
    for (int i = threadIdx.x; i < 3 * n_cam; i += blockDim.x) { s += b_c[i] * b_c[i]; m = fmax(m, fabs(b_c[i])); }
    const double rc = block_sum(s, red);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_down(m, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    // (the partials through LDS: 512 dependent global loads by one thread took 36 us; same summation order)
    __shared__ double sp[2 * CG_PARTS];
    for (int i = threadIdx.x; i < n_part; i += blockDim.x) { sp[i] = part[i]; sp[CG_PARTS + i] = part[CG_PARTS + i]; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0, mt = 0.0, mc = 0.0;
        for (int i = 0; i < n_part; ++i) { t += sp[i]; mt = fmax(mt, sp[CG_PARTS + i]); }
        for (int i = 0; i < (int)(blockDim.x >> 6); ++i) mc = fmax(mc, red[i]);
        st->rho = 0; st->rho_prev = 0; st->pq = 0; st->alpha = 0; st->beta = 0; st->bnorm2 = 0; st->atol2 = 0;
        st->rr_cam = rc; st->pq_time = 0; st->rr_time = t;
        st->rmax_cam = mc; st->rmax_time = mt; st->pmax = 0; st->qscale = 1; st->qinv = 1; st->wmax = wmax;
        st->iter = 0; st->done = 0; st->first = 1; st->lo_bits = 48;
    }
}
extern "C" int vican_cg_init(int32_t n_cam, int32_t n_time, const double* b_c, const double* b_t, double* x_c,
                             double* x_t, double* r_c, double* r_t, double* p_c, double* p_t, vican_cg_state_t* st,
                             double* ws /* >= 2*CG_PARTS doubles */, double wmax, void* stream) {
    if (n_cam <= 0 || n_time < 0 || !b_c || !b_t || !x_c || !x_t || !r_c || !r_t || !p_c || !p_t || !st || !ws)
        return set_err(VICAN_ERR_ARG, "vican_cg_init: bad argument");
    long long n = 3LL * n_time;
    int nb = (int)((n + 255) / 256); if (nb < 1) nb = 1; if (nb > CG_PARTS) nb = CG_PARTS;
    hipLaunchKernelGGL(cg_init_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, n_cam, n_time, b_c, b_t, x_c, x_t,
                       r_c, r_t, p_c, p_t, ws);
    hipLaunchKernelGGL(cg_init_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, n_cam, b_c, ws, nb, st, wmax);
    LAUNCH_CHECK("vican_cg_init");
    return VICAN_OK;
}

__device__ __forceinline__ void cg_close_iteration(const double* __restrict__ rr_part, int n_part, vican_cg_state_t* st) {
    double t = 0.0, m = 0.0, mp = 0.0;
    for (int i = 0; i < n_part; ++i) { t += rr_part[i]; m = fmax(m, rr_part[CG_PARTS + i]); mp = fmax(mp, rr_part[2 * CG_PARTS + i]); }
    st->rr_time = t; st->rmax_time = m; st->pmax_time = mp; st->iter += 1; st->rho_prev = st->rho; st->first = 0;
}

// ---------------------------------------------------------------------------
// Camera-side loops of the CG kernels: 3C (<= 3072 untiled) elements by ONE workgroup of 256 threads, thread t takes elements
// t, t + 256, ... in that order.  Written as plain loops they ran one element per memory round trip - a store into x_c / r_c /
// p_c may alias the next element's operands, so the compiler cannot hoist the loads, and 12 dependent L2 round trips made
// cg_begin 9.4 us and the camera workgroup of cg_step 8 us on the stress graph.  Here the operands of CG_CAMK elements are loaded
// into registers before the first use: same expressions, same per-thread order of the floating-point sums, same bits.
// ---------------------------------------------------------------------------
#define CG_CAMK 12
// this thread's part of p_c . q_c, q_c = deg_c p_c - qc  (cg_cam_step / cg_step / cg_fold2)
__device__ __forceinline__ double cg_cam_dot(int nc, const double* __restrict__ deg_c, const double* __restrict__ p_c, const double* qc) {
    double s = 0.0;
    for (int base = 0; base < nc; base += 256 * CG_CAMK) {
        double pv[CG_CAMK], dv[CG_CAMK], qv[CG_CAMK];
#pragma unroll
        for (int k = 0; k < CG_CAMK; ++k) {
            const int i = base + (int)threadIdx.x + 256 * k;
            const bool in = i < nc;
            pv[k] = in ? p_c[i] : 0.0; dv[k] = in ? deg_c[i / 3] : 0.0; qv[k] = in ? qc[i] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < CG_CAMK; ++k) {
            const int i = base + (int)threadIdx.x + 256 * k;
            if (i < nc) { const double q = dv[k] * pv[k] - qv[k]; s += pv[k] * q; }
        }
    }
    return s;
}
// x_c += alpha p_c, r_c -= alpha q_c; this thread's parts of r_c . r_c and max |r_c|
__device__ __forceinline__ void cg_cam_update(int nc, double alpha, const double* __restrict__ deg_c, const double* __restrict__ qc,
                                              const double* __restrict__ p_c, double* x_c, double* r_c, double& rr, double& mx) {
    for (int base = 0; base < nc; base += 256 * CG_CAMK) {
        double pv[CG_CAMK], dv[CG_CAMK], qv[CG_CAMK], xv[CG_CAMK], rv[CG_CAMK];
#pragma unroll
        for (int k = 0; k < CG_CAMK; ++k) {
            const int i = base + (int)threadIdx.x + 256 * k;
            const bool in = i < nc;
            pv[k] = in ? p_c[i] : 0.0; dv[k] = in ? deg_c[i / 3] : 0.0; qv[k] = in ? qc[i] : 0.0; xv[k] = in ? x_c[i] : 0.0; rv[k] = in ? r_c[i] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < CG_CAMK; ++k) {
            const int i = base + (int)threadIdx.x + 256 * k;
            if (i < nc) {
                const double p = pv[k];
                const double q = dv[k] * p - qv[k];
                x_c[i] = mul_add_2r(alpha, p, xv[k]);
                const double r = mul_add_2r(-alpha, q, rv[k]);
                r_c[i] = r;
                rr += r * r; mx = fmax(mx, fabs(r));
            }
        }
    }
}

// top of an iteration: (optionally close the previous one), convergence test, beta, p_c update,
// and the fixed-point scale of this iteration's sweep from the running bound on max|p|
__device__ __forceinline__ void cg_begin_body(int n_cam, const double* __restrict__ r_c, double* p_c, double rtol,
                                              const double* __restrict__ rr_part, int n_part, double n_add, vican_cg_state_t* st,
                                              double* red /* [8] */, double* sh_beta, int* sh_go) {
    // close the previous iteration: fixed-order block reduction of the partials (sum, max |r_t|, max |p_t|)
    double ps = 0.0, pm = 0.0, pp = 0.0;
    for (int i = threadIdx.x; i < n_part; i += 256) {
        ps += rr_part[i]; pm = fmax(pm, rr_part[CG_PARTS + i]); pp = fmax(pp, rr_part[2 * CG_PARTS + i]);
    }
    const double tsum = block_sum(ps, red);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { pm = fmax(pm, __shfl_down(pm, o, 64)); pp = fmax(pp, __shfl_down(pp, o, 64)); }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = pm; red[4 + (threadIdx.x >> 6)] = pp; }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (n_part > 0) {
            st->rr_time = tsum; st->rmax_time = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
            st->pmax_time = fmax(fmax(red[4], red[5]), fmax(red[6], red[7]));
            st->iter += 1; st->rho_prev = st->rho; st->first = 0;
        }
        const double rho = st->rr_cam + st->rr_time;
        if (st->iter == 0 && st->first) { st->bnorm2 = rho; st->atol2 = rtol * rtol * rho; }
        st->rho = rho;
        int go = 1;
        if (sqrt(rho) < sqrt(st->atol2) || rho == 0.0) { st->done = 1; go = 0; }    // scipy: norm(r) < atol
        else if (!(rho == rho)) { st->done = -2; go = 0; }                          // NaN: scipy would spin to maxiter; stop and report
        double beta = 0.0;
        if (go && !st->first) beta = rho / st->rho_prev;
        st->beta = beta;
        *sh_beta = beta; *sh_go = go && !st->first;
    }
    __syncthreads();
    // p_c = r_c + beta p_c (p = r on the first iteration) and its exact maximum
    const double beta = *sh_beta;
    const int go = *sh_go;
    double mc = 0.0;
    for (int base = 0; base < 3 * n_cam; base += 256 * CG_CAMK) {          // (operands staged: see cg_cam_dot)
        double pv[CG_CAMK], rv[CG_CAMK];
#pragma unroll
        for (int k = 0; k < CG_CAMK; ++k) {
            const int i = base + (int)threadIdx.x + 256 * k;
            const bool in = i < 3 * n_cam;
            pv[k] = in ? p_c[i] : 0.0; rv[k] = (in && go) ? r_c[i] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < CG_CAMK; ++k) {
            const int i = base + (int)threadIdx.x + 256 * k;
            if (i < 3 * n_cam) {
                double p = pv[k];
                if (go) { p = mul_add_2r(beta, p, rv[k]); p_c[i] = p; }
                mc = fmax(mc, fabs(p));
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mc = fmax(mc, __shfl_down(mc, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mc;
    __syncthreads();
    if (threadIdx.x == 0) {
        // Fixed-point scale of this iteration's sweep: the hi word sits 49 bits below a bound on max |w p| (a lane pre-sums up
        // to four same-row contributions: 4 * 2^49 < 2^51, the range of the magic-number conversion), the lo word carries
        // lo_bits more (to_fix2, vican_sweep_common.h).  The bound uses the cameras' exact max |p_c| and, for the timestep
        // side (updated inside the sweep), |p_new| <= max|r_t| + beta * max|p_t| with the MEASURED max |p_t| of the current
        // iterate (cg_step).  History: one accumulator 47 bits below a compounding bound stopped the large_shop-scale
        // golden at 118 iterations, 49 bits below this tight bound at 111, scipy's own window is 101-106 (a Laplacian
        // product is a difference of nearly equal terms, and terms far below the global bound lose their bits): hence two words.
        const double pc = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
        const double pt = st->first ? st->rmax_time : st->rmax_time + beta * st->pmax_time;
        st->pmax = fmax(pc, pt);
        double inv;
        st->qscale = fix_scale(st->wmax * st->pmax, n_add, &inv, 49);
        st->lo_bits = fix2_lo_bits(n_add);
        st->qinv = inv;
    }
}
__global__ __launch_bounds__(256) void cg_begin_kernel(int n_cam, const double* __restrict__ r_c, double* p_c,
                                                       double rtol, const double* __restrict__ rr_part, int n_part,
                                                       double n_add, vican_cg_state_t* st) {
    __shared__ double sh_beta;
    __shared__ int sh_go;
    __shared__ double red[8];
    if (st->done) return;
    cg_begin_body(n_cam, r_c, p_c, rtol, rr_part, n_part, n_add, st, red, &sh_beta, &sh_go);
}
extern "C" int vican_cg_begin(int32_t n_cam, const double* r_c, double* p_c, double rtol, const double* rr_part,
                              int32_t n_part, double n_add, vican_cg_state_t* st, void* stream) {
    if (n_cam <= 0 || !r_c || !p_c || !st || (n_part > 0 && !rr_part)) return set_err(VICAN_ERR_ARG, "vican_cg_begin: bad argument");
    hipLaunchKernelGGL(cg_begin_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, n_cam, r_c, p_c, rtol, rr_part,
                       n_part, n_add, st);
    LAUNCH_CHECK("vican_cg_begin");
    return VICAN_OK;
}

// Timestep-major Laplacian sweep (12 algorithmic bytes per edge: weight + packed index).
// Software pipeline: while chunk k is processed, the edge words of chunk k+1 and the p_t / r_t / deg_t values
// of its rows are already in flight (registers); they are committed to the second staging buffer in the fold
// phase.  Two barriers per chunk and no global-memory latency on the critical path (the first version staged
// each chunk's rows behind a barrier: ~3.6 us per chunk of pure latency, 117 us per launch on the stress graph).
template <int EPL>
struct CgRegs { double w[EPL]; uint32_t id[EPL]; };

template <int BLOCK, int EPL, int NR>
__global__ __launch_bounds__(BLOCK) void cg_sweep_kernel(vican_graph_t g, const double* __restrict__ w,
                                                         const double* __restrict__ deg_t,
                                                         const double* __restrict__ p_c, const double* __restrict__ r_t,
                                                         double* __restrict__ p_t, double* __restrict__ q_t,
                                                         u64* __restrict__ qc_part, double* __restrict__ pq_part,
                                                         const vican_cg_state_t* __restrict__ st, const int partial) {
    // partial != 0 (camera-tiled graphs, vican_cg_sweep_partial): g holds the edges of ONE camera tile; p_t is read as it is
    // (already updated), q_t receives the tile's row sums sum_{c in tile} w p_c alone and no p.q partial is formed - the
    // caller combines the tiles (vican_cg_combine_rows).  The camera sums of the tile's cameras are complete either way.
    extern __shared__ __align__(16) unsigned char lds_raw[];
    if (st->done) return;
    const int C = g.n_cam, ncopy = g.n_copy, cmask = ncopy - 1, mr3 = 3 * g.max_rows;
    u64* qc = (u64*)lds_raw;                               // [2][3][C] planes: hi words, lo words
    u64* qt = qc + 6 * C;                                  // [2 buffers][2 words][max_rows*3][ncopy]
    double* pcs = (double*)(qt + (size_t)4 * mr3 * ncopy); // [3][C] planes
    double* pts = pcs + 3 * C;                             // [2][max_rows*3]   p of the chunk's rows
    double* dps = pts + 2 * mr3;                           // [2][max_rows*3]   deg * p
    double* red = dps + 2 * mr3;                           // [16]
    const int tid = threadIdx.x, lane_copy = tid & cmask;
    const uint32_t pad_cam = (uint32_t)((tid & 31) < g.n_cam ? (tid & 31) : 0);   // padding slots: a valid camera, zero weight
    const bool upd = !st->first && !partial;
    const double beta = st->beta, scale = st->qscale, inv = st->qinv;
    const int lob = st->lo_bits;
    const double lo_scale = ldexp(1.0, lob);
    const int lo_c = 3 * C, lo_t = mr3 * ncopy;            // offsets of the lo planes behind the hi planes
    for (int i = tid; i < 3 * C; i += BLOCK) { pcs[(i % 3) * C + i / 3] = p_c[i]; qc[i] = 0ull; qc[lo_c + i] = 0ull; }
    for (int i = tid; i < 4 * mr3 * ncopy; i += BLOCK) qt[i] = 0ull;
    const int k0 = (int)(((long long)blockIdx.x * g.n_chunk) / gridDim.x);
    const int k1 = (int)(((long long)(blockIdx.x + 1) * g.n_chunk) / gridDim.x);

    // (edge words: non-temporal loads when the stream cannot stay cached between two CG iterations - g.stream_nt)
    auto load_edges = [&](CgRegs<EPL>& e, int k) {
        const size_t s = (size_t)k * g.slots + (size_t)tid * EPL;
        if (EPL == 4) {
            uint4 t; double2 a, b;
            if (g.stream_nt) { t = stream_load((const uint4*)(g.idx + s)); a = stream_load((const double2*)(w + s)); b = stream_load((const double2*)(w + s + 2)); }
            else { t = *(const uint4*)(g.idx + s); a = *(const double2*)(w + s); b = *(const double2*)(w + s + 2); }
            e.id[0] = t.x; e.id[1] = t.y; e.id[2] = t.z; e.id[3] = t.w;
            e.w[0] = a.x; e.w[1] = a.y; e.w[2] = b.x; e.w[3] = b.y;
        } else {
            uint2 t; double2 a;
            if (g.stream_nt) { t = stream_load((const uint2*)(g.idx + s)); a = stream_load((const double2*)(w + s)); }
            else { t = *(const uint2*)(g.idx + s); a = *(const double2*)(w + s); }
            e.id[0] = t.x; e.id[1] = t.y; e.w[0] = a.x; e.w[1] = a.y;
        }
    };
    // rows of chunk k: p (updated p = r + beta p on all but the first iteration) and deg
    struct RowRegs { double p[NR], d[NR]; };
    auto load_rows = [&](RowRegs& rr, int k) {
        const int r0 = g.chunk_row0[k], n3 = 3 * (g.chunk_row0[k + 1] - r0);
#pragma unroll
        for (int m = 0; m < NR; ++m) {
            const int i = tid + m * BLOCK;
            rr.p[m] = 0.0; rr.d[m] = 0.0;
            if (i < n3) {
                const size_t gi = (size_t)r0 * 3 + i;
                double p = p_t[gi];
                if (upd) p = mul_add_2r(beta, p, r_t[gi]);
                rr.p[m] = p; rr.d[m] = deg_t[r0 + i / 3];
            }
        }
    };
    auto commit_rows = [&](const RowRegs& rr, int k, int buf) {
        const int r0 = g.chunk_row0[k], n3 = 3 * (g.chunk_row0[k + 1] - r0);
#pragma unroll
        for (int m = 0; m < NR; ++m) {
            const int i = tid + m * BLOCK;
            if (i < n3) {
                pts[buf * mr3 + i] = rr.p[m]; dps[buf * mr3 + i] = rr.d[m] * rr.p[m];
                if (upd) p_t[(size_t)r0 * 3 + i] = rr.p[m];
            }
        }
    };

    // One chunk of edge words and row values is in flight ahead of the chunk being processed (a second one was
    // measured: no gain on the stress graph, slower on sparse graphs).  The fold of a chunk's row accumulators
    // (a few threads, a chain of LDS reads and a global store - nothing in the chunk depends on it) is deferred
    // into the next chunk's body, where it overlaps with everybody's edge work: ONE barrier per chunk; the row
    // accumulators are double-buffered for that.
    // The row values are requested TWO bodies ahead: with one barrier per chunk a body lasts about as long as a
    // global-memory round trip, and the commit at its end would wait for loads issued at its start.
    CgRegs<EPL> ea, eb;
    RowRegs ra, rb;
    double pq = 0.0;
    if (k0 < k1) { load_edges(ea, k0); load_rows(ra, k0); commit_rows(ra, k0, 0); }
    if (k0 + 1 < k1) load_rows(ra, k0 + 1);
    __syncthreads();

    auto fold = [&](const int k, const int buf) {
        const int r0 = g.chunk_row0[k], nrows = g.chunk_row0[k + 1] - r0;
        u64* q = qt + (size_t)buf * 2 * lo_t;
        for (int i = tid; i < 3 * nrows; i += BLOCK) {
            long long sum = 0, slo = 0;
            for (int c = 0; c < ncopy; ++c) {
                const int a = i * ncopy + ((c + i) & cmask);
                sum += (long long)q[a]; slo += (long long)q[lo_t + a];
                q[a] = 0ull; q[lo_t + a] = 0ull;
            }
            const double acc_ = fix2_value(sum, slo, lob, inv);
            const double qv = partial ? acc_ : dps[buf * mr3 + i] - acc_;
            q_t[(size_t)r0 * 3 + i] = qv;
            pq += pts[buf * mr3 + i] * qv;
        }
    };
    // body(k): edges `cur`; `rx` holds the rows of chunk k+1 (committed at the end), `ry` receives those of k+2
    auto body = [&](CgRegs<EPL>& cur, CgRegs<EPL>& nxt, RowRegs& rx, RowRegs& ry, const int k, const int buf) {
        load_edges(nxt, k + 1 < k1 ? k + 1 : k);                               // in flight during this chunk (unconditional: exact vmcnt waits)
        if (k + 2 < k1) load_rows(ry, k + 2);
        const double* pt = pts + buf * mr3;
        u64* qtb = qt + (size_t)buf * 2 * lo_t;
        double acc[3] = {0, 0, 0};
        uint32_t prow = 0xFFFFFFFFu;
        auto flush_row = [&](const uint32_t r) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const Fix2 f = to_fix2(acc[i], scale, lo_scale);
                u64* a = &qtb[(r * 3 + i) * ncopy + lane_copy];
                lds_add_fix(a, f.hi); lds_add_fix(a + lo_t, f.lo);
            }
        };
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            const bool pad = cur.id[j] == VICAN_PAD_SLOT;
            const uint32_t cam = pad ? pad_cam : (cur.id[j] & 0xFFFFu), row = pad ? 0u : (cur.id[j] >> 16);
            const double wj = pad ? 0.0 : cur.w[j];
            if (row != prow) {
                if (prow != 0xFFFFFFFFu) flush_row(prow);
                prow = row; acc[0] = acc[1] = acc[2] = 0.0;
            }
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                acc[i] += wj * pcs[i * C + cam];
                const Fix2 f = to_fix2(wj * pt[row * 3 + i], scale, lo_scale);
                lds_add_fix(&qc[i * C + cam], f.hi); lds_add_fix(&qc[lo_c + i * C + cam], f.lo);
            }
        }
        if (prow != 0xFFFFFFFFu) flush_row(prow);
        // previous chunk's fold, then the next chunk's rows into the staging buffer the fold has just read
        // (same index -> same thread in both loops, so no cross-thread hazard on that buffer)
        if (k > k0) fold(k - 1, buf ^ 1);
        if (k + 1 < k1) commit_rows(rx, k + 1, buf ^ 1);
        __syncthreads();
    };
#pragma unroll 1
    for (int k = k0; k < k1; k += 2) {
        body(ea, eb, ra, rb, k, 0);
        if (k + 1 < k1) body(eb, ea, rb, ra, k + 1, 1);
    }
    if (k0 < k1) fold(k1 - 1, (k1 - 1 - k0) & 1);
    __syncthreads();
    for (int i = tid; i < 6 * C; i += BLOCK) qc_part[(size_t)blockIdx.x * 6 * C + i] = qc[i];
    const double t = block_sum(pq, red);
    if (tid == 0 && !partial) pq_part[blockIdx.x] = t;
}
extern "C" __attribute__((visibility("hidden"))) int vican_cg_wsweep(const vican_graph_t* g, const double* w, const double* deg_t,
                                                                     const double* p_c, const double* r_t, double* p_t, double* q_t,
                                                                     void* qc_part, double* pq_part, const vican_cg_state_t* st,
                                                                     void* stream, int partial);       // vican_wtrans.hip
static int cg_sweep_launch(const vican_graph_t* g, const double* w, const double* deg_t, const double* p_c,
                           const double* r_t, double* p_t, double* q_t, void* qc_part, double* pq_part,
                           const vican_cg_state_t* st, void* stream, const int partial) {
    if (int r = vican_check_graph(g, "vican_cg_sweep")) return r;
    if (!w || !deg_t || !p_c || !r_t || !p_t || !q_t || !qc_part || !pq_part || !st)
        return set_err(VICAN_ERR_ARG, "vican_cg_sweep: null pointer");
    if (g->layout == VICAN_LAYOUT_WAVE) {
        return vican_cg_wsweep(g, w, deg_t, p_c, r_t, p_t, q_t, qc_part, pq_part, st, stream, partial);
    }
    const size_t lds = (size_t)cg_lds_bytes(g->n_cam, g->max_rows, g->n_copy);
    const int epl = g->slots / g->block_threads;
    const int nr = (3 * g->max_rows + g->block_threads - 1) / g->block_threads;   // row values per thread (<= 3 EPL)
    hipStream_t s = (hipStream_t)stream;
#define CG_LAUNCH3(B, E, R)                                                                                      \
    do {                                                                                                         \
        auto kern = cg_sweep_kernel<B, E, R>;                                                                    \
        static size_t conf = 0;                                                                                  \
        if (lds > conf) { hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); conf = lds; } \
        hipLaunchKernelGGL(kern, dim3(g->n_wg), dim3(B), lds, s, *g, w, deg_t, p_c, r_t, p_t, q_t, (u64*)qc_part, pq_part, st, partial); \
    } while (0)
#define CG_LAUNCH(B, E)                                                                                          \
    do {                                                                                                         \
        if (nr <= 1) CG_LAUNCH3(B, E, 1); else if (nr <= 4) CG_LAUNCH3(B, E, 4); else CG_LAUNCH3(B, E, 12);      \
    } while (0)
    if (nr > 12) return set_err(VICAN_ERR_CAPACITY, "vican_cg_sweep: more than 4 rows per lane in a chunk");
    if (g->block_threads == 1024)     { if (epl == 4) CG_LAUNCH(1024, 4); else CG_LAUNCH(1024, 2); }
    else if (g->block_threads == 768) { if (epl == 4) CG_LAUNCH(768, 4);  else CG_LAUNCH(768, 2); }
    else if (g->block_threads == 512) { if (epl == 4) CG_LAUNCH(512, 4);  else CG_LAUNCH(512, 2); }
    else                              { if (epl == 4) CG_LAUNCH(256, 4);  else CG_LAUNCH(256, 2); }
#undef CG_LAUNCH
#undef CG_LAUNCH3
    LAUNCH_CHECK("vican_cg_sweep");
    return VICAN_OK;
}
extern "C" int vican_cg_sweep(const vican_graph_t* g, const double* w, const double* deg_t, const double* p_c,
                              const double* r_t, double* p_t, double* q_t, void* qc_part, double* pq_part,
                              const vican_cg_state_t* st, void* stream) {
    return cg_sweep_launch(g, w, deg_t, p_c, r_t, p_t, q_t, qc_part, pq_part, st, stream, 0);
}

// ---- camera-tiled graphs (more cameras than one LDS table holds): the product q = A p tile by tile ----------------------
// p_t <- r_t + beta p_t (not on the first iteration): the update the untiled sweep performs while it loads its rows
__global__ void cg_update_pt_kernel(long long n, const double* __restrict__ r_t, double* __restrict__ p_t, const vican_cg_state_t* __restrict__ st) {
    if (st->done || st->first) return;
    const double beta = st->beta;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p_t[i] = mul_add_2r(beta, p_t[i], r_t[i]);
}
extern "C" int vican_cg_update_pt(int32_t n_time, const double* r_t, double* p_t, const vican_cg_state_t* st, void* stream) {
    if (n_time < 0 || !r_t || !p_t || !st) return set_err(VICAN_ERR_ARG, "vican_cg_update_pt: bad argument");
    const long long n = 3LL * n_time;
    if (n == 0) return VICAN_OK;
    long long nb = (n + 255) / 256; if (nb > 2048) nb = 2048;
    hipLaunchKernelGGL(cg_update_pt_kernel, dim3((unsigned)nb), dim3(256), 0, (hipStream_t)stream, n, r_t, p_t, st);
    LAUNCH_CHECK("vican_cg_update_pt");
    return VICAN_OK;
}
// one camera tile: acc_t [T][3] = sum_{c in tile} w_ct p_c (exact double-word sums, rounded once), qc_part = the
// double-word slabs of sum_t w_ct p_t for the tile's cameras (complete: every edge of a camera lies in its tile; fold with
// vican_cg_fold(pq_part = NULL)).  p_c: the tile's slice of the camera vector; p_t: already updated (vican_cg_update_pt).
extern "C" int vican_cg_sweep_partial(const vican_graph_t* g, const double* w, const double* p_c, const double* p_t, double* acc_t,
                                      void* qc_part, const vican_cg_state_t* st, void* stream) {
    if (!acc_t) return set_err(VICAN_ERR_ARG, "vican_cg_sweep_partial: null pointer");
    // (deg_t, r_t and pq_part are not touched in partial mode: any valid pointers)
    return cg_sweep_launch(g, w, p_t, p_c, p_t, const_cast<double*>(p_t), acc_t, qc_part, acc_t, st, stream, 1);
}
// q_t = deg_t p_t - sum over the tiles of acc (tile order: deterministic), pq_part[block] = partial p_t.q_t
__global__ __launch_bounds__(256) void cg_combine_rows_kernel(long long n, int n_tile, long long tile_stride, const double* __restrict__ deg_t,
                                                              const double* __restrict__ p_t, const double* __restrict__ acc,
                                                              double* __restrict__ q_t, double* __restrict__ pq_part,
                                                              const vican_cg_state_t* __restrict__ st) {
    __shared__ double red[8];
    if (st->done) return;
    double pq = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        double s = 0.0;
        for (int k = 0; k < n_tile; ++k) s += acc[(size_t)k * tile_stride + i];
        const double p = p_t[i], q = deg_t[i / 3] * p - s;
        q_t[i] = q;
        pq += p * q;
    }
    const double t = block_sum(pq, red);
    if (threadIdx.x == 0) pq_part[blockIdx.x] = t;
}
extern "C" int vican_cg_combine_rows(int32_t n_time, int32_t n_tile, int64_t tile_stride, const double* deg_t, const double* p_t,
                                     const double* acc, double* q_t, double* pq_part, int32_t part_cap, const vican_cg_state_t* st,
                                     void* stream) {
    if (n_time < 0 || n_tile <= 0 || !deg_t || !p_t || !acc || !q_t || !pq_part || part_cap < 1 || !st)
        return set_err(VICAN_ERR_ARG, "vican_cg_combine_rows: bad argument");
    const long long n = 3LL * n_time;
    int nb = (int)((n + 1023) / 1024); if (nb < 1) nb = 1; if (nb > part_cap) nb = part_cap; if (nb > 1024) nb = 1024;
    hipLaunchKernelGGL(cg_combine_rows_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, n, (int)n_tile, (long long)tile_stride, deg_t, p_t,
                       acc, q_t, pq_part, st);
    LAUNCH_CHECK("vican_cg_combine_rows");
    return nb;
}

__global__ void cg_reduce_pq_kernel(const double* __restrict__ pq_part, int n_part, double* __restrict__ out,
                                    const vican_cg_state_t* __restrict__ st) {
    if (st->done) return;
    __shared__ double sp[1024];                 // partials through LDS (loaded in parallel), summed in the same fixed order
    double t = 0.0;
    for (int k0 = 0; k0 < n_part; k0 += 1024) {
        const int m = n_part - k0 < 1024 ? n_part - k0 : 1024;
        for (int i = threadIdx.x; i < m; i += blockDim.x) sp[i] = pq_part[k0 + i];
        __syncthreads();
        if (threadIdx.x == 0) for (int i = 0; i < m; ++i) t += sp[i];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = t;
}
extern "C" int vican_cg_reduce_pq(const double* pq_part, int32_t n_part, double* out, const vican_cg_state_t* st,
                                  void* stream) {
    if (!pq_part || n_part <= 0 || !st || !out) return set_err(VICAN_ERR_ARG, "vican_cg_reduce_pq: bad argument");
    hipLaunchKernelGGL(cg_reduce_pq_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, pq_part, n_part, out, st);
    LAUNCH_CHECK("vican_cg_reduce_pq");
    return VICAN_OK;
}

__global__ __launch_bounds__(256) void cg_cam_step_kernel(int n_cam, const double* __restrict__ deg_c,
                                                          const double* __restrict__ qc_sum,
                                                          const double* __restrict__ pq_time,
                                                          const double* __restrict__ p_c, double* x_c, double* r_c,
                                                          vican_cg_state_t* st) {
    __shared__ double red[8];
    __shared__ double sh_alpha;
    if (st->done) return;
    const int n = 3 * n_cam;
    const double s = cg_cam_dot(n, deg_c, p_c, qc_sum);
    const double pqc = block_sum(s, red);
    if (threadIdx.x == 0) {
        const double pq = *pq_time + pqc;
        st->pq_time = *pq_time;
        st->pq = pq;
        st->alpha = st->rho / pq;
        sh_alpha = st->alpha;
    }
    __syncthreads();
    const double alpha = sh_alpha;
    double rr = 0.0, m = 0.0;
    cg_cam_update(n, alpha, deg_c, qc_sum, p_c, x_c, r_c, rr, m);
    const double t = block_sum(rr, red);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_down(m, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) { st->rr_cam = t; st->rmax_cam = fmax(fmax(red[0], red[1]), fmax(red[2], red[3])); }
}
extern "C" int vican_cg_cam_step(int32_t n_cam, const double* deg_c, const double* qc_sum, const double* pq_time,
                                 const double* p_c, double* x_c, double* r_c, vican_cg_state_t* st, void* stream) {
    if (n_cam <= 0 || !deg_c || !qc_sum || !pq_time || !p_c || !x_c || !r_c || !st)
        return set_err(VICAN_ERR_ARG, "vican_cg_cam_step: bad argument");
    hipLaunchKernelGGL(cg_cam_step_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, n_cam, deg_c, qc_sum, pq_time,
                       p_c, x_c, r_c, st);
    LAUNCH_CHECK("vican_cg_cam_step");
    return VICAN_OK;
}

__global__ __launch_bounds__(256) void cg_time_step_kernel(long long n, const double* __restrict__ p_t,
                                                           const double* __restrict__ q_t, double* x_t, double* r_t,
                                                           double* __restrict__ rr_part,
                                                           const vican_cg_state_t* __restrict__ st) {
    __shared__ double red[8];
    if (st->done) return;
    const double alpha = st->alpha;
    double rr = 0.0, m = 0.0, mp = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const double pv = p_t[i];
        x_t[i] = mul_add_2r(alpha, pv, x_t[i]);
        mp = fmax(mp, fabs(pv));
        const double r = mul_add_2r(-alpha, q_t[i], r_t[i]);
        r_t[i] = r;
        rr += r * r; m = fmax(m, fabs(r));
    }
    const double t = block_sum(rr, red);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { m = fmax(m, __shfl_down(m, o, 64)); mp = fmax(mp, __shfl_down(mp, o, 64)); }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = m; red[4 + (threadIdx.x >> 6)] = mp; }
    __syncthreads();
    if (threadIdx.x == 0) { rr_part[blockIdx.x] = t; rr_part[CG_PARTS + blockIdx.x] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
                            rr_part[2 * CG_PARTS + blockIdx.x] = fmax(fmax(red[4], red[5]), fmax(red[6], red[7])); }
}
extern "C" int vican_cg_time_step(int32_t n_time, const double* p_t, const double* q_t, double* x_t, double* r_t,
                                  double* rr_part /* >= 3*512 doubles */, int32_t part_cap, const vican_cg_state_t* st,
                                  void* stream) {
    if (n_time < 0 || !p_t || !q_t || !x_t || !r_t || !rr_part || part_cap < 3 * CG_PARTS || !st)
        return set_err(VICAN_ERR_ARG, "vican_cg_time_step: bad argument");
    const long long n = 3LL * n_time;
    int nb = (int)((n + 1023) / 1024); if (nb < 1) nb = 1; if (nb > CG_PARTS) nb = CG_PARTS;
    hipLaunchKernelGGL(cg_time_step_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, n, p_t, q_t, x_t, r_t, rr_part, st);
    LAUNCH_CHECK("vican_cg_time_step");
    return nb;
}

// cg_cam_step + cg_time_step in ONE launch: every block derives alpha itself (the camera-side dot product is
// 3C elements: cheaper to recompute per block than to wait for another kernel); the LAST block (an extra one: the time side is
// strided over gridDim.x - 1 blocks) updates the camera vectors and their part of the state - as block 0's epilogue it
// stretched the launch by its 3-4 us.  Same thread mappings and summation orders as the two separate kernels => same bits.
__global__ __launch_bounds__(256) void cg_step_kernel(int n_cam, long long n, const double* __restrict__ deg_c,
                                                      const double* __restrict__ qc_sum, const double* __restrict__ pq_time,
                                                      const double* __restrict__ p_c, double* x_c, double* r_c,
                                                      const double* __restrict__ p_t, const double* __restrict__ q_t,
                                                      double* __restrict__ x_t, double* __restrict__ r_t, double* __restrict__ rr_part,
                                                      vican_cg_state_t* st) {
    __shared__ double red[8];
    __shared__ double sh_alpha;
    if (st->done) return;
    const int nc = 3 * n_cam;
    const double s = cg_cam_dot(nc, deg_c, p_c, qc_sum);
    const double pqc = block_sum(s, red);
    if (threadIdx.x == 0) {
        const double pq = *pq_time + pqc;
        sh_alpha = st->rho / pq;
        if (blockIdx.x == 0) { st->pq_time = *pq_time; st->pq = pq; st->alpha = sh_alpha; }
    }
    __syncthreads();
    const double alpha = sh_alpha;
    const int nb = (int)gridDim.x - 1;                     // blocks of the time side
    if ((int)blockIdx.x < nb) {
    double rr = 0.0, m = 0.0, mp = 0.0;
#pragma unroll 4
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)nb * 256) {
        const double pv = p_t[i];
        x_t[i] = mul_add_2r(alpha, pv, x_t[i]);
        mp = fmax(mp, fabs(pv));
        const double r = mul_add_2r(-alpha, q_t[i], r_t[i]);
        r_t[i] = r;
        rr += r * r; m = fmax(m, fabs(r));
    }
    const double t = block_sum(rr, red);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { m = fmax(m, __shfl_down(m, o, 64)); mp = fmax(mp, __shfl_down(mp, o, 64)); }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = m; red[4 + (threadIdx.x >> 6)] = mp; }
    __syncthreads();
    if (threadIdx.x == 0) { rr_part[blockIdx.x] = t; rr_part[CG_PARTS + blockIdx.x] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
                            rr_part[2 * CG_PARTS + blockIdx.x] = fmax(fmax(red[4], red[5]), fmax(red[6], red[7])); }
    return;
    }
    double rc2 = 0.0, mc = 0.0;
    cg_cam_update(nc, alpha, deg_c, qc_sum, p_c, x_c, r_c, rc2, mc);
    const double tc = block_sum(rc2, red);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mc = fmax(mc, __shfl_down(mc, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mc;
    __syncthreads();
    if (threadIdx.x == 0) { st->rr_cam = tc; st->rmax_cam = fmax(fmax(red[0], red[1]), fmax(red[2], red[3])); }
}

__global__ void cg_end_kernel(const double* __restrict__ rr_part, int n_part, vican_cg_state_t* st) {
    if (st->done) return;
    // (the partials through LDS: up to 3 x 512 dependent global loads by one thread took ~36 us per iteration of a sharded
    //  solve; same summation order)
    __shared__ double sp[3 * CG_PARTS];
    for (int i = threadIdx.x; i < n_part; i += blockDim.x) {
        sp[i] = rr_part[i]; sp[CG_PARTS + i] = rr_part[CG_PARTS + i]; sp[2 * CG_PARTS + i] = rr_part[2 * CG_PARTS + i];
    }
    __syncthreads();
    if (threadIdx.x == 0) cg_close_iteration(sp, n_part, st);
}
extern "C" int vican_cg_end(const double* rr_part, int32_t n_part, vican_cg_state_t* st, void* stream) {
    if (!rr_part || n_part <= 0 || !st) return set_err(VICAN_ERR_ARG, "vican_cg_end: bad argument");
    hipLaunchKernelGGL(cg_end_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, rr_part, n_part, st);
    LAUNCH_CHECK("vican_cg_end");
    return VICAN_OK;
}

// ---------------------------------------------------------------------------
// ONE message per iteration for sharded solves: the Chronopoulos-Gear arrangement of CG.  scipy's recurrence needs two
// reductions per iteration ([q_c | p.q] after the product, r.r after the update; the second one feeds the beta that the NEXT
// product's p = r + beta p needs).  Forming s = A r instead of q = A p removes the dependency: with gamma = r.r and
// delta = r.s of the SAME residual in one message,
//     beta_k = gamma_k / gamma_{k-1},   alpha_k = gamma_k / (delta_k - beta_k gamma_k / alpha_{k-1}),
//     p_k = r_k + beta_k p_{k-1},  q_k = s_k + beta_k q_{k-1} (= A p_k),  x += alpha_k p_k,  r -= alpha_k q_k
// - the same iterates in exact arithmetic, other roundings (q by recurrence, alpha from delta): another trajectory inside
// the band in which the reference's loosely converged CG moves anyway.  The stopping test is scipy's
// (|r_k| < rtol |b| on the directly formed r.r, before iteration k's update).  The product is the same sweep kernel, run
// on r with a "first iteration" view of the state (`sw`: no p update, its own fixed-point scale from the measured max |r|).
//   cg1_prepare : close the previous update (r_t.r_t and max |r_t| of this rank from the step kernel's partials) -> msg[3C+1],
//                 scale of the sweep -> sw
//   sweep + fold: msg[0:3C] = sum_t w r_t (this rank), msg[3C] = r_t.s_t (this rank);  s_t = deg_t r_t - sum_c w r_c
//   (all-reduce of msg[0:3C+2] by the caller)
//   cg1_step    : every block derives gamma, delta, beta, alpha itself; vector updates; block 0 the camera side and the state
// sc: 4 doubles, gamma and alpha of the last two iterations by parity of k (block 0 of launch k writes slot k & 1 while the
// other blocks still read slot (k-1) & 1); for the same reason the camera residual is written to a second buffer (r_c_new):
// every block reads r_c for its scalars, and blocks start whenever the device lets them (a shared device delays them by
// whole kernels).
__global__ __launch_bounds__(256) void cg1_prepare_kernel(const double* __restrict__ rr_part, int n_part, double n_add,
                                                          const vican_cg_state_t* __restrict__ st, vican_cg_state_t* sw,
                                                          double* msg_rr) {
    __shared__ double red[8];
    if (st->done) { if (threadIdx.x == 0) sw->done = st->done; return; }
    double ps = 0.0, pm = 0.0;
    for (int i = threadIdx.x; i < n_part; i += 256) { ps += rr_part[i]; pm = fmax(pm, rr_part[CG_PARTS + i]); }
    const double tsum = block_sum(ps, red);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) pm = fmax(pm, __shfl_down(pm, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = pm;
    __syncthreads();
    if (threadIdx.x == 0) {
        const double rr = n_part > 0 ? tsum : st->rr_time;                     // (before the first iteration: from cg_init)
        const double rmax_t = n_part > 0 ? fmax(fmax(red[0], red[1]), fmax(red[2], red[3])) : st->rmax_time;
        *msg_rr = rr;
        double inv;
        sw->qscale = fix_scale(st->wmax * fmax(st->rmax_cam, rmax_t), n_add, &inv, 49);
        sw->qinv = inv; sw->lo_bits = fix2_lo_bits(n_add);
        sw->beta = 0.0; sw->first = 1; sw->done = 0;
    }
}

__global__ __launch_bounds__(256) void cg1_step_kernel(int n_cam, long long n, int k, double rtol, const double* __restrict__ deg_c,
                                                       const double* __restrict__ msg, const double* __restrict__ r_c,
                                                       double* __restrict__ r_c_new, double* p_c, double* q_c, double* x_c,
                                                       double* __restrict__ r_t, const double* __restrict__ s_t, double* __restrict__ p_t,
                                                       double* __restrict__ q_t, double* __restrict__ x_t, double* __restrict__ rr_part,
                                                       double* sc, vican_cg_state_t* st) {
    __shared__ double red[8];
    __shared__ double sh_alpha, sh_beta;
    __shared__ int sh_go;
    if (st->done) return;
    const int nc = 3 * n_cam;
    double g = 0.0, d = 0.0;
    for (int i = threadIdx.x; i < nc; i += 256) {
        const double r = r_c[i];
        const double s = deg_c[i / 3] * r - msg[i];
        g += r * r; d += r * s;
    }
    const double gc = block_sum(g, red);
    const double dc = block_sum(d, red);
    if (threadIdx.x == 0) {
        const double gamma = gc + msg[nc + 1], delta = dc + msg[nc];
        const double atol2 = k == 0 ? rtol * rtol * gamma : st->atol2;
        int stop = 0;
        if (sqrt(gamma) < sqrt(atol2) || gamma == 0.0) stop = 1;                 // scipy: norm(r) < atol, at the top of iteration k
        else if (!(gamma == gamma)) stop = -2;
        double beta = 0.0, alpha = 0.0;
        if (!stop) {
            if (k == 0) alpha = gamma / delta;
            else { beta = gamma / sc[(k - 1) & 1]; alpha = gamma / (delta - beta * gamma / sc[2 + ((k - 1) & 1)]); }
        }
        sh_alpha = alpha; sh_beta = beta; sh_go = !stop;
        if (blockIdx.x == 0) {
            if (k == 0) { st->bnorm2 = gamma; st->atol2 = atol2; }
            st->rho_prev = st->rho; st->rho = gamma; st->rr_cam = gc; st->rr_time = msg[nc + 1];
            if (stop) st->done = stop;
            else {
                sc[k & 1] = gamma; sc[2 + (k & 1)] = alpha;
                st->alpha = alpha; st->beta = beta; st->pq = gamma / alpha; st->iter = k + 1; st->first = 0;
            }
        }
    }
    __syncthreads();
    if (!sh_go) return;
    const double alpha = sh_alpha, beta = sh_beta;
    double rr = 0.0, m = 0.0;
#pragma unroll 4
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        double p = r_t[i], q = s_t[i];
        if (k) { p = mul_add_2r(beta, p_t[i], p); q = mul_add_2r(beta, q_t[i], q); }
        p_t[i] = p; q_t[i] = q;
        x_t[i] = mul_add_2r(alpha, p, x_t[i]);
        const double r = mul_add_2r(-alpha, q, r_t[i]);
        r_t[i] = r;
        rr += r * r; m = fmax(m, fabs(r));
    }
    const double t = block_sum(rr, red);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmax(m, __shfl_down(m, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) { rr_part[blockIdx.x] = t; rr_part[CG_PARTS + blockIdx.x] = fmax(fmax(red[0], red[1]), fmax(red[2], red[3])); }
    if (blockIdx.x != 0) return;
    __syncthreads();
    double mc = 0.0;
    for (int i = threadIdx.x; i < nc; i += 256) {
        const double rv = r_c[i];
        const double s = deg_c[i / 3] * rv - msg[i];
        double p = rv, q = s;
        if (k) { p = mul_add_2r(beta, p_c[i], p); q = mul_add_2r(beta, q_c[i], q); }
        p_c[i] = p; q_c[i] = q;
        x_c[i] = mul_add_2r(alpha, p, x_c[i]);
        const double r = mul_add_2r(-alpha, q, rv);
        r_c_new[i] = r;               // (never in place: the other blocks derive gamma and delta from r_c, some of them later than this)
        mc = fmax(mc, fabs(r));
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mc = fmax(mc, __shfl_down(mc, o, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mc;
    __syncthreads();
    if (threadIdx.x == 0) st->rmax_cam = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
}

extern "C" int vican_cg1_iter_local(const vican_graph_t* g, const double* w, const double* deg_t, const double* r_c,
                                    const double* r_t, double* s_t, void* qc_part, double* pq_part, double* msg,
                                    const double* rr_part, int32_t n_part, double n_add, const vican_cg_state_t* st,
                                    vican_cg_state_t* sw, void* stream) {
    if (!g || !w || !deg_t || !r_c || !r_t || !s_t || !qc_part || !pq_part || !msg || !st || !sw || (n_part > 0 && !rr_part) ||
        n_part > CG_PARTS)
        return set_err(VICAN_ERR_ARG, "vican_cg1_iter_local: bad argument");
    hipLaunchKernelGGL(cg1_prepare_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, rr_part, n_part, n_add, st, sw,
                       msg + 3 * (size_t)g->n_cam + 1);
    LAUNCH_CHECK("vican_cg1_iter_local");
    int rc;
    // the product on the residual: the sweep's "first iteration" form reads p_t and never writes it
    if ((rc = vican_cg_sweep(g, w, deg_t, r_c, r_t, const_cast<double*>(r_t), s_t, qc_part, pq_part, sw, stream)) < 0) return rc;
    return vican_cg_fold(qc_part, g->n_wg, g->n_cam, pq_part, msg, sw, stream);
}

extern "C" int vican_cg1_iter_finish(int32_t n_cam, int32_t n_time, int32_t k, double rtol, const double* deg_c, const double* msg,
                                     const double* r_c, double* r_c_new, double* p_c, double* q_c, double* x_c, double* r_t, const double* s_t,
                                     double* p_t, double* q_t, double* x_t, double* rr_part, int32_t part_cap, double* sc,
                                     vican_cg_state_t* st, void* stream) {
    if (n_cam <= 0 || n_time < 0 || k < 0 || !deg_c || !msg || !r_c || !r_c_new || r_c_new == r_c || !p_c || !q_c || !x_c || !r_t || !s_t || !p_t || !q_t ||
        !x_t || !rr_part || part_cap < 2 * CG_PARTS || !sc || !st)
        return set_err(VICAN_ERR_ARG, "vican_cg1_iter_finish: bad argument");
    const long long n = 3LL * n_time;
    int nb = (int)((n + 1023) / 1024); if (nb < 1) nb = 1; if (nb > CG_PARTS) nb = CG_PARTS;
    hipLaunchKernelGGL(cg1_step_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, n_cam, n, k, rtol, deg_c, msg, r_c, r_c_new, p_c, q_c,
                       x_c, r_t, s_t, p_t, q_t, x_t, rr_part, sc, st);
    LAUNCH_CHECK("vican_cg1_iter_finish");
    return nb;
}

// fold of one CG sweep in ONE launch: qcpq[0:3C] = sum_wg qc_part (double-word planes [2][3][C] per workgroup -> row-major
// [C][3] doubles, rounded once) and qcpq[3C] = sum_wg pq_part (block 0).  Integer sums: exact, any order, overflow-proof
// (fix3_add / fix3_value, vican_sweep_common.h).
template <int COLS>          // columns per workgroup (1024 / COLS lane groups stride over the slabs): see slab_reduce_fx_kernel
__device__ __forceinline__ void cg_fold_columns(const long long* __restrict__ part, int n_slab, long long n, int lob, long long (*sh)[1024],
                                                long long& t, long long& b, long long& l, long long& col, bool& owner) {
    constexpr int NG = 1024 / COLS;
    const int e = threadIdx.x % COLS, grp = threadIdx.x / COLS;
    const long long i = (long long)blockIdx.x * COLS + e;
    Fix3 a = {0, 0, 0};
    if (i < n) {
        // (two slabs per pass: four independent loads in flight per lane instead of two)
        int k = grp;
        for (; k + NG < n_slab; k += 2 * NG) {
            const long long h0 = part[(size_t)k * 2 * n + i], l0 = part[(size_t)k * 2 * n + n + i];
            const long long h1 = part[(size_t)(k + NG) * 2 * n + i], l1 = part[(size_t)(k + NG) * 2 * n + n + i];
            fix3_add(a, h0, l0, lob); fix3_add(a, h1, l1, lob);
        }
        if (k < n_slab) fix3_add(a, part[(size_t)k * 2 * n + i], part[(size_t)k * 2 * n + n + i], lob);
    }
    sh[0][threadIdx.x] = a.top; sh[1][threadIdx.x] = a.bot; sh[2][threadIdx.x] = a.lo;
    __syncthreads();
    owner = grp == 0 && i < n;
    col = i;
    t = b = l = 0;
    if (owner) {
#pragma unroll
        for (int k = 0; k < NG; ++k) { t += sh[0][k * COLS + e]; b += sh[1][k * COLS + e]; l += sh[2][k * COLS + e]; }
    }
}
template <int COLS>
__global__ __launch_bounds__(1024) void cg_fold_kernel(const long long* __restrict__ part, int n_slab, int n_cam,
                                                       const double* __restrict__ pq_part, double* __restrict__ qcpq,
                                                       const vican_cg_state_t* __restrict__ st) {
    __shared__ long long sh[3][1024];
    if (st->done) return;
    const long long n = 3LL * n_cam;
    const int lob = st->lo_bits;
    long long t, b, l, i; bool owner;
    cg_fold_columns<COLS>(part, n_slab, n, lob, sh, t, b, l, i, owner);
    if (owner) {
        const long long q = i / n_cam, cam = i % n_cam;
        qcpq[cam * 3 + q] = fix3_value(t, b, l, lob, st->qinv);
    }
    if (blockIdx.x == 0 && pq_part != nullptr) {   // p.q partials: loaded in parallel, summed in a fixed order
        __shared__ double pq[1024];
        __syncthreads();
        for (int k0 = 0; k0 < n_slab; k0 += 1024) {
            const int k = k0 + threadIdx.x;
            pq[threadIdx.x] = k < n_slab ? pq_part[k] : 0.0;
            __syncthreads();
            if (threadIdx.x == 0) {
                double t = k0 ? qcpq[n] : 0.0;
                const int m = n_slab - k0 < 1024 ? n_slab - k0 : 1024;
                for (int j = 0; j < m; ++j) t += pq[j];
                qcpq[n] = t;
            }
            __syncthreads();
        }
    }
}
// (measured on the stress graph, 256 slabs x 3000 columns: 64-column workgroups 9.7 us, 16-column ones 21.9 us - 128-byte pieces
//  of 256 slabs 48 KB apart; the narrower instantiations stay for tests)
static inline int cg_fold_cols(long long n, int n_slab) { (void)n; (void)n_slab; return 64; }

extern "C" int vican_cg_fold(const void* qc_part, int32_t n_slab, int32_t n_cam, const double* pq_part, double* qcpq,
                             const vican_cg_state_t* st, void* stream) {
    if (!qc_part || n_slab <= 0 || n_cam <= 0 || !qcpq || !st) return set_err(VICAN_ERR_ARG, "vican_cg_fold: bad argument");
    const long long n = 3LL * n_cam;
#define CGF_LAUNCH(COLS_) hipLaunchKernelGGL(cg_fold_kernel<COLS_>, dim3((unsigned)((n + COLS_ - 1) / COLS_)), dim3(1024), 0, (hipStream_t)stream, \
                                             (const long long*)qc_part, n_slab, n_cam, pq_part, qcpq, st)
    const int cols = cg_fold_cols(n, n_slab);
    if (cols == 64) CGF_LAUNCH(64); else if (cols == 32) CGF_LAUNCH(32); else CGF_LAUNCH(16);
#undef CGF_LAUNCH
    LAUNCH_CHECK("vican_cg_fold");
    return VICAN_OK;
}

// the folds of the camera tiles of one CG product (vican_cg_sweep_tiles) in ONE launch: blockIdx.y = tile, tile k's cameras at
// qcpq[3 cam0_k ...] (consecutive camera ranges); same sums, same bits as vican_cg_fold per tile
struct CgFoldTiles { const long long* part[64]; int n_cam[64]; int cam0[64]; };
__global__ __launch_bounds__(1024) void cg_fold_tiles_kernel(CgFoldTiles T, int n_slab, double* __restrict__ qcpq,
                                                             const vican_cg_state_t* __restrict__ st) {
    __shared__ long long sh[3][1024];
    if (st->done) return;
    const int tile = (int)blockIdx.y, n_cam = T.n_cam[tile];
    const long long n = 3LL * n_cam;
    if ((long long)blockIdx.x * 64 >= n) return;
    const int lob = st->lo_bits;
    long long t, b, l, i; bool owner;
    cg_fold_columns<64>(T.part[tile], n_slab, n, lob, sh, t, b, l, i, owner);
    if (owner) {
        const long long q = i / n_cam, cam = i % n_cam;
        qcpq[(T.cam0[tile] + cam) * 3 + q] = fix3_value(t, b, l, lob, st->qinv);
    }
}
extern "C" int vican_cg_fold_tiles(const void* const* qc_parts, const int32_t* n_cams, int32_t n_tile, int32_t n_slab, double* qcpq,
                                   const vican_cg_state_t* st, void* stream) {
    if (!qc_parts || !n_cams || n_tile <= 0 || n_tile > 64 || n_slab <= 0 || !qcpq || !st) return set_err(VICAN_ERR_ARG, "vican_cg_fold_tiles: bad argument");
    CgFoldTiles T;
    int cam0 = 0, cmax = 0;
    for (int k = 0; k < n_tile; ++k) {
        if (!qc_parts[k] || n_cams[k] <= 0) return set_err(VICAN_ERR_ARG, "vican_cg_fold_tiles: bad argument");
        T.part[k] = (const long long*)qc_parts[k]; T.n_cam[k] = n_cams[k]; T.cam0[k] = cam0;
        cam0 += n_cams[k];
        cmax = n_cams[k] > cmax ? n_cams[k] : cmax;
    }
    hipLaunchKernelGGL(cg_fold_tiles_kernel, dim3((unsigned)((3LL * cmax + 63) / 64), (unsigned)n_tile), dim3(1024), 0, (hipStream_t)stream,
                       T, (int)n_slab, qcpq, st);
    LAUNCH_CHECK("vican_cg_fold_tiles");
    return VICAN_OK;
}

// ---------------------------------------------------------------------------
// Single-rank iteration behind one host call (begin, sweep, fold, step), bit-reproducible from run to run:
//   sweep                      (unchanged)
//   cg_fold2  = cg_fold + partials of p_t.q_t over FIXED slices (the sweep's own partial depends on the order in which its
//               wavefronts drew their chunk tickets: the only sum of the iteration that was not reproducible)
//   cg_step2  = cg_step with alpha from those slices (by every workgroup for itself)
// Measured on the stress graph (profiles/r05_cg_tail.txt): the tail kernels are bound by their own dependent memory round
// trips, not by launch gaps (1.3 us each).  A version in which the step's last workgroup (agent-scope ticket, sc1 hand-over
// words) also ran the next iteration's head - three launches instead of four - was built in round 5, tested bit-identical, measured
// SLOWER (the head's 8 us of dependent round trips stay serial behind the last step workgroup, the ticket adds 3 us) and
// removed in round 6 (tools/lab_patches/cg_handover.patch keeps it).  What made the tail shorter was staging the camera-side
// loops (cg_cam_dot / cg_cam_update: cg_begin 9.4 -> 8.2 us, cg_step 13.3 -> 10.6 us).
// Same thread mappings, summation orders and expressions as the four kernels => same bits
// (tests/test_kernels_gpu.py::test_fused_cg_iteration).
// ---------------------------------------------------------------------------
// p_t . q_t is formed HERE, over fixed slices in a fixed order, not taken from the sweep's pq_part: the sweeps hand their chunks
// to wavefronts by ticket, so which wavefront adds which rows' p q - and with it the last bits of the workgroup's floating-
// point partial - depends on timing (every other sum of the sweep is an exact integer sum).  4.8 MB of extra reads on the
// stress graph; the single-rank CG is bit-reproducible from run to run.
template <int COLS>
__global__ __launch_bounds__(1024) void cg_fold2_kernel(const long long* __restrict__ part, int n_slab, int n_cam,
                                                        const double* __restrict__ p_t, const double* __restrict__ q_t, long long n_t, int n_pq,
                                                        double* __restrict__ pq_part, double* __restrict__ qcpq,
                                                        const vican_cg_state_t* __restrict__ st, int n_pq_pad) {
    __shared__ long long sh[3][1024];
    __shared__ double red[16];
    if (st->done) return;
    // (sharded runs all-reduce the slices element by element: the ones this rank does not fill must be zero every time)
    if (blockIdx.x == 0 && (int)threadIdx.x >= n_pq && (int)threadIdx.x < n_pq_pad) pq_part[threadIdx.x] = 0.0;
    const long long n = 3LL * n_cam;
    const int lob = st->lo_bits;
    const int e = threadIdx.x & 63, grp = threadIdx.x >> 6;
    if ((long long)blockIdx.x * COLS < n) {                    // (workgroups beyond the columns only take a slice of p_t.q_t)
        long long t, b, l, i; bool owner;
        cg_fold_columns<COLS>(part, n_slab, n, lob, sh, t, b, l, i, owner);
        if (owner) {
            const long long q = i / n_cam, cam = i % n_cam;
            qcpq[cam * 3 + q] = fix3_value(t, b, l, lob, st->qinv);
        }
    }
    if ((int)blockIdx.x < n_pq) {   // this workgroup's slice of p_t . q_t (n_pq slices: enough workgroups for long captures of few cameras)
        double s = 0.0;
#pragma unroll 4
        for (long long i = (long long)blockIdx.x * 1024 + threadIdx.x; i < n_t; i += (long long)n_pq * 1024) s += p_t[i] * q_t[i];
        s = wave_sum(s);
        if (e == 0) red[grp] = s;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int k = 0; k < 16; ++k) t += red[k];
            pq_part[blockIdx.x] = t;
        }
    }
}

__global__ __launch_bounds__(256) void cg_step2_kernel(int n_cam, long long n, const double* __restrict__ deg_c,
                                                       const double* __restrict__ qc_sum, double* p_c, double* x_c, double* r_c,
                                                       const double* __restrict__ p_t, const double* __restrict__ q_t,
                                                       double* __restrict__ x_t, double* __restrict__ r_t, double* rr_part,
                                                       const double* __restrict__ pq_part, int n_pq,
                                                       vican_cg_state_t* st, int rr_pad) {
    __shared__ double red[8];
    if (st->done) return;
    const int nc = 3 * n_cam;
    // (sharded runs all-reduce rr_part[0:rr_pad] element by element: the partials beyond this rank's blocks are zero every time)
    if (blockIdx.x == gridDim.x - 1)
        for (int i = (int)gridDim.x - 1 + (int)threadIdx.x; i < rr_pad; i += 256) rr_part[i] = 0.0;
    // alpha = rho / (p_t.q_t + p_c.q_c), by every workgroup for itself from the same numbers in the same order (as cg_step):
    // the partials of p_t.q_t that the fold's workgroups left (fixed slices, workgroup order) and the camera part
    __shared__ double sh_alpha;
    {
        double tq = 0.0;
        for (int k = threadIdx.x; k < n_pq; k += 256) tq += pq_part[k];     // (n_pq <= 192: at most one term per thread)
        __shared__ double pqs[256];
        pqs[threadIdx.x] = tq;
        const double sdot = cg_cam_dot(nc, deg_c, p_c, qc_sum);
        const double pqc = block_sum(sdot, red);           // (its barriers also publish pqs)
        if (threadIdx.x == 0) {
            double pq_time = 0.0;
            const int m = n_pq < 256 ? n_pq : 256;
            for (int k = 0; k < m; ++k) pq_time += pqs[k];
            const double pq = pq_time + pqc;
            sh_alpha = st->rho / pq;
            if (blockIdx.x == 0) { st->pq_time = pq_time; st->pq = pq; st->alpha = sh_alpha; }
        }
        __syncthreads();
    }
    const double alpha = sh_alpha;
    const int nb = (int)gridDim.x - 1;                     // blocks of the time side
    if ((int)blockIdx.x < nb) {
        double rr = 0.0, m = 0.0, mp = 0.0;
#pragma unroll 4
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)nb * 256) {
            const double pv = p_t[i];
            x_t[i] = mul_add_2r(alpha, pv, x_t[i]);
            mp = fmax(mp, fabs(pv));
            const double r = mul_add_2r(-alpha, q_t[i], r_t[i]);
            r_t[i] = r;
            rr += r * r; m = fmax(m, fabs(r));
        }
        const double t = block_sum(rr, red);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { m = fmax(m, __shfl_down(m, o, 64)); mp = fmax(mp, __shfl_down(mp, o, 64)); }
        __syncthreads();
        if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = m; red[4 + (threadIdx.x >> 6)] = mp; }
        __syncthreads();
        if (threadIdx.x == 0) {
            const double rm = fmax(fmax(red[0], red[1]), fmax(red[2], red[3])), pm = fmax(fmax(red[4], red[5]), fmax(red[6], red[7]));
            rr_part[blockIdx.x] = t; rr_part[CG_PARTS + blockIdx.x] = rm; rr_part[2 * CG_PARTS + blockIdx.x] = pm;
        }
    } else {
        double rc2 = 0.0, mc = 0.0;
        cg_cam_update(nc, alpha, deg_c, qc_sum, p_c, x_c, r_c, rc2, mc);
        const double tc = block_sum(rc2, red);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mc = fmax(mc, __shfl_down(mc, o, 64));
        __syncthreads();
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mc;
        __syncthreads();
        if (threadIdx.x == 0) {
            const double rm = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
            st->rr_cam = tc; st->rmax_cam = rm;
        }
    }
}

// One CG iteration of a single rank: cg_begin + sweep + cg_fold2 + cg_step2.  `first` != 0: the call that follows
// vican_cg_init.  ws: 1024 bytes owned by this solve, 8-byte aligned - from byte 256 the fold's partials of p_t.q_t (the first
// 256 bytes are unused: the hand-over words of round 5's variant lived there).
// The state after k calls equals the state after k x (vican_cg_iter_local, vican_cg_iter_finish) plus the vican_cg_begin
// of the next call, bit for bit.
extern "C" int vican_cg_iter_fused(const vican_graph_t* g, const double* w, const double* deg_t, const double* deg_c,
                                   double* r_c, double* p_c, double* x_c, double* r_t, double* p_t, double* q_t, double* x_t,
                                   void* qc_part, double* pq_part, double* qcpq, double rtol, double* rr_part, int32_t part_cap,
                                   double n_add, int32_t first, vican_cg_state_t* st, uint32_t* ws, void* stream) {
    if (!g || !deg_c || !r_c || !p_c || !x_c || !r_t || !p_t || !q_t || !x_t || !qcpq || !rr_part || part_cap < 3 * CG_PARTS || !st || !ws)
        return set_err(VICAN_ERR_ARG, "vican_cg_iter_fused: bad argument");
    int rc;
    hipStream_t s = (hipStream_t)stream;
    const long long n = 3LL * g->n_time, nc = 3LL * g->n_cam;
    int nb = (int)((n + 1023) / 1024); if (nb < 1) nb = 1; if (nb > CG_PARTS) nb = CG_PARTS;        // as vican_cg_time_step
    if ((rc = vican_cg_begin(g->n_cam, r_c, p_c, rtol, rr_part, (first & 1) ? 0 : nb, n_add, st, stream)) < 0) return rc;
    if ((rc = vican_cg_sweep(g, w, deg_t, p_c, r_t, p_t, q_t, qc_part, pq_part, st, stream)) < 0) return rc;
    // partials of p_t . q_t, one per workgroup of the fold (<= 96: C <= 2048): doubles 32.. of the solve's workspace
    double* pq_part2 = (double*)(ws + 64);
    const int n_fold = (int)((nc + 63) / 64);
    int n_pq = (int)((n + 8191) / 8192); if (n_pq < 1) n_pq = 1; if (n_pq > 96) n_pq = 96;       // slices of p_t.q_t (8 elements per thread and pass)
    hipLaunchKernelGGL(cg_fold2_kernel<64>, dim3((unsigned)(n_fold > n_pq ? n_fold : n_pq)), dim3(1024), 0, s, (const long long*)qc_part, (int)g->n_wg,
                       (int)g->n_cam, p_t, q_t, n, n_pq, pq_part2, qcpq, st, 0);
    hipLaunchKernelGGL(cg_step2_kernel, dim3(nb + 1), dim3(256), 0, s, (int)g->n_cam, n, deg_c, qcpq, p_c, x_c, r_c, p_t, q_t, x_t, r_t,
                       rr_part, pq_part2, n_pq, st, 0);
    LAUNCH_CHECK("vican_cg_iter_fused");
    return VICAN_OK;
}

// ---------------------------------------------------------------------------
// One CG iteration of a SHARDED solve behind one host call (timestep rows split over the ranks of `comm`, camera side
// replicated): the single-rank sequence above with the two sums that cross ranks all-reduced in stream order -
//   [cg_begin] sweep, cg_fold2 -> msg = [q_c partial (3C) | slices of p_t.q_t (VICAN_CG_PQ_SLICES, zero-padded)] -> all-reduce
//   cg_step2 (alpha from the reduced message, by every workgroup for itself) -> rr_part[0:VICAN_CG_RR_SLICES] (zero-padded) -> all-reduce
// and the next call's cg_begin closes the iteration from the reduced partials.  scipy's recurrence (two messages per
// iteration, bipgo.py:477).  Every partial that enters a message is formed over FIXED slices in a fixed order and the peer
// exchange sums ranks in rank order: the iterates are bit-reproducible from run to run and bit-identical on every rank.
// first != 0: the call that follows vican_cg_init AND the all-reduce of the state's rr_time (the caller's).
// msg: 3C + VICAN_CG_PQ_SLICES doubles.  The maxima that bound the fixed-point scale stay rank-local (each rank converts its
// own partials to doubles before they travel).
// ---------------------------------------------------------------------------
extern "C" int vican_cg_iter_comm(const vican_graph_t* g, const double* w, const double* deg_t, const double* deg_c,
                                  double* r_c, double* p_c, double* x_c, double* r_t, double* p_t, double* q_t, double* x_t,
                                  void* qc_part, double* pq_part, double* msg, double rtol, double* rr_part, int32_t part_cap,
                                  double n_add, int32_t first, vican_cg_state_t* st, vican_comm_t* comm, void* stream) {
    if (!g || !deg_c || !r_c || !p_c || !x_c || !r_t || !p_t || !q_t || !x_t || !msg || !rr_part || part_cap < 3 * CG_PARTS || !st)
        return set_err(VICAN_ERR_ARG, "vican_cg_iter_comm: bad argument");
    int rc;
    hipStream_t s = (hipStream_t)stream;
    const long long n = 3LL * g->n_time, nc = 3LL * g->n_cam;
    int nb = (int)((n + 1023) / 1024); if (nb < 1) nb = 1; if (nb > CG_PARTS) nb = CG_PARTS;        // as vican_cg_time_step
    if ((rc = vican_cg_begin(g->n_cam, r_c, p_c, rtol, rr_part, first ? 0 : VICAN_CG_RR_SLICES, n_add, st, stream)) < 0) return rc;
    if ((rc = vican_cg_sweep(g, w, deg_t, p_c, r_t, p_t, q_t, qc_part, pq_part, st, stream)) < 0) return rc;
    double* pq_slices = msg + nc;
    const int n_fold = (int)((nc + 63) / 64);
    int n_pq = (int)((n + 8191) / 8192); if (n_pq < 1) n_pq = 1; if (n_pq > VICAN_CG_PQ_SLICES) n_pq = VICAN_CG_PQ_SLICES;
    hipLaunchKernelGGL(cg_fold2_kernel<64>, dim3((unsigned)(n_fold > n_pq ? n_fold : n_pq)), dim3(1024), 0, s, (const long long*)qc_part, (int)g->n_wg,
                       (int)g->n_cam, p_t, q_t, n, n_pq, pq_slices, msg, st, VICAN_CG_PQ_SLICES);
    LAUNCH_CHECK("vican_cg_iter_comm");
    if (comm && (rc = vican_comm_allreduce_sum(comm, msg, nc + VICAN_CG_PQ_SLICES, stream)) < 0) return rc;
    hipLaunchKernelGGL(cg_step2_kernel, dim3(nb + 1), dim3(256), 0, s, (int)g->n_cam, n, deg_c, msg, p_c, x_c, r_c, p_t, q_t, x_t, r_t,
                       rr_part, pq_slices, VICAN_CG_PQ_SLICES, st, VICAN_CG_RR_SLICES);
    LAUNCH_CHECK("vican_cg_iter_comm");
    if (comm && (rc = vican_comm_allreduce_sum(comm, rr_part, VICAN_CG_RR_SLICES, stream)) < 0) return rc;
    return VICAN_OK;
}

// ---------------------------------------------------------------------------
// composites: one CG iteration as two host calls (local half up to the point where a sharded
// run all-reduces [q_c | p.q]; finishing half).  qcpq: [3C + 1] doubles.
// ---------------------------------------------------------------------------
extern "C" int vican_cg_iter_local(const vican_graph_t* g, const double* w, const double* deg_t, const double* r_c,
                                   double* p_c, const double* r_t, double* p_t, double* q_t, void* qc_part,
                                   double* pq_part, double* qcpq, double rtol, const double* rr_part, int32_t n_part,
                                   double n_add, vican_cg_state_t* st, void* stream) {
    int rc;
    if ((rc = vican_cg_begin(g->n_cam, r_c, p_c, rtol, rr_part, n_part, n_add, st, stream)) < 0) return rc;
    if ((rc = vican_cg_sweep(g, w, deg_t, p_c, r_t, p_t, q_t, qc_part, pq_part, st, stream)) < 0) return rc;
    return vican_cg_fold(qc_part, g->n_wg, g->n_cam, pq_part, qcpq, st, stream);
}
extern "C" int vican_cg_iter_finish(int32_t n_cam, int32_t n_time, const double* deg_c, const double* qcpq,
                                    const double* p_c, double* x_c, double* r_c, const double* p_t, const double* q_t,
                                    double* x_t, double* r_t, double* rr_part, int32_t part_cap, vican_cg_state_t* st,
                                    void* stream) {
    if (n_cam <= 0 || n_time < 0 || !deg_c || !qcpq || !p_c || !x_c || !r_c || !p_t || !q_t || !x_t || !r_t || !rr_part ||
        part_cap < 3 * CG_PARTS || !st)
        return set_err(VICAN_ERR_ARG, "vican_cg_iter_finish: bad argument");
    const long long n = 3LL * n_time;
    int nb = (int)((n + 1023) / 1024); if (nb < 1) nb = 1; if (nb > CG_PARTS) nb = CG_PARTS;        // as vican_cg_time_step
    hipLaunchKernelGGL(cg_step_kernel, dim3(nb + 1), dim3(256), 0, (hipStream_t)stream, n_cam, n, deg_c, qcpq, qcpq + 3 * n_cam,
                       p_c, x_c, r_c, p_t, q_t, x_t, r_t, rr_part, st);
    LAUNCH_CHECK("vican_cg_iter_finish");
    return nb;
}
