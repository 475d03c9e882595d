// vican_kernels.hip - camera-side and translation-stage kernels: batched 3x3 polar /
// gauge fix, dense helpers of the block Lanczos iteration, right-hand side and the
// conjugate-gradient kernels.  The hot edge sweep lives in vican_sweep.hip.
#include "common.cuh"

// ---------------------------------------------------------------------------
// batched polar / gauge
// ---------------------------------------------------------------------------
__global__ void polar_dual_kernel(int n, const double* __restrict__ in, double* __restrict__ R_out,
                                  double* __restrict__ lam_out, int mode) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double A[9], R[9], lam[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) A[q] = in[(size_t)i * 9 + q];
    polar_dual3(A, R, lam, mode);
    if (R_out)
#pragma unroll
        for (int q = 0; q < 9; ++q) R_out[(size_t)i * 9 + q] = R[q];
    if (lam_out && mode)
#pragma unroll
        for (int q = 0; q < 9; ++q) lam_out[(size_t)i * 9 + q] = lam[q];
}
extern "C" int vican_polar_dual(int32_t n, const double* in, double* R_out, double* lam_out, int32_t mode,
                                void* stream) {
    if (n < 0 || !in || mode < 0 || mode > 2) return set_err(VICAN_ERR_ARG, "vican_polar_dual: bad argument");
    if (n == 0) return VICAN_OK;
    hipLaunchKernelGGL(polar_dual_kernel, dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, n, in, R_out,
                       lam_out, mode);
    LAUNCH_CHECK("vican_polar_dual");
    return VICAN_OK;
}

__global__ void gauge_project_kernel(int n_cam, const double* __restrict__ xin, double* __restrict__ xout) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cam) return;
    double g0[9], gi[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) g0[q] = xin[q];
    const double d = det3(g0), id = 1.0 / d;
    gi[0] = (g0[4] * g0[8] - g0[5] * g0[7]) * id; gi[1] = (g0[2] * g0[7] - g0[1] * g0[8]) * id; gi[2] = (g0[1] * g0[5] - g0[2] * g0[4]) * id;
    gi[3] = (g0[5] * g0[6] - g0[3] * g0[8]) * id; gi[4] = (g0[0] * g0[8] - g0[2] * g0[6]) * id; gi[5] = (g0[2] * g0[3] - g0[0] * g0[5]) * id;
    gi[6] = (g0[3] * g0[7] - g0[4] * g0[6]) * id; gi[7] = (g0[1] * g0[6] - g0[0] * g0[7]) * id; gi[8] = (g0[0] * g0[4] - g0[1] * g0[3]) * id;
    double xc[9], A[9], R[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) xc[q] = xin[(size_t)c * 9 + q];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) A[i * 3 + j] = xc[i * 3] * gi[j] + xc[i * 3 + 1] * gi[3 + j] + xc[i * 3 + 2] * gi[6 + j];
    polar_dual3(A, R, nullptr, 0);
#pragma unroll
    for (int q = 0; q < 9; ++q) xout[(size_t)c * 9 + q] = R[q];
}
extern "C" int vican_gauge_project(int32_t n_cam, const double* x_in, double* x_out, void* stream) {
    if (n_cam <= 0 || !x_in || !x_out) return set_err(VICAN_ERR_ARG, "vican_gauge_project: bad argument");
    // x_in may alias x_out: every thread reads block 0 before any thread of ANOTHER
    // workgroup may overwrite it only if they do not alias; require distinct buffers.
    if (x_in == x_out) return set_err(VICAN_ERR_ARG, "vican_gauge_project: in-place not supported");
    hipLaunchKernelGGL(gauge_project_kernel, dim3((n_cam + 127) / 128), dim3(128), 0, (hipStream_t)stream, n_cam, x_in,
                       x_out);
    LAUNCH_CHECK("vican_gauge_project");
    return VICAN_OK;
}

// ---------------------------------------------------------------------------
// camera-side dense helpers for block Lanczos.  V: column-major basis, column k
// at V + k*ld (ld >= n).  R (work block) is column-major [3][n]; x/z are row-major [n][3].
// ---------------------------------------------------------------------------
__global__ void lap_apply_kernel(int n_cam, const double* __restrict__ lamC, const double* __restrict__ V, int ld,
                                 int col0, const double* __restrict__ z, double* __restrict__ aq, int n) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= n_cam) return;
    double L[9], q[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) L[k] = lamC[(size_t)c * 9 + k];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int b = 0; b < 3; ++b) q[i * 3 + b] = V[(size_t)(col0 + b) * ld + 3 * c + i];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int b = 0; b < 3; ++b)
            aq[(size_t)b * n + 3 * c + i] = L[i * 3] * q[b] + L[i * 3 + 1] * q[3 + b] + L[i * 3 + 2] * q[6 + b] -
                                           z[(size_t)(3 * c + i) * 3 + b];
}
extern "C" int vican_lap_apply(int32_t n_cam, const double* lamC, const double* V, int32_t ld, int32_t col0,
                               const double* z, double* aq, void* stream) {
    if (n_cam <= 0 || !lamC || !V || !z || !aq || ld < 3 * n_cam || col0 < 0)
        return set_err(VICAN_ERR_ARG, "vican_lap_apply: bad argument");
    hipLaunchKernelGGL(lap_apply_kernel, dim3((n_cam + 127) / 128), dim3(128), 0, (hipStream_t)stream, n_cam, lamC, V,
                       ld, col0, z, aq, 3 * n_cam);
    LAUNCH_CHECK("vican_lap_apply");
    return VICAN_OK;
}

// H[k][c] = V[:,k] . R[:,c]   one workgroup per basis column k (deterministic order)
__global__ __launch_bounds__(256) void tall_gram_kernel(int n, const double* __restrict__ V, int ld,
                                                        const double* __restrict__ R, double* __restrict__ H) {
    __shared__ double red[8];
    const int k = blockIdx.x;
    const double* v = V + (size_t)k * ld;
    double s0 = 0, s1 = 0, s2 = 0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const double vi = v[i];
        s0 += vi * R[i]; s1 += vi * R[(size_t)n + i]; s2 += vi * R[(size_t)2 * n + i];
    }
    const double t0 = block_sum(s0, red), t1 = block_sum(s1, red), t2 = block_sum(s2, red);
    if (threadIdx.x == 0) { H[k * 3] = t0; H[k * 3 + 1] = t1; H[k * 3 + 2] = t2; }
}
extern "C" int vican_tall_gram(int32_t n, const double* V, int32_t ld, int32_t ka, const double* R, double* H,
                               void* stream) {
    if (n <= 0 || !V || !R || !H || ka <= 0 || ld < n) return set_err(VICAN_ERR_ARG, "vican_tall_gram: bad argument");
    hipLaunchKernelGGL(tall_gram_kernel, dim3(ka), dim3(256), 0, (hipStream_t)stream, n, V, ld, R, H);
    LAUNCH_CHECK("vican_tall_gram");
    return VICAN_OK;
}

#define KA_MAX 192
__global__ __launch_bounds__(256) void tall_update_kernel(int n, const double* __restrict__ V, int ld, int ka,
                                                          const double* __restrict__ H, double* __restrict__ R,
                                                          double* __restrict__ H_out, int accumulate) {
    __shared__ double h[KA_MAX * 3];
    for (int i = threadIdx.x; i < ka * 3; i += 256) h[i] = H[i];
    __syncthreads();
    if (blockIdx.x == 0 && H_out)
        for (int i = threadIdx.x; i < ka * 3; i += 256) H_out[i] = accumulate ? H_out[i] + h[i] : h[i];
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double r0 = R[i], r1 = R[(size_t)n + i], r2 = R[(size_t)2 * n + i];
    for (int k = 0; k < ka; ++k) {
        const double v = V[(size_t)k * ld + i];
        r0 -= v * h[k * 3]; r1 -= v * h[k * 3 + 1]; r2 -= v * h[k * 3 + 2];
    }
    R[i] = r0; R[(size_t)n + i] = r1; R[(size_t)2 * n + i] = r2;
}
extern "C" int vican_tall_update(int32_t n, const double* V, int32_t ld, int32_t ka, const double* part,
                                 double* R, double* H_out, int32_t accumulate, void* stream) {
    if (n <= 0 || !V || !R || !part || ka <= 0 || ka > KA_MAX || ld < n)
        return set_err(VICAN_ERR_ARG, "vican_tall_update: bad argument");
    hipLaunchKernelGGL(tall_update_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, V, ld, ka, part,
                       R, H_out, accumulate);
    LAUNCH_CHECK("vican_tall_update");
    return VICAN_OK;
}

// G = R^T R (3x3, from tall_gram with V := R) -> upper Cholesky beta (G = beta^T beta);
// Q = R beta^-1 written to basis columns col0..col0+2 and, row-major, to x_out for the next sweep.
__global__ __launch_bounds__(256) void chol_qr3_kernel(int n, const double* __restrict__ R, const double* __restrict__ G,
                                                       double* __restrict__ V, int ld, int col0,
                                                       double* __restrict__ beta_out, double* __restrict__ x_out,
                                                       double pivot_floor) {
    const double g00 = G[0], g01 = G[1], g02 = G[2], g11 = G[4], g12 = G[5], g22 = G[8];
    const double tr = g00 + g11 + g22, floor_ = fmax(1e-28 * tr, pivot_floor);
    double b00 = 0, b01 = 0, b02 = 0, b11 = 0, b12 = 0, b22 = 0, i00 = 0, i11 = 0, i22 = 0;
    if (g00 > floor_) { b00 = sqrt(g00); i00 = 1.0 / b00; b01 = g01 * i00; b02 = g02 * i00; }
    const double d11 = g11 - b01 * b01;
    if (d11 > floor_) { b11 = sqrt(d11); i11 = 1.0 / b11; b12 = (g12 - b01 * b02) * i11; }
    const double d22 = g22 - b02 * b02 - b12 * b12;
    if (d22 > floor_) { b22 = sqrt(d22); i22 = 1.0 / b22; }
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0 && beta_out) {
        beta_out[0] = b00; beta_out[1] = b01; beta_out[2] = b02; beta_out[3] = 0; beta_out[4] = b11; beta_out[5] = b12;
        beta_out[6] = 0; beta_out[7] = 0; beta_out[8] = b22;
    }
    if (i >= n) return;
    const double r0 = R[i], r1 = R[(size_t)n + i], r2 = R[(size_t)2 * n + i];
    const double q0 = r0 * i00;
    const double q1 = (r1 - q0 * b01) * i11;
    const double q2 = (r2 - q0 * b02 - q1 * b12) * i22;
    V[(size_t)col0 * ld + i] = q0; V[(size_t)(col0 + 1) * ld + i] = q1; V[(size_t)(col0 + 2) * ld + i] = q2;
    if (x_out) { x_out[(size_t)i * 3] = q0; x_out[(size_t)i * 3 + 1] = q1; x_out[(size_t)i * 3 + 2] = q2; }
}
extern "C" int vican_chol_qr3(int32_t n, const double* R, const double* G, double* V, int32_t ld, int32_t col0,
                              double* beta_out, double* x_out, double pivot_floor, void* stream) {
    if (n <= 0 || !R || !G || !V || ld < n || col0 < 0) return set_err(VICAN_ERR_ARG, "vican_chol_qr3: bad argument");
    hipLaunchKernelGGL(chol_qr3_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, R, G, V, ld, col0,
                       beta_out, x_out, pivot_floor);
    LAUNCH_CHECK("vican_chol_qr3");
    return VICAN_OK;
}

// X[n][3] (row-major) = V[:, :ka] Y[ka][3]
__global__ __launch_bounds__(256) void tall_combine_kernel(int n, const double* __restrict__ V, int ld, int ka,
                                                           const double* __restrict__ Y, double* __restrict__ X) {
    __shared__ double y[KA_MAX * 3];
    for (int i = threadIdx.x; i < ka * 3; i += 256) y[i] = Y[i];
    __syncthreads();
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    double x0 = 0, x1 = 0, x2 = 0;
    for (int k = 0; k < ka; ++k) {
        const double v = V[(size_t)k * ld + i];
        x0 += v * y[k * 3]; x1 += v * y[k * 3 + 1]; x2 += v * y[k * 3 + 2];
    }
    X[(size_t)i * 3] = x0; X[(size_t)i * 3 + 1] = x1; X[(size_t)i * 3 + 2] = x2;
}
extern "C" int vican_tall_combine(int32_t n, const double* V, int32_t ld, int32_t ka, const double* Y, double* X,
                                  void* stream) {
    if (n <= 0 || !V || !Y || !X || ka <= 0 || ka > KA_MAX || ld < n)
        return set_err(VICAN_ERR_ARG, "vican_tall_combine: bad argument");
    hipLaunchKernelGGL(tall_combine_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, V, ld, ka, Y, X);
    LAUNCH_CHECK("vican_tall_combine");
    return VICAN_OK;
}

// row-major [n][3] -> basis columns col0..col0+2 (used to seed the Krylov space)
__global__ void rows_to_cols_kernel(int n, const double* __restrict__ X, double* __restrict__ V, int ld, int col0) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    V[(size_t)col0 * ld + i] = X[(size_t)i * 3];
    V[(size_t)(col0 + 1) * ld + i] = X[(size_t)i * 3 + 1];
    V[(size_t)(col0 + 2) * ld + i] = X[(size_t)i * 3 + 2];
}
extern "C" int vican_rows_to_cols(int32_t n, const double* X, double* V, int32_t ld, int32_t col0, void* stream) {
    if (n <= 0 || !X || !V || ld < n || col0 < 0) return set_err(VICAN_ERR_ARG, "vican_rows_to_cols: bad argument");
    hipLaunchKernelGGL(rows_to_cols_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, X, V, ld, col0);
    LAUNCH_CHECK("vican_rows_to_cols");
    return VICAN_OK;
}

// ---------------------------------------------------------------------------
// translation stage: right-hand side
// ---------------------------------------------------------------------------
template <int BLOCK, int EPL>
__global__ __launch_bounds__(BLOCK) void trans_rhs_kernel(vican_graph_t g, const double* __restrict__ u,
                                                          const double* __restrict__ v, const double* __restrict__ rc,
                                                          const double* __restrict__ rt, double* __restrict__ rhs_t,
                                                          double* __restrict__ rhs_c_part) {
    extern __shared__ double lds[];
    const int nx = 9 * g.n_cam, nc3 = 3 * g.n_cam;
    double* rcs = lds;                 // [C][9]
    double* gc = lds + nx;             // [C][3]
    double* rts = gc + nc3;            // [max_rows][9]
    double* gt = rts + 9 * g.max_rows; // [max_rows][3]
    const int tid = threadIdx.x;
    for (int i = tid; i < nx; i += BLOCK) rcs[i] = rc[i];
    for (int i = tid; i < nc3; i += BLOCK) gc[i] = 0.0;
    const int k0 = (int)(((long long)blockIdx.x * g.n_chunk) / gridDim.x);
    const int k1 = (int)(((long long)(blockIdx.x + 1) * g.n_chunk) / gridDim.x);
    for (int k = k0; k < k1; ++k) {
        const int r0 = g.chunk_row0[k], nrows = g.chunk_row0[k + 1] - r0;
        __syncthreads();
        for (int i = tid; i < 9 * nrows; i += BLOCK) rts[i] = rt[(size_t)r0 * 9 + i];
        for (int i = tid; i < 3 * nrows; i += BLOCK) gt[i] = 0.0;
        __syncthreads();
        for (int j = 0; j < EPL; ++j) {
            const int s = tid * EPL + j;
            const uint32_t id = g.idx[(size_t)k * g.slots + s];
            if (id == VICAN_PAD_SLOT) continue;
            const uint32_t cam = id & 0xFFFFu, row = id >> 16;
            double uu[3], vv[3];
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                uu[p] = u[((size_t)k * 3 + p) * g.slots + s];
                vv[p] = v[((size_t)k * 3 + p) * g.slots + s];
            }
            const double* A = rcs + cam * 9;   // world<-cam = A^T
            const double* B = rts + row * 9;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const double gi = A[0 * 3 + i] * uu[0] + A[1 * 3 + i] * uu[1] + A[2 * 3 + i] * uu[2] +
                                  B[0 * 3 + i] * vv[0] + B[1 * 3 + i] * vv[1] + B[2 * 3 + i] * vv[2];
                lds_add(&gt[row * 3 + i], gi);
                lds_add(&gc[cam * 3 + i], -gi);
            }
        }
        __syncthreads();
        for (int i = tid; i < 3 * nrows; i += BLOCK) rhs_t[(size_t)r0 * 3 + i] = gt[i];
    }
    __syncthreads();
    for (int i = tid; i < nc3; i += BLOCK) rhs_c_part[(size_t)blockIdx.x * nc3 + i] = gc[i];
}

extern "C" int vican_trans_rhs(const vican_graph_t* g, const double* u, const double* v, const double* rc,
                               const double* rt, double* rhs_t, double* rhs_c_part, void* stream) {
    if (int r = vican_check_graph(g, "vican_trans_rhs")) return r;
    if (!u || !v || !rc || !rt || !rhs_t || !rhs_c_part) return set_err(VICAN_ERR_ARG, "vican_trans_rhs: null pointer");
    const size_t lds = (size_t)rhs_lds_bytes(g->n_cam, g->max_rows);
    const int epl = g->slots / g->block_threads;
    hipStream_t st = (hipStream_t)stream;
#define RHS_LAUNCH(B, E)                                                                                         \
    do {                                                                                                         \
        auto kern = trans_rhs_kernel<B, E>;                                                                      \
        static size_t conf = 0;                                                                                  \
        if (lds > conf) { hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); conf = lds; } \
        hipLaunchKernelGGL(kern, dim3(g->n_wg), dim3(B), lds, st, *g, u, v, rc, rt, rhs_t, rhs_c_part);         \
    } while (0)
    if (g->block_threads == 1024)     { if (epl == 4) RHS_LAUNCH(1024, 4); else RHS_LAUNCH(1024, 2); }
    else if (g->block_threads == 768) { if (epl == 4) RHS_LAUNCH(768, 4);  else RHS_LAUNCH(768, 2); }
    else if (g->block_threads == 512) { if (epl == 4) RHS_LAUNCH(512, 4);  else RHS_LAUNCH(512, 2); }
    else                              { if (epl == 4) RHS_LAUNCH(256, 4);  else RHS_LAUNCH(256, 2); }
#undef RHS_LAUNCH
    LAUNCH_CHECK("vican_trans_rhs");
    return VICAN_OK;
}

// ---------------------------------------------------------------------------
// translation stage: conjugate gradients (scipy.sparse.linalg.cg recurrence)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void cg_init_kernel(int n_cam, int n_time, const double* __restrict__ b_c,
                                                      const double* __restrict__ b_t, double* x_c, double* x_t,
                                                      double* r_c, double* r_t, double* p_c, double* p_t,
                                                      double* __restrict__ part) {
    __shared__ double red[8];
    const long long n = 3LL * n_time, nc = 3LL * n_cam;
    double s = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const double b = b_t[i];
        x_t[i] = 0.0; r_t[i] = b; p_t[i] = b; s += b * b;
    }
    if (blockIdx.x == 0)
        for (long long i = threadIdx.x; i < nc; i += 256) { const double b = b_c[i]; x_c[i] = 0.0; r_c[i] = b; p_c[i] = b; }
    const double t = block_sum(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}
__global__ void cg_init_finish_kernel(int n_cam, const double* __restrict__ b_c, const double* __restrict__ part,
                                      int n_part, vican_cg_state_t* st) {
    __shared__ double red[8];
    double s = 0.0;
    for (int i = threadIdx.x; i < 3 * n_cam; i += blockDim.x) s += b_c[i] * b_c[i];
    const double rc = block_sum(s, red);
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < n_part; ++i) t += part[i];
        st->rho = 0; st->rho_prev = 0; st->pq = 0; st->alpha = 0; st->beta = 0; st->bnorm2 = 0; st->atol2 = 0;
        st->rr_cam = rc; st->pq_time = 0; st->rr_time = t; st->iter = 0; st->done = 0; st->first = 1; st->pad = 0;
    }
}
#define CG_PARTS 512
extern "C" int vican_cg_init(int32_t n_cam, int32_t n_time, const double* b_c, const double* b_t, double* x_c,
                             double* x_t, double* r_c, double* r_t, double* p_c, double* p_t, vican_cg_state_t* st,
                             double* ws /* >= CG_PARTS doubles */, void* stream) {
    if (n_cam <= 0 || n_time < 0 || !b_c || !b_t || !x_c || !x_t || !r_c || !r_t || !p_c || !p_t || !st || !ws)
        return set_err(VICAN_ERR_ARG, "vican_cg_init: bad argument");
    long long n = 3LL * n_time;
    int nb = (int)((n + 255) / 256); if (nb < 1) nb = 1; if (nb > CG_PARTS) nb = CG_PARTS;
    hipLaunchKernelGGL(cg_init_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, n_cam, n_time, b_c, b_t, x_c, x_t,
                       r_c, r_t, p_c, p_t, ws);
    hipLaunchKernelGGL(cg_init_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, n_cam, b_c, ws, nb, st);
    LAUNCH_CHECK("vican_cg_init");
    return VICAN_OK;
}

// rho = rr_cam + rr_time (rr_time = sum of rr_part when n_part > 0, which also closes the
// previous iteration: iter++, rho_prev = rho); convergence test; beta; p_c update.
__global__ __launch_bounds__(256) void cg_begin_kernel(int n_cam, const double* __restrict__ r_c, double* p_c,
                                                       double rtol, const double* __restrict__ rr_part, int n_part,
                                                       vican_cg_state_t* st) {
    __shared__ double sh_beta;
    __shared__ int sh_go;
    if (st->done) return;
    if (threadIdx.x == 0) {
        if (n_part > 0) {
            double t = 0.0;
            for (int i = 0; i < n_part; ++i) t += rr_part[i];
            st->rr_time = t; st->iter += 1; st->rho_prev = st->rho; st->first = 0;
        }
        const double rho = st->rr_cam + st->rr_time;
        if (st->iter == 0 && st->first) { st->bnorm2 = rho; st->atol2 = rtol * rtol * rho; }
        st->rho = rho;
        int go = 1;
        // scipy: if norm(r) < atol: done   (atol = rtol*|b|)
        if (sqrt(rho) < sqrt(st->atol2) || rho == 0.0) { st->done = 1; go = 0; }
        double beta = 0.0;
        if (go && !st->first) beta = rho / st->rho_prev;
        st->beta = beta;
        sh_beta = beta; sh_go = go && !st->first;
    }
    __syncthreads();
    if (!sh_go) return;
    const double beta = sh_beta;
    for (int i = threadIdx.x; i < 3 * n_cam; i += 256) p_c[i] = r_c[i] + beta * p_c[i];
}
extern "C" int vican_cg_begin(int32_t n_cam, const double* r_c, double* p_c, double rtol, const double* rr_part,
                              int32_t n_part, vican_cg_state_t* st, void* stream) {
    if (n_cam <= 0 || !r_c || !p_c || !st || (n_part > 0 && !rr_part)) return set_err(VICAN_ERR_ARG, "vican_cg_begin: bad argument");
    hipLaunchKernelGGL(cg_begin_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, n_cam, r_c, p_c, rtol, rr_part,
                       n_part, st);
    LAUNCH_CHECK("vican_cg_begin");
    return VICAN_OK;
}

template <int BLOCK, int EPL>
__global__ __launch_bounds__(BLOCK) void cg_sweep_kernel(vican_graph_t g, const double* __restrict__ w,
                                                         const double* __restrict__ deg_t,
                                                         const double* __restrict__ p_c, const double* __restrict__ r_t,
                                                         double* __restrict__ p_t, double* __restrict__ q_t,
                                                         double* __restrict__ qc_part, double* __restrict__ pq_part,
                                                         const vican_cg_state_t* __restrict__ st) {
    extern __shared__ double lds[];
    if (st->done) return;
    const int nc3 = 3 * g.n_cam;
    double* pcs = lds;                  // [C][3]
    double* qc = lds + nc3;             // [C][3]
    double* pts = qc + nc3;             // [max_rows][3]
    double* qt = pts + 3 * g.max_rows;  // [max_rows][3]
    double* red = qt + 3 * g.max_rows;  // [16]
    const int tid = threadIdx.x;
    const bool upd = !st->first;
    const double beta = st->beta;
    for (int i = tid; i < nc3; i += BLOCK) { pcs[i] = p_c[i]; qc[i] = 0.0; }
    const int k0 = (int)(((long long)blockIdx.x * g.n_chunk) / gridDim.x);
    const int k1 = (int)(((long long)(blockIdx.x + 1) * g.n_chunk) / gridDim.x);
    double pq = 0.0;
    for (int k = k0; k < k1; ++k) {
        const int r0 = g.chunk_row0[k], nrows = g.chunk_row0[k + 1] - r0;
        __syncthreads();
        for (int i = tid; i < 3 * nrows; i += BLOCK) {
            const size_t gi = (size_t)r0 * 3 + i;
            double p = p_t[gi];
            if (upd) { p = r_t[gi] + beta * p; p_t[gi] = p; }
            pts[i] = p; qt[i] = 0.0;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            const size_t s = (size_t)k * g.slots + (size_t)tid * EPL + j;
            const uint32_t id = g.idx[s];
            if (id == VICAN_PAD_SLOT) continue;
            const uint32_t cam = id & 0xFFFFu, row = id >> 16;
            const double ww = w[s];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                lds_add(&qt[row * 3 + i], ww * pcs[cam * 3 + i]);
                lds_add(&qc[cam * 3 + i], ww * pts[row * 3 + i]);
            }
        }
        __syncthreads();
        for (int i = tid; i < 3 * nrows; i += BLOCK) {
            const double q = deg_t[r0 + i / 3] * pts[i] - qt[i];
            q_t[(size_t)r0 * 3 + i] = q;
            pq += pts[i] * q;
        }
    }
    __syncthreads();
    for (int i = tid; i < nc3; i += BLOCK) qc_part[(size_t)blockIdx.x * nc3 + i] = qc[i];
    const double t = block_sum(pq, red);
    if (tid == 0) pq_part[blockIdx.x] = t;
}
extern "C" int vican_cg_sweep(const vican_graph_t* g, const double* w, const double* deg_t, const double* p_c,
                              const double* r_t, double* p_t, double* q_t, double* qc_part, double* pq_part,
                              const vican_cg_state_t* st, void* stream) {
    if (int r = vican_check_graph(g, "vican_cg_sweep")) return r;
    if (!w || !deg_t || !p_c || !r_t || !p_t || !q_t || !qc_part || !pq_part || !st)
        return set_err(VICAN_ERR_ARG, "vican_cg_sweep: null pointer");
    const size_t lds = (size_t)cg_lds_bytes(g->n_cam, g->max_rows);
    const int epl = g->slots / g->block_threads;
    hipStream_t s = (hipStream_t)stream;
#define CG_LAUNCH(B, E)                                                                                          \
    do {                                                                                                         \
        auto kern = cg_sweep_kernel<B, E>;                                                                       \
        static size_t conf = 0;                                                                                  \
        if (lds > conf) { hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); conf = lds; } \
        hipLaunchKernelGGL(kern, dim3(g->n_wg), dim3(B), lds, s, *g, w, deg_t, p_c, r_t, p_t, q_t, qc_part, pq_part, st); \
    } while (0)
    if (g->block_threads == 1024)     { if (epl == 4) CG_LAUNCH(1024, 4); else CG_LAUNCH(1024, 2); }
    else if (g->block_threads == 768) { if (epl == 4) CG_LAUNCH(768, 4);  else CG_LAUNCH(768, 2); }
    else if (g->block_threads == 512) { if (epl == 4) CG_LAUNCH(512, 4);  else CG_LAUNCH(512, 2); }
    else                              { if (epl == 4) CG_LAUNCH(256, 4);  else CG_LAUNCH(256, 2); }
#undef CG_LAUNCH
    LAUNCH_CHECK("vican_cg_sweep");
    return VICAN_OK;
}

__global__ void cg_reduce_pq_kernel(const double* __restrict__ pq_part, int n_part, double* __restrict__ out,
                                    const vican_cg_state_t* __restrict__ st) {
    if (st->done) return;
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < n_part; ++i) t += pq_part[i];
        *out = t;
    }
}
extern "C" int vican_cg_reduce_pq(const double* pq_part, int32_t n_part, double* out, const vican_cg_state_t* st,
                                  void* stream) {
    if (!pq_part || n_part <= 0 || !st || !out) return set_err(VICAN_ERR_ARG, "vican_cg_reduce_pq: bad argument");
    hipLaunchKernelGGL(cg_reduce_pq_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, pq_part, n_part, out, st);
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
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const double q = deg_c[i / 3] * p_c[i] - qc_sum[i];
        s += p_c[i] * q;
    }
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
    double rr = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const double p = p_c[i];
        const double q = deg_c[i / 3] * p - qc_sum[i];
        x_c[i] += alpha * p;
        const double r = r_c[i] - alpha * q;
        r_c[i] = r;
        rr += r * r;
    }
    const double t = block_sum(rr, red);
    if (threadIdx.x == 0) st->rr_cam = t;
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
    double rr = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        x_t[i] += alpha * p_t[i];
        const double r = r_t[i] - alpha * q_t[i];
        r_t[i] = r;
        rr += r * r;
    }
    const double t = block_sum(rr, red);
    if (threadIdx.x == 0) rr_part[blockIdx.x] = t;
}
extern "C" int vican_cg_time_step(int32_t n_time, const double* p_t, const double* q_t, double* x_t, double* r_t,
                                  double* rr_part, int32_t part_cap, const vican_cg_state_t* st, void* stream) {
    if (n_time < 0 || !p_t || !q_t || !x_t || !r_t || !rr_part || part_cap <= 0 || !st)
        return set_err(VICAN_ERR_ARG, "vican_cg_time_step: bad argument");
    const long long n = 3LL * n_time;
    int nb = (int)((n + 1023) / 1024); if (nb < 1) nb = 1; if (nb > part_cap) nb = part_cap; if (nb > CG_PARTS) nb = CG_PARTS;
    hipLaunchKernelGGL(cg_time_step_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, n, p_t, q_t, x_t, r_t, rr_part, st);
    LAUNCH_CHECK("vican_cg_time_step");
    return nb;
}

__global__ void cg_end_kernel(const double* __restrict__ rr_part, int n_part, vican_cg_state_t* st) {
    if (st->done) return;
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < n_part; ++i) t += rr_part[i];
        st->rr_time = t; st->iter += 1; st->rho_prev = st->rho; st->first = 0;
    }
}
extern "C" int vican_cg_end(const double* rr_part, int32_t n_part, vican_cg_state_t* st, void* stream) {
    if (!rr_part || n_part <= 0 || !st) return set_err(VICAN_ERR_ARG, "vican_cg_end: bad argument");
    hipLaunchKernelGGL(cg_end_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, rr_part, n_part, st);
    LAUNCH_CHECK("vican_cg_end");
    return VICAN_OK;
}
