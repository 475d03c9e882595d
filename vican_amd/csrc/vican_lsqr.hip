// vican_lsqr.hip - kernels of the LSQR translation solve (lsqr_solver="direct", reference
// bipgo.py:479-480 = scipy.sparse.linalg.lsqr on the 3E' x 3N incidence matrix J).
//
// LSQR is run on the MERGED system  J~ p = b~  (one 3-row block per merged (camera,timestep)
// edge: s_e (p_t - p_c) = g_e / s_e with s_e = sqrt(w_e), g_e = Rc^T u_e + Rt^T v_e), which has the
// same normal equations J~^T J~ = J^T J, J~^T b~ = J^T b and therefore the same iterates x_k as
// the reference's system; residual norms differ by the constant |b|^2 - |b~|^2, which the host
// driver (vican_amd/solver.py) adds back in the stopping tests.  The Golub-Kahan scalars live on
// the host (two small device->host reads per iteration); everything O(E) runs here:
//   u-step   u <- s (v_t - v_c) - coef * u          (12 + 2*24 algorithmic bytes / edge)
//   v-step   v <- J~^T (u / beta) - beta v           (12 + 24 bytes / edge, fixed-point sums)
#include "common.cuh"

#define LSQR_PARTS 1024

// ---------------------------------------------------------------------------
// u_1 (unnormalised) = b~ = (Rc^T u_e + Rt^T v_e) / sqrt(w_e);  partial |u|^2 per workgroup
// ---------------------------------------------------------------------------
template <int BLOCK, int EPL>
__global__ __launch_bounds__(BLOCK) void lsqr_init_u_kernel(vican_graph_t g, const double* __restrict__ w,
                                                            const double* __restrict__ ue, const double* __restrict__ ve,
                                                            const double* __restrict__ rc, const double* __restrict__ rt,
                                                            double* __restrict__ u, double* __restrict__ part) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int C = g.n_cam;
    double* rcs = (double*)lds_raw;                 // [9][C] planes
    double* rts = rcs + 9 * C;                      // [max_rows][9]
    double* red = rts + 9 * g.max_rows;             // [16]
    const int tid = threadIdx.x;
    for (int i = tid; i < 9 * C; i += BLOCK) rcs[(i % 9) * C + i / 9] = rc[i];
    const int k0 = (int)(((long long)blockIdx.x * g.n_chunk) / gridDim.x);
    const int k1 = (int)(((long long)(blockIdx.x + 1) * g.n_chunk) / gridDim.x);
    double nrm = 0.0;
    for (int k = k0; k < k1; ++k) {
        const int r0 = g.chunk_row0[k], nrows = g.chunk_row0[k + 1] - r0;
        __syncthreads();
        for (int i = tid; i < 9 * nrows; i += BLOCK) rts[i] = rt[(size_t)r0 * 9 + i];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            const int s = tid * EPL + j;
            const size_t e = (size_t)k * g.slots + s;
            const uint32_t id = g.idx[e];
            double out[3] = {0, 0, 0};
            if (id != VICAN_PAD_SLOT) {
                const uint32_t cam = id & 0xFFFFu, row = id >> 16;
                const double inv_s = 1.0 / sqrt(w[e]);
                double uu[3], vv[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) { uu[p] = ue[((size_t)k * 3 + p) * g.slots + s]; vv[p] = ve[((size_t)k * 3 + p) * g.slots + s]; }
                const double* B = rts + row * 9;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const double gi = rcs[(0 * 3 + i) * C + cam] * uu[0] + rcs[(1 * 3 + i) * C + cam] * uu[1] +
                                      rcs[(2 * 3 + i) * C + cam] * uu[2] + B[0 * 3 + i] * vv[0] + B[1 * 3 + i] * vv[1] +
                                      B[2 * 3 + i] * vv[2];
                    out[i] = gi * inv_s;
                    nrm += out[i] * out[i];
                }
            }
#pragma unroll
            for (int p = 0; p < 3; ++p) u[((size_t)k * 3 + p) * g.slots + s] = out[p];
        }
    }
    const double t = block_sum(nrm, red);
    if (tid == 0) part[blockIdx.x] = t;
}

// u <- s (v_t - v_c) - coef * u ; partial |u|^2
template <int BLOCK, int EPL>
__global__ __launch_bounds__(BLOCK) void lsqr_u_step_kernel(vican_graph_t g, const double* __restrict__ w,
                                                            const double* __restrict__ v_c, const double* __restrict__ v_t,
                                                            double coef, double* __restrict__ u, double* __restrict__ part) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int C = g.n_cam;
    double* vcs = (double*)lds_raw;                 // [3][C] planes
    double* vts = vcs + 3 * C;                      // [max_rows][3]
    double* red = vts + 3 * g.max_rows;
    const int tid = threadIdx.x;
    for (int i = tid; i < 3 * C; i += BLOCK) vcs[(i % 3) * C + i / 3] = v_c[i];
    const int k0 = (int)(((long long)blockIdx.x * g.n_chunk) / gridDim.x);
    const int k1 = (int)(((long long)(blockIdx.x + 1) * g.n_chunk) / gridDim.x);
    double nrm = 0.0;
    for (int k = k0; k < k1; ++k) {
        const int r0 = g.chunk_row0[k], nrows = g.chunk_row0[k + 1] - r0;
        __syncthreads();
        for (int i = tid; i < 3 * nrows; i += BLOCK) vts[i] = v_t[(size_t)r0 * 3 + i];
        __syncthreads();
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            const int s = tid * EPL + j;
            const size_t e = (size_t)k * g.slots + s;
            const uint32_t id = g.idx[e];
            if (id == VICAN_PAD_SLOT) continue;
            const uint32_t cam = id & 0xFFFFu, row = id >> 16;
            const double sq = sqrt(w[e]);
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const size_t a = ((size_t)k * 3 + p) * g.slots + s;
                const double un = sq * (vts[row * 3 + p] - vcs[p * C + cam]) - coef * u[a];
                u[a] = un;
                nrm += un * un;
            }
        }
    }
    const double t = block_sum(nrm, red);
    if (tid == 0) part[blockIdx.x] = t;
}

// v_raw = J~^T (u * inv_beta) - beta * v : rows finished here (v_t in place, partial |v_t|^2),
// camera side as fixed-point slabs of  -sum_t s u inv_beta.   Bound: |u inv_beta| <= 1, |s| <= smax.
template <int BLOCK, int EPL>
__global__ __launch_bounds__(BLOCK) void lsqr_v_step_kernel(vican_graph_t g, const double* __restrict__ w,
                                                            const double* __restrict__ u, double inv_beta, double beta,
                                                            double* __restrict__ v_t, u64* __restrict__ vc_part,
                                                            double* __restrict__ part, double scale, double inv) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int C = g.n_cam, ncopy = g.n_copy, cmask = ncopy - 1;
    u64* vc = (u64*)lds_raw;                          // [3][C] planes
    u64* vt = vc + 3 * C;                             // [max_rows*3][ncopy]
    double* red = (double*)(vt + (size_t)3 * g.max_rows * ncopy);
    const int tid = threadIdx.x, lane_copy = tid & cmask;
    for (int i = tid; i < 3 * C; i += BLOCK) vc[i] = 0ull;
    for (int i = tid; i < 3 * g.max_rows * ncopy; i += BLOCK) vt[i] = 0ull;
    const int k0 = (int)(((long long)blockIdx.x * g.n_chunk) / gridDim.x);
    const int k1 = (int)(((long long)(blockIdx.x + 1) * g.n_chunk) / gridDim.x);
    double nrm = 0.0;
    __syncthreads();
    for (int k = k0; k < k1; ++k) {
        const int r0 = g.chunk_row0[k], nrows = g.chunk_row0[k + 1] - r0;
        double acc[3] = {0, 0, 0};
        uint32_t prow = 0xFFFFFFFFu;
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            const int s = tid * EPL + j;
            const size_t e = (size_t)k * g.slots + s;
            const uint32_t id = g.idx[e];
            if (id == VICAN_PAD_SLOT) continue;
            const uint32_t cam = id & 0xFFFFu, row = id >> 16;
            const double sq = sqrt(w[e]) * inv_beta;
            if (row != prow) {
                if (prow != 0xFFFFFFFFu)
#pragma unroll
                    for (int i = 0; i < 3; ++i) lds_add_fix(&vt[(prow * 3 + i) * ncopy + lane_copy], to_fix(acc[i], scale));
                prow = row; acc[0] = acc[1] = acc[2] = 0.0;
            }
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const double a = sq * u[((size_t)k * 3 + p) * g.slots + s];
                acc[p] += a;
                lds_add_fix(&vc[p * C + cam], to_fix(-a, scale));
            }
        }
        if (prow != 0xFFFFFFFFu)
#pragma unroll
            for (int i = 0; i < 3; ++i) lds_add_fix(&vt[(prow * 3 + i) * ncopy + lane_copy], to_fix(acc[i], scale));
        __syncthreads();
        for (int i = tid; i < 3 * nrows; i += BLOCK) {
            long long sum = 0;
            for (int c = 0; c < ncopy; ++c) {
                const int a = i * ncopy + ((c + i) & cmask);
                sum += (long long)vt[a];
                vt[a] = 0ull;
            }
            const size_t gi = (size_t)r0 * 3 + i;
            const double vn = (double)sum * inv - beta * v_t[gi];
            v_t[gi] = vn;
            nrm += vn * vn;
        }
        __syncthreads();
    }
    for (int i = tid; i < 3 * C; i += BLOCK) vc_part[(size_t)blockIdx.x * 3 * C + i] = vc[i];
    const double t = block_sum(nrm, red);
    if (tid == 0) part[blockIdx.x] = t;
}

#define LSQR_DISPATCH(KERN, LDS, ...)                                                                            \
    do {                                                                                                         \
        const int epl_ = g->slots / g->block_threads;                                                            \
        auto launch_ = [&](auto kern, int B) {                                                                   \
            hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS));      \
            hipLaunchKernelGGL(kern, dim3(g->n_wg), dim3(B), (LDS), st, __VA_ARGS__);                            \
        };                                                                                                       \
        if (g->block_threads == 1024)     { if (epl_ == 4) launch_(KERN<1024, 4>, 1024); else launch_(KERN<1024, 2>, 1024); } \
        else if (g->block_threads == 768) { if (epl_ == 4) launch_(KERN<768, 4>, 768);   else launch_(KERN<768, 2>, 768); }   \
        else if (g->block_threads == 512) { if (epl_ == 4) launch_(KERN<512, 4>, 512);   else launch_(KERN<512, 2>, 512); }   \
        else                              { if (epl_ == 4) launch_(KERN<256, 4>, 256);   else launch_(KERN<256, 2>, 256); }   \
    } while (0)

// out[0] = sum part[0..n)   (fixed order)
__global__ void sum_partials_kernel(const double* __restrict__ part, int n, double* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < n; ++i) t += part[i];
        out[0] = t;
    }
}

extern "C" int vican_lsqr_init_u(const vican_graph_t* g, const double* w, const double* ue, const double* ve,
                                 const double* rc, const double* rt, double* u, double* part, double* nrm2_out,
                                 void* stream) {
    if (int r = vican_check_graph(g, "vican_lsqr_init_u")) return r;
    if (!w || !ue || !ve || !rc || !rt || !u || !part || !nrm2_out) return set_err(VICAN_ERR_ARG, "vican_lsqr_init_u: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)8 * (9 * g->n_cam + 9 * g->max_rows + 16);
    LSQR_DISPATCH(lsqr_init_u_kernel, lds, *g, w, ue, ve, rc, rt, u, part);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(64), 0, st, part, g->n_wg, nrm2_out);
    LAUNCH_CHECK("vican_lsqr_init_u");
    return VICAN_OK;
}

extern "C" int vican_lsqr_u_step(const vican_graph_t* g, const double* w, const double* v_c, const double* v_t,
                                 double coef, double* u, double* part, double* nrm2_out, void* stream) {
    if (int r = vican_check_graph(g, "vican_lsqr_u_step")) return r;
    if (!w || !v_c || !v_t || !u || !part || !nrm2_out) return set_err(VICAN_ERR_ARG, "vican_lsqr_u_step: null pointer");
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)8 * (3 * g->n_cam + 3 * g->max_rows + 16);
    LSQR_DISPATCH(lsqr_u_step_kernel, lds, *g, w, v_c, v_t, coef, u, part);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(64), 0, st, part, g->n_wg, nrm2_out);
    LAUNCH_CHECK("vican_lsqr_u_step");
    return VICAN_OK;
}

extern "C" int vican_lsqr_v_step(const vican_graph_t* g, const double* w, const double* u, double inv_beta, double beta,
                                 double* v_t, void* vc_part, double* part, double* nrm2_t_out, double smax, double n_add,
                                 double* inv_out, void* stream) {
    if (int r = vican_check_graph(g, "vican_lsqr_v_step")) return r;
    if (!w || !u || !v_t || !vc_part || !part || !nrm2_t_out || !inv_out) return set_err(VICAN_ERR_ARG, "vican_lsqr_v_step: null pointer");
    hipStream_t st = (hipStream_t)stream;
    double c = smax > 1e-300 ? smax : 1e-300;
    int e = 47 - (int)ceil(log2(c));
    const int e2 = 61 - (int)ceil(log2(c * (n_add > 1 ? n_add : 1)));
    if (e2 < e) e = e2;
    const double scale = ldexp(1.0, e), inv = ldexp(1.0, -e);
    *inv_out = inv;
    const size_t lds = (size_t)cg_lds_bytes(g->n_cam, g->max_rows, g->n_copy);
    LSQR_DISPATCH(lsqr_v_step_kernel, lds, *g, w, u, inv_beta, beta, v_t, (u64*)vc_part, part, scale, inv);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(64), 0, st, part, g->n_wg, nrm2_t_out);
    LAUNCH_CHECK("vican_lsqr_v_step");
    return VICAN_OK;
}

// camera side of the v-step: v_c <- acc - beta v_c ; out[0] = |v_c|^2        (acc: reduced slabs)
__global__ __launch_bounds__(256) void lsqr_cam_v_kernel(int n, const double* __restrict__ acc, double beta, double* v_c,
                                                         double* __restrict__ out) {
    __shared__ double red[8];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) { const double vn = acc[i] - beta * v_c[i]; v_c[i] = vn; s += vn * vn; }
    const double t = block_sum(s, red);
    if (threadIdx.x == 0) out[0] = t;
}
extern "C" int vican_lsqr_cam_v(int32_t n_cam, const double* acc, double beta, double* v_c, double* nrm2_out, void* stream) {
    if (n_cam <= 0 || !acc || !v_c || !nrm2_out) return set_err(VICAN_ERR_ARG, "vican_lsqr_cam_v: bad argument");
    hipLaunchKernelGGL(lsqr_cam_v_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, 3 * n_cam, acc, beta, v_c, nrm2_out);
    LAUNCH_CHECK("vican_lsqr_cam_v");
    return VICAN_OK;
}

// v *= inv_alfa ; x += t1 w ; w = v + t2 w ; partial |w_new|^2     (any length n)
__global__ __launch_bounds__(256) void lsqr_update_kernel(long long n, double inv_alfa, double t1, double t2, double* v,
                                                          double* w, double* x, double* __restrict__ part) {
    __shared__ double red[8];
    double s = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const double vn = v[i] * inv_alfa, wo = w[i];
        v[i] = vn;
        x[i] += t1 * wo;
        const double wn = vn + t2 * wo;
        w[i] = wn;
        s += wn * wn;
    }
    const double t = block_sum(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = t;
}
extern "C" int vican_lsqr_update(int64_t n, double inv_alfa, double t1, double t2, double* v, double* w, double* x,
                                 double* part, double* nrm2_w_out, void* stream) {
    if (n < 0 || !v || !w || !x || !part || !nrm2_w_out) return set_err(VICAN_ERR_ARG, "vican_lsqr_update: bad argument");
    int nb = (int)((n + 1023) / 1024); if (nb < 1) nb = 1; if (nb > LSQR_PARTS) nb = LSQR_PARTS;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(lsqr_update_kernel, dim3(nb), dim3(256), 0, st, (long long)n, inv_alfa, t1, t2, v, w, x, part);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(64), 0, st, part, nb, nrm2_w_out);
    LAUNCH_CHECK("vican_lsqr_update");
    return VICAN_OK;
}
