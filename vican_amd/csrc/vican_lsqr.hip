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
#include "vican_sweep_common.h"
#include <type_traits>

#define LSQR_PARTS 1024

// ---------------------------------------------------------------------------
// u_1 (unnormalised) = b~ = (Rc^T u_e + Rt^T v_e) / sqrt(w_e);  partial |u|^2 per workgroup
// ---------------------------------------------------------------------------
template <int BLOCK, int EPL>
__global__ __launch_bounds__(BLOCK) void lsqr_init_u_kernel(vican_graph_t g, const double* __restrict__ w,
                                                            const double* __restrict__ ue, const double* __restrict__ ve,
                                                            const double* __restrict__ rc, const double* __restrict__ rt,
                                                            double* __restrict__ u, double* __restrict__ sw,
                                                            double* __restrict__ part) {
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
            double out[3] = {0, 0, 0}, sq = 0.0;
            if (id != VICAN_PAD_SLOT) {
                const uint32_t cam = id & 0xFFFFu, row = id >> 16;
                sq = sqrt(w[e]);
                const double inv_s = 1.0 / sq;
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
            sw[e] = sq;                               // s_e = sqrt(w_e) (0 on padding): the steps below never take a root
        }
    }
    const double t = block_sum(nrm, red);
    if (tid == 0) part[blockIdx.x] = t;
}

// u <- s (v_t - v_c) - coef * u ; partial |u|^2
// Both per-iteration kernels are software pipelines like cg_sweep_kernel (vican_trans.hip): the edge words of
// chunk k+1 (16-byte loads) and the row values of the following chunk are in flight while chunk k is processed,
// one barrier per chunk; sw = sqrt(w) is precomputed by vican_lsqr_init_u.
template <int EPL>
struct LsqrRegs { double u[3][EPL], s[EPL]; uint32_t id[EPL]; };

template <int EPL>
__device__ __forceinline__ void lsqr_load_edges(LsqrRegs<EPL>& e, const vican_graph_t& g, const double* __restrict__ sw,
                                                const double* __restrict__ u, int k, int tid) {
    const size_t s = (size_t)k * g.slots + (size_t)tid * EPL;
    if (EPL == 4) { const uint4 t = *(const uint4*)(g.idx + s); e.id[0] = t.x; e.id[1] = t.y; e.id[2] = t.z; e.id[3] = t.w; }
    else          { const uint2 t = *(const uint2*)(g.idx + s); e.id[0] = t.x; e.id[1] = t.y; }
#pragma unroll
    for (int j = 0; j < EPL; j += 2) { const double2 a = *(const double2*)(sw + s + j); e.s[j] = a.x; e.s[j + 1] = a.y; }
#pragma unroll
    for (int p = 0; p < 3; ++p) {
        const size_t o = ((size_t)k * 3 + p) * g.slots + (size_t)tid * EPL;
#pragma unroll
        for (int j = 0; j < EPL; j += 2) { const double2 a = *(const double2*)(u + o + j); e.u[p][j] = a.x; e.u[p][j + 1] = a.y; }
    }
}

template <int BLOCK, int EPL, int NR>
__global__ __launch_bounds__(BLOCK) void lsqr_u_step_kernel(vican_graph_t g, const double* __restrict__ sw,
                                                            const double* __restrict__ v_c, const double* __restrict__ v_t,
                                                            double coef, double* __restrict__ u, double* __restrict__ part) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int C = g.n_cam, mr3 = 3 * g.max_rows;
    double* vcs = (double*)lds_raw;                 // [3][C] planes
    double* vts = vcs + 3 * C;                      // [2][max_rows][3]
    double* red = vts + 2 * mr3;
    const int tid = threadIdx.x;
    for (int i = tid; i < 3 * C; i += BLOCK) vcs[(i % 3) * C + i / 3] = v_c[i];
    const int k0 = (int)(((long long)blockIdx.x * g.n_chunk) / gridDim.x);
    const int k1 = (int)(((long long)(blockIdx.x + 1) * g.n_chunk) / gridDim.x);
    double rv[NR];
    auto load_rows = [&](int k) {
        const int r0 = g.chunk_row0[k], n3 = 3 * (g.chunk_row0[k + 1] - r0);
#pragma unroll
        for (int m = 0; m < NR; ++m) { const int i = tid + m * BLOCK; rv[m] = i < n3 ? v_t[(size_t)r0 * 3 + i] : 0.0; }
    };
    auto commit_rows = [&](int k, int buf) {
        const int n3 = 3 * (g.chunk_row0[k + 1] - g.chunk_row0[k]);
#pragma unroll
        for (int m = 0; m < NR; ++m) { const int i = tid + m * BLOCK; if (i < n3) vts[buf * mr3 + i] = rv[m]; }
    };
    LsqrRegs<EPL> ea, eb;
    double nrm = 0.0;
    if (k0 < k1) { lsqr_load_edges<EPL>(ea, g, sw, u, k0, tid); load_rows(k0); commit_rows(k0, 0); }
    __syncthreads();
    auto body = [&](LsqrRegs<EPL>& cur, LsqrRegs<EPL>& nxt, const int k, const int buf) {
        if (k + 1 < k1) { lsqr_load_edges<EPL>(nxt, g, sw, u, k + 1, tid); load_rows(k + 1); }
        const double* vt = vts + buf * mr3;
        double un[3][EPL];
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            const bool pad = cur.id[j] == VICAN_PAD_SLOT;
            const uint32_t cam = pad ? 0u : (cur.id[j] & 0xFFFFu), row = pad ? 0u : (cur.id[j] >> 16);
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const double v = pad ? 0.0 : cur.s[j] * (vt[row * 3 + p] - vcs[p * C + cam]) - coef * cur.u[p][j];
                un[p][j] = v;
                nrm += v * v;
            }
        }
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            double* o = u + ((size_t)k * 3 + p) * g.slots + (size_t)tid * EPL;
#pragma unroll
            for (int j = 0; j < EPL; j += 2) *(double2*)(o + j) = make_double2(un[p][j], un[p][j + 1]);
        }
        if (k + 1 < k1) commit_rows(k + 1, buf ^ 1);
        __syncthreads();
    };
#pragma unroll 1
    for (int k = k0; k < k1; k += 2) {
        body(ea, eb, k, 0);
        if (k + 1 < k1) body(eb, ea, k + 1, 1);
    }
    const double t = block_sum(nrm, red);
    if (tid == 0) part[blockIdx.x] = t;
}

// v_raw = J~^T (u * inv_beta) - beta * v : rows finished here (v_t in place, partial |v_t|^2),
// camera side as fixed-point slabs of  -sum_t s u inv_beta.   Bound: |u inv_beta| <= 1, |s| <= smax.
template <int BLOCK, int EPL, int NR>
__global__ __launch_bounds__(BLOCK) void lsqr_v_step_kernel(vican_graph_t g, const double* __restrict__ sw,
                                                            const double* __restrict__ u, double inv_beta, double beta,
                                                            double* __restrict__ v_t, u64* __restrict__ vc_part,
                                                            double* __restrict__ part, double scale, double inv) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int C = g.n_cam, ncopy = g.n_copy, cmask = ncopy - 1, mr3 = 3 * g.max_rows;
    u64* vc = (u64*)lds_raw;                          // [3][C] planes
    u64* vt = vc + 3 * C;                             // [2][max_rows*3][ncopy]
    double* red = (double*)(vt + (size_t)2 * mr3 * ncopy);
    const int tid = threadIdx.x, lane_copy = tid & cmask;
    const uint32_t pad_cam = (uint32_t)((tid & 31) < g.n_cam ? (tid & 31) : 0);
    for (int i = tid; i < 3 * C; i += BLOCK) vc[i] = 0ull;
    for (int i = tid; i < 2 * mr3 * ncopy; i += BLOCK) vt[i] = 0ull;
    const int k0 = (int)(((long long)blockIdx.x * g.n_chunk) / gridDim.x);
    const int k1 = (int)(((long long)(blockIdx.x + 1) * g.n_chunk) / gridDim.x);
    struct RowRegs { double v[NR]; };
    auto load_rows = [&](RowRegs& rr, int k) {          // old v_t of the chunk's rows, consumed by its (deferred) fold
        const int r0 = g.chunk_row0[k], n3 = 3 * (g.chunk_row0[k + 1] - r0);
#pragma unroll
        for (int m = 0; m < NR; ++m) { const int i = tid + m * BLOCK; rr.v[m] = i < n3 ? v_t[(size_t)r0 * 3 + i] : 0.0; }
    };
    double nrm = 0.0;
    auto fold = [&](const RowRegs& rr, const int k, const int buf) {
        const int r0 = g.chunk_row0[k], n3 = 3 * (g.chunk_row0[k + 1] - r0);
        u64* q = vt + (size_t)buf * mr3 * ncopy;
#pragma unroll
        for (int m = 0; m < NR; ++m) {
            const int i = tid + m * BLOCK;
            if (i < n3) {
                long long sum = 0;
                for (int c = 0; c < ncopy; ++c) {
                    const int a = i * ncopy + ((c + i) & cmask);
                    sum += (long long)q[a];
                    q[a] = 0ull;
                }
                const double vn = (double)sum * inv - beta * rr.v[m];
                v_t[(size_t)r0 * 3 + i] = vn;
                nrm += vn * vn;
            }
        }
    };
    LsqrRegs<EPL> ea, eb;
    RowRegs ra, rb;
    if (k0 < k1) { lsqr_load_edges<EPL>(ea, g, sw, u, k0, tid); load_rows(ra, k0); }
    __syncthreads();
    // body(k): edges `cur`; `rs` holds the old v_t of chunk k-1 (folded here) and is then refilled with those of
    // chunk k+1 (consumed one body later, by the fold in body k+2)
    auto body = [&](LsqrRegs<EPL>& cur, LsqrRegs<EPL>& nxt, RowRegs& rs, const int k, const int buf) {
        if (k + 1 < k1) lsqr_load_edges<EPL>(nxt, g, sw, u, k + 1, tid);
        u64* vtb = vt + (size_t)buf * mr3 * ncopy;
        double acc[3] = {0, 0, 0};
        uint32_t prow = 0xFFFFFFFFu;
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            const bool pad = cur.id[j] == VICAN_PAD_SLOT;
            const uint32_t cam = pad ? pad_cam : (cur.id[j] & 0xFFFFu), row = pad ? 0u : (cur.id[j] >> 16);
            const double sq = pad ? 0.0 : cur.s[j] * inv_beta;
            if (row != prow) {
                if (prow != 0xFFFFFFFFu)
#pragma unroll
                    for (int i = 0; i < 3; ++i) lds_add_fix(&vtb[(prow * 3 + i) * ncopy + lane_copy], to_fix(acc[i], scale));
                prow = row; acc[0] = acc[1] = acc[2] = 0.0;
            }
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const double a = sq * cur.u[p][j];
                acc[p] += a;
                lds_add_fix(&vc[p * C + cam], to_fix(-a, scale));
            }
        }
        if (prow != 0xFFFFFFFFu)
#pragma unroll
            for (int i = 0; i < 3; ++i) lds_add_fix(&vtb[(prow * 3 + i) * ncopy + lane_copy], to_fix(acc[i], scale));
        if (k > k0) fold(rs, k - 1, buf ^ 1);             // deferred: overlaps with everybody's edge work
        if (k + 1 < k1) load_rows(rs, k + 1);
        __syncthreads();
    };
    // register sets: chunk k's old rows live in ra for even (k-k0), rb for odd
#pragma unroll 1
    for (int k = k0; k < k1; k += 2) {
        body(ea, eb, rb, k, 0);
        if (k + 1 < k1) body(eb, ea, ra, k + 1, 1);
    }
    if (k0 < k1) { if ((k1 - 1 - k0) & 1) fold(rb, k1 - 1, 1); else fold(ra, k1 - 1, 0); }
    __syncthreads();
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
// the pipelined step kernels additionally carry NR = row values per thread (3 max_rows / block, <= 3 EPL)
#define LSQR_DISPATCH_NR(KERN, LDS, ...)                                                                         \
    do {                                                                                                         \
        const int epl_ = g->slots / g->block_threads;                                                            \
        const int nr_ = (3 * g->max_rows + g->block_threads - 1) / g->block_threads;                             \
        auto launch_ = [&](auto kern, int B) {                                                                   \
            hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(LDS));      \
            hipLaunchKernelGGL(kern, dim3(g->n_wg), dim3(B), (LDS), st, __VA_ARGS__);                            \
        };                                                                                                       \
        auto pick_ = [&](auto b_, auto e_) {                                                                     \
            constexpr int B = decltype(b_)::value, E = decltype(e_)::value;                                      \
            if (nr_ <= 1) launch_(KERN<B, E, 1>, B); else if (nr_ <= 4) launch_(KERN<B, E, 4>, B); else launch_(KERN<B, E, 12>, B); \
        };                                                                                                       \
        using I2 = std::integral_constant<int, 2>; using I4 = std::integral_constant<int, 4>;                    \
        if (g->block_threads == 1024)     { if (epl_ == 4) pick_(std::integral_constant<int, 1024>{}, I4{}); else pick_(std::integral_constant<int, 1024>{}, I2{}); } \
        else if (g->block_threads == 768) { if (epl_ == 4) pick_(std::integral_constant<int, 768>{}, I4{});  else pick_(std::integral_constant<int, 768>{}, I2{}); }  \
        else if (g->block_threads == 512) { if (epl_ == 4) pick_(std::integral_constant<int, 512>{}, I4{});  else pick_(std::integral_constant<int, 512>{}, I2{}); }  \
        else                              { if (epl_ == 4) pick_(std::integral_constant<int, 256>{}, I4{});  else pick_(std::integral_constant<int, 256>{}, I2{}); }  \
    } while (0)

// out[0] = sum part[0..n)   (fixed order)
__global__ void sum_partials_kernel(const double* __restrict__ part, int n, double* __restrict__ out) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < n; ++i) t += part[i];
        out[0] = t;
    }
}

extern "C" __attribute__((visibility("hidden"))) int vican_lsqr_winit(const vican_graph_t* g, const double* w, const double* ue, const double* ve,
                                                                      const double* rc, const double* rt, double* u, double* sw, double* part,
                                                                      void* stream);          // vican_wtrans.hip
extern "C" __attribute__((visibility("hidden"))) int vican_lsqr_wstep(const vican_graph_t* g, const double* sw, double* u, const double* v_c,
                                                                      const double* v_t, double* z_t, void* zc_part, double* part,
                                                                      const vican_lsqr_state_t* st, void* stream);
extern "C" int vican_lsqr_init_u(const vican_graph_t* g, const double* w, const double* ue, const double* ve,
                                 const double* rc, const double* rt, double* u, double* sw, double* part, double* nrm2_out,
                                 void* stream) {
    if (int r = vican_check_graph(g, "vican_lsqr_init_u")) return r;
    if (!w || !ue || !ve || !rc || !rt || !u || !sw || !part || !nrm2_out) return set_err(VICAN_ERR_ARG, "vican_lsqr_init_u: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (g->layout == VICAN_LAYOUT_WAVE) {
        if (int r = vican_lsqr_winit(g, w, ue, ve, rc, rt, u, sw, part, stream)) return r;
        hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(64), 0, st, part, g->n_wg, nrm2_out);
        LAUNCH_CHECK("vican_lsqr_init_u");
        return VICAN_OK;
    }
    const size_t lds = (size_t)8 * (9 * g->n_cam + 9 * g->max_rows + 16);
    LSQR_DISPATCH(lsqr_init_u_kernel, lds, *g, w, ue, ve, rc, rt, u, sw, part);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(64), 0, st, part, g->n_wg, nrm2_out);
    LAUNCH_CHECK("vican_lsqr_init_u");
    return VICAN_OK;
}

extern "C" int vican_lsqr_u_step(const vican_graph_t* g, const double* sw, const double* v_c, const double* v_t,
                                 double coef, double* u, double* part, double* nrm2_out, void* stream) {
    if (int r = vican_check_block_graph(g, "vican_lsqr_u_step")) return r;
    if (!sw || !v_c || !v_t || !u || !part || !nrm2_out) return set_err(VICAN_ERR_ARG, "vican_lsqr_u_step: null pointer");
    if (3 * g->max_rows > 12 * g->block_threads) return set_err(VICAN_ERR_CAPACITY, "vican_lsqr_u_step: more than 4 rows per lane in a chunk");
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = (size_t)8 * (3 * g->n_cam + 6 * g->max_rows + 16);
    LSQR_DISPATCH_NR(lsqr_u_step_kernel, lds, *g, sw, v_c, v_t, coef, u, part);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(64), 0, st, part, g->n_wg, nrm2_out);
    LAUNCH_CHECK("vican_lsqr_u_step");
    return VICAN_OK;
}

extern "C" int vican_lsqr_v_step(const vican_graph_t* g, const double* sw, const double* u, double inv_beta, double beta,
                                 double* v_t, void* vc_part, double* part, double* nrm2_t_out, double smax, double n_add,
                                 double* inv_out, void* stream) {
    if (int r = vican_check_block_graph(g, "vican_lsqr_v_step")) return r;
    if (!sw || !u || !v_t || !vc_part || !part || !nrm2_t_out || !inv_out) return set_err(VICAN_ERR_ARG, "vican_lsqr_v_step: null pointer");
    if (3 * g->max_rows > 12 * g->block_threads) return set_err(VICAN_ERR_CAPACITY, "vican_lsqr_v_step: more than 4 rows per lane in a chunk");
    hipStream_t st = (hipStream_t)stream;
    double c = smax > 1e-300 ? smax : 1e-300;
    int e = 47 - (int)ceil(log2(c));
    const int e2 = 61 - (int)ceil(log2(c * (n_add > 1 ? n_add : 1)));
    if (e2 < e) e = e2;
    const double scale = ldexp(1.0, e), inv = ldexp(1.0, -e);
    *inv_out = inv;
    const size_t lds = (size_t)cg_lds_bytes(g->n_cam, g->max_rows, g->n_copy);
    LSQR_DISPATCH_NR(lsqr_v_step_kernel, lds, *g, sw, u, inv_beta, beta, v_t, (u64*)vc_part, part, scale, inv);
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

// ===========================================================================================================================
// Device-resident LSQR iteration (no host read on the critical path; one pass over the edges per iteration)
// ===========================================================================================================================
// scipy's loop (bidiagonalisation  beta u = A v - alfa u,  alfa v = A^T u - beta v,  then the QR / norm / stopping-test
// scalars) with every scalar in a device struct (vican_lsqr_state_t) and the two edge passes of an iteration FUSED: the edge
// vector is stored UNNORMALISED (u~_i = beta_i u_i), a pass forms  u^ = J~ v_i - (alfa_i / beta_i) u~_i  (= beta_{i+1} u_{i+1}),
// writes it, and in the same pass accumulates |u^|^2 and z = J~^T u^ (row sums written out, camera sums as double-word
// fixed-point slabs); the node-side kernel then finishes  v~ = z / beta_{i+1} - beta_{i+1} v_i.  60 bytes per edge and
// iteration (packed index 4, sqrt(w) 8, u~ read 24 + written 24) instead of 60 + 36 in two passes.  (J~^T u^) / beta and
// J~^T (u^ / beta) differ by rounding only; iterates agree with scipy's to ~1e-15 relative per iteration.
static_assert(sizeof(vican_lsqr_state_t) == 8 * 28 + 4 * 8, "vican_lsqr_state_t layout");

template <int BLOCK, int EPL, int NR>
__global__ __launch_bounds__(BLOCK) void lsqr_step_kernel(vican_graph_t g, const double* __restrict__ sw, double* __restrict__ u,
                                                          const double* __restrict__ v_c, const double* __restrict__ v_t,
                                                          double* __restrict__ z_t, u64* __restrict__ zc_part,
                                                          double* __restrict__ part, const vican_lsqr_state_t* __restrict__ st) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    if (st->done) return;
    const int C = g.n_cam, ncopy = g.n_copy, cmask = ncopy - 1, mr3 = 3 * g.max_rows;
    const int lo_c = 3 * C, lo_t = mr3 * ncopy;
    u64* zc = (u64*)lds_raw;                               // [2][3][C] planes: hi, lo
    u64* zt = zc + 6 * C;                                  // [2 buffers][2][max_rows*3][ncopy]
    double* vcs = (double*)(zt + (size_t)4 * lo_t);        // [3][C] planes
    double* vts = vcs + 3 * C;                             // [2][max_rows*3]
    double* red = vts + 2 * mr3;                           // [16]
    const int tid = threadIdx.x, lane_copy = tid & cmask;
    const uint32_t pad_cam = (uint32_t)((tid & 31) < g.n_cam ? (tid & 31) : 0);
    const double coef = st->coef, scale = st->qscale, inv = st->qinv;
    const int lob = st->lo_bits;
    const double lo_scale = ldexp(1.0, lob);
    for (int i = tid; i < 3 * C; i += BLOCK) { vcs[(i % 3) * C + i / 3] = v_c[i]; zc[i] = 0ull; zc[lo_c + i] = 0ull; }
    for (int i = tid; i < 4 * lo_t; i += BLOCK) zt[i] = 0ull;
    const int k0 = (int)(((long long)blockIdx.x * g.n_chunk) / gridDim.x);
    const int k1 = (int)(((long long)(blockIdx.x + 1) * g.n_chunk) / gridDim.x);
    double rv[NR];
    auto load_rows = [&](int k) {
        const int r0 = g.chunk_row0[k], n3 = 3 * (g.chunk_row0[k + 1] - r0);
#pragma unroll
        for (int m = 0; m < NR; ++m) { const int i = tid + m * BLOCK; rv[m] = i < n3 ? v_t[(size_t)r0 * 3 + i] : 0.0; }
    };
    auto commit_rows = [&](int k, int buf) {
        const int n3 = 3 * (g.chunk_row0[k + 1] - g.chunk_row0[k]);
#pragma unroll
        for (int m = 0; m < NR; ++m) { const int i = tid + m * BLOCK; if (i < n3) vts[buf * mr3 + i] = rv[m]; }
    };
    auto fold = [&](const int k, const int buf) {
        const int r0 = g.chunk_row0[k], n3 = 3 * (g.chunk_row0[k + 1] - r0);
        u64* q = zt + (size_t)buf * 2 * lo_t;
        for (int i = tid; i < n3; i += BLOCK) {
            long long sum = 0, slo = 0;
            for (int c = 0; c < ncopy; ++c) {
                const int a = i * ncopy + ((c + i) & cmask);
                sum += (long long)q[a]; slo += (long long)q[lo_t + a];
                q[a] = 0ull; q[lo_t + a] = 0ull;
            }
            z_t[(size_t)r0 * 3 + i] = fix2_value(sum, slo, lob, inv);
        }
    };
    LsqrRegs<EPL> ea, eb;
    double nrm = 0.0;
    if (k0 < k1) { lsqr_load_edges<EPL>(ea, g, sw, u, k0, tid); load_rows(k0); commit_rows(k0, 0); }
    __syncthreads();
    auto body = [&](LsqrRegs<EPL>& cur, LsqrRegs<EPL>& nxt, const int k, const int buf) {
        if (k + 1 < k1) { lsqr_load_edges<EPL>(nxt, g, sw, u, k + 1, tid); load_rows(k + 1); }
        const double* vt = vts + buf * mr3;
        u64* ztb = zt + (size_t)buf * 2 * lo_t;
        double un[3][EPL], acc[3] = {0, 0, 0};
        uint32_t prow = 0xFFFFFFFFu;
        auto flush_row = [&](const uint32_t r) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const Fix2 f = to_fix2(acc[i], scale, lo_scale);
                u64* a = &ztb[(r * 3 + i) * ncopy + lane_copy];
                lds_add_fix(a, f.hi); lds_add_fix(a + lo_t, f.lo);
            }
        };
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            const bool pad = cur.id[j] == VICAN_PAD_SLOT;
            const uint32_t cam = pad ? pad_cam : (cur.id[j] & 0xFFFFu), row = pad ? 0u : (cur.id[j] >> 16);
            const double sj = pad ? 0.0 : cur.s[j];
            if (row != prow) {
                if (prow != 0xFFFFFFFFu) flush_row(prow);
                prow = row; acc[0] = acc[1] = acc[2] = 0.0;
            }
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const double uh = pad ? 0.0 : sj * (vt[row * 3 + p] - vcs[p * C + cam]) - coef * cur.u[p][j];
                un[p][j] = uh;
                nrm += uh * uh;
                const double a = sj * uh;
                acc[p] += a;
                const Fix2 f = to_fix2(-a, scale, lo_scale);
                lds_add_fix(&zc[p * C + cam], f.hi); lds_add_fix(&zc[lo_c + p * C + cam], f.lo);
            }
        }
        if (prow != 0xFFFFFFFFu) flush_row(prow);
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            double* o = u + ((size_t)k * 3 + p) * g.slots + (size_t)tid * EPL;
#pragma unroll
            for (int j = 0; j < EPL; j += 2) *(double2*)(o + j) = make_double2(un[p][j], un[p][j + 1]);
        }
        if (k > k0) fold(k - 1, buf ^ 1);                  // deferred: overlaps with everybody's edge work
        if (k + 1 < k1) commit_rows(k + 1, buf ^ 1);
        __syncthreads();
    };
#pragma unroll 1
    for (int k = k0; k < k1; k += 2) {
        body(ea, eb, k, 0);
        if (k + 1 < k1) body(eb, ea, k + 1, 1);
    }
    if (k0 < k1) fold(k1 - 1, (k1 - 1 - k0) & 1);
    __syncthreads();
    for (int i = tid; i < 6 * C; i += BLOCK) zc_part[(size_t)blockIdx.x * 6 * C + i] = zc[i];
    const double t = block_sum(nrm, red);
    if (tid == 0) part[blockIdx.x] = t;
}
static inline int64_t lsqr_step_lds_bytes(int32_t n_cam, int32_t max_rows, int32_t n_copy) {
    return 72LL * n_cam + (int64_t)max_rows * (96LL * n_copy + 48) + 256;
}

// camera slabs [n_slab][2][3][C] -> acc[0 : 3C] ([C][3], exact sums rounded once); acc[3C] = sum of the |u^|^2 partials
__global__ __launch_bounds__(1024) void lsqr_fold_kernel(const long long* __restrict__ slab, int n_slab, int n_cam, const double* __restrict__ part,
                                                         double* __restrict__ acc, const vican_lsqr_state_t* __restrict__ st) {
    __shared__ long long sh[3][1024];
    if (st->done) return;
    const long long n = 3LL * n_cam;
    const int lob = st->lo_bits;
    const int e = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const long long i = (long long)blockIdx.x * 64 + e;
    Fix3 a = {0, 0, 0};
    if (i < n)
        for (int k = grp; k < n_slab; k += 16) fix3_add(a, slab[(size_t)k * 2 * n + i], slab[(size_t)k * 2 * n + n + i], lob);
    sh[0][threadIdx.x] = a.top; sh[1][threadIdx.x] = a.bot; sh[2][threadIdx.x] = a.lo;
    __syncthreads();
    if (grp == 0 && i < n) {
        long long t = 0, b = 0, l = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) { t += sh[0][k * 64 + e]; b += sh[1][k * 64 + e]; l += sh[2][k * 64 + e]; }
        const long long q = i / n_cam, cam = i % n_cam;
        acc[cam * 3 + q] = fix3_value(t, b, l, lob, st->qinv);
    }
    if (blockIdx.x == 0) {
        __shared__ double pp[1024];
        __syncthreads();
        double tot = 0.0;
        for (int k0 = 0; k0 < n_slab; k0 += 1024) {
            const int k = k0 + threadIdx.x;
            pp[threadIdx.x] = k < n_slab ? part[k] : 0.0;
            __syncthreads();
            if (threadIdx.x == 0) { const int m = n_slab - k0 < 1024 ? n_slab - k0 : 1024; for (int j = 0; j < m; ++j) tot += pp[j]; }
            __syncthreads();
        }
        if (threadIdx.x == 0) acc[n] = tot;
    }
}

// node side of the fused step: beta' = sqrt(acc[3C]) (all-reduced by the caller when sharded);
// v_t <- z_t / beta' - beta' v_t, v_c <- acc_c / beta' - beta' v_c (block 0); part2[b] = partial |v_t|^2, part2[LSQR_PARTS] = |v_c|^2
__global__ __launch_bounds__(256) void lsqr_nodes_kernel(long long n_t, int n_c, const double* __restrict__ z_t, const double* __restrict__ acc,
                                                         double* __restrict__ v_t, double* __restrict__ v_c, double* __restrict__ part2,
                                                         const vican_lsqr_state_t* __restrict__ st) {
    __shared__ double red[8];
    if (st->done) return;
    const double b2 = acc[n_c];
    const double beta = sqrt(b2), ib = beta > 0.0 ? 1.0 / beta : 0.0;
    double s = 0.0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n_t; i += (long long)gridDim.x * 256) {
        const double vn = z_t[i] * ib - beta * v_t[i];
        v_t[i] = vn; s += vn * vn;
    }
    const double t = block_sum(s, red);
    if (threadIdx.x == 0) part2[blockIdx.x] = t;
    if (blockIdx.x == 0) {
        __syncthreads();
        double sc = 0.0;
        for (int i = threadIdx.x; i < n_c; i += 256) { const double vn = acc[i] * ib - beta * v_c[i]; v_c[i] = vn; sc += vn * vn; }
        const double tc = block_sum(sc, red);
        if (threadIdx.x == 0) part2[LSQR_PARTS] = tc;
    }
}

__device__ __forceinline__ void lsqr_sym_ortho(double a, double b, double& c, double& s, double& r) {
    if (b == 0.0) { c = a > 0 ? 1.0 : (a < 0 ? -1.0 : 0.0); s = 0.0; r = fabs(a); return; }
    if (a == 0.0) { c = 0.0; s = b > 0 ? 1.0 : -1.0; r = fabs(b); return; }
    if (fabs(b) > fabs(a)) { const double tau = a / b; s = (b > 0 ? 1.0 : -1.0) / sqrt(1.0 + tau * tau); c = s * tau; r = b / s; }
    else { const double tau = b / a; c = (a > 0 ? 1.0 : -1.0) / sqrt(1.0 + tau * tau); s = c * tau; r = a / c; }
}
// the scalars of one iteration (scipy.sparse.linalg.lsqr's loop body after the bidiagonalisation step) + stopping tests.
// nv2 = |v~|^2 timestep part = sum of part2[0:n_part] (or *tsum when the caller all-reduced it) + part2[LSQR_PARTS];
// wn2 = |w|^2 of the current direction = sum wpart_t[0:n_wt] + sum wpart_c[0:n_wc] (or *wsum_t when all-reduced).
__global__ __launch_bounds__(256) void lsqr_scalars_kernel(const double* __restrict__ acc_beta2, const double* __restrict__ part2, int n_part,
                                                           const double* __restrict__ tsum, const double* __restrict__ wpart_t, int n_wt,
                                                           const double* __restrict__ wpart_c, int n_wc, const double* __restrict__ wsum_t,
                                                           vican_lsqr_state_t* st) {
    __shared__ double red[8];
    if (st->done) {                                        // (the update of the iteration that raised `done` has run: no more updates)
        if (threadIdx.x == 0) st->update = 0;
        return;
    }
    double a = 0.0, b = 0.0;
    if (!tsum) for (int i = threadIdx.x; i < n_part; i += 256) a += part2[i];
    if (!wsum_t) for (int i = threadIdx.x; i < n_wt; i += 256) b += wpart_t[i];
    for (int i = threadIdx.x; i < n_wc; i += 256) b += wpart_c[i];
    const double sa = block_sum(a, red);
    const double sb = block_sum(b, red);
    if (threadIdx.x != 0) return;
    const double eps = 2.220446049250313e-16;
    const double beta = sqrt(*acc_beta2);
    const double nv2 = (tsum ? *tsum : sa) + part2[LSQR_PARTS];
    const double wnorm2 = sb + (wsum_t ? *wsum_t : 0.0);
    double alfa = st->alfa, anorm = st->anorm;
    if (beta > 0.0) { anorm = sqrt(anorm * anorm + alfa * alfa + beta * beta); alfa = sqrt(nv2); }
    double cs, sn, rho;
    lsqr_sym_ortho(st->rhobar, beta, cs, sn, rho);
    const double theta = sn * alfa;
    const double rhobar = -cs * alfa, phi = cs * st->phibar, phibar = sn * st->phibar, tau = sn * phi;
    const double t1 = phi / rho, t2 = -theta / rho;
    const double ddnorm = st->ddnorm + wnorm2 / (rho * rho);
    const double delta = st->sn2 * rho, gambar = -st->cs2 * rho, rhs = phi - delta * st->z, zbar = rhs / gambar;
    const double xnorm = sqrt(st->xxnorm + zbar * zbar);
    const double gamma = sqrt(gambar * gambar + theta * theta);
    const double z = rhs / gamma;
    const double acond = anorm * sqrt(ddnorm);
    const double rnorm = sqrt(phibar * phibar + st->c2), arnorm = alfa * fabs(tau);
    const double test1 = rnorm / st->bnorm, test2 = arnorm / (anorm * rnorm + eps), test3 = 1.0 / (acond + eps);
    const double tt1 = test1 / (1.0 + anorm * xnorm / st->bnorm), rtol = st->btol + st->atol * anorm * xnorm / st->bnorm;
    const int itn = st->itn + 1;
    int istop = 0;
    if (itn >= st->iter_lim) istop = 7;
    if (1.0 + test3 <= 1.0) istop = 6;
    if (1.0 + test2 <= 1.0) istop = 5;
    if (1.0 + tt1 <= 1.0) istop = 4;
    if (test3 <= st->ctol) istop = 3;
    if (test2 <= st->atol) istop = 2;
    if (test1 <= rtol) istop = 1;
    if (!(rnorm == rnorm)) istop = 8;                      // NaN: stop and report (scipy would run to iter_lim)
    st->coef = beta > 0.0 ? alfa / beta : 0.0;             // next step: u^ = J~ v - (alfa / beta) u~
    st->inv_alfa = alfa > 0.0 ? 1.0 / alfa : 1.0; st->t1 = t1; st->t2 = t2;
    st->alfa = alfa; st->beta = beta; st->anorm = anorm; st->rhobar = rhobar; st->phibar = phibar; st->ddnorm = ddnorm;
    st->cs2 = gambar / gamma; st->sn2 = theta / gamma; st->z = z; st->xxnorm = st->xxnorm + z * z;
    st->rnorm = rnorm; st->arnorm = arnorm; st->acond = acond; st->xnorm = xnorm;
    // fixed-point scale of the next step's sums: |s u^| <= smax (2 smax |v|_max + alfa |u|_max) <= smax (2 smax + alfa)   (|v| = |u| = 1)
    double qi;
    st->qscale = fix_scale(st->smax * (2.0 * st->smax + alfa), st->n_add, &qi, 49);
    st->qinv = qi;
    st->itn = itn; st->istop = istop;
    st->update = 1;                                        // this iteration's x / w update is still to run
    if (istop) st->done = 1;
}

// v *= inv_alfa ; x += t1 w ; w = v + t2 w ; partial |w_new|^2 - coefficients from the state; runs for the iteration whose
// scalars were just computed even if that iteration raised `done` (scipy updates x before it tests)
__global__ __launch_bounds__(256) void lsqr_update_st_kernel(long long n, double* v, double* w, double* x, double* __restrict__ part,
                                                             const vican_lsqr_state_t* __restrict__ st) {
    __shared__ double red[8];
    if (!st->update) return;
    const double inv_alfa = st->inv_alfa, t1 = st->t1, t2 = st->t2;
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

extern "C" int vican_lsqr_step(const vican_graph_t* g, const double* sw, double* u, const double* v_c, const double* v_t, double* z_t,
                               void* zc_part, double* part, double* acc, const vican_lsqr_state_t* state, void* stream) {
    const vican_lsqr_state_t* st = state;
    if (int r = vican_check_graph(g, "vican_lsqr_step")) return r;
    if (!sw || !u || !v_c || !v_t || !z_t || !zc_part || !part || !acc || !st) return set_err(VICAN_ERR_ARG, "vican_lsqr_step: null pointer");
    if (g->layout == VICAN_LAYOUT_WAVE) {
        if (int r = vican_lsqr_wstep(g, sw, u, v_c, v_t, z_t, zc_part, part, st, stream)) return r;
        const long long n = 3LL * g->n_cam;
        hipLaunchKernelGGL(lsqr_fold_kernel, dim3((unsigned)((n + 63) / 64)), dim3(1024), 0, (hipStream_t)stream, (const long long*)zc_part, g->n_wg,
                           g->n_cam, part, acc, st);
        LAUNCH_CHECK("vican_lsqr_step");
        return VICAN_OK;
    }
    if (3 * g->max_rows > 12 * g->block_threads) return set_err(VICAN_ERR_CAPACITY, "vican_lsqr_step: more than 4 rows per lane in a chunk");
    hipStream_t st_ = (hipStream_t)stream;
    const size_t lds = (size_t)lsqr_step_lds_bytes(g->n_cam, g->max_rows, g->n_copy);
    if ((int64_t)lds > vican_lds_limit_bytes()) return set_err(VICAN_ERR_CAPACITY, "vican_lsqr_step: camera tables / row staging do not fit in LDS");
    {
        hipStream_t st = st_;                               // (the dispatch macro launches on `st`)
        LSQR_DISPATCH_NR(lsqr_step_kernel, lds, *g, sw, u, v_c, v_t, z_t, (u64*)zc_part, part, state);
    }
    const long long n = 3LL * g->n_cam;
    hipLaunchKernelGGL(lsqr_fold_kernel, dim3((unsigned)((n + 63) / 64)), dim3(1024), 0, st_, (const long long*)zc_part, g->n_wg, g->n_cam, part, acc, st);
    LAUNCH_CHECK("vican_lsqr_step");
    return VICAN_OK;
}
extern "C" int vican_lsqr_nodes(int32_t n_cam, int32_t n_time, const double* z_t, const double* acc, double* v_t, double* v_c, double* part2,
                                const vican_lsqr_state_t* st, void* stream) {
    if (n_cam <= 0 || n_time < 0 || !z_t || !acc || !v_t || !v_c || !part2 || !st) return set_err(VICAN_ERR_ARG, "vican_lsqr_nodes: bad argument");
    const long long n = 3LL * n_time;
    int nb = (int)((n + 1023) / 1024); if (nb < 1) nb = 1; if (nb > LSQR_PARTS) nb = LSQR_PARTS;
    hipLaunchKernelGGL(lsqr_nodes_kernel, dim3(nb), dim3(256), 0, (hipStream_t)stream, n, 3 * n_cam, z_t, acc, v_t, v_c, part2, st);
    LAUNCH_CHECK("vican_lsqr_nodes");
    return nb;
}
extern "C" int vican_lsqr_scalars(int32_t n_cam, const double* acc, const double* part2, int32_t n_part, const double* tsum,
                                  const double* wpart_t, int32_t n_wt, const double* wpart_c, int32_t n_wc, const double* wsum_t,
                                  vican_lsqr_state_t* st, void* stream) {
    if (n_cam <= 0 || !acc || !part2 || !wpart_c || (!wpart_t && !wsum_t) || !st) return set_err(VICAN_ERR_ARG, "vican_lsqr_scalars: bad argument");
    hipLaunchKernelGGL(lsqr_scalars_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, acc + 3 * (size_t)n_cam, part2, (int)n_part, tsum, wpart_t,
                       (int)n_wt, wpart_c, (int)n_wc, wsum_t, st);
    LAUNCH_CHECK("vican_lsqr_scalars");
    return VICAN_OK;
}
extern "C" int vican_lsqr_update_st(int64_t n, double* v, double* w, double* x, double* part, int32_t last, vican_lsqr_state_t* st,
                                    void* stream) {
    if (n < 0 || !v || !w || !x || !part || !st) return set_err(VICAN_ERR_ARG, "vican_lsqr_update_st: bad argument");
    int nb = (int)((n + 1023) / 1024); if (nb < 1) nb = 1; if (nb > LSQR_PARTS) nb = LSQR_PARTS;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(lsqr_update_st_kernel, dim3(nb), dim3(256), 0, s, (long long)n, v, w, x, part, st);
    (void)last;          // (the flag is cleared by the next vican_lsqr_scalars call once `done` is up: no extra launch)
    LAUNCH_CHECK("vican_lsqr_update_st");
    return nb;
}
