// vican_kernels.hip - hand-written CDNA4 (gfx950) kernels of the primal-dual
// bipartite SE(3) solver.  C ABI in include/vican_hip.h; design notes in DESIGN.md.
//
// Nothing here is GEMM-shaped: the hot loops stream 3x3 edge blocks from HBM once
// per sweep (coalesced 16 B/lane plane loads), keep the camera-side vectors in
// LDS, and reduce with LDS atomics + wavefront shuffles.  64-wide wavefronts
// throughout; no MFMA, no CUDA compatibility paths.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "vican_hip.h"

// ---------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int set_err(int code, const char* fmt, const char* a = "") {
    snprintf(g_err, sizeof(g_err), fmt, a);
    return code;
}
#define LAUNCH_CHECK(name)                                                         \
    do {                                                                           \
        hipError_t e_ = hipGetLastError();                                         \
        if (e_ != hipSuccess) {                                                    \
            snprintf(g_err, sizeof(g_err), "%s: %s", name, hipGetErrorString(e_)); \
            return VICAN_ERR_LAUNCH;                                               \
        }                                                                          \
    } while (0)

extern "C" const char* vican_last_error(void) { return g_err; }
extern "C" int vican_abi_version(void) { return 1; }

// ---------------------------------------------------------------------------
// small device helpers
// ---------------------------------------------------------------------------
#define WAVE 64

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = WAVE / 2; o > 0; o >>= 1) v += __shfl_down(v, o, WAVE);
    return v;
}

// Sum over the workgroup; result valid in thread 0.  `red` holds >= blockDim/64 doubles.
__device__ __forceinline__ double block_sum(double v, double* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0) {
        const int nw = (blockDim.x + WAVE - 1) / WAVE;
        for (int i = 0; i < nw; ++i) t += red[i];   // fixed order
    }
    return t;
}

__device__ __forceinline__ void lds_add(double* p, double v) {
    // ds_add_f64 (no return) under -munsafe-fp-atomics
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// --- 3x3 SVD by one-sided Jacobi (Hestenes), double precision ---------------
// A (row-major) = U diag(s) V^T, s sorted descending.  High relative accuracy
// (no A^T A squaring); U completed to an orthonormal basis when A is rank deficient.
__device__ void svd3(const double* A, double* U, double* s, double* V) {
    double a[3][3], v[3][3];   // a[j] = column j of the working matrix, v[j] = column j of V
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int i = 0; i < 3; ++i) { a[j][i] = A[i * 3 + j]; v[j][i] = (i == j) ? 1.0 : 0.0; }
    for (int sweep = 0; sweep < 30; ++sweep) {
        bool rotated = false;
#pragma unroll
        for (int pq = 0; pq < 3; ++pq) {
            const int p = (pq == 2) ? 1 : 0, q = (pq == 0) ? 1 : 2;
            const double al = a[p][0] * a[p][0] + a[p][1] * a[p][1] + a[p][2] * a[p][2];
            const double be = a[q][0] * a[q][0] + a[q][1] * a[q][1] + a[q][2] * a[q][2];
            const double ga = a[p][0] * a[q][0] + a[p][1] * a[q][1] + a[p][2] * a[q][2];
            if (ga != 0.0 && fabs(ga) > 1e-16 * sqrt(al * be)) {
                rotated = true;
                const double zeta = (be - al) / (2.0 * ga);
                const double t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const double ap = a[p][i], aq = a[q][i];
                    a[p][i] = c * ap - sn * aq;
                    a[q][i] = sn * ap + c * aq;
                    const double vp = v[p][i], vq = v[q][i];
                    v[p][i] = c * vp - sn * vq;
                    v[q][i] = sn * vp + c * vq;
                }
            }
        }
        if (!rotated) break;
    }
    double n[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) n[j] = sqrt(a[j][0] * a[j][0] + a[j][1] * a[j][1] + a[j][2] * a[j][2]);
    // sort descending (3-element network), permuting columns of a and v together
#define SWAPCOL(x, y)                                                      \
    if (n[x] < n[y]) {                                                     \
        double tn = n[x]; n[x] = n[y]; n[y] = tn;                          \
        for (int i = 0; i < 3; ++i) {                                      \
            double ta = a[x][i]; a[x][i] = a[y][i]; a[y][i] = ta;          \
            double tv = v[x][i]; v[x][i] = v[y][i]; v[y][i] = tv;          \
        }                                                                  \
    }
    SWAPCOL(0, 1) SWAPCOL(1, 2) SWAPCOL(0, 1)
#undef SWAPCOL
    double u[3][3];
    const double tiny = 1e-300;
    if (n[0] > tiny) { for (int i = 0; i < 3; ++i) u[0][i] = a[0][i] / n[0]; }
    else { u[0][0] = 1.0; u[0][1] = 0.0; u[0][2] = 0.0; }
    if (n[1] > tiny && n[1] > 1e-15 * n[0]) { for (int i = 0; i < 3; ++i) u[1][i] = a[1][i] / n[1]; }
    else {   // any unit vector orthogonal to u0
        int k = 0; double m = fabs(u[0][0]);
        if (fabs(u[0][1]) < m) { k = 1; m = fabs(u[0][1]); }
        if (fabs(u[0][2]) < m) { k = 2; }
        double e[3] = {0.0, 0.0, 0.0}; e[k] = 1.0;
        const double d = u[0][k];
        double w[3] = {e[0] - d * u[0][0], e[1] - d * u[0][1], e[2] - d * u[0][2]};
        const double wn = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
        for (int i = 0; i < 3; ++i) u[1][i] = w[i] / wn;
    }
    if (n[2] > tiny && n[2] > 1e-15 * n[0]) { for (int i = 0; i < 3; ++i) u[2][i] = a[2][i] / n[2]; }
    else {   // u2 = u0 x u1 (sign is irrelevant for U diag(1,1,det) V^T and U f(S) U^T)
        u[2][0] = u[0][1] * u[1][2] - u[0][2] * u[1][1];
        u[2][1] = u[0][2] * u[1][0] - u[0][0] * u[1][2];
        u[2][2] = u[0][0] * u[1][1] - u[0][1] * u[1][0];
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        s[j] = n[j];
#pragma unroll
        for (int i = 0; i < 3; ++i) { U[i * 3 + j] = u[j][i]; V[i * 3 + j] = v[j][i]; }
    }
}

__device__ __forceinline__ double det3(const double* m) {
    return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) +
           m[2] * (m[3] * m[7] - m[4] * m[6]);
}

// Polar rotation with det fix and dual block from one SVD.
// mode: 0 none, 1 lam = U S U^T, 2 lam = U S^-1 U^T   (S NOT sign corrected: bipgo.py:312,329)
__device__ void polar_dual3(const double* A, double* R, double* lam, int mode) {
    double U[9], s[3], V[9];
    svd3(A, U, s, V);
    const double d = (det3(U) * det3(V) < 0.0) ? -1.0 : 1.0;
    if (R) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                R[i * 3 + j] = U[i * 3 + 0] * V[j * 3 + 0] + U[i * 3 + 1] * V[j * 3 + 1] +
                               d * U[i * 3 + 2] * V[j * 3 + 2];
    }
    if (lam && mode) {
        double f[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) f[k] = (mode == 1) ? s[k] : 1.0 / s[k];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                lam[i * 3 + j] = f[0] * U[i * 3 + 0] * U[j * 3 + 0] + f[1] * U[i * 3 + 1] * U[j * 3 + 1] +
                                 f[2] * U[i * 3 + 2] * U[j * 3 + 2];
    }
}

// ---------------------------------------------------------------------------
// host-side planning
// ---------------------------------------------------------------------------
extern "C" int vican_plan_chunks(int32_t n_time, const int32_t* rp, int32_t slots, int32_t max_rows,
                                 int32_t* out, int32_t cap) {
    if (n_time < 0 || !rp || slots <= 0 || max_rows <= 0 || max_rows > 65535)
        return set_err(VICAN_ERR_ARG, "vican_plan_chunks: bad argument");
    int32_t nc = 0, r = 0;
    while (r < n_time) {
        if (out) { if (nc >= cap) return set_err(VICAN_ERR_ARG, "vican_plan_chunks: output too small"); out[nc] = r; }
        const int32_t e0 = rp[r];
        int32_t r1 = r;
        while (r1 < n_time && (r1 - r) < max_rows && (rp[r1 + 1] - e0) <= slots) ++r1;
        if (r1 == r) return set_err(VICAN_ERR_CAPACITY, "vican_plan_chunks: a timestep row has more edges than a chunk holds");
        r = r1;
        ++nc;
    }
    if (out) { if (nc >= cap) return set_err(VICAN_ERR_ARG, "vican_plan_chunks: output too small"); out[nc] = n_time; }
    return nc;
}

extern "C" int64_t vican_sweep_lds_bytes(int32_t n_cam, int32_t max_rows) {
    // x table 9C + z accumulators 9C + per-row staging 9*max_rows, doubles; + reduction scratch
    return (int64_t)8 * (18LL * n_cam + 9LL * max_rows) + 256;
}
extern "C" int64_t vican_lds_limit_bytes(void) { return 160 * 1024; }
static int64_t rhs_lds_bytes(int32_t n_cam, int32_t max_rows) { return (int64_t)8 * (12LL * n_cam + 12LL * max_rows) + 256; }
static int64_t cg_lds_bytes(int32_t n_cam, int32_t max_rows) { return (int64_t)8 * (6LL * n_cam + 6LL * max_rows + 16); }
extern "C" int32_t vican_max_rows_for(int32_t n_cam) {
    const int64_t lim = vican_lds_limit_bytes();
    int64_t a = (lim - 256 - 144LL * n_cam) / 72, b = (lim - 256 - 96LL * n_cam) / 96, c = (lim - 128 - 48LL * n_cam) / 48;
    int64_t m = a < b ? a : b; if (c < m) m = c; if (m > 65535) m = 65535;
    return (int32_t)m;
}

static int check_graph(const vican_graph_t* g, const char* who) {
    if (!g || g->n_cam <= 0 || g->n_cam > 65535 || g->n_time < 0 || g->n_chunk < 0 || !g->idx || !g->blk ||
        !g->chunk_row0 || g->n_wg <= 0)
        return set_err(VICAN_ERR_ARG, "%s: bad graph descriptor", who);
    const int epl = (g->storage == VICAN_STORE_F32) ? 4 : 2;
    if ((g->block_threads != 256 && g->block_threads != 1024) || g->slots != g->block_threads * epl)
        return set_err(VICAN_ERR_ARG, "%s: slots must be block_threads * (16 / sizeof(storage))", who);
    if (g->max_rows <= 0 || g->max_rows > 65535) return set_err(VICAN_ERR_ARG, "%s: bad max_rows", who);
    if (g->max_rows > vican_max_rows_for(g->n_cam))
        return set_err(VICAN_ERR_CAPACITY, "%s: camera tables do not fit in LDS", who);
    return 0;
}

// ---------------------------------------------------------------------------
// layout: CSR -> chunked planes
// ---------------------------------------------------------------------------
template <typename S>
__global__ void pack_edges_kernel(vican_graph_t g, const int32_t* __restrict__ row_ptr,
                                  const int32_t* __restrict__ col, const S* __restrict__ blk_csr,
                                  const S* __restrict__ a_csr, const double* __restrict__ w_csr,
                                  const double* __restrict__ u_csr, const double* __restrict__ v_csr,
                                  S* __restrict__ a_out, double* __restrict__ w_out,
                                  double* __restrict__ u_out, double* __restrict__ v_out) {
    const int k = blockIdx.y;
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= g.slots) return;
    const int r0 = g.chunk_row0[k], r1 = g.chunk_row0[k + 1];
    const int e0 = row_ptr[r0], e1 = row_ptr[r1];
    const int e = e0 + s;
    S* blk = (S*)g.blk;
    uint32_t* idx = (uint32_t*)g.idx;
    const size_t base = (size_t)k * g.slots + s;
    if (e < e1) {
        int lo = r0, hi = r1;           // row_ptr[lo] <= e < row_ptr[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (row_ptr[mid] <= e) lo = mid; else hi = mid;
        }
        idx[base] = (uint32_t)col[e] | ((uint32_t)(lo - r0) << 16);
#pragma unroll
        for (int p = 0; p < 9; ++p) blk[((size_t)k * 9 + p) * g.slots + s] = blk_csr[(size_t)e * 9 + p];
        if (a_out) a_out[base] = a_csr[e];
        if (w_out) w_out[base] = w_csr[e];
        if (u_out)
            for (int p = 0; p < 3; ++p) u_out[((size_t)k * 3 + p) * g.slots + s] = u_csr[(size_t)e * 3 + p];
        if (v_out)
            for (int p = 0; p < 3; ++p) v_out[((size_t)k * 3 + p) * g.slots + s] = v_csr[(size_t)e * 3 + p];
    } else {
        idx[base] = VICAN_PAD_SLOT;
#pragma unroll
        for (int p = 0; p < 9; ++p) blk[((size_t)k * 9 + p) * g.slots + s] = (S)0;
        if (a_out) a_out[base] = (S)0;
        if (w_out) w_out[base] = 0.0;
        if (u_out) for (int p = 0; p < 3; ++p) u_out[((size_t)k * 3 + p) * g.slots + s] = 0.0;
        if (v_out) for (int p = 0; p < 3; ++p) v_out[((size_t)k * 3 + p) * g.slots + s] = 0.0;
    }
}

extern "C" int vican_pack_edges(const vican_graph_t* g, const int32_t* row_ptr, const int32_t* col,
                                const void* blk_csr, const void* a_csr, const double* w_csr,
                                const double* u_csr, const double* v_csr, void* a_out, double* w_out,
                                double* u_out, double* v_out, void* stream) {
    if (int rc = check_graph(g, "vican_pack_edges")) return rc;
    if (!row_ptr || !col || !blk_csr) return set_err(VICAN_ERR_ARG, "vican_pack_edges: null input");
    if (g->n_chunk == 0) return VICAN_OK;
    dim3 grid((g->slots + 255) / 256, g->n_chunk), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (g->storage == VICAN_STORE_F32)
        hipLaunchKernelGGL(pack_edges_kernel<float>, grid, block, 0, st, *g, row_ptr, col, (const float*)blk_csr,
                           (const float*)a_csr, w_csr, u_csr, v_csr, (float*)a_out, w_out, u_out, v_out);
    else
        hipLaunchKernelGGL(pack_edges_kernel<double>, grid, block, 0, st, *g, row_ptr, col, (const double*)blk_csr,
                           (const double*)a_csr, w_csr, u_csr, v_csr, (double*)a_out, w_out, u_out, v_out);
    LAUNCH_CHECK("vican_pack_edges");
    return VICAN_OK;
}

// ---------------------------------------------------------------------------
// per-chunk scalar sums: row sums (written) + camera sums (global atomics, one-off)
// ---------------------------------------------------------------------------
template <typename S, bool IDENTITY_OUT>
__global__ void row_cam_sums_kernel(vican_graph_t g, const S* __restrict__ val, double* __restrict__ row_out,
                                    double* __restrict__ cam_acc) {
    extern __shared__ double lds[];
    const int k = blockIdx.x;
    const int r0 = g.chunk_row0[k], nrows = g.chunk_row0[k + 1] - r0;
    for (int r = threadIdx.x; r < nrows; r += blockDim.x) lds[r] = 0.0;
    __syncthreads();
    const size_t base = (size_t)k * g.slots;
    for (int s = threadIdx.x; s < g.slots; s += blockDim.x) {
        const uint32_t id = g.idx[base + s];
        if (id == VICAN_PAD_SLOT) continue;
        const double v = (double)val[base + s];
        lds_add(&lds[id >> 16], v);
        unsafeAtomicAdd(&cam_acc[id & 0xFFFFu], v);
    }
    __syncthreads();
    for (int r = threadIdx.x; r < nrows; r += blockDim.x) {
        const double d = lds[r];
        if (IDENTITY_OUT) {
            double* o = row_out + (size_t)(r0 + r) * 9;
            const double inv = 1.0 / d;
            o[0] = inv; o[1] = 0; o[2] = 0; o[3] = 0; o[4] = inv; o[5] = 0; o[6] = 0; o[7] = 0; o[8] = inv;
        } else {
            row_out[r0 + r] = d;
        }
    }
}

extern "C" int vican_init_duals(const vican_graph_t* g, const void* a, double* lamT_inv, double* cam_deg,
                                void* stream) {
    if (int rc = check_graph(g, "vican_init_duals")) return rc;
    if (!a || !lamT_inv || !cam_deg) return set_err(VICAN_ERR_ARG, "vican_init_duals: null pointer");
    if (g->n_chunk == 0) return VICAN_OK;
    const size_t lds = (size_t)g->max_rows * 8;
    if (g->storage == VICAN_STORE_F32)
        hipLaunchKernelGGL((row_cam_sums_kernel<float, true>), dim3(g->n_chunk), dim3(256), lds, (hipStream_t)stream,
                           *g, (const float*)a, lamT_inv, cam_deg);
    else
        hipLaunchKernelGGL((row_cam_sums_kernel<double, true>), dim3(g->n_chunk), dim3(256), lds, (hipStream_t)stream,
                           *g, (const double*)a, lamT_inv, cam_deg);
    LAUNCH_CHECK("vican_init_duals");
    return VICAN_OK;
}

extern "C" int vican_trans_degrees(const vican_graph_t* g, const double* w, double* deg_t, double* deg_c,
                                   void* stream) {
    if (int rc = check_graph(g, "vican_trans_degrees")) return rc;
    if (!w || !deg_t || !deg_c) return set_err(VICAN_ERR_ARG, "vican_trans_degrees: null pointer");
    if (g->n_chunk == 0) return VICAN_OK;
    hipLaunchKernelGGL((row_cam_sums_kernel<double, false>), dim3(g->n_chunk), dim3(256), (size_t)g->max_rows * 8,
                       (hipStream_t)stream, *g, w, deg_t, deg_c);
    LAUNCH_CHECK("vican_trans_degrees");
    return VICAN_OK;
}

__global__ void scaled_identity_kernel(int n, const double* __restrict__ sc, double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double s = sc[i];
    double* o = out + (size_t)i * 9;
    o[0] = s; o[1] = 0; o[2] = 0; o[3] = 0; o[4] = s; o[5] = 0; o[6] = 0; o[7] = 0; o[8] = s;
}
extern "C" int vican_scaled_identity(int32_t n, const double* scale, double* out, void* stream) {
    if (n < 0 || !scale || !out) return set_err(VICAN_ERR_ARG, "vican_scaled_identity: bad argument");
    if (n == 0) return VICAN_OK;
    hipLaunchKernelGGL(scaled_identity_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, scale, out);
    LAUNCH_CHECK("vican_scaled_identity");
    return VICAN_OK;
}

// ---------------------------------------------------------------------------
// THE HOT KERNEL: fused timestep-major block operator / dual update
// ---------------------------------------------------------------------------
template <typename S> struct Vec;
template <> struct Vec<float>  { typedef float4  type; static constexpr int N = 4; };
template <> struct Vec<double> { typedef double2 type; static constexpr int N = 2; };

template <typename S> __device__ __forceinline__ double vget(const typename Vec<S>::type& v, int j);
template <> __device__ __forceinline__ double vget<float>(const float4& v, int j) {
    return (double)(j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w);
}
template <> __device__ __forceinline__ double vget<double>(const double2& v, int j) { return j == 0 ? v.x : v.y; }

// MODE 0: zpart[wg] = sum M * (lamT_inv * (sum M^T x))      (operator P x)
// MODE 1: per row SVD of (sum M^T x) -> Rt, lamT_inv          (dual update)
template <typename S, int BLOCK, int MODE>
__global__ __launch_bounds__(BLOCK) void block_sweep_kernel(vican_graph_t g, const double* __restrict__ lamT_inv,
                                                            const double* __restrict__ x,
                                                            double* __restrict__ zpart,
                                                            double* __restrict__ Rt_out,
                                                            double* __restrict__ lamT_out) {
    typedef typename Vec<S>::type V;
    constexpr int EPL = Vec<S>::N;
    extern __shared__ double lds[];
    const int nx = 9 * g.n_cam;
    double* xs = lds;                // [C][3][3] camera-side input vectors
    double* zs = lds + nx;           // [C][3][3] camera-side accumulators
    double* ys = lds + 2 * nx;       // [max_rows][3][3] per-row staging (y, then w)
    const int tid = threadIdx.x;

    for (int i = tid; i < nx; i += BLOCK) { xs[i] = x[i]; if (MODE == 0) zs[i] = 0.0; }

    const int k0 = (int)(((long long)blockIdx.x * g.n_chunk) / gridDim.x);
    const int k1 = (int)(((long long)(blockIdx.x + 1) * g.n_chunk) / gridDim.x);
    const S* __restrict__ blk = (const S*)g.blk;

    for (int k = k0; k < k1; ++k) {
        const int r0 = g.chunk_row0[k];
        const int nrows = g.chunk_row0[k + 1] - r0;
        // issue the global loads first so they overlap the LDS zeroing + barrier
        V m[9];
        const size_t pbase = (size_t)k * 9 * g.slots + (size_t)tid * EPL;
#pragma unroll
        for (int p = 0; p < 9; ++p) m[p] = *(const V*)(blk + pbase + (size_t)p * g.slots);
        const uint32_t* ip = g.idx + (size_t)k * g.slots + (size_t)tid * EPL;
        uint32_t id[EPL];
        if (EPL == 4) { const uint4 t = *(const uint4*)ip; id[0] = t.x; id[1] = t.y; id[2] = t.z; id[3] = t.w; }
        else          { const uint2 t = *(const uint2*)ip; id[0] = t.x; id[1] = t.y; }

        __syncthreads();                       // previous chunk's phase 3 done with ys
        for (int i = tid; i < 9 * nrows; i += BLOCK) ys[i] = 0.0;
        __syncthreads();

        // ---- phase 1: y_row += M^T x_cam, consecutive same-row edges pre-summed in registers
        {
            double acc[9];
            uint32_t cur = 0xFFFFFFFFu;
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
                if (id[j] == VICAN_PAD_SLOT) continue;
                const uint32_t cam = id[j] & 0xFFFFu, row = id[j] >> 16;
                const double* xc = xs + cam * 9;
                double c[9];
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int b = 0; b < 3; ++b)
                        c[a * 3 + b] = vget<S>(m[0 + a], j) * xc[b] + vget<S>(m[3 + a], j) * xc[3 + b] +
                                       vget<S>(m[6 + a], j) * xc[6 + b];
                if (row != cur) {
                    if (cur != 0xFFFFFFFFu) {
#pragma unroll
                        for (int q = 0; q < 9; ++q) lds_add(&ys[cur * 9 + q], acc[q]);
                    }
                    cur = row;
#pragma unroll
                    for (int q = 0; q < 9; ++q) acc[q] = c[q];
                } else {
#pragma unroll
                    for (int q = 0; q < 9; ++q) acc[q] += c[q];
                }
            }
            if (cur != 0xFFFFFFFFu) {
#pragma unroll
                for (int q = 0; q < 9; ++q) lds_add(&ys[cur * 9 + q], acc[q]);
            }
        }
        __syncthreads();

        // ---- phase 2: per row, w = lamT_inv * y   (or SVD -> Rt, lamT_inv)
        for (int r = tid; r < nrows; r += BLOCK) {
            double y[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) y[q] = ys[r * 9 + q];
            if (MODE == 0) {
                const double* L = lamT_inv + (size_t)(r0 + r) * 9;
                double l[9];
#pragma unroll
                for (int q = 0; q < 9; ++q) l[q] = L[q];
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int b = 0; b < 3; ++b)
                        ys[r * 9 + a * 3 + b] = l[a * 3 + 0] * y[b] + l[a * 3 + 1] * y[3 + b] + l[a * 3 + 2] * y[6 + b];
            } else {
                double R[9], lam[9];
                polar_dual3(y, R, lam, 2);
                double* Ro = Rt_out + (size_t)(r0 + r) * 9;
                double* Lo = lamT_out + (size_t)(r0 + r) * 9;
#pragma unroll
                for (int q = 0; q < 9; ++q) { Ro[q] = R[q]; Lo[q] = lam[q]; }
            }
        }
        if (MODE == 0) {
            __syncthreads();
            // ---- phase 3: z_cam += M w_row   (blocks still in registers: read from HBM once)
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
                if (id[j] == VICAN_PAD_SLOT) continue;
                const uint32_t cam = id[j] & 0xFFFFu, row = id[j] >> 16;
                const double* w = ys + row * 9;
                double* zc = zs + cam * 9;
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int b = 0; b < 3; ++b)
                        lds_add(&zc[i * 3 + b], vget<S>(m[i * 3 + 0], j) * w[b] + vget<S>(m[i * 3 + 1], j) * w[3 + b] +
                                                    vget<S>(m[i * 3 + 2], j) * w[6 + b]);
            }
        }
    }
    if (MODE == 0) {
        __syncthreads();
        double* zp = zpart + (size_t)blockIdx.x * nx;
        for (int i = tid; i < nx; i += BLOCK) zp[i] = zs[i];
    }
}

template <typename S, int BLOCK, int MODE>
static int launch_sweep(const vican_graph_t* g, const double* lamT_inv, const double* x, double* zpart, double* Rt,
                        double* lamT_out, hipStream_t st) {
    const size_t lds = (size_t)vican_sweep_lds_bytes(g->n_cam, g->max_rows);
    auto kern = block_sweep_kernel<S, BLOCK, MODE>;
    static size_t configured = 0;       // per instantiation
    if (lds > configured) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return set_err(VICAN_ERR_LAUNCH, "%s: cannot raise dynamic LDS limit", "vican sweep");
        configured = lds;
    }
    hipLaunchKernelGGL(kern, dim3(g->n_wg), dim3(BLOCK), lds, st, *g, lamT_inv, x, zpart, Rt, lamT_out);
    return 0;
}

template <int MODE>
static int dispatch_sweep(const vican_graph_t* g, const double* lamT_inv, const double* x, double* zpart, double* Rt,
                          double* lamT_out, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    int rc;
    if (g->storage == VICAN_STORE_F32)
        rc = (g->block_threads == 1024) ? launch_sweep<float, 1024, MODE>(g, lamT_inv, x, zpart, Rt, lamT_out, st)
                                        : launch_sweep<float, 256, MODE>(g, lamT_inv, x, zpart, Rt, lamT_out, st);
    else
        rc = (g->block_threads == 1024) ? launch_sweep<double, 1024, MODE>(g, lamT_inv, x, zpart, Rt, lamT_out, st)
                                        : launch_sweep<double, 256, MODE>(g, lamT_inv, x, zpart, Rt, lamT_out, st);
    return rc;
}

extern "C" int vican_block_op(const vican_graph_t* g, const double* lamT_inv, const double* x, double* zpart,
                              void* stream) {
    if (int rc = check_graph(g, "vican_block_op")) return rc;
    if (!lamT_inv || !x || !zpart) return set_err(VICAN_ERR_ARG, "vican_block_op: null pointer");
    if (int rc = dispatch_sweep<0>(g, lamT_inv, x, zpart, nullptr, nullptr, stream)) return rc;
    LAUNCH_CHECK("vican_block_op");
    return VICAN_OK;
}

extern "C" int vican_dual_update(const vican_graph_t* g, const double* rc_, double* Rt, double* lamT_inv,
                                 void* stream) {
    if (int rc = check_graph(g, "vican_dual_update")) return rc;
    if (!rc_ || !Rt || !lamT_inv) return set_err(VICAN_ERR_ARG, "vican_dual_update: null pointer");
    if (g->n_chunk == 0) return VICAN_OK;
    if (int rc = dispatch_sweep<1>(g, nullptr, rc_, nullptr, Rt, lamT_inv, stream)) return rc;
    LAUNCH_CHECK("vican_dual_update");
    return VICAN_OK;
}

// ---------------------------------------------------------------------------
// slab reduction (fixed order => bitwise reproducible)
// ---------------------------------------------------------------------------
__global__ void slab_reduce_kernel(const double* __restrict__ part, int n_slab, long long n, double* __restrict__ out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int k = 0; k < n_slab; ++k) s += part[(size_t)k * n + i];
    out[i] = s;
}
extern "C" int vican_slab_reduce(const double* part, int32_t n_slab, int64_t n, double* out, void* stream) {
    if (!part || !out || n_slab <= 0 || n < 0) return set_err(VICAN_ERR_ARG, "vican_slab_reduce: bad argument");
    if (n == 0) return VICAN_OK;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, part,
                       n_slab, (long long)n, out);
    LAUNCH_CHECK("vican_slab_reduce");
    return VICAN_OK;
}

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
    if (int r = check_graph(g, "vican_trans_rhs")) return r;
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
    if (g->block_threads == 1024) { if (epl == 4) RHS_LAUNCH(1024, 4); else RHS_LAUNCH(1024, 2); }
    else                          { if (epl == 4) RHS_LAUNCH(256, 4);  else RHS_LAUNCH(256, 2); }
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
    if (int r = check_graph(g, "vican_cg_sweep")) return r;
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
    if (g->block_threads == 1024) { if (epl == 4) CG_LAUNCH(1024, 4); else CG_LAUNCH(1024, 2); }
    else                          { if (epl == 4) CG_LAUNCH(256, 4);  else CG_LAUNCH(256, 2); }
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
