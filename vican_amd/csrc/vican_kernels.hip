// vican_kernels.hip - camera-side and translation-stage kernels: batched 3x3 polar /
// gauge fix, dense helpers of the block Lanczos iteration, right-hand side and the
// conjugate-gradient kernels.  The hot edge sweep lives in vican_sweep.hip.
#include "vican_common.h"
#include <cstdlib>
#include <type_traits>

// ---------------------------------------------------------------------------
// batched polar / gauge
// ---------------------------------------------------------------------------
__global__ void polar_dual_kernel(const int32_t* __restrict__ gate, int n, const double* __restrict__ in, double* __restrict__ R_out,
                                  double* __restrict__ lam_out, int mode) {
    GATE_RETURN(gate);
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double A[9], R[9], lam[9];
#pragma unroll
    for (int q = 0; q < 9; ++q) A[q] = in[(size_t)i * 9 + q];
    polar_dual3_fast(A, R, lam, mode);
    if (R_out)
#pragma unroll
        for (int q = 0; q < 9; ++q) R_out[(size_t)i * 9 + q] = R[q];
    if (lam_out && (mode & 3))
#pragma unroll
        for (int q = 0; q < 9; ++q) lam_out[(size_t)i * 9 + q] = lam[q];
}
extern "C" int vican_polar_dual(int32_t n, const double* in, double* R_out, double* lam_out, int32_t mode,
                                void* stream) {
    if (n < 0 || !in || mode < 0 || mode > 6 || (mode & 3) == 3) return set_err(VICAN_ERR_ARG, "vican_polar_dual: bad argument");
    if (n == 0) return VICAN_OK;
    hipLaunchKernelGGL(polar_dual_kernel, dim3((n + 127) / 128), dim3(128), 0, (hipStream_t)stream, g_vican_gate, n, in, R_out,
                       lam_out, mode);
    LAUNCH_CHECK("vican_polar_dual");
    return VICAN_OK;
}

__global__ void gauge_project_kernel(const int32_t* __restrict__ gate, int n_cam, const double* __restrict__ xin,
                                     double* __restrict__ xout) {
    GATE_RETURN(gate);
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
    polar_dual3_fast(A, R, nullptr, 0);
#pragma unroll
    for (int q = 0; q < 9; ++q) xout[(size_t)c * 9 + q] = R[q];
}
extern "C" int vican_gauge_project(int32_t n_cam, const double* x_in, double* x_out, void* stream) {
    if (n_cam <= 0 || !x_in || !x_out) return set_err(VICAN_ERR_ARG, "vican_gauge_project: bad argument");
    // x_in may alias x_out: every thread reads block 0 before any thread of ANOTHER
    // workgroup may overwrite it only if they do not alias; require distinct buffers.
    if (x_in == x_out) return set_err(VICAN_ERR_ARG, "vican_gauge_project: in-place not supported");
    hipLaunchKernelGGL(gauge_project_kernel, dim3((n_cam + 127) / 128), dim3(128), 0, (hipStream_t)stream, g_vican_gate, n_cam, x_in,
                       x_out);
    LAUNCH_CHECK("vican_gauge_project");
    return VICAN_OK;
}

// Z[i][:] = X[i][:] * beta^-1 for the upper-triangular 3x3 beta of vican_chol_qr3 (row-major [n][3]): turns
// A (R beta^-1) = (A R) beta^-1 around, so that an operator application computed for the un-normalised start block R
// (vican_dual_update_op) serves the orthonormalised one.  A zero pivot (dropped column) gives a zero column.
__global__ void right_solve3_kernel(int n, const double* __restrict__ X, const double* __restrict__ beta, double* __restrict__ Z) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double b00 = beta[0], b01 = beta[1], b02 = beta[2], b11 = beta[4], b12 = beta[5], b22 = beta[8];
    const double x0 = X[(size_t)i * 3], x1 = X[(size_t)i * 3 + 1], x2 = X[(size_t)i * 3 + 2];
    const double z0 = b00 != 0.0 ? x0 / b00 : 0.0;
    const double z1 = b11 != 0.0 ? (x1 - z0 * b01) / b11 : 0.0;
    const double z2 = b22 != 0.0 ? (x2 - z0 * b02 - z1 * b12) / b22 : 0.0;
    Z[(size_t)i * 3] = z0; Z[(size_t)i * 3 + 1] = z1; Z[(size_t)i * 3 + 2] = z2;
}
extern "C" int vican_right_solve3(int32_t n, const double* X, const double* beta, double* Z, void* stream) {
    if (n <= 0 || !X || !beta || !Z) return set_err(VICAN_ERR_ARG, "vican_right_solve3: bad argument");
    hipLaunchKernelGGL(right_solve3_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, X, beta, Z);
    LAUNCH_CHECK("vican_right_solve3");
    return VICAN_OK;
}

// out[r] = A[r] * (sum_k B[k * b_stride + r])   (3x3 blocks, A != NULL, width 9)   or   out = sum_k B[k * b_stride + .]
// (A == NULL, any width): per-row partial results of the camera tiles (graphs with more cameras than the LDS-resident
// sweep holds, device.TiledBackend) summed in tile order and, for the operator, multiplied by the row's dual block.
__global__ void sum_apply3_kernel(const int32_t* __restrict__ gate, long long n_rows, int width, const double* __restrict__ A,
                                  const double* __restrict__ B, int n_b, long long b_stride, double* __restrict__ out) {
    GATE_RETURN(gate);
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (A) {
        if (r >= n_rows) return;
        double y[9], a[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) { y[q] = 0.0; a[q] = A[r * 9 + q]; }
        for (int k = 0; k < n_b; ++k)
#pragma unroll
            for (int q = 0; q < 9; ++q) y[q] += B[(size_t)k * b_stride + r * 9 + q];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int b = 0; b < 3; ++b) out[r * 9 + i * 3 + b] = a[i * 3] * y[b] + a[i * 3 + 1] * y[3 + b] + a[i * 3 + 2] * y[6 + b];
    } else {
        if (r >= n_rows * width) return;
        double t = 0.0;
        for (int k = 0; k < n_b; ++k) t += B[(size_t)k * b_stride + r];
        out[r] = t;
    }
}
extern "C" int vican_sum_apply3(int64_t n_rows, int32_t width, const double* A, const double* B, int32_t n_b, int64_t b_stride,
                                double* out, void* stream) {
    if (n_rows < 0 || width <= 0 || !B || !out || n_b <= 0 || (A && width != 9)) return set_err(VICAN_ERR_ARG, "vican_sum_apply3: bad argument");
    if (n_rows == 0) return VICAN_OK;
    const long long n = A ? n_rows : n_rows * width;
    hipLaunchKernelGGL(sum_apply3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, g_vican_gate,
                       (long long)n_rows, width, A, B, n_b, (long long)b_stride, out);
    LAUNCH_CHECK("vican_sum_apply3");
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
// Long vectors (the non-eliminated solver: n = 3(C+T) ~ 1e5..1e6): one workgroup per (group of GRAM_COLS basis
// columns, slice of rows) so that the whole chip streams V once and R GRAM_COLS times less often than above
// (measured at n = 303000, ka = 48: 400 us with one workgroup per column); partial sums per slice are folded in
// slice order by tall_gram_fold_kernel - deterministic.
#define GRAM_COLS 4
__global__ __launch_bounds__(256) void tall_gram_slice_kernel(int n, const double* __restrict__ V, int ld, int ka,
                                                              const double* __restrict__ R, int rows_per_slice,
                                                              double* __restrict__ part) {
    __shared__ double red[4][GRAM_COLS * 3];
    const int k0 = blockIdx.x * GRAM_COLS, s = blockIdx.y;
    const int i0 = s * rows_per_slice, i1 = min(n, i0 + rows_per_slice);
    double acc[GRAM_COLS][3];
#pragma unroll
    for (int c = 0; c < GRAM_COLS; ++c) acc[c][0] = acc[c][1] = acc[c][2] = 0.0;
    for (int i = i0 + (int)threadIdx.x; i < i1; i += 256) {
        const double r0 = R[i], r1 = R[(size_t)n + i], r2 = R[(size_t)2 * n + i];
#pragma unroll
        for (int c = 0; c < GRAM_COLS; ++c) {
            const double v = (k0 + c < ka) ? V[(size_t)(k0 + c) * ld + i] : 0.0;
            acc[c][0] += v * r0; acc[c][1] += v * r1; acc[c][2] += v * r2;
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < GRAM_COLS; ++c)
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const double t = wave_sum(acc[c][q]);
            if (lane == 0) red[wave][c * 3 + q] = t;
        }
    __syncthreads();
    const int t = threadIdx.x;
    if (t < GRAM_COLS * 3 && k0 + t / 3 < ka)
        part[((size_t)s * ka + k0) * 3 + t] = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
}
__global__ void tall_gram_fold_kernel(const double* __restrict__ part, int n_slice, int m, double* __restrict__ H) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    double s = 0.0;
    for (int k = 0; k < n_slice; ++k) s += part[(size_t)k * m + i];
    H[i] = s;
}
#define GRAM_SLICE_MIN_N 16384
#define GRAM_MAX_SLICES 128

extern "C" int vican_tall_gram(int32_t n, const double* V, int32_t ld, int32_t ka, const double* R, double* H,
                               double* ws, int64_t ws_doubles, void* stream) {
    if (n <= 0 || !V || !R || !H || ka <= 0 || ld < n) return set_err(VICAN_ERR_ARG, "vican_tall_gram: bad argument");
    if (ws && n >= GRAM_SLICE_MIN_N) {
        int rows = 4096;
        while ((n + rows - 1) / rows > GRAM_MAX_SLICES) rows *= 2;
        const int n_slice = (n + rows - 1) / rows;
        if ((int64_t)n_slice * ka * 3 <= ws_doubles) {
            hipLaunchKernelGGL(tall_gram_slice_kernel, dim3((ka + GRAM_COLS - 1) / GRAM_COLS, n_slice), dim3(256), 0,
                               (hipStream_t)stream, n, V, ld, ka, R, rows, ws);
            hipLaunchKernelGGL(tall_gram_fold_kernel, dim3((ka * 3 + 63) / 64), dim3(64), 0, (hipStream_t)stream, ws, n_slice,
                               ka * 3, H);
            LAUNCH_CHECK("vican_tall_gram");
            return VICAN_OK;
        }
    }
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

// Start block of an eigen-solve in ONE launch (n <= VICAN_SEED_MAX_N): G = X0^T X0, upper Cholesky G = beta^T beta,
// Q0 = X0 beta^-1 -> basis columns 0..2 and the row-major sweep input, beta0; optionally also Z = Zraw beta^-1
// (vican_right_solve3).  Replaces vican_rows_to_cols + vican_tall_gram + vican_chol_qr3 (+ vican_right_solve3): four
// dependent launches of 5-8 us each at the head of every primal-dual iteration.  One workgroup; the six Gram entries
// are reduced in a fixed order.
#define VICAN_SEED_THREADS 1024
__global__ __launch_bounds__(VICAN_SEED_THREADS) void lanczos_seed_kernel(int n, const double* __restrict__ X0, double* __restrict__ V,
                                                                            int ld, double* __restrict__ beta_out,
                                                                            double* __restrict__ x_out, const double* __restrict__ Zraw,
                                                                            double* __restrict__ Z, unsigned int* __restrict__ rearm) {
    __shared__ double red[16][6];
    __shared__ double sb[9];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (rearm && tid < 2) rearm[tid] = 0u;      // barrier counters of the cooperative step (first use: the next launch)
    double g[6] = {0, 0, 0, 0, 0, 0};
    for (int i = tid; i < n; i += VICAN_SEED_THREADS) {
        const double a = X0[(size_t)i * 3], b = X0[(size_t)i * 3 + 1], c = X0[(size_t)i * 3 + 2];
        g[0] += a * a; g[1] += a * b; g[2] += a * c; g[3] += b * b; g[4] += b * c; g[5] += c * c;
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const double t = wave_sum(g[k]);
        if (lane == 0) red[wave][k] = t;
    }
    __syncthreads();
    if (tid == 0) {
        double G6[6];
        for (int k = 0; k < 6; ++k) { double t = 0.0; for (int w = 0; w < VICAN_SEED_THREADS / 64; ++w) t += red[w][k]; G6[k] = t; }
        const double g00 = G6[0], g01 = G6[1], g02 = G6[2], g11 = G6[3], g12 = G6[4], g22 = G6[5];
        const double floor_ = 1e-28 * (g00 + g11 + g22);                // same pivot rule as chol_qr3_kernel
        double b00 = 0, b01 = 0, b02 = 0, b11 = 0, b12 = 0, b22 = 0;
        if (g00 > floor_) { b00 = sqrt(g00); b01 = g01 / b00; b02 = g02 / b00; }
        const double d11 = g11 - b01 * b01;
        if (d11 > floor_) { b11 = sqrt(d11); b12 = (g12 - b01 * b02) / b11; }
        const double d22 = g22 - b02 * b02 - b12 * b12;
        if (d22 > floor_) b22 = sqrt(d22);
        sb[0] = b00; sb[1] = b01; sb[2] = b02; sb[3] = 0; sb[4] = b11; sb[5] = b12; sb[6] = 0; sb[7] = 0; sb[8] = b22;
        for (int k = 0; k < 9; ++k) beta_out[k] = sb[k];
    }
    __syncthreads();
    const double b00 = sb[0], b01 = sb[1], b02 = sb[2], b11 = sb[4], b12 = sb[5], b22 = sb[8];
    const double i00 = b00 != 0.0 ? 1.0 / b00 : 0.0, i11 = b11 != 0.0 ? 1.0 / b11 : 0.0, i22 = b22 != 0.0 ? 1.0 / b22 : 0.0;
    for (int i = tid; i < n; i += VICAN_SEED_THREADS) {
        const double r0 = X0[(size_t)i * 3], r1 = X0[(size_t)i * 3 + 1], r2 = X0[(size_t)i * 3 + 2];
        const double q0 = r0 * i00, q1 = (r1 - q0 * b01) * i11, q2 = (r2 - q0 * b02 - q1 * b12) * i22;
        V[i] = q0; V[(size_t)ld + i] = q1; V[(size_t)2 * ld + i] = q2;
        x_out[(size_t)i * 3] = q0; x_out[(size_t)i * 3 + 1] = q1; x_out[(size_t)i * 3 + 2] = q2;
        if (Zraw) {
            const double x0 = Zraw[(size_t)i * 3], x1 = Zraw[(size_t)i * 3 + 1], x2 = Zraw[(size_t)i * 3 + 2];
            const double z0 = x0 * i00, z1 = (x1 - z0 * b01) * i11, z2 = (x2 - z0 * b02 - z1 * b12) * i22;
            Z[(size_t)i * 3] = z0; Z[(size_t)i * 3 + 1] = z1; Z[(size_t)i * 3 + 2] = z2;
        }
    }
}
extern "C" int vican_lanczos_seed(int32_t n, const double* X0, double* V, int32_t ld, double* beta_out, double* x_out,
                                  const double* Zraw, double* Z, void* coop_sync, void* stream) {
    if (n <= 0 || n > VICAN_SEED_MAX_N || !X0 || !V || !beta_out || !x_out || ld < n || (Zraw && !Z) || X0 == x_out)
        return set_err(VICAN_ERR_ARG, "vican_lanczos_seed: bad argument");
    hipLaunchKernelGGL(lanczos_seed_kernel, dim3(1), dim3(VICAN_SEED_THREADS), 0, (hipStream_t)stream, n, X0, V, ld, beta_out,
                       x_out, Zraw, Z, (unsigned int*)coop_sync);
    LAUNCH_CHECK("vican_lanczos_seed");
    return VICAN_OK;
}

// X[n][3] (row-major) = V[:, :ka] Y[ka][3]
__global__ __launch_bounds__(256) void tall_combine_kernel(const int32_t* __restrict__ gate, int n, const double* __restrict__ V, int ld, int ka,
                                                           const double* __restrict__ Y, double* __restrict__ X) {
    GATE_RETURN(gate);
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
    hipLaunchKernelGGL(tall_combine_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, g_vican_gate, n, V, ld, ka, Y, X);
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
// The camera-side half of one block-Lanczos step in ONE single-workgroup launch.
// The seven fine-grained kernels above (kept as entry points) are each 5-7 us of mostly launch
// latency on a 3C x 3(j+1) problem; fused, the whole step is one launch of 1024 threads with
// workgroup barriers between the stages.  Same arithmetic and the same (fixed) summation orders
// per stage, hence reproducible; z = reduced P Q_j; writes basis block j+1, x_out (next sweep
// input), Hcol (this step's projected column, 3(j+1) x 3) and beta (3 x 3).
// ---------------------------------------------------------------------------
__device__ __forceinline__ void fused_gram(int n, const double* __restrict__ V, int ld, int ka,
                                           const double* __restrict__ R, double* __restrict__ h /* LDS [ka][3] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int k = wave; k < ka; k += nw) {
        const double* v = V + (size_t)k * ld;
        double s0 = 0, s1 = 0, s2 = 0;
        for (int i = lane; i < n; i += 64) {
            const double vi = v[i];
            s0 += vi * R[i]; s1 += vi * R[(size_t)n + i]; s2 += vi * R[(size_t)2 * n + i];
        }
        s0 = wave_sum(s0); s1 = wave_sum(s1); s2 = wave_sum(s2);
        if (lane == 0) { h[k * 3] = s0; h[k * 3 + 1] = s1; h[k * 3 + 2] = s2; }
    }
}
__device__ __forceinline__ void fused_update(int n, const double* __restrict__ V, int ld, int ka,
                                             const double* __restrict__ h, double* __restrict__ R) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        double r0 = R[i], r1 = R[(size_t)n + i], r2 = R[(size_t)2 * n + i];
        for (int k = 0; k < ka; ++k) {
            const double v = V[(size_t)k * ld + i];
            r0 -= v * h[k * 3]; r1 -= v * h[k * 3 + 1]; r2 -= v * h[k * 3 + 2];
        }
        R[i] = r0; R[(size_t)n + i] = r1; R[(size_t)2 * n + i] = r2;
    }
}
#define VICAN_FUSED_CAM_MAX 400
__global__ __launch_bounds__(1024) void lanczos_cam_fused_kernel(int n_cam, const double* __restrict__ lamC, double* V,
                                                                 int ld, int j, const double* __restrict__ z, double* R,
                                                                 double* __restrict__ Hcol, double* __restrict__ beta_out,
                                                                 double* __restrict__ x_out, double pivot_floor) {
    __shared__ double h[KA_MAX * 3];
    __shared__ double h2[KA_MAX * 3];
    __shared__ double gpart[16][6];
    const int n = 3 * n_cam, ka = 3 * (j + 1), tid = threadIdx.x;
    // A Q_j = Lambda_C Q_j - z    (R column-major [3][n])
    for (int c = tid; c < n_cam; c += blockDim.x) {
        double L[9], q[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) L[k] = lamC[(size_t)c * 9 + k];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int b = 0; b < 3; ++b) q[i * 3 + b] = V[(size_t)(3 * j + b) * ld + 3 * c + i];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int b = 0; b < 3; ++b)
                R[(size_t)b * n + 3 * c + i] = L[i * 3] * q[b] + L[i * 3 + 1] * q[3 + b] + L[i * 3 + 2] * q[6 + b] -
                                              z[(size_t)(3 * c + i) * 3 + b];
    }
    __syncthreads();
    fused_gram(n, V, ld, ka, R, h);                 // Gram-Schmidt pass 1
    __syncthreads();
    fused_update(n, V, ld, ka, h, R);
    __syncthreads();
    fused_gram(n, V, ld, ka, R, h2);                // pass 2 ("twice is enough")
    __syncthreads();
    fused_update(n, V, ld, ka, h2, R);
    for (int i = tid; i < ka * 3; i += blockDim.x) Hcol[i] = h[i] + h2[i];
    __syncthreads();
    // G = R^T R  (6 unique entries), fixed order: per-thread -> wave -> 16 waves
    double g[6] = {0, 0, 0, 0, 0, 0};
    for (int i = tid; i < n; i += blockDim.x) {
        const double r0 = R[i], r1 = R[(size_t)n + i], r2 = R[(size_t)2 * n + i];
        g[0] += r0 * r0; g[1] += r0 * r1; g[2] += r0 * r2; g[3] += r1 * r1; g[4] += r1 * r2; g[5] += r2 * r2;
    }
#pragma unroll
    for (int q = 0; q < 6; ++q) g[q] = wave_sum(g[q]);
    if ((tid & 63) == 0)
#pragma unroll
        for (int q = 0; q < 6; ++q) gpart[tid >> 6][q] = g[q];
    __syncthreads();
    double G6[6] = {0, 0, 0, 0, 0, 0};
    for (int wv = 0; wv < (int)(blockDim.x >> 6); ++wv)
#pragma unroll
        for (int q = 0; q < 6; ++q) G6[q] += gpart[wv][q];
    // upper Cholesky G = beta^T beta, Q = R beta^-1  (same pivot rule as chol_qr3_kernel)
    const double g00 = G6[0], g01 = G6[1], g02 = G6[2], g11 = G6[3], g12 = G6[4], g22 = G6[5];
    const double tr = g00 + g11 + g22, floor_ = fmax(1e-28 * tr, pivot_floor);
    double b00 = 0, b01 = 0, b02 = 0, b11 = 0, b12 = 0, b22 = 0, i00 = 0, i11 = 0, i22 = 0;
    if (g00 > floor_) { b00 = sqrt(g00); i00 = 1.0 / b00; b01 = g01 * i00; b02 = g02 * i00; }
    const double d11 = g11 - b01 * b01;
    if (d11 > floor_) { b11 = sqrt(d11); i11 = 1.0 / b11; b12 = (g12 - b01 * b02) * i11; }
    const double d22 = g22 - b02 * b02 - b12 * b12;
    if (d22 > floor_) { b22 = sqrt(d22); i22 = 1.0 / b22; }
    if (tid == 0) {
        beta_out[0] = b00; beta_out[1] = b01; beta_out[2] = b02; beta_out[3] = 0; beta_out[4] = b11; beta_out[5] = b12;
        beta_out[6] = 0; beta_out[7] = 0; beta_out[8] = b22;
    }
    const int col0 = 3 * (j + 1);
    for (int i = tid; i < n; i += blockDim.x) {
        const double r0 = R[i], r1 = R[(size_t)n + i], r2 = R[(size_t)2 * n + i];
        const double q0 = r0 * i00;
        const double q1 = (r1 - q0 * b01) * i11;
        const double q2 = (r2 - q0 * b02 - q1 * b12) * i22;
        V[(size_t)col0 * ld + i] = q0; V[(size_t)(col0 + 1) * ld + i] = q1; V[(size_t)(col0 + 2) * ld + i] = q2;
        x_out[(size_t)i * 3] = q0; x_out[(size_t)i * 3 + 1] = q1; x_out[(size_t)i * 3 + 2] = q2;
    }
}

// ---------------------------------------------------------------------------
// Cooperative camera-side Lanczos step: the seven launches of the fine-grained sequence (each a few
// microseconds of work behind ~5 us of launch latency) as ONE kernel of <= 32 co-resident workgroups that
// meet at two grid barriers (one per Gram-Schmidt pass; the 3x3 Gram of the final block rides on the second - device counter + spin;
// the grid is far smaller than the chip, so all
// workgroups are resident).  Each workgroup owns a slice of <= 32 cameras (96 rows of the 3C x 3 block):
// its rows of R live in LDS for the whole step; only the Gram partials (3 ka doubles per workgroup and pass)
// and the 3x3 Gram of R cross workgroups, summed in a fixed order => deterministic.
// ---------------------------------------------------------------------------
#define COOP_CAMS 32
#define COOP_MAX_WG 256                 /* workgroups of the cooperative camera-side step: 8192 cameras */
#define COOP_ROWS (3 * COOP_CAMS)
// Everything that crosses workgroups (the partial sums) is written and read with device-scope atomics, which
// go through to the memory-side coherence point by themselves; a device-scope FENCE would instead write back the
// whole L2 of the XCD - which still holds the megabytes of slabs of the sweep that ran just before (measured:
// ~10 us per fence, 33 us for an otherwise empty step).  __syncthreads() waits for the workgroup's own stores.
__device__ __forceinline__ void st_agent(double* p, double v) {
    __hip_atomic_store((unsigned long long*)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double ld_agent(const double* p) {
    return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
// Grid barrier of the cooperative step.  Memory ordering, deliberately NOT release/acquire: every datum that crosses
// workgroups here is written with st_agent (an agent-scope atomic store: write-through to the device coherence point)
// and read with ld_agent (an agent-scope atomic load: never served from this CU's L1), the __syncthreads() in front of
// the arrival makes every wavefront wait for its outstanding stores (s_waitcnt vmcnt(0)) before thread 0 adds to the
// counter, and the counter itself is an agent-scope atomic - so the data is at the coherence point before any other
// workgroup can see the count.  A release on the arrival / acquire on the exit would be the portable form under the HIP
// memory model, but at agent scope they lower to an L2 write-back / invalidate of everything this XCD holds - the sweep's
// 18 MB of slabs included, ~10 us per barrier.  The reliance on this ISA behaviour is pinned by
// tests/test_kernels_gpu.py::test_cooperative_step_repeats_bit_identically (3000 repetitions, bit-identical), and the
// counter is zeroed at the start of every eigen-solve (by the seed kernel, vican_lanczos_seed(coop_sync)).
// Graphs with few slabs (<= 64 workgroups in the sweep, i.e. little dirty data in L2) take the fenced form of the barrier
// anyway (`fenced`: +0.4-0.8 us per barrier there); the spin itself is bounded (vican_grid_sync, vican_common.h).
// partial H = V[:, :ka]^T R over this workgroup's rows, to part[ka*3]; vs: this workgroup's rows of the basis, staged in
// LDS as [ka][COOP_ROWS].  8 lanes per element: strided rows, then a DPP sum over the 8 lanes.  (The first version gave a
// wavefront to each basis vector and reduced three sums over 64 lanes with shuffles - 36 ds_bpermute per vector, ~1.2 us,
// i.e. 11 us per Gram-Schmidt pass at 36 vectors: the camera-side step grew from 12 to 30 us along an eigen-solve.)
__device__ __forceinline__ double coop_dpp8_sum(double v) {
    auto mv = [](double x, auto ctrl) -> double {
        const unsigned long long b = (unsigned long long)__double_as_longlong(x);
        const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, decltype(ctrl)::value, 0xF, 0xF, true);
        const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), decltype(ctrl)::value, 0xF, 0xF, true);
        return __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
    };
    v += mv(v, std::integral_constant<int, 0xB1>());       // lane ^ 1
    v += mv(v, std::integral_constant<int, 0x4E>());       // lane ^ 2
    v += mv(v, std::integral_constant<int, 0x141>());      // mirror inside 8 lanes
    return v;
}
// sum over the 64 lanes, in every lane: DPP inside groups of 16, two cross-row shuffles
__device__ __forceinline__ double wave_allsum_dpp(double v) {
    v = coop_dpp8_sum(v);
    {
        const unsigned long long b = (unsigned long long)__double_as_longlong(v);
        const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, 0x140, 0xF, 0xF, true);          // row_mirror
        const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), 0x140, 0xF, 0xF, true);
        v += __longlong_as_double((long long)(((unsigned long long)(unsigned)hi << 32) | (unsigned)lo));
    }
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    return v;
}
__device__ __forceinline__ void coop_gram(const double* __restrict__ vs, int ka, int nsl,
                                          const double (*rs)[COOP_ROWS], double* __restrict__ part, int nwg, int wg) {
    const int seg = threadIdx.x & 7;
    for (int e = threadIdx.x >> 3; e < ka * 3; e += (int)(blockDim.x >> 3)) {
        const int k = e / 3, c = e - 3 * k;
        const double* v = vs + k * COOP_ROWS;
        double s = 0.0;
        for (int i = seg; i < nsl; i += 8) s += v[i] * rs[c][i];
        s = coop_dpp8_sum(s);
        if (seg == 0) st_agent(part + (size_t)e * nwg + wg, s);       // layout [3 ka][nwg]: one element's partials are contiguous
    }
}
// out[t] = sum_w part[t][w] (fixed order).  The device-scope loads cost ~150 ns each when one thread issues them
// back to back, so all threads fetch a window of COOP_STAGE values at once (coalesced) and the sums run from LDS.
#define COOP_STAGE 2048
__device__ __forceinline__ void coop_reduce(const double* __restrict__ part, int hs, int nwg, double* __restrict__ stage,
                                            double* __restrict__ out) {
    const int per = COOP_STAGE / nwg;                      // elements per window
    for (int t0 = 0; t0 < hs; t0 += per) {
        const int nt = hs - t0 < per ? hs - t0 : per, cnt = nt * nwg;
        double v[COOP_STAGE / 256];
#pragma unroll
        for (int m = 0; m < COOP_STAGE / 256; ++m) {
            const int e = threadIdx.x + m * 256;
            v[m] = e < cnt ? ld_agent(part + (size_t)t0 * nwg + e) : 0.0;
        }
#pragma unroll
        for (int m = 0; m < COOP_STAGE / 256; ++m) stage[threadIdx.x + m * 256] = v[m];
        __syncthreads();
        for (int t = threadIdx.x; t < nt; t += blockDim.x) {
            double a = 0.0;
            for (int w = 0; w < nwg; ++w) a += stage[t * nwg + w];
            out[t0 + t] = a;
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void lanczos_cam_coop_kernel(int n_cam, const double* __restrict__ lamC, double* V, int ld,
                                                               int j, const double* __restrict__ z, double* ws,
                                                               double* __restrict__ Hcol, double* __restrict__ beta_out,
                                                               double* __restrict__ x_out, double pivot_floor,
                                                               unsigned int* sync, const long long* __restrict__ zpart,
                                                               int n_slab, const double* __restrict__ pa,
                                                               const double* __restrict__ pb, uint32_t* abort_word,
                                                               unsigned long long spin_limit, int fenced) {
    extern __shared__ double vs[];
    const vican_sync_t sy = {sync, abort_word, spin_limit};                       // [ka][COOP_ROWS]: this workgroup's rows of the basis,
    __shared__ double rs[3][COOP_ROWS];                  // read from global memory ONCE for all four uses
    __shared__ double h[KA_MAX * 3], h2[KA_MAX * 3];
    __shared__ double g6[4][6], G6s[6];
    __shared__ double stage[COOP_STAGE];
    __shared__ long long zred[8][9 * COOP_CAMS];         // slab fold (zpart != NULL): partial sums of 8 slab groups
    __shared__ double zl[9 * COOP_CAMS];                 // ... and this workgroup's rows of z
    const int nwg = (int)gridDim.x, wg = (int)blockIdx.x, tid = threadIdx.x;
    const int ka = 3 * (j + 1), hs = 3 * ka;
    const int c0 = (int)(((long long)wg * n_cam) / nwg), c1 = (int)(((long long)(wg + 1) * n_cam) / nwg);
    const int row0 = 3 * c0, nsl = 3 * (c1 - c0);
    double* part1 = ws;                                  // [nwg][hs]
    double* part2 = ws + (size_t)nwg * hs;               // [nwg][hs]
    double* partG = ws + (size_t)2 * nwg * hs;           // [nwg][8]
    // z straight from the sweep's fixed-point slabs [n_slab][9][n_cam] (what vican_slab_reduce_fx would produce - one
    // launch less per Lanczos step): this workgroup's cameras only, exact integer sums, same conversion.  The slab loads
    // are issued first and five slabs at a time (45 independent loads per thread: the fold was five dependent round
    // trips, 3 of the step's 15 us), the basis rows are staged while they are in flight.
    // (every global load that does not depend on another one is issued up front: the step is a chain of round trips)
    double Lc[9], qc9[9];
    const double sc = 1.0 * (pa ? *pa : 1.0) * (pb ? *pb : 1.0);
    {
        const int c = tid < c1 - c0 ? c0 + tid : c0;
#pragma unroll
        for (int k = 0; k < 9; ++k) Lc[k] = lamC[(size_t)c * 9 + k];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int b = 0; b < 3; ++b) qc9[i * 3 + b] = V[(size_t)(3 * j + b) * ld + 3 * c + i];
    }
    long long zacc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (zpart) {
        const int lane = tid & 31, grp = tid >> 5, ncl = c1 - c0;
        if (lane < ncl) {
#pragma unroll 5
            for (int s = grp; s < n_slab; s += 8) {
                const long long* sp = zpart + (size_t)s * 9 * n_cam + c0 + lane;
#pragma unroll
                for (int q = 0; q < 9; ++q) zacc[q] += sp[(size_t)q * n_cam];
            }
        }
    }
    for (int t = tid; t < ka * COOP_ROWS; t += blockDim.x) {
        const int k = t / COOP_ROWS, i = t - k * COOP_ROWS;
        vs[t] = i < nsl ? V[(size_t)k * ld + row0 + i] : 0.0;
    }
    if (zpart) {
        const int lane = tid & 31, grp = tid >> 5;
#pragma unroll
        for (int q = 0; q < 9; ++q) zred[grp][q * COOP_CAMS + lane] = zacc[q];
        __syncthreads();
        for (int t = tid; t < 9 * COOP_CAMS; t += blockDim.x) {
            long long sum = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k) sum += zred[k][t];
            const int q = t / COOP_CAMS, cl = t - q * COOP_CAMS;
            zl[cl * 9 + q] = (double)sum * sc;
        }
        __syncthreads();
    }
    // A Q_j = Lambda_C Q_j - z on this slice
    if (tid < c1 - c0) {
        const int c = c0 + tid;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int b = 0; b < 3; ++b)
                rs[b][3 * tid + i] = Lc[i * 3] * qc9[b] + Lc[i * 3 + 1] * qc9[3 + b] + Lc[i * 3 + 2] * qc9[6 + b] -
                                     (zpart ? zl[tid * 9 + i * 3 + b] : z[(size_t)(3 * c + i) * 3 + b]);
    }
    __syncthreads();
    // two Gram-Schmidt passes against the whole basis.  G = R^T R of the FINAL block rides on the second pass's barrier: with an
    // orthonormal basis (R' - V h2)^T (R' - V h2) = R'^T R' - h2^T h2, and h2 - what the first pass left along the basis - is
    // rounding-sized against R' (no cancellation: the correction is ~1e-32 of the leading term; where the Krylov space is exhausted
    // both forms sit far below the pivot floor).  Two grid barriers per step instead of three.
    for (int pass = 0; pass < 2; ++pass) {
        double* part = pass ? part2 : part1;
        double* hh = pass ? h2 : h;
        coop_gram(vs, ka, nsl, rs, part, nwg, wg);
        if (pass == 1) {                                       // slice partial of R'^T R' (six numbers) -> all workgroups
            double g[6] = {0, 0, 0, 0, 0, 0};
            if (tid < nsl) {
                const double r0 = rs[0][tid], r1 = rs[1][tid], r2 = rs[2][tid];
                g[0] = r0 * r0; g[1] = r0 * r1; g[2] = r0 * r2; g[3] = r1 * r1; g[4] = r1 * r2; g[5] = r2 * r2;
            }
#pragma unroll
            for (int q = 0; q < 6; ++q) g[q] = wave_sum(g[q]);
            if ((tid & 63) == 0)
#pragma unroll
                for (int q = 0; q < 6; ++q) g6[tid >> 6][q] = g[q];
            __syncthreads();
            if (tid < 6) st_agent(partG + (size_t)tid * nwg + wg, (g6[0][tid] + g6[1][tid]) + (g6[2][tid] + g6[3][tid]));
        }
        if (!vican_grid_sync(sy, (unsigned)((pass + 1) * nwg), fenced != 0)) return;
        coop_reduce(part, hs, nwg, stage, hh);
        if (pass == 1) coop_reduce(partG, 6, nwg, stage, G6s);
        if (tid < nsl) {
            double r0 = rs[0][tid], r1 = rs[1][tid], r2 = rs[2][tid];
            for (int k = 0; k < ka; ++k) {
                const double v = vs[k * COOP_ROWS + tid];
                r0 -= v * hh[k * 3]; r1 -= v * hh[k * 3 + 1]; r2 -= v * hh[k * 3 + 2];
            }
            rs[0][tid] = r0; rs[1][tid] = r1; rs[2][tid] = r2;
        }
        __syncthreads();
    }
    if (wg == 0) for (int t = tid; t < hs; t += blockDim.x) Hcol[t] = h[t] + h2[t];
    // G = R'^T R' - h2^T h2 (every workgroup for itself, from the same numbers in the same order)
    double c00 = 0, c01 = 0, c02 = 0, c11 = 0, c12 = 0, c22 = 0;
    for (int k = 0; k < ka; ++k) {
        const double a = h2[k * 3], b = h2[k * 3 + 1], c = h2[k * 3 + 2];
        c00 += a * a; c01 += a * b; c02 += a * c; c11 += b * b; c12 += b * c; c22 += c * c;
    }
    // upper Cholesky G = beta^T beta, Q = R beta^-1  (same pivot rule as chol_qr3_kernel)
    const double g00 = G6s[0] - c00, g01 = G6s[1] - c01, g02 = G6s[2] - c02, g11 = G6s[3] - c11, g12 = G6s[4] - c12, g22 = G6s[5] - c22;
    const double tr = g00 + g11 + g22, floor_ = fmax(1e-28 * tr, pivot_floor);
    double b00 = 0, b01 = 0, b02 = 0, b11 = 0, b12 = 0, b22 = 0, i00 = 0, i11 = 0, i22 = 0;
    if (g00 > floor_) { b00 = sqrt(g00); i00 = 1.0 / b00; b01 = g01 * i00; b02 = g02 * i00; }
    const double d11 = g11 - b01 * b01;
    if (d11 > floor_) { b11 = sqrt(d11); i11 = 1.0 / b11; b12 = (g12 - b01 * b02) * i11; }
    const double d22 = g22 - b02 * b02 - b12 * b12;
    if (d22 > floor_) { b22 = sqrt(d22); i22 = 1.0 / b22; }
    if (wg == 0 && tid == 0) {
        beta_out[0] = b00; beta_out[1] = b01; beta_out[2] = b02; beta_out[3] = 0; beta_out[4] = b11; beta_out[5] = b12;
        beta_out[6] = 0; beta_out[7] = 0; beta_out[8] = b22;
    }
    if (tid < nsl) {
        const int col0 = 3 * (j + 1), i = row0 + tid;
        const double q0 = rs[0][tid] * i00;
        const double q1 = (rs[1][tid] - q0 * b01) * i11;
        const double q2 = (rs[2][tid] - q0 * b02 - q1 * b12) * i22;
        V[(size_t)col0 * ld + i] = q0; V[(size_t)(col0 + 1) * ld + i] = q1; V[(size_t)(col0 + 2) * ld + i] = q2;
        x_out[(size_t)i * 3] = q0; x_out[(size_t)i * 3 + 1] = q1; x_out[(size_t)i * 3 + 2] = q2;
    }
    // the last workgroup out re-arms the barrier counter for the next launch
    if (tid == 0 && __hip_atomic_fetch_add(&sync[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)nwg - 1u) {
        __hip_atomic_store(&sync[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&sync[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
extern "C" int64_t vican_lanczos_coop_ws_doubles(int32_t n_cam) {
    const int64_t nwg = (n_cam + COOP_CAMS - 1) / COOP_CAMS;
    return nwg * (2LL * 3 * KA_MAX + 8);
}
extern "C" int vican_lanczos_cam_coop(int32_t n_cam, const double* lamC, double* V, int32_t ld, int32_t j, const double* z,
                                      double* ws, double* Hcol, double* beta, double* x_out, double pivot_floor,
                                      uint32_t* sync_ws, const void* zpart, int32_t n_slab, const double* pa, const double* pb,
                                      int32_t fenced, void* stream) {
    if (n_cam <= 0 || n_cam > COOP_MAX_WG * COOP_CAMS || !lamC || !V || (!z && !zpart) || (zpart && n_slab <= 0) || !ws || !Hcol || !beta ||
        !x_out || !sync_ws || j < 0 ||
        3 * (j + 1) > 128 || ld < 3 * n_cam)                 // basis slice in LDS: 128 x 96 doubles = 96 KB
        return set_err(VICAN_ERR_ARG, "vican_lanczos_cam_coop: bad argument");
    const int nwg = (n_cam + COOP_CAMS - 1) / COOP_CAMS;                    // <= 256 workgroups (camera-tiled graphs: 4000 cameras = 125)
    const size_t lds = (size_t)3 * (j + 1) * COOP_ROWS * sizeof(double);
    static size_t configured = 0;
    if (lds > 32 * 1024 && lds > configured) {
        if (hipFuncSetAttribute((const void*)lanczos_cam_coop_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * COOP_ROWS * 8) != hipSuccess)
            return set_err(VICAN_ERR_LAUNCH, "vican_lanczos_cam_coop: cannot raise dynamic LDS limit");
        configured = 128 * COOP_ROWS * 8;
    }
    // co-residency of the grid on an idle device (cached per kernel, LDS size, grid and device inside vican_coresident_ok)
    if (int rc = vican_coresident_ok((const void*)lanczos_cam_coop_kernel, 256, lds, nwg, "vican_lanczos_cam_coop")) return rc;
    hipLaunchKernelGGL(lanczos_cam_coop_kernel, dim3(nwg), dim3(256), lds, (hipStream_t)stream, n_cam, lamC, V, ld, j, z, ws, Hcol,
                       beta, x_out, pivot_floor, sync_ws, (const long long*)zpart, n_slab, pa, pb, g_vican_abort_word,
                       g_vican_sync_ticks, (int)fenced);
    LAUNCH_CHECK("vican_lanczos_cam_coop");
    return VICAN_OK;
}

// One Lanczos step of a single rank behind ONE host call: the operator sweep (vican_block_op: slabs) and the cooperative camera-side
// step that folds them itself.  Capture-sized graphs are bound by the host's launch rate once their kernels take 10-15 us each
// (large_shop: 97 launches in 1.19 ms of GPU time); a ctypes call costs the Python driver ~5 us.  VICAN_ERR_CAPACITY: the sweep
// HAS run (its slabs are in zpart), the cooperative grid was refused - fold the slabs and take vican_lanczos_cam_step.
extern "C" int vican_lanczos_step_slabs(const vican_graph_t* g, const double* lamT_inv, const double* x, void* zpart, double* fx,
                                        const double* lamC, double* V, int32_t ld, int32_t j, double* ws, double* Hcol, double* beta,
                                        double* x_out, double pivot_floor, uint32_t* sync_ws, int32_t fenced, void* stream) {
    if (!g || !fx) return set_err(VICAN_ERR_ARG, "vican_lanczos_step_slabs: bad argument");
    if (int rc = vican_block_op(g, lamT_inv, x, zpart, fx, stream)) return rc;
    return vican_lanczos_cam_coop(g->n_cam, lamC, V, ld, j, nullptr, ws, Hcol, beta, x_out, pivot_floor, sync_ws, zpart, g->n_wg, fx + 3, fx + 7,
                                  fenced, stream);
}

extern "C" int vican_lanczos_cam_step(int32_t n_cam, const double* lamC, double* V, int32_t ld, int32_t j,
                                      const double* z, double* R, double* H, double* G, double* Hcol, double* beta,
                                      double* x_out, double pivot_floor, double* ws, int64_t ws_doubles, void* stream) {
    if (n_cam <= 0 || !lamC || !V || !z || !R || !H || !G || !Hcol || !beta || !x_out || j < 0 || 3 * (j + 1) > KA_MAX ||
        ld < 3 * n_cam)
        return set_err(VICAN_ERR_ARG, "vican_lanczos_cam_step: bad argument");
    if (n_cam > VICAN_FUSED_CAM_MAX) {
        // large camera sets: the stages are bandwidth-bound enough to want many workgroups
        const int n = 3 * n_cam, ka = 3 * (j + 1);
        int rc;
        if ((rc = vican_lap_apply(n_cam, lamC, V, ld, 3 * j, z, R, stream)) < 0) return rc;
        if ((rc = vican_tall_gram(n, V, ld, ka, R, H, ws, ws_doubles, stream)) < 0) return rc;
        if ((rc = vican_tall_update(n, V, ld, ka, H, R, Hcol, 0, stream)) < 0) return rc;
        if ((rc = vican_tall_gram(n, V, ld, ka, R, H, ws, ws_doubles, stream)) < 0) return rc;          // second Gram-Schmidt pass
        if ((rc = vican_tall_update(n, V, ld, ka, H, R, Hcol, 1, stream)) < 0) return rc;
        if ((rc = vican_tall_gram(n, R, n, 3, R, G, ws, ws_doubles, stream)) < 0) return rc;
        return vican_chol_qr3(n, R, G, V, ld, 3 * (j + 1), beta, x_out, pivot_floor, stream);
    }
    hipLaunchKernelGGL(lanczos_cam_fused_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, n_cam, lamC, V, ld, j, z, R,
                       Hcol, beta, x_out, pivot_floor);
    LAUNCH_CHECK("vican_lanczos_cam_step");
    return VICAN_OK;
}

// ---------------------------------------------------------------------------
// Ritz step of the block Lanczos iteration, entirely on the device
// ---------------------------------------------------------------------------
// Assembles the projected matrix T = V^T L V (3 eff x 3 eff, eff <= VICAN_RITZ_MAX_STEPS) from the
// columns recorded by the Lanczos steps, diagonalises it with a parallel two-sided Jacobi iteration
// in LDS, forms the Ritz residuals of the three smallest pairs and takes the stop / converged
// decision that the host used to take after a synchronous round trip (reference: the ARPACK call at
// bipgo.py:288; tolerance semantics as in solver.RotationSolver).  `gate` receives 1 iff the
// eigen-solve is finished AND converged: work enqueued behind this kernel under
// vican_set_gate(gate) then runs, otherwise it cancels itself.
//
// Jacobi layout (Brent-Luk): round-robin pairing, N/2 disjoint pivots (p_k, q_k) per round, N-1
// rounds per sweep.  Thread k computes the rotation of pair k from its three pivot entries; after a
// barrier one thread per 2x2 block (pair k_i x pair k_j) applies B <- J_i^T B J_j in place - a block
// is read and written by its owner only - and rotates the same 2x2 block of the eigenvector matrix
// by J_j.  A round is bound by instruction issue and LDS latency of a few wavefronts (measured
// ~0.45-0.9 us per round), so index arithmetic is recomputed instead of loaded and every LDS read of
// a phase is independent of the others (one round trip per phase).
// The rotation angle comes from a fast f32 evaluation (it only has to annihilate the pivot to a
// relative 1e-7 per visit - the iteration converges quadratically anyway), but (c, s) are formed in
// f64 from the half-angle tangent u as ((1-u^2), 2u) / (1+u^2), which is orthogonal to rounding for ANY
// u, so the eigenvector matrix stays orthonormal.  Pivots below 2e-15 |T|_F (above the rounding noise
// of the update, LAPACK-grade absolute accuracy) are skipped.
__device__ __forceinline__ void ritz_pair(int k, int r, int N, int& p, int& q) {
    int a = r + k, b = r - k + (N - 1);
    if (a >= N - 1) a -= N - 1;
    if (b >= N - 1) b -= N - 1;
    if (k == 0) { a = N - 1; b = r; }
    p = a < b ? a : b; q = a < b ? b : a;
}

// Jacobi rotation annihilating the pivot a_pq (to ~1e-7 relative): c, s exactly orthogonal.  Returns 1 if it rotates.
__device__ __forceinline__ int ritz_rotation(double app, double aqq, double apq, bool live, double thr, double inv,
                                             double& c, double& s) {
    c = 1.0; s = 0.0;
    if (!live || !(fabs(apq) > thr)) return 0;
    const float df = (float)((aqq - app) * inv), hf = (float)(2.0 * apq * inv);
    const float tau = df * __builtin_amdgcn_rcpf(hf);
    const float at = fabsf(tau);
    float t = __builtin_amdgcn_rcpf(at + __builtin_amdgcn_sqrtf(1.0f + at * at));      // tan(theta), |t| <= 1
    t = tau < 0.0f ? -t : t;
    const float uf = t * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_sqrtf(1.0f + t * t));   // tan(theta/2)
    const double u = (double)uf, u2 = u * u, x = 1.0 + u2;
    double w = (double)__builtin_amdgcn_rcpf((float)x);
    w = w * (2.0 - x * w); w = w * (2.0 - x * w);            // 1e-7 -> 1e-14 -> rounding
    c = (1.0 - u2) * w; s = 2.0 * u * w;
    return 1;
}

// ---- fast path of the Ritz step: Householder tridiagonalisation (Q accumulated) + Sturm-count bisection for the five
// smallest and two largest eigenvalues + inverse iteration for the three smallest pairs, validated against T itself
// (residual and orthonormality); the Jacobi iteration below stays as the fallback.  Jacobi is latency-bound per round
// (N - 1 rounds x 6-9 sweeps x ~0.8 us: 250 us at n = 36); this path is ~n dependent steps of two short matrix-vector
// products plus a few microseconds of scalar recurrences.
#define RITZ_FAST_NMAX 96          /* (3 x VICAN_RITZ_MAX_STEPS; element i of the serial recurrences lives in lane i % 64) */
#define RITZ_FAST_NMIN 9          /* tools/ritz_bench.py, round 4: n = 12 44 vs 42 us (Jacobi), 15: 55 vs 63, 24: 85 vs 116, 36: 134 vs 251, 48: 185 vs 391; round 5 (one-wavefront reduction, division-free Sturm counts): profiles/r05_ritz_bench.txt */
__device__ __forceinline__ double ritz_rcp(double x) {                 // 1/x to ~1 ulp: v_rcp_f64 + one Newton step
    double r = __builtin_amdgcn_rcp(x);
    r = fma(fma(-x, r, 1.0), r, r);
    return r;
}
__device__ __forceinline__ double ritz_readlane(double v, int lane) {   // lane: wave-uniform
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, lane);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)((unsigned long long)b >> 32), lane);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// element i of a lane-distributed array of up to 128 entries: slot i / 64 of lane i % 64 (i: wave-uniform)
__device__ __forceinline__ double ritz_get(double v0, double v1, int i) { return i < 64 ? ritz_readlane(v0, i) : ritz_readlane(v1, i - 64); }
// number of eigenvalues of the symmetric tridiagonal (d, e) below sigma (LAPACK dlaebz recurrence); element i of d and of
// e^2 lives in lane i % 64's registers (NS slots: n <= 64 NS) and is read with v_readlane: no memory latency in the chain
// Round 5 measured two division-free forms of this count (the characteristic polynomials p_i = (d_i - sigma) p_{i-1} -
// e_{i-1}^2 p_{i-2} with power-of-two rescaling, sign changes counted with boolean logic or with integer operations on the
// high words): both SLOWER than the ratio recurrence below - bisection at n = 24: 32.8 / 26.0 against 20.1 us
// (profiles/r05_ritz_bench.txt).  The chain is not bound by the reciprocal but by the four v_readlane per element that feed it.
template <int NS>
__device__ __forceinline__ int ritz_sturm(const double* d, const double* e2, int n, double sigma, double pivmin) {
    double q = ritz_readlane(d[0], 0) - sigma;
    int cnt = 0;
    if (q <= pivmin) { ++cnt; q = fmin(q, -pivmin); }
    for (int i = 1; i < n; ++i) {
        const double di = NS == 1 ? ritz_readlane(d[0], i) : ritz_get(d[0], d[NS - 1], i);
        const double ei = NS == 1 ? ritz_readlane(e2[0], i - 1) : ritz_get(e2[0], e2[NS - 1], i - 1);
        q = di - sigma - ei * ritz_rcp(q);
        if (q <= pivmin) { ++cnt; q = fmin(q, -pivmin); }
    }
    return cnt;
}

// Householder tridiagonalisation of an n x n (n <= NM <= 32) symmetric matrix by ONE wavefront, everything in registers:
// lane i < 32 holds row i of A, lane 32 + i row i of Q (= I on entry); the Householder vector lives one element per lane and
// element j is read with v_readlane (j is a compile-time constant of the unrolled loops: v_j = 0 for j <= k makes every
// sum and update run over all NM columns without predicates); element k of a lane's row - a run-time index - comes out of
// an NM-way select.  A first single-wavefront version kept the rows in LDS and read one element per loop trip: one LDS
// round trip per element, 51 us at n = 24 against 34 for the cooperative form (profiles/r05_ritz_bench.txt).
// On return A (rows > k of the trailing blocks: the tridiagonal part that the caller reads) and Q are back in LDS, fd / fe hold
// the first n - 2 diagonal / sub-diagonal entries.
template <int NM>
__device__ __forceinline__ void ritz_house_regs(double* A, double* V, const int ld, const int n, double* fd, double* fe, const int ln) {
    const bool arow = ln < 32;
    const int ri = ln & 31;
    const bool rowlive = ri < n;
    double* rowp = (arow ? A : V) + ri * ld;
    double row[NM];
#pragma unroll
    for (int j = 0; j < NM; ++j) row[j] = (rowlive && j < n) ? rowp[j] : 0.0;
    for (int k = 0; k + 2 < n; ++k) {
        double sel = 0.0;                                        // element k of this lane's row
#pragma unroll
        for (int j = 0; j < NM; ++j) sel = (j == k) ? row[j] : sel;
        const double akk = ritz_readlane(sel, k), x0 = ritz_readlane(sel, k + 1);     // A[k][k], A[k+1][k] (lanes k, k + 1 < 32)
        const double aik = (arow && ri >= k + 2) ? sel : 0.0;   // column k below the sub-diagonal
        const double sg = wave_allsum_dpp(aik * aik);
        double tau = 0.0, alpha = x0, v0 = 0.0;
        if (sg > 0.0) {                                          // (no reflection when the tail of the column is exactly zero)
            const double mu = sqrt(x0 * x0 + sg);
            alpha = x0 <= 0.0 ? mu : -mu;
            v0 = x0 - alpha;
            tau = 2.0 / (v0 * v0 + sg);
        }
        if (ln == 0) { fd[k] = akk; fe[k] = alpha; }
        if (tau == 0.0) continue;                                // (uniform)
        double vi = (arow && ri == k + 1) ? v0 : aik;           // v_i in lane i: 0 above k + 1
        vi = __shfl(vi, ri, 64);                                 // ... and in lane 32 + i
        const bool live = rowlive && (!arow || ri > k);
        double acc = 0.0;                                        // p_i = tau (A v)_i (lanes < 32, rows > k), g_i = tau (Q v)_i (lanes >= 32)
#pragma unroll
        for (int j = 0; j < NM; ++j) acc += row[j] * ritz_readlane(vi, j);
        acc = live ? acc * tau : 0.0;
        const double K = 0.5 * tau * wave_allsum_dpp(arow ? acc * vi : 0.0);
        const double wi = arow ? acc - K * vi : 0.0;             // w_i (lanes < 32)
        const double fa = arow ? vi : acc, fb = arow ? wi : 0.0; // row -= fa w_j + fb v_j (A)  /  g_i v_j (Q: fa = g_i on v_j)
#pragma unroll
        for (int j = 0; j < NM; ++j) {
            const double vj = ritz_readlane(vi, j), wj = ritz_readlane(wi, j);
            row[j] -= arow ? (fa * wj + fb * vj) : (fa * vj);
        }
    }
#pragma unroll
    for (int j = 0; j < NM; ++j) if (rowlive && j < n) rowp[j] = row[j];
}

__global__ __launch_bounds__(1024) void ritz_kernel(const double* __restrict__ HB, int stride, int hw, int steps, int flags,
                                                    double eig_tol, double floor_tol, double floor_level, double stall_ratio,
                                                    double* __restrict__ Y, double* __restrict__ status,
                                                    int32_t* __restrict__ gate, int fast_lds) {
    extern __shared__ double sm[];
    __shared__ int s_eff, s_idx[5];
    __shared__ double s_th4, s_th5;
    __shared__ double s_red[16];
    const int tid = threadIdx.x, B = blockDim.x;
    if (tid == 0) {
        // a vanished pivot in beta_j: the Krylov space was exhausted at block j+1, the rest is zero padding
        int eff = steps;
        for (int j = 0; j < steps; ++j) {
            const double* b = HB + (size_t)j * stride + hw;
            if (b[0] == 0.0 || b[4] == 0.0 || b[8] == 0.0) { eff = j + 1; break; }
        }
        s_eff = eff;
    }
    __syncthreads();
    const int eff = s_eff, n = 3 * eff, N = n + (n & 1), ld = N | 1, half = N / 2;
    double* A = sm;                      // [N][ld]  (row/column n is a decoupled zero dummy when n is odd)
    double* V = A + (size_t)N * ld;      // [N][ld]
    double* cs = V + (size_t)N * ld;     // [half][2]

    double nf = 0.0;
    for (int idx = tid; idx < N * N; idx += B) {
        const int i = idx / N, c = idx - i * N;
        double v = 0.0;
        if (i < n && c < n) {
            const int lo = i < c ? i : c, hi = i < c ? c : i;         // symmetric completion of the upper triangle
            v = HB[(size_t)(hi / 3) * stride + lo * 3 + (hi % 3)];
        }
        A[i * ld + c] = v;
        V[i * ld + c] = (i == c) ? 1.0 : 0.0;
        nf += v * v;
    }
    nf = block_sum(nf, s_red);
    if (tid == 0) s_red[0] = nf;
    __syncthreads();
    const double normf = sqrt(s_red[0]);
    const double thr = 2e-15 * normf;
    // power-of-two normalisation for the f32 angle evaluation
    int ex = 0;
    if (normf > 0.0) frexp(normf, &ex);
    const double inv = ldexp(1.0, -ex);
    __syncthreads();


    // ---- fast path (see above ritz_kernel): tridiagonalise, bisect, inverse-iterate, validate
    __shared__ int s_fast;
    __shared__ double s_tau, s_gl, s_gu, s_piv;
    if (tid == 0) s_fast = 0;
    if (n >= RITZ_FAST_NMIN && n <= RITZ_FAST_NMAX && fast_lds) {
        double* fd = (double*)((((size_t)(cs + 2 * half)) + (size_t)half * sizeof(int) + 15) & ~(size_t)15);
        double* fe = fd + N; double* fe2 = fe + N; double* fv = fe2 + N; double* fp = fv + N; double* fg = fp + N;
        double* fth = fg + N;                   // [8] eigenvalues: five smallest, second largest, largest
        double* zt = fth + 8;                   // [3][N] eigenvectors of the tridiagonal matrix
        double* ys = zt + 3 * N;                // [3][N] ... and of T
        const int ln = tid & 63, wv = tid >> 6, nwv = B >> 6;
        // Householder tridiagonalisation T = Q^T A Q (A overwritten, Q accumulated in V = I).
        // n <= 32 (up to 10 Lanczos steps - every check of the capture-sized graphs): ONE wavefront does the whole reduction
        // out of REGISTERS (ritz_house_regs): no workgroup barrier, no LDS round trip per element.  The three barriers per step
        // of the cooperative form below (16 wavefronts) are 1.5 us per step, 34 us of an 82 us check at n = 24.
#ifdef RITZ_NO_WAVE1
        if (false) {
#else
        if (n <= 32) {
#endif
          if (wv == 0) { if (n <= 16) ritz_house_regs<16>(A, V, ld, n, fd, fe, ln); else ritz_house_regs<32>(A, V, ld, n, fd, fe, ln); }
          __syncthreads();
        } else
        for (int k = 0; k + 2 < n; ++k) {
            if (wv == 0) {                                  // norm of the column below the subdiagonal; v into fv
                double sg = 0.0;
                for (int i = k + 2 + ln; i < n; i += 64) { const double x = A[i * ld + k]; fv[i] = x; sg += x * x; }
                sg = wave_allsum_dpp(sg);
                if (ln == 0) {
                    const double x0 = A[(k + 1) * ld + k];
                    double tau = 0.0, alpha = x0, v0 = 0.0;
                    if (sg > 0.0) {                         // (no reflection when the tail of the column is exactly zero)
                        const double mu = sqrt(x0 * x0 + sg);
                        alpha = x0 <= 0.0 ? mu : -mu;
                        v0 = x0 - alpha;
                        tau = 2.0 / (v0 * v0 + sg);
                    }
                    s_tau = tau; fv[k + 1] = v0;
                    fd[k] = A[k * ld + k]; fe[k] = alpha;
                }
            }
            __syncthreads();
            const double tau = s_tau;
            if (tau != 0.0) {                               // (uniform)
                for (int i = tid >> 3; i < n; i += B >> 3) {    // p = tau A v (rows below k), g = tau Q v: 8 lanes per row
                    double pa = 0.0, ga = 0.0;
                    const double* Ar = A + i * ld;
                    const double* Vr = V + i * ld;
                    for (int j = k + 1 + (tid & 7); j < n; j += 8) { const double vj = fv[j]; pa += Ar[j] * vj; ga += Vr[j] * vj; }
                    pa = coop_dpp8_sum(pa); ga = coop_dpp8_sum(ga);
                    if ((tid & 7) == 0) { fp[i] = tau * pa; fg[i] = tau * ga; }
                }
                __syncthreads();
                double kk = 0.0;                            // K = tau/2 p.v, by every wavefront for itself (same order: same bits)
                for (int i = k + 1 + ln; i < n; i += 64) kk += fp[i] * fv[i];
                kk = wave_allsum_dpp(kk);
                const double K = 0.5 * tau * kk;
                // A -= v w^T + w v^T on the trailing block, Q -= g v^T on its columns: 32 columns x (B / 32) rows per pass
                for (int i = wv * 2 + (ln >> 5); i < n; i += 2 * nwv) {
                    const double vi = i > k ? fv[i] : 0.0, wi = i > k ? fp[i] - K * vi : 0.0, gi = fg[i];
                    for (int j = k + 1 + (ln & 31); j < n; j += 32) {
                        const double vj = fv[j], wj = fp[j] - K * vj;
                        if (i > k) A[i * ld + j] -= vi * wj + wi * vj;
                        V[i * ld + j] -= gi * vj;
                    }
                }
            }
            __syncthreads();
        }
        // 0: tridiagonalisation
        if (tid == 0) {
            fd[n - 2] = A[(n - 2) * ld + n - 2]; fe[n - 2] = A[(n - 1) * ld + n - 2];
            fd[n - 1] = A[(n - 1) * ld + n - 1]; fe[n - 1] = 0.0;
        }
        __syncthreads();
        if (wv == 0) {                                      // e^2, Gershgorin interval, pivot floor of the Sturm recurrence
            double gl = 1e300, gu = -1e300, em = 0.0;
            for (int i = ln; i < n; i += 64) {
                const double ei = fe[i], r = (i > 0 ? fabs(fe[i - 1]) : 0.0) + (i + 1 < n ? fabs(ei) : 0.0);
                fe2[i] = ei * ei;
                gl = fmin(gl, fd[i] - r); gu = fmax(gu, fd[i] + r); em = fmax(em, ei * ei);
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                gl = fmin(gl, __shfl_xor(gl, o, 64)); gu = fmax(gu, __shfl_xor(gu, o, 64)); em = fmax(em, __shfl_xor(em, o, 64));
            }
            if (ln == 0) {
                const double piv = 2.2250738585072014e-308 * fmax(1.0, em);
                const double w = fmax(fabs(gl), fabs(gu)) * 2.220446049250313e-16 * n * 2.0 + 2.0 * piv;
                s_gl = gl - w; s_gu = gu + w; s_piv = piv;
            }
        }
        __syncthreads();
        // 1: Gershgorin
        auto small_solve = [&](auto ns_) {                  // NS register slots per lane: 1 for n <= 64, 2 up to 128
            constexpr int NS = decltype(ns_)::value;
        {   // bisection, 64 trial shifts per round: a wavefront per wanted eigenvalue
            const double piv = s_piv;
            double dl[NS], e2l[NS];
#pragma unroll
            for (int sl = 0; sl < NS; ++sl) { const int e = ln + 64 * sl; dl[sl] = e < n ? fd[e] : 0.0; e2l[sl] = e < n ? fe2[e] : 0.0; }
            const double btol = 1.1102230246251565e-16 * normf;                       // absolute accuracy asked of an eigenvalue
            for (int t = wv; t < 7; t += nwv) {
                const int kk = t < 5 ? t : n - 7 + t;       // 0..4, n-2, n-1
                double lo = s_gl, hi = s_gu;
                for (int round = 0; round < 14; ++round) {
                    const double sig = lo + (hi - lo) * ((double)(ln + 1) * (1.0 / 65.0));
                    const int c = ritz_sturm<NS>(dl, e2l, n, sig, piv);
                    const unsigned long long mk = __ballot(c >= kk + 1);
                    const int f = mk ? __ffsll((long long)mk) - 1 : 64;
                    const double nhi = f < 64 ? __shfl(sig, f, 64) : hi;
                    const double nlo = f > 0 ? __shfl(sig, f - 1, 64) : lo;
                    lo = nlo; hi = nhi;
                    if (hi - lo <= fmax(4.440892098500626e-16 * fmax(fabs(lo), fabs(hi)), btol) + 2.0 * piv) break;
                }
                if (ln == 0) fth[t] = 0.5 * (lo + hi);
            }
        }
        __syncthreads();
        // 2: bisection
        // inverse iteration for the three smallest pairs (dlagtf / dlagts: LU with partial pivoting of T - theta I), one
        // wavefront per pair, element i of every array in the REGISTERS of lane i % 64 (slot i / 64): the serial recurrences
        // read their operands with v_readlane and write back under a lane predicate - no memory latency in the chain
        {
            const double tiny = fmax(2.220446049250313e-16 * normf, 1e-300);
            double la[NS], lb[NS], lc[NS], l2[NS], ly[NS];
            int lin[NS];
#pragma unroll
            for (int sl = 0; sl < NS; ++sl) { la[sl] = lb[sl] = lc[sl] = l2[sl] = ly[sl] = 0.0; lin[sl] = 0; }
#define RG(a, i) (NS == 1 ? ritz_readlane((a)[0], (i)) : ritz_get((a)[0], (a)[NS - 1], (i)))
#define RSET(a, i, v) do { if (ln == ((i) & 63)) { if (NS == 1 || (i) < 64) (a)[0] = (v); else (a)[NS - 1] = (v); } } while (0)
            if (wv < 3) {
                const double th = fth[wv];
#pragma unroll
                for (int sl = 0; sl < NS; ++sl) {
                    const int e = ln + 64 * sl;
                    if (e < n) { la[sl] = fd[e] - th; lb[sl] = fe[e]; lc[sl] = fe[e]; ly[sl] = 1.0 + 0.125 * (double)((e * 7 + wv * 3) % 11); }
                }
                for (int i = 0; i + 1 < n; ++i) {
                    const double ai = RG(la, i), ci = RG(lc, i), bi = RG(lb, i), a1 = RG(la, i + 1), b1 = RG(lb, i + 1);
                    const bool sw = fabs(ci) > fabs(ai);
                    const double piv_ = sw ? ci : (ai != 0.0 ? ai : tiny);
                    const double mult = (sw ? ai : ci) * ritz_rcp(piv_);
                    const double na1 = sw ? bi - mult * a1 : a1 - mult * bi;
                    RSET(la, i, piv_); RSET(lc, i, mult);
                    if (ln == (i & 63)) { if (NS == 1 || i < 64) lin[0] = sw ? 1 : 0; else lin[NS - 1] = sw ? 1 : 0; }
                    if (sw) { RSET(lb, i, a1); RSET(l2, i, b1); }
                    RSET(la, i + 1, na1);
                    if (sw) RSET(lb, i + 1, -mult * b1);
                }
                { const double an = RG(la, n - 1); if (fabs(an) < tiny) RSET(la, n - 1, tiny); }
            }
            for (int it = 0; it < 2; ++it) {                // (theta is accurate to rounding: the second iterate is converged;
                if (wv < 3) {                               //  the validation below catches the rest)
                    for (int i = 0; i + 1 < n; ++i) {       // forward substitution with the recorded interchanges
                        const double yi = RG(ly, i), y1 = RG(ly, i + 1), ci = RG(lc, i);
                        const int sw = (NS == 1 || i < 64) ? __builtin_amdgcn_readlane(lin[0], i & 63) : __builtin_amdgcn_readlane(lin[NS - 1], i - 64);
                        const double nyi = sw ? y1 : yi, ny1 = sw ? yi - ci * y1 : y1 - ci * yi;
                        RSET(ly, i, nyi);
                        RSET(ly, i + 1, ny1);
                    }
                    double x1 = 0.0, x2 = 0.0;              // back substitution: x_{i+1}, x_{i+2} carried as uniform values
                    for (int i = n - 1; i >= 0; --i) {
                        double pv = RG(la, i);
                        if (fabs(pv) < tiny) pv = pv < 0.0 ? -tiny : tiny;
                        const double xi = (RG(ly, i) - RG(lb, i) * x1 - RG(l2, i) * x2) * ritz_rcp(pv);
                        RSET(ly, i, xi);
                        x2 = x1; x1 = xi;
                    }
                    // scale against overflow before the products below
                    double mx = 0.0;
#pragma unroll
                    for (int sl = 0; sl < NS; ++sl) mx = fmax(mx, ln + 64 * sl < n ? fabs(ly[sl]) : 0.0);
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) mx = fmax(mx, __shfl_xor(mx, o, 64));
                    const double isc = mx > 0.0 ? 1.0 / mx : 0.0;
#pragma unroll
                    for (int sl = 0; sl < NS; ++sl) ly[sl] *= isc;
                }
                // modified Gram-Schmidt among the three (a cluster of nearly equal eigenvalues shares its vectors otherwise)
                for (int q = 0; q < 3; ++q) {
                    if (wv == q) {
                        for (int p2 = 0; p2 < q; ++p2) {
                            double zp[NS], dt = 0.0;
#pragma unroll
                            for (int sl = 0; sl < NS; ++sl) { const int e = ln + 64 * sl; zp[sl] = e < n ? zt[p2 * N + e] : 0.0; dt += e < n ? ly[sl] * zp[sl] : 0.0; }
#pragma unroll
                            for (int o = 32; o > 0; o >>= 1) dt += __shfl_xor(dt, o, 64);
#pragma unroll
                            for (int sl = 0; sl < NS; ++sl) ly[sl] -= dt * zp[sl];
                        }
                        double nn = 0.0;
#pragma unroll
                        for (int sl = 0; sl < NS; ++sl) nn += ln + 64 * sl < n ? ly[sl] * ly[sl] : 0.0;
#pragma unroll
                        for (int o = 32; o > 0; o >>= 1) nn += __shfl_xor(nn, o, 64);
                        const double inn = nn > 0.0 ? 1.0 / sqrt(nn) : 0.0;
#pragma unroll
                        for (int sl = 0; sl < NS; ++sl) { ly[sl] *= inn; if (ln + 64 * sl < n) zt[q * N + ln + 64 * sl] = ly[sl]; }
                    }
                    __syncthreads();
                }
            }
#undef RG
#undef RSET
        }
        };
        if (n <= 64) small_solve(std::integral_constant<int, 1>()); else small_solve(std::integral_constant<int, 2>());
        // 3: inverse iteration
        // back-transformation y = Q z, then validation against T itself
        for (int idx = tid >> 3; idx < 3 * n; idx += B >> 3) {
            const int k = idx / n, i = idx - k * n;
            const double* Vr = V + i * ld;
            const double* z = zt + k * N;
            double acc = 0.0;
            for (int j = tid & 7; j < n; j += 8) acc += Vr[j] * z[j];
            acc = coop_dpp8_sum(acc);
            if ((tid & 7) == 0) ys[k * N + i] = acc;
        }
        for (int i = wv; i < n; i += nwv)                   // T again (the tridiagonalisation overwrote it)
            for (int c = ln; c < n; c += 64) {
                const int lo = i < c ? i : c, hi = i < c ? c : i;
                A[i * ld + c] = HB[(size_t)(hi / 3) * stride + lo * 3 + (hi % 3)];
            }
        __syncthreads();
        double bad = 0.0;
        for (int idx = tid >> 3; idx < 3 * n; idx += B >> 3) {
            const int k = idx / n, i = idx - k * n;
            const double* Ar = A + i * ld;
            const double* y = ys + k * N;
            double acc = 0.0;
            for (int j = tid & 7; j < n; j += 8) acc += Ar[j] * y[j];
            acc = coop_dpp8_sum(acc);
            bad = fmax(bad, fabs(acc - fth[k] * y[i]));
        }
        if (tid < 9) {
            const int p2 = tid / 3, q = tid - 3 * p2;
            double dt = 0.0;
            for (int i = 0; i < n; ++i) dt += ys[p2 * N + i] * ys[q * N + i];
            bad = fmax(bad, fabs(dt - (p2 == q ? 1.0 : 0.0)) * normf);
        }
        if (!(bad == bad)) bad = 1e300;                     // NaN
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) bad = fmax(bad, __shfl_xor(bad, o, 64));
        if (ln == 0) s_red[wv] = bad;
        __syncthreads();
        if (tid == 0) {
            double m = 0.0;
            for (int w = 0; w < nwv; ++w) m = fmax(m, s_red[w]);
            bool ok = m <= 1e-13 * normf;
            for (int t = 0; t < 6 && ok; ++t) if (!(fth[t] <= fth[t + 1])) ok = false;       // ordered: 5 smallest, 2 largest
            s_fast = ok ? 1 : 0;
        }
        __syncthreads();
        // 4: back-transformation + validation
        if (s_fast) {
            // hand the result to the common tail in the Jacobi layout: eigenvalue k on the diagonal, its vector in column k
            for (int idx = tid; idx < 3 * n; idx += B) { const int k = idx / n, i = idx - k * n; V[i * ld + k] = ys[k * N + i]; }
            if (tid == 0) {
                for (int k = 0; k < 3; ++k) { A[k * ld + k] = fth[k]; s_idx[k] = k; }
                A[3 * ld + 3] = fth[5]; A[4 * ld + 4] = fth[6]; s_idx[3] = 3; s_idx[4] = 4;
                s_th4 = fth[3]; s_th5 = fth[4];
            }
        } else {
            for (int idx = tid; idx < N * N; idx += B) { const int i = idx / N, c = idx - i * N; V[i * ld + c] = (i == c) ? 1.0 : 0.0; }
        }
        __syncthreads();
    }
    const bool fast = s_fast != 0;
    if (!fast) {
    // Termination: a sweep without rotations, or - cheaper - a sweep whose largest pivot was already
    // below 1e-8 |T|_F: Jacobi converges quadratically, so that sweep left the off-diagonal part at
    // the 1e-16 |T|_F (|T|_F / gap) level and neither a polishing nor a verifying sweep is needed.
    const double small = 1e-8 * normf;
    int sweeps = 0;
    for (; sweeps < 30; ++sweeps) {
        int state = 0;                          // bit 0: rotated, bit 1: a pivot above `small` was seen
        for (int r = 0; r < N - 1; ++r) {
            if (tid < half) {                   // thread k: rotation of pair k from its three pivot entries
                int p, q;
                ritz_pair(tid, r, N, p, q);
                const double apq = A[p * ld + q], app = A[p * ld + p], aqq = A[q * ld + q];    // one LDS latency
                double c, s;
                if (ritz_rotation(app, aqq, apq, q < n, thr, inv, c, s)) state |= fabs(apq) > small ? 3 : 1;
                cs[2 * tid] = c; cs[2 * tid + 1] = s;
            }
            __syncthreads();
            for (int b = tid; b < half * half; b += B) {
                const int ki = b / half, kj = b - ki * half;
                int pi, qi, pj, qj;
                ritz_pair(ki, r, N, pi, qi);                  // recomputed (ALU) so that every LDS read below
                ritz_pair(kj, r, N, pj, qj);                  // is independent: one latency for all of them
                double* a0 = A + pi * ld; double* a1 = A + qi * ld;
                double* v0 = V + pi * ld; double* v1 = V + qi * ld;
                const double ci = cs[2 * ki], si = cs[2 * ki + 1], cj = cs[2 * kj], sj = cs[2 * kj + 1];
                const double b00 = a0[pj], b01 = a0[qj], b10 = a1[pj], b11 = a1[qj];
                const double e00 = v0[pj], e01 = v0[qj], e10 = v1[pj], e11 = v1[qj];
                // columns (J_j), then rows (J_i^T); identity rotations reproduce the entries exactly
                const double t00 = cj * b00 - sj * b01, t01 = sj * b00 + cj * b01;
                const double t10 = cj * b10 - sj * b11, t11 = sj * b10 + cj * b11;
                double n00 = ci * t00 - si * t10, n01 = ci * t01 - si * t11;
                double n10 = si * t00 + ci * t10, n11 = si * t01 + ci * t11;
                if (ki == kj) { const double o = 0.5 * (n01 + n10); n01 = o; n10 = o; }
                a0[pj] = n00; a0[qj] = n01; a1[pj] = n10; a1[qj] = n11;
                v0[pj] = cj * e00 - sj * e01; v0[qj] = sj * e00 + cj * e01;
                v1[pj] = cj * e10 - sj * e11; v1[qj] = sj * e10 + cj * e11;
            }
            __syncthreads();
        }
        const int any_rot = __syncthreads_or(state & 1), any_big = __syncthreads_or(state & 2);
        if (!any_rot || !any_big) { ++sweeps; break; }
    }

    // ranks of the eigenvalues (ties broken by index): one thread per eigenvalue
    if (tid < 5) s_idx[tid] = -1;
    if (tid == 0) s_th4 = s_th5 = __longlong_as_double(0x7ff8000000000000LL);
    __syncthreads();
    if (tid < n) {
        const double d = A[tid * ld + tid];
        int rank = 0;
        for (int j = 0; j < n; ++j) {
            const double e = A[j * ld + j];
            rank += (e < d || (e == d && j < tid)) ? 1 : 0;
        }
        if (rank < 3) s_idx[rank] = tid;                                    // three smallest
        if (rank == 3) s_th4 = d;                                           // fourth smallest (reported only)
        if (rank == 4) s_th5 = d;                                           // fifth smallest (the reference's max_eval exit)
        if (n >= 2 && rank >= n - 2) s_idx[3 + (rank - (n - 2))] = tid;     // second largest, largest
        if (n == 1) s_idx[4] = tid;
    }
    __syncthreads();
    }
    // Ritz vectors of the three smallest values -> Y[3 steps][3] (zero rows beyond the effective basis)
    for (int idx = tid; idx < 3 * steps * 3; idx += B) {
        const int i = idx / 3, k = idx - 3 * i;
        Y[idx] = (i < n && s_idx[k] >= 0) ? V[i * ld + s_idx[k]] : 0.0;
    }
    if (tid == 0) {
        const double* beta = HB + (size_t)(eff - 1) * stride + hw;
        double resmax = 0.0;
        for (int k = 0; k < 3; ++k) {
            if (s_idx[k] < 0) continue;
            double q = 0.0;
            for (int a = 0; a < 3; ++a) {
                double v = 0.0;
                for (int b = 0; b < 3; ++b) v += beta[a * 3 + b] * V[(n - 3 + b) * ld + s_idx[k]];
                q += v * v;
            }
            resmax = fmax(resmax, sqrt(q));
        }
        const double th_min = A[s_idx[0] * ld + s_idx[0]], th_max = A[s_idx[4] * ld + s_idx[4]];
        const double scale = fmax(fmax(fabs(th_min), fabs(th_max)), 1e-300);
        const double r = resmax / scale;
        const bool first = flags & 1, at_max = flags & 2;
        const bool breakdown = beta[0] == 0.0 && beta[4] == 0.0 && beta[8] == 0.0;
        const double prev = status[12];
        bool floor_hit = !first && r > stall_ratio * prev && r <= floor_tol;
        if (floor_level >= 0.0 && r <= 2.0 * floor_level) floor_hit = true;
        const bool stop = eff < steps || breakdown || r <= eig_tol || floor_hit || at_max;
        const bool converged = breakdown || floor_hit || resmax <= eig_tol * scale;
        const double nan = __longlong_as_double(0x7ff8000000000000LL);
        status[0] = r; status[1] = scale; status[2] = stop; status[3] = converged; status[4] = floor_hit;
        status[5] = eff; status[6] = breakdown;
        for (int k = 0; k < 3; ++k) status[7 + k] = s_idx[k] >= 0 ? A[s_idx[k] * ld + s_idx[k]] : nan;
        status[10] = n >= 5 ? A[s_idx[3] * ld + s_idx[3]] : nan;
        status[11] = n >= 5 ? th_max : nan;
        status[12] = r; status[13] = s_th5; status[14] = resmax; status[15] = s_th4;
        *gate = (stop && converged) ? 1 : 0;
    }
}

extern "C" int vican_ritz(const double* HB, int32_t row_stride, int32_t hw, int32_t steps, int32_t flags, double eig_tol,
                          double floor_tol, double floor_level, double stall_ratio, double* Y, double* status, int32_t* gate,
                          void* stream) {
    if (!HB || !Y || !status || !gate || !(stall_ratio > 0.0 && stall_ratio < 1.0) || steps < 1 || steps > VICAN_RITZ_MAX_STEPS || hw < 9 * steps || row_stride < hw + 9)
        return set_err(VICAN_ERR_ARG, "vican_ritz: bad argument");
    const int n = 3 * steps, N = n + (n & 1), ld = N | 1, half = N / 2;
    size_t lds = ((size_t)2 * N * ld + 2 * half) * sizeof(double) + (size_t)half * sizeof(int) + 16;
    const int fast_lds = n >= RITZ_FAST_NMIN && n <= RITZ_FAST_NMAX;     // workspace of the tridiagonalisation path
    if (fast_lds) lds += ((size_t)12 * N + 8) * sizeof(double) + 32;   // d, e, e^2, v, p, g, 8 eigenvalues, z[3], y[3]
    static size_t configured = 0;
    if (lds > 64 * 1024 && lds > configured) {
        if (hipFuncSetAttribute((const void*)ritz_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return set_err(VICAN_ERR_LAUNCH, "vican_ritz: cannot raise dynamic LDS limit");
        configured = lds;
    }
    int threads = ((half * half + 63) / 64) * 64;            // one thread per 2x2 block, at most 1024
    threads = threads > 1024 ? 1024 : threads;
    if (fast_lds && threads < 448) threads = 448;                      // seven wavefronts: one per wanted eigenvalue
    hipLaunchKernelGGL(ritz_kernel, dim3(1), dim3(threads), lds, (hipStream_t)stream, HB, row_stride, hw, steps,
                       flags, eig_tol, floor_tol, floor_level, stall_ratio, Y, status, gate, fast_lds);
    LAUNCH_CHECK("vican_ritz");
    return VICAN_OK;
}
