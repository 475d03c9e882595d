// vican_common.h - shared helpers of the vican HIP kernels (error plumbing, wavefront
// reductions, 3x3 SVD / polar factor, LDS budget).  Included by every .hip file.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "vican_hip.h"

// ---------------------------------------------------------------------------
// error plumbing (one message buffer per host thread, defined in vican_sweep.hip)
// ---------------------------------------------------------------------------
extern thread_local char g_vican_err[512];

static inline int set_err(int code, const char* fmt, const char* a = "") {
    snprintf(g_vican_err, sizeof(g_vican_err), fmt, a);
    return code;
}
#define LAUNCH_CHECK(name)                                                                     \
    do {                                                                                       \
        hipError_t e_ = hipGetLastError();                                                     \
        if (e_ != hipSuccess) {                                                                \
            snprintf(g_vican_err, sizeof(g_vican_err), "%s: %s", name, hipGetErrorString(e_)); \
            return VICAN_ERR_LAUNCH;                                                           \
        }                                                                                      \
    } while (0)

int vican_check_graph(const vican_graph_t* g, const char* who);   // vican_sweep.hip
int vican_check_block_graph(const vican_graph_t* g, const char* who);   // ... and layout == VICAN_LAYOUT_BLOCK

// Launch gate (vican_set_gate, per host thread, defined in vican_sweep.hip): kernels that honour it take
// `const int32_t* gate` as their last parameter and start with GATE_RETURN - the whole grid exits
// unless *gate == 1 at execution time (speculatively enqueued work that a device-side decision cancels).
extern thread_local const int32_t* g_vican_gate;
#define GATE_RETURN(gate) do { if ((gate) != nullptr && *(gate) != 1) return; } while (0)
// Launch timer (vican_set_launch_events, per host thread, defined in vican_sweep.hip): the NEXT edge-kernel launch (operator /
// dual-update sweeps, and on the wave layout the CG product, the right-hand side and the fused LSQR pass) binds
// these two HIP events to its own dispatch (hipExtLaunchKernelGGL: start / stop = begin / end of the kernel on the
// device, the timestamps a kernel trace shows) and clears them.  An event pair recorded around a launch instead also
// times the 5-8 us the queue idles between an event command and the next dispatch.
extern thread_local hipEvent_t g_vican_ev_start, g_vican_ev_stop;
#define VICAN_LAUNCH_SWEEP(kern, grid, block, lds, st, ...)                                                        \
    do {                                                                                                           \
        if (g_vican_ev_start && g_vican_ev_stop) {                                                                 \
            hipExtLaunchKernelGGL(kern, grid, block, (uint32_t)(lds), st, g_vican_ev_start, g_vican_ev_stop, 0, __VA_ARGS__); \
            g_vican_ev_start = g_vican_ev_stop = nullptr;                                                          \
        } else {                                                                                                   \
            hipLaunchKernelGGL(kern, grid, block, lds, st, __VA_ARGS__);                                           \
        }                                                                                                          \
    } while (0)
// (double-word camera sums [2][3][C] + R_c planes [9][C]; double-word row stripes + double-buffered R_t staging)
static inline int64_t rhs_lds_bytes(int32_t n_cam, int32_t max_rows, int32_t n_copy) { return 120LL * n_cam + (int64_t)max_rows * (48LL * n_copy + 144) + 256; }
// (double-word camera sums [2][3][C] + p_c planes; double-buffered double-word row stripes + two staging arrays)
static inline int64_t cg_lds_bytes(int32_t n_cam, int32_t max_rows, int32_t n_copy) { return 72LL * n_cam + (int64_t)max_rows * (96LL * n_copy + 96) + 256; }
// ---------------------------------------------------------------------------
// Grid barriers of the cooperative kernels (lanczos_cam_coop_kernel, cg_resident_kernel, tiled_sweep_kernel)
// ---------------------------------------------------------------------------
// These kernels spin on a device counter, which only terminates if every workgroup of the grid is resident.  Three
// layers make that safe:
//   1. the launchers refuse grids that cannot be co-resident on an idle device (vican_coresident_ok: occupancy query x
//      compute units), so the host falls back to the launch-sequence paths;
//   2. every spin is BOUNDED: after `limit` ticks of the 100 MHz real-time counter (vican_set_barrier_abort, default 2 s) a
//      workgroup raises the abort word and leaves; every other workgroup sees the word in its own spin loop (or at its
//      first barrier if it was dispatched late) and leaves too - a grid that shares the device with something that keeps
//      its workgroups out (another process's resident kernel, a CU mask) ends with an error instead of hanging the queue;
//   3. the abort word lives in host-visible memory: the host polls it for free, re-runs on the launch-sequence path and
//      stops using the cooperative kernels (device.HipBackend.barrier_aborted).
// `fenced`: agent-scope RELEASE before the arrival / ACQUIRE after the exit (the portable ordering; costs an L2 write-back,
// 0.4-0.8 us with little dirty data).  Without it the callers' cross-workgroup data must travel as agent-scope atomics.
extern thread_local uint32_t* g_vican_abort_word;          // vican_sweep.hip (vican_set_barrier_abort)
extern thread_local unsigned long long g_vican_sync_ticks;
struct vican_sync_t { unsigned int* counter; uint32_t* abort_word; unsigned long long limit; };
static inline vican_sync_t vican_sync_args(unsigned int* counter) { return vican_sync_t{counter, g_vican_abort_word, g_vican_sync_ticks}; }
int vican_coresident_ok(const void* kernel, int block_threads, size_t lds_bytes, int grid, const char* who);   // vican_sweep.hip

// returns true when the barrier was passed, false when the launch is aborted (the caller returns from the kernel)
__device__ __forceinline__ bool vican_grid_sync(const vican_sync_t& sy, unsigned int target, bool fenced) {
    __shared__ int s_pass;
    __syncthreads();
    if (threadIdx.x == 0) {
        if (fenced) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __hip_atomic_fetch_add(sy.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        int pass = 1;
        if (__hip_atomic_load(sy.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            unsigned int spins = 0;
            while (__hip_atomic_load(sy.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if ((++spins & 1023u) == 0 && sy.abort_word) {           // (host memory: looked at every ~1000 polls only)
                    if (__hip_atomic_load(sy.abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) { pass = 0; break; }
                    if (__builtin_amdgcn_s_memrealtime() - t0 > sy.limit) {
                        __hip_atomic_store(sy.abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        pass = 0; break;
                    }
                }
            }
        }
        if (fenced) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        s_pass = pass;
    }
    __syncthreads();
    return s_pass != 0;
}

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
// a b + c with TWO roundings (product, then sum) - scipy's CG forms its vector updates that way (p *= beta; p += r;
// x += alpha * p; r -= alpha * q), and the loosely converged iterate is a chaotic function of such roundings (DESIGN.md section
// 2): the recurrence's updates are spelled like scipy's instead of being left to -ffp-contract=fast
__device__ __forceinline__ double mul_add_2r(double a, double b, double c) {
#pragma clang fp contract(off)
    const double t = a * b;
    return t + c;
}
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
    for (int sweep = 0; sweep < 20; ++sweep) {
        bool rotated = false;
#pragma unroll
        for (int pq = 0; pq < 3; ++pq) {
            const int p = (pq == 2) ? 1 : 0, q = (pq == 0) ? 1 : 2;
            const double al = a[p][0] * a[p][0] + a[p][1] * a[p][1] + a[p][2] * a[p][2];
            const double be = a[q][0] * a[q][0] + a[q][1] * a[q][1] + a[q][2] * a[q][2];
            const double ga = a[p][0] * a[q][0] + a[p][1] * a[q][1] + a[p][2] * a[q][2];
            // columns count as orthogonal at |a_p.a_q| <= 4 eps |a_p||a_q| (quadratic convergence:
            // typically 4-6 sweeps; the old 1e-16 threshold sat below rounding noise and always ran 30)
            if (ga != 0.0 && fabs(ga) > 8.9e-16 * sqrt(al * be)) {
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
// mode & 3: 0 none, 1 lam = U S U^T, 2 lam = U S^-1 U^T   (S NOT sign corrected: bipgo.py:312,329)
// mode & 4: R = U V^T without the det fix (the non-eliminated solver, bipgo.py:126-127)
__device__ void polar_dual3(const double* A, double* R, double* lam, int mode) {
    double U[9], s[3], V[9];
    svd3(A, U, s, V);
    const double d = (!(mode & 4) && det3(U) * det3(V) < 0.0) ? -1.0 : 1.0;
    if (R) {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                R[i * 3 + j] = U[i * 3 + 0] * V[j * 3 + 0] + U[i * 3 + 1] * V[j * 3 + 1] +
                               d * U[i * 3 + 2] * V[j * 3 + 2];
    }
    if (lam && (mode & 3)) {
        double f[3];
#pragma unroll
        for (int k = 0; k < 3; ++k) f[k] = ((mode & 3) == 1) ? s[k] : 1.0 / s[k];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j)
                lam[i * 3 + j] = f[0] * U[i * 3 + 0] * U[j * 3 + 0] + f[1] * U[i * 3 + 1] * U[j * 3 + 1] +
                                 f[2] * U[i * 3 + 2] * U[j * 3 + 2];
    }
}

// Orthogonal polar factor U V^T of A (NO det fix: a reflection for det A < 0) by the Newton iteration
// X <- (X + X^-T) / 2 started from A scaled to |A|_F = sqrt 3.  Timestep sums Z_t = sum_c M_ct^T R_c and the
// camera-side products are close to a scaled rotation, where it converges quadratically in 3-4 steps of ~80 flops -
// cheap enough to run inside the streaming sweep (the Jacobi SVD is ~20x that).  Only used for well-conditioned
// blocks (normalised determinant > 1e-3, i.e. cond <~ 1e3: the early iterates invert X, and their rounding errors
// (~cond eps) perturb the matrix whose polar factor the iteration converges to); returns false otherwise.
__device__ inline bool polar_newton_core(const double* A, double* X) {
    double n2 = 0.0;
#pragma unroll
    for (int q = 0; q < 9; ++q) n2 += A[q] * A[q];
    if (!(n2 > 0.0 && n2 < 1e300)) return false;
    const double sc = sqrt(3.0 / n2);
#pragma unroll
    for (int q = 0; q < 9; ++q) X[q] = A[q] * sc;
    for (int it = 0; it < 24; ++it) {
        double Cf[9];
        Cf[0] = X[4] * X[8] - X[5] * X[7]; Cf[1] = X[5] * X[6] - X[3] * X[8]; Cf[2] = X[3] * X[7] - X[4] * X[6];
        Cf[3] = X[2] * X[7] - X[1] * X[8]; Cf[4] = X[0] * X[8] - X[2] * X[6]; Cf[5] = X[1] * X[6] - X[0] * X[7];
        Cf[6] = X[1] * X[5] - X[2] * X[4]; Cf[7] = X[2] * X[3] - X[0] * X[5]; Cf[8] = X[0] * X[4] - X[1] * X[3];
        const double det = X[0] * Cf[0] + X[1] * Cf[1] + X[2] * Cf[2];
        if (!(fabs(det) > 1e-3)) return false;          // |det| only grows towards 1 along the iteration: decided at it = 0
        const double h = 0.5 / det;
        double delta = 0.0;
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const double nx = 0.5 * X[q] + h * Cf[q];
            const double d = nx - X[q];
            delta += d * d;
            X[q] = nx;
        }
        if (delta <= 1e-16) return true;                // |step| <= 1e-8: quadratic convergence leaves the new iterate at ~step^2/2
    }
    return false;
}

// out of line on purpose: the (never taken in practice) SVD path must not raise the register pressure of the caller
__device__ __attribute__((noinline)) void polar_svd_fallback(const double* A, double* R) { polar_dual3(A, R, nullptr, 4); }

__device__ inline void polar_newton3(const double* A, double* R) {
    double X[9];
    if (polar_newton_core(A, X)) {
#pragma unroll
        for (int q = 0; q < 9; ++q) R[q] = X[q];
    } else {
        polar_svd_fallback(A, R);
    }
}

// polar_dual3 without the SVD where the block allows it: Q = U V^T from the Newton iteration; for det A > 0 that is
// the det-fixed rotation, A Q^T = U S U^T is the dual block (mode 1) and its inverse U S^-1 U^T (mode 2).
// Reflections (det A < 0 with the det fix requested: needs the third singular pair) and ill-conditioned blocks
// take the SVD path.
__device__ inline void polar_dual3_fast(const double* A, double* R, double* lam, int mode) {
    double Q[9];
    const bool ok = polar_newton_core(A, Q);
    if (!ok || (!(mode & 4) && det3(Q) < 0.0)) { polar_dual3(A, R, lam, mode); return; }
    if (R) {
#pragma unroll
        for (int q = 0; q < 9; ++q) R[q] = Q[q];
    }
    if (lam && (mode & 3)) {
        double H[9];
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) H[i * 3 + j] = A[i * 3] * Q[j * 3] + A[i * 3 + 1] * Q[j * 3 + 1] + A[i * 3 + 2] * Q[j * 3 + 2];
        const double h01 = 0.5 * (H[1] + H[3]), h02 = 0.5 * (H[2] + H[6]), h12 = 0.5 * (H[5] + H[7]);
        if ((mode & 3) == 1) {
            lam[0] = H[0]; lam[1] = h01; lam[2] = h02; lam[3] = h01; lam[4] = H[4]; lam[5] = h12; lam[6] = h02; lam[7] = h12; lam[8] = H[8];
        } else {                                        // inverse of the symmetric positive definite H by cofactors
            const double c00 = H[4] * H[8] - h12 * h12, c01 = h02 * h12 - h01 * H[8], c02 = h01 * h12 - h02 * H[4];
            const double c11 = H[0] * H[8] - h02 * h02, c12 = h01 * h02 - H[0] * h12, c22 = H[0] * H[4] - h01 * h01;
            const double id = 1.0 / (H[0] * c00 + h01 * c01 + h02 * c02);
            lam[0] = c00 * id; lam[1] = c01 * id; lam[2] = c02 * id; lam[3] = c01 * id; lam[4] = c11 * id; lam[5] = c12 * id;
            lam[6] = c02 * id; lam[7] = c12 * id; lam[8] = c22 * id;
        }
    }
}

// --- 64-bit fixed-point accumulation helpers (see vican_sweep.hip header) -----
typedef unsigned long long u64;

// double -> 64-bit fixed point (round to nearest) by the magic-number trick; |v*scale| < 2^51
__device__ __forceinline__ u64 to_fix(double v, double scale) {
    const double magic = 6755399441055744.0;          // 1.5 * 2^52
    return (u64)(__double_as_longlong(fma(v, scale, magic)) - __double_as_longlong(magic));
}
__device__ __forceinline__ void lds_add_fix(u64* p, u64 v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);     // ds_add_u64
}


__device__ __forceinline__ void atomic_max_pos(double* addr, double v) {
    // non-negative doubles order like their bit patterns
    atomicMax((unsigned long long*)addr, (unsigned long long)__double_as_longlong(v));
}

