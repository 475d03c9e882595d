// vican_merge.hip - the numeric half of the front-end on the device (reference bipgo.py:203-221 and 445-469: the two per-edge
// Python loops that apply the marker constraints, weight, and merge multi-marker detections of one (camera, timestep) pair).
//
// Input: one entry per KEPT source edge as arrays - camera / timestep / marker INDEX (the host keeps the string -> index
// maps and the user's callables, frontend.index_edges), measured rotation R~ and translation t~, the two weights k_r, k_t -
// plus the per-marker tables R_m^T R_root and (R_root^T R_m) tau_m.  Output: the merged timestep-major CSR problem
// (row_ptr, col, M_ct, a_ct, w_ct, u_ct, v_ct) and the diagonal of the reference's J^T J (deg_c, deg_t), all in HBM, ready
// for LocalGraph.
//
// Bit-for-bit the arithmetic of frontend.merge_host (tests/test_merge_gpu.py): every sum runs SEQUENTIALLY in source-edge
// order (stable radix sorts keep that order inside a segment; one thread walks a segment), products and sums are single
// IEEE operations in the order spelled out there (no fused multiply-add: `fp contract(off)`), and the entries of J^T J are
// accumulated in the matrix dtype like scipy's csr_matmat does (float32 products summed in float32 for dtype=float32).
// Sorting: rocPRIM radix sort (stable) on (timestep * C + camera) for the merge, on camera / on timestep for the diagonal.
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

#include "vican_common.h"

#pragma clang fp contract(off)

__global__ void merge_keys_kernel(long long n, int n_cam, const int32_t* __restrict__ cam, const int32_t* __restrict__ tim,
                                  unsigned long long* __restrict__ key, uint32_t* __restrict__ kc, uint32_t* __restrict__ kt,
                                  uint32_t* __restrict__ idx) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    key[i] = (unsigned long long)tim[i] * (unsigned long long)n_cam + (unsigned long long)cam[i];
    kc[i] = (uint32_t)cam[i]; kt[i] = (uint32_t)tim[i]; idx[i] = (uint32_t)i;
}
__global__ void merge_heads_kernel(long long n, const unsigned long long* __restrict__ key, int32_t* __restrict__ head) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) head[i] = (i == 0 || key[i] != key[i - 1]) ? 1 : 0;
}
// seg[i] = inclusive scan of head = 1-based segment number of sorted position i
__global__ void merge_starts_kernel(long long n, int n_cam, int n_time, const unsigned long long* __restrict__ key,
                                    const int32_t* __restrict__ seg, int32_t* __restrict__ start, int32_t* __restrict__ n_merged,
                                    int32_t* __restrict__ row_ptr, int32_t* __restrict__ col) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int e = seg[i] - 1;
    if (i == 0 || key[i] != key[i - 1]) {
        start[e] = (int32_t)i;
        col[e] = (int32_t)(key[i] % (unsigned long long)n_cam);
        const long long row = (long long)(key[i] / (unsigned long long)n_cam);
        if (i == 0 || (long long)(key[i - 1] / (unsigned long long)n_cam) != row) row_ptr[row] = e;
    }
    if (i == n - 1) { start[e + 1] = (int32_t)n; *n_merged = e + 1; row_ptr[n_time] = e + 1; }
}

template <typename W>
__global__ void merge_segments_kernel(long long n, const int32_t* __restrict__ n_merged, const int32_t* __restrict__ start,
                                      const uint32_t* __restrict__ idx, const int32_t* __restrict__ marker,
                                      const double* __restrict__ R, const double* __restrict__ t, const double* __restrict__ kr,
                                      const uint8_t* __restrict__ kr_f32,
                                      const double* __restrict__ kt, const double* __restrict__ CmT, const double* __restrict__ qtau,
                                      double* __restrict__ blk, double* __restrict__ a, double* __restrict__ w, double* __restrict__ u,
                                      double* __restrict__ v) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= *n_merged) return;
    double M[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0}, sa = 0.0, su[3] = {0, 0, 0}, sv[3] = {0, 0, 0};
    W sw = (W)0;
    for (int i = start[e]; i < start[e + 1]; ++i) {
        const uint32_t s = idx[i];
        const int m = marker[s];
        const double k = kr[s], ktv = kt[s];
        double A[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) A[q] = k * R[(size_t)s * 9 + q];
        // a float32 rotation weighted by a Python scalar: numpy forms `k_r * R` (bipgo.py:213) in float32 - weight and product
        // rounded to float32 (object mode: SE3.inv() returns float32 rotations, geometry.py:239-243)
        if (kr_f32 && kr_f32[s]) {
            const float kf = (float)k;
#pragma unroll
            for (int q = 0; q < 9; ++q) A[q] = (double)__fmul_rn(kf, (float)R[(size_t)s * 9 + q]);
        }
        const double* B = CmT + (size_t)m * 9;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const double p0 = A[r * 3 + 0] * B[0 * 3 + c], p1 = A[r * 3 + 1] * B[1 * 3 + c], p2 = A[r * 3 + 2] * B[2 * 3 + c];
                M[r * 3 + c] = M[r * 3 + c] + ((p0 + p1) + p2);
            }
        sa = sa + k;
        const W kd = (W)ktv;
        sw = sw + kd * kd;
        const double kk = (double)kd * ktv;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            su[c] = su[c] + kk * t[(size_t)s * 3 + c];
            sv[c] = sv[c] + kk * qtau[(size_t)m * 3 + c];
        }
    }
#pragma unroll
    for (int q = 0; q < 9; ++q) blk[(size_t)e * 9 + q] = M[q];
    a[e] = sa; w[e] = (double)sw;
#pragma unroll
    for (int c = 0; c < 3; ++c) { u[(size_t)e * 3 + c] = su[c]; v[(size_t)e * 3 + c] = sv[c]; }
}

// diagonal of J^T J at one node set: node j sums kd^2 over its source edges in source order (keys sorted stably by node)
template <typename W>
__global__ void merge_degree_kernel(long long n, int n_node, const uint32_t* __restrict__ key_sorted, const uint32_t* __restrict__ idx,
                                    const double* __restrict__ kt, double* __restrict__ deg) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n_node) return;
    long long lo = 0, hi = n;                                   // first sorted position with key >= j
    while (lo < hi) { const long long mid = (lo + hi) >> 1; if (key_sorted[mid] < (uint32_t)j) lo = mid + 1; else hi = mid; }
    W s = (W)0;
    for (long long i = lo; i < n && key_sorted[i] == (uint32_t)j; ++i) { const W kd = (W)kt[idx[i]]; s = s + kd * kd; }
    deg[j] = (double)s;
}

static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }
static inline int bits_for(unsigned long long v) { int b = 1; while (b < 64 && (v >> b)) ++b; return b; }

struct MergeWs {
    unsigned long long *key, *key2; uint32_t *kc, *kc2, *kt, *kt2, *idx, *i1, *i2, *i3; int32_t *head, *seg, *start; void* temp; size_t temp_bytes;
    size_t total;
};
static hipError_t merge_layout(long long n, int n_cam, int n_time, void* base, MergeWs& w) {
    size_t t1 = 0, t2 = 0, t3 = 0;
    hipError_t e = rocprim::radix_sort_pairs(nullptr, t1, (unsigned long long*)nullptr, (unsigned long long*)nullptr, (uint32_t*)nullptr,
                                             (uint32_t*)nullptr, (size_t)n, 0, bits_for((unsigned long long)n_time * n_cam), (hipStream_t)0);
    if (e != hipSuccess) return e;
    e = rocprim::radix_sort_pairs(nullptr, t2, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (uint32_t*)nullptr, (size_t)n, 0, 32,
                                  (hipStream_t)0);
    if (e != hipSuccess) return e;
    e = rocprim::inclusive_scan(nullptr, t3, (int32_t*)nullptr, (int32_t*)nullptr, (size_t)n, rocprim::plus<int32_t>(), (hipStream_t)0);
    if (e != hipSuccess) return e;
    w.temp_bytes = t1 > t2 ? t1 : t2; if (t3 > w.temp_bytes) w.temp_bytes = t3;
    size_t off = 0;
    auto take = [&](size_t bytes) { void* p = base ? (void*)((char*)base + off) : nullptr; off += al256(bytes); return p; };
    w.key = (unsigned long long*)take(8 * n); w.key2 = (unsigned long long*)take(8 * n);
    w.kc = (uint32_t*)take(4 * n); w.kc2 = (uint32_t*)take(4 * n); w.kt = (uint32_t*)take(4 * n); w.kt2 = (uint32_t*)take(4 * n);
    w.idx = (uint32_t*)take(4 * n); w.i1 = (uint32_t*)take(4 * n); w.i2 = (uint32_t*)take(4 * n); w.i3 = (uint32_t*)take(4 * n);
    w.head = (int32_t*)take(4 * n); w.seg = (int32_t*)take(4 * n); w.start = (int32_t*)take(4 * (n + 1));
    w.temp = take(w.temp_bytes);
    w.total = off;
    return hipSuccess;
}

extern "C" int64_t vican_merge_ws_bytes(int64_t n, int32_t n_cam, int32_t n_time) {
    if (n <= 0 || n_cam <= 0 || n_time <= 0) return 0;
    MergeWs w;
    if (merge_layout(n, n_cam, n_time, nullptr, w) != hipSuccess) return -1;
    return (int64_t)w.total;
}

extern "C" int vican_merge_edges(int64_t n, int32_t n_cam, int32_t n_time, int32_t n_marker, int32_t storage, const int32_t* cam,
                                 const int32_t* tim, const int32_t* marker, const double* R, const double* t, const double* kr,
                                 const uint8_t* kr_f32, const double* kt, const double* CmT, const double* qtau, void* ws, int64_t ws_bytes, int32_t* n_merged,
                                 int32_t* row_ptr, int32_t* col, double* blk, double* a, double* w, double* u, double* v, double* deg_c,
                                 double* deg_t, void* stream) {
    if (n <= 0 || n_cam <= 0 || n_time <= 0 || n_marker <= 0 || n > 0x7FFFFFF0LL || !cam || !tim || !marker || !R || !t || !kr || !kt ||
        !CmT || !qtau || !ws || !n_merged || !row_ptr || !col || !blk || !a || !w || !u || !v || !deg_c || !deg_t)
        return set_err(VICAN_ERR_ARG, "vican_merge_edges: bad argument");
    MergeWs m;
    if (merge_layout(n, n_cam, n_time, ws, m) != hipSuccess || (int64_t)m.total > ws_bytes)
        return set_err(VICAN_ERR_ARG, "vican_merge_edges: workspace too small (vican_merge_ws_bytes)");
    hipStream_t s = (hipStream_t)stream;
    const unsigned nb = (unsigned)((n + 255) / 256);
    hipLaunchKernelGGL(merge_keys_kernel, dim3(nb), dim3(256), 0, s, (long long)n, (int)n_cam, cam, tim, m.key, m.kc, m.kt, m.idx);
    size_t tb = m.temp_bytes;
    // stable sorts: equal keys keep source order
    if (rocprim::radix_sort_pairs(m.temp, tb, m.key, m.key2, m.idx, m.i1, (size_t)n, 0, bits_for((unsigned long long)n_time * n_cam), s) != hipSuccess ||
        rocprim::radix_sort_pairs(m.temp, tb, m.kc, m.kc2, m.idx, m.i2, (size_t)n, 0, bits_for((unsigned long long)n_cam), s) != hipSuccess ||
        rocprim::radix_sort_pairs(m.temp, tb, m.kt, m.kt2, m.idx, m.i3, (size_t)n, 0, bits_for((unsigned long long)n_time), s) != hipSuccess)
        return set_err(VICAN_ERR_LAUNCH, "vican_merge_edges: radix sort failed");
    hipLaunchKernelGGL(merge_heads_kernel, dim3(nb), dim3(256), 0, s, (long long)n, m.key2, m.head);
    if (rocprim::inclusive_scan(m.temp, tb, m.head, m.seg, (size_t)n, rocprim::plus<int32_t>(), s) != hipSuccess)
        return set_err(VICAN_ERR_LAUNCH, "vican_merge_edges: scan failed");
    hipLaunchKernelGGL(merge_starts_kernel, dim3(nb), dim3(256), 0, s, (long long)n, (int)n_cam, (int)n_time, m.key2, m.seg, m.start, n_merged,
                       row_ptr, col);
    if (storage == VICAN_STORE_F32) {
        hipLaunchKernelGGL(merge_segments_kernel<float>, dim3(nb), dim3(256), 0, s, (long long)n, n_merged, m.start, m.i1, marker, R, t, kr, kr_f32, kt,
                           CmT, qtau, blk, a, w, u, v);
        hipLaunchKernelGGL(merge_degree_kernel<float>, dim3((n_cam + 63) / 64), dim3(64), 0, s, (long long)n, (int)n_cam, m.kc2, m.i2, kt, deg_c);
        hipLaunchKernelGGL(merge_degree_kernel<float>, dim3((n_time + 63) / 64), dim3(64), 0, s, (long long)n, (int)n_time, m.kt2, m.i3, kt, deg_t);
    } else {
        hipLaunchKernelGGL(merge_segments_kernel<double>, dim3(nb), dim3(256), 0, s, (long long)n, n_merged, m.start, m.i1, marker, R, t, kr, kr_f32, kt,
                           CmT, qtau, blk, a, w, u, v);
        hipLaunchKernelGGL(merge_degree_kernel<double>, dim3((n_cam + 63) / 64), dim3(64), 0, s, (long long)n, (int)n_cam, m.kc2, m.i2, kt, deg_c);
        hipLaunchKernelGGL(merge_degree_kernel<double>, dim3((n_time + 63) / 64), dim3(64), 0, s, (long long)n, (int)n_time, m.kt2, m.i3, kt, deg_t);
    }
    LAUNCH_CHECK("vican_merge_edges");
    return VICAN_OK;
}
