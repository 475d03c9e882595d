// Microbenchmark (design input for the translation arrays' layout): one wavefront per 256-slot chunk reads 8-byte-per-slot
// data (w; u, v planes) as two 16-byte loads per lane.  (A) lane l reads bytes [32 l, 32 l + 16) and [32 l + 16, 32 l + 32):
// every load instruction covers 2 KB at 50 % density (what slots 4l .. 4l+3 of a plane give); (B) the same bytes permuted so
// that instruction h reads [1024 h + 16 l, +16): dense.  Non-temporal and plain loads; 12 wavefronts per workgroup, chunks
// of one wavefront NW apart like cg_wsweep_kernel.
// hipcc --offload-arch=gfx950 -O3 tools/stride_read_bench.hip -o /tmp/srb && /tmp/srb
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v2d __attribute__((ext_vector_type(2)));
template <bool DENSE, bool NT, int PLANES>
__global__ __launch_bounds__(768) void rd(const double* __restrict__ p, int n_chunk, double* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c0 = (int)((long long)blockIdx.x * n_chunk / gridDim.x), c1 = (int)((long long)(blockIdx.x + 1) * n_chunk / gridDim.x);
    double acc = 0;
    for (int k = c0 + wave; k < c1; k += 12) {
        const double* c = p + (size_t)k * 256 * PLANES;
        v2d v[2 * PLANES];
#pragma unroll
        for (int q = 0; q < PLANES; ++q)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const v2d* a = (const v2d*)(c + q * 256 + (DENSE ? h * 128 + lane * 2 : lane * 4 + h * 2));
                v[2 * q + h] = NT ? __builtin_nontemporal_load(a) : *a;
            }
#pragma unroll
        for (int i = 0; i < 2 * PLANES; ++i) acc += v[i].x + v[i].y;
    }
    if (acc == 123.456) out[0] = acc;
}
int main() {
    const size_t bytes = 3ull << 29;                    // 1.5 GB
    double *p, *out; hipMalloc(&p, bytes); hipMalloc(&out, 8); hipMemset(p, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms;
#define RUN(name, D, N, P) do { const int nch = (int)(bytes / (256 * 8 * P)); \
        hipLaunchKernelGGL((rd<D, N, P>), dim3(256), dim3(768), 0, 0, p, nch, out); hipEventRecord(e0); \
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((rd<D, N, P>), dim3(256), dim3(768), 0, 0, p, nch, out); \
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); \
        printf("%-58s %7.3f ms  %6.0f GB/s\n", name, ms / 5, bytes / (ms / 5) / 1e6); } while (0)
    RUN("1 plane  (w),   half-dense instructions, plain", false, false, 1);
    RUN("1 plane  (w),   dense instructions,      plain", true, false, 1);
    RUN("1 plane  (w),   half-dense instructions, nt", false, true, 1);
    RUN("1 plane  (w),   dense instructions,      nt", true, true, 1);
    RUN("6 planes (u,v), half-dense instructions, plain", false, false, 6);
    RUN("6 planes (u,v), dense instructions,      plain", true, false, 6);
    RUN("6 planes (u,v), half-dense instructions, nt", false, true, 6);
    RUN("6 planes (u,v), dense instructions,      nt", true, true, 6);
    return 0;
}
