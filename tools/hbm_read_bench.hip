// Pure HBM read-rate microbenchmark (design input): grid-stride float4 reads of a 1 GB buffer
// with several launch shapes, to compare with the sweep's persistent 256 x 768-thread pattern.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int UNROLL>
__global__ void rd(const float4* __restrict__ p, size_t n, float* out) {
    float acc = 0.f;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, stride = (size_t)gridDim.x * blockDim.x;
    for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        float4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = p[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
    }
    if (acc == 123.456f) out[0] = acc;
}
// persistent: each block walks contiguous 120 KB chunks, 10 x 16 B per thread per chunk (like the sweep)
__global__ void rd_chunks(const float4* __restrict__ p, int nchunk, int f4_per_chunk, float* out) {
    float acc = 0.f;
    const int k0 = (int)((long long)blockIdx.x * nchunk / gridDim.x), k1 = (int)((long long)(blockIdx.x + 1) * nchunk / gridDim.x);
    for (int k = k0; k < k1; ++k) {
        const float4* c = p + (size_t)k * f4_per_chunk;
        float4 v[10];
#pragma unroll
        for (int u = 0; u < 10; ++u) v[u] = c[u * blockDim.x + threadIdx.x];
#pragma unroll
        for (int u = 0; u < 10; ++u) acc += v[u].x + v[u].w;
    }
    if (acc == 123.456f) out[0] = acc;
}
int main() {
    const size_t bytes = 1ull << 30, n = bytes / 16;
    float4* p; float* out; hipMalloc(&p, bytes); hipMalloc(&out, 4); hipMemset(p, 1, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto report = [&](const char* name, float ms) { printf("%-46s %7.3f ms  %6.0f GB/s\n", name, ms, bytes / ms / 1e6); };
    float ms;
#define RUN(name, ...) do { __VA_ARGS__; hipEventRecord(e0); for (int r = 0; r < 5; ++r) { __VA_ARGS__; } hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1); report(name, ms / 5); } while (0)
    RUN("grid 2048x256 unroll4", hipLaunchKernelGGL(rd<4>, dim3(2048), dim3(256), 0, 0, p, n, out));
    RUN("grid 2048x256 unroll8", hipLaunchKernelGGL(rd<8>, dim3(2048), dim3(256), 0, 0, p, n, out));
    RUN("grid 1024x512 unroll8", hipLaunchKernelGGL(rd<8>, dim3(1024), dim3(512), 0, 0, p, n, out));
    RUN("grid 256x1024 unroll8", hipLaunchKernelGGL(rd<8>, dim3(256), dim3(1024), 0, 0, p, n, out));
    RUN("grid 256x768 unroll10", hipLaunchKernelGGL(rd<10>, dim3(256), dim3(768), 0, 0, p, n, out));
    RUN("grid 512x768 unroll10", hipLaunchKernelGGL(rd<10>, dim3(512), dim3(768), 0, 0, p, n, out));
    { const int f4 = 768 * 10; const int nch = (int)(n / f4);
      RUN("persistent 256x768, 120KB chunks", hipLaunchKernelGGL(rd_chunks, dim3(256), dim3(768), 0, 0, p, nch, f4, out));
      RUN("persistent 512x768, 120KB chunks", hipLaunchKernelGGL(rd_chunks, dim3(512), dim3(768), 0, 0, p, nch, f4, out));
      RUN("persistent 768x256 (3/CU), 40KB chunks", hipLaunchKernelGGL(rd_chunks, dim3(768), dim3(256), 0, 0, p, (int)(n / 2560), 2560, out)); }
    return 0;
}
