// Microbenchmark: LDS atomic / read throughput on gfx950 (design input for the sweep kernel).
// hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics tools/lds_atomic_bench.hip -o /tmp/ldsb && /tmp/ldsb
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define N_ELEM 9000
template <typename T, int MODE, int STRIDE>
__global__ __launch_bounds__(1024) void k(T* out, const int* __restrict__ offs, int iters) {
    __shared__ T lds[N_ELEM * (sizeof(T) == 8 ? 1 : 2)];
    const int n = N_ELEM * (sizeof(T) == 8 ? 1 : 2);
    for (int i = threadIdx.x; i < n; i += 1024) lds[i] = 0;
    __syncthreads();
    int a = (threadIdx.x * STRIDE) % n;
    const int r = offs[threadIdx.x];
    T acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 9; ++u) {
            int idx = a + u * (MODE == 2 ? 1000 : 1);      // MODE 2: component-major (SoA) style
            if (idx >= n) idx -= n;
            if (MODE == 3) acc += lds[idx];
            else __hip_atomic_fetch_add(&lds[idx], (T)1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        a += r; if (a >= n) a -= n;
    }
    __syncthreads();
    if (MODE == 3) out[blockIdx.x * 1024 + threadIdx.x] = acc;
    else if (threadIdx.x == 0) out[blockIdx.x] = lds[5];
}
template <typename T, int MODE, int STRIDE>
void run(const char* name, int* d_offs) {
    T* out; hipMalloc(&out, 256 * 1024 * sizeof(T));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    hipLaunchKernelGGL((k<T, MODE, STRIDE>), dim3(256), dim3(1024), 0, 0, out, d_offs, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<T, MODE, STRIDE>), dim3(256), dim3(1024), 0, 0, out, d_offs, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double ops = 1024.0 * 9 * iters;                     // per CU
    printf("%-44s %8.3f ms  %7.2f lane-ops/ns/CU  (%.1f clk/wave-instr @2.4GHz)\n", name, ms, ops / (ms * 1e6),
           64.0 / (ops / (ms * 1e6) / 2.4));
    hipFree(out);
}
int main() {
    int h[1024]; srand(1);
    int* d; hipMalloc(&d, sizeof(h));
    for (int i = 0; i < 1024; ++i) h[i] = 9 * 64;         // all lanes advance together: pattern stays regular
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    run<unsigned long long, 0, 1>("u64 atomic, lane stride 1", d);
    run<unsigned long long, 0, 9>("u64 atomic, lane stride 9", d);
    run<unsigned long long, 0, 36>("u64 atomic, lane stride 36", d);
    run<unsigned int, 0, 1>("u32 atomic, lane stride 1", d);
    run<unsigned int, 0, 36>("u32 atomic, lane stride 36", d);
    run<double, 0, 1>("f64 atomic, lane stride 1 elem (no conflict)", d);
    run<double, 0, 9>("f64 atomic, lane stride 9 (AoS, consecutive cams)", d);
    run<double, 0, 36>("f64 atomic, lane stride 36 (AoS, every 4th cam)", d);
    run<double, 2, 4>("f64 atomic, SoA planes, lane stride 4", d);
    run<float, 0, 1>("f32 atomic, lane stride 1", d);
    run<float, 0, 9>("f32 atomic, lane stride 9", d);
    run<float, 0, 36>("f32 atomic, lane stride 36", d);
    run<float, 2, 4>("f32 atomic, SoA planes, lane stride 4", d);
    run<double, 3, 1>("f64 read, lane stride 1", d);
    run<double, 3, 9>("f64 read, lane stride 9", d);
    run<double, 3, 36>("f64 read, lane stride 36", d);
    run<float, 3, 1>("f32 read, lane stride 1", d);
    run<float, 3, 36>("f32 read, lane stride 36", d);
    for (int i = 0; i < 1024; ++i) h[i] = (rand() % 997) * 9;   // random cams per lane
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    run<unsigned long long, 0, 9>("u64 atomic, AoS, random cams", d);
    run<unsigned int, 0, 9>("u32 atomic, AoS, random cams", d);
    run<double, 0, 9>("f64 atomic, AoS, random cams", d);
    run<double, 2, 1>("f64 atomic, SoA, random cams", d);
    run<float, 0, 9>("f32 atomic, AoS, random cams", d);
    run<float, 2, 1>("f32 atomic, SoA, random cams", d);
    return 0;
}
