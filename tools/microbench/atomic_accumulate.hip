// device-scope 64-bit integer atomics: W workgroups each add N values onto the SAME N addresses (the camera accumulators of a sweep)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void add_kernel(unsigned long long* acc, int n, int rep) {
    for (int r = 0; r < rep; ++r)
        for (int i = threadIdx.x; i < n; i += blockDim.x)
            __hip_atomic_fetch_add(acc + i, (unsigned long long)(blockIdx.x + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void slab_kernel(unsigned long long* slab, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) slab[(size_t)blockIdx.x * n + i] = blockIdx.x + 1;
}
int main() {
    for (int n : {3060, 6144, 9000, 18000}) {
        for (int W : {40, 256}) {
            unsigned long long *acc, *slab;
            hipMalloc(&acc, (size_t)n * 8); hipMemset(acc, 0, (size_t)n * 8);
            hipMalloc(&slab, (size_t)n * 8 * W);
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int t = 0; t < 3; ++t) hipLaunchKernelGGL(add_kernel, dim3(W), dim3(768), 0, 0, acc, n, 1);
            hipDeviceSynchronize();
            float ms_a = 0, ms_s = 0;
            hipEventRecord(e0); for (int t = 0; t < 50; ++t) hipLaunchKernelGGL(add_kernel, dim3(W), dim3(768), 0, 0, acc, n, 1); hipEventRecord(e1);
            hipEventSynchronize(e1); hipEventElapsedTime(&ms_a, e0, e1);
            hipEventRecord(e0); for (int t = 0; t < 50; ++t) hipLaunchKernelGGL(slab_kernel, dim3(W), dim3(768), 0, 0, slab, n); hipEventRecord(e1);
            hipEventSynchronize(e1); hipEventElapsedTime(&ms_s, e0, e1);
            std::vector<unsigned long long> h(n);
            hipMemcpy(h.data(), acc, (size_t)n * 8, hipMemcpyDeviceToHost);
            printf("n %6d W %3d: atomics %.2f us per launch, slab writes %.2f us per launch (sum check %s)\n", n, W, ms_a / 50 * 1e3, ms_s / 50 * 1e3,
                   h[0] == (unsigned long long)53 * W * (W + 1) / 2 ? "ok" : "BAD");
            hipFree(acc); hipFree(slab);
        }
    }
    return 0;
}
