// Latency of a device-side grid barrier (relaxed agent-scope counter, as coop_grid_sync / cgr_grid_sync) on this chip:
//   hipcc --offload-arch=gfx950 -O3 tools/barrier_bench.hip -o tools/barb && tools/barb
// mode 0: nwg workgroups, one per blockIdx (spread over the 8 XCDs round-robin)
// mode 1: 8 nwg workgroups launched, only those with blockIdx % 8 == 0 take part (all on one XCD)
// mode 2: as 0, plus every workgroup publishes a double per round and reads all of them back (the gather of a reduction)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ void gsync(unsigned* c, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(c, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}
__global__ void k(unsigned* c, double* part, double* out, int rounds, int mode, int nwg) {
    int wg = blockIdx.x;
    if (mode == 1) { if (blockIdx.x & 7) return; wg = blockIdx.x >> 3; }
    double acc = 0.0;
    for (int r = 0; r < rounds; ++r) {
        if (mode == 2) {
            if (threadIdx.x == 0) __hip_atomic_store(part + (r & 1) * 1024 + wg, (double)(r + wg), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        gsync(c, (unsigned)(r + 1) * nwg);
        if (mode == 2) {
            double v = threadIdx.x < nwg ? __hip_atomic_load(part + (r & 1) * 1024 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0.0;
            for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
            acc += v;
        }
    }
    if (threadIdx.x == 0) out[wg] = acc;
}
int main() {
    unsigned* c; double *part, *out;
    hipMalloc(&c, 64); hipMalloc(&part, 2048 * 8); hipMalloc(&out, 4096 * 8);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int rounds = 2000;
    for (int mode = 0; mode < 3; ++mode)
        for (int nwg : {1, 8, 16, 32, 40, 64, 128, 256}) {
            if (mode == 1 && nwg > 32) continue;
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                hipMemset(c, 0, 64);
                hipEventRecord(a);
                hipLaunchKernelGGL(k, dim3(mode == 1 ? 8 * nwg : nwg), dim3(256), 0, 0, c, part, out, rounds, mode, nwg);
                hipEventRecord(b); hipEventSynchronize(b);
                float ms; hipEventElapsedTime(&ms, a, b); if (ms < best) best = ms;
            }
            printf("mode %d nwg %3d: %.2f us per barrier\n", mode, nwg, best * 1e3 / rounds);
        }
    return 0;
}
