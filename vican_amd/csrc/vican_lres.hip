// vican_lres.hip - a run of block Lanczos steps (operator sweep + camera-side step, reference: the ARPACK iteration behind
// eigs(k=5, sigma=-1e-6) at bipgo.py:288) as ONE cooperative launch, for graphs that are small enough to be latency-bound.
//
// On a capture-sized graph (large_shop: 340 cameras, 10 000 timesteps, 40 000 merged edges) a Lanczos step of the
// multi-kernel path is a sweep launch (vican_block_op, 11.8 us for 2.4 MB) and a cooperative camera-side launch
// (vican_lanczos_cam_coop, 20 us) - 26 such pairs per solve, every one bound by launch latency, table staging and the
// re-reading of the basis.  Here the grid of the sweep stays resident for a whole run of steps j0 .. j1 - 1:
//   * every wavefront keeps the blocks, packed indices and dual rows of its first two chunks IN REGISTERS (a capture-sized
//     graph has one or two chunks per wavefront) - after the first step the sweep touches no global memory for the graph;
//   * the first ceil(C / 32) workgroups double as the camera side: their rows of the basis V stay in LDS and grow by one
//     block per step; the sweep tables (x planes, fixed-point z planes) are rebuilt per step, the camera side's scratch
//     overlays them;
//   * per step five grid barriers (relaxed agent-scope counter, tools/barrier_bench.hip: ~1 us at 40 workgroups) order
//     what crosses workgroups - the fixed-point z slabs, the Gram-Schmidt partials (twice), the 3x3 Gram of the new block,
//     the new block itself (the next sweep's input) - all of it written and read with agent-scope atomics.
// Arithmetic: the sweep is wave_sweep_kernel<MODE 0> (same products, same exact 64-bit fixed-point sums and scales - z is
// bit-identical), the camera side is lanczos_cam_coop_kernel (same slices of 32 cameras, same fixed summation orders).
#include "vican_sweep_common.h"

#define LR_THREADS 256
#define LR_NW 4
#define LR_CAMS 32
#define LR_ROWS (3 * LR_CAMS)
#define LR_STAGE 2048
#define LR_KA_MAX 192
#define LR_SCRATCH0 (8 * 9 * LR_CAMS > LR_STAGE ? 8 * 9 * LR_CAMS : LR_STAGE)     /* doubles: max(stage, zred) */

__device__ __forceinline__ void lr_st(double* p, double v) {
    __hip_atomic_store((unsigned long long*)p, (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ double lr_ld(const double* p) {
    return __longlong_as_double((long long)__hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void lr_st(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 lr_ld(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// grid barrier: vican_grid_sync (vican_common.h), fenced form, bounded spin (0.8 us per barrier on a capture-sized graph;
// see vican_cgres.hip)

// partial H = V[:, :ka]^T R over this workgroup's rows -> part[3 ka][ncw]: 8 lanes per element (strided rows, then a
// DPP sum over the 8 lanes) - the same order as coop_gram in vican_kernels.hip, so both paths agree to the bit
__device__ __forceinline__ double lr_dpp8_sum(double v) {
    v += __longlong_as_double((long long)dpp_u64<0xB1>((u64)__double_as_longlong(v)));
    v += __longlong_as_double((long long)dpp_u64<0x4E>((u64)__double_as_longlong(v)));
    v += __longlong_as_double((long long)dpp_u64<0x141>((u64)__double_as_longlong(v)));
    return v;
}
__device__ __forceinline__ void lr_gram(const double* __restrict__ vs, int ka, int nsl, const double (*rs)[LR_ROWS],
                                        double* __restrict__ part, int ncw, int wg) {
    const int seg = threadIdx.x & 7;
    for (int e = threadIdx.x >> 3; e < ka * 3; e += LR_THREADS / 8) {
        const int k = e / 3, c = e - 3 * k;
        const double* v = vs + k * LR_ROWS;
        double s = 0.0;
        for (int i = seg; i < nsl; i += 8) s += v[i] * rs[c][i];
        s = lr_dpp8_sum(s);
        if (seg == 0) lr_st(part + (size_t)e * ncw + wg, s);
    }
}
// out[t] = sum_w part[t][w] in a fixed order: windows of LR_STAGE values fetched by all threads at once, summed from LDS
__device__ __forceinline__ void lr_reduce(const double* __restrict__ part, int hs, int ncw, double* __restrict__ stage,
                                          double* __restrict__ out) {
    const int per = LR_STAGE / ncw;
    for (int t0 = 0; t0 < hs; t0 += per) {
        const int nt = hs - t0 < per ? hs - t0 : per, cnt = nt * ncw;
        double v[LR_STAGE / LR_THREADS];
#pragma unroll
        for (int m = 0; m < LR_STAGE / LR_THREADS; ++m) {
            const int e = threadIdx.x + m * LR_THREADS;
            v[m] = e < cnt ? lr_ld(part + (size_t)t0 * ncw + e) : 0.0;
        }
#pragma unroll
        for (int m = 0; m < LR_STAGE / LR_THREADS; ++m) stage[threadIdx.x + m * LR_THREADS] = v[m];
        __syncthreads();
        for (int t = threadIdx.x; t < nt; t += LR_THREADS) {
            double a = 0.0;
            for (int w = 0; w < ncw; ++w) a += stage[t * ncw + w];
            out[t0 + t] = a;
        }
        __syncthreads();
    }
}

extern "C" int64_t vican_lanczos_resident_lds_bytes(int32_t n_cam, int32_t max_rows, int32_t storage, int32_t n_copy, int32_t j1) {
    const int64_t s = ssize(storage), cp = plane_stride(n_cam);
    const int64_t per_wave = (((int64_t)max_rows * 9 * (8LL * n_copy + 8 + s)) + 15) & ~15LL;
    int64_t tables = 9LL * cp * (s + 8);
    const int64_t scratch = 8LL * (LR_SCRATCH0 + 9 * LR_CAMS);
    if (tables < scratch) tables = scratch;
    return tables + LR_NW * per_wave + 8LL * 3 * j1 * LR_ROWS + 256;
}
extern "C" int64_t vican_lanczos_resident_ws_doubles(int32_t n_cam) {
    const int64_t ncw = (n_cam + LR_CAMS - 1) / LR_CAMS;
    return ncw * (2LL * 3 * LR_KA_MAX + 8) + 8 + 12 * 64;    // (+ stamp area of diagnostic builds)
}

template <typename S, int EPL, int TRIPS>
struct LrChunk {
    ChunkRegs<S, EPL> r;
    double L[TRIPS][3];
    int r0, nrows;
};

template <typename S, int CP, int TRIPS>
__global__ __launch_bounds__(LR_THREADS) void lanczos_resident_kernel(
    const int32_t* __restrict__ gate, vican_graph_t g, const double* __restrict__ lamT_inv, const double* __restrict__ lamC, double* V,
    int ld, int j0, int j1, double* xrow, double* HB, int hb_stride, int hw, u64* zpart, double* ws, unsigned int* sync,
    const double* __restrict__ fx, double pivot_floor, uint32_t* abort_word, unsigned long long spin_limit) {
    GATE_RETURN(gate);
    constexpr int EPL = Vec<S>::N;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    __shared__ double s_xm[LR_NW];
    __shared__ double rs[3][LR_ROWS];
    __shared__ double h[LR_KA_MAX * 3], h2[LR_KA_MAX * 3];
    __shared__ double g6[4][6], G6s[6];
    const int C = g.n_cam, nx = 9 * CP, ncopy = g.n_copy, cmask = ncopy - 1, RW = g.max_rows;
    const int tid = threadIdx.x, lane = tid & 63, lane_copy = lane & cmask;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwg = (int)gridDim.x, wg = (int)blockIdx.x;
    // LDS map: [z planes | x planes] (the camera side's scratch overlays them) | per-wavefront row regions | basis rows
    u64* zs = (u64*)lds_raw;
    S* xs = (S*)(zs + nx);
    size_t tables = (size_t)nx * (8 + sizeof(S));
    const size_t scratch = 8 * (size_t)(LR_SCRATCH0 + 9 * LR_CAMS);
    if (tables < scratch) tables = scratch;
    double* stage = (double*)lds_raw;                          // (camera side, after the slab is written)
    long long* zred = (long long*)lds_raw;                     // [8][9 LR_CAMS] slab-group partials of the fold (before `stage` is used)
    double* zl = stage + LR_SCRATCH0;                          // [LR_CAMS][9] this workgroup's rows of z
    const size_t per_wave = (((size_t)RW * 9 * (8 * ncopy + 8 + sizeof(S))) + 15) & ~(size_t)15;
    unsigned char* wbase = lds_raw + tables + (size_t)wave * per_wave;
    u64* ys = (u64*)wbase;
    double* yv = (double*)(ys + (size_t)RW * 9 * ncopy);
    S* wv = (S*)(yv + (size_t)RW * 9);
    double* vs = (double*)(lds_raw + tables + (size_t)LR_NW * per_wave);      // [3 j1][LR_ROWS]
    const int ka_cap = 3 * j1;
    const uint32_t pad_cam = (uint32_t)((lane & 31) < C ? (lane & 31) : 0);

    // camera side: workgroups 0 .. ncw-1 own slices of <= 32 cameras
    const int ncw = (C + LR_CAMS - 1) / LR_CAMS;
    const bool is_cam = wg < ncw;
    const int cc0 = is_cam ? (int)(((long long)wg * C) / ncw) : 0, cc1 = is_cam ? (int)(((long long)(wg + 1) * C) / ncw) : 0;
    const int row0 = 3 * cc0, nsl = 3 * (cc1 - cc0);
    const int hs_cap = 3 * LR_KA_MAX;
    double* part1 = ws;
    double* part2 = ws + (size_t)ncw * hs_cap;
    double* partG = ws + (size_t)2 * ncw * hs_cap;
    unsigned int nbar = 0;
    const vican_sync_t sy = {sync, abort_word, spin_limit};
    auto gsync = [&]() -> bool { ++nbar; return vican_grid_sync(sy, nbar * (unsigned)nwg, true); };
#ifdef VICAN_LRSTAMP    /* diagnostic build: wall clock (100 MHz ticks) per phase and step of workgroup 0 -> ws tail [step][12] */
    unsigned long long st_t = __builtin_amdgcn_s_memrealtime();
    double* st_out = ws + (size_t)ncw * (2 * hs_cap + 8);
#define LSTAMP(i) do { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); \
                       if (wg == 0 && tid == 0 && j - j0 < 64) st_out[(j - j0) * 12 + (i)] = (double)(t_ - st_t); st_t = t_; __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define LSTAMP(i) do {} while (0)
#endif

    // this workgroup's chunks; the first two of every wavefront stay in registers
    const int c0 = (int)(((long long)wg * g.n_chunk) / nwg), c1 = (int)(((long long)(wg + 1) * g.n_chunk) / nwg);
    const int kmax = g.n_chunk - 1;
    auto fetch = [&](LrChunk<S, EPL, TRIPS>& ch, int kc) {
        kc = kc < kmax ? kc : kmax;
        ch.r0 = g.chunk_row0[kc]; ch.nrows = g.chunk_row0[kc + 1] - ch.r0;
        load_chunk<S, EPL>(ch.r, g, kc, lane);
#pragma unroll
        for (int t = 0; t < TRIPS; ++t) {
            int j = lane + 64 * t;
            j = j < ch.nrows * 3 ? j : 0;
            const double* Lp = lamT_inv + (size_t)ch.r0 * 9 + (size_t)j * 3;
            ch.L[t][0] = Lp[0]; ch.L[t][1] = Lp[1]; ch.L[t][2] = Lp[2];
        }
    };
    LrChunk<S, EPL, TRIPS> ca, cb;
    fetch(ca, c0 + wave);
    fetch(cb, c0 + wave + LR_NW);

    const double fx0 = fx[0], fx1 = fx[1], fx2 = fx[2], fx3 = fx[3], fx8 = fx[8];
    for (int i = lane; i < 9 * RW * ncopy; i += 64) ys[i] = 0ull;
    // basis rows of this camera slice: columns 0 .. 3 (j0 + 1) - 1 were written by earlier launches
    if (is_cam) {
        const int ka0 = 3 * (j0 + 1) < ka_cap ? 3 * (j0 + 1) : ka_cap;
        for (int t = tid; t < ka0 * LR_ROWS; t += LR_THREADS) {
            const int k = t / LR_ROWS, i = t - k * LR_ROWS;
            vs[t] = i < nsl ? V[(size_t)k * ld + row0 + i] : 0.0;
        }
    }

    for (int j = j0; j < j1; ++j) {
        const int ka = 3 * (j + 1), hs = 3 * ka;
        // ---- stage the sweep input (rows written by the camera workgroups at the end of the previous step)
        constexpr int XC = (CP + LR_THREADS - 1) / LR_THREADS;
        double xv[XC][9];
#pragma unroll
        for (int m = 0; m < XC; ++m) {
            const int c = tid + m * LR_THREADS;
#pragma unroll
            for (int i = 0; i < 9; ++i) xv[m][i] = c < C ? lr_ld(xrow + (size_t)c * 9 + i) : 0.0;    // (never a plain load: a line
                                                                     // cached in this XCD's L2 would go stale when another XCD rewrites it)
        }
        double xm2 = 0.0;
#pragma unroll
        for (int m = 0; m < XC; ++m) {
            double q = 0.0;
#pragma unroll
            for (int i = 0; i < 9; ++i) q += xv[m][i] * xv[m][i];
            xm2 = fmax(xm2, q);
        }
#pragma unroll
        for (int o2 = 32; o2 > 0; o2 >>= 1) xm2 = fmax(xm2, __shfl_xor(xm2, o2, 64));
        __syncthreads();                                       // (the camera side's overlay of the previous step is done)
        if (lane == 0) s_xm[wave] = xm2;
#pragma unroll
        for (int m = 0; m < XC; ++m) {
            const int c = tid + m * LR_THREADS;
            if (c < C) {
#pragma unroll
                for (int i = 0; i < 9; ++i) xs[i * CP + c] = pre_scale<S>(xv[m][i], 1.0);
            }
        }
        for (int i = tid; i < nx; i += LR_THREADS) zs[i] = 0ull;
        __syncthreads();
        xm2 = fmax(fmax(s_xm[0], s_xm[1]), fmax(s_xm[2], s_xm[3]));
        int shift = 0;
        if (xm2 > 0.0) { const double r2 = fx8 * fx8 / xm2; shift = r2 >= 1.0 ? (ilogb(r2) >> 1) : 0; }
        shift = shift < 0 ? 0 : (shift > 40 ? 40 : shift);
        const double up = ldexp(1.0, shift);
        const double y_scale = fx0 * up, y_inv = fx1 / up, z_scale = fx2 * up, z_conv = fx3 / up;
        LSTAMP(0);

        // ---- sweep: the three phases of wave_sweep_kernel<MODE 0> on one chunk
        auto chunk = [&](const LrChunk<S, EPL, TRIPS>& ch) {
            const ChunkRegs<S, EPL>& cur = ch.r;
            const int nrows = ch.nrows;
            uint32_t cam[EPL], row[EPL];
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                const bool pad = cur.id[e] == VICAN_PAD_SLOT;
                cam[e] = pad ? pad_cam : (cur.id[e] & 0xFFFFu); row[e] = pad ? 0u : (cur.id[e] >> 16);
            }
            {
                S acc[9], xc[9], xn[9];
#pragma unroll
                for (int q = 0; q < 9; ++q) xc[q] = xs[q * CP + cam[0]];
#pragma unroll
                for (int e = 0; e < EPL; ++e) {
                    if (e + 1 < EPL) {
#pragma unroll
                        for (int q = 0; q < 9; ++q) xn[q] = xs[q * CP + cam[e + 1]];
                    }
                    const bool cont = e > 0 && row[e] == row[e - 1];
#pragma unroll
                    for (int a = 0; a < 3; ++a)
#pragma unroll
                        for (int b = 0; b < 3; ++b) {
                            const S c = dot3<S>(vget<S>(cur.m[0 + a], e), xc[b], vget<S>(cur.m[3 + a], e), xc[3 + b],
                                                vget<S>(cur.m[6 + a], e), xc[6 + b]);
                            acc[a * 3 + b] = cont ? acc[a * 3 + b] + c : c;
                        }
                    const bool last = (e == EPL - 1) || row[e + 1 < EPL ? e + 1 : e] != row[e];
                    if (last) {
                        u64* yr = ys + (size_t)(row[e] * 9) * ncopy + lane_copy;
#pragma unroll
                        for (int q = 0; q < 9; ++q) lds_add_fix(yr + q * ncopy, fix_of<S>(acc[q], y_scale));
                    }
                    if (e + 1 < EPL) {
#pragma unroll
                        for (int q = 0; q < 9; ++q) xc[q] = xn[q];
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            for (int i = lane; i < nrows * 9; i += 64) {
                const int oo = i % 9;
                long long s = 0;
                for (int c = 0; c < ncopy; ++c) {
                    const int a = i * ncopy + ((c + oo) & cmask);
                    s += (long long)ys[a];
                    ys[a] = 0ull;
                }
                yv[i] = (double)fix_total<S>(s) * y_inv;
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int t = 0; t < TRIPS; ++t) {
                const int i = lane + 64 * t;
                if (i < nrows * 3) {
                    const double* yr = yv + (i / 3) * 9;
#pragma unroll
                    for (int b3 = 0; b3 < 3; ++b3)
                        wv[i * 3 + b3] = pre_scale<S>(dot3<double>(ch.L[t][0], yr[b3], ch.L[t][1], yr[3 + b3], ch.L[t][2], yr[6 + b3]), z_scale);
                }
            }
            __builtin_amdgcn_wave_barrier();
            S w[9];
            uint32_t prow = 0xFFFFFFFFu;
#pragma unroll
            for (int e = 0; e < EPL; ++e) {
                if (row[e] != prow) {
                    prow = row[e];
#pragma unroll
                    for (int q = 0; q < 9; ++q) w[q] = wv[row[e] * 9 + q];
                }
                u64* zc = zs + cam[e];
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int b = 0; b < 3; ++b) {
                        const S v = dot3<S>(vget<S>(cur.m[i * 3 + 0], e), w[b], vget<S>(cur.m[i * 3 + 1], e), w[3 + b],
                                            vget<S>(cur.m[i * 3 + 2], e), w[6 + b]);
                        lds_add_fix(&zc[(i * 3 + b) * CP], fix_of<S>(v, z_scale));
                    }
            }
            __builtin_amdgcn_wave_barrier();
        };
        {
            int kc = c0 + wave;
            if (kc < c1) chunk(ca);
            kc += LR_NW;
            if (kc < c1) chunk(cb);
            for (kc += LR_NW; kc < c1; kc += LR_NW) { LrChunk<S, EPL, TRIPS> ch; fetch(ch, kc); chunk(ch); }
        }
        __syncthreads();
        LSTAMP(1);
        // ---- this workgroup's z slab [9][C] (the totals: exact integers)
        {
            u64* zp = zpart + (size_t)wg * 9 * C;
#pragma unroll
            for (int q = 0; q < 9; ++q)
                for (int c = tid; c < C; c += LR_THREADS) lr_st(zp + q * C + c, (u64)fix_total<S>((long long)zs[q * CP + c]));
        }
        LSTAMP(2);
        if (!gsync()) return;
        LSTAMP(3);

        // ---- camera side (lanczos_cam_coop_kernel on the first ncw workgroups; its scratch overlays the sweep tables)
        if (is_cam) {
            // z of the own cameras from all slabs (exact integer sums): 8 slab groups x 32 cameras, the nine components of a
            // slab as independent loads (agent-scope loads of one thread are not overlapped across loop iterations)
            {
                const int cl = tid & 31, grp = tid >> 5, ncl = cc1 - cc0;
                long long acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
                if (cl < ncl)
                    for (int s = grp; s < nwg; s += 8) {
                        const u64* sp = zpart + (size_t)s * 9 * C + cc0 + cl;
                        u64 v[9];
#pragma unroll
                        for (int q = 0; q < 9; ++q) v[q] = lr_ld(sp + (size_t)q * C);
#pragma unroll
                        for (int q = 0; q < 9; ++q) acc[q] += (long long)v[q];
                    }
#pragma unroll
                for (int q = 0; q < 9; ++q) zred[grp * (9 * LR_CAMS) + q * LR_CAMS + cl] = acc[q];
                __syncthreads();
                for (int t = tid; t < 9 * LR_CAMS; t += LR_THREADS) {
                    long long sum = 0;
#pragma unroll
                    for (int k = 0; k < 8; ++k) sum += zred[k * (9 * LR_CAMS) + t];
                    const int q = t / LR_CAMS, c2 = t - q * LR_CAMS;
                    zl[c2 * 9 + q] = (double)sum * z_conv;
                }
            }
            __syncthreads();
            // A Q_j = Lambda_C Q_j - z on this slice
            if (tid < cc1 - cc0) {
                const int c = cc0 + tid;
                double L[9], q[9];
#pragma unroll
                for (int k = 0; k < 9; ++k) L[k] = lamC[(size_t)c * 9 + k];
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int b = 0; b < 3; ++b) q[i * 3 + b] = vs[(3 * j + b) * LR_ROWS + 3 * tid + i];
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int b = 0; b < 3; ++b)
                        rs[b][3 * tid + i] = L[i * 3] * q[b] + L[i * 3 + 1] * q[3 + b] + L[i * 3 + 2] * q[6 + b] - zl[tid * 9 + i * 3 + b];
            }
            __syncthreads();
            lr_gram(vs, ka, nsl, rs, part1, ncw, wg);
        }
        LSTAMP(4);
        if (!gsync()) return;
        LSTAMP(5);
        if (is_cam) {
            lr_reduce(part1, hs, ncw, stage, h);
            if (tid < nsl) {
                double r0 = rs[0][tid], r1 = rs[1][tid], r2 = rs[2][tid];
                for (int k = 0; k < ka; ++k) {
                    const double v = vs[k * LR_ROWS + tid];
                    r0 -= v * h[k * 3]; r1 -= v * h[k * 3 + 1]; r2 -= v * h[k * 3 + 2];
                }
                rs[0][tid] = r0; rs[1][tid] = r1; rs[2][tid] = r2;
            }
            __syncthreads();
            lr_gram(vs, ka, nsl, rs, part2, ncw, wg);
        }
        LSTAMP(6);
        if (!gsync()) return;
        LSTAMP(7);
        if (is_cam) {
            lr_reduce(part2, hs, ncw, stage, h2);
            if (tid < nsl) {
                double r0 = rs[0][tid], r1 = rs[1][tid], r2 = rs[2][tid];
                for (int k = 0; k < ka; ++k) {
                    const double v = vs[k * LR_ROWS + tid];
                    r0 -= v * h2[k * 3]; r1 -= v * h2[k * 3 + 1]; r2 -= v * h2[k * 3 + 2];
                }
                rs[0][tid] = r0; rs[1][tid] = r1; rs[2][tid] = r2;
            }
            __syncthreads();
            if (wg == 0) for (int t = tid; t < hs; t += LR_THREADS) HB[(size_t)j * hb_stride + t] = h[t] + h2[t];
            double gq[6] = {0, 0, 0, 0, 0, 0};
            if (tid < nsl) {
                const double r0 = rs[0][tid], r1 = rs[1][tid], r2 = rs[2][tid];
                gq[0] = r0 * r0; gq[1] = r0 * r1; gq[2] = r0 * r2; gq[3] = r1 * r1; gq[4] = r1 * r2; gq[5] = r2 * r2;
            }
#pragma unroll
            for (int q = 0; q < 6; ++q) gq[q] = wave_sum(gq[q]);
            if ((tid & 63) == 0)
#pragma unroll
                for (int q = 0; q < 6; ++q) g6[tid >> 6][q] = gq[q];
            __syncthreads();
            if (tid < 6) lr_st(partG + (size_t)tid * ncw + wg, (g6[0][tid] + g6[1][tid]) + (g6[2][tid] + g6[3][tid]));
        }
        LSTAMP(8);
        if (!gsync()) return;
        LSTAMP(9);
        if (is_cam) {
            lr_reduce(partG, 6, ncw, stage, G6s);
            // upper Cholesky G = beta^T beta, Q = R beta^-1 (pivot rule of chol_qr3_kernel)
            const double g00 = G6s[0], g01 = G6s[1], g02 = G6s[2], g11 = G6s[3], g12 = G6s[4], g22 = G6s[5];
            const double tr = g00 + g11 + g22, floor_ = fmax(1e-28 * tr, pivot_floor);
            double b00 = 0, b01 = 0, b02 = 0, b11 = 0, b12 = 0, b22 = 0, i00 = 0, i11 = 0, i22 = 0;
            if (g00 > floor_) { b00 = sqrt(g00); i00 = 1.0 / b00; b01 = g01 * i00; b02 = g02 * i00; }
            const double d11 = g11 - b01 * b01;
            if (d11 > floor_) { b11 = sqrt(d11); i11 = 1.0 / b11; b12 = (g12 - b01 * b02) * i11; }
            const double d22 = g22 - b02 * b02 - b12 * b12;
            if (d22 > floor_) { b22 = sqrt(d22); i22 = 1.0 / b22; }
            if (wg == 0 && tid == 0) {
                double* bo = HB + (size_t)j * hb_stride + hw;
                bo[0] = b00; bo[1] = b01; bo[2] = b02; bo[3] = 0; bo[4] = b11; bo[5] = b12; bo[6] = 0; bo[7] = 0; bo[8] = b22;
            }
            if (tid < nsl) {
                const int col0 = 3 * (j + 1), i = row0 + tid;
                const double q0 = rs[0][tid] * i00;
                const double q1 = (rs[1][tid] - q0 * b01) * i11;
                const double q2 = (rs[2][tid] - q0 * b02 - q1 * b12) * i22;
                V[(size_t)col0 * ld + i] = q0; V[(size_t)(col0 + 1) * ld + i] = q1; V[(size_t)(col0 + 2) * ld + i] = q2;
                if (col0 + 2 < ka_cap) { vs[col0 * LR_ROWS + tid] = q0; vs[(col0 + 1) * LR_ROWS + tid] = q1; vs[(col0 + 2) * LR_ROWS + tid] = q2; }
                lr_st(xrow + (size_t)i * 3, q0); lr_st(xrow + (size_t)i * 3 + 1, q1); lr_st(xrow + (size_t)i * 3 + 2, q2);
            }
        }
        LSTAMP(10);
        if (!gsync()) return;
        LSTAMP(11);
    }
    // the last workgroup out re-arms the barrier counter for the next launch
    __syncthreads();
    if (tid == 0 && __hip_atomic_fetch_add(&sync[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)nwg - 1u) {
        __hip_atomic_store(&sync[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&sync[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// Lanczos steps j0 .. j1 - 1 in one launch.  Preconditions as for vican_block_op (fx finished for this operator) and
// vican_lanczos_cam_coop: V [3 (m + 1)][ld] column-major basis with blocks 0 .. j0 filled, xrow [3C][3] = block j0 row-major
// (in) / block j1 (out), HB [m][hb_stride]: row j receives the projected column (3 * 3 (j + 1) doubles) and, from offset hw,
// beta_j (9 doubles) - the layout vican_ritz reads.  zpart: n_wg * 9C 64-bit words; ws: vican_lanczos_resident_ws_doubles()
// doubles; sync_ws: two zeroed 32-bit words (the barrier counters - the same pair vican_lanczos_cam_coop uses, re-armed by
// vican_lanczos_seed at the start of every eigen-solve).  Honours the launch gate (vican_set_gate).
extern "C" int vican_lanczos_resident(const vican_graph_t* g, const double* lamT_inv, const double* lamC, double* V, int32_t ld,
                                      int32_t j0, int32_t j1, double* xrow, double* HB, int32_t hb_stride, int32_t hw, void* zpart,
                                      double* ws, uint32_t* sync_ws, const double* fx, double pivot_floor, void* stream) {
    if (int rc = vican_check_graph(g, "vican_lanczos_resident")) return rc;
    if (g->layout != VICAN_LAYOUT_WAVE || !g->blk) return set_err(VICAN_ERR_ARG, "vican_lanczos_resident: wave layout with block planes only");
    if (!lamT_inv || !lamC || !V || !xrow || !HB || !zpart || !ws || !sync_ws || !fx || j0 < 0 || j1 <= j0 || 3 * j1 > LR_KA_MAX ||
        ld < 3 * g->n_cam || hw < 9 * j1 || hb_stride < hw + 9)
        return set_err(VICAN_ERR_ARG, "vican_lanczos_resident: bad argument");
    if (g->n_chunk == 0) return set_err(VICAN_ERR_ARG, "vican_lanczos_resident: graph without edges");
    int dev = 0, n_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return set_err(VICAN_ERR_LAUNCH, "vican_lanczos_resident: cannot query the device");
    const int ncw = (g->n_cam + LR_CAMS - 1) / LR_CAMS;
    if (g->n_wg > n_cu || ncw > g->n_wg || g->n_cam > 512)
        return set_err(VICAN_ERR_CAPACITY, "%s: the grid must be co-resident, hold ceil(C / 32) camera workgroups, and C <= 512", "vican_lanczos_resident");
    const size_t lds = (size_t)vican_lanczos_resident_lds_bytes(g->n_cam, g->max_rows, g->storage, g->n_copy, j1);
    if (lds + 14 * 1024 > 160 * 1024) return set_err(VICAN_ERR_CAPACITY, "vican_lanczos_resident: tables + basis rows do not fit in LDS");
    const int trips = (3 * g->max_rows + 63) / 64;
    if (trips > 3) return set_err(VICAN_ERR_CAPACITY, "vican_lanczos_resident: more than 64 rows per chunk");
    hipStream_t s = (hipStream_t)stream;
#define LR_LAUNCH(S_, CP_, T_)                                                                                            \
    do {                                                                                                                  \
        auto kern = lanczos_resident_kernel<S_, CP_, T_>;                                                                 \
        static size_t conf = 0;                                                                                           \
        if (lds > conf) {                                                                                                 \
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
                return set_err(VICAN_ERR_LAUNCH, "vican_lanczos_resident: cannot raise dynamic LDS limit");               \
            conf = lds;                                                                                                   \
        }                                                                                                                 \
        if (int rc_ = vican_coresident_ok((const void*)kern, LR_THREADS, lds, g->n_wg, "vican_lanczos_resident")) return rc_; \
        hipLaunchKernelGGL(kern, dim3(g->n_wg), dim3(LR_THREADS), lds, s, g_vican_gate, *g, lamT_inv, lamC, V, (int)ld, (int)j0, \
                           (int)j1, xrow, HB, (int)hb_stride, (int)hw, (u64*)zpart, ws, sync_ws, fx, pivot_floor,         \
                           g_vican_abort_word, g_vican_sync_ticks);                                                       \
    } while (0)
#define LR_PICK2(S_, CP_) do { if (trips <= 1) LR_LAUNCH(S_, CP_, 1); else if (trips == 2) LR_LAUNCH(S_, CP_, 2); else LR_LAUNCH(S_, CP_, 3); } while (0)
#define LR_PICK(S_) do { if (g->n_cam <= 256) LR_PICK2(S_, 256); else LR_PICK2(S_, 512); } while (0)
    if (g->storage == VICAN_STORE_F32) LR_PICK(float); else LR_PICK(double);
#undef LR_PICK
#undef LR_PICK2
#undef LR_LAUNCH
    LAUNCH_CHECK("vican_lanczos_resident");
    return VICAN_OK;
}
