// vican_tsweep.hip - the operator z = sum_t M_.t Lambda_t^-1 (sum_c M_ct^T x_c) on CAMERA-TILED graphs (more cameras than one
// LDS table holds; the reference has no camera limit, bipgo.py:225-232) in ONE launch that reads every block ONCE.
//
// The untiled sweep keeps a chunk's blocks in registers from phase 1 (row sums) to phase 3 (camera sums).  With camera
// tiles the row sums need ALL tiles before any tile can run phase 3, so rounds 2-4 made two passes over the blocks (a rows
// pass and a camera pass per tile: vican_tile_rows / vican_tile_cams).  Here the tiles share their chunking (chunk k covers
// the same timestep rows in every tile: vican_plan_chunks_multi), a workgroup is bound to one tile, and the wavefront that
// handles chunk k of its tile
//     phase 1    forms its tile's share of the chunk's row sums and PUBLISHES it (72 B per row, write-through stores),
//     ...        goes on with phase 1 of its NEXT chunk (the blocks of chunk k stay in registers: TS_SETS register sets
//                rotate - waiting for phase 3, in phase 1, prefetch in flight [, with four: arrived for the next phase 1]),
//     phase 2/3  one iteration later reads the shares of ALL tiles for chunk k's rows, applies Lambda_t^-1 and runs
//                phase 3 on the blocks it still holds.
// Exchange without counters, fences or atomics: the shares live in two buffers used by alternate launches; a wavefront
// that publishes rows into this launch's buffer overwrites the same rows of the OTHER buffer with a sentinel (a NaN with a
// payload no arithmetic produces), so at the next launch every value of its buffer is either the sentinel or new.  A
// reader simply reloads until none of its values is the sentinel - each value validates itself, no ordering between
// stores is needed, and in the common case (the partner wavefronts of the other tiles run the same static schedule) the
// values have been there for a whole iteration.  Progress: every wavefront publishes chunk k BEFORE it waits for chunk
// k - 1's shares, so the wavefront owning the globally oldest unpublished chunk is never blocked; the grid must be
// co-resident (checked at launch: one workgroup per compute unit) and every spin is bounded (vican_set_barrier_abort).
#include "vican_sweep_common.h"

#ifndef TS_NW
#define TS_NW 8
#endif
#ifndef TS_SETS
#define TS_SETS 3                     /* register sets per wavefront: 3 = prefetch one iteration ahead, 4 = two (measured on the wide
                                         benchmark: 227.5 us with 3, 234.9 us with 4 - every wavefront ends with TS_SETS - 2 clamped loads) */
#endif
#define TS_PRE 4                      /* tiles whose shares are requested ahead of phase 1 (more tiles: the reload loop) */
#define TS_SENTINEL 0x7FF8C0DEFACE0001ull     /* quiet NaN, payload never produced by arithmetic */

// Pointers that reach the kernel through the descriptor array in memory are GENERIC pointers to the compiler (it promotes
// pointers loaded from a kernel argument to the global address space only if nothing in the kernel may have overwritten
// them): every access became a FLAT instruction, whose results may return out of order with respect to global and LDS
// operations - the waitcnt pass then waits with vmcnt(0) / lgkmcnt(0) everywhere and the prefetch pipeline is gone.  The
// round trip through address space 1 tells it what they are.
#define TS_G __attribute__((address_space(1)))
__device__ __forceinline__ void ts_st(TS_G double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double ts_ld(TS_G const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
// load_chunk16 (vican_sweep_common.h) on address-space-1 pointers
template <typename S, int EPL, bool NT>
__device__ __forceinline__ void ts_load_chunk(ChunkRegs<S, EPL>& c, TS_G const S* blk, TS_G const uint16_t* idx16, const int slots, const int k, const int lane) {
    const size_t pbase = (size_t)k * 9 * slots + (size_t)lane * EPL;
    TS_G const uint16_t* ip = idx16 + (size_t)k * slots + (size_t)lane * EPL;
    uint32_t lo, hi = 0;
    if constexpr (sizeof(S) == 4) {
        typedef float v4f __attribute__((ext_vector_type(4)));
        typedef unsigned int v2u __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int p = 0; p < 9; ++p) {
            TS_G const v4f* q = (TS_G const v4f*)(blk + pbase + (size_t)p * slots);
            const v4f t = NT ? __builtin_nontemporal_load(q) : *q;
            c.m[p] = make_float4(t.x, t.y, t.z, t.w);
        }
        const v2u t = NT ? __builtin_nontemporal_load((TS_G const v2u*)ip) : *(TS_G const v2u*)ip;
        lo = t.x; hi = t.y;
    } else {
        typedef double v2d __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int p = 0; p < 9; ++p) {
            TS_G const v2d* q = (TS_G const v2d*)(blk + pbase + (size_t)p * slots);
            const v2d t = NT ? __builtin_nontemporal_load(q) : *q;
            c.m[p] = make_double2(t.x, t.y);
        }
        lo = NT ? __builtin_nontemporal_load((TS_G const uint32_t*)ip) : *(TS_G const uint32_t*)ip;
    }
    const uint32_t h[4] = {lo & 0xFFFFu, lo >> 16, hi & 0xFFFFu, hi >> 16};
#pragma unroll
    for (int j = 0; j < EPL; ++j) c.id[j] = idx16_expand(h[j]);
}

extern "C" int64_t vican_tiled_op_lds_bytes(int32_t n_cam, int32_t max_rows, int32_t storage, int32_t n_copy) {
    const int64_t s = ssize(storage), cp = plane_stride(n_cam);
    const int64_t per_wave = (((int64_t)max_rows * 9 * (8LL * n_copy + 8 + s)) + 15) & ~15LL;
    return 9LL * cp * (s + 8) + (int64_t)TS_NW * per_wave + 256;
}

template <typename S, int CP, int TRIPS, bool NT>
__global__ __launch_bounds__(TS_NW * 64) void tiled_sweep_kernel(const vican_tile_t* __restrict__ tiles, const int n_tile,
                                                                 const double* __restrict__ lamT_inv, const double* __restrict__ x_base,
                                                                 const int parity, uint32_t* abort_word, const unsigned long long spin_limit) {
    constexpr int NW = TS_NW, EPL = Vec<S>::N, BLOCK = NW * 64;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    __shared__ const double* s_yp[64];                        // this launch's share buffer of every tile
    const int tile = (int)blockIdx.x % n_tile, wgt = (int)blockIdx.x / n_tile, nwgt = (int)gridDim.x / n_tile;
    const vican_tile_t* Tp = tiles + tile;
    const vican_graph_t g = Tp->g;
    TS_G const S* const g_blk = (TS_G const S*)g.blk;
    TS_G const uint16_t* const g_idx = (TS_G const uint16_t*)g.idx16;
    TS_G const int32_t* const g_row0 = (TS_G const int32_t*)g.chunk_row0;
    // the operand: the tile's slice of x_base ([3C][3], tiles = consecutive camera ranges in descriptor order), else the descriptor's
    int cam0 = 0;
    for (int t = 0; t < tile; ++t) cam0 += tiles[t].g.n_cam;
    TS_G const double* const x = x_base ? (TS_G const double*)x_base + (size_t)9 * cam0 : (TS_G const double*)Tp->x;
    TS_G double* const yp_pub = (TS_G double*)Tp->ypart[parity];       // [T][9] this tile's shares, this launch
    TS_G double* const yp_clr = (TS_G double*)Tp->ypart[parity ^ 1];   // ... re-armed for the next launch
    TS_G double* const fx = (TS_G double*)Tp->fx;
    const int C = g.n_cam, nx = 9 * CP, ncopy = g.n_copy, cmask = ncopy - 1, RW = g.max_rows;
    const int tid = threadIdx.x, lane = tid & 63, lane_copy = lane & cmask;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    u64* zs = (u64*)lds_raw;                                   // [9][CP] camera accumulators of this workgroup's tile
    S* xs = (S*)(zs + nx);                                     // [9][CP] x of the tile's cameras
    const size_t per_wave = (((size_t)RW * 9 * (8 * ncopy + 8 + sizeof(S))) + 15) & ~(size_t)15;
    unsigned char* wbase = (unsigned char*)(xs + nx) + (size_t)wave * per_wave;
    u64* ys = (u64*)wbase;                                     // [RW * 9][ncopy] striped row accumulators (this wave's)
    double* yv = (double*)(ys + (size_t)RW * 9 * ncopy);       // [RW * 9] row sums over ALL tiles
    S* wv = (S*)(yv + (size_t)RW * 9);                         // [RW * 9] phase-3 operand
    const uint32_t pad_cam = (uint32_t)((lane & 31) < C ? (lane & 31) : 0);
    const double sentinel = __longlong_as_double((long long)TS_SENTINEL);

    for (int c = tid; c < C; c += BLOCK) {
#pragma unroll
        for (int i = 0; i < 9; ++i) xs[i * CP + c] = pre_scale<S>(x[(size_t)c * 9 + i], 1.0);
    }
    for (int i = tid; i < nx; i += BLOCK) zs[i] = 0ull;
    for (int i = lane; i < 9 * RW * ncopy; i += 64) ys[i] = 0ull;
    if (tid < n_tile) s_yp[tid] = tiles[tid].ypart[parity];
    // (no measured shift: the phase-3 operand depends on every tile's x - scales as vican_bip_apply / sweep MODE 4)
    const double y_scale = fx[0], y_inv = fx[1], z_scale = fx[2];
    if (wgt == 0 && tid == 0) fx[7] = 1.0;
    __syncthreads();

    const int nchunk = g.n_chunk, kmax = nchunk - 1;
    const int kstride = nwgt * NW;
    auto clampk = [&](int k) -> int { return k < kmax ? k : kmax; };
    typedef int ts_v2i __attribute__((ext_vector_type(2)));
    auto load_rows = [&](int k) -> int2 { const ts_v2i t = *(TS_G const ts_v2i*)(g_row0 + clampk(k)); return make_int2(t.x, t.y); };
    auto cam_of = [&](uint32_t id) -> uint32_t { return id == VICAN_PAD_SLOT ? pad_cam : (id & 0xFFFFu); };
    auto row_of = [&](uint32_t id) -> uint32_t { return id == VICAN_PAD_SLOT ? 0u : (id >> 16); };
    bool aborted = false;

    // phase 1 of chunk `cur` (rows r0 .. r0 + nrows): this tile's share of the row sums, published + the other buffer re-armed
    auto phase1 = [&](const ChunkRegs<S, EPL>& cur, const int r0, const int nrows) {
        uint32_t cam[EPL], row[EPL];
#pragma unroll
        for (int j = 0; j < EPL; ++j) { cam[j] = cam_of(cur.id[j]); row[j] = row_of(cur.id[j]); }
        S acc[9], xc[9], xn[9];
        bool run_real = false;
#pragma unroll
        for (int q = 0; q < 9; ++q) xc[q] = xs[q * CP + cam[0]];
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            if (j + 1 < EPL) {
#pragma unroll
                for (int q = 0; q < 9; ++q) xn[q] = xs[q * CP + cam[j + 1]];
            }
            const bool cont = j > 0 && row[j] == row[j - 1];
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) {
                    const S c = dot3<S>(vget<S>(cur.m[0 + a], j), xc[b], vget<S>(cur.m[3 + a], j), xc[3 + b],
                                        vget<S>(cur.m[6 + a], j), xc[6 + b]);
                    acc[a * 3 + b] = cont ? acc[a * 3 + b] + c : c;
                }
            // (a shared chunking pads more than a tile's own would - a quarter of the slots on the wide workload; padding slots
            //  sit at the END of a chunk in the row-major slot order, i.e. whole lanes hold nothing else: their zero sums are not sent)
            run_real = (cont && run_real) || cur.id[j] != VICAN_PAD_SLOT;     // (padding slots report row 0: a run may mix both)
            const bool last = ((j == EPL - 1) || row[j + 1 < EPL ? j + 1 : j] != row[j]) && run_real;
            if (last) {
                u64* yr = ys + (size_t)(row[j] * 9) * ncopy + lane_copy;
#pragma unroll
                for (int q = 0; q < 9; ++q) lds_add_fix(yr + q * ncopy, fix_of<S>(acc[q], y_scale));
            }
            if (j + 1 < EPL) {
#pragma unroll
                for (int q = 0; q < 9; ++q) xc[q] = xn[q];
            }
        }
        __builtin_amdgcn_wave_barrier();
        for (int i = lane; i < nrows * 9; i += 64) {
            const int oo = i % 9;
            long long s = 0;
            for (int c = 0; c < ncopy; ++c)
                s += (long long)__hip_atomic_exchange(&ys[i * ncopy + ((c + oo) & cmask)], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            ts_st(yp_pub + (size_t)r0 * 9 + i, (double)fix_total<S>(s) * y_inv);
            ts_st(yp_clr + (size_t)r0 * 9 + i, sentinel);
        }
        __builtin_amdgcn_wave_barrier();
    };

    // phases 2 + 3 of chunk `prev` (published an iteration ago): the row sums of all tiles, Lambda_t^-1, camera sums
    // `pre`: the shares of the first TS_PRE tiles for item `lane` of the chunk's rows, requested BEFORE phase 1 of the current
    // chunk (their round trip hides behind it); usable when the chunk has at most 64 row items and there are at most TS_PRE
    // tiles and none of the values is still the sentinel - otherwise the loop below (re)loads
    auto phase23 = [&](const ChunkRegs<S, EPL>& prev, const int r0, const int nrows, const double (&L)[TRIPS][3], const double (&pre)[TS_PRE]) {
        bool fast = nrows * 9 <= 64 && n_tile <= TS_PRE;
        if (fast) {
            bool pending = false;
            double y = 0.0;
#pragma unroll
            for (int t = 0; t < TS_PRE; ++t)
                if (t < n_tile) { pending |= (unsigned long long)__double_as_longlong(pre[t]) == TS_SENTINEL; y += pre[t]; }   // tile order
            pending = pending && lane < nrows * 9;
            if (__any(pending)) fast = false;
            else if (lane < nrows * 9) yv[lane] = y;
        }
        for (int base = 0; !fast && base < nrows * 9; base += 64) {
            const int i = base + lane;
            const bool live = i < nrows * 9;
            double y = 0.0;
            unsigned long long t0 = 0ull;
            unsigned int spins = 0;
            for (;;) {
                bool pending = false;
                y = 0.0;
                if (live)
                    for (int t = 0; t < n_tile; ++t) {               // tile order: deterministic
                        const double v = ts_ld((TS_G const double*)s_yp[t] + (size_t)r0 * 9 + i);
                        pending |= (unsigned long long)__double_as_longlong(v) == TS_SENTINEL;
                        y += v;
                    }
                if (!__any(pending)) break;
                __builtin_amdgcn_s_sleep(2);
                if ((++spins & 255u) == 0) {
                    if (t0 == 0ull) t0 = __builtin_amdgcn_s_memrealtime();
                    bool stop = false;
                    if (abort_word) {
                        stop = __hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u;
                        if (!stop && __builtin_amdgcn_s_memrealtime() - t0 > spin_limit) {
                            __hip_atomic_store(abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            stop = true;
                        }
                    }
                    if (stop) { aborted = true; break; }
                }
            }
            if (live) yv[i] = y;
            if (aborted) return;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int t = 0; t < TRIPS; ++t) {
            const int j = lane + 64 * t;
            if (j < nrows * 3) {
                const double* yr = yv + (j / 3) * 9;
#pragma unroll
                for (int b3 = 0; b3 < 3; ++b3)
                    wv[j * 3 + b3] = pre_scale<S>(dot3<double>(L[t][0], yr[b3], L[t][1], yr[3 + b3], L[t][2], yr[6 + b3]), z_scale);
            }
        }
        __builtin_amdgcn_wave_barrier();
    };
    auto phase3 = [&](const ChunkRegs<S, EPL>& prev) {
        S w[9];
        uint32_t prow = 0xFFFFFFFFu;
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            const uint32_t rowj = row_of(prev.id[j]), camj = cam_of(prev.id[j]);
            if (rowj != prow) {
                prow = rowj;
#pragma unroll
                for (int q = 0; q < 9; ++q) w[q] = wv[rowj * 9 + q];
            }
            u64* zc = zs + camj;
            if (prev.id[j] != VICAN_PAD_SLOT) {
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int b = 0; b < 3; ++b) {
                        const S v = dot3<S>(vget<S>(prev.m[i * 3 + 0], j), w[b], vget<S>(prev.m[i * 3 + 1], j), w[3 + b],
                                            vget<S>(prev.m[i * 3 + 2], j), w[6 + b]);
                        lds_add_fix(&zc[(i * 3 + b) * CP], fix_of<S>(v, z_scale));
                    }
            }
        }
        __builtin_amdgcn_wave_barrier();
    };

    // one iteration: requests (shares and dual blocks of chunk kc - kstride, then the prefetch of chunk kc + (TS_SETS - 2) kstride into the
    // set `fre` that finished its phase 3 an iteration ago), phase 1 of chunk kc (set `cur`), phases 2 + 3 of chunk kc - kstride
    // (set `prev`).  Loads are unconditional (clamped).
    auto body = [&](const ChunkRegs<S, EPL>& cur, const int2 vcur, const ChunkRegs<S, EPL>& prev, const int2 vprev,
                    ChunkRegs<S, EPL>& fre, int2& vfre, const int kc) {
        const int kp = kc - kstride;
        const bool have_prev = kp >= 0 && kp < nchunk;
        // requests for the pending chunk's phase 2 (unconditional: vprev always holds valid row bounds): dual blocks, shares
        const int r0p = __builtin_amdgcn_readfirstlane(vprev.x), nrp = __builtin_amdgcn_readfirstlane(vprev.y) - r0p;
        double L[TRIPS][3], pre[TS_PRE];
#pragma unroll
        for (int t = 0; t < TRIPS; ++t) {
            int j = lane + 64 * t;
            j = j < nrp * 3 ? j : 0;
            const double* Lp = lamT_inv + (size_t)r0p * 9 + (size_t)j * 3;
            L[t][0] = Lp[0]; L[t][1] = Lp[1]; L[t][2] = Lp[2];
        }
        {
            const int i = lane < nrp * 9 ? lane : 0;
#pragma unroll
            for (int t = 0; t < TS_PRE; ++t) pre[t] = ts_ld((TS_G const double*)s_yp[t < n_tile ? t : 0] + (size_t)r0p * 9 + i);
        }
        // ... then the prefetch of chunk kc + (TS_SETS - 2) kstride into the free set.  Vector-memory results return in issue order: the
        // shares (older) can be consumed after phase 1 while this prefetch stays in flight until the NEXT iteration's shares
        // are consumed - a whole iteration and a phase of latency hidden.  (Issued after phase 2 instead, the prefetch sat in
        // front of the next iteration's share requests and had to land within a phase: 386 -> 349 us with the share requests
        // moved ahead of phase 1, -> this order.)
        __builtin_amdgcn_sched_barrier(0);
        vfre = load_rows(kc + (TS_SETS - 2) * kstride);
        ts_load_chunk<S, EPL, NT>(fre, g_blk, g_idx, g.slots, clampk(kc + (TS_SETS - 2) * kstride), lane);
        __builtin_amdgcn_sched_barrier(0);
        if (kc < nchunk) {
            const int r0 = __builtin_amdgcn_readfirstlane(vcur.x);
            phase1(cur, r0, __builtin_amdgcn_readfirstlane(vcur.y) - r0);
        }
        if (have_prev) phase23(prev, r0p, nrp, L, pre);
        if (have_prev && !aborted) phase3(prev);
    };

#if TS_SETS == 3
    ChunkRegs<S, EPL> A, B, Cc;
    int kc = wgt * NW + wave;
    int2 vA = load_rows(kc), vB = vA, vC = vA;
    ts_load_chunk<S, EPL, NT>(A, g_blk, g_idx, g.slots, clampk(kc), lane);
    // (iteration i: cur = set i % 3, prev = set (i - 1) % 3, free = set (i + 1) % 3)
#pragma unroll 1
    while (kc - kstride < nchunk && !aborted) {
        body(A, vA, Cc, vC, B, vB, kc); kc += kstride;
        if (!(kc - kstride < nchunk) || aborted) break;
        body(B, vB, A, vA, Cc, vC, kc); kc += kstride;
        if (!(kc - kstride < nchunk) || aborted) break;
        body(Cc, vC, B, vB, A, vA, kc); kc += kstride;
    }
#else
    ChunkRegs<S, EPL> A, B, Cc, D;
    int kc = wgt * NW + wave;
    int2 vA = load_rows(kc), vB = load_rows(kc + kstride), vC = vA, vD = vA;
    ts_load_chunk<S, EPL, NT>(A, g_blk, g_idx, g.slots, clampk(kc), lane);
    ts_load_chunk<S, EPL, NT>(B, g_blk, g_idx, g.slots, clampk(kc + kstride), lane);
    // (iteration i: cur = set i % 4, prev = set (i - 1) % 4, free = set (i + 2) % 4; ends when neither a chunk nor a pending one is left)
#pragma unroll 1
    while (kc - kstride < nchunk && !aborted) {
        body(A, vA, D, vD, Cc, vC, kc); kc += kstride;
        if (!(kc - kstride < nchunk) || aborted) break;
        body(B, vB, A, vA, D, vD, kc); kc += kstride;
        if (!(kc - kstride < nchunk) || aborted) break;
        body(Cc, vC, B, vB, A, vA, kc); kc += kstride;
        if (!(kc - kstride < nchunk) || aborted) break;
        body(D, vD, Cc, vC, B, vB, kc); kc += kstride;
    }
#endif
    __syncthreads();
    TS_G u64* zp = (TS_G u64*)Tp->zpart + (size_t)wgt * 9 * C;         // slab layout [9][C]
#pragma unroll
    for (int q = 0; q < 9; ++q)
        for (int c = tid; c < C; c += BLOCK) zp[q * C + c] = (u64)fix_total<S>((long long)zs[q * CP + c]);
}

// The tiled operator in one launch.  tiles_host / tiles_dev: the same n_tile descriptors in host and device memory (all wave
// layouts with the SAME chunking - n_chunk, chunk rows, slots, storage; zpart of every tile holds n_wg_tile slabs; ypart[2]:
// the tile's two share buffers [T][9], both filled with the sentinel (vican_tiled_op_sentinel) before the first launch and
// after an aborted one).  parity: 0, 1, 0, ... on successive launches.  z_cam of tile k: slab-reduce its zpart over n_wg_tile
// slabs afterwards (vican_slab_reduce_fx with fx + 3, fx + 7).  n_wg_tile workgroups per tile, n_tile * n_wg_tile <= CUs.
static int tiled_op_launch(const vican_tile_t* tiles_host, const vican_tile_t* tiles_dev, int32_t n_tile, int32_t n_wg_tile,
                           const double* lamT_inv, const double* x_base, int32_t parity, void* stream) {
    if (!tiles_host || !tiles_dev || n_tile <= 0 || n_tile > 64 || n_wg_tile <= 0 || !lamT_inv || (parity != 0 && parity != 1))
        return set_err(VICAN_ERR_ARG, "vican_tiled_op: bad argument");
    const vican_graph_t& g0 = tiles_host[0].g;
    int cp = 0, max_rows = 0, n_copy = 0, n_cam = 0;
    for (int k = 0; k < n_tile; ++k) {
        const vican_tile_t& t = tiles_host[k];
        if (int rc = vican_check_graph(&t.g, "vican_tiled_op")) return rc;
        if (t.g.layout != VICAN_LAYOUT_WAVE || !t.g.blk || !t.g.idx16 || (!t.x && !x_base) || !t.zpart || !t.fx || !t.ypart[0] || !t.ypart[1])
            return set_err(VICAN_ERR_ARG, "vican_tiled_op: tiles must be wave layouts with all buffers set");
        if (t.g.n_chunk != g0.n_chunk || t.g.slots != g0.slots || t.g.storage != g0.storage || t.g.n_time != g0.n_time ||
            t.g.stream_nt != g0.stream_nt || t.g.n_chunk == 0)
            return set_err(VICAN_ERR_ARG, "vican_tiled_op: the tiles do not share one chunking");
        cp = (int)plane_stride(t.g.n_cam) > cp ? (int)plane_stride(t.g.n_cam) : cp;
        max_rows = t.g.max_rows > max_rows ? t.g.max_rows : max_rows;
        n_copy = t.g.n_copy > n_copy ? t.g.n_copy : n_copy;
        n_cam = t.g.n_cam > n_cam ? t.g.n_cam : n_cam;
    }
    const size_t lds = (size_t)vican_tiled_op_lds_bytes(n_cam, max_rows, g0.storage, n_copy);
    if ((int64_t)lds > vican_lds_limit_bytes()) return set_err(VICAN_ERR_CAPACITY, "vican_tiled_op: camera tables / row staging do not fit in LDS");
    const int trips = (3 * max_rows + 63) / 64, grid = n_tile * n_wg_tile;
    if (trips > 3) return set_err(VICAN_ERR_CAPACITY, "vican_tiled_op: more than 64 rows per chunk");
    hipStream_t st = (hipStream_t)stream;
#define TS_LAUNCH4(S_, CP_, T_, NT_)                                                                                         \
    do {                                                                                                                     \
        auto kern = tiled_sweep_kernel<S_, CP_, T_, NT_>;                                                                    \
        static size_t conf = 0;                                                                                              \
        if (lds > conf) {                                                                                                    \
            if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)  \
                return set_err(VICAN_ERR_LAUNCH, "vican_tiled_op: cannot raise dynamic LDS limit");                          \
            conf = lds;                                                                                                      \
        }                                                                                                                    \
        /* co-residency on EVERY launch, as the other cooperative launchers do: it depends on the grid and the device, not */ \
        /* only on the LDS size (the workgroups spin on each other's shares: a grid that is not resident would hang)       */ \
        if (int rc = vican_coresident_ok((const void*)kern, TS_NW * 64, lds, grid, "vican_tiled_op")) return rc;             \
        VICAN_LAUNCH_SWEEP(kern, dim3(grid), dim3(TS_NW * 64), lds, st, tiles_dev, (int)n_tile, lamT_inv, x_base, (int)parity, \
                           g_vican_abort_word, g_vican_sync_ticks);                                                          \
    } while (0)
#define TS_LAUNCH3(S_, CP_, T_) do { if (g0.stream_nt) TS_LAUNCH4(S_, CP_, T_, true); else TS_LAUNCH4(S_, CP_, T_, false); } while (0)
#define TS_LAUNCH2(S_, CP_) do { if (trips <= 1) TS_LAUNCH3(S_, CP_, 1); else if (trips == 2) TS_LAUNCH3(S_, CP_, 2); else TS_LAUNCH3(S_, CP_, 3); } while (0)
#define TS_LAUNCH1(S_) do { if (cp == 256) TS_LAUNCH2(S_, 256); else if (cp == 512) TS_LAUNCH2(S_, 512); else TS_LAUNCH2(S_, 1024); } while (0)
    if (g0.storage == VICAN_STORE_F32) TS_LAUNCH1(float); else TS_LAUNCH1(double);
#undef TS_LAUNCH1
#undef TS_LAUNCH2
#undef TS_LAUNCH3
#undef TS_LAUNCH4
    LAUNCH_CHECK("vican_tiled_op");
    return VICAN_OK;
}
extern "C" int vican_tiled_op(const vican_tile_t* tiles_host, const vican_tile_t* tiles_dev, int32_t n_tile, int32_t n_wg_tile,
                              const double* lamT_inv, int32_t parity, void* stream) {
    return tiled_op_launch(tiles_host, tiles_dev, n_tile, n_wg_tile, lamT_inv, nullptr, parity, stream);
}

// The slab folds of all tiles in one launch: z[cam0_k + c][q] = scale_k * sum over the n_wg_tile slabs of tile k (exact integer
// sums; scale_k = fx_k[3] * fx_k[7] as vican_slab_reduce_fx(zpart_k, n_wg_tile, C_k, 9, 1.0, fx_k + 3, fx_k + 7, z_k)).
// 1024 threads = 64 columns x 16 groups striding over the slabs; blockIdx.y = tile.
__global__ __launch_bounds__(1024) void tiled_fold_kernel(const int32_t* __restrict__ gate, const vican_tile_t* __restrict__ tiles,
                                                          const int n_slab, double* __restrict__ z) {
    GATE_RETURN(gate);
    __shared__ long long sh[1024];
    const int tile = (int)blockIdx.y;
    const long long C = tiles[tile].g.n_cam, n = 9 * C;
    if ((long long)blockIdx.x * 64 >= n) return;
    int cam0 = 0;
    for (int t = 0; t < tile; ++t) cam0 += tiles[t].g.n_cam;
    const long long* part = (const long long*)tiles[tile].zpart;
    const double* fx = tiles[tile].fx;
    const int e = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const long long i = (long long)blockIdx.x * 64 + e;
    long long s = 0;
    if (i < n)
        for (int k = grp; k < n_slab; k += 16) s += part[(size_t)k * n + i];
    sh[threadIdx.x] = s;
    __syncthreads();
    if (grp == 0 && i < n) {
        long long t = 0;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += sh[k * 64 + e];
        const long long q = i / C, cam = i % C;
        z[(size_t)(cam0 + cam) * 9 + q] = (double)t * (fx[3] * fx[7]);
    }
}

// The tiled operator with operand and result in the caller's arrays: z = P x for x, z [3C][3] doubles (tiles = consecutive camera
// ranges in descriptor order; the descriptors' x fields are not read) - the fused launch and ONE fold launch for all tiles.
extern "C" int vican_tiled_op_z(const vican_tile_t* tiles_host, const vican_tile_t* tiles_dev, int32_t n_tile, int32_t n_wg_tile,
                                const double* lamT_inv, const double* x, double* z, int32_t parity, void* stream) {
    if (!x || !z) return set_err(VICAN_ERR_ARG, "vican_tiled_op_z: NULL operand or result");
    if (int rc = tiled_op_launch(tiles_host, tiles_dev, n_tile, n_wg_tile, lamT_inv, x, parity, stream)) return rc;
    int n_cam = 0;
    for (int k = 0; k < n_tile; ++k) n_cam = tiles_host[k].g.n_cam > n_cam ? tiles_host[k].g.n_cam : n_cam;
    hipLaunchKernelGGL(tiled_fold_kernel, dim3((unsigned)((9LL * n_cam + 63) / 64), (unsigned)n_tile), dim3(1024), 0, (hipStream_t)stream,
                       g_vican_gate, tiles_dev, (int)n_wg_tile, z);
    LAUNCH_CHECK("vican_tiled_op_z");
    return VICAN_OK;
}

__global__ void tiled_sentinel_kernel(double* p, long long n) {
    const double s = __longlong_as_double((long long)TS_SENTINEL);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) p[i] = s;
}
extern "C" int vican_tiled_op_sentinel(double* ypart, int64_t n, void* stream) {
    if (!ypart || n < 0) return set_err(VICAN_ERR_ARG, "vican_tiled_op_sentinel: bad argument");
    if (n == 0) return VICAN_OK;
    hipLaunchKernelGGL(tiled_sentinel_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, ypart, (long long)n);
    LAUNCH_CHECK("vican_tiled_op_sentinel");
    return VICAN_OK;
}
