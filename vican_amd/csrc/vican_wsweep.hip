// vican_wsweep.hip - the edge sweep with ONE WAVEFRONT PER CHUNK ("wave layout", vican_graph_t.layout == 1).
//
// Same mathematics, same data layout and the same exact 64-bit fixed-point accumulation as block_sweep_kernel
// (vican_sweep.hip), but a chunk is 64 lanes x EPL slots = whole timestep rows with at most 256 (f32) / 128 (f64)
// edges, and a wavefront takes a chunk through all three phases on its own:
//
//     phase 1   y_row += M^T x_cam           x gathered from the workgroup's LDS planes, striped row accumulators in
//                                            the WAVE's private LDS region
//     phase 2   w_row  = Lambda_T^-1 y_row   lanes of the same wavefront: fold the stripes, 3x3 product per row
//     phase 3   z_cam += M w_row             blocks still in registers; z accumulators shared by the workgroup
//
// LDS operations of one wavefront execute in issue order, so the phases need NO workgroup barrier: the only
// __syncthreads() of the kernel are in the prologue and before the slab write-out.  Why: the block sweep is neither
// HBM- nor issue-bound (PMC, round 1: VALU 55 % busy, LDS 43 %, HBM at 85 % of the streaming ceiling) - its 12
// wavefronts move through the phases in lock-step, so the LDS-atomic phase, the VALU phase and the row fold never
// overlap and every barrier costs the skew of the slowest wavefront.  Here the wavefronts of a workgroup drift
// apart: while one issues its phase-3 atomics another multiplies and a third waits for its next chunk, each with
// its own prefetch in flight.
//
// Scheduling.  Chunks are handed out in two levels, both dynamic, neither on the critical path:
//   * device level: one counter (fx[12]) hands out RANGES of NW or 2 NW consecutive chunks to workgroups - a few
//     thousand device-scope atomics per launch, each issued by ONE wavefront of the workgroup a full round before the
//     workgroup runs dry, so nobody waits for it;
//   * workgroup level: the wavefronts draw single chunks of the published ranges through an LDS ticket counter.
// Why both: wavefronts of one CU progress unevenly (static round-robin inside the workgroup: 194-199 us, LDS tickets
// 185 us on the same box), and XCDs get different shares of the HBM bandwidth (static ranges per workgroup: mean finish
// time by XCD between 181 and 202 us of a 212 us launch).  A first attempt that let wavefronts steal from a pool with
// blocking device-scope atomics made the launch SLOWER the larger the pool (10 % pool 208 us, 28 % 236 us): the
// round trip of an atomic under a saturated memory system is many microseconds.
// Sums are exact integers, so which wavefront adds a chunk cannot change a bit of the result; a cap on the ranges a
// workgroup may take keeps the number of adds into one z accumulator within the bound fx_finish assumed.
#include "vican_sweep_common.h"

#ifndef VICAN_WSWEEP_PART

extern "C" int64_t vican_wsweep_lds_bytes(int32_t n_cam, int32_t max_rows, int32_t storage, int32_t n_copy, int32_t n_waves) {
    const int64_t s = ssize(storage), cp = plane_stride(n_cam);
    const int64_t per_wave = (((int64_t)max_rows * 9 * (8LL * n_copy + 8 + s)) + 15) & ~15LL;
    return 9LL * cp * (s + 8) + (int64_t)n_waves * per_wave + 256;
}

#endif

#if !defined(VICAN_WSWEEP_SPLIT) || defined(VICAN_WSWEEP_PART)

// MODE 0: zpart[wg] = sum M (lamT_inv (sum M^T x))          operator P x
// MODE 1: lamT_out[t] = Z_t = sum_c M_ct^T x_c               dual update (SVDs in dual_svd_kernel)
// MODE 3: MODE 1 and zpart[wg] = sum M polar(Z_t)            dual update fused with the next operator application
// MODE 4: zpart[wg] = sum M w_t, w_t GIVEN (lamT_inv = [T][9])  camera half of the operator on a camera tile (more cameras than one
//         LDS table holds: rows pass = MODE 1 per tile, w_t = Lambda_t^-1 (sum of the tiles' Z_t), then this) - no x table,
//         no phase 1; |w_t|_F <= omega * x_bound, the bound fx[2] was sized for
// FB (MODE 3): rows whose Newton polar iteration does not apply (ill-conditioned Z_t) take the SVD inside the kernel.
// That path is a function call, and a call site makes the register allocator spill in the streaming loop at 12
// wavefronts (168 VGPRs: 89 spilled, 774 us against 181 us for MODE 0; without the call 164 VGPRs, no spills).  The
// 12-wavefront instantiation is therefore built with FB = false: such a row raises the word sched[4] instead, and the
// launcher enqueues the 8-wavefront FB = true instantiation right behind it, gated on that word - it reruns the whole
// sweep (same outputs) only when a row needed it; dual_svd_kernel clears the word.
// ONE: every chunk of the graph holds exactly ONE row (n_chunk == n_time: dense rows, e.g. 250 cameras per timestep).  The
// row's sums then need no accumulators at all: the lanes' partial sums are reduced over the wavefront by three shared DPP
// butterflies (wave_total3: nine f64 sums, fixed order, deterministic), the row's 3x3 product / polar factor is formed by every
// lane for its column and handed to phase 3 through v_readlane (wave-uniform operands), and what is left in LDS is the x gather
// and the z atomics: per chunk 72 LDS wave-instructions instead of ~100, no phase-2 LDS round trip (stamps of the general
// kernel on the stress graph: phases 1 + 2 = 4.0 k of 9.4 k cycles per chunk and wavefront; counters: 44 % of the LDS-active
// cycles were bank-conflict cycles, most of them the same-address pairs of the striped row accumulators).
// The edge stream is read through the 2-byte index of the wave layout (vican_graph_t.idx16, load_chunk16): 38 instead of 40 bytes
// per edge in f32 (operator sweep on the stress graph 167.8 -> 155.9 us on the same box).
// NT: non-temporal loads of the edge stream, chosen at compile time (load_chunk).
template <typename S, int NW, int MODE, int CP, int TRIPS, bool FB = true, bool ONE = false, bool NT = false>
__global__ __launch_bounds__(NW * 64) void wave_sweep_kernel(const int32_t* __restrict__ gate, vican_graph_t g,
                                                             const double* __restrict__ lamT_inv,
                                                             const double* __restrict__ x, u64* __restrict__ zpart,
                                                             double* __restrict__ lamT_out, double* __restrict__ fx) {
    GATE_RETURN(gate);
    constexpr int EPL = Vec<S>::N, BLOCK = NW * 64;
    constexpr bool HAS_Z = (MODE == 0 || MODE == 3 || MODE == 4);
    // TRIPS: (row, dual-block row) items per lane in phase 2 = ceil(3 max_rows / 64), 1..3 (max_rows <= 64)
    extern __shared__ __align__(16) unsigned char lds_raw[];
    __shared__ double s_xm[16];
    const int C = g.n_cam, nx = 9 * CP, ncopy = g.n_copy, cmask = ncopy - 1, RW = g.max_rows;
    const int tid = threadIdx.x, lane = tid & 63, lane_copy = lane & cmask;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // wave-uniform: scalar registers, not one VGPR per use
    u64* zs = (u64*)lds_raw;                                   // [9][CP] planes, shared by the workgroup
    S* xs = (S*)(zs + (HAS_Z ? nx : 0));                       // [9][CP] planes
    const size_t per_wave = (((size_t)RW * 9 * (8 * ncopy + 8 + sizeof(S))) + 15) & ~(size_t)15;
    unsigned char* wbase = (unsigned char*)(xs + nx) + (size_t)wave * per_wave;
    u64* ys = (u64*)wbase;                                     // [RW * 9][ncopy] striped row accumulators (this wave's)
    double* yv = (double*)(ys + (size_t)RW * 9 * ncopy);       // [RW * 9] folded row sums
    S* wv = (S*)(yv + (size_t)RW * 9);                         // [RW * 9] phase-3 operand
    const uint32_t pad_cam = (uint32_t)((lane & 31) < C ? (lane & 31) : 0);

    // ---- chunk scheduler (see the header).  Ranges are published in a ring of 8 LDS entries {first ticket, end ticket,
    // first chunk}; the wavefront that draws the ticket NW before the end of the newest range fetches the next one.
    const int nwg = (int)gridDim.x, nchunk = g.n_chunk;
    unsigned int* sched = (unsigned int*)(fx + 12);            // [0] ranges handed out (units of NW chunks), [8] workgroups finished
    constexpr int NONE = 0x7fffffff, RING = 8;
    const int unit0 = nchunk >= 4 * NW * nwg ? 2 : 1;          // implicit first range of workgroup w: units [unit0 w, unit0 (w + 1))
    const int cap_units = g.wg_chunk_cap > 0 ? g.wg_chunk_cap / NW : 0x3fffffff;
    __shared__ int s_cnt, s_npub, s_done, s_units, s_v0[RING], s_v1[RING], s_base[RING];
    // first range of workgroup w (implicit, no atomic): units [unit0 w, unit0 (w + 1)); its first NW tickets are implicit
    // too - wavefront i holds ticket i - so that the first chunk is requested before the prologue barrier
    const int base0 = (int)blockIdx.x * unit0 * NW;
    const int len0 = nchunk - base0 < 0 ? 0 : (nchunk - base0 > unit0 * NW ? unit0 * NW : nchunk - base0);
    if (tid == 0) {
        s_cnt = NW; s_v0[0] = 0; s_v1[0] = len0; s_base[0] = base0; s_npub = 1; s_units = unit0;
        s_done = len0 < unit0 * NW ? 1 : 0;                    // the queue ends inside (or before) this range
    }
    int my_r = 0;                                              // newest ring entry this wavefront has looked at (wave-uniform)
    // ticket -> chunk.  Tickets are unique in the workgroup; entries are published in ticket order.
    auto resolve = [&](const int v) -> int {
        for (;;) {
            const int npub = __hip_atomic_load(&s_npub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("" ::: "memory");                       // (the entries are read after the count that covers them)
#define LDSV(x) __hip_atomic_load(&(x), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)     /* never cached in a register */
            while (my_r < npub - 1 && v >= LDSV(s_v1[my_r & (RING - 1)])) ++my_r;
            const int e = my_r & (RING - 1), v0 = LDSV(s_v0[e]), v1 = LDSV(s_v1[e]);
            if (v < v1) {
                const int k = LDSV(s_base[e]) + (v - v0);
                // the one wavefront whose ticket sits a round before the end of the NEWEST range fetches the next range
                const int trig = v1 - v0 > NW ? v1 - NW : v0;
                if (v == trig && my_r == npub - 1 && !__hip_atomic_load(&s_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) {
                    const int taken = LDSV(s_units);
                    int done = 1, nb = 0, nl = 0;
                    if (taken < cap_units) {
                        // two units while the queue is long (fewer atomics), one towards its end (finer balance)
                        // (never past the cap: the host sizes the fixed-point head-room of the z accumulators from exactly cap_units)
                        int want = (long long)(k + 2 * nwg * NW) * 8 < (long long)nchunk * 7 ? 2 : 1;
                        if (want > cap_units - taken) want = cap_units - taken;
                        unsigned int p = 0;
                        if (lane == 0) p = __hip_atomic_fetch_add(&sched[0], (unsigned)want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        p = (unsigned int)__builtin_amdgcn_readfirstlane((int)p);
                        const long long first = ((long long)unit0 * nwg + p) * NW;
                        if (first < nchunk) {
                            nb = (int)first; nl = (int)((long long)nchunk - first < want * NW ? nchunk - first : want * NW);
                            done = nl < want * NW;
                        }
                    }
                    if (lane == 0) {
                        if (nl > 0) {
                            const int e2 = (my_r + 1) & (RING - 1);
                            s_v0[e2] = v1; s_v1[e2] = v1 + nl; s_base[e2] = nb; s_units = taken + (nl + NW - 1) / NW;
                        }
                        // ORDER: the entry, then s_npub, then s_done - a wavefront that finds s_done set must also find the last
                        // range published (it reads s_done first, s_npub second: below).  With s_done stored first, a wavefront
                        // polling between the two stores saw "done, nothing new" and left with its ticket inside the range
                        // being published: that chunk's rows were missing from the sums (about one sweep in 10^4).
                        // (LDS operations of one wavefront execute in order; the compiler barriers keep the stores in THIS order - a
                        //  release store would also wait for the wavefront's prefetch of the next chunk, vmcnt(0))
                        asm volatile("" ::: "memory");
                        if (nl > 0) __hip_atomic_store(&s_npub, npub + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        asm volatile("" ::: "memory");
                        if (done) __hip_atomic_store(&s_done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
                return k;
            }
            if (my_r < npub - 1) continue;                     // (entries were published meanwhile)
            // (s_done FIRST, s_npub second - the order of the two stores reversed; loads of one wavefront return in order)
            const int done_now = __hip_atomic_load(&s_done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            asm volatile("" ::: "memory");
            if (done_now && __hip_atomic_load(&s_npub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) == npub) return NONE;
            __builtin_amdgcn_s_sleep(8);                       // the next range is on its way
        }
    };
    auto draw = [&]() -> int {
        int t = 0;
        if (lane == 0) t = __hip_atomic_fetch_add(&s_cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return t;                                              // (valid in lane 0; resolved later)
    };
    int kc = wave < len0 ? base0 + wave : NONE, kn1 = NONE;
    ChunkRegs<S, EPL> ra, rb;
    // first row of a chunk and of the chunk behind it: VECTOR loads (every lane the same address) issued one body
    // ahead.  Scalar loads would share lgkmcnt with the LDS operations and, returning out of order, turn every LDS wait
    // of the body into lgkmcnt(0) until they land (thousands of cycles under a saturated memory system).
    const int kmax = nchunk - 1;
    auto load_rows = [&](int k) -> int2 { k = k < kmax ? k : kmax; return *(const int2*)(g.chunk_row0 + k); };
    int2 va = load_rows(kc), vb;              // (chunk_row0 has n_chunk + 1 entries; 8-byte loads may be unaligned: fine)

    // ---- prologue (as block_sweep_kernel): measure max_c |x_c|_F, stage x as planes, zero the accumulators
    constexpr int XC = MODE == 4 ? 1 : (CP + BLOCK - 1) / BLOCK;
    double xv[XC][9];
#pragma unroll
    for (int m = 0; m < XC; ++m) {
        const int c = tid + m * BLOCK;
#pragma unroll
        for (int i = 0; i < 9; ++i) xv[m][i] = (MODE != 4 && c < C) ? x[(size_t)c * 9 + i] : 0.0;
    }
    const double fx0 = fx[0], fx1 = fx[1], fx2 = fx[2], fx8 = fx[8], fx9 = (MODE == 3) ? fx[9] : 0.0;
    __builtin_amdgcn_sched_barrier(0);
    load_chunk16<S, EPL, NT>(ra, g, kc < kmax ? kc : kmax, lane);
    __builtin_amdgcn_sched_barrier(0);
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
    if (lane == 0) s_xm[wave] = xm2;
#pragma unroll
    for (int m = 0; m < XC; ++m) {
        const int c = tid + m * BLOCK;
        if (MODE != 4 && c < C) {
#pragma unroll
            for (int i = 0; i < 9; ++i) xs[i * CP + c] = pre_scale<S>(xv[m][i], 1.0);
        }
    }
    if (HAS_Z) for (int i = tid; i < nx; i += BLOCK) zs[i] = 0ull;
    if (!ONE && MODE != 4) for (int i = lane; i < 9 * RW * ncopy; i += 64) ys[i] = 0ull;
    int vzero = 0;                                             // opaque per-lane zero: keeps wave-uniform addresses on VECTOR loads
    if (ONE) asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));
    __syncthreads();
    if (kc != NONE) {
        if (wave == 0 && len0 <= NW) (void)resolve(0);         // (a short first range: ticket 0 is the one that fetches the next)
        kn1 = resolve(__builtin_amdgcn_readfirstlane(draw()));
    }
    xm2 = 0.0;
#pragma unroll
    for (int i = 0; i < NW; ++i) xm2 = fmax(xm2, s_xm[i]);
    int shift = 0;
    if (xm2 > 0.0) { const double r2 = fx8 * fx8 / xm2; shift = r2 >= 1.0 ? (ilogb(r2) >> 1) : 0; }
    shift = shift < 0 ? 0 : (shift > 40 ? 40 : shift);
    if (MODE == 3 || MODE == 4) shift = 0;                     // phase-3 operand = polar factors, |.|_F = x_bound / the given rows
    const double up = ldexp(1.0, shift);
    // (wave-uniform doubles that live through the whole loop: into scalar registers)
    auto uni = [](double v) -> double {
        const long long b = __double_as_longlong(v);
        const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)((unsigned long long)b >> 32));
        return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
    };
    const double y_scale = uni(fx0 * up), y_inv = uni(fx1 / up), z_scale = uni((MODE == 3) ? fx9 : fx2 * up);
    if ((MODE == 0 || MODE == 4) && blockIdx.x == 0 && tid == 0) fx[7] = 1.0 / up;
    if ((MODE == 1 || MODE == 3) && blockIdx.x == 0 && tid == 0) fx[4] = 0.0;   // omega bound: raised by dual_svd_kernel

    // body: this wavefront processes the chunk held in `cur` (its row bounds in `vrow`, loaded a body ago), prefetches
    // chunk kpref into `nxt` and requests kpref's row bounds (returned; consumed by the next body).  Every
    // vector-memory operation of the body is UNCONDITIONAL (indices clamped instead): results retire in issue order,
    // and a load that may or may not have been issued makes the compiler wait for vmcnt(0) - i.e. for the whole
    // prefetch - wherever an older result is needed.
    auto body = [&](ChunkRegs<S, EPL>& cur, ChunkRegs<S, EPL>& nxt, const int2 vrow, const int kpref, int& ticket) -> int2 {
        ticket = draw();                                       // for the chunk after next; resolved after the body
        const int r0 = __builtin_amdgcn_readfirstlane(vrow.x);
        const int nrows = __builtin_amdgcn_readfirstlane(vrow.y) - r0;
        const int kp = kpref < kmax ? kpref : kmax;
        const int2 vnext = load_rows(kp);
        // dual-block rows for phase 2, one (row, block row) item = 24 contiguous bytes per lane: the chunk's rows are
        // consecutive, so these loads are fully coalesced.  Issued BEFORE the prefetch (in-order retirement again).
        double L[TRIPS][3];
        double L1[9];                                          // ONE: the row's whole dual block in every lane
        double W4[MODE == 4 ? 3 * TRIPS : 1];                  // MODE 4: the given phase-3 operand of the chunk's rows, 9 nrows doubles
        if ((MODE == 0 || MODE == 4) && ONE) {
            const double* Lp = lamT_inv + (size_t)r0 * 9 + vzero;
#pragma unroll
            for (int q = 0; q < 9; ++q) L1[q] = Lp[q];
        } else if (MODE == 4) {
#pragma unroll
            for (int t = 0; t < 3 * TRIPS; ++t) {
                int j = lane + 64 * t;
                j = j < nrows * 9 ? j : 0;                  // (lanes without an item re-read item 0)
                W4[t] = lamT_inv[(size_t)r0 * 9 + j];
            }
        } else if (MODE == 0) {
#pragma unroll
            for (int t = 0; t < TRIPS; ++t) {
                int j = lane + 64 * t;
                j = j < nrows * 3 ? j : 0;                  // lanes without an item re-read item 0 (never used)
                const double* Lp = lamT_inv + (size_t)r0 * 9 + (size_t)j * 3;
                L[t][0] = Lp[0]; L[t][1] = Lp[1]; L[t][2] = Lp[2];
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        load_chunk16<S, EPL, NT>(nxt, g, kp, lane);

        // ---- phase 1
        // descriptors, dual loads, prefetch issue
        // LEAN (16 wavefronts = 128 VGPRs): no double-buffered x gather, camera / row indices re-derived in phase 3
        constexpr bool LEAN = NW >= 16;
        auto cam_of = [&](uint32_t id) -> uint32_t { return id == VICAN_PAD_SLOT ? pad_cam : (id & 0xFFFFu); };
        auto row_of = [&](uint32_t id) -> uint32_t { return id == VICAN_PAD_SLOT ? 0u : (id >> 16); };
        uint32_t cam[EPL], row[EPL];
#pragma unroll
        for (int j = 0; j < EPL; ++j) { cam[j] = cam_of(cur.id[j]); row[j] = row_of(cur.id[j]); }
        double ycol[3] = {0, 0, 0};                            // ONE: column min(lane & 3, 2) of the row's 3x3 sum Z_t
        if constexpr (MODE != 4) {
            S acc[9], xc[9], xn[9];
#pragma unroll
            for (int q = 0; q < 9; ++q) xc[q] = xs[q * CP + cam[0]];
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
                if (!LEAN && j + 1 < EPL) {
#pragma unroll
                    for (int q = 0; q < 9; ++q) xn[q] = xs[q * CP + cam[j + 1]];
                }
                const bool cont = ONE ? j > 0 : (j > 0 && row[j] == row[j - 1]);
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int b = 0; b < 3; ++b) {
                        const S c = dot3<S>(vget<S>(cur.m[0 + a], j), xc[b], vget<S>(cur.m[3 + a], j), xc[3 + b],
                                            vget<S>(cur.m[6 + a], j), xc[6 + b]);
                        acc[a * 3 + b] = cont ? acc[a * 3 + b] + c : c;
                    }
                const bool last = !ONE && ((j == EPL - 1) || row[j + 1 < EPL ? j + 1 : j] != row[j]);
                if (last) {
                    u64* yr = ys + (size_t)(row[j] * 9) * ncopy + lane_copy;
#pragma unroll
                    for (int q = 0; q < 9; ++q) lds_add_fix(yr + q * ncopy, fix_of<S>(acc[q], y_scale));
                }
                if (j + 1 < EPL) {
#pragma unroll
                    for (int q = 0; q < 9; ++q) xc[q] = LEAN ? xs[q * CP + cam[j + 1]] : xn[q];
                }
            }
            if (ONE) {
                // the lane's slots all belong to the chunk's one row (padding blocks are zero): nine wave sums in f64, three
                // streams per butterfly - lane l ends up with column min(l & 3, 2) of the row's 3x3 sum
#pragma unroll
                for (int a = 0; a < 3; ++a) ycol[a] = wave_total3((double)acc[a * 3 + 0], (double)acc[a * 3 + 1], (double)acc[a * 3 + 2], lane);
            }
        }
        __builtin_amdgcn_wave_barrier();
        // phase 1

        // ---- phase 2
        S w1[9];                                               // ONE: the phase-3 operand, wave-uniform
        if constexpr (ONE) {
            const int m = (lane & 3) < 2 ? (lane & 3) : 2;     // the column this lane holds
            if ((MODE == 1 || MODE == 3) && lane < 3) {        // Z_t for dual_svd_kernel
#pragma unroll
                for (int a = 0; a < 3; ++a) lamT_out[(size_t)r0 * 9 + a * 3 + lane] = ycol[a];
            }
            double wcol[3] = {0, 0, 0};
            if (MODE == 0) {
#pragma unroll
                for (int i = 0; i < 3; ++i) wcol[i] = dot3<double>(L1[i * 3 + 0], ycol[0], L1[i * 3 + 1], ycol[1], L1[i * 3 + 2], ycol[2]);
            } else if (MODE == 3) {
                // every lane forms the polar factor of the whole Z_t (nine wave-uniform doubles through v_readlane)
                double Zt[9], R[9];
#pragma unroll
                for (int a = 0; a < 3; ++a)
#pragma unroll
                    for (int b3 = 0; b3 < 3; ++b3) Zt[a * 3 + b3] = lane_bcast(ycol[a], b3);
                if (FB) {
                    polar_newton3(Zt, R);
                } else if (!polar_newton_core(Zt, R)) {
#pragma unroll
                    for (int q = 0; q < 9; ++q) R[q] = 0.0;
                    sched[VICAN_SCHED_REDO] = 1u;               // the gated FB instantiation redoes this sweep
                }
#pragma unroll
                for (int i = 0; i < 3; ++i) wcol[i] = m == 0 ? R[i * 3] : (m == 1 ? R[i * 3 + 1] : R[i * 3 + 2]);
            }
            if (MODE == 4) {
#pragma unroll
                for (int q = 0; q < 9; ++q) w1[q] = pre_scale<S>(L1[q], z_scale);      // (wave-uniform already)
            } else if (HAS_Z) {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const S ws = pre_scale<S>(wcol[i], z_scale);
#pragma unroll
                    for (int b3 = 0; b3 < 3; ++b3) {
                        if (sizeof(S) == 4) w1[i * 3 + b3] = (S)__builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, (float)ws), b3));
                        else w1[i * 3 + b3] = (S)lane_bcast((double)ws, b3);
                    }
                }
            }
        } else {
        // ---- phase 2 (this wavefront's rows only; LDS operations of a wavefront execute in order)
        if (MODE == 4) {
#pragma unroll
            for (int t = 0; t < 3 * TRIPS; ++t) {
                const int j = lane + 64 * t;
                if (j < nrows * 9) wv[j] = pre_scale<S>(W4[t], z_scale);
            }
        } else
        for (int i = lane; i < nrows * 9; i += 64) {
            const int oo = i % 9;
            long long s = 0;
            for (int c = 0; c < ncopy; ++c)          // read and clear in ONE LDS operation (ds_wrxchg_rtn_b64)
                s += (long long)__hip_atomic_exchange(&ys[i * ncopy + ((c + oo) & cmask)], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            const double y = (double)fix_total<S>(s) * y_inv;
            yv[i] = y;
            if (MODE != 0) lamT_out[(size_t)r0 * 9 + i] = y;        // Z_t for dual_svd_kernel (contiguous over the chunk)
        }
        __builtin_amdgcn_wave_barrier();
        if (MODE == 0) {
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
        }
        if (MODE == 3) {
            for (int r = lane; r < nrows; r += 64) {
                double R[9];
                if (FB) {
                    polar_newton3(yv + r * 9, R);
                } else if (!polar_newton_core(yv + r * 9, R)) {
#pragma unroll
                    for (int q = 0; q < 9; ++q) R[q] = 0.0;
                    sched[VICAN_SCHED_REDO] = 1u;               // the gated FB instantiation redoes this sweep
                }
#pragma unroll
                for (int q = 0; q < 9; ++q) wv[r * 9 + q] = pre_scale<S>(R[q], z_scale);
            }
        }
        }
        __builtin_amdgcn_wave_barrier();
        // phase 2

        if (HAS_Z) {
            // ---- phase 3
            S w[9];
            uint32_t prow = 0xFFFFFFFFu;
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
                const uint32_t rowj = LEAN ? row_of(cur.id[j]) : row[j], camj = LEAN ? cam_of(cur.id[j]) : cam[j];
                if (ONE) {
#pragma unroll
                    for (int q = 0; q < 9; ++q) w[q] = w1[q];
                } else if (rowj != prow) {
                    prow = rowj;
#pragma unroll
                    for (int q = 0; q < 9; ++q) w[q] = wv[rowj * 9 + q];
                }
                u64* zc = zs + camj;
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int b = 0; b < 3; ++b) {
                        const S v = dot3<S>(vget<S>(cur.m[i * 3 + 0], j), w[b], vget<S>(cur.m[i * 3 + 1], j), w[3 + b],
                                            vget<S>(cur.m[i * 3 + 2], j), w[6 + b]);
                        lds_add_fix(&zc[(i * 3 + b) * CP], fix_of<S>(v, z_scale));
                    }
            }
        }
        __builtin_amdgcn_wave_barrier();
        // phase 3
        return vnext;
    };

#define WCOUNT() ((void)0)
#pragma unroll 1
    while (kc != NONE) {
        int t;
        vb = body(ra, rb, va, kn1, t);       // (va: rows of kc; returns the rows of kn1)
        WCOUNT();
        kc = kn1; kn1 = resolve(__builtin_amdgcn_readfirstlane(t));
        if (kc == NONE) break;
        va = body(rb, ra, vb, kn1, t);
        WCOUNT();
        kc = kn1; kn1 = resolve(__builtin_amdgcn_readfirstlane(t));
    }
    // the last workgroup to get here re-arms the range counter for the next launch (device-scope atomics only - a
    // device-scope fence would write back this XCD's whole L2)
    __syncthreads();
    if (tid == 0 && __hip_atomic_fetch_add(&sched[8], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)nwg - 1u) {
        __hip_atomic_store(&sched[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&sched[8], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (HAS_Z) {
        u64* zp = zpart + (size_t)blockIdx.x * 9 * C;          // slab layout [9][C]
#pragma unroll
        for (int q = 0; q < 9; ++q)
            for (int c = tid; c < C; c += BLOCK) zp[q * C + c] = (u64)fix_total<S>((long long)zs[q * CP + c]);
    }
}

static thread_local const int32_t* w_gate_override = nullptr;     // the redo launch of MODE 3 runs under its own gate
template <typename S, int NW, int MODE, int CP, int TRIPS, bool FB, bool ONE, bool NT>
static int launch_wsweep5(const vican_graph_t* g, const double* lamT_inv, const double* x, u64* zpart, double* lamT_out,
                          double* fx, hipStream_t st) {
    const size_t lds = (size_t)vican_wsweep_lds_bytes(g->n_cam, g->max_rows, g->storage, g->n_copy, NW);
    auto kern = wave_sweep_kernel<S, NW, MODE, CP, TRIPS, FB, ONE, NT>;
    static size_t configured = 0;       // per instantiation
    if (lds > configured) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return set_err(VICAN_ERR_LAUNCH, "%s: cannot raise dynamic LDS limit", "vican wave sweep");
        configured = lds;
    }
    VICAN_LAUNCH_SWEEP(kern, dim3(g->n_wg), dim3(NW * 64), lds, st, w_gate_override ? w_gate_override : g_vican_gate, *g, lamT_inv, x,
                       zpart, lamT_out, fx);
    return 0;
}
template <typename S, int NW, int MODE, int CP, int TRIPS, bool FB = true, bool ONE = false>
static int launch_wsweep4(const vican_graph_t* g, const double* lamT_inv, const double* x, u64* zpart, double* lamT_out,
                          double* fx, hipStream_t st) {
    // (streams larger than the caches exist only on graphs planned for >= 8 wavefronts: no NT instantiation at 4)
    if (NW >= 8 && g->stream_nt) return launch_wsweep5<S, NW, MODE, CP, TRIPS, FB, ONE, (NW >= 8)>(g, lamT_inv, x, zpart, lamT_out, fx, st);
    return launch_wsweep5<S, NW, MODE, CP, TRIPS, FB, ONE, false>(g, lamT_inv, x, zpart, lamT_out, fx, st);
}
template <typename S, int NW, int MODE, int CP, bool FB>
static int launch_wsweep3(const vican_graph_t* g, const double* lamT_inv, const double* x, u64* zpart, double* lamT_out,
                          double* fx, hipStream_t st) {
    // one row per chunk everywhere (n_chunk == n_time): the accumulator-free instantiation
    if (NW >= 8 && NW <= 12 && g->n_chunk == g->n_time)
        return launch_wsweep4<S, NW, MODE, CP, 1, FB, true>(g, lamT_inv, x, zpart, lamT_out, fx, st);
    // TRIPS = ceil(3 max_rows / 64) row items per lane: dual-block rows (MODE 0) / thirds of the given row operands (MODE 4)
    if ((MODE != 0 && MODE != 4) || 3 * g->max_rows <= 64) return launch_wsweep4<S, NW, MODE, CP, 1, FB>(g, lamT_inv, x, zpart, lamT_out, fx, st);
    if (3 * g->max_rows <= 128) return launch_wsweep4<S, NW, MODE, CP, 2, FB>(g, lamT_inv, x, zpart, lamT_out, fx, st);
    return launch_wsweep4<S, NW, MODE, CP, 3, FB>(g, lamT_inv, x, zpart, lamT_out, fx, st);
}
template <typename S, int NW, int MODE, bool FB = true>
static int launch_wsweep2(const vican_graph_t* g, const double* lamT_inv, const double* x, u64* zpart, double* lamT_out,
                          double* fx, hipStream_t st) {
    if (g->n_cam <= 256) return launch_wsweep3<S, NW, MODE, 256, FB>(g, lamT_inv, x, zpart, lamT_out, fx, st);
    if (g->n_cam <= 512) return launch_wsweep3<S, NW, MODE, 512, FB>(g, lamT_inv, x, zpart, lamT_out, fx, st);
    return launch_wsweep3<S, NW, MODE, 1024, FB>(g, lamT_inv, x, zpart, lamT_out, fx, st);
}
template <typename S, int MODE>
static int launch_wsweep1(const vican_graph_t* g, const double* lamT_inv, const double* x, u64* zpart, double* lamT_out,
                          double* fx, hipStream_t st) {
    // the per-wavefront LDS regions and the chunk cap do not depend on the launch shape: a graph planned for 12 wavefronts
    // can be swept by the 8-wavefront instantiation
    if (MODE == 3 && g->wg_waves >= 12) {
        // 12 wavefronts without the in-kernel SVD path, then the 8-wavefront kernel with it under the redo word (see FB)
        if (int rc = launch_wsweep2<S, 12, MODE, false>(g, lamT_inv, x, zpart, lamT_out, fx, st)) return rc;
        w_gate_override = (const int32_t*)((const unsigned int*)(fx + 12) + VICAN_SCHED_REDO);
        const int rc = launch_wsweep2<S, 8, MODE, true>(g, lamT_inv, x, zpart, lamT_out, fx, st);
        w_gate_override = nullptr;
        return rc;
    }
    if (g->wg_waves == 16 && (MODE == 0 || MODE == 4)) return launch_wsweep2<S, 16, MODE>(g, lamT_inv, x, zpart, lamT_out, fx, st);
    if (g->wg_waves >= 12 && MODE != 3) return launch_wsweep2<S, 12, MODE>(g, lamT_inv, x, zpart, lamT_out, fx, st);
    if (g->wg_waves >= 8) return launch_wsweep2<S, 8, MODE>(g, lamT_inv, x, zpart, lamT_out, fx, st);
    return launch_wsweep2<S, 4, MODE>(g, lamT_inv, x, zpart, lamT_out, fx, st);
}
template <int MODE>
static int dispatch_wsweep(const vican_graph_t* g, const double* lamT_inv, const double* x, u64* zpart, double* lamT_out,
                           double* fx, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (g->storage == VICAN_STORE_F32) return launch_wsweep1<float, MODE>(g, lamT_inv, x, zpart, lamT_out, fx, st);
    return launch_wsweep1<double, MODE>(g, lamT_inv, x, zpart, lamT_out, fx, st);
}
#endif

#define WSWEEP_PART_ARGS const vican_graph_t* g, const double* lamT_inv, const double* x, u64* zpart, double* lamT_out, \
                         double* fx, void* stream
#if defined(VICAN_WSWEEP_PART)
#define WSWEEP_CAT2(a, b) a##b
#define WSWEEP_CAT(a, b) WSWEEP_CAT2(a, b)
extern "C" __attribute__((visibility("hidden"))) int WSWEEP_CAT(vican_wsweep_part_, VICAN_WSWEEP_PART)(WSWEEP_PART_ARGS) {
    return dispatch_wsweep<VICAN_WSWEEP_PART>(g, lamT_inv, x, zpart, lamT_out, fx, stream);
}
#else
#if defined(VICAN_WSWEEP_SPLIT)
extern "C" {
__attribute__((visibility("hidden"))) int vican_wsweep_part_0(WSWEEP_PART_ARGS);
__attribute__((visibility("hidden"))) int vican_wsweep_part_1(WSWEEP_PART_ARGS);
__attribute__((visibility("hidden"))) int vican_wsweep_part_3(WSWEEP_PART_ARGS);
__attribute__((visibility("hidden"))) int vican_wsweep_part_4(WSWEEP_PART_ARGS);
}
#endif
// entry used by vican_sweep.hip's dispatcher for graphs in the wave layout
extern "C" __attribute__((visibility("hidden"))) int vican_wsweep(int mode, WSWEEP_PART_ARGS) {
    if (!g->idx16) return set_err(VICAN_ERR_ARG, "%s: the wave layout needs vican_graph_t.idx16 (vican_pack_idx16)", "vican wave sweep");
#if defined(VICAN_WSWEEP_SPLIT)
    if (mode == 0) return vican_wsweep_part_0(g, lamT_inv, x, zpart, lamT_out, fx, stream);
    if (mode == 1) return vican_wsweep_part_1(g, lamT_inv, x, zpart, lamT_out, fx, stream);
    if (mode == 3) return vican_wsweep_part_3(g, lamT_inv, x, zpart, lamT_out, fx, stream);
    if (mode == 4) return vican_wsweep_part_4(g, lamT_inv, x, zpart, lamT_out, fx, stream);
#else
    if (mode == 0) return dispatch_wsweep<0>(g, lamT_inv, x, zpart, lamT_out, fx, stream);
    if (mode == 1) return dispatch_wsweep<1>(g, lamT_inv, x, zpart, lamT_out, fx, stream);
    if (mode == 3) return dispatch_wsweep<3>(g, lamT_inv, x, zpart, lamT_out, fx, stream);
    if (mode == 4) return dispatch_wsweep<4>(g, lamT_inv, x, zpart, lamT_out, fx, stream);
#endif
    return set_err(VICAN_ERR_ARG, "%s: this sweep mode needs the block layout", "vican wave sweep");
}
#endif
