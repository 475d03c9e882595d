// vican_cgres.hip - the whole translation CG (reference bipgo.py:476-478: scipy.sparse.linalg.cg on the normal equations)
// as ONE cooperative launch, for graphs that are small enough to be latency-bound.
//
// On a capture-sized graph (large_shop: 340 cameras, 10 000 timesteps, 40 000 merged edges) a CG iteration of the
// multi-kernel path (vican_cg_iter_local + vican_cg_iter_finish) is four dependent launches of 5-8 us each, none of which
// moves more than a few hundred kilobytes: 23.6 us per iteration, 0.40 ms per solve - all of it launch latency.  Here the
// grid stays resident for the whole solve:
//   * a workgroup owns a contiguous range of chunks (whole timestep rows) for all iterations; the timestep vectors
//     x_t, r_t, p_t, q_t of its rows are touched by this workgroup only, so they live in global memory behind the
//     compute unit's own L1 without any cross-workgroup coherence traffic;
//   * the camera vectors (3C numbers each) are replicated: every workgroup keeps p_c, r_c, x_c in LDS and performs
//     the same camera-side updates from the same inputs in the same order, so they agree to the bit;
//   * two device-side grid barriers per iteration carry what has to cross workgroups: the fixed-point camera partials
//     sum_t w p_t (every workgroup ADDS its sums into one of two global accumulator sets with integer atomics - exact, any
//     order - and everybody reads the 3C totals behind the barrier; round 5: slab per workgroup, barrier, a slice folded per
//     workgroup, barrier) and the partial dot products / maxima.  Everything that crosses is written and read with agent-scope
//     atomics (write-through / L1-bypassing), the barrier is the relaxed counter of lanczos_cam_coop_kernel.
// Arithmetic is that of the multi-kernel path: contributions w p in f64, exact double-word fixed-point accumulation (to_fix2:
// hi word 49 bits below wmax * max(max|p_c|, max|r_t| + beta max|p_t|) as cg_begin_kernel, lo word 48 more), scipy's recurrences and its
// stopping test |r| < rtol |b| at the top of every iteration.  Only the grouping of the floating-point partial sums of
// r.r and p.q differs (per workgroup here, per 1024 elements there), i.e. the iterates agree to rounding, not to the bit.
#include "vican_sweep_common.h"

#define CGR_THREADS 256
#define CGR_NW 4

__device__ __forceinline__ void cgr_st(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double cgr_ld(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void cgr_st(u64* p, u64 v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ u64 cgr_ld(const u64* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// grid barrier: vican_grid_sync (vican_common.h) in its fenced form - a counter that only grows during a launch, with an
// agent-scope RELEASE before the arrival and ACQUIRE after the exit, and a BOUNDED spin.  On a capture-sized graph the L2
// holds little dirty data, so the write-back / invalidate these fences lower to costs 0.4 us per barrier (the cooperative
// Lanczos step on the stress graph, with 18 MB of slabs in L2, would pay 10 us); what crosses workgroups is still written
// and read with agent-scope atomics, the fences make the ordering a property of the memory model instead of the ISA.

// deterministic block reductions of up to three sums and three maxima in one pass: 4 wavefronts
struct Cgr6 { double s0, s1, s2, a, b, c; };
__device__ __forceinline__ Cgr6 cgr_reduce6(double s0, double s1, double s2, double a, double b, double c, double* red /* [24] */) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s0 += __shfl_down(s0, o, 64); s1 += __shfl_down(s1, o, 64); s2 += __shfl_down(s2, o, 64);
        a = fmax(a, __shfl_down(a, o, 64)); b = fmax(b, __shfl_down(b, o, 64)); c = fmax(c, __shfl_down(c, o, 64));
    }
    __syncthreads();                        // red may still be read by the previous reduction's consumers
    if ((threadIdx.x & 63) == 0) {
        const int wv = threadIdx.x >> 6;
        red[wv] = s0; red[4 + wv] = s1; red[8 + wv] = s2; red[12 + wv] = a; red[16 + wv] = b; red[20 + wv] = c;
    }
    __syncthreads();
    Cgr6 r;
    r.s0 = (red[0] + red[1]) + (red[2] + red[3]);
    r.s1 = (red[4] + red[5]) + (red[6] + red[7]);
    r.s2 = (red[8] + red[9]) + (red[10] + red[11]);
    r.a = fmax(fmax(red[12], red[13]), fmax(red[14], red[15]));
    r.b = fmax(fmax(red[16], red[17]), fmax(red[18], red[19]));
    r.c = fmax(fmax(red[20], red[21]), fmax(red[22], red[23]));
    return r;
}
__device__ __forceinline__ double cgr_sum(double s, double* red) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_down(s, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    return (red[0] + red[1]) + (red[2] + red[3]);
}

// LDS: camera side 8 (5 * 3C + C) bytes; per wavefront the striped row accumulators and the staging of a chunk's rows;
// the four timestep vectors of the workgroup's own rows
extern "C" int64_t vican_cg_resident_lds_bytes(int32_t n_cam, int32_t max_rows, int32_t n_copy, int32_t rows_per_wg) {
    const int64_t per_wave = (((int64_t)max_rows * 3 * (16LL * n_copy + 8)) + 15) & ~15LL;
    return 8LL * (6 * 3 * (int64_t)n_cam + n_cam) + CGR_NW * per_wave + 4LL * 24 * rows_per_wg + 256;
}
extern "C" int64_t vican_cg_resident_ws_doubles(int32_t n_cam, int32_t n_wg) { return 18LL * n_cam + 4LL * n_wg + 8 + 16; }

template <int EPL>
struct CgrEdges { uint32_t id[EPL]; double w[EPL]; };

template <int EPL, int TRIPS>
__global__ __launch_bounds__(CGR_THREADS) void cg_resident_kernel(
    vican_graph_t g, const double* __restrict__ w, const double* __restrict__ deg_t, const double* __restrict__ deg_c,
    const double* __restrict__ b_c, const double* __restrict__ b_t, double* __restrict__ x_c, double* __restrict__ x_t, u64* slab,
    double* ws, double rtol, int max_iter, double n_add, double wmax, int rows_cap, vican_cg_state_t* st, uint32_t* abort_word,
    unsigned long long spin_limit) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    __shared__ double red[24];
    const int C = g.n_cam, n3 = 3 * C, ncopy = g.n_copy, cmask = ncopy - 1, RW = g.max_rows;
    const int tid = threadIdx.x, lane = tid & 63, lane_copy = lane & cmask;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nwg = gridDim.x, wg = blockIdx.x;
    u64* qc = (u64*)lds_raw;                    // [2][3][C] planes: camera partial sums of this workgroup (hi words, lo words)
    double* pcs = (double*)(qc + 2 * n3);       // [3][C] planes: p_c
    double* rcs = pcs + n3;                     // r_c
    double* xcs = rcs + n3;                     // x_c
    double* qcs = xcs + n3;                     // q_c of the current iteration
    double* dgc = qcs + n3;                     // [C] deg_c
    double* xts = dgc + C;                      // the workgroup's own rows: x_t, r_t, p_t, q_t, [3 rows_cap] each
    double* rts = xts + 3 * (size_t)rows_cap;
    double* pts = rts + 3 * (size_t)rows_cap;
    double* qts = pts + 3 * (size_t)rows_cap;
    const size_t per_wave = (((size_t)RW * 3 * (16 * ncopy + 8)) + 15) & ~(size_t)15;
    unsigned char* wbase = (unsigned char*)(qts + 3 * (size_t)rows_cap) + (size_t)wave * per_wave;
    u64* qt = (u64*)wbase;                      // [2][RW * 3][ncopy] striped row accumulators (this wavefront's): hi, lo
    const int lo_t = 3 * RW * ncopy;
    double* dps = (double*)(qt + (size_t)2 * lo_t);           // [RW * 3] deg_t p of the chunk's rows
    // workspace: two sets (alternate iterations) of [3][3C] camera sums (hi in two halves, lo: what fix3_add accumulates) that the
    // workgroups ADD their slabs into (agent-scope integer atomics: exact, any order), [nwg][4] partials (r.r, max|r_t|, p.q, max|p_t|),
    // barrier counter
    u64* qc_acc = (u64*)ws;
    double* part = ws + 6 * n3;
    unsigned int* sync = (unsigned int*)(part + 4 * (size_t)nwg);
    unsigned int nbar = 0;
    const vican_sync_t sy = {sync, abort_word, spin_limit};
    // a barrier that was not passed (bounded spin, vican_set_barrier_abort) ends the launch: done = -1 tells the host
    auto gsync = [&]() -> bool {
        ++nbar;
        if (vican_grid_sync(sy, nbar * (unsigned)nwg, true)) return true;
        if (wg == 0 && tid == 0) { st->done = -1; st->iter = 0; }
        return false;
    };

    const int c0 = (int)(((long long)wg * g.n_chunk) / nwg), c1 = (int)(((long long)(wg + 1) * g.n_chunk) / nwg);
    const int row0 = g.chunk_row0[c0];
    const long long i0 = 3LL * row0;
    const int ni = 3 * (g.chunk_row0[c1] - row0);                       // own elements (<= 3 rows_cap)
    const uint32_t pad_cam = (uint32_t)((lane & 31) < C ? (lane & 31) : 0);
    const int j0 = (int)(((long long)wg * n3) / nwg), j1 = (int)(((long long)(wg + 1) * n3) / nwg);     // camera slice folded here

    auto load_edges = [&](CgrEdges<EPL>& e, const int kc) {
        const size_t sl = (size_t)kc * g.slots + (size_t)lane * EPL;
        if constexpr (EPL == 4) {
            const double* wk = w + (size_t)kc * g.slots + (size_t)lane * 2;          // permuted storage of w (slot_pos8)
            const uint4 a = *(const uint4*)(g.idx + sl); const double2 b0 = *(const double2*)wk, b1 = *(const double2*)(wk + 128);
            e.id[0] = a.x; e.id[1] = a.y; e.id[2] = a.z; e.id[3] = a.w; e.w[0] = b0.x; e.w[1] = b0.y; e.w[2] = b1.x; e.w[3] = b1.y;
        } else {
            const uint2 a = *(const uint2*)(g.idx + sl); const double2 b0 = *(const double2*)(w + sl);
            e.id[0] = a.x; e.id[1] = a.y; e.w[0] = b0.x; e.w[1] = b0.y;
        }
    };
    // the first two chunks of this wavefront stay in registers for the whole solve (a capture-sized graph has one or two)
    CgrEdges<EPL> e0, e1;
    {
        const int kmax = g.n_chunk - 1, ka = c0 + wave, kb = c0 + wave + CGR_NW;
        load_edges(e0, ka < kmax ? ka : kmax);
        load_edges(e1, kb < kmax ? kb : kmax);
    }

    // ---- x = 0, r = p = b
    double s = 0.0, m = 0.0, sc = 0.0, mc = 0.0;
    for (int i = tid; i < ni; i += CGR_THREADS) {
        const double b = b_t[i0 + i];
        xts[i] = 0.0; rts[i] = b; pts[i] = b; s += b * b; m = fmax(m, fabs(b));
    }
    for (int j = tid; j < n3; j += CGR_THREADS) {
        const int comp = j / C, cam = j - comp * C;
        const double b = b_c[cam * 3 + comp];
        rcs[j] = b; pcs[j] = b; xcs[j] = 0.0; sc += b * b; mc = fmax(mc, fabs(b));
        if (comp == 0) dgc[cam] = deg_c[cam];
    }
    for (int i = lane; i < 2 * lo_t; i += 64) qt[i] = 0ull;
    for (int e = j0 + tid; e < j1; e += CGR_THREADS)            // both accumulator sets start at zero (this workgroup's slice; the barrier below publishes it)
#pragma unroll
        for (int q = 0; q < 6; ++q) cgr_st(qc_acc + (size_t)q * n3 + e, 0ull);
    Cgr6 t = cgr_reduce6(s, sc, 0.0, m, mc, 0.0, red);
    if (tid == 0) { cgr_st(part + 4 * wg, t.s0); cgr_st(part + 4 * wg + 1, t.a); }
    double rr_cam = t.s1, rmax_cam = t.b, pcmax = t.b;
    if (!gsync()) return;
    t = cgr_reduce6(tid < nwg ? cgr_ld(part + 4 * tid) : 0.0, 0.0, 0.0, tid < nwg ? cgr_ld(part + 4 * tid + 1) : 0.0, 0.0, 0.0, red);
    double rr_time = t.s0, rmax_time = t.a, pmax_time = 0.0;
    double rho = 0.0, rho_prev = 0.0, bnorm2 = 0.0, atol2 = 0.0, beta = 0.0, alpha = 0.0, pq = 0.0, pmax = 0.0;
    double scale = 1.0, inv = 1.0;
    const int lob = fix2_lo_bits(n_add);
    const double lo_scale = ldexp(1.0, lob);
    int iter = 0, done = 0;

    for (int k = 0;; ++k) {
        rho = rr_cam + rr_time;
        if (k == 0) { bnorm2 = rho; atol2 = rtol * rtol * rho; }
        if (k >= max_iter) break;                                                    // scipy: range(maxiter), no test behind the last update
        if (sqrt(rho) < sqrt(atol2) || rho == 0.0) { done = 1; break; }            // scipy: norm(r) < atol, before the step
        if (!(rho == rho)) { done = -2; break; }                                     // NaN input: scipy would spin to maxiter; report instead
        const bool first = k == 0;
        beta = first ? 0.0 : rho / rho_prev;
        // fixed-point scale of this iteration (49 bits below a bound on max |w p|, as cg_begin_kernel): both node sets
        // through |p_new| <= max|r| + beta max|p| with the MEASURED maxima of the current iterate
        pmax = first ? fmax(rmax_cam, rmax_time) : fmax(rmax_cam + beta * pcmax, rmax_time + beta * pmax_time);
        scale = fix_scale(wmax * pmax, n_add, &inv, 49);
        // p = r + beta p on both node sets
        for (int j = tid; j < n3; j += CGR_THREADS) {
            if (!first) pcs[j] = mul_add_2r(beta, pcs[j], rcs[j]);
            qc[j] = 0ull; qc[n3 + j] = 0ull;
        }
        if (!first)
            for (int i = tid; i < ni; i += CGR_THREADS) pts[i] = mul_add_2r(beta, pts[i], rts[i]);
        __syncthreads();

        // ---- sweep of this workgroup's chunks: a wavefront per chunk (the arithmetic of cg_wsweep_kernel)
        double pqs = 0.0;
        auto chunk = [&](const CgrEdges<EPL>& e, const int kc) {
            const int r0 = g.chunk_row0[kc], nr3 = 3 * (g.chunk_row0[kc + 1] - r0), l0 = 3 * (r0 - row0);
            const double* pl = pts + l0;
#pragma unroll
            for (int tt = 0; tt < TRIPS; ++tt) {
                const int i = lane + 64 * tt;
                if (i < nr3) dps[i] = deg_t[r0 + i / 3] * pl[i];
            }
            uint32_t cam[EPL], row[EPL];
            double wj[EPL], pc[EPL][3], pr[EPL][3];
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
                const bool pad = e.id[j] == VICAN_PAD_SLOT;
                cam[j] = pad ? pad_cam : (e.id[j] & 0xFFFFu); row[j] = pad ? 0u : (e.id[j] >> 16);
                wj[j] = pad ? 0.0 : e.w[j];
#pragma unroll
                for (int i = 0; i < 3; ++i) pc[j][i] = pcs[i * C + cam[j]];
            }
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
                if (j == 0 || row[j] != row[j - 1]) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) pr[j][i] = pl[row[j] * 3 + i];
                } else {
#pragma unroll
                    for (int i = 0; i < 3; ++i) pr[j][i] = pr[j - 1][i];
                }
            }
            Fix2 fc[EPL][3];
            double ar[EPL][3];
            {
                double acc[3] = {0, 0, 0};
#pragma unroll
                for (int j = 0; j < EPL; ++j) {
                    if (j > 0 && row[j] != row[j - 1]) acc[0] = acc[1] = acc[2] = 0.0;
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        acc[i] += wj[j] * pc[j][i];
                        fc[j][i] = to_fix2(wj[j] * pr[j][i], scale, lo_scale);
                        ar[j][i] = acc[i];
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
#pragma unroll
                for (int i = 0; i < 3; ++i) { lds_add_fix(&qc[i * C + cam[j]], fc[j][i].hi); lds_add_fix(&qc[n3 + i * C + cam[j]], fc[j][i].lo); }
                if (j == EPL - 1 || row[j] != row[j + 1]) {
#pragma unroll
                    for (int i = 0; i < 3; ++i) {
                        const Fix2 f = to_fix2(ar[j][i], scale, lo_scale);
                        u64* a = &qt[(row[j] * 3 + i) * ncopy + lane_copy];
                        lds_add_fix(a, f.hi); lds_add_fix(a + lo_t, f.lo);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
            for (int base = 0; base < nr3 * ncopy; base += 64) {
                const int a = base + lane;
                const bool live = a < nr3 * ncopy;
                u64 sum = 0ull, slo = 0ull;
                if (live) { sum = qt[a]; slo = qt[lo_t + a]; qt[a] = 0ull; qt[lo_t + a] = 0ull; }
                sum = stripe_sum(sum, ncopy); slo = stripe_sum(slo, ncopy);
                if (live && (a & cmask) == 0) {
                    const int i = a / ncopy;
                    const double qv = dps[i] - fix2_value((long long)sum, (long long)slo, lob, inv);
                    qts[l0 + i] = qv;
                    pqs += pl[i] * qv;
                }
            }
            __builtin_amdgcn_wave_barrier();
        };
        {
            int kc = c0 + wave;
            if (kc < c1) chunk(e0, kc);
            kc += CGR_NW;
            if (kc < c1) chunk(e1, kc);
            for (kc += CGR_NW; kc < c1; kc += CGR_NW) { CgrEdges<EPL> e; load_edges(e, kc); chunk(e, kc); }
        }
        __syncthreads();
        // ---- this workgroup's camera sums INTO the iteration's accumulator set: the three integers fix3_add forms from a slab's
        // (hi, lo) words, added with agent-scope atomics - exact integer sums, so the totals are the ones a fold over the slabs
        // gives, whatever the order (round 5 wrote slabs, met at a barrier, folded a slice per workgroup and met again: one grid
        // barrier and one round of dependent loads more per iteration)
        u64* const acc = qc_acc + (size_t)(k & 1) * 3 * n3;
        for (int j = tid; j < n3; j += CGR_THREADS) {
            long long h = (long long)qc[j], l = (long long)qc[n3 + j];
            const long long c = l >> lob;
            h += c; l -= c << lob;
            __hip_atomic_fetch_add(acc + j, (u64)(h >> 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(acc + n3 + j, (u64)(h & 0xFFFFFFFFll), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_fetch_add(acc + 2 * n3 + j, (u64)l, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        const double pq_loc = cgr_sum(pqs, red);
        if (tid == 0) cgr_st(part + 4 * wg + 2, pq_loc);
        if (!gsync()) return;
        // (the OTHER set - read by everybody an iteration ago, added into an iteration from now - is zeroed here: every workgroup
        //  is past the barriers that ended those reads, and passes another one before the next adds)
        {
            u64* const nxt = qc_acc + (size_t)((k + 1) & 1) * 3 * n3;
            for (int e = j0 + tid; e < j1; e += CGR_THREADS) { cgr_st(nxt + e, 0ull); cgr_st(nxt + n3 + e, 0ull); cgr_st(nxt + 2 * n3 + e, 0ull); }
        }

        // ---- alpha, x += alpha p, r -= alpha q (camera side replicated, timestep side on the own rows)
        double sq = tid < nwg ? cgr_ld(part + 4 * tid + 2) : 0.0;               // p.q: timestep partials + camera terms
        for (int j = tid; j < n3; j += CGR_THREADS) {
            const int comp = j / C, cam = j - comp * C;
            const double q = dgc[cam] * pcs[j] - fix3_value((long long)cgr_ld(acc + j), (long long)cgr_ld(acc + n3 + j), (long long)cgr_ld(acc + 2 * n3 + j), lob, inv);
            qcs[j] = q; sq += pcs[j] * q;
        }
        pq = cgr_sum(sq, red);
        alpha = rho / pq;
        double sr = 0.0, mr = 0.0, mpc = 0.0;
        for (int j = tid; j < n3; j += CGR_THREADS) {
            const double p = pcs[j];
            xcs[j] = mul_add_2r(alpha, p, xcs[j]); mpc = fmax(mpc, fabs(p));
            const double r = mul_add_2r(-alpha, qcs[j], rcs[j]);
            rcs[j] = r; sr += r * r; mr = fmax(mr, fabs(r));
        }
        double mp = 0.0;
        s = 0.0; m = 0.0;
        for (int i = tid; i < ni; i += CGR_THREADS) {
            const double pv = pts[i];
            xts[i] = mul_add_2r(alpha, pv, xts[i]); mp = fmax(mp, fabs(pv));
            const double r = mul_add_2r(-alpha, qts[i], rts[i]);
            rts[i] = r; s += r * r; m = fmax(m, fabs(r));
        }
        t = cgr_reduce6(s, sr, 0.0, m, mp, fmax(mr, 0.0), red);
        if (tid == 0) { cgr_st(part + 4 * wg, t.s0); cgr_st(part + 4 * wg + 1, t.a); cgr_st(part + 4 * wg + 3, t.b); }
        rr_cam = t.s1; rmax_cam = t.c;
        // (max |p_c| of the iterate: all lanes hold the same camera vectors - one more wave-level maximum, no exchange)
        {
            double v = mpc;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o, 64));
            __syncthreads();
            if (lane == 0) red[wave] = v;
            __syncthreads();
            pcmax = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
        }
        if (!gsync()) return;
        t = cgr_reduce6(tid < nwg ? cgr_ld(part + 4 * tid) : 0.0, 0.0, 0.0, tid < nwg ? cgr_ld(part + 4 * tid + 1) : 0.0,
                        tid < nwg ? cgr_ld(part + 4 * tid + 3) : 0.0, 0.0, red);
        rr_time = t.s0; rmax_time = t.a; pmax_time = t.b;
        rho_prev = rho;
        ++iter;
    }

    // ---- results: the solution and the state (workgroup 0); the barrier counter is re-armed by the last one out
    for (int i = tid; i < ni; i += CGR_THREADS) x_t[i0 + i] = xts[i];
    if (wg == 0) {
        for (int j = tid; j < n3; j += CGR_THREADS) { const int comp = j / C, cam = j - comp * C; x_c[cam * 3 + comp] = xcs[j]; }
        if (tid == 0) {
            st->rho = rho; st->rho_prev = rho_prev; st->pq = pq; st->alpha = alpha; st->beta = beta; st->bnorm2 = bnorm2;
            st->atol2 = atol2; st->rr_cam = rr_cam; st->pq_time = 0.0; st->rr_time = rr_time; st->rmax_cam = rmax_cam;
            st->rmax_time = rmax_time; st->pmax = pmax; st->qscale = scale; st->qinv = inv; st->wmax = wmax;
            st->pmax_time = pmax_time; st->iter = iter; st->done = done; st->first = iter == 0; st->lo_bits = lob;
        }
    }
    __syncthreads();
    if (tid == 0 && __hip_atomic_fetch_add(&sync[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)nwg - 1u) {
        __hip_atomic_store(&sync[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&sync[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

// The whole CG solve in one launch (wave-layout graphs whose grid is co-resident: n_wg <= compute units).
// x_c [C][3], x_t [T][3]: solution; slab: n_wg * 6C 64-bit words (double-word camera partials); ws: vican_cg_resident_ws_doubles doubles (the barrier counter
// in it is zeroed in-stream by every call); rows_per_wg: max rows of one workgroup's chunk range [n_chunk b / n_wg,
// n_chunk (b + 1) / n_wg); st: receives the final state (iter, done, rho, bnorm2, ...).
extern "C" int vican_cg_resident(const vican_graph_t* g, const double* w, const double* deg_t, const double* deg_c, const double* b_c,
                                 const double* b_t, double* x_c, double* x_t, void* slab, double* ws, double rtol, int32_t max_iter,
                                 double n_add, double wmax, int32_t rows_per_wg, vican_cg_state_t* st, void* stream) {
    if (int rc = vican_check_graph(g, "vican_cg_resident")) return rc;
    if (g->layout != VICAN_LAYOUT_WAVE) return set_err(VICAN_ERR_ARG, "vican_cg_resident: wave layout only");
    if (!w || !deg_t || !deg_c || !b_c || !b_t || !x_c || !x_t || !slab || !ws || !st || max_iter < 0 || rows_per_wg < 1)
        return set_err(VICAN_ERR_ARG, "vican_cg_resident: bad argument");
    if (g->n_chunk == 0) return set_err(VICAN_ERR_ARG, "vican_cg_resident: graph without edges");
    int dev = 0, n_cu = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
        return set_err(VICAN_ERR_LAUNCH, "vican_cg_resident: cannot query the device");
    if (g->n_wg > n_cu || g->n_wg > CGR_THREADS)
        return set_err(VICAN_ERR_CAPACITY, "%s: more workgroups than compute units (or than 256): the grid would not be co-resident", "vican_cg_resident");
    const size_t lds = (size_t)vican_cg_resident_lds_bytes(g->n_cam, g->max_rows, g->n_copy, rows_per_wg);
    if ((int64_t)lds > vican_lds_limit_bytes()) return set_err(VICAN_ERR_CAPACITY, "vican_cg_resident: camera vectors / own rows do not fit in LDS");
    const int epl = g->slots / 64, trips = (3 * g->max_rows + 63) / 64;
    if (trips > 3) return set_err(VICAN_ERR_CAPACITY, "vican_cg_resident: more than 64 rows per chunk");
    hipStream_t s = (hipStream_t)stream;
    // the barrier counter (behind the [3C] sums and the [n_wg][4] partials): armed here, so that a launch that was torn down
    // cannot make the next one pass its barriers early
    if (hipMemsetAsync(ws + 9 * (size_t)g->n_cam + 4 * (size_t)g->n_wg, 0, 8, s) != hipSuccess)
        return set_err(VICAN_ERR_LAUNCH, "vican_cg_resident: memset failed");
#define CGR_LAUNCH(E_, T_)                                                                                                \
    do {                                                                                                                  \
        auto kern = cg_resident_kernel<E_, T_>;                                                                           \
        static size_t conf = 0;                                                                                           \
        if (lds > conf) { hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); conf = lds; } \
        if (int rc_ = vican_coresident_ok((const void*)kern, CGR_THREADS, lds, g->n_wg, "vican_cg_resident")) return rc_;  \
        hipLaunchKernelGGL(kern, dim3(g->n_wg), dim3(CGR_THREADS), lds, s, *g, w, deg_t, deg_c, b_c, b_t, x_c, x_t, (u64*)slab, ws,  \
                           rtol, (int)max_iter, n_add, wmax, (int)rows_per_wg, st, g_vican_abort_word, g_vican_sync_ticks); \
    } while (0)
    if (epl == 4) { if (trips <= 1) CGR_LAUNCH(4, 1); else if (trips == 2) CGR_LAUNCH(4, 2); else CGR_LAUNCH(4, 3); }
    else          { if (trips <= 1) CGR_LAUNCH(2, 1); else if (trips == 2) CGR_LAUNCH(2, 2); else CGR_LAUNCH(2, 3); }
#undef CGR_LAUNCH
    LAUNCH_CHECK("vican_cg_resident");
    return VICAN_OK;
}
