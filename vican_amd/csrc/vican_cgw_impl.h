// vican_cgw_impl.h - the CG Laplacian product of the wave layout as a device function (see vican_wtrans.hip for the design):
// shared by cg_wsweep_kernel (vican_wtrans.hip) and the single-launch product over camera tiles (vican_tcg.hip).
#pragma once
#include "vican_sweep_common.h"

template <int EPL>
struct CgWRegs { double w[EPL]; uint32_t id[EPL]; };

// CP: stride of the camera planes in LDS (plane_stride(): 256 / 512 / 1024 entries) - a compile-time constant, so that the
// nine planes (hi and lo words of the three sums, the three components of p_c) are reached with IMMEDIATE offsets from one
// address register per edge: the kernel is bound by VALU issue (310 instructions per wavefront and chunk before, of which ~35
// were plane address arithmetic and ~100 the three separate wave reductions of the one-row path - wave_total3).
// NT: the edge stream is read with non-temporal loads (graphs whose streams exceed the caches, vican_graph_t.stream_nt) - a
// COMPILE-TIME choice: with a run-time branch around the two forms of the loads the compiler's vmcnt bookkeeping collapses at
// the join (s_waitcnt vmcnt(0..2) where 5-11 younger loads may stay in flight: every prefetch landed before its chunk's
// predecessor was processed - found in round 4 in the ISA of all wave-layout kernels; round 3 had introduced the branch).
// (the body of cg_wsweep_kernel as a device function: workgroup `wg` of `nwg` over the chunks of g - the plain kernel passes
//  blockIdx.x / gridDim.x, the camera-tile kernel of vican_tcg.hip one tile's share of its grid)
template <int NW, int EPL, int TRIPS, int CP, bool NT>
__device__ __forceinline__ void cg_wsweep_impl(const vican_graph_t& g, const double* __restrict__ w,
                                               const double* __restrict__ deg_t, const double* __restrict__ p_c,
                                               const double* __restrict__ r_t, double* __restrict__ p_t,
                                               double* __restrict__ q_t, u64* __restrict__ qc_part,
                                               double* __restrict__ pq_part, const vican_cg_state_t* __restrict__ st,
                                               const int partial, const int wg, const int nwg) {
    // partial != 0 (camera tiles, vican_cg_sweep_partial - as in cg_sweep_kernel, vican_trans.hip): g holds the edges of ONE camera
    // tile; p_t is read as it is (already updated; r_t / deg_t are not used: any readable pointers), q_t receives the tile's
    // row sums sum_{c in tile} w p_c alone, no p.q partial; the camera sums of the tile's cameras are complete either way
    extern __shared__ __align__(16) unsigned char lds_raw[];
    __shared__ double red[16];
    __shared__ int s_ticket;
    if (st->done) return;
    const int C = g.n_cam, ncopy = g.n_copy, cmask = ncopy - 1, RW = g.max_rows;
    const int tid = threadIdx.x, lane = tid & 63, lane_copy = lane & cmask;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    u64* qc = (u64*)lds_raw;                                   // [2][3][CP] planes (hi words, lo words), shared by the workgroup
    double* pcs = (double*)(qc + 6 * CP);                      // [3][CP] planes
    const size_t per_wave = (((size_t)RW * 3 * (16 * ncopy + 16)) + 15) & ~(size_t)15;
    unsigned char* wbase = (unsigned char*)(pcs + 3 * CP) + (size_t)wave * per_wave;
    u64* qt = (u64*)wbase;                                     // [2][RW * 3][ncopy] striped row accumulators (this wave's): hi, lo
    double* pts = (double*)(qt + (size_t)2 * RW * 3 * ncopy);  // [RW * 3] p of the chunk's rows
    double* dps = pts + RW * 3;                                // [RW * 3] deg * p
    const uint32_t pad_cam = (uint32_t)((lane & 31) < C ? (lane & 31) : 0);
    const bool upd = !st->first && !partial;
    const double beta = st->beta, scale = st->qscale, inv = st->qinv;
    const int lob = st->lo_bits;
    const double lo_scale = ldexp(1.0, lob);
    constexpr int lo_c = 3 * CP;                               // offset of the lo planes behind the hi planes
    const int lo_t = 3 * RW * ncopy;
    for (int i = tid; i < 3 * C; i += NW * 64) pcs[(i % 3) * CP + i / 3] = p_c[i];
    for (int i = tid; i < 6 * CP; i += NW * 64) qc[i] = 0ull;
    for (int i = lane; i < 2 * lo_t; i += 64) qt[i] = 0ull;
    const int c0 = (int)(((long long)wg * g.n_chunk) / nwg);
    const int c1 = (int)(((long long)(wg + 1) * g.n_chunk) / nwg);
    const int kmax = g.n_chunk - 1;

    auto load_rows = [&](int k) -> int2 { k = k < kmax ? k : kmax; return *(const int2*)(g.chunk_row0 + k); };
    auto load_edges = [&](CgWRegs<EPL>& e, int k) {
        k = k < kmax ? k : kmax;
        const size_t s = (size_t)k * g.slots + (size_t)lane * EPL;
        if (EPL == 4) {
            double2 a, b;
            const double* wk = w + (size_t)k * g.slots + (size_t)lane * 2;        // permuted storage (slot_pos8): dense 16-byte loads
            load_idx16<EPL, NT>(e.id, g.idx16 + s);                                // (2-byte index of the wave layout)
            if (NT) { a = stream_load((const double2*)wk); b = stream_load((const double2*)(wk + 128)); }
            else { a = *(const double2*)wk; b = *(const double2*)(wk + 128); }
            e.w[0] = a.x; e.w[1] = a.y; e.w[2] = b.x; e.w[3] = b.y;
        } else {
            double2 a;
            load_idx16<EPL, NT>(e.id, g.idx16 + s);
            if (NT) a = stream_load((const double2*)(w + s)); else a = *(const double2*)(w + s);
            e.w[0] = a.x; e.w[1] = a.y;
        }
    };
    // row values of a chunk: lane + 64 t < 3 nrows holds (p_t, r_t, deg_t) of one (row, component) item
    struct RowVals { double p[TRIPS], r[TRIPS], d[TRIPS]; };
    auto load_rowvals = [&](RowVals& rv, const int2 vrow) {
        const int r0 = __builtin_amdgcn_readfirstlane(vrow.x), n3 = 3 * (__builtin_amdgcn_readfirstlane(vrow.y) - r0);
#pragma unroll
        for (int t = 0; t < TRIPS; ++t) {
            int i = lane + 64 * t;
            i = i < n3 ? i : 0;                                 // lanes without an item re-read item 0 (never used)
            const size_t gi = (size_t)r0 * 3 + i;
            rv.p[t] = p_t[gi]; rv.r[t] = r_t[gi]; rv.d[t] = deg_t[r0 + i / 3];
        }
    };
    if (tid == 0) s_ticket = c0 + 3 * NW;
    __syncthreads();

    // Chunks of the workgroup's range are handed to its wavefronts by an LDS ticket (the first three rounds are static):
    // with a static interleaved assignment the wavefronts of a workgroup finished 64-84 us into an 86 us launch (the SIMD
    // arbiter favours some), and the workgroup ends with its last wavefront.  A ticket is drawn two bodies before its chunk is
    // processed (row bounds are requested two chunks ahead), its value is read at the end of the body that drew it.
    int k = c0 + wave, kb = k + NW, kc = k + 2 * NW;
    CgWRegs<EPL> ea, eb;
    RowVals ra, rb;
    int2 v0 = load_rows(k), v1 = load_rows(kb), v2;
    load_edges(ea, k);
    load_rowvals(ra, v0);
    double pq = 0.0;

    // body: chunk k (edges `cur`, row values `rv`, row bounds `vrow`); requests the edge words and row values of chunk
    // k + NW (row bounds `vnext`, loaded a body ago) and the row bounds of chunk k + 2 NW (returned)
    auto body = [&](CgWRegs<EPL>& cur, CgWRegs<EPL>& nxt, RowVals& rv, RowVals& rvn, const int2 vrow, const int2 vnext, const int k_next,
                    const int k_after) -> int2 {
        const int r0 = __builtin_amdgcn_readfirstlane(vrow.x), n3 = 3 * (__builtin_amdgcn_readfirstlane(vrow.y) - r0);
        const int2 vnn = load_rows(k_after);
        load_rowvals(rvn, vnext);
        __builtin_amdgcn_sched_barrier(0);
        load_edges(nxt, k_next);
        // row bounds (a wait), issue of the prefetches
        if (n3 == 3) {
            // ONE row in the chunk (dense rows: every lane's slots belong to it): no LDS staging, no row accumulators, no fold -
            // lanes 0..2 hold p / r / deg of the row's three components, the row sum sum_c w p_c is a wave reduction of the
            // lanes' partial sums (DPP, fixed order: deterministic; plain f64 like scipy's row sum), camera side as below
            const double pn = upd ? mul_add_2r(beta, rv.p[0], rv.r[0]) : rv.p[0];
            if (upd && lane < 3) p_t[(size_t)r0 * 3 + lane] = pn;
            const double prow[3] = {lane_bcast(pn, 0), lane_bcast(pn, 1), lane_bcast(pn, 2)};
            uint32_t cam[EPL];
            double wj[EPL], acc[3] = {0, 0, 0};
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
                const bool pad = cur.id[j] == VICAN_PAD_SLOT;
                cam[j] = pad ? pad_cam : (cur.id[j] & 0xFFFFu);
                wj[j] = pad ? 0.0 : cur.w[j];
#pragma unroll
                for (int i = 0; i < 3; ++i) acc[i] += wj[j] * pcs[i * CP + cam[j]];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const Fix2 f = to_fix2(wj[j] * prow[i], scale, lo_scale);
                    lds_add_fix(&qc[i * CP + cam[j]], f.hi); lds_add_fix(&qc[lo_c + i * CP + cam[j]], f.lo);
                }
            }
            const double srow = wave_total3(acc[0], acc[1], acc[2], lane);     // lanes 0, 1, 2: the row sums of the three components
            if (lane < 3) {
                const double qv = partial ? srow : rv.d[0] * pn - srow;
                q_t[(size_t)r0 * 3 + lane] = qv;
                pq += pn * qv;
            }
            return vnn;
        }
        // commit this chunk's rows: p (updated), deg p into the wavefront's staging; the updated p back to memory
#pragma unroll
        for (int t = 0; t < TRIPS; ++t) {
            const int i = lane + 64 * t;
            if (i < n3) {
                const double p = upd ? mul_add_2r(beta, rv.p[t], rv.r[t]) : rv.p[t];
                pts[i] = p; dps[i] = rv.d[t] * p;
                if (upd) p_t[(size_t)r0 * 3 + i] = p;
            }
        }
        __builtin_amdgcn_wave_barrier();
        // commit
        // edges.  LDS operations of a wavefront return in order, so a read issued after an atomic waits for it: all the
        // reads of the lane's EPL edges first (camera values; the row's p where the row changes), then the arithmetic,
        // then nothing but atomics - camera contributions one by one, same-row contributions of a lane pre-summed.
        uint32_t cam[EPL], row[EPL];
        double wj[EPL], pc[EPL][3], pr[EPL][3];
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            const bool pad = cur.id[j] == VICAN_PAD_SLOT;
            cam[j] = pad ? pad_cam : (cur.id[j] & 0xFFFFu); row[j] = pad ? 0u : (cur.id[j] >> 16);
            wj[j] = pad ? 0.0 : cur.w[j];
#pragma unroll
            for (int i = 0; i < 3; ++i) pc[j][i] = pcs[i * CP + cam[j]];
        }
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            if (j == 0 || row[j] != row[j - 1]) {               // a lane's slots are consecutive edges: mostly one row
#pragma unroll
                for (int i = 0; i < 3; ++i) pr[j][i] = pts[row[j] * 3 + i];
            } else {
#pragma unroll
                for (int i = 0; i < 3; ++i) pr[j][i] = pr[j - 1][i];
            }
        }
        Fix2 fc[EPL][3];
        double ar[EPL][3];                                       // the lane's running row sum after edge j
        {
            double acc[3] = {0, 0, 0};
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
                if (j > 0 && row[j] != row[j - 1]) acc[0] = acc[1] = acc[2] = 0.0;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    acc[i] += wj[j] * pc[j][i];
                    fc[j][i] = to_fix2(wj[j] * pr[j][i], scale, lo_scale);
                    ar[j][i] = acc[i];                           // converted only where the lane's run of this row ends
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                lds_add_fix(&qc[i * CP + cam[j]], fc[j][i].hi);
                lds_add_fix(&qc[lo_c + i * CP + cam[j]], fc[j][i].lo);
            }
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
        // edges
        // fold this chunk's row sums (exact integer sums of the stripes), q_t, p.q: one lane per (item, stripe) word,
        // the n_copy words of an item summed across neighbouring lanes
        for (int base = 0; base < n3 * ncopy; base += 64) {
            const int a = base + lane;
            const bool live = a < n3 * ncopy;
            u64 sum = 0ull, slo = 0ull;
            if (live) {                                          // read and clear in one LDS operation each (ds_wrxchg_rtn_b64)
                sum = __hip_atomic_exchange(&qt[a], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                slo = __hip_atomic_exchange(&qt[lo_t + a], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            }
            sum = stripe_sum(sum, ncopy); slo = stripe_sum(slo, ncopy);
            if (live && (a & cmask) == 0) {
                const int i = a / ncopy;
                const double sv = fix2_value((long long)sum, (long long)slo, lob, inv);
                const double qv = partial ? sv : dps[i] - sv;
                q_t[(size_t)r0 * 3 + i] = qv;
                pq += pts[i] * qv;
            }
        }
        __builtin_amdgcn_wave_barrier();
        // fold
        return vnn;
    };
    auto draw = [&]() -> int {
        int t = 0;
        if (lane == 0) t = __hip_atomic_fetch_add(&s_ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return t;                                              // (valid in lane 0; read with readfirstlane after the body)
    };
#pragma unroll 1
    while (k < c1) {
        int t = draw();
        v2 = body(ea, eb, ra, rb, v0, v1, kb, kc);             // chunk k; v2 = row bounds of kc
        k = kb; kb = kc; kc = __builtin_amdgcn_readfirstlane(t);
        if (k >= c1) break;
        t = draw();
        v0 = body(eb, ea, rb, ra, v1, v2, kb, kc);             // chunk k (the former kb); v0 = row bounds of the new kc
        k = kb; kb = kc; kc = __builtin_amdgcn_readfirstlane(t);
        // rotate the row-bound registers: (v0, v1, v2) hold the bounds of (kb, -, k) -> bring them back to (k, kb)
        const int2 tmp = v0; v0 = v2; v1 = tmp;
    }
    __syncthreads();
    for (int pl = 0; pl < 6; ++pl)                             // the slab keeps planes of stride C ([2][3][C]: cg_fold_kernel)
        for (int i = tid; i < C; i += NW * 64) qc_part[((size_t)wg * 6 + pl) * C + i] = qc[pl * CP + i];
    const double t = block_sum(pq, red);
    if (tid == 0 && !partial) pq_part[wg] = t;
}

