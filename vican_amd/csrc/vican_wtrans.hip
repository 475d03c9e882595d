// vican_wtrans.hip - the CG Laplacian product q = A p (reference bipgo.py:476-478, one application per scipy cg
// iteration) on graphs in the WAVE layout: one wavefront per chunk, no workgroup barrier in the loop.
//
// Same arithmetic as cg_sweep_kernel (vican_trans.hip): contributions w p in f64, exact DOUBLE-WORD fixed-point accumulation
// (to_fix2, vican_sweep_common.h: hi and lo words in separate planes, so both keep the bank pattern of the slot order;
// camera sums in the workgroup's LDS table, row sums in the wavefront's own LDS region), q_t = deg_t p_t - sum_c w p_c,
// the timestep part of p.q, and the update p_t <- r_t + beta p_t folded into the row loads.  What changes is the
// schedule: the block kernel's 12 wavefronts share a 3072-slot chunk and meet at a barrier per chunk; here a wavefront
// owns a 256-slot chunk (whole rows), stages its rows' p and deg p in its private LDS region and folds its own row sums -
// LDS operations of one wavefront execute in order, so nothing has to be synchronised.  Row bounds are requested two
// chunks ahead and row values (p_t, r_t, deg_t) one chunk ahead, edge words (index + weight, 12 B per edge) one chunk
// ahead; every vector-memory operation is unconditional (clamped indices) so that the compiler's vmcnt waits are exact.
// A workgroup owns a contiguous range of chunks and its wavefront i takes chunks c0 + i, c0 + i + NW, ...
#include "vican_sweep_common.h"

extern "C" int64_t vican_cg_wsweep_lds_bytes(int32_t n_cam, int32_t max_rows, int32_t n_copy, int32_t n_waves) {
    const int64_t per_wave = (((int64_t)max_rows * 3 * (16LL * n_copy + 16)) + 15) & ~15LL;
    return 72LL * plane_stride(n_cam) + (int64_t)n_waves * per_wave + 256;       // nine camera planes of compile-time stride
}


#include "vican_cgw_impl.h"

template <int NW, int EPL, int TRIPS, int CP, bool NT>
__global__ __launch_bounds__(NW * 64) void cg_wsweep_kernel(vican_graph_t g, const double* __restrict__ w,
                                                            const double* __restrict__ deg_t, const double* __restrict__ p_c,
                                                            const double* __restrict__ r_t, double* __restrict__ p_t,
                                                            double* __restrict__ q_t, u64* __restrict__ qc_part,
                                                            double* __restrict__ pq_part, const vican_cg_state_t* __restrict__ st,
                                                            const int partial) {
    cg_wsweep_impl<NW, EPL, TRIPS, CP, NT>(g, w, deg_t, p_c, r_t, p_t, q_t, qc_part, pq_part, st, partial, (int)blockIdx.x, (int)gridDim.x);
}

// ---------------------------------------------------------------------------
// The same product on graphs whose chunks hold exactly ONE row each (n_chunk == n_time: dense rows - the stress graph, 250
// cameras per timestep).  Round-4 counters (profiles/r04_cg_counters.json) showed the general kernel bound by VALU issue and
// memory latency, not by the LDS atomic pipe: 310 VALU wave-instructions per chunk, 70 % VALU-busy at three wavefronts per SIMD,
// removing every camera atomic gained 2 %.  This variant has no row accumulators, no staging, no fold and no row bounds
// (chunk k is row k), takes the three row sums through ONE shared butterfly (wave_total3), reaches the nine camera planes with
// immediate offsets, and prefetches TWO chunks ahead (three register sets of 12 + 6 VGPRs; the wait for a chunk's data was
// 950 of 3700 cycles per chunk with one).  Same arithmetic and summation structure as the one-row path of cg_wsweep_kernel
// (exact double-word camera sums; the row sum a fixed-order f64 wave reduction).
// W32: the weights come as float32 (vican_graph_t.w32: exact copies, plain slot order - 16 bytes per lane instead of 32), are kept
// as loaded in the register set and converted where they are used (a conversion at the load would wait for it)
template <int EPL, bool W32>
struct CgW1Regs { typename std::conditional<W32, float, double>::type w[EPL]; uint32_t id[EPL]; };

template <int NW, int EPL, int CP, bool NT, bool W32>
__global__ __launch_bounds__(NW * 64) void cg_wsweep1_kernel(vican_graph_t g, const double* __restrict__ w,
                                                             const double* __restrict__ deg_t, const double* __restrict__ p_c,
                                                             const double* __restrict__ r_t, double* __restrict__ p_t,
                                                             double* __restrict__ q_t, u64* __restrict__ qc_part,
                                                             double* __restrict__ pq_part, const vican_cg_state_t* __restrict__ st) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    __shared__ double red[16];
    __shared__ int s_ticket;
    if (st->done) return;
    const int C = g.n_cam;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    u64* qc = (u64*)lds_raw;                                   // [2][3][CP] planes (hi words, lo words)
    double* pcs = (double*)(qc + 6 * CP);                      // [3][CP] planes
    constexpr int lo_c = 3 * CP;
    const uint32_t pad_cam = (uint32_t)((lane & 31) < C ? (lane & 31) : 0);
    const bool upd = !st->first;
    const double beta = st->beta, scale = st->qscale;
    const double lo_scale = ldexp(1.0, st->lo_bits);
    for (int i = tid; i < 3 * C; i += NW * 64) pcs[(i % 3) * CP + i / 3] = p_c[i];
    for (int i = tid; i < 6 * CP; i += NW * 64) qc[i] = 0ull;
    const int c0 = (int)(((long long)blockIdx.x * g.n_chunk) / gridDim.x);
    const int c1 = (int)(((long long)(blockIdx.x + 1) * g.n_chunk) / gridDim.x);
    const int kmax = g.n_chunk - 1;
    const int l3 = lane < 3 ? lane : 0;
    // deg_t[k] has a wave-uniform address: left alone it becomes a SCALAR load, which under a saturated memory system takes
    // thousands of cycles and shares lgkmcnt with the LDS operations (returning out of order, it turns every LDS wait into
    // lgkmcnt(0)) - an opaque per-lane zero keeps it a vector load
    int vzero;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vzero));

    struct RowVals { double p, r, d; };                        // lanes 0..2: component `lane` of the row's p_t, r_t; deg_t
    auto load_edges = [&](CgW1Regs<EPL, W32>& e, int k) {
        k = k < kmax ? k : kmax;
        const size_t s = (size_t)k * g.slots + (size_t)lane * EPL;
        if constexpr (EPL == 4 && W32) {
            uint2 t2; float4 a;
            if (NT) { t2 = stream_load((const uint2*)(g.idx16 + s)); a = stream_load((const float4*)(g.w32 + s)); }
            else { t2 = *(const uint2*)(g.idx16 + s); a = *(const float4*)(g.w32 + s); }
            const uint32_t h[4] = {t2.x & 0xFFFFu, t2.x >> 16, t2.y & 0xFFFFu, t2.y >> 16};
#pragma unroll
            for (int j = 0; j < 4; ++j) e.id[j] = h[j] == 0xFFFFu ? VICAN_PAD_SLOT : h[j];
            e.w[0] = a.x; e.w[1] = a.y; e.w[2] = a.z; e.w[3] = a.w;
        } else if constexpr (EPL == 4) {
            uint2 t2; double2 a, b;
            const double* wk = w + (size_t)k * g.slots + (size_t)lane * 2;        // permuted storage (slot_pos8): dense 16-byte loads
            // (2-byte camera indices of one-row graphs, vican_graph_t.idx16: 8 instead of 16 bytes per lane)
            if (NT) { t2 = stream_load((const uint2*)(g.idx16 + s)); a = stream_load((const double2*)wk); b = stream_load((const double2*)(wk + 128)); }
            else { t2 = *(const uint2*)(g.idx16 + s); a = *(const double2*)wk; b = *(const double2*)(wk + 128); }
            const uint32_t h[4] = {t2.x & 0xFFFFu, t2.x >> 16, t2.y & 0xFFFFu, t2.y >> 16};
#pragma unroll
            for (int j = 0; j < 4; ++j) e.id[j] = h[j] == 0xFFFFu ? VICAN_PAD_SLOT : h[j];
            e.w[0] = a.x; e.w[1] = a.y; e.w[2] = b.x; e.w[3] = b.y;
        } else {
            uint32_t t1; double2 a;
            if (NT) { t1 = __builtin_nontemporal_load((const uint32_t*)(g.idx16 + s)); a = stream_load((const double2*)(w + s)); }
            else { t1 = *(const uint32_t*)(g.idx16 + s); a = *(const double2*)(w + s); }
            const uint32_t h0 = t1 & 0xFFFFu, h1 = t1 >> 16;
            e.id[0] = h0 == 0xFFFFu ? VICAN_PAD_SLOT : h0; e.id[1] = h1 == 0xFFFFu ? VICAN_PAD_SLOT : h1;
            e.w[0] = a.x; e.w[1] = a.y;
        }
    };
    auto load_rowvals = [&](RowVals& rv, int k) {
        k = k < kmax ? k : kmax;
        const size_t gi = (size_t)k * 3 + l3;
        rv.p = p_t[gi]; rv.r = r_t[gi]; rv.d = deg_t[k + vzero];
    };
    if (tid == 0) s_ticket = c0 + 4 * NW;        // (three register sets: two chunks in flight beside the one being processed)
    __syncthreads();

    // chunks of the workgroup's range by LDS ticket (the first four rounds are static); a ticket is drawn three bodies before
    // its chunk is processed and read at the end of the body that drew it
    int q0 = c0 + wave, q1 = q0 + NW, q2 = q1 + NW, q3 = q2 + NW;
    CgW1Regs<EPL, W32> ea, eb, ec;
    RowVals ra, rb, rc;
    load_edges(ea, q0); load_rowvals(ra, q0);
    load_edges(eb, q1); load_rowvals(rb, q1);
    double pq = 0.0;

    auto body = [&](const CgW1Regs<EPL, W32>& cur, const RowVals& rv, CgW1Regs<EPL, W32>& fill, RowVals& rvf, const int k, const int k_fill) {
        load_rowvals(rvf, k_fill);
        load_edges(fill, k_fill);
        __builtin_amdgcn_sched_barrier(0);
        const double pn = upd ? mul_add_2r(beta, rv.p, rv.r) : rv.p;
        if (upd && lane < 3) p_t[(size_t)k * 3 + lane] = pn;
        const double prow[3] = {lane_bcast(pn, 0), lane_bcast(pn, 1), lane_bcast(pn, 2)};
        uint32_t cam[EPL];
        double wj[EPL], acc[3] = {0, 0, 0};
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            const bool pad = cur.id[j] == VICAN_PAD_SLOT;
            cam[j] = pad ? pad_cam : (cur.id[j] & 0xFFFFu);
            wj[j] = pad ? 0.0 : (double)cur.w[j];
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
        const double srow = wave_total3(acc[0], acc[1], acc[2], lane);         // lanes 0, 1, 2: the three row sums
        if (lane < 3) {
            const double qv = rv.d * pn - srow;
            q_t[(size_t)k * 3 + lane] = qv;
            pq += pn * qv;
        }
    };
    auto draw = [&]() -> int {
        int t = 0;
        if (lane == 0) t = __hip_atomic_fetch_add(&s_ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return t;
    };
#define CGW1_NEXT(t) do { q0 = q1; q1 = q2; q2 = q3; q3 = __builtin_amdgcn_readfirstlane(t); } while (0)
#pragma unroll 1
    while (q0 < c1) {
        int t = draw();
        body(ea, ra, ec, rc, q0, q2);
        CGW1_NEXT(t);
        if (q0 >= c1) break;
        t = draw();
        body(eb, rb, ea, ra, q0, q2);
        CGW1_NEXT(t);
        if (q0 >= c1) break;
        t = draw();
        body(ec, rc, eb, rb, q0, q2);
        CGW1_NEXT(t);
    }
#undef CGW1_NEXT
    __syncthreads();
    for (int pl = 0; pl < 6; ++pl)
        for (int i = tid; i < C; i += NW * 64) qc_part[((size_t)blockIdx.x * 6 + pl) * C + i] = qc[pl * CP + i];
    const double t = block_sum(pq, red);
    if (tid == 0) pq_part[blockIdx.x] = t;
}

// ---------------------------------------------------------------------------
// right-hand side J^T b on the wave layout (vican_trans_rhs, reference bipgo.py:451-461 + J^T):
//   g_ct = Rc_c^T u_ct + Rt_t^T v_ct ;  rhs_t = sum_c g_ct ;  rhs_c = -sum_t g_ct
// ---------------------------------------------------------------------------
// A pure stream like the dual-update sweep: 52 B per edge (packed index + u + v, in the rotation layout's slot order),
// one wavefront per chunk, the chunk's R_t blocks staged in the wavefront's own LDS region, R_c as planes shared by the
// workgroup, sums in double-word fixed point (to_fix2).  Edge words and R_t values one chunk ahead, row bounds two.
// 8 wavefronts per workgroup (4 on small graphs): two register sets of 24 doubles per lane do not fit 12.
template <int EPL>
struct RhsWRegs { double u[3][EPL], v[3][EPL]; uint32_t id[EPL]; };

static inline int64_t wrhs_lds_bytes(int32_t n_cam, int32_t max_rows, int32_t n_copy, int32_t n_waves) {
    const int64_t per_wave = (((int64_t)max_rows * (48LL * n_copy + 72)) + 15) & ~15LL;
    return 120LL * n_cam + (int64_t)n_waves * per_wave + 256;
}

template <int NW, int EPL, int TRIPS, bool NT>
__global__ __launch_bounds__(NW * 64) void trans_wrhs_kernel(vican_graph_t g, const double* __restrict__ u, const double* __restrict__ v,
                                                             const double* __restrict__ rc, const double* __restrict__ rt,
                                                             double* __restrict__ rhs_t, u64* __restrict__ rhs_c_part, double scale,
                                                             double inv, int lob) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    const int C = g.n_cam, ncopy = g.n_copy, cmask = ncopy - 1, RW = g.max_rows;
    const int tid = threadIdx.x, lane = tid & 63, lane_copy = lane & cmask;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const double lo_scale = ldexp(1.0, lob);
    const int lo_c = 3 * C, lo_t = 3 * RW * ncopy;
    u64* gc = (u64*)lds_raw;                                   // [2][3][C] planes (hi, lo), shared by the workgroup
    double* rcs = (double*)(gc + 6 * C);                       // [9][C] planes
    const size_t per_wave = (((size_t)RW * (48 * ncopy + 72)) + 15) & ~(size_t)15;
    unsigned char* wbase = (unsigned char*)(rcs + 9 * C) + (size_t)wave * per_wave;
    u64* gt = (u64*)wbase;                                     // [2][RW * 3][ncopy] striped row accumulators (this wave's)
    double* rts = (double*)(gt + (size_t)2 * lo_t);            // [RW][9] R_t of the chunk's rows
    const uint32_t pad_cam = (uint32_t)((lane & 31) < C ? (lane & 31) : 0);
    for (int i = tid; i < 9 * C; i += NW * 64) rcs[(i % 9) * C + i / 9] = rc[i];
    for (int i = tid; i < 6 * C; i += NW * 64) gc[i] = 0ull;
    for (int i = lane; i < 2 * lo_t; i += 64) gt[i] = 0ull;
    const int c0 = (int)(((long long)blockIdx.x * g.n_chunk) / gridDim.x);
    const int c1 = (int)(((long long)(blockIdx.x + 1) * g.n_chunk) / gridDim.x);
    const int kmax = g.n_chunk - 1;

    auto load_rows = [&](int k) -> int2 { k = k < kmax ? k : kmax; return *(const int2*)(g.chunk_row0 + k); };
    auto load_edges = [&](RhsWRegs<EPL>& e, int k) {
        k = k < kmax ? k : kmax;
        const size_t s = (size_t)k * g.slots + (size_t)lane * EPL;
        constexpr bool nt = NT;
        load_idx16<EPL, NT>(e.id, g.idx16 + s);                                    // (2-byte index of the wave layout)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            // EPL 4: permuted storage (slot_pos8): slots (4l+j, 4l+j+1) at doubles [64 j + 2 l, +2) - dense 16-byte loads
            const size_t o = ((size_t)k * 3 + p) * g.slots + (EPL == 4 ? (size_t)lane * 2 : (size_t)lane * EPL);
#pragma unroll
            for (int j = 0; j < EPL; j += 2) {
                const size_t oj = o + (EPL == 4 ? 64 * j : j);
                const double2 a = nt ? stream_load((const double2*)(u + oj)) : *(const double2*)(u + oj);
                const double2 b = nt ? stream_load((const double2*)(v + oj)) : *(const double2*)(v + oj);
                e.u[p][j] = a.x; e.u[p][j + 1] = a.y; e.v[p][j] = b.x; e.v[p][j + 1] = b.y;
            }
        }
    };
    struct RowVals { double t[TRIPS]; };
    auto load_rowvals = [&](RowVals& rv, const int2 vrow) {
        const int r0 = __builtin_amdgcn_readfirstlane(vrow.x), n9 = 9 * (__builtin_amdgcn_readfirstlane(vrow.y) - r0);
#pragma unroll
        for (int t = 0; t < TRIPS; ++t) {
            int i = lane + 64 * t;
            i = i < n9 ? i : 0;
            rv.t[t] = rt[(size_t)r0 * 9 + i];
        }
    };
    __syncthreads();

    int k = c0 + wave;
    RhsWRegs<EPL> ea, eb;
    RowVals ra, rb;
    int2 v0 = load_rows(k), v1 = load_rows(k + NW), v2;
    load_edges(ea, k);
    load_rowvals(ra, v0);

    auto body = [&](RhsWRegs<EPL>& cur, RhsWRegs<EPL>& nxt, RowVals& rv, RowVals& rvn, const int2 vrow, const int2 vnext, const int kk) -> int2 {
        const int r0 = __builtin_amdgcn_readfirstlane(vrow.x), nrows = __builtin_amdgcn_readfirstlane(vrow.y) - r0;
        const int n9 = 9 * nrows, n3 = 3 * nrows;
        const int2 vnn = load_rows(kk + 2 * NW);
        load_rowvals(rvn, vnext);
        __builtin_amdgcn_sched_barrier(0);
        load_edges(nxt, kk + NW);
#pragma unroll
        for (int t = 0; t < TRIPS; ++t) {
            const int i = lane + 64 * t;
            if (i < n9) rts[i] = rv.t[t];
        }
        __builtin_amdgcn_wave_barrier();
        // all LDS reads and the arithmetic first, then nothing but atomics (LDS operations of a wavefront return in order)
        uint32_t cam[EPL], row[EPL];
        double gn[EPL][3], ar[EPL][3];                       // -g of the edge (converted in the atomic phase), running row sum
        {
            double acc[3] = {0, 0, 0}, B[9];
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
                const bool pad = cur.id[j] == VICAN_PAD_SLOT;
                cam[j] = pad ? pad_cam : (cur.id[j] & 0xFFFFu); row[j] = pad ? 0u : (cur.id[j] >> 16);
                const double uj[3] = {pad ? 0.0 : cur.u[0][j], pad ? 0.0 : cur.u[1][j], pad ? 0.0 : cur.u[2][j]};
                const double vj[3] = {pad ? 0.0 : cur.v[0][j], pad ? 0.0 : cur.v[1][j], pad ? 0.0 : cur.v[2][j]};
                if (j == 0 || row[j] != row[j - 1]) {
#pragma unroll
                    for (int q = 0; q < 9; ++q) B[q] = rts[row[j] * 9 + q];
                    acc[0] = acc[1] = acc[2] = 0.0;
                }
#pragma unroll
                for (int i = 0; i < 3; ++i) {       // world<-node = transpose of the stored blocks
                    const double gi = rcs[(0 * 3 + i) * C + cam[j]] * uj[0] + rcs[(1 * 3 + i) * C + cam[j]] * uj[1] +
                                      rcs[(2 * 3 + i) * C + cam[j]] * uj[2] + B[0 * 3 + i] * vj[0] + B[1 * 3 + i] * vj[1] + B[2 * 3 + i] * vj[2];
                    acc[i] += gi;
                    gn[j][i] = -gi;
                    ar[j][i] = acc[i];
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const Fix2 f = to_fix2(gn[j][i], scale, lo_scale);
                lds_add_fix(&gc[i * C + cam[j]], f.hi); lds_add_fix(&gc[lo_c + i * C + cam[j]], f.lo);
            }
            if (j == EPL - 1 || row[j] != row[j + 1]) {
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const Fix2 f = to_fix2(ar[j][i], scale, lo_scale);
                    u64* a = &gt[(row[j] * 3 + i) * ncopy + lane_copy];
                    lds_add_fix(a, f.hi); lds_add_fix(a + lo_t, f.lo);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
        for (int base = 0; base < n3 * ncopy; base += 64) {
            const int a = base + lane;
            const bool live = a < n3 * ncopy;
            u64 sum = 0ull, slo = 0ull;
            if (live) { sum = gt[a]; slo = gt[lo_t + a]; gt[a] = 0ull; gt[lo_t + a] = 0ull; }
            sum = stripe_sum(sum, ncopy); slo = stripe_sum(slo, ncopy);
            if (live && (a & cmask) == 0) rhs_t[(size_t)r0 * 3 + a / ncopy] = fix2_value((long long)sum, (long long)slo, lob, inv);
        }
        __builtin_amdgcn_wave_barrier();
        return vnn;
    };
#pragma unroll 1
    while (k < c1) {
        v2 = body(ea, eb, ra, rb, v0, v1, k);
        k += NW;
        if (k >= c1) break;
        v0 = body(eb, ea, rb, ra, v1, v2, k);
        k += NW;
        const int2 tmp = v0; v0 = v2; v1 = tmp;
    }
    __syncthreads();
    for (int i = tid; i < 6 * C; i += NW * 64) rhs_c_part[(size_t)blockIdx.x * 6 * C + i] = gc[i];
}

extern "C" __attribute__((visibility("hidden"))) int vican_trans_wrhs(const vican_graph_t* g, const double* u, const double* v, const double* rc,
                                                                      const double* rt, double* rhs_t, void* rhs_c_part, double scale,
                                                                      double inv, int lob, void* stream) {
    if (!g->idx16) return set_err(VICAN_ERR_ARG, "%s: the wave layout needs vican_graph_t.idx16 (vican_pack_idx16)", "vican_trans_wrhs");
    const int epl = g->slots / 64, trips = (9 * g->max_rows + 63) / 64;
    // (measured: 8 wavefronts beat 12 even where 12 fit the register budget: stress 240 vs 253 us, sparse 207 vs 302 us)
    int nw = g->wg_waves >= 8 ? 8 : 4;
    while (nw > 4 && wrhs_lds_bytes(g->n_cam, g->max_rows, g->n_copy, nw) > vican_lds_limit_bytes()) nw -= 4;
    const size_t lds = (size_t)wrhs_lds_bytes(g->n_cam, g->max_rows, g->n_copy, nw);
    if ((int64_t)lds > vican_lds_limit_bytes()) return set_err(VICAN_ERR_CAPACITY, "%s: camera tables / row staging do not fit in LDS", "vican_trans_rhs (wave layout)");
    if (trips > 9) return set_err(VICAN_ERR_CAPACITY, "vican_trans_rhs: more than 64 rows per chunk");
    // the grid of the rotation layout has one workgroup per 12 / 8 / 4 wavefronts' worth of chunks; this kernel has nw
    hipStream_t s = (hipStream_t)stream;
#define WRHS_LAUNCH_(NW_, E_, T_, NT_)                                                                                    \
    do {                                                                                                                  \
        auto kern = trans_wrhs_kernel<NW_, E_, T_, NT_>;                                                                  \
        static size_t conf = 0;                                                                                           \
        if (lds > conf) { hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); conf = lds; } \
        VICAN_LAUNCH_SWEEP(kern, dim3(g->n_wg), dim3(NW_ * 64), lds, s, *g, u, v, rc, rt, rhs_t, (u64*)rhs_c_part, scale, inv, lob); \
    } while (0)
#define WRHS_LAUNCH(NW_, E_, T_) do { if (g->stream_nt) WRHS_LAUNCH_(NW_, E_, T_, true); else WRHS_LAUNCH_(NW_, E_, T_, false); } while (0)
#define WRHS_PICK(NW_)                                                                                                    \
    do {                                                                                                                  \
        if (epl == 4) { if (trips <= 1) WRHS_LAUNCH(NW_, 4, 1); else if (trips <= 3) WRHS_LAUNCH(NW_, 4, 3); else WRHS_LAUNCH(NW_, 4, 9); } \
        else          { if (trips <= 1) WRHS_LAUNCH(NW_, 2, 1); else if (trips <= 3) WRHS_LAUNCH(NW_, 2, 3); else WRHS_LAUNCH(NW_, 2, 9); } \
    } while (0)
    if (nw == 12) WRHS_PICK(12); else if (nw == 8) WRHS_PICK(8); else WRHS_PICK(4);
#undef WRHS_PICK
#undef WRHS_LAUNCH
#undef WRHS_LAUNCH_
    LAUNCH_CHECK("vican_trans_rhs");
    return VICAN_OK;
}

// ---------------------------------------------------------------------------
// LSQR on the wave layout (lsqr_solver="direct", reference bipgo.py:479-480; the block-layout kernels and the mathematics are in
// vican_lsqr.hip): u~_1 = b~ and the fused step  u^ = J~ v - coef u~,  |u^|^2,  z = J~^T u^  - one wavefront per chunk, the edge
// vector u~ and sqrt(w) in the rotation layout's slot order (dense permuted storage, slot_pos8), 60 B per edge and step.
// ---------------------------------------------------------------------------
template <int EPL>
struct LsqrWRegs { double u[3][EPL], s[EPL]; uint32_t id[EPL]; };

template <int EPL>
__device__ __forceinline__ size_t wpos8(int lane, int j) { return EPL == 4 ? (size_t)(64 * (j & 2) + lane * 2 + (j & 1)) : (size_t)(lane * EPL + j); }

// u~_1 = b~ = (Rc^T u_e + Rt^T v_e) / sqrt(w_e) per edge, sw = sqrt(w), part[block] = partial |b~|^2
template <int NW, int EPL, int TRIPS>
__global__ __launch_bounds__(NW * 64) void lsqr_winit_kernel(vican_graph_t g, const double* __restrict__ w, const double* __restrict__ ue,
                                                             const double* __restrict__ ve, const double* __restrict__ rc,
                                                             const double* __restrict__ rt, double* __restrict__ u, double* __restrict__ sw,
                                                             double* __restrict__ part) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    __shared__ double red[16];
    const int C = g.n_cam, RW = g.max_rows;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    double* rcs = (double*)lds_raw;                            // [9][C] planes
    double* rts = rcs + 9 * C + (size_t)wave * RW * 9;         // [RW][9] this wavefront's rows
    for (int i = tid; i < 9 * C; i += NW * 64) rcs[(i % 9) * C + i / 9] = rc[i];
    __syncthreads();
    const int c0 = (int)(((long long)blockIdx.x * g.n_chunk) / gridDim.x);
    const int c1 = (int)(((long long)(blockIdx.x + 1) * g.n_chunk) / gridDim.x);
    double nrm = 0.0;
    for (int k = c0 + wave; k < c1; k += NW) {                 // (runs once per solve: not pipelined)
        const int r0 = g.chunk_row0[k], n9 = 9 * (g.chunk_row0[k + 1] - r0);
        for (int i = lane; i < n9; i += 64) rts[i] = rt[(size_t)r0 * 9 + i];
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            const size_t e = (size_t)k * g.slots + (size_t)lane * EPL + j, e8 = (size_t)k * g.slots + wpos8<EPL>(lane, j);
            const uint32_t id = g.idx[e];
            double out[3] = {0, 0, 0}, sq = 0.0;
            if (id != VICAN_PAD_SLOT) {
                const uint32_t cam = id & 0xFFFFu, row = id >> 16;
                sq = sqrt(w[e8]);
                const double inv_s = 1.0 / sq;
                double uu[3], vv[3];
#pragma unroll
                for (int p = 0; p < 3; ++p) { uu[p] = ue[((size_t)k * 3 + p) * g.slots + wpos8<EPL>(lane, j)]; vv[p] = ve[((size_t)k * 3 + p) * g.slots + wpos8<EPL>(lane, j)]; }
                const double* B = rts + row * 9;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    const double gi = rcs[(0 * 3 + i) * C + cam] * uu[0] + rcs[(1 * 3 + i) * C + cam] * uu[1] + rcs[(2 * 3 + i) * C + cam] * uu[2] +
                                      B[0 * 3 + i] * vv[0] + B[1 * 3 + i] * vv[1] + B[2 * 3 + i] * vv[2];
                    out[i] = gi * inv_s;
                    nrm += out[i] * out[i];
                }
            }
#pragma unroll
            for (int p = 0; p < 3; ++p) u[((size_t)k * 3 + p) * g.slots + wpos8<EPL>(lane, j)] = out[p];
            sw[e8] = sq;
        }
        __builtin_amdgcn_wave_barrier();
    }
    const double t = block_sum(nrm, red);
    if (tid == 0) part[blockIdx.x] = t;
}

static inline int64_t lsqr_wstep_lds_bytes(int32_t n_cam, int32_t max_rows, int32_t n_copy, int32_t n_waves) {
    const int64_t per_wave = (((int64_t)max_rows * 3 * (16LL * n_copy + 8)) + 15) & ~15LL;
    return 72LL * n_cam + (int64_t)n_waves * per_wave + 256;
}

template <int NW, int EPL, int TRIPS, bool NT>
__global__ __launch_bounds__(NW * 64) void lsqr_wstep_kernel(vican_graph_t g, const double* __restrict__ sw, double* __restrict__ u,
                                                             const double* __restrict__ v_c, const double* __restrict__ v_t,
                                                             double* __restrict__ z_t, u64* __restrict__ zc_part, double* __restrict__ part,
                                                             const vican_lsqr_state_t* __restrict__ st) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    __shared__ double red[16];
    __shared__ int s_ticket;
    if (st->done) return;
    const int C = g.n_cam, ncopy = g.n_copy, cmask = ncopy - 1, RW = g.max_rows;
    const int tid = threadIdx.x, lane = tid & 63, lane_copy = lane & cmask;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const double coef = st->coef, scale = st->qscale, inv = st->qinv;
    const int lob = st->lo_bits;
    const double lo_scale = ldexp(1.0, lob);
    const int lo_c = 3 * C, lo_t = 3 * RW * ncopy;
    u64* zc = (u64*)lds_raw;                                   // [2][3][C] planes (hi, lo), shared by the workgroup
    double* vcs = (double*)(zc + 6 * C);                       // [3][C] planes
    const size_t per_wave = (((size_t)RW * 3 * (16 * ncopy + 8)) + 15) & ~(size_t)15;
    unsigned char* wbase = (unsigned char*)(vcs + 3 * C) + (size_t)wave * per_wave;
    u64* zt = (u64*)wbase;                                     // [2][RW * 3][ncopy] striped row accumulators (this wave's)
    double* vts = (double*)(zt + (size_t)2 * lo_t);            // [RW * 3] v_t of the chunk's rows
    const uint32_t pad_cam = (uint32_t)((lane & 31) < C ? (lane & 31) : 0);
    for (int i = tid; i < 3 * C; i += NW * 64) { vcs[(i % 3) * C + i / 3] = v_c[i]; zc[i] = 0ull; zc[lo_c + i] = 0ull; }
    for (int i = lane; i < 2 * lo_t; i += 64) zt[i] = 0ull;
    const int c0 = (int)(((long long)blockIdx.x * g.n_chunk) / gridDim.x);
    const int c1 = (int)(((long long)(blockIdx.x + 1) * g.n_chunk) / gridDim.x);
    const int kmax = g.n_chunk - 1;
    if (tid == 0) s_ticket = c0 + 3 * NW;
    // the edge vector u~ (24 B per edge, read AND written every step): streamed past the caches when the graph's streams are
    // (a vector of more than the 256 MB Infinity Cache cannot be found there again a step later; it only evicts what could)
    constexpr bool nt_u = NT;

    auto load_rows = [&](int k) -> int2 { k = k < kmax ? k : kmax; return *(const int2*)(g.chunk_row0 + k); };
    auto load_edges = [&](LsqrWRegs<EPL>& e, int k) {
        k = k < kmax ? k : kmax;
        const size_t s = (size_t)k * g.slots + (size_t)lane * EPL;
        constexpr bool nt = NT;
        load_idx16<EPL, NT>(e.id, g.idx16 + s);                                    // (2-byte index of the wave layout)
#pragma unroll
        for (int j = 0; j < EPL; j += 2) {
            const double* a = sw + (size_t)k * g.slots + wpos8<EPL>(lane, j);
            const double2 t = nt ? stream_load((const double2*)a) : *(const double2*)a;
            e.s[j] = t.x; e.s[j + 1] = t.y;
        }
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int j = 0; j < EPL; j += 2) {
                const double* a = u + ((size_t)k * 3 + p) * g.slots + wpos8<EPL>(lane, j);
                const double2 t = nt_u ? stream_load((const double2*)a) : *(const double2*)a;
                e.u[p][j] = t.x; e.u[p][j + 1] = t.y;
            }
    };
    struct RowVals { double v[TRIPS]; };
    auto load_rowvals = [&](RowVals& rv, const int2 vrow) {
        const int r0 = __builtin_amdgcn_readfirstlane(vrow.x), n3 = 3 * (__builtin_amdgcn_readfirstlane(vrow.y) - r0);
#pragma unroll
        for (int t = 0; t < TRIPS; ++t) {
            int i = lane + 64 * t;
            i = i < n3 ? i : 0;
            rv.v[t] = v_t[(size_t)r0 * 3 + i];
        }
    };
    __syncthreads();

    int k = c0 + wave, kb = k + NW, kc = k + 2 * NW;
    LsqrWRegs<EPL> ea, eb;
    RowVals ra, rb;
    int2 v0 = load_rows(k), v1 = load_rows(kb), v2;
    load_edges(ea, k);
    load_rowvals(ra, v0);
    double nrm = 0.0;

    auto body = [&](LsqrWRegs<EPL>& cur, LsqrWRegs<EPL>& nxt, RowVals& rv, RowVals& rvn, const int2 vrow, const int2 vnext, const int kcur,
                    const int k_next, const int k_after) -> int2 {
        const int r0 = __builtin_amdgcn_readfirstlane(vrow.x), n3 = 3 * (__builtin_amdgcn_readfirstlane(vrow.y) - r0);
        const int2 vnn = load_rows(k_after);
        load_rowvals(rvn, vnext);
        __builtin_amdgcn_sched_barrier(0);
        load_edges(nxt, k_next);
        const bool single = n3 == 3;                           // one row in the chunk: its v_t broadcast from lanes 0..2, row sums by DPP
        double vrow1[3] = {0, 0, 0};
        if (single) {
            vrow1[0] = lane_bcast(rv.v[0], 0); vrow1[1] = lane_bcast(rv.v[0], 1); vrow1[2] = lane_bcast(rv.v[0], 2);
        } else {
#pragma unroll
            for (int t = 0; t < TRIPS; ++t) { const int i = lane + 64 * t; if (i < n3) vts[i] = rv.v[t]; }
            __builtin_amdgcn_wave_barrier();
        }
        uint32_t cam[EPL], row[EPL];
        double un[3][EPL], sa[EPL][3], ar[EPL][3];
        {
            double acc[3] = {0, 0, 0}, vt3[3] = {vrow1[0], vrow1[1], vrow1[2]};
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
                const bool pad = cur.id[j] == VICAN_PAD_SLOT;
                cam[j] = pad ? pad_cam : (cur.id[j] & 0xFFFFu); row[j] = pad ? 0u : (cur.id[j] >> 16);
                const double sj = pad ? 0.0 : cur.s[j];
                if (!single && (j == 0 || row[j] != row[j - 1])) {
#pragma unroll
                    for (int p = 0; p < 3; ++p) vt3[p] = vts[row[j] * 3 + p];
                    acc[0] = acc[1] = acc[2] = 0.0;
                }
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    const double uh = pad ? 0.0 : sj * (vt3[p] - vcs[p * C + cam[j]]) - coef * cur.u[p][j];
                    un[p][j] = uh;
                    nrm += uh * uh;
                    const double a = sj * uh;
                    acc[p] += a;
                    sa[j][p] = -a;
                    ar[j][p] = acc[p];
                }
            }
        }
        // the new edge vector back to memory (dense permuted positions)
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int j = 0; j < EPL; j += 2)
                if (nt_u) stream_store((double2*)(u + ((size_t)kcur * 3 + p) * g.slots + wpos8<EPL>(lane, j)), make_double2(un[p][j], un[p][j + 1]));
                else *(double2*)(u + ((size_t)kcur * 3 + p) * g.slots + wpos8<EPL>(lane, j)) = make_double2(un[p][j], un[p][j + 1]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const Fix2 f = to_fix2(sa[j][p], scale, lo_scale);
                lds_add_fix(&zc[p * C + cam[j]], f.hi); lds_add_fix(&zc[lo_c + p * C + cam[j]], f.lo);
            }
            if (!single && (j == EPL - 1 || row[j] != row[j + 1])) {
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    const Fix2 f = to_fix2(ar[j][p], scale, lo_scale);
                    u64* a = &zt[(row[j] * 3 + p) * ncopy + lane_copy];
                    lds_add_fix(a, f.hi); lds_add_fix(a + lo_t, f.lo);
                }
            }
        }
        if (single) {
            const double srow = wave_total3(ar[EPL - 1][0], ar[EPL - 1][1], ar[EPL - 1][2], lane);
            if (lane < 3) z_t[(size_t)r0 * 3 + lane] = srow;
        } else {
            __builtin_amdgcn_wave_barrier();
            for (int base = 0; base < n3 * ncopy; base += 64) {
                const int a = base + lane;
                const bool live = a < n3 * ncopy;
                u64 sum = 0ull, slo = 0ull;
                if (live) { sum = zt[a]; slo = zt[lo_t + a]; zt[a] = 0ull; zt[lo_t + a] = 0ull; }
                sum = stripe_sum(sum, ncopy); slo = stripe_sum(slo, ncopy);
                if (live && (a & cmask) == 0) z_t[(size_t)r0 * 3 + a / ncopy] = fix2_value((long long)sum, (long long)slo, lob, inv);
            }
            __builtin_amdgcn_wave_barrier();
        }
        return vnn;
    };
    auto draw = [&]() -> int {
        int t = 0;
        if (lane == 0) t = __hip_atomic_fetch_add(&s_ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        return t;
    };
#pragma unroll 1
    while (k < c1) {
        int t = draw();
        v2 = body(ea, eb, ra, rb, v0, v1, k, kb, kc);
        k = kb; kb = kc; kc = __builtin_amdgcn_readfirstlane(t);
        if (k >= c1) break;
        t = draw();
        v0 = body(eb, ea, rb, ra, v1, v2, k, kb, kc);
        k = kb; kb = kc; kc = __builtin_amdgcn_readfirstlane(t);
        const int2 tmp = v0; v0 = v2; v1 = tmp;
    }
    __syncthreads();
    for (int i = tid; i < 6 * C; i += NW * 64) zc_part[(size_t)blockIdx.x * 6 * C + i] = zc[i];
    const double t = block_sum(nrm, red);
    if (tid == 0) part[blockIdx.x] = t;
}

extern "C" __attribute__((visibility("hidden"))) int vican_lsqr_winit(const vican_graph_t* g, const double* w, const double* ue, const double* ve,
                                                                      const double* rc, const double* rt, double* u, double* sw, double* part,
                                                                      void* stream) {
    const int nw = g->wg_waves >= 8 ? 8 : 4;
    const size_t lds = (size_t)8 * (9 * g->n_cam + (size_t)nw * 9 * g->max_rows) + 256;
    if ((int64_t)lds > vican_lds_limit_bytes()) return set_err(VICAN_ERR_CAPACITY, "%s: camera tables / row staging do not fit in LDS", "vican_lsqr_init_u (wave layout)");
    const int epl = g->slots / 64;
    hipStream_t s = (hipStream_t)stream;
#define WINIT_LAUNCH(NW_, E_)                                                                                             \
    do {                                                                                                                  \
        auto kern = lsqr_winit_kernel<NW_, E_, 1>;                                                                        \
        static size_t conf = 0;                                                                                           \
        if (lds > conf) { hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); conf = lds; } \
        hipLaunchKernelGGL(kern, dim3(g->n_wg), dim3(NW_ * 64), lds, s, *g, w, ue, ve, rc, rt, u, sw, part);             \
    } while (0)
    if (nw == 8) { if (epl == 4) WINIT_LAUNCH(8, 4); else WINIT_LAUNCH(8, 2); }
    else         { if (epl == 4) WINIT_LAUNCH(4, 4); else WINIT_LAUNCH(4, 2); }
#undef WINIT_LAUNCH
    LAUNCH_CHECK("vican_lsqr_init_u");
    return VICAN_OK;
}
extern "C" __attribute__((visibility("hidden"))) int vican_lsqr_wstep(const vican_graph_t* g, const double* sw, double* u, const double* v_c,
                                                                      const double* v_t, double* z_t, void* zc_part, double* part,
                                                                      const vican_lsqr_state_t* st, void* stream) {
    if (!g->idx16) return set_err(VICAN_ERR_ARG, "%s: the wave layout needs vican_graph_t.idx16 (vican_pack_idx16)", "vican_lsqr_wstep");
    int nw = g->wg_waves >= 8 ? 8 : 4;
    while (nw > 4 && lsqr_wstep_lds_bytes(g->n_cam, g->max_rows, g->n_copy, nw) > vican_lds_limit_bytes()) nw -= 4;
    const size_t lds = (size_t)lsqr_wstep_lds_bytes(g->n_cam, g->max_rows, g->n_copy, nw);
    if ((int64_t)lds > vican_lds_limit_bytes()) return set_err(VICAN_ERR_CAPACITY, "%s: camera tables / row staging do not fit in LDS", "vican_lsqr_step (wave layout)");
    const int epl = g->slots / 64, trips = (3 * g->max_rows + 63) / 64;
    if (trips > 3) return set_err(VICAN_ERR_CAPACITY, "vican_lsqr_step: more than 64 rows per chunk");
    hipStream_t s = (hipStream_t)stream;
#define WSTEP_LAUNCH_(NW_, E_, T_, NT_)                                                                                   \
    do {                                                                                                                  \
        auto kern = lsqr_wstep_kernel<NW_, E_, T_, NT_>;                                                                  \
        static size_t conf = 0;                                                                                           \
        if (lds > conf) { hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); conf = lds; } \
        VICAN_LAUNCH_SWEEP(kern, dim3(g->n_wg), dim3(NW_ * 64), lds, s, *g, sw, u, v_c, v_t, z_t, (u64*)zc_part, part, st); \
    } while (0)
#define WSTEP_LAUNCH(NW_, E_, T_) do { if (g->stream_nt) WSTEP_LAUNCH_(NW_, E_, T_, true); else WSTEP_LAUNCH_(NW_, E_, T_, false); } while (0)
#define WSTEP_PICK(NW_)                                                                                                   \
    do {                                                                                                                  \
        if (epl == 4) { if (trips <= 1) WSTEP_LAUNCH(NW_, 4, 1); else if (trips == 2) WSTEP_LAUNCH(NW_, 4, 2); else WSTEP_LAUNCH(NW_, 4, 3); } \
        else          { if (trips <= 1) WSTEP_LAUNCH(NW_, 2, 1); else if (trips == 2) WSTEP_LAUNCH(NW_, 2, 2); else WSTEP_LAUNCH(NW_, 2, 3); } \
    } while (0)
    if (nw == 8) WSTEP_PICK(8); else WSTEP_PICK(4);
#undef WSTEP_PICK
#undef WSTEP_LAUNCH
#undef WSTEP_LAUNCH_
    LAUNCH_CHECK("vican_lsqr_step");
    return VICAN_OK;
}

// launcher: called by vican_cg_sweep for graphs in the wave layout
extern "C" __attribute__((visibility("hidden"))) int vican_cg_wsweep(const vican_graph_t* g, const double* w, const double* deg_t,
                                                                     const double* p_c, const double* r_t, double* p_t, double* q_t,
                                                                     void* qc_part, double* pq_part, const vican_cg_state_t* st,
                                                                     void* stream, int partial) {
    if (!g->idx16) return set_err(VICAN_ERR_ARG, "%s: the wave layout needs vican_graph_t.idx16 (vican_pack_idx16)", "vican_cg_wsweep");
    const int nw = g->wg_waves >= 12 ? 12 : (g->wg_waves >= 8 ? 8 : 4);
    const size_t lds = (size_t)vican_cg_wsweep_lds_bytes(g->n_cam, g->max_rows, g->n_copy, nw);
    if ((int64_t)lds > 160 * 1024) return set_err(VICAN_ERR_CAPACITY, "%s: camera tables / row staging do not fit in LDS", "vican_cg_sweep (wave layout)");
    const int epl = g->slots / 64, trips = (3 * g->max_rows + 63) / 64, cp = (int)plane_stride(g->n_cam);
    const bool nt = g->stream_nt != 0;
    hipStream_t s = (hipStream_t)stream;
    if (g->n_chunk == g->n_time && nw >= 8 && !partial && g->idx16) {
        // every chunk is one row: the specialised kernel (no row staging: LDS = the nine camera planes)
        const size_t lds1 = (size_t)72 * cp + 256;
        // (float32 weight stream: only for the very array w32 was made from, 4 edges per lane)
        const bool w32 = g->w32 != nullptr && g->w32_src == w && epl == 4;
#define CGW1_LAUNCH_(NW_, E_, CP_, NT_)                                                                                   \
        do {                                                                                                              \
            if (w32 && E_ == 4) CGW1_LAUNCH__(NW_, 4, CP_, NT_, true); else CGW1_LAUNCH__(NW_, E_, CP_, NT_, false);      \
        } while (0)
#define CGW1_LAUNCH__(NW_, E_, CP_, NT_, W32_)                                                                            \
        do {                                                                                                              \
            auto kern = cg_wsweep1_kernel<NW_, E_, CP_, NT_, W32_>;                                                       \
            static size_t conf = 0;                                                                                       \
            if (lds1 > conf) { hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1); conf = lds1; } \
            VICAN_LAUNCH_SWEEP(kern, dim3(g->n_wg), dim3(NW_ * 64), lds1, s, *g, w, deg_t, p_c, r_t, p_t, q_t, (u64*)qc_part, pq_part, st); \
        } while (0)
#define CGW1_LAUNCH(NW_, E_, CP_) do { if (nt) CGW1_LAUNCH_(NW_, E_, CP_, true); else CGW1_LAUNCH_(NW_, E_, CP_, false); } while (0)
#define CGW1_PICK(NW_)                                                                                                    \
        do {                                                                                                              \
            if (epl == 4) { if (cp == 256) CGW1_LAUNCH(NW_, 4, 256); else if (cp == 512) CGW1_LAUNCH(NW_, 4, 512); else CGW1_LAUNCH(NW_, 4, 1024); } \
            else          { if (cp == 256) CGW1_LAUNCH(NW_, 2, 256); else if (cp == 512) CGW1_LAUNCH(NW_, 2, 512); else CGW1_LAUNCH(NW_, 2, 1024); } \
        } while (0)
        CGW1_PICK(12);                  // (12 wavefronts per workgroup: 8 and 16 measured slower - profiles/NOTES.md, round 4)
#undef CGW1_PICK
#undef CGW1_LAUNCH
#undef CGW1_LAUNCH_
#undef CGW1_LAUNCH__
        LAUNCH_CHECK("vican_cg_sweep");
        return VICAN_OK;
    }
#define CGW_LAUNCH_(NW_, E_, T_, CP_, NT_)                                                                                \
    do {                                                                                                                  \
        auto kern = cg_wsweep_kernel<NW_, E_, T_, CP_, NT_>;                                                              \
        static size_t conf = 0;                                                                                           \
        if (lds > conf) { hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); conf = lds; } \
        VICAN_LAUNCH_SWEEP(kern, dim3(g->n_wg), dim3(NW_ * 64), lds, s, *g, w, deg_t, p_c, r_t, p_t, q_t, (u64*)qc_part, pq_part, st, partial); \
    } while (0)
#define CGW_LAUNCH(NW_, E_, T_)                                                                                           \
    do {                                                                                                                  \
        if (nt) { if (cp == 256) CGW_LAUNCH_(NW_, E_, T_, 256, true); else if (cp == 512) CGW_LAUNCH_(NW_, E_, T_, 512, true); else CGW_LAUNCH_(NW_, E_, T_, 1024, true); } \
        else    { if (cp == 256) CGW_LAUNCH_(NW_, E_, T_, 256, false); else if (cp == 512) CGW_LAUNCH_(NW_, E_, T_, 512, false); else CGW_LAUNCH_(NW_, E_, T_, 1024, false); } \
    } while (0)
#define CGW_PICK(NW_)                                                                                                     \
    do {                                                                                                                  \
        if (epl == 4) { if (trips <= 1) CGW_LAUNCH(NW_, 4, 1); else if (trips == 2) CGW_LAUNCH(NW_, 4, 2); else CGW_LAUNCH(NW_, 4, 3); } \
        else          { if (trips <= 1) CGW_LAUNCH(NW_, 2, 1); else if (trips == 2) CGW_LAUNCH(NW_, 2, 2); else CGW_LAUNCH(NW_, 2, 3); } \
    } while (0)
    if (nw == 12) CGW_PICK(12); else if (nw == 8) CGW_PICK(8); else CGW_PICK(4);
#undef CGW_PICK
#undef CGW_LAUNCH
#undef CGW_LAUNCH_
    LAUNCH_CHECK("vican_cg_sweep");
    return VICAN_OK;
}
