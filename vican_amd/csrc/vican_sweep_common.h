// vican_sweep_common.h - register-level helpers shared by the two edge sweeps (vican_sweep.hip: one workgroup per
// chunk of 256..1024 lanes; vican_wsweep.hip: one wavefront per chunk): vector loads of a chunk's planes, the
// fixed-point conversions and the explicitly rounded 3-term dot product.
#pragma once
#include "vican_common.h"

// words of the scheduler block (unsigned int*)(fx + 12) shared by the sweeps: [0] chunk ranges handed out, [8] workgroups
// finished, [VICAN_SCHED_REDO] raised by the fused dual-update sweep when a row needs the in-kernel SVD (vican_wsweep.hip)
#define VICAN_SCHED_REDO 4

static inline int64_t ssize(int32_t storage) { return storage == VICAN_STORE_F32 ? 4 : 8; }
// camera planes are padded to a compile-time stride CP (256, 512 or 1024 entries) so that the nine
// component planes are reached with LDS immediate offsets instead of per-access address math
static inline int64_t plane_stride(int32_t n_cam) { return n_cam <= 256 ? 256 : (n_cam <= 512 ? 512 : 1024); }

// Position of slot s inside its chunk for the 8-BYTE-PER-SLOT arrays of the translation stage (w; the planes of u, v).
// Wave layout with four slots per lane (f32 block storage, 256 slots): a lane's four doubles would be 32 contiguous bytes,
// i.e. each of its two 16-byte loads would cover only every other 16 bytes of a 2 KB span.  Measured (tools/
// stride_read_bench.hip, non-temporal loads, 12 wavefronts per workgroup): 5.74-5.81 TB/s that way against 6.60-6.93 TB/s
// when every load instruction of a wavefront is dense - so these arrays are stored permuted: slots (4l+2h, 4l+2h+1) of lane
// l at doubles [128 h + 2 l, +2).  Every other layout / storage keeps plain slot order (already dense).
__host__ __device__ inline int slot_pos8(const vican_graph_t& g, int s) {
    if (g.layout == VICAN_LAYOUT_WAVE && g.slots == 256) return ((s & 2) << 6) + ((s >> 2) << 1) + (s & 1);
    return s;
}

template <typename S> struct Vec;
template <> struct Vec<float>  { typedef float4  type; static constexpr int N = 4; };
template <> struct Vec<double> { typedef double2 type; static constexpr int N = 2; };

template <typename S> __device__ __forceinline__ S vget(const typename Vec<S>::type& v, int j);
template <> __device__ __forceinline__ float vget<float>(const float4& v, int j) {
    return j == 0 ? v.x : j == 1 ? v.y : j == 2 ? v.z : v.w;
}
template <> __device__ __forceinline__ double vget<double>(const double2& v, int j) { return j == 0 ? v.x : v.y; }

template <typename S, int EPL>
struct ChunkRegs {
    typename Vec<S>::type m[9];
    uint32_t id[EPL];
};

// 16-byte (8-byte) loads of the edge stream, non-temporal form
#ifdef VICAN_NO_NT
template <typename T> __device__ __forceinline__ T stream_load(const T* p) { return *p; }
#else
typedef float vican_v4f __attribute__((ext_vector_type(4)));
typedef double vican_v2d __attribute__((ext_vector_type(2)));
typedef unsigned int vican_v4u __attribute__((ext_vector_type(4)));
typedef unsigned int vican_v2u __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float4 stream_load(const float4* p) {
    const vican_v4f t = __builtin_nontemporal_load((const vican_v4f*)p); return make_float4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ double2 stream_load(const double2* p) {
    const vican_v2d t = __builtin_nontemporal_load((const vican_v2d*)p); return make_double2(t.x, t.y);
}
__device__ __forceinline__ void stream_store(double2* p, const double2 v) {
    vican_v2d t; t.x = v.x; t.y = v.y;
    __builtin_nontemporal_store(t, (vican_v2d*)p);
}
__device__ __forceinline__ uint4 stream_load(const uint4* p) {
    const vican_v4u t = __builtin_nontemporal_load((const vican_v4u*)p); return make_uint4(t.x, t.y, t.z, t.w);
}
__device__ __forceinline__ uint2 stream_load(const uint2* p) {
    const vican_v2u t = __builtin_nontemporal_load((const vican_v2u*)p); return make_uint2(t.x, t.y);
}
#endif

// An edge stream far larger than the caches (g.stream_nt, set by the host above ~192 MB) is read with non-temporal
// loads (global_load ... nt): it is touched exactly once per sweep, and this way it neither evicts the tables the
// sweep re-reads (x, duals) nor pays for allocating lines it will never hit - measured on the 1 GB stress graph:
// wave sweep 184 -> 175 us, block sweep 195 -> 178 us.  Cache-resident graphs (large_shop: 2.4 MB) keep plain loads -
// their blocks ARE re-read from L2 / Infinity Cache by the next sweep.  Both branches issue the same ten loads, so the
// compiler's vmcnt bookkeeping stays exact.
// NT = 0 / 1: the choice is made at COMPILE time (wave_sweep_kernel): behind a run-time branch the compiler's vmcnt bookkeeping
// collapses at the join - the wave sweep waited with vmcnt(0), i.e. for its whole prefetch, in front of phase 2 (round-4 ISA;
// with the branch resolved it waits with vmcnt(10)).  NT = -1: run-time (block sweep: one barrier-synchronised wait per chunk).
template <typename S, int EPL, int NT = -1>
__device__ __forceinline__ void load_chunk(ChunkRegs<S, EPL>& c, const vican_graph_t& g, int k, int tid) {
    typedef typename Vec<S>::type V;
    const S* blk = (const S*)g.blk;
    const size_t pbase = (size_t)k * 9 * g.slots + (size_t)tid * EPL;
    const uint32_t* ip = g.idx + (size_t)k * g.slots + (size_t)tid * EPL;
    if (NT < 0 ? g.stream_nt != 0 : NT != 0) {
#pragma unroll
        for (int p = 0; p < 9; ++p) c.m[p] = stream_load((const V*)(blk + pbase + (size_t)p * g.slots));
        if (EPL == 4) { const uint4 t = stream_load((const uint4*)ip); c.id[0] = t.x; c.id[1] = t.y; c.id[2] = t.z; c.id[3] = t.w; }
        else          { const uint2 t = stream_load((const uint2*)ip); c.id[0] = t.x; c.id[1] = t.y; }
    } else {
#pragma unroll
        for (int p = 0; p < 9; ++p) c.m[p] = *(const V*)(blk + pbase + (size_t)p * g.slots);
        if (EPL == 4) { const uint4 t = *(const uint4*)ip; c.id[0] = t.x; c.id[1] = t.y; c.id[2] = t.z; c.id[3] = t.w; }
        else          { const uint2 t = *(const uint2*)ip; c.id[0] = t.x; c.id[1] = t.y; }
    }
}

// A wave-layout chunk through the 2-byte index (vican_graph_t.idx16: camera | row << 10): the nine planes as load_chunk, 8 (4)
// bytes of index per lane instead of 16 (8); ids are expanded to the 32-bit form of idx so that the kernels' bodies do not change.
__device__ __forceinline__ uint32_t idx16_expand(uint32_t h) { return h == 0xFFFFu ? VICAN_PAD_SLOT : ((h & 0x3FFu) | ((h >> 10) << 16)); }
// the EPL ids of a lane from the 2-byte index, expanded to the 32-bit form (ip: the lane's first slot)
template <int EPL, bool NT>
__device__ __forceinline__ void load_idx16(uint32_t (&id)[EPL], const uint16_t* ip) {
    uint32_t lo, hi = 0;
    if (EPL == 4) { const uint2 t = NT ? stream_load((const uint2*)ip) : *(const uint2*)ip; lo = t.x; hi = t.y; }
    else          { lo = NT ? __builtin_nontemporal_load((const uint32_t*)ip) : *(const uint32_t*)ip; }
    const uint32_t h[4] = {lo & 0xFFFFu, lo >> 16, hi & 0xFFFFu, hi >> 16};
#pragma unroll
    for (int j = 0; j < EPL; ++j) id[j] = idx16_expand(h[j]);
}
template <typename S, int EPL, bool NT>
__device__ __forceinline__ void load_chunk16(ChunkRegs<S, EPL>& c, const vican_graph_t& g, int k, int lane) {
    typedef typename Vec<S>::type V;
    const S* blk = (const S*)g.blk;
    const size_t pbase = (size_t)k * 9 * g.slots + (size_t)lane * EPL;
    const uint16_t* ip = g.idx16 + (size_t)k * g.slots + (size_t)lane * EPL;
#pragma unroll
    for (int p = 0; p < 9; ++p) c.m[p] = NT ? stream_load((const V*)(blk + pbase + (size_t)p * g.slots)) : *(const V*)(blk + pbase + (size_t)p * g.slots);
    uint32_t lo, hi = 0;
    if (EPL == 4) { const uint2 t = NT ? stream_load((const uint2*)ip) : *(const uint2*)ip; lo = t.x; hi = t.y; }
    else          { lo = NT ? __builtin_nontemporal_load((const uint32_t*)ip) : *(const uint32_t*)ip; }
    const uint32_t h[4] = {lo & 0xFFFFu, lo >> 16, hi & 0xFFFFu, hi >> 16};
#pragma unroll
    for (int j = 0; j < EPL; ++j) c.id[j] = idx16_expand(h[j]);
}

template <typename S> __device__ __forceinline__ S pre_scale(double v, double scale);
template <> __device__ __forceinline__ float pre_scale<float>(double v, double) { return (float)v; }
template <> __device__ __forceinline__ double pre_scale<double>(double v, double) { return v; }
// Contribution -> 64-bit fixed point.  f64 blocks: magic-number conversion (vican_common.h to_fix).
// f32 blocks: the raw bit pattern of fma(v, scale, 1.5*2^52) WITHOUT subtracting the bias - a sum of N
// patterns is off by N * 0x4338'0000'0000'0000, which only touches bits 48..63, and the true total
// (|.| < 2^46 by the choice of scale in fx_finish) is the sign-extended low 48 bits (fix_total).  One VALU
// instruction less per contribution; a 24-bit f32 product is still represented without loss down to
// 2^-46 of the TOTAL bound.
template <typename S> __device__ __forceinline__ u64 fix_of(S v, double scale);
template <> __device__ __forceinline__ u64 fix_of<double>(double v, double scale) { return to_fix(v, scale); }
template <> __device__ __forceinline__ u64 fix_of<float>(float v, double scale) {
    return (u64)__double_as_longlong(fma((double)v, scale, 6755399441055744.0));
}
template <typename S> __device__ __forceinline__ long long fix_total(long long s);
template <> __device__ __forceinline__ long long fix_total<float>(long long s) { return (long long)((u64)s << 16) >> 16; }
template <> __device__ __forceinline__ long long fix_total<double>(long long s) { return s; }

// a0 b0 + a1 b1 + a2 b2 with the roundings spelled out (one product, two fused multiply-adds).  The chunk body
// below is instantiated once per register set of the ping-pong ring, and WHICH instance processes a given chunk
// depends on the order the tickets are drawn; left to -ffp-contract=fast the two instances were compiled to
// different mixes of (packed) mul/add/fma for one pair of outputs in one instantiation (<float,768,0,512>), i.e.
// launches differed in the last f32 bit of z components (1,1),(1,2) about once in 30 - found by repetition.
template <typename S>
__device__ __forceinline__ S dot3(S a0, S b0, S a1, S b1, S a2, S b2) {
#ifdef VICAN_DOT3_PLAIN          /* A/B only: lets the compiler contract (and reproduces the nondeterminism) */
    return a0 * b0 + a1 * b1 + a2 * b2;
#endif
#pragma clang fp contract(off)
    // (measured against mul/mul/fma/add and mul/mul/mul/add/add formulations and against the compiler's own choice:
    //  all within the +-3 % run-to-run spread of the launch time)
    S t = a0 * b0;
    t = __builtin_elementwise_fma(a1, b1, t);
    return __builtin_elementwise_fma(a2, b2, t);
}

// scale = 2^e with  c*2^e <= 2^47  (one contribution; lanes pre-sum <= 4 of them: < 2^51)  and
// c*n_add*2^e <= 2^61 (one accumulator)
__device__ __host__ inline double fix_scale(double c, double n_add, double* inv, int bits = 47) {
    c = c > 1e-300 ? c : 1e-300;
    int e = bits - (int)ceil(log2(c));
    const int e2 = 61 - (int)ceil(log2(c * (n_add > 1 ? n_add : 1)));
    e = e < e2 ? e : e2;
    e = e > 1000 ? 1000 : (e < -1000 ? -1000 : e);
    *inv = ldexp(1.0, -e);
    return ldexp(1.0, e);
}

// ---- double-word fixed point (translation stage: CG product, right-hand side) ----------------------------------------
// The reference's translation solve is scipy's f64 CSR product: every TERM w p keeps 53 bits.  One 64-bit accumulator
// resolves 49 bits below a GLOBAL bound on max |w p| - rows whose terms are far below that bound (mild weights already
// span 2^14, the CG direction varies over the graph) lose exactly those bits, and the loosely converged CG (rtol 1e-5 on a
// singular Laplacian) shows it: 111 iterations on the large_shop-scale golden where the reference's own window is 101-106.
// Contributions are therefore split into two integers that are accumulated separately:
//     v * scale = hi + lo * 2^-lob ,   hi = rint(v * scale)  (|hi| < 2^51),   lo = rint((v * scale - hi) * 2^lob)
// hi through the magic-number conversion as before; the residual v * scale - hi is EXACT (fma; scale is a power of two),
// so the pair carries v to 49 + lob bits below the bound (lob = 48 unless an accumulator takes more than 2^14 adds) - a
// term 2^-44 below the bound still has its 53 bits.  Sums of hi and of lo are exact integer sums (associative: bit-
// reproducible whatever the order), and (sum hi, sum lo) is rounded to f64 ONCE per output (fix2_value).
struct Fix2 { u64 hi, lo; };
__device__ __forceinline__ Fix2 to_fix2(double v, double scale, double lo_scale) {
    const double magic = 6755399441055744.0;          // 1.5 * 2^52
    const double t = fma(v, scale, magic);
    const double h = t - magic;                       // rint(v * scale), exact
    Fix2 f;
    f.hi = (u64)(__double_as_longlong(t) - __double_as_longlong(magic));
    f.lo = to_fix(fma(v, scale, -h), lo_scale);       // exact residual in [-1/2, 1/2] -> integer below 2^(lob-1)
    return f;
}
// bits of the lo word for accumulators that take up to n_add contributions: |sum lo| <= n_add 2^(lob-1) < 2^62
__device__ __host__ inline int fix2_lo_bits(double n_add) {
    const int h = (int)ceil(log2(n_add > 1 ? n_add : 1));
    const int b = 62 - h;
    return b > 48 ? 48 : (b < 8 ? 8 : b);
}
// (sum hi + sum lo 2^-lob) / scale as one double: lo's carry moves into hi, the fraction is added last
__device__ __forceinline__ double fix2_value(long long hi, long long lo, int lob, double inv) {
    const long long carry = lo >> lob;                // floor
    hi += carry; lo -= carry << lob;                  // 0 <= lo < 2^lob
    return ((double)hi + ldexp((double)lo, -lob)) * inv;
}

// Sums over the slabs of many workgroups: the lo word of every slab is normalised into [0, 2^lob) first (its carry moves
// to hi) and hi is summed in two halves (hi >> 32, hi & 0xFFFFFFFF), so that no sum can overflow whatever the number of
// slabs and however close to the bound the terms are.  (top, bot, lo) -> double, rounded once: the two integer parts are
// exact doubles, the rounding error of their sum is recovered (two-sum) and joins the fraction.
struct Fix3 { long long top, bot, lo; };
__device__ __forceinline__ void fix3_add(Fix3& a, long long h, long long l, int lob) {
    const long long c = l >> lob;
    h += c; l -= c << lob;
    a.top += h >> 32; a.bot += h & 0xFFFFFFFFll; a.lo += l;
}
__device__ __forceinline__ double fix3_value(long long top, long long bot, long long lo, int lob, double inv) {
    const long long c = lo >> lob;
    bot += c; lo -= c << lob;
    const double x = (double)top * 4294967296.0, y = (double)bot;
    const double sxy = x + y, bb = sxy - x, err = (x - (sxy - bb)) + (y - bb);
    return (sxy + (err + ldexp((double)lo, -lob))) * inv;
}

// Sum of a 64-bit value over the aligned group of `ncopy` (power of two <= 32) neighbouring lanes, in every lane of the
// group: DPP quad permutes (lane ^ 1, lane ^ 2), mirrors inside 8 and 16 lanes, one cross-row shuffle for 32.  Used to
// fold the lane-striped fixed-point row accumulators with one LDS read per lane instead of ncopy dependent ones.
template <int CTRL>
__device__ __forceinline__ u64 dpp_u64(u64 v) {
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), CTRL, 0xF, 0xF, true);
    return ((u64)(uint32_t)hi << 32) | (uint32_t)lo;
}
// Sum of a double over the 64 lanes of a wavefront WITHOUT the LDS pipe, in a fixed order (deterministic): DPP inside the rows
// of 16 lanes (lane ^ 1, lane ^ 2, half-row mirror, row mirror), then row_bcast15 / row_bcast31 carry the row totals
// upwards; the total arrives in lane 63 and is broadcast with v_readlane.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_f64(double v) {
    const u64 b = (u64)__double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, CTRL, ROW_MASK, 0xF, false);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), CTRL, ROW_MASK, 0xF, false);
    return __longlong_as_double((long long)(((u64)(uint32_t)hi << 32) | (uint32_t)lo));
}
__device__ __forceinline__ double wave_total(double v) {
    v += dpp_f64<0xB1, 0xF>(v);                     // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E, 0xF>(v);                     // quad_perm [2,3,0,1]
    v += dpp_f64<0x141, 0xF>(v);                    // row_half_mirror
    v += dpp_f64<0x140, 0xF>(v);                    // row_mirror: every lane of a row holds the row's total
    v += dpp_f64<0x142, 0xA>(v);                    // row_bcast15 into rows 1 and 3 (other rows add the `old` value 0)
    v += dpp_f64<0x143, 0xC>(v);                    // row_bcast31 into rows 2 and 3: lane 63 = total
    const u64 b = (u64)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, 63), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), 63);
    return __longlong_as_double((long long)(((u64)hi << 32) | lo));
}
// Sums of THREE doubles over the 64 lanes of a wavefront in one butterfly (the row sums of a one-row chunk: vican_wtrans.hip).
// Three wave_total() calls cost ~100 VALU instructions (six DPP steps of two moves and an add each, the `old` operands of the
// masked row_bcast steps) - a third of the CG product's instruction count, and that kernel is bound by VALU issue (rocprofv3
// counters, profiles/r04_cg_counters.json).  Here the three streams share the butterfly: after the first exchange (lane ^ 1) even
// lanes carry a, odd lanes b; after the second (lane ^ 2) lanes = 0, 1 (mod 4) carry the quad's sums of a, b and lanes = 2, 3
// (mod 4) the quad's sum of c; row_ror:4 / row_ror:8 add the four quads of a row of 16 lanes without moving a stream off its
// lanes, v_permlane16_swap / v_permlane32_swap (gfx950) add the four rows.  Result: lane l holds the total of stream
// min(l mod 4, 2) - lanes 0, 1, 2 the totals of a, b, c.  Fixed order: deterministic.  ~35 VALU instructions.
template <int CTRL>
__device__ __forceinline__ double dpp_f64_all(double v) {        // every lane has a valid source lane: no `old` operand
    const u64 b = (u64)__double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)b, CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(uint32_t)(b >> 32), CTRL, 0xF, 0xF, true);
    return __longlong_as_double((long long)(((u64)(uint32_t)hi << 32) | (uint32_t)lo));
}
__device__ __forceinline__ double rows_total(double v) {         // v + the same lane of the other three rows of 16 lanes
    {
        const u64 b = (u64)__double_as_longlong(v);
        const auto lo = __builtin_amdgcn_permlane16_swap((uint32_t)b, (uint32_t)b, false, false);
        const auto hi = __builtin_amdgcn_permlane16_swap((uint32_t)(b >> 32), (uint32_t)(b >> 32), false, false);
        v = __longlong_as_double((long long)(((u64)hi[0] << 32) | lo[0])) + __longlong_as_double((long long)(((u64)hi[1] << 32) | lo[1]));
    }
    {
        const u64 b = (u64)__double_as_longlong(v);
        const auto lo = __builtin_amdgcn_permlane32_swap((uint32_t)b, (uint32_t)b, false, false);
        const auto hi = __builtin_amdgcn_permlane32_swap((uint32_t)(b >> 32), (uint32_t)(b >> 32), false, false);
        v = __longlong_as_double((long long)(((u64)hi[0] << 32) | lo[0])) + __longlong_as_double((long long)(((u64)hi[1] << 32) | lo[1]));
    }
    return v;
}
__device__ __forceinline__ double wave_total3(const double a, const double b, const double c, const int lane) {
    const bool b0 = lane & 1, b1 = lane & 2;
    const double x = (b0 ? b : a) + dpp_f64_all<0xB1>(b0 ? a : b);       // quad_perm [1,0,3,2]: even lanes a, odd lanes b
    const double c1 = c + dpp_f64_all<0xB1>(c);
    double y = (b1 ? c1 : x) + dpp_f64_all<0x4E>(b1 ? x : c1);           // quad_perm [2,3,0,1]: lanes 0, 1 (mod 4) a, b; lanes 2, 3 c
    y += dpp_f64_all<0x124>(y);                                           // row_ror:4
    y += dpp_f64_all<0x128>(y);                                           // row_ror:8: the row's four quads
    return rows_total(y);
}
__device__ __forceinline__ double lane_bcast(double v, const int src_lane) {
    const u64 b = (u64)__double_as_longlong(v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)b, src_lane), hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(b >> 32), src_lane);
    return __longlong_as_double((long long)(((u64)hi << 32) | lo));
}

__device__ __forceinline__ u64 stripe_sum(u64 v, const int ncopy) {
    if (ncopy >= 2) v += dpp_u64<0xB1>(v);           // quad_perm [1,0,3,2]
    if (ncopy >= 4) v += dpp_u64<0x4E>(v);           // quad_perm [2,3,0,1]
    if (ncopy >= 8) v += dpp_u64<0x141>(v);          // row_half_mirror: the other quad's total
    if (ncopy >= 16) v += dpp_u64<0x140>(v);         // row_mirror: the other half row's total
    if (ncopy >= 32) v += (u64)__shfl_xor((long long)v, 16, 64);
    return v;
}
