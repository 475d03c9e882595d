// vican_sweep.hip - edge layout + THE hot kernel of the solver: the fused timestep-major
// sweep over the chunked CSR-of-3x3-blocks (operator z = R~ Lambda_T^-1 R~^T x and the
// per-timestep dual update).  Design notes: DESIGN.md section 3.
//
// What the sweep is built around (measured on MI355X, tools/lds_atomic_bench.hip):
//   * HBM-bound streaming of the block planes: 9 x 16 B coalesced loads per lane per chunk,
//     next chunk prefetched into a second register set while the current one is processed;
//   * LDS float atomics are the scarce resource: ds_add_f64 costs 9 clk/wave-instr without
//     conflicts, 23 with random camera indices and serialises completely when all lanes hit
//     one address; ds_add_f32 is ~190 clk; ds_add_u64 is 7.7 (12 with random cameras).
//     => every accumulator is a 64-bit FIXED-POINT integer: fastest atomic on this chip
//     and, because integer addition is associative, every sweep is bit-reproducible;
//   * the row accumulators (y_t) are striped over `n_copy` lane-indexed copies so the 64
//     lanes of a wavefront that sit in the same timestep row never share an address;
//   * phase 2 (w_t = Lambda_T^-1 y_t) runs on 9*rows threads from LDS-staged duals.
#include "vican_common.h"

// Build layout.  Compiled as ONE translation unit this file contains everything.  The in-tree build
// (vican_amd/_lib.py) splits it to compile in parallel - the hot kernel has ~190 instantiations:
//   -DVICAN_SWEEP_SPLIT                       everything except the hot kernel; the sweep modes are reached through
//                                             vican_sweep_part_<mode>()
//   -DVICAN_SWEEP_SPLIT -DVICAN_SWEEP_PART=m  only the hot kernel, its launchers and vican_sweep_part_<m>()
#if defined(VICAN_SWEEP_PART) && !defined(VICAN_SWEEP_SPLIT)
#error "VICAN_SWEEP_PART needs VICAN_SWEEP_SPLIT"
#endif
#define SWEEP_PART_ARGS const vican_graph_t* g, const double* lamT_inv, const double* x, u64* zpart, double* Rt, \
                        double* lamT_out, const double* rnorm, double* fx, void* stream
#ifndef VICAN_SWEEP_PART

#include <algorithm>
#include <vector>
#include "vican_sweep_common.h"
thread_local char g_vican_err[512] = "";
extern "C" const char* vican_last_error(void) { return g_vican_err; }
extern "C" int vican_abi_version(void) { return VICAN_ABI_VERSION; }

thread_local uint32_t* g_vican_abort_word = nullptr;
thread_local unsigned long long g_vican_sync_ticks = 200000000ull;          // 2 s of the 100 MHz real-time counter
extern "C" int vican_set_barrier_abort(uint32_t* abort_word, int64_t timeout_us) {
    g_vican_abort_word = abort_word;
    g_vican_sync_ticks = timeout_us > 0 ? (unsigned long long)timeout_us * 100ull : 200000000ull;
    return VICAN_OK;
}
// Can `grid` workgroups of this kernel be resident at once on an idle device?  (Necessary for its grid barriers; the
// bounded spin of vican_grid_sync covers a device that is NOT idle.)
// Every cooperative launcher asks on EVERY launch; the answer is a property of (kernel, block size, LDS bytes, device) - the
// workgroups one compute unit holds - and is kept per such key (the occupancy query costs ~2 us of host time on the critical
// path of an operator application; statics inside the launchers, as round 5 had them, were not keyed by the device).
int vican_coresident_ok(const void* kernel, int block_threads, size_t lds_bytes, int grid, const char* who) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return set_err(VICAN_ERR_LAUNCH, "%s: cannot query the occupancy of a cooperative kernel", who);
    struct Key { const void* k; int bt; size_t lds; int dev; long long capacity; };
    static thread_local std::vector<Key> seen;              // (per host thread, like the gate and the launch timer: no lock)
    long long capacity = -1;
    for (const Key& e : seen)
        if (e.k == kernel && e.bt == block_threads && e.lds == lds_bytes && e.dev == dev) { capacity = e.capacity; break; }
    if (capacity < 0) {
        int n_cu = 0, per_cu = 0;
        if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block_threads, lds_bytes) != hipSuccess)
            return set_err(VICAN_ERR_LAUNCH, "%s: cannot query the occupancy of a cooperative kernel", who);
        capacity = (long long)per_cu * n_cu;
        if (seen.size() < 256) seen.push_back(Key{kernel, block_threads, lds_bytes, dev, capacity});
    }
    if (capacity < grid)
        return set_err(VICAN_ERR_CAPACITY, "%s: the grid of this cooperative kernel cannot be co-resident on the device", who);
    return VICAN_OK;
}

// diagnostic: workgroups that only stay resident (tests of the bounded grid barriers)
__global__ void occupy_kernel(unsigned long long ticks, int lds_words) {
    extern __shared__ unsigned int occ_lds[];
    if (lds_words > 0) occ_lds[threadIdx.x % lds_words] = threadIdx.x;          // (keeps the allocation)
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
extern "C" int vican_test_occupy(int32_t n_wg, int32_t threads, int32_t lds_bytes, int64_t microseconds, void* stream) {
    if (n_wg <= 0 || threads <= 0 || threads > 1024 || lds_bytes < 0 || lds_bytes > 160 * 1024 || microseconds < 0)
        return set_err(VICAN_ERR_ARG, "vican_test_occupy: bad argument");
    static int conf = 0;
    if (lds_bytes > conf) { hipFuncSetAttribute((const void*)occupy_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes); conf = lds_bytes; }
    hipLaunchKernelGGL(occupy_kernel, dim3(n_wg), dim3(threads), (size_t)lds_bytes, (hipStream_t)stream,
                       (unsigned long long)microseconds * 100ull, lds_bytes / 4);
    LAUNCH_CHECK("vican_test_occupy");
    return VICAN_OK;
}

thread_local const int32_t* g_vican_gate = nullptr;
extern "C" int vican_set_gate(const int32_t* gate) { g_vican_gate = gate; return VICAN_OK; }
thread_local hipEvent_t g_vican_ev_start = nullptr, g_vican_ev_stop = nullptr;
extern "C" int vican_set_launch_events(void* start_event, void* stop_event) {
    g_vican_ev_start = (hipEvent_t)start_event; g_vican_ev_stop = (hipEvent_t)stop_event;
    return VICAN_OK;
}

// ---------------------------------------------------------------------------
// host-side planning + LDS budget
// ---------------------------------------------------------------------------
extern "C" int vican_plan_chunks(int32_t n_time, const int32_t* rp, int32_t slots, int32_t max_rows,
                                 int32_t* out, int32_t cap) {
    if (n_time < 0 || !rp || slots <= 0 || max_rows <= 0 || max_rows > 65535)
        return set_err(VICAN_ERR_ARG, "vican_plan_chunks: bad argument");
    int32_t nc = 0, r = 0;
    while (r < n_time) {
        if (out) { if (nc >= cap) return set_err(VICAN_ERR_ARG, "vican_plan_chunks: output too small"); out[nc] = r; }
        const int32_t e0 = rp[r];
        int32_t r1 = r;
        while (r1 < n_time && (r1 - r) < max_rows && (rp[r1 + 1] - e0) <= slots) ++r1;
        if (r1 == r) return set_err(VICAN_ERR_CAPACITY, "vican_plan_chunks: a timestep row has more edges than a chunk holds");
        r = r1;
        ++nc;
    }
    if (out) { if (nc >= cap) return set_err(VICAN_ERR_ARG, "vican_plan_chunks: output too small"); out[nc] = n_time; }
    return nc;
}

__global__ void pack_idx16_kernel(const uint32_t* __restrict__ idx, uint16_t* __restrict__ out, long long n) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const uint32_t v = idx[i];
        out[i] = v == VICAN_PAD_SLOT ? (uint16_t)0xFFFF : (uint16_t)((v & 0x3FFu) | ((v >> 16) << 10));
    }
}
extern "C" int vican_pack_idx16(const vican_graph_t* g, uint16_t* out, void* stream) {
    if (int rc = vican_check_graph(g, "vican_pack_idx16")) return rc;
    if (!out) return set_err(VICAN_ERR_ARG, "vican_pack_idx16: null pointer");
    if (g->layout != VICAN_LAYOUT_WAVE || (g->n_cam == 1024 && g->max_rows == 64))
        return set_err(VICAN_ERR_ARG, "vican_pack_idx16: needs a wave-layout graph (with <= 63 rows per chunk if it has 1024 cameras)");
    const long long n = (long long)g->n_chunk * g->slots;
    if (n == 0) return VICAN_OK;
    hipLaunchKernelGGL(pack_idx16_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, g->idx, out, n);
    LAUNCH_CHECK("vican_pack_idx16");
    return VICAN_OK;
}

// vican_graph_t.w32: the translation weights of a wave-layout graph with 4 edges per lane (w: [n_chunk][256] doubles in the
// layout's slot_pos8 order - [half][lane][2]) as float32 in PLAIN slot order ([lane][4]); *inexact (device word, zeroed here)
// becomes 1 if any weight is not a float32 value - the copy must not be used then.
__global__ void pack_w32_kernel(const double* __restrict__ w, float* __restrict__ out, long long n, int32_t* inexact) {
    bool bad = false;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const long long k = i >> 8;
        const int s = (int)(i & 255), lane = s >> 2, j = s & 3;
        const double v = w[k * 256 + (j < 2 ? lane * 2 + j : 128 + lane * 2 + (j - 2))];
        const float f = (float)v;
        out[i] = f;
        bad = bad || (double)f != v;
    }
    if (bad) *inexact = 1;
}
extern "C" int vican_pack_w32(const vican_graph_t* g, const double* w, float* out, int32_t* inexact, void* stream) {
    if (int rc = vican_check_graph(g, "vican_pack_w32")) return rc;
    if (!w || !out || !inexact) return set_err(VICAN_ERR_ARG, "vican_pack_w32: null pointer");
    if (g->layout != VICAN_LAYOUT_WAVE || g->slots != 256) return set_err(VICAN_ERR_ARG, "vican_pack_w32: needs a wave-layout graph with 4 edges per lane");
    if (hipMemsetAsync(inexact, 0, sizeof(int32_t), (hipStream_t)stream) != hipSuccess) return set_err(VICAN_ERR_LAUNCH, "vican_pack_w32: memset failed");
    const long long n = (long long)g->n_chunk * g->slots;
    if (n == 0) return VICAN_OK;
    hipLaunchKernelGGL(pack_w32_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, w, out, n, inexact);
    LAUNCH_CHECK("vican_pack_w32");
    return VICAN_OK;
}

// The same for camera tiles that must SHARE their chunking (vican_tiled_op): chunk k covers the same timestep rows in every
// tile; a row joins the open chunk while every tile's edges of the chunk still fit its slots.  rps: n_tile row-pointer arrays.
extern "C" int vican_plan_chunks_multi(int32_t n_time, int32_t n_tile, const int32_t* const* rps, int32_t slots, int32_t max_rows,
                                       int32_t* out, int32_t cap) {
    if (n_time < 0 || n_tile <= 0 || !rps || slots <= 0 || max_rows <= 0 || max_rows > 65535)
        return set_err(VICAN_ERR_ARG, "vican_plan_chunks_multi: bad argument");
    for (int k = 0; k < n_tile; ++k) if (!rps[k]) return set_err(VICAN_ERR_ARG, "vican_plan_chunks_multi: null row pointer array");
    int32_t nc = 0, r = 0;
    while (r < n_time) {
        if (out) { if (nc >= cap) return set_err(VICAN_ERR_ARG, "vican_plan_chunks_multi: output too small"); out[nc] = r; }
        int32_t r1 = r;
        while (r1 < n_time && (r1 - r) < max_rows) {
            bool fits = true;
            for (int k = 0; k < n_tile && fits; ++k) fits = (rps[k][r1 + 1] - rps[k][r]) <= slots;
            if (!fits) break;
            ++r1;
        }
        if (r1 == r) return set_err(VICAN_ERR_CAPACITY, "vican_plan_chunks_multi: a timestep row has more edges than a chunk holds");
        r = r1;
        ++nc;
    }
    if (out) { if (nc >= cap) return set_err(VICAN_ERR_ARG, "vican_plan_chunks_multi: output too small"); out[nc] = n_time; }
    return nc;
}

// A shared chunking of camera tiles over CONSECUTIVE rows pads badly when a row's edges split evenly over the tiles: a chunk
// takes a row only while EVERY tile's edges still fit, and with 4 tiles of 62.5 +- 6.8 edges per row four rows (250 +- 14) fit
// all four tiles one time in five - three rows per chunk, 1.33 slots per edge on the wide benchmark graph.  Rows need not be
// consecutive: the operator is a sum over rows.  vican_plan_rows_multi fills one chunk at a time from a pool of the next
// `window` unassigned rows: the first of the pool opens the chunk; while the fullest tile still has room for two average rows
// the chunk takes the row of the pool that leaves its tiles most evenly filled (loads in units of the tile's mean row), then
// the largest row that still fits - 1.03 slots per edge on that graph (four rows in every chunk), against 1.06 for first-fit
// into a window of open chunks.  Returns the row ORDER (perm[new] = old row; rows of a chunk ascending) with the chunk
// boundaries in the new numbering; window = 1 is the consecutive chunking.  The caller builds the tiles from the rows in this
// order and undoes it on whatever it hands back per row (vican_amd.device.TiledGraph.row_perm).
extern "C" int vican_plan_rows_multi(int32_t n_time, int32_t n_tile, const int32_t* const* rps, int32_t slots, int32_t max_rows,
                                     int32_t window, int32_t* perm_out, int32_t* chunk_row0_out, int32_t cap) {
    if (n_time < 0 || n_tile <= 0 || n_tile > 64 || !rps || slots <= 0 || max_rows <= 0 || max_rows > 65535 || window < 1 || !perm_out || !chunk_row0_out)
        return set_err(VICAN_ERR_ARG, "vican_plan_rows_multi: bad argument");
    for (int k = 0; k < n_tile; ++k) if (!rps[k]) return set_err(VICAN_ERR_ARG, "vican_plan_rows_multi: null row pointer array");
    std::vector<double> inv_mean((size_t)n_tile);               // 1 / (mean edges per row) of every tile
    for (int k = 0; k < n_tile; ++k) {
        const double m = n_time ? (double)(rps[k][n_time] - rps[k][0]) / n_time : 0.0;
        inv_mean[k] = 1.0 / (m > 1e-9 ? m : 1e-9);
    }
    auto count = [&](int32_t r, int k) { return rps[k][r + 1] - rps[k][r]; };
    for (int32_t r = 0; r < n_time; ++r)
        for (int k = 0; k < n_tile; ++k)
            if (count(r, k) > slots) return set_err(VICAN_ERR_CAPACITY, "vican_plan_rows_multi: a timestep row has more edges than a chunk holds");
    std::vector<int32_t> pool, sum((size_t)n_tile), rows;       // pool: unassigned rows, ascending
    int32_t next = 0, pos = 0, nc = 0;
    auto refill = [&]() { while ((int)pool.size() < window && next < n_time) pool.push_back(next++); };
    refill();
    while (!pool.empty()) {
        if ((int64_t)nc + 2 > cap) return set_err(VICAN_ERR_ARG, "vican_plan_rows_multi: output too small");
        rows.assign(1, pool.front());
        for (int k = 0; k < n_tile; ++k) sum[k] = count(pool.front(), k);
        pool.erase(pool.begin());
        refill();
        while ((int)rows.size() < max_rows && !pool.empty()) {
            double more = 1e300;                                // average rows the fullest tile still holds
            for (int k = 0; k < n_tile; ++k) { const double m = (slots - sum[k]) * inv_mean[k]; if (m < more) more = m; }
            int best = -1;
            double best_score = 0.0;
            for (int i = 0; i < (int)pool.size(); ++i) {
                const int32_t r = pool[i];
                bool fits = true;
                double hi = 0.0, mean = 0.0;
                int64_t total = 0;
                for (int k = 0; k < n_tile; ++k) {
                    const int32_t c = sum[k] + count(r, k);
                    if (c > slots) { fits = false; break; }
                    const double load = c * inv_mean[k];
                    if (load > hi) hi = load;
                    mean += load;
                    total += c;
                }
                if (!fits) continue;
                const double score = more < 2.0 ? -(double)total : hi - mean / n_tile;
                if (best < 0 || score < best_score) { best = i; best_score = score; }
            }
            if (best < 0) break;
            for (int k = 0; k < n_tile; ++k) sum[k] += count(pool[best], k);
            rows.push_back(pool[best]);
            pool.erase(pool.begin() + best);
            refill();
        }
        std::sort(rows.begin(), rows.end());
        chunk_row0_out[nc++] = pos;
        for (int32_t r : rows) perm_out[pos++] = r;
    }
    chunk_row0_out[nc] = pos;
    return nc;
}

// (dynamic LDS a kernel may ask for: the 160 KB of a CU minus room for the kernels' few static __shared__ words - a graph that
//  fitted the 160 KB exactly, C = 1024 in f64 with two accumulator copies, failed at hipFuncSetAttribute)
extern "C" int64_t vican_lds_limit_bytes(void) { return 160 * 1024 - 1024; }


// operator sweep: x table (storage type) + z accumulators (u64) per camera; per row:
// y sums (f64) + w (storage type) + n_copy striped u64 accumulators
// camera planes are padded to a compile-time stride CP (256 or 1024 entries) so that the nine
// component planes are reached with LDS immediate offsets instead of per-access address math
extern "C" int64_t vican_sweep_lds_bytes(int32_t n_cam, int32_t max_rows, int32_t storage, int32_t n_copy) {
    const int64_t s = ssize(storage), cp = plane_stride(n_cam);
    return 9LL * cp * (s + 8) + (int64_t)max_rows * (72 + 9 * s + 72LL * n_copy) + 256;
}
extern "C" int32_t vican_max_rows_for(int32_t n_cam, int32_t storage, int32_t n_copy) {
    const int64_t lim = vican_lds_limit_bytes(), s = ssize(storage);
    if (n_cam > 1024) return 0;
    int64_t a = (lim - 256 - 9LL * plane_stride(n_cam) * (s + 8)) / (72 + 9 * s + 72LL * n_copy);
    int64_t b = (lim - 256 - 120LL * n_cam) / (48LL * n_copy + 144); // rhs kernel (rhs_lds_bytes, vican_common.h)
    int64_t c = (lim - 256 - 72LL * n_cam) / (96LL * n_copy + 96);   // CG sweep (cg_lds_bytes)
    int64_t m = a < b ? a : b; if (c < m) m = c; if (m > 65535) m = 65535;
    return (int32_t)m;
}

extern "C" int64_t vican_wsweep_lds_bytes(int32_t n_cam, int32_t max_rows, int32_t storage, int32_t n_copy, int32_t n_waves);

int vican_check_graph(const vican_graph_t* g, const char* who) {
    if (!g || g->n_cam <= 0 || g->n_cam > 65535 || g->n_time < 0 || g->n_chunk < 0 || !g->idx ||
        !g->chunk_row0 || g->n_wg <= 0)
        return set_err(VICAN_ERR_ARG, "%s: bad graph descriptor", who);
    const int epl = (g->storage == VICAN_STORE_F32) ? 4 : 2;
    const int nc = g->n_copy;
    if (nc < 1 || nc > 32 || (nc & (nc - 1))) return set_err(VICAN_ERR_ARG, "%s: n_copy must be a power of two <= 32", who);
    if (g->layout == VICAN_LAYOUT_WAVE) {
        if ((g->wg_waves != 4 && g->wg_waves != 8 && g->wg_waves != 12 && g->wg_waves != 16) || g->block_threads != 64 * g->wg_waves || g->slots != 64 * epl)
            return set_err(VICAN_ERR_ARG, "%s: wave layout needs wg_waves in {4, 8, 12}, block_threads = 64 wg_waves, slots = 64 * (16 / sizeof(storage))", who);
        if (g->n_cam > 1024 || g->max_rows <= 0 || g->max_rows > 64 ||
            vican_wsweep_lds_bytes(g->n_cam, g->max_rows, g->storage, nc, g->wg_waves) > vican_lds_limit_bytes())
            return set_err(VICAN_ERR_CAPACITY, "%s: camera tables / row staging do not fit in LDS (wave layout)", who);
        return 0;
    }
    if (g->layout != VICAN_LAYOUT_BLOCK) return set_err(VICAN_ERR_ARG, "%s: unknown layout", who);
    if ((g->block_threads != 256 && g->block_threads != 512 && g->block_threads != 768 && g->block_threads != 1024) ||
        g->slots != g->block_threads * epl)
        return set_err(VICAN_ERR_ARG, "%s: slots must be block_threads * (16 / sizeof(storage))", who);
    if (!g->blk) {
        // a layout that only carries CG weights (camera tiling: the rotation blocks live in per-tile graphs): the CG sweep's
        // LDS budget alone decides (48 B per camera: C <= ~3300)
        if (g->max_rows <= 0 || cg_lds_bytes(g->n_cam, g->max_rows, nc) > vican_lds_limit_bytes())
            return set_err(VICAN_ERR_CAPACITY, "%s: camera tables / row staging of the CG sweep do not fit in LDS", who);
        return 0;
    }
    if (g->max_rows <= 0 || g->max_rows > vican_max_rows_for(g->n_cam, g->storage, nc))
        return set_err(VICAN_ERR_CAPACITY, "%s: camera tables / row staging do not fit in LDS", who);
    return 0;
}
// entry points that only take the block layout (translation kernels, vican_bip_apply)
int vican_check_block_graph(const vican_graph_t* g, const char* who) {
    if (int rc = vican_check_graph(g, who)) return rc;
    if (g->layout != VICAN_LAYOUT_BLOCK) return set_err(VICAN_ERR_ARG, "%s: needs a block-layout graph", who);
    return 0;
}

// ---------------------------------------------------------------------------
// layout: CSR -> chunked planes
//
// Slot assignment inside a chunk is BANK-AWARE: lane l of the sweep workgroup owns slots
// EPL*l .. EPL*l+EPL-1 and is fed edges whose camera index is congruent to l modulo 32.
// With the camera-side LDS tables stored as planes [component][camera], every gather of
// x_cam, every fixed-point atomic on z_cam and every striped atomic on y_row then hits 32
// distinct banks per 32-lane group - the conflict-free rate instead of the 4x slower
// "random" one (PMC: 75 % of LDS cycles were bank conflicts before this).
// It is also ROW-AWARE: of the n edges a row has in a class, the first EPL*floor(n/EPL) fill
// whole lanes ("pure" lanes: one row per lane), allocated from one end of the class's lane groups;
// the n mod EPL leftovers of all rows are packed continuously into lanes allocated from the other end
// (mirrored at the end so that the leftovers sit in the first, i.e. oldest, wavefronts).  The sweep flushes
// a lane's row sum whenever the row changes inside the lane (masked, but at full instruction cost); with this
// order the lanes that change row sit in few wavefronts and all others skip those flushes (execz branches): 12 x 36 -> about
// 10 x 9 + 2 x 36 row-accumulator atomics per chunk.  Classes that overflow their lanes spill into the
// free slots of the others.
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(64) void plan_slots_kernel(vican_graph_t g, const int32_t* __restrict__ row_ptr,
                                                        const int32_t* __restrict__ col, int32_t* __restrict__ perm,
                                                        int epl) {
    extern __shared__ int32_t sh[];                 // overflow list [slots]
    const int k = blockIdx.x, lane = threadIdx.x;
    int novf = 0;                                   // wave-uniform
    const int r_first = g.chunk_row0[k], r_last = g.chunk_row0[k + 1];
    int32_t* pm = perm + (size_t)k * g.slots;
    for (int s = lane; s < g.slots; s += 64) pm[s] = -1;
    __syncthreads();
    const int G = g.slots / epl / 32;               // lane groups (of 32 lanes) = lanes per camera class
    int pure_lanes = 0, left_cnt = 0;               // lane c (< 32): state of class c
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int r = r_first; r < r_last; ++r) {
        const int e0 = row_ptr[r], e1 = row_ptr[r + 1];
        if (e0 == e1) continue;
        int n_c = 0;                                // lane kk: edges of this row in class kk
        for (int base = e0; base < e1; base += 64) {
            const int e = base + lane;
            const int c = e < e1 ? (col[e] & 31) : -1;
            for (int kk = 0; kk < 32; ++kk) {
                const unsigned long long m = __ballot(c == kk);
                if (lane == kk) n_c += __popcll(m);
            }
        }
        int seen = 0;
        for (int base = e0; base < e1; base += 64) {
            const int e = base + lane;
            const bool valid = e < e1;
            const int c = valid ? (col[e] & 31) : -1;
            int pos = 0, ncls = 0, pl = 0, lc = 0;
            for (int kk = 0; kk < 32; ++kk) {
                const unsigned long long m = __ballot(c == kk);
                const int before = __shfl(seen, kk, 64), nk = __shfl(n_c, kk, 64);
                const int plk = __shfl(pure_lanes, kk, 64), lck = __shfl(left_cnt, kk, 64);
                if (c == kk) { pos = before + __popcll(m & lt); ncls = nk; pl = plk; lc = lck; }
                if (lane == kk) seen += __popcll(m);
            }
            bool spill = false;
            int grp = 0, slot = 0;
            if (valid) {
                const int full = (ncls / epl) * epl, rem = ncls - full;
                const int back_groups = (lc + rem + epl - 1) / epl;         // leftover lane groups in use after this row
                if (pos < full) {                   // pure lane, from the front
                    grp = pl + pos / epl; slot = pos % epl;
                    spill = grp >= G - back_groups;
                } else {                            // leftover stream, from the back
                    const int lr = lc + (pos - full);
                    grp = G - 1 - lr / epl; slot = lr % epl;
                    spill = grp < pl + ncls / epl;
                }
            }
            const unsigned long long sm = __ballot(spill);      // ordered compaction: deterministic layout
            // lane groups are mirrored: the leftover lanes end up in the FIRST wavefronts - the oldest ones, which
            // the SIMD arbiter favours, so the wavefronts with the extra flush work finish first (-3 % vs last)
            if (valid && !spill) pm[(c + 32 * (G - 1 - grp)) * epl + slot] = e;
            if (spill) sh[novf + __popcll(sm & lt)] = e;
            novf += __popcll(sm);
        }
        if (lane < 32) { pure_lanes += n_c / epl; left_cnt += n_c % epl; }
    }
    __syncthreads();
    if (lane == 0 && novf > 0) {                    // spill: few edges; fill free slots from the front (the
        int o = 0;                                  // wavefronts that change rows anyway)
        for (int s = 0; s < g.slots && o < novf; ++s)
            if (pm[s] < 0) pm[s] = sh[o++];
    }
}

// BANK-AWARE order for wave-layout chunks that hold ONE row (dense rows: the stress graph, 250 of 1000 cameras per timestep):
// plan_slots_kernel above fills 32 camera classes of 2 x EPL slots each and drops whatever does not fit (13 % of the edges at
// 98 % fill: a class holds 7.8 +- 2.8 cameras for 8 slots) into free slots anywhere - nearly every 16-lane group of every LDS
// instruction then contains a stranger, and a stranger is a 2-way bank conflict: half of all LDS-array cycles of the sweeps
// were conflict cycles (profiles/r04_cg_counters.json, r04_sweep_counters.json).  What the hardware asks for is weaker than
// "lane = camera mod 32": a 64-bit LDS atomic is served in groups of 16 lanes and conflict-free iff the 16 cameras of a group
// differ mod 16 (64-bit gathers: 32 lanes, mod 32).  So the chunk is seen as NC = 4 EPL cells of 16 lanes (cell k = slot
// j = k / 4 of lanes 16 (k % 4) .. + 15) and the edges of residue r = camera mod 16 go ONE PER CELL, cells 0, 1, 2, ... at lane
// offset r (alternating between the two mod-32 halves of the residue, so that the two cells of a 32-lane gather differ mod 32
// while both halves last): a residue with n_r <= NC members causes no conflict at all.  The members beyond NC of an over-full
// residue take the holes that under-full residues leave - all in the LAST cells - at most one per cell: such a cell costs one
// extra cycle, and the extra cycles of different residues overlap.  Simulated on the stress graph's rows (tools: see
// profiles/NOTES round 5): LDS cycles of a 64-bit atomic per chunk 27.6 -> 22.2 (ideal 16).
// One THREAD per chunk, sequential (a row has <= 256 edges; packing is not on any timed path).
__global__ void plan_slots_wave1_kernel(vican_graph_t g, const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
                                        int32_t* __restrict__ perm, int epl) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= g.n_chunk) return;
    const int r0 = g.chunk_row0[k];
    const int e0 = row_ptr[r0], e1 = row_ptr[r0 + 1];
    int32_t* pm = perm + (size_t)k * g.slots;
    for (int s = 0; s < g.slots; ++s) pm[s] = -1;
    const int NC = 4 * epl;                                     // cells of 16 lanes
    int nA[16], nB[16];
    for (int r = 0; r < 16; ++r) nA[r] = nB[r] = 0;
    for (int e = e0; e < e1; ++e) { const int c = col[e]; if (c & 16) ++nB[c & 15]; else ++nA[c & 15]; }
    auto slot_of = [&](int cell, int off) { const int j = cell >> 2, grp = cell & 3; return (16 * grp + off) * epl + j; };
    // free[cell]: lane offsets (= residues) without a primary entry in that cell
    unsigned freem[16];
    for (int c = 0; c < 16; ++c) {
        unsigned m = 0;
        for (int r = 0; r < 16; ++r) if (nA[r] + nB[r] <= c) m |= 1u << r;
        freem[c] = c < NC ? m : 0u;
    }
    int rkA[16], rkB[16], nextc[16];
    for (int r = 0; r < 16; ++r) { rkA[r] = rkB[r] = 0; nextc[r] = NC - 1; }
    for (int e = e0; e < e1; ++e) {
        const int c = col[e], r = c & 15;
        const bool isB = (c & 16) != 0;
        const int m = nA[r] < nB[r] ? nA[r] : nB[r];
        const int rk = isB ? rkB[r]++ : rkA[r]++;
        const int i = rk < m ? 2 * rk + (isB ? 1 : 0) : 2 * m + (rk - m);       // member index inside the residue
        if (i < NC) { pm[slot_of(i, r)] = e; continue; }
        // excess member: a hole of the highest cell this residue has not used for an excess yet
        int cell = nextc[r];
        while (cell >= 0 && freem[cell] == 0u) --cell;
        if (cell < 0) {                                         // no hole left there: any hole (cells from the top)
            cell = NC - 1;
            while (cell >= 0 && freem[cell] == 0u) --cell;
        } else nextc[r] = cell - 1;
        if (cell >= 0) {
            const int off = __ffs((int)freem[cell]) - 1;
            freem[cell] &= ~(1u << off);
            pm[slot_of(cell, off)] = e;
        }
    }
}

// ROW-MAJOR slot order (vican_graph_t.slot_order == 1): the chunk's edges in CSR order, i.e. lane l holds EPL consecutive
// edges of (mostly) ONE row.  For graphs of short rows: with a few edges per row a (row, camera class) pair holds at most one
// edge, so the bank-aware order above has no pure lanes at all - every lane flushes EPL row sums (9 LDS atomics each) and
// re-reads EPL row operands per chunk - while here a lane changes row about once; the price is that the camera-side
// gathers and atomics hit random banks.
// ... and inside that order, WHICH of a row's slots each of its edges takes is free (a row's set of slots - hence every lane's
// rows - stays as it is): round 5 uses the freedom against LDS bank conflicts on wave-layout chunks.  A 64-bit LDS atomic is
// served in groups of 16 lanes and conflicts where two cameras of a group agree mod 16; in CSR order a group's 16 cameras are
// random: 49 LDS cycles per 64-bit atomic and chunk where 16 would be ideal (simulated, rows of 4-60 edges).  One thread per
// chunk walks the rows in order and gives each edge the first free slot of its row whose cell (slot j of a 16-lane group) does
// not hold its residue yet: 35-42 cycles.  (Pack time only; sequential, O(row length^2) per row.)
__global__ void plan_slots_rows_greedy_kernel(vican_graph_t g, const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
                                              int32_t* __restrict__ perm, int epl) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= g.n_chunk) return;
    const int r0 = g.chunk_row0[k], r1 = g.chunk_row0[k + 1];
    int32_t* pm = perm + (size_t)k * g.slots;
    for (int s = 0; s < g.slots; ++s) pm[s] = -1;
    unsigned cellmask[16];                                      // cell (j, 16-lane group) = j * 4 + group: residues present
    for (int c = 0; c < 16; ++c) cellmask[c] = 0u;
    int pos = 0;
    for (int r = r0; r < r1; ++r) {
        const int e0 = row_ptr[r], e1 = row_ptr[r + 1], deg = e1 - e0;
        for (int e = e0; e < e1; ++e) {
            const unsigned bit = 1u << (col[e] & 15);
            int pick = -1, first_free = -1;
            for (int s = pos; s < pos + deg; ++s) {
                if (pm[s] >= 0) continue;
                if (first_free < 0) first_free = s;
                const int lane = s / epl, j = s - lane * epl;
                if (!(cellmask[j * 4 + (lane >> 4)] & bit)) { pick = s; break; }
            }
            if (pick < 0) pick = first_free;
            const int lane = pick / epl, j = pick - lane * epl;
            cellmask[j * 4 + (lane >> 4)] |= bit;
            pm[pick] = e;
        }
        pos += deg;
    }
}

__global__ void plan_slots_rows_kernel(vican_graph_t g, const int32_t* __restrict__ row_ptr, int32_t* __restrict__ perm) {
    const int k = blockIdx.y, s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= g.slots) return;
    const int e0 = row_ptr[g.chunk_row0[k]], e1 = row_ptr[g.chunk_row0[k + 1]];
    perm[(size_t)k * g.slots + s] = e0 + s < e1 ? e0 + s : -1;
}

template <typename S>
__global__ void pack_edges_kernel(vican_graph_t g, const int32_t* __restrict__ row_ptr,
                                  const int32_t* __restrict__ col, const S* __restrict__ blk_csr,
                                  const S* __restrict__ a_csr, const double* __restrict__ w_csr,
                                  const double* __restrict__ u_csr, const double* __restrict__ v_csr,
                                  S* __restrict__ a_out, double* __restrict__ w_out,
                                  double* __restrict__ u_out, double* __restrict__ v_out,
                                  const int32_t* __restrict__ perm) {
    const int k = blockIdx.y;
    const int s = blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= g.slots) return;
    const int r0 = g.chunk_row0[k], r1 = g.chunk_row0[k + 1];
    S* blk = blk_csr ? (S*)g.blk : nullptr;      // a layout that only carries the translation arrays has no block planes
    uint32_t* idx = (uint32_t*)g.idx;
    const size_t base = (size_t)k * g.slots + s;
    const int s8 = slot_pos8(g, s);                 // position in the 8-byte-per-slot arrays (w, planes of u / v)
    const size_t base8 = (size_t)k * g.slots + s8;
    const int e = perm[base];
    if (e >= 0) {
        int lo = r0, hi = r1;           // row_ptr[lo] <= e < row_ptr[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (row_ptr[mid] <= e) lo = mid; else hi = mid;
        }
        idx[base] = (uint32_t)col[e] | ((uint32_t)(lo - r0) << 16);
        if (blk) {
#pragma unroll
            for (int p = 0; p < 9; ++p) blk[((size_t)k * 9 + p) * g.slots + s] = blk_csr[(size_t)e * 9 + p];
        }
        if (a_out) a_out[base] = a_csr[e];
        if (w_out) w_out[base8] = w_csr[e];
        if (u_out)
            for (int p = 0; p < 3; ++p) u_out[((size_t)k * 3 + p) * g.slots + s8] = u_csr[(size_t)e * 3 + p];
        if (v_out)
            for (int p = 0; p < 3; ++p) v_out[((size_t)k * 3 + p) * g.slots + s8] = v_csr[(size_t)e * 3 + p];
    } else {
        idx[base] = VICAN_PAD_SLOT;
        if (blk) {
#pragma unroll
            for (int p = 0; p < 9; ++p) blk[((size_t)k * 9 + p) * g.slots + s] = (S)0;
        }
        if (a_out) a_out[base] = (S)0;
        if (w_out) w_out[base8] = 0.0;
        if (u_out) for (int p = 0; p < 3; ++p) u_out[((size_t)k * 3 + p) * g.slots + s8] = 0.0;
        if (v_out) for (int p = 0; p < 3; ++p) v_out[((size_t)k * 3 + p) * g.slots + s8] = 0.0;
    }
}

extern "C" int vican_pack_edges(const vican_graph_t* g, const int32_t* row_ptr, const int32_t* col,
                                const void* blk_csr, const void* a_csr, const double* w_csr,
                                const double* u_csr, const double* v_csr, void* a_out, double* w_out,
                                double* u_out, double* v_out, int32_t* perm_ws, void* stream) {
    if (int rc = vican_check_graph(g, "vican_pack_edges")) return rc;
    if (!row_ptr || !col || !perm_ws || (blk_csr && !g->blk)) return set_err(VICAN_ERR_ARG, "vican_pack_edges: null input");
    if (g->n_chunk == 0) return VICAN_OK;
    dim3 grid((g->slots + 255) / 256, g->n_chunk), block(256);
    hipStream_t st = (hipStream_t)stream;
    const int epl = (g->storage == VICAN_STORE_F32) ? 4 : 2;
    if (g->slot_order == 1 && g->layout == VICAN_LAYOUT_WAVE && g->slots == 64 * epl)
        hipLaunchKernelGGL(plan_slots_rows_greedy_kernel, dim3((g->n_chunk + 63) / 64), dim3(64), 0, st, *g, row_ptr, col, perm_ws, epl);
    else if (g->slot_order == 1)
        hipLaunchKernelGGL(plan_slots_rows_kernel, grid, block, 0, st, *g, row_ptr, perm_ws);
    else if (g->layout == VICAN_LAYOUT_WAVE && g->n_chunk == g->n_time && g->slots == 64 * epl)
        hipLaunchKernelGGL(plan_slots_wave1_kernel, dim3((g->n_chunk + 63) / 64), dim3(64), 0, st, *g, row_ptr, col, perm_ws, epl);
    else
        hipLaunchKernelGGL(plan_slots_kernel, dim3(g->n_chunk), dim3(64), (size_t)g->slots * 4, st, *g, row_ptr, col, perm_ws, epl);
    const int32_t* perm = perm_ws;
    if (g->storage == VICAN_STORE_F32)
        hipLaunchKernelGGL(pack_edges_kernel<float>, grid, block, 0, st, *g, row_ptr, col, (const float*)blk_csr,
                           (const float*)a_csr, w_csr, u_csr, v_csr, (float*)a_out, w_out, u_out, v_out, perm);
    else
        hipLaunchKernelGGL(pack_edges_kernel<double>, grid, block, 0, st, *g, row_ptr, col, (const double*)blk_csr,
                           (const double*)a_csr, w_csr, u_csr, v_csr, (double*)a_out, w_out, u_out, v_out, perm);
    LAUNCH_CHECK("vican_pack_edges");
    return VICAN_OK;
}

// ---------------------------------------------------------------------------
// graph constants (computed once per graph at pack time, not per solve):
//   row sums / camera sums of a per-edge scalar, and the block-norm bounds that size
//   the fixed-point scales.
// ---------------------------------------------------------------------------
// Deterministic (order-independent) sums: 64-bit fixed point relative to vmax >= max|val|.
// Persistent workgroups: camera sums are collected in an LDS table over all the chunks a workgroup handles and
// flushed once (the first version issued one global atomic per edge onto n_cam addresses: 2.9 ms for 25 M edges).
template <typename S>
__global__ __launch_bounds__(256) void edge_sums_kernel(vican_graph_t g, const S* __restrict__ val, double scale, double inv,
                                                        double* __restrict__ row_out, u64* __restrict__ cam_acc) {
    extern __shared__ __align__(16) unsigned char lds_raw[];
    u64* cams = (u64*)lds_raw;                        // [n_cam]
    u64* rows = cams + g.n_cam;                       // [max_rows]
    for (int c = threadIdx.x; c < g.n_cam; c += blockDim.x) cams[c] = 0ull;
    for (int k = blockIdx.x; k < g.n_chunk; k += gridDim.x) {
        const int r0 = g.chunk_row0[k], nrows = g.chunk_row0[k + 1] - r0;
        for (int r = threadIdx.x; r < nrows; r += blockDim.x) rows[r] = 0ull;
        __syncthreads();
        const size_t base = (size_t)k * g.slots;
        for (int s = threadIdx.x; s < g.slots; s += blockDim.x) {
            const uint32_t id = g.idx[base + s];
            if (id == VICAN_PAD_SLOT) continue;
            // (an 8-byte array on a 4-slots-per-lane wave layout is the weights w in their permuted order, slot_pos8)
            const u64 f = to_fix((double)val[base + (sizeof(S) == 8 ? slot_pos8(g, s) : s)], scale);
            lds_add_fix(&rows[id >> 16], f);
            lds_add_fix(&cams[id & 0xFFFFu], f);
        }
        __syncthreads();
        for (int r = threadIdx.x; r < nrows; r += blockDim.x) row_out[r0 + r] = (double)(long long)rows[r] * inv;
        __syncthreads();
    }
    __syncthreads();
    for (int c = threadIdx.x; c < g.n_cam; c += blockDim.x)
        if (cams[c] != 0ull) atomicAdd(&cam_acc[c], cams[c]);
}
__global__ void fix_to_double_kernel(int n, const u64* __restrict__ in, double inv, double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (double)(long long)in[i] * inv;
}

extern "C" int vican_edge_sums(const vican_graph_t* g, const void* val, int32_t val_is_f64, double vmax, double* row_sum,
                               double* cam_sum, void* cam_ws, void* stream) {
    if (int rc = vican_check_graph(g, "vican_edge_sums")) return rc;
    if (!val || !row_sum || !cam_sum || !cam_ws || !(vmax >= 0)) return set_err(VICAN_ERR_ARG, "vican_edge_sums: bad argument");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(cam_ws, 0, (size_t)g->n_cam * 8, st) != hipSuccess) return set_err(VICAN_ERR_LAUNCH, "vican_edge_sums: memset failed");
    // one value <= 2^44 so that a camera may collect up to 2^17 rows... bounded by n_time adds
    double c = vmax > 1e-300 ? vmax : 1e-300;
    int e = 47 - (int)ceil(log2(c));
    const int e2 = 61 - (int)ceil(log2(c * (g->n_time > 1 ? (double)g->n_time : 1.0)));
    if (e2 < e) e = e2;
    const double scale = ldexp(1.0, e), inv = ldexp(1.0, -e);
    if (g->n_chunk > 0) {
        const size_t lds = ((size_t)g->max_rows + g->n_cam) * 8;
        const int grid = g->n_chunk < 2048 ? g->n_chunk : 2048;
        if (val_is_f64)
            hipLaunchKernelGGL(edge_sums_kernel<double>, dim3(grid), dim3(256), lds, st, *g, (const double*)val, scale, inv,
                               row_sum, (u64*)cam_ws);
        else
            hipLaunchKernelGGL(edge_sums_kernel<float>, dim3(grid), dim3(256), lds, st, *g, (const float*)val, scale, inv,
                               row_sum, (u64*)cam_ws);
    }
    hipLaunchKernelGGL(fix_to_double_kernel, dim3((g->n_cam + 255) / 256), dim3(256), 0, st, g->n_cam, (const u64*)cam_ws, inv, cam_sum);
    LAUNCH_CHECK("vican_edge_sums");
    return VICAN_OK;
}

// rnorm[t] = sum_c |M_ct|_F ; fx[5] = max |M_ct|_F ; fx[6] = max_t rnorm[t]
// Raise a bound (fx[4..6]) to the workgroup's maximum with ONE atomic: same-address device-scope atomics are serialised at the memory
// side (~10 ns each; measured: 1600 per-wavefront atomics = most of a 20 us kernel at T = 100000).
__device__ __forceinline__ void wg_raise_bound(double om, double* target) {
    __shared__ double sh_max[16];
    __syncthreads();                        // callable twice in a row
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) om = fmax(om, __shfl_down(om, o, 64));
    if ((threadIdx.x & 63) == 0) sh_max[threadIdx.x >> 6] = om;
    __syncthreads();
    if (threadIdx.x == 0) {
        double m = 0.0;
        for (int i = 0; i < (int)(blockDim.x + 63) / 64; ++i) m = fmax(m, sh_max[i]);
        if (m > 0.0) atomic_max_pos(target, m);
    }
}

template <typename S>
__global__ void block_norms_kernel(vican_graph_t g, double* __restrict__ rnorm, double* __restrict__ fx) {
    extern __shared__ double lds[];
    const int k = blockIdx.x;
    const int r0 = g.chunk_row0[k], nrows = g.chunk_row0[k + 1] - r0;
    for (int r = threadIdx.x; r < nrows; r += blockDim.x) lds[r] = 0.0;
    __syncthreads();
    const S* blk = (const S*)g.blk;
    double amax = 0.0;
    for (int s = threadIdx.x; s < g.slots; s += blockDim.x) {
        const uint32_t id = g.idx[(size_t)k * g.slots + s];
        if (id == VICAN_PAD_SLOT) continue;
        double q = 0.0;
#pragma unroll
        for (int p = 0; p < 9; ++p) { const double v = (double)blk[((size_t)k * 9 + p) * g.slots + s]; q += v * v; }
        q = sqrt(q);
        amax = fmax(amax, q);
        lds_add(&lds[id >> 16], q);
    }
    __syncthreads();
    double rmax = 0.0;
    for (int r = threadIdx.x; r < nrows; r += blockDim.x) { rnorm[r0 + r] = lds[r]; rmax = fmax(rmax, lds[r]); }
    wg_raise_bound(amax, &fx[5]);
    wg_raise_bound(rmax, &fx[6]);
}

extern "C" int vican_block_norms(const vican_graph_t* g, double* rnorm, double* fx, void* stream) {
    if (int rc = vican_check_graph(g, "vican_block_norms")) return rc;
    if (!rnorm || !fx || !g->blk) return set_err(VICAN_ERR_ARG, "vican_block_norms: null pointer");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(fx, 0, VICAN_FX_DOUBLES * sizeof(double), st) != hipSuccess)
        return set_err(VICAN_ERR_LAUNCH, "vican_block_norms: memset failed");
    if (g->n_chunk == 0) return VICAN_OK;
    const size_t lds = (size_t)g->max_rows * 8;
    if (g->storage == VICAN_STORE_F32)
        hipLaunchKernelGGL(block_norms_kernel<float>, dim3(g->n_chunk), dim3(256), lds, st, *g, rnorm, fx);
    else
        hipLaunchKernelGGL(block_norms_kernel<double>, dim3(g->n_chunk), dim3(256), lds, st, *g, rnorm, fx);
    LAUNCH_CHECK("vican_block_norms");
    return VICAN_OK;
}

// Initial duals from the stored row sums d_t (bipgo.py:271-276): lamT_inv[t] = I/d_t, and the
// fixed-point bound omega = max_t |lamT_inv[t]|_F * rnorm[t].
__global__ void init_duals_kernel(int n_time, const double* __restrict__ d, const double* __restrict__ rnorm,
                                  double* __restrict__ lamT_inv, double* __restrict__ fx) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    double om = 0.0;
    if (t < n_time && d[t] > 0.0) om = 1.7320508075688772 / d[t] * rnorm[t];
    // contiguous stores: element i of the [T][9] array belongs to row i / 9 (diagonal entries i % 9 in {0, 4, 8})
    const int r0 = blockIdx.x * blockDim.x, nloc = min((int)blockDim.x, n_time - r0) * 9;       // 32-bit index arithmetic
    for (int i = threadIdx.x; i < nloc; i += blockDim.x) {
        const int r = i / 9, q = i - 9 * r;
        lamT_inv[(size_t)r0 * 9 + i] = (q == 0 || q == 4 || q == 8) ? 1.0 / d[r0 + r] : 0.0;
    }
    wg_raise_bound(om, &fx[4]);
}

extern "C" int vican_init_duals(int32_t n_time, const double* row_sum_a, const double* rnorm, double* lamT_inv,
                                double* fx, void* stream) {
    if (n_time < 0 || !row_sum_a || !rnorm || !lamT_inv || !fx) return set_err(VICAN_ERR_ARG, "vican_init_duals: bad argument");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(fx + 4, 0, sizeof(double), st) != hipSuccess) return set_err(VICAN_ERR_LAUNCH, "vican_init_duals: memset failed");
    if (n_time == 0) return VICAN_OK;
    hipLaunchKernelGGL(init_duals_kernel, dim3((n_time + 255) / 256), dim3(256), 0, st, n_time, row_sum_a, rnorm, lamT_inv, fx);
    LAUNCH_CHECK("vican_init_duals");
    return VICAN_OK;
}

// omega for caller-supplied duals: fx[4] = max_t |lamT_inv[t]|_F * rnorm[t]
__global__ void duals_bound_kernel(int n_time, const double* __restrict__ lamT_inv, const double* __restrict__ rnorm,
                                   double* __restrict__ fx) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    double om = 0.0;
    if (t < n_time) {
        double q = 0.0;
        for (int i = 0; i < 9; ++i) { const double v = lamT_inv[(size_t)t * 9 + i]; q += v * v; }
        om = sqrt(q) * rnorm[t];
        if (!(om >= 0.0) || om > 1e300) om = 0.0;          // rows without edges carry inf/nan duals
    }
    wg_raise_bound(om, &fx[4]);
}
extern "C" int vican_duals_bound(int32_t n_time, const double* lamT_inv, const double* rnorm, double* fx, void* stream) {
    if (n_time < 0 || !lamT_inv || !rnorm || !fx) return set_err(VICAN_ERR_ARG, "vican_duals_bound: bad argument");
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(fx + 4, 0, sizeof(double), st) != hipSuccess) return set_err(VICAN_ERR_LAUNCH, "vican_duals_bound: memset failed");
    if (n_time == 0) return VICAN_OK;
    hipLaunchKernelGGL(duals_bound_kernel, dim3((n_time + 255) / 256), dim3(256), 0, st, n_time, lamT_inv, rnorm, fx);
    LAUNCH_CHECK("vican_duals_bound");
    return VICAN_OK;
}

__global__ void scaled_identity_kernel(int n, const double* __restrict__ sc, double* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double s = sc[i];
    double* o = out + (size_t)i * 9;
    o[0] = s; o[1] = 0; o[2] = 0; o[3] = 0; o[4] = s; o[5] = 0; o[6] = 0; o[7] = 0; o[8] = s;
}
extern "C" int vican_scaled_identity(int32_t n, const double* scale, double* out, void* stream) {
    if (n < 0 || !scale || !out) return set_err(VICAN_ERR_ARG, "vican_scaled_identity: bad argument");
    if (n == 0) return VICAN_OK;
    hipLaunchKernelGGL(scaled_identity_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, scale, out);
    LAUNCH_CHECK("vican_scaled_identity");
    return VICAN_OK;
}

// Fixed-point scales (powers of two) from the recorded bounds:
//   y contributions  |M^T x|      <= amax * xb          row totals <= rmax * xb
//   z contributions  |M w|        <= amax * omega * xb  per-accumulator totals <= n_add * that
// A single contribution must stay below 2^51 (magic-number conversion), a total below 2^62.
__device__ __forceinline__ void fx_finish_one(double* fx, double x_bound, double n_add, int bits, int tot) {
    const double amax = fmax(fx[5], 1e-300), rmax = fmax(fx[6], 1e-300), om = fmax(fx[4], 1e-300);
    const double cy = amax * x_bound, ty = rmax * x_bound;
    const double cz = amax * om * x_bound, tz = cz * n_add;
    // a lane pre-sums up to 4 contributions before converting: 4 * 2^47 < 2^51 (magic-number conversion)
    int ey = min(bits - (int)ceil(log2(cy)), tot - (int)ceil(log2(ty)));
    int ez = min(bits - (int)ceil(log2(cz)), tot - (int)ceil(log2(tz)));
    ey = max(min(ey, 100), -100); ez = max(min(ez, 100), -100);     // pre-scaled f32 tables stay finite
    fx[0] = ldexp(1.0, ey); fx[1] = ldexp(1.0, -ey);
    fx[2] = ldexp(1.0, ez); fx[3] = ldexp(1.0, -ez);
    fx[7] = 1.0; fx[8] = x_bound;
    // the same for a phase-3 operand bounded by x_bound itself (omega = 1): MODE 3 sweeps (vican_dual_update_op)
    const double cz1 = amax * x_bound, tz1 = cz1 * n_add;
    int e1 = min(bits - (int)ceil(log2(cz1)), tot - (int)ceil(log2(tz1)));
    e1 = max(min(e1, 100), -100);
    fx[9] = ldexp(1.0, e1); fx[11] = ldexp(1.0, -e1);
}
__global__ void fx_finish_kernel(const int32_t* __restrict__ gate, double* fx, double x_bound, double n_add, int bits, int tot) {
    GATE_RETURN(gate);
    if (threadIdx.x || blockIdx.x) return;
    fx_finish_one(fx, x_bound, n_add, bits, tot);
}
// the scale buffers of several graphs over the SAME rows (camera tiles) in one launch: omega (fx[4]) is taken from the first
struct FxList { double* fx[64]; double n_add[64]; };
__global__ void fx_finish_multi_kernel(const int32_t* __restrict__ gate, FxList L, int n, double x_bound, int bits, int tot) {
    GATE_RETURN(gate);
    const int k = threadIdx.x;
    if (blockIdx.x || k >= n) return;
    if (k > 0) L.fx[k][4] = L.fx[0][4];
    fx_finish_one(L.fx[k], x_bound, L.n_add[k], bits, tot);
}
extern "C" int vican_fx_finish_multi(double* const* fx, const double* n_add, int32_t n, double x_bound, int32_t storage, void* stream) {
    if (!fx || !n_add || n <= 0 || n > 64 || !(x_bound > 0)) return set_err(VICAN_ERR_ARG, "vican_fx_finish_multi: bad argument");
    FxList L;
    for (int k = 0; k < n; ++k) {
        if (!fx[k] || !(n_add[k] >= 1)) return set_err(VICAN_ERR_ARG, "vican_fx_finish_multi: bad argument");
        L.fx[k] = fx[k]; L.n_add[k] = n_add[k];
    }
    const int bits = 47, tot = storage == VICAN_STORE_F32 ? 46 : 61;
    hipLaunchKernelGGL(fx_finish_multi_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, g_vican_gate, L, (int)n, x_bound, bits, tot);
    LAUNCH_CHECK("vican_fx_finish_multi");
    return VICAN_OK;
}
extern "C" int vican_fx_finish(double* fx, double x_bound, double n_add, int32_t storage, void* stream) {
    if (!fx || !(x_bound > 0) || !(n_add >= 1)) return set_err(VICAN_ERR_ARG, "vican_fx_finish: bad argument");
    // f32 blocks: the accumulators hold raw magic-biased bit patterns and totals are recovered from the low
    // 48 bits (fix_of<float>), so a total must stay below 2^46; f64 blocks: true 64-bit totals
    const int bits = 47, tot = storage == VICAN_STORE_F32 ? 46 : 61;
    hipLaunchKernelGGL(fx_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, g_vican_gate, fx, x_bound, n_add, bits, tot);
    LAUNCH_CHECK("vican_fx_finish");
    return VICAN_OK;
}

#endif  // !VICAN_SWEEP_PART

#if !defined(VICAN_SWEEP_SPLIT) || defined(VICAN_SWEEP_PART)
// ---------------------------------------------------------------------------
// THE HOT KERNEL
// ---------------------------------------------------------------------------
#include "vican_sweep_common.h"

// MODE 0: zpart[wg] (fixed point) = sum M * (lamT_inv * (sum M^T x))      (operator P x)
// MODE 1: per row SVD of (sum M^T x) -> Rt, lamT_inv, omega bound         (dual update)
// MODE 2: y_row = sum M^T x_cam -> lamT_out,  zpart[wg] = sum M * xt_row    (both halves of the symmetric
//         (C+T)-node operator R~ [x_cam; x_time] of the non-eliminated solver in ONE pass over the blocks;
//         `lamT_inv` carries x_time [T][9])
// MODE 3: MODE 1 (Z_t = sum M^T R_c -> lamT_out, for dual_svd_kernel) AND zpart[wg] = sum M * polar(Z_t) in the same
//         pass: with the new duals Lambda_t = U S^-1 U^T (Z_t = U S V^T) the product Lambda_t Z_t is the polar
//         factor U V^T, so this IS the first operator application P_new R_c of the next eigen-solve (up to the 3x3
//         normalisation of the start block, applied afterwards) - one pass over the blocks instead of two
// Arithmetic in the storage type S (f32 products for f32 blocks), accumulation in 64-bit fixed point.
//
// Per chunk:  [phase 3 of the previous chunk | phase 1]  barrier  [phase 2: one wavefront per
// row, no workgroup barrier inside]  barrier  ...   - two barriers per chunk; the register
// set of the next chunk is loaded while the current one is processed (ping-pong, no copies).
template <typename S, int BLOCK, int MODE, int CP, bool ROWPAR>
__global__ __launch_bounds__(BLOCK) void block_sweep_kernel(const int32_t* __restrict__ gate, vican_graph_t g,
                                                            const double* __restrict__ lamT_inv,
                                                            const double* __restrict__ x,
                                                            u64* __restrict__ zpart,
                                                            double* __restrict__ Rt_out,
                                                            double* __restrict__ lamT_out,
                                                            const double* __restrict__ rnorm,
                                                            double* __restrict__ fx) {
    GATE_RETURN(gate);
    constexpr int EPL = Vec<S>::N;
    constexpr int NWAVE = BLOCK / 64;
    extern __shared__ __align__(16) unsigned char lds_raw[];
    constexpr bool HAS_Z = (MODE == 0 || MODE == 2 || MODE == 3);            // camera-side accumulators + phase 3
    const int C = g.n_cam, nx = 9 * CP, ncopy = g.n_copy, cmask = ncopy - 1;
    // 8-byte arrays first, then the storage-type tables
    u64* zs = (u64*)lds_raw;                                   // [9][CP] planes (MODE 0)
    double* ysum = (double*)(zs + (HAS_Z ? nx : 0));       // [max_rows][9] (MODE 1)
    u64* ys = (u64*)(ysum + 9 * g.max_rows);                   // [max_rows*9][ncopy]
    S* xs = (S*)(ys + (size_t)9 * g.max_rows * ncopy);         // [9][CP] planes
    S* wv = xs + nx;                                           // [max_rows][9] (MODE 0; pre-scaled likewise)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lane_copy = tid & cmask;
    const uint32_t pad_cam = (uint32_t)((tid & 31) < g.n_cam ? (tid & 31) : 0);   // padding slots: a valid camera, zero block
    const int part_n = ncopy < 4 ? ncopy : 4, per = ncopy / part_n;   // phase 2: part_n lanes per accumulator

    // Chunks are handed out dynamically: DEPTH+1 per workgroup up front (chunk = workgroup + i * grid), then one
    // ticket per processed chunk from a device counter (fx[10]), fetched one body ahead of the prefetch that
    // needs it.  Measured with static ranges: workgroups finished between 192 and 212 us of a 214 us launch
    // (32 vs 33 chunks each, and CU-to-CU speed differences) - with tickets all finish within one chunk time.
    // Sums are exact integers, so WHICH workgroup accumulates a chunk cannot change the result; the
    // per-workgroup cap keeps the number of adds into one accumulator within the bound fx_finish assumed.
    constexpr int DEPTH = (BLOCK <= 512) ? 2 : 1;
    __shared__ int s_knext;
    unsigned int* sched = (unsigned int*)(fx + 10);            // [0] next ticket, [1] workgroups finished
    const int nchunk = g.n_chunk, nwg = (int)gridDim.x;
    const int cap = g.wg_chunk_cap > DEPTH ? g.wg_chunk_cap : 0x7fffffff;
    int kc = (int)blockIdx.x, kn1 = kc + nwg, kn2 = kc + 2 * nwg, taken = DEPTH + 1;
    ChunkRegs<S, EPL> ra, rb, rc;

    // The scales in fx assume |x_c|_F <= x_bound (sqrt 3).  Krylov vectors are far smaller
    // (~sqrt(3/C)), so every workgroup measures max_c |x_c|_F of THIS input (all find the same
    // value) and shifts both scales up by the power of two that still keeps it below the bound:
    // 4-5 more bits of fixed-point resolution for free.  Workgroup 0 records 2^-shift in fx[7]
    // for vican_slab_reduce_fx.  x is read once (registers), measured, then staged as planes.
    constexpr int XC = (CP + BLOCK - 1) / BLOCK;                // cameras per thread
    double xv[XC][9];
#pragma unroll
    for (int m = 0; m < XC; ++m) {
        const int c = tid + m * BLOCK;
#pragma unroll
        for (int i = 0; i < 9; ++i) xv[m][i] = c < C ? x[(size_t)c * 9 + i] : 0.0;
    }
    const double fx0 = fx[0], fx1 = fx[1], fx2 = fx[2], fx8 = fx[8], fx9 = (MODE == 3) ? fx[9] : 0.0;
    // The first chunk(s) are requested right behind the x loads (vector-memory results retire in issue
    // order, so x - an L2 hit - must be the older request): their HBM latency overlaps the table staging.
    __builtin_amdgcn_sched_barrier(0);
    if (kc < nchunk) load_chunk<S, EPL>(ra, g, kc, tid);
    if (DEPTH == 2 && kn1 < nchunk) load_chunk<S, EPL>(rb, g, kn1, tid);
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
    if (lane == 0) ysum[wave] = xm2;                     // scratch: ysum is not used before phase 2
    // tables staged BEFORE the (single) prologue barrier: they do not depend on the measured norm
    // (the wavefronts of a 768-thread workgroup start up to ~3 us apart: every barrier here costs that skew)
#pragma unroll
    for (int m = 0; m < XC; ++m) {
        const int c = tid + m * BLOCK;
        if (c < C) {
#pragma unroll
            for (int i = 0; i < 9; ++i) xs[i * CP + c] = pre_scale<S>(xv[m][i], 1.0);
        }
    }
    if (HAS_Z) for (int i = tid; i < nx; i += BLOCK) zs[i] = 0ull;
    for (int i = tid; i < 9 * g.max_rows * ncopy; i += BLOCK) ys[i] = 0ull;
    __syncthreads();

    xm2 = 0.0;
    for (int i = 0; i < NWAVE; ++i) xm2 = fmax(xm2, ysum[i]);
    int shift = 0;
    // floor(log2(x_bound / sqrt(xm2))) = floor(log2(x_bound^2 / xm2) / 2): one division + exponent extraction
    if (xm2 > 0.0) { const double r2 = fx8 * fx8 / xm2; shift = r2 >= 1.0 ? (ilogb(r2) >> 1) : 0; }     // fx[8] = x_bound
    shift = shift < 0 ? 0 : (shift > 40 ? 40 : shift);
    if (MODE == 2 || MODE == 3) shift = 0;      // phase-3 operand not measured here (x_time / polar factors, |.|_F = x_bound)
    const double up = ldexp(1.0, shift);
    const double y_scale = fx0 * up, y_inv = fx1 / up, z_scale = (MODE == 3) ? fx9 : fx2 * up;     // fx[9]: scale for omega = 1
    if ((MODE == 0 || MODE == 2) && blockIdx.x == 0 && tid == 0) fx[7] = 1.0 / up;
    if ((MODE == 1 || MODE == 3) && blockIdx.x == 0 && tid == 0) fx[4] = 0.0;       // omega bound: raised by dual_svd_kernel afterwards

    // Register ring: DEPTH chunks in flight ahead of the one being processed.  512-thread workgroups
    // have 256 VGPRs per lane and keep two chunks in flight (the memory system then always has work
    // from this CU); 768/1024-thread workgroups only have room for one.

    // body: process chunk k (registers `cur`), prefetch chunk kpref into `nxt`, draw the ticket for the chunk
    // the NEXT body will prefetch; returns that chunk index (>= nchunk: nothing left / cap reached)
    auto body = [&](ChunkRegs<S, EPL>& cur, ChunkRegs<S, EPL>& nxt, const int k, const int kpref) -> int {
        unsigned int ticket = 0;
        if (tid == 0 && taken < cap) ticket = atomicAdd(&sched[0], 1u);     // oldest vector-memory op of this body
        const int r0 = g.chunk_row0[k];
        const int nrows = g.chunk_row0[k + 1] - r0;
        // duals of the row this wavefront will fold in phase 2 (lane -> accumulator o = lane / part_n).
        // Issued BEFORE the chunk prefetch: vector-memory results retire in issue order, so a wait for
        // these three doubles placed after the prefetch would also wait for the whole next chunk.
        const int o = lane / part_n, oa = o / 3, ob = o - 3 * oa;
        double L0 = 0, L1 = 0, L2 = 0;
        if (MODE == 0 && wave < nrows && o < 9) {
            const double* L = lamT_inv + (size_t)(r0 + wave) * 9 + oa * 3;
            L0 = L[0]; L1 = L[1]; L2 = L[2];
        }
        if (MODE == 2 && wave < nrows && o < 9) L0 = lamT_inv[(size_t)(r0 + wave) * 9 + o];     // x_time entry o of the row
        if (ROWPAR && MODE == 0 && tid < nrows * 3) {       // many short rows: this thread's first (row, dual-block row) item
            const double* L = lamT_inv + (size_t)r0 * 9 + (size_t)tid * 3;
            L0 = L[0]; L1 = L[1]; L2 = L[2];
        }
        __builtin_amdgcn_sched_barrier(0);                              // keep these loads ahead of the prefetch
        if (kpref < nchunk) load_chunk<S, EPL>(nxt, g, kpref, tid);      // prefetch: lands during this/next chunk

        // ---- phase 1: y_row += M^T x_cam ; same-row edges of a lane pre-summed in registers,
        //      then ONE striped fixed-point atomic group per (lane,row).  Padding slots carry zero
        //      blocks and are processed like edges of (row 0, camera lane%32): no branches; the
        //      x gather of edge j+1 is in flight while edge j is multiplied.
        // inter-chunk prologue: descriptors, dual loads, prefetch issue
        uint32_t cam[EPL], row[EPL];
#pragma unroll
        for (int j = 0; j < EPL; ++j) {
            const bool pad = cur.id[j] == VICAN_PAD_SLOT;
            cam[j] = pad ? pad_cam : (cur.id[j] & 0xFFFFu);
            row[j] = pad ? 0u : (cur.id[j] >> 16);
        }
        {
            S acc[9], xc[9], xn[9];
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
                const bool last = (j == EPL - 1) || row[j + 1 < EPL ? j + 1 : j] != row[j];
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
        }
        // phase 1 (incl. waiting for this chunk's loads)
        if (tid == 0) s_knext = taken < cap ? (DEPTH + 1) * nwg + (int)ticket : 0x7fffffff;
        __syncthreads();
        // read between the two barriers of this body (rewritten after the second); wave-uniform -> scalar register,
        // so that the chunk addresses derived from it are scalar arithmetic
        const int knew = __builtin_amdgcn_readfirstlane(s_knext);
        // barrier A

        // ---- phase 2: one wavefront per row: fold the striped copies (exact integer sum, copy index
        //      rotated by the accumulator index => distinct banks), re-zero them, then
        //      w = lamT_inv * y (or the SVD) with wave shuffles only - no workgroup barrier inside
        if constexpr (ROWPAR) {
            // many short rows (sparse graphs; chosen at launch when max_rows > 2 wavefronts' worth): a wavefront per
            // row would run nrows / NWAVE rounds back to back; instead one thread per (row, accumulator):
            // fold -> barrier -> w = lamT_inv * y, all rows at once
            for (int i = tid; i < nrows * 9; i += BLOCK) {
                const int oo = i % 9;
                long long s = 0;
                for (int c = 0; c < ncopy; ++c) {
                    const int a = i * ncopy + ((c + oo) & cmask);
                    s += (long long)ys[a];
                    ys[a] = 0ull;
                }
                ysum[i] = (double)fix_total<S>(s) * y_inv;
            }
            // MODE 0: one thread per (row, row of its dual block): 24 contiguous bytes of lamT_inv per thread, three
            // outputs - a third of the loop trips (and exposed L2 latencies) of one thread per output; the first trip's
            // dual entries are requested before the barrier
            // (first trip: requested at the top of the body, ahead of the chunk prefetch; second trip: before the barrier)
            double l0 = L0, l1 = L1, l2 = L2, m0 = 0.0, m1 = 0.0, m2 = 0.0;
            if (MODE == 0 && tid + BLOCK < nrows * 3) {
                const double* L = lamT_inv + (size_t)r0 * 9 + (size_t)(tid + BLOCK) * 3;
                m0 = L[0]; m1 = L[1]; m2 = L[2];
            }
            __syncthreads();
            if (MODE == 0) {
                for (int i = tid; i < nrows * 3; i += BLOCK) {
                    if (i == tid + BLOCK) { l0 = m0; l1 = m1; l2 = m2; }
                    else if (i != tid) {
                        const double* L = lamT_inv + (size_t)r0 * 9 + (size_t)i * 3;
                        l0 = L[0]; l1 = L[1]; l2 = L[2];
                    }
                    const int r = i / 3;
                    const double* yr = ysum + r * 9;
#pragma unroll
                    for (int b3 = 0; b3 < 3; ++b3)
                        wv[i * 3 + b3] = pre_scale<S>(dot3<double>(l0, yr[b3], l1, yr[3 + b3], l2, yr[6 + b3]), z_scale);
                }
            } else {
                for (int i = tid; i < nrows * 9; i += BLOCK) {
                    lamT_out[(size_t)r0 * 9 + i] = ysum[i];
                    if (MODE == 2) wv[i] = pre_scale<S>(lamT_inv[(size_t)r0 * 9 + i], z_scale);
                }
            }
            if (MODE == 3) {                            // one thread per row: polar factor of Z_t -> phase-3 operand
                for (int r = tid; r < nrows; r += BLOCK) {
                    double R[9];
                    polar_newton3(ysum + r * 9, R);
#pragma unroll
                    for (int q = 0; q < 9; ++q) wv[r * 9 + q] = pre_scale<S>(R[q], z_scale);
                }
            }
        } else
        for (int r = wave; r < nrows; r += NWAVE) {
            long long s = 0;
            if (o < 9) {
                const int part = lane - o * part_n;
                for (int c = 0; c < per; ++c) {
                    const int a = (r * 9 + o) * ncopy + ((part * per + c + o) & cmask);
                    s += (long long)ys[a];
                    ys[a] = 0ull;
                }
            }
            if (part_n > 1) s += __shfl_xor(s, 1, 64);
            if (part_n > 2) s += __shfl_xor(s, 2, 64);
            s = fix_total<S>(s);
            const double y = (double)s * y_inv;                  // valid where lane == o * part_n
            if (MODE == 0) {
                if (r != wave && o < 9) {
                    const double* L = lamT_inv + (size_t)(r0 + r) * 9 + oa * 3;
                    L0 = L[0]; L1 = L[1]; L2 = L[2];
                }
                const double y0 = __shfl(y, (0 + ob) * part_n, 64), y1 = __shfl(y, (3 + ob) * part_n, 64),
                             y2 = __shfl(y, (6 + ob) * part_n, 64);
                if (o < 9 && lane == o * part_n) wv[r * 9 + o] = pre_scale<S>(dot3<double>(L0, y0, L1, y1, L2, y2), z_scale);
            } else {
                // Z_t goes to global memory; the per-row SVDs run afterwards in a fully parallel
                // kernel (one thread per row) instead of serialising this streaming sweep
                if (o < 9 && lane == o * part_n) lamT_out[(size_t)(r0 + r) * 9 + o] = y;
                if (MODE == 3 && o < 9 && lane == o * part_n) ysum[r * 9 + o] = y;
                if (MODE == 2) {
                    if (r != wave && o < 9) L0 = lamT_inv[(size_t)(r0 + r) * 9 + o];
                    if (o < 9 && lane == o * part_n) wv[r * 9 + o] = pre_scale<S>(L0, z_scale);
                }
            }
        }
        if (MODE == 3 && !ROWPAR) {                     // rows of the chunk side by side in one wavefront: polar factors
            __syncthreads();
            if (tid < nrows) {
                double R[9];
                polar_newton3(ysum + tid * 9, R);
#pragma unroll
                for (int q = 0; q < 9; ++q) wv[tid * 9 + q] = pre_scale<S>(R[q], z_scale);
            }
        }
        // all rows folded and re-zeroed before anyone starts the next phase 1 / reads w
        // phase 2
        __syncthreads();
        // barrier B
        if (HAS_Z) {
            // ---- phase 3: z_cam += M w_row   (blocks still in registers: read from HBM once)
            S w[9];
            uint32_t prow = 0xFFFFFFFFu;
#pragma unroll
            for (int j = 0; j < EPL; ++j) {
                const bool pad = cur.id[j] == VICAN_PAD_SLOT;
                const uint32_t camj = pad ? pad_cam : (cur.id[j] & 0xFFFFu);
                const uint32_t rowj = pad ? 0u : (cur.id[j] >> 16);
                if (rowj != prow) {
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
        // phase 3
        // (the barrier after the next chunk's phase 1 separates this chunk's phase 3 reads of wv
        //  from the next phase 2 writes; ys is already re-zeroed for the next phase 1)
        return knew;
    };

#define VICAN_ADVANCE1(knew_) do { const int kn_ = (knew_); kc = kn1; kn1 = kn_; taken += kn_ < nchunk ? 1 : 0; } while (0)
#define VICAN_ADVANCE2(knew_) do { const int kn_ = (knew_); kc = kn1; kn1 = kn2; kn2 = kn_; taken += kn_ < nchunk ? 1 : 0; } while (0)
    if (DEPTH == 1) {
#pragma unroll 1
        while (kc < nchunk) {
            VICAN_ADVANCE1(body(ra, rb, kc, kn1));
            if (kc >= nchunk) break;
            VICAN_ADVANCE1(body(rb, ra, kc, kn1));
        }
    } else {
#pragma unroll 1
        while (kc < nchunk) {                       // body(current, set to refill with the chunk two ahead)
            VICAN_ADVANCE2(body(ra, rc, kc, kn2));
            if (kc >= nchunk) break;
            VICAN_ADVANCE2(body(rb, ra, kc, kn2));
            if (kc >= nchunk) break;
            VICAN_ADVANCE2(body(rc, rb, kc, kn2));
        }
    }
#undef VICAN_ADVANCE1
#undef VICAN_ADVANCE2
    // the last workgroup to get here re-arms the ticket counter for the next launch (every ticket of every
    // workgroup has been drawn by then)
    // (device-scope atomics only - a device-scope fence would write back this XCD's whole L2, slabs included)
    if (tid == 0 && __hip_atomic_fetch_add(&sched[1], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == (unsigned)nwg - 1u) {
        __hip_atomic_store(&sched[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&sched[1], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (HAS_Z) {
        __syncthreads();
        u64* zp = zpart + (size_t)blockIdx.x * 9 * C;          // slab layout stays [9][C]
#pragma unroll
        for (int q = 0; q < 9; ++q)
            for (int c = tid; c < C; c += BLOCK) zp[q * C + c] = (u64)fix_total<S>((long long)zs[q * CP + c]);
    }
}

#endif  // hot kernel

#ifndef VICAN_SWEEP_PART
// Rt[t], lamT_inv[t] from Z_t (stored in lamT_inv by the MODE 1 sweep), in place; omega bound.
__global__ __launch_bounds__(256) void dual_svd_kernel(const int32_t* __restrict__ gate, int n_time, double* __restrict__ Rt,
                                                       double* __restrict__ lamT_inv, const double* __restrict__ rnorm,
                                                       double* __restrict__ fx) {
    GATE_RETURN(gate);
    // rows are 72-byte records: per-thread loads/stores of one record each run at ~0.3 TB/s (8-byte accesses 72 bytes
    // apart; measured 21 us for 100000 rows, arithmetic ~2 us) - the workgroup's 256 records go through LDS instead,
    // read and written as contiguous words
    __shared__ double sh[2][256 * 9];
    const int t0 = blockIdx.x * 256, t = t0 + threadIdx.x;
    const int nloc = min(256, n_time - t0) * 9;
    for (int i = threadIdx.x; i < nloc; i += 256) sh[0][i] = lamT_inv[(size_t)t0 * 9 + i];
    __syncthreads();
    double om = 0.0;
    if (t < n_time) {
        double Z[9], R[9], lam[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) Z[q] = sh[0][threadIdx.x * 9 + q];
        polar_dual3_fast(Z, R, lam, 2);
        double fro = 0.0;
#pragma unroll
        for (int q = 0; q < 9; ++q) { sh[1][threadIdx.x * 9 + q] = R[q]; sh[0][threadIdx.x * 9 + q] = lam[q]; fro += lam[q] * lam[q]; }
        om = sqrt(fro) * rnorm[t];
        if (!(om >= 0.0) || om > 1e300) om = 0.0;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nloc; i += 256) { Rt[(size_t)t0 * 9 + i] = sh[1][i]; lamT_inv[(size_t)t0 * 9 + i] = sh[0][i]; }
    wg_raise_bound(om, &fx[4]);
    if (blockIdx.x == 0 && threadIdx.x == 0) ((unsigned int*)(fx + 12))[VICAN_SCHED_REDO] = 0u;   // re-arm the fused sweep's redo word
}
#endif  // !VICAN_SWEEP_PART

#if !defined(VICAN_SWEEP_SPLIT) || defined(VICAN_SWEEP_PART)
template <typename S, int BLOCK, int MODE, int CP, bool ROWPAR>
static int launch_sweep2(const vican_graph_t* g, const double* lamT_inv, const double* x, u64* zpart, double* Rt,
                         double* lamT_out, const double* rnorm, double* fx, hipStream_t st) {
    const size_t lds = (size_t)vican_sweep_lds_bytes(g->n_cam, g->max_rows, g->storage, g->n_copy);
    auto kern = block_sweep_kernel<S, BLOCK, MODE, CP, ROWPAR>;
    static size_t configured = 0;       // per instantiation
    if (lds > configured) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return set_err(VICAN_ERR_LAUNCH, "%s: cannot raise dynamic LDS limit", "vican sweep");
        configured = lds;
    }
    VICAN_LAUNCH_SWEEP(kern, dim3(g->n_wg), dim3(BLOCK), lds, st, g_vican_gate, *g, lamT_inv, x, zpart, Rt, lamT_out, rnorm, fx);
    return 0;
}

template <typename S, int BLOCK, int MODE, int CP>
static int launch_sweep1(const vican_graph_t* g, const double* lamT_inv, const double* x, u64* zpart, double* Rt,
                        double* lamT_out, const double* rnorm, double* fx, hipStream_t st) {
    if (g->max_rows > 2 * (BLOCK / 64)) return launch_sweep2<S, BLOCK, MODE, CP, true>(g, lamT_inv, x, zpart, Rt, lamT_out, rnorm, fx, st);
    return launch_sweep2<S, BLOCK, MODE, CP, false>(g, lamT_inv, x, zpart, Rt, lamT_out, rnorm, fx, st);
}
template <typename S, int BLOCK, int MODE>
static int launch_sweep(const vican_graph_t* g, const double* lamT_inv, const double* x, u64* zpart, double* Rt,
                        double* lamT_out, const double* rnorm, double* fx, hipStream_t st) {
    if (g->n_cam <= 256) return launch_sweep1<S, BLOCK, MODE, 256>(g, lamT_inv, x, zpart, Rt, lamT_out, rnorm, fx, st);
    if (g->n_cam <= 512) return launch_sweep1<S, BLOCK, MODE, 512>(g, lamT_inv, x, zpart, Rt, lamT_out, rnorm, fx, st);
    return launch_sweep1<S, BLOCK, MODE, 1024>(g, lamT_inv, x, zpart, Rt, lamT_out, rnorm, fx, st);
}

template <int MODE>
static int dispatch_sweep(const vican_graph_t* g, const double* lamT_inv, const double* x, u64* zpart, double* Rt,
                          double* lamT_out, const double* rnorm, double* fx, void* stream) {
    hipStream_t st = (hipStream_t)stream;
#define SWEEP_ARGS g, lamT_inv, x, zpart, Rt, lamT_out, rnorm, fx, st
    if (g->storage == VICAN_STORE_F32) {
        if (g->block_threads == 1024) return launch_sweep<float, 1024, MODE>(SWEEP_ARGS);
        if (g->block_threads == 768) return launch_sweep<float, 768, MODE>(SWEEP_ARGS);
        if (g->block_threads == 512) return launch_sweep<float, 512, MODE>(SWEEP_ARGS);
        return launch_sweep<float, 256, MODE>(SWEEP_ARGS);
    }
    if (g->block_threads == 1024) return launch_sweep<double, 1024, MODE>(SWEEP_ARGS);
    if (g->block_threads == 768) return launch_sweep<double, 768, MODE>(SWEEP_ARGS);
    if (g->block_threads == 512) return launch_sweep<double, 512, MODE>(SWEEP_ARGS);
    return launch_sweep<double, 256, MODE>(SWEEP_ARGS);
#undef SWEEP_ARGS
}
#endif  // hot kernel launchers

#if defined(VICAN_SWEEP_PART)
#define SWEEP_CAT2(a, b) a##b
#define SWEEP_CAT(a, b) SWEEP_CAT2(a, b)
extern "C" __attribute__((visibility("hidden"))) int SWEEP_CAT(vican_sweep_part_, VICAN_SWEEP_PART)(SWEEP_PART_ARGS) {
    return dispatch_sweep<VICAN_SWEEP_PART>(g, lamT_inv, x, zpart, Rt, lamT_out, rnorm, fx, stream);
}
#else   // everything below: not in the kernel-only parts
#if defined(VICAN_SWEEP_SPLIT)
extern "C" {
__attribute__((visibility("hidden"))) int vican_sweep_part_0(SWEEP_PART_ARGS);
__attribute__((visibility("hidden"))) int vican_sweep_part_1(SWEEP_PART_ARGS);
__attribute__((visibility("hidden"))) int vican_sweep_part_2(SWEEP_PART_ARGS);
__attribute__((visibility("hidden"))) int vican_sweep_part_3(SWEEP_PART_ARGS);
}
template <int MODE>
static int dispatch_sweep(SWEEP_PART_ARGS) {
    if (MODE == 0) return vican_sweep_part_0(g, lamT_inv, x, zpart, Rt, lamT_out, rnorm, fx, stream);
    if (MODE == 1) return vican_sweep_part_1(g, lamT_inv, x, zpart, Rt, lamT_out, rnorm, fx, stream);
    if (MODE == 2) return vican_sweep_part_2(g, lamT_inv, x, zpart, Rt, lamT_out, rnorm, fx, stream);
    return vican_sweep_part_3(g, lamT_inv, x, zpart, Rt, lamT_out, rnorm, fx, stream);
}
#endif

// graphs in the wave layout (one wavefront per chunk): vican_wsweep.hip
extern "C" __attribute__((visibility("hidden"))) int vican_wsweep(int mode, const vican_graph_t* g, const double* lamT_inv, const double* x,
                                                                  u64* zpart, double* lamT_out, double* fx, void* stream);
template <int MODE>
static int sweep_any(SWEEP_PART_ARGS) {
    if (!g->blk) return set_err(VICAN_ERR_ARG, "%s: graph without block planes", "vican sweep");
    if (g->layout == VICAN_LAYOUT_WAVE) return vican_wsweep(MODE, g, lamT_inv, x, zpart, lamT_out, fx, stream);
    return dispatch_sweep<MODE>(g, lamT_inv, x, zpart, Rt, lamT_out, rnorm, fx, stream);
}

extern "C" int vican_block_op(const vican_graph_t* g, const double* lamT_inv, const double* x, void* zpart,
                              double* fx, void* stream) {
    if (int rc = vican_check_graph(g, "vican_block_op")) return rc;
    if (!lamT_inv || !x || !zpart || !fx) return set_err(VICAN_ERR_ARG, "vican_block_op: null pointer");
    if (int rc = sweep_any<0>(g, lamT_inv, x, (u64*)zpart, nullptr, nullptr, nullptr, fx, stream)) return rc;
    LAUNCH_CHECK("vican_block_op");
    return VICAN_OK;
}


// Both halves of R~ [x_cam; x_time] (non-eliminated solver): y_time[t] = sum_c M_ct^T x_cam[c] (exact fixed-point
// row sums -> f64), z_cam = slab-reduced sum_t M_ct x_time[t].  fx must hold the scales of vican_bip_scales.
extern "C" int vican_bip_apply(const vican_graph_t* g, const double* x_cam, const double* x_time, void* zpart,
                               double* fx, double* z_cam, double* y_time, void* stream) {
    if (int rc = vican_check_block_graph(g, "vican_bip_apply")) return rc;
    if (!g->blk) return set_err(VICAN_ERR_ARG, "vican_bip_apply: graph without block planes");
    if (!x_cam || !x_time || !zpart || !fx || !z_cam || !y_time) return set_err(VICAN_ERR_ARG, "vican_bip_apply: null pointer");
    if (g->n_chunk == 0) return set_err(VICAN_ERR_ARG, "vican_bip_apply: graph without edges");
    if (int rc = dispatch_sweep<2>(g, x_time, x_cam, (u64*)zpart, nullptr, y_time, nullptr, fx, stream)) return rc;
    LAUNCH_CHECK("vican_bip_apply");
    return vican_slab_reduce_fx(zpart, g->n_wg, g->n_cam, 9, 1.0, fx + 3, fx + 7, z_cam, stream);
}

// ---- camera-tiled operator (more cameras than one LDS table holds), one tile in the WAVE layout -----------------------------
// z = sum_t M_.t Lambda_t^-1 (sum_c M_ct^T x_c) needs the row sums over ALL cameras before any camera's output: per tile a rows
// pass (sweep MODE 1: y_time = the tile's share of Z_t), the caller sums the tiles' shares and applies Lambda_t^-1
// (vican_sum_apply3), then a camera pass per tile (sweep MODE 4: z_cam = sum_t M_ct w_t).  Both passes stream the tile's
// blocks through the wave-layout kernel (the round-3 path ran the two-sided block-layout sweep, MODE 2, twice with a zero
// operand on one side each time).  fx: the tile's scales for the CURRENT duals, bounds taken over all tiles' rows
// (vican_duals_bound with the global row norms + vican_fx_finish).
extern "C" int vican_tile_rows(const vican_graph_t* g, const double* x_cam, double* y_time, double* fx, void* stream) {
    if (int rc = vican_check_graph(g, "vican_tile_rows")) return rc;
    if (g->layout != VICAN_LAYOUT_WAVE) return set_err(VICAN_ERR_ARG, "vican_tile_rows: wave layout only (block-layout tiles: vican_bip_apply)");
    if (!x_cam || !y_time || !fx) return set_err(VICAN_ERR_ARG, "vican_tile_rows: null pointer");
    if (g->n_chunk == 0) return set_err(VICAN_ERR_ARG, "vican_tile_rows: graph without edges");
    if (int rc = sweep_any<1>(g, nullptr, x_cam, nullptr, nullptr, y_time, nullptr, fx, stream)) return rc;
    LAUNCH_CHECK("vican_tile_rows");
    return VICAN_OK;
}
extern "C" int vican_tile_cams(const vican_graph_t* g, const double* w_time, void* zpart, double* fx, double* z_cam, void* stream) {
    if (int rc = vican_check_graph(g, "vican_tile_cams")) return rc;
    if (g->layout != VICAN_LAYOUT_WAVE) return set_err(VICAN_ERR_ARG, "vican_tile_cams: wave layout only (block-layout tiles: vican_bip_apply)");
    if (!w_time || !zpart || !fx || !z_cam) return set_err(VICAN_ERR_ARG, "vican_tile_cams: null pointer");
    if (g->n_chunk == 0) return set_err(VICAN_ERR_ARG, "vican_tile_cams: graph without edges");
    if (int rc = vican_wsweep(4, g, w_time, nullptr, (u64*)zpart, nullptr, fx, stream)) return rc;
    LAUNCH_CHECK("vican_tile_cams");
    return vican_slab_reduce_fx(zpart, g->n_wg, g->n_cam, 9, 1.0, fx + 3, fx + 7, z_cam, stream);
}

__global__ void set_double_kernel(double* p, double v) { if (threadIdx.x == 0 && blockIdx.x == 0) *p = v; }
// Fixed-point scales for vican_bip_apply: the phase-3 operand is x_time itself (|x_t|_F <= x_bound), i.e. omega = 1.
extern "C" int vican_bip_scales(double* fx, double x_bound, double n_add, int32_t storage, void* stream) {
    if (!fx) return set_err(VICAN_ERR_ARG, "vican_bip_scales: null pointer");
    hipLaunchKernelGGL(set_double_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, fx + 4, 1.0);
    return vican_fx_finish(fx, x_bound, n_add, storage, stream);
}

extern "C" int vican_dual_update(const vican_graph_t* g, const double* rc_, double* Rt, double* lamT_inv,
                                 const double* rnorm, double* fx, void* stream) {
    if (int rc = vican_check_graph(g, "vican_dual_update")) return rc;
    if (!rc_ || !Rt || !lamT_inv || !rnorm || !fx) return set_err(VICAN_ERR_ARG, "vican_dual_update: null pointer");
    if (g->n_chunk == 0) {                       // a rank without rows: omega bound 0 (memsets cannot be gated; nothing to cancel)
        if (hipMemsetAsync(fx + 4, 0, sizeof(double), (hipStream_t)stream) != hipSuccess)
            return set_err(VICAN_ERR_LAUNCH, "vican_dual_update: memset failed");
        return VICAN_OK;
    }
    if (int rc = sweep_any<1>(g, nullptr, rc_, nullptr, Rt, lamT_inv, rnorm, fx, stream)) return rc;   // zeroes fx[4]
    hipLaunchKernelGGL(dual_svd_kernel, dim3((g->n_time + 255) / 256), dim3(256), 0, (hipStream_t)stream, g_vican_gate,
                       g->n_time, Rt, lamT_inv, rnorm, fx);
    LAUNCH_CHECK("vican_dual_update");
    return VICAN_OK;
}

// Dual update fused with the first operator application of the next eigen-solve (MODE 3): as vican_dual_update,
// and z_raw[3C][3] = this rank's partial of  sum_t M_ct polar(Z_t)  =  P_new R_c  for the NEW duals.
extern "C" int vican_dual_update_op(const vican_graph_t* g, const double* rc_, double* Rt, double* lamT_inv,
                                    const double* rnorm, double* fx, void* zpart, double* z_raw, void* stream) {
    if (int rc = vican_check_graph(g, "vican_dual_update_op")) return rc;
    if (!rc_ || !Rt || !lamT_inv || !rnorm || !fx || !zpart || !z_raw) return set_err(VICAN_ERR_ARG, "vican_dual_update_op: null pointer");
    if (g->n_chunk == 0) {                       // a rank without rows (memsets cannot be gated; z_raw is only read after a converged gate)
        if (hipMemsetAsync(fx + 4, 0, sizeof(double), (hipStream_t)stream) != hipSuccess ||
            hipMemsetAsync(z_raw, 0, sizeof(double) * 9 * g->n_cam, (hipStream_t)stream) != hipSuccess)
            return set_err(VICAN_ERR_LAUNCH, "vican_dual_update_op: memset failed");
        return VICAN_OK;
    }
    if (int rc = sweep_any<3>(g, nullptr, rc_, (u64*)zpart, Rt, lamT_inv, rnorm, fx, stream)) return rc;   // zeroes fx[4]
    LAUNCH_CHECK("vican_dual_update_op");
    if (int rc = vican_slab_reduce_fx(zpart, g->n_wg, g->n_cam, 9, 1.0, fx + 11, nullptr, z_raw, stream)) return rc;
    hipLaunchKernelGGL(dual_svd_kernel, dim3((g->n_time + 255) / 256), dim3(256), 0, (hipStream_t)stream, g_vican_gate,
                       g->n_time, Rt, lamT_inv, rnorm, fx);
    LAUNCH_CHECK("vican_dual_update_op");
    return VICAN_OK;
}

// ---------------------------------------------------------------------------
// slab reductions (fixed order; the fixed-point one is exact)
// ---------------------------------------------------------------------------
// 256 threads = 64 elements x 4 slab groups
__global__ __launch_bounds__(256) void slab_reduce_kernel(const double* __restrict__ part, int n_slab, long long n,
                                                          double* __restrict__ out) {
    __shared__ double sh[256];
    const int e = threadIdx.x & 63, grp = threadIdx.x >> 6;
    const long long i = (long long)blockIdx.x * 64 + e;
    double s = 0.0;
    if (i < n)
        for (int k = grp; k < n_slab; k += 4) s += part[(size_t)k * n + i];
    sh[threadIdx.x] = s;
    __syncthreads();
    if (grp == 0 && i < n) out[i] = (sh[e] + sh[64 + e]) + (sh[128 + e] + sh[192 + e]);
}
extern "C" int vican_slab_reduce(const double* part, int32_t n_slab, int64_t n, double* out, void* stream) {
    if (!part || !out || n_slab <= 0 || n < 0) return set_err(VICAN_ERR_ARG, "vican_slab_reduce: bad argument");
    if (n == 0) return VICAN_OK;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, (hipStream_t)stream, part,
                       n_slab, (long long)n, out);
    LAUNCH_CHECK("vican_slab_reduce");
    return VICAN_OK;
}

// slabs hold planes [ncomp][C]; the output is the row-major camera vector [C][ncomp].
// 1024 threads = 64 elements x 16 slab groups (256 slabs -> 16 loads per thread in flight).
// COLS columns per workgroup (1024 / COLS groups of lanes stride over the slabs)
template <int COLS>
__global__ __launch_bounds__(1024) void slab_reduce_fx_kernel(const int32_t* __restrict__ gate, const long long* __restrict__ part, int n_slab, long long n,
                                                              int ncomp, double scale, const double* __restrict__ pa,
                                                              const double* __restrict__ pb, double* __restrict__ out) {
    GATE_RETURN(gate);
    __shared__ long long sh[1024];
    constexpr int NG = 1024 / COLS;
    const int e = threadIdx.x % COLS, grp = threadIdx.x / COLS;
    const long long i = (long long)blockIdx.x * COLS + e;
    long long s = 0;                            // (integer sums: exact, any order)
    if (i < n)
        for (int k = grp; k < n_slab; k += NG) s += part[(size_t)k * n + i];
    sh[threadIdx.x] = s;
    __syncthreads();
    if (grp == 0 && i < n) {
        long long t = 0;
#pragma unroll
        for (int k = 0; k < NG; ++k) t += sh[k * COLS + e];
        const long long C = n / ncomp, q = i / C, cam = i % C;
        const double sc = scale * (pa ? *pa : 1.0) * (pb ? *pb : 1.0);
        out[cam * ncomp + q] = (double)t * sc;
    }
}
extern "C" int vican_slab_reduce_fx(const void* part, int32_t n_slab, int32_t n_cam, int32_t ncomp, double scale,
                                    const double* pa, const double* pb, double* out, void* stream) {
    if (!part || !out || n_slab <= 0 || n_cam <= 0 || ncomp <= 0) return set_err(VICAN_ERR_ARG, "vican_slab_reduce_fx: bad argument");
    const long long n = (long long)n_cam * ncomp;
#define SRF_LAUNCH(COLS_) hipLaunchKernelGGL(slab_reduce_fx_kernel<COLS_>, dim3((unsigned)((n + COLS_ - 1) / COLS_)), dim3(1024), 0, (hipStream_t)stream, \
                                             g_vican_gate, (const long long*)part, n_slab, n, ncomp, scale, pa, pb, out)
    // (64-column workgroups always: 32-column ones measured 9.0 against 8.8 us on the stress graph's 256 slabs x 9000 columns, and the
    //  CG fold 21.9 against 9.7 us with 16 columns - narrow pieces of slabs that lie far apart)
    SRF_LAUNCH(64);
#undef SRF_LAUNCH
    LAUNCH_CHECK("vican_slab_reduce_fx");
    return VICAN_OK;
}

// composite: operator sweep + slab fold as one host call (z = local P x, row-major [3C][3])
extern "C" int vican_block_op_z(const vican_graph_t* g, const double* lamT_inv, const double* x, void* zpart,
                                double* fx, double* z, void* stream) {
    int rc = vican_block_op(g, lamT_inv, x, zpart, fx, stream);
    if (rc < 0) return rc;
    return vican_slab_reduce_fx(zpart, g->n_wg, g->n_cam, 9, 1.0, fx + 3, fx + 7, z, stream);
}
#endif  // !VICAN_SWEEP_PART
