// vican_facade_tiles.hip - the four-call boundary (vican_facade.hip) on graphs with more cameras than the LDS-resident sweeps
// hold (C > 1024; the reference has no camera limit, bipgo.py:225-232): the plan cuts the edge set by camera range into tiles of
// equal width that share ONE chunking of the timestep rows (vican_plan_chunks_multi), packs every tile in the wave layout and
// runs the schedule of vican_amd/tiled.py (TiledGraph / TiledBackend) from C:
//   operator      one launch that reads every block once (vican_tiled_op_z); where that grid is not co-resident a rows pass
//                 per tile, w_t = Lambda_t^-1 (sum of the tiles' row partials), a camera pass per tile
//   dual update   rows pass per tile, sum in tile order, batched 3x3 SVDs (vican_polar_dual)
//   J^T b         per tile, rows summed in tile order
//   CG product    the tiles' sweeps in one launch (vican_cg_sweep_tiles, 2..4 tiles) or tile by tile, rows combined in tile order
// Everything camera-sided (Lanczos step, Ritz, gauge, polar) has no camera limit and stays in vican_facade.hip.
// The timestep rows live in an ORDER OF THE PLAN'S OWN where that packs the shared chunking tighter (vican_plan_rows_multi: 1.33 ->
// 1.03 slots per edge on 4 tiles x 62 edges per row); everything per row inside the plan is in that order, the calls translate
// their per-row arguments (deg_t, Rt, x_t) at the boundary.
//   LSQR          vican_solve_trans_lsqr (vican_facade.hip): every pass over the edges tile by tile, each tile its own edge vector
// Not here (the Python driver's): tiles in the block layout (a tile row of more than 64 * EPL edges, a tile without edges).
#include "vican_facade_impl.h"

namespace {

int g_tile_cams = 1024;         // cameras per tile (the LDS tables of the sweeps); vican_facade_set_tile_cams: tests force small tiles

#define CK(call) do { const int rc_ = (call); if (rc_ < 0) return rc_; } while (0)
#define HIPCK(call, what) do { if ((call) != hipSuccess) return ferr(VICAN_ERR_LAUNCH, "%s: %s failed", what, #call); } while (0)

// edges of every timestep row per camera tile (tiles = camera ranges of equal width): one wavefront per row, lane k counts tile k
__global__ void tile_count_kernel(int T, int nt, int tile, const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
                                  int32_t* __restrict__ cnt /*[nt][T]*/) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (t >= T) return;
    const int e0 = row_ptr[t], e1 = row_ptr[t + 1];
    int mine = 0;
    for (int e = e0; e < e1; e += 64) {
        const int k = e + lane < e1 ? col[e + lane] / tile : -1;
        for (int q = 0; q < nt; ++q) {
            const unsigned long long m = __ballot(k == q);
            if (lane == q) mine += __popcll(m);
        }
    }
    if (lane < nt) cnt[(size_t)lane * T + t] = mine;
}

// the edges of ONE tile (cameras [c0, c1)) as CSR arrays of their own, rows and the order inside a row kept: one wavefront per row
// (perm != NULL: row t of the tile = row perm[t] of the caller's arrays)
template <typename S>
__global__ void tile_gather_kernel(int T, int c0, int c1, const int32_t* __restrict__ perm, const int32_t* __restrict__ row_ptr, const int32_t* __restrict__ col,
                                   const S* __restrict__ blk, const S* __restrict__ a, const double* __restrict__ w,
                                   const double* __restrict__ u, const double* __restrict__ v, const int32_t* __restrict__ rp_k,
                                   int32_t* __restrict__ col_k, S* __restrict__ blk_k, S* __restrict__ a_k, double* __restrict__ w_k,
                                   double* __restrict__ u_k, double* __restrict__ v_k) {
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (t >= T) return;
    const int ts = perm ? perm[t] : t;
    const int e0 = row_ptr[ts], e1 = row_ptr[ts + 1];
    int base = rp_k[t];
    for (int e = e0; e < e1; e += 64) {
        const int i = e + lane;
        const int c = i < e1 ? col[i] : -1;
        const bool in = c >= c0 && c < c1;
        const unsigned long long m = __ballot(in);
        if (in) {
            const size_t d = (size_t)base + __popcll(m & ((1ull << lane) - 1ull));
            col_k[d] = c - c0;
            a_k[d] = a[i];
#pragma unroll
            for (int q = 0; q < 9; ++q) blk_k[d * 9 + q] = blk[(size_t)i * 9 + q];
            if (w) {
                w_k[d] = w[i];
#pragma unroll
                for (int q = 0; q < 3; ++q) { u_k[d * 3 + q] = u[(size_t)i * 3 + q]; v_k[d * 3 + q] = v[(size_t)i * 3 + q]; }
            }
        }
        base += __popcll(m);
    }
}

// dst[r] = src[perm[r]] (gather: the caller's order -> the plan's) or dst[perm[r]] = src[r] (scatter: back), rows of `width` doubles
__global__ void rows_permute_kernel(long long n, int width, const int32_t* __restrict__ perm, const double* __restrict__ src,
                                    double* __restrict__ dst, int scatter) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n * width; i += (long long)gridDim.x * blockDim.x) {
        const long long r = i / width, q = i - r * width, o = (long long)perm[r] * width + q;
        if (scatter) dst[o] = src[i]; else dst[i] = src[o];
    }
}

// vican_amd/layout.py _wave_params: what the LDS of a compute unit allows for a wave-layout graph
int wave_params(int n_cam, double deg_avg, long long n_edges, int storage, int ncu, int* slots_out, int* rows_out, int* n_copy_out, int* waves_out) {
    const int epl = storage == VICAN_STORE_F32 ? 4 : 2, slots = 64 * epl;
    const long long lim = vican_lds_limit_bytes();
    int rows = std::max(1, std::min(n_cam >= 1024 ? 63 : 64, (int)std::ceil(1.25 * slots / deg_avg) + 1));
    int n_copy = 1;
    while (n_copy < 8 && n_copy * epl < deg_avg) n_copy *= 2;
    int waves = 12;
    if (n_edges < 12LL * slots * ncu) waves = n_edges >= 8LL * slots * ncu ? 8 : 4;
    auto fits = [&](int r, int nc, int nw) { return vican_wsweep_lds_bytes(n_cam, r, storage, nc, nw) <= lim; };
    while (!fits(rows, n_copy, waves) && n_copy > 1) n_copy /= 2;
    if (!fits(rows, n_copy, waves)) {
        int r = rows;
        while (r > 1 && !fits(r, n_copy, waves)) --r;
        if (fits(r, n_copy, waves) && r * deg_avg >= 1.05 * slots) rows = r;
    }
    while (!fits(rows, n_copy, waves) && waves > 4) waves -= 4;
    while (!fits(rows, n_copy, waves) && rows > 1) --rows;
    if (!fits(rows, n_copy, waves)) return ferr(VICAN_ERR_CAPACITY, "vican_plan_create: camera tables (C=%d) do not fit in LDS", n_cam);
    *slots_out = slots; *rows_out = rows; *n_copy_out = n_copy; *waves_out = waves;
    return VICAN_OK;
}

}  // namespace

extern "C" int vican_facade_set_tile_cams(int32_t n) {
    if (n < 1 || n > 1024) return ferr(VICAN_ERR_ARG, "vican_facade_set_tile_cams: 1..1024");
    g_tile_cams = n;
    return VICAN_OK;
}

int vican_facade_tile_cams() { return g_tile_cams; }

// ---- planning: tiles, the shared chunking, every tile's descriptor (host; one small kernel for the per-tile row lengths) ----------
int vican_facade_tiles_layout(vican_plan* P, const int32_t* row_ptr, const int32_t* col, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    const int C = P->C, T = P->T, storage = P->storage, epl = P->epl, ncu = vican_facade::n_cu();
    // tiles of EQUAL width (4000 cameras: 4 x 1000, not 3 x 1024 + 928): a row's edges split evenly and the shared chunking fills
    // every tile's slots at the same pace (vican_amd/tiled.py)
    int nt = std::max(1, (C + g_tile_cams - 1) / g_tile_cams);
    const int tile = std::min(g_tile_cams, (((C + nt - 1) / nt) + 7) / 8 * 8);
    nt = (C + tile - 1) / tile;
    if (nt > 64) return ferr(VICAN_ERR_CAPACITY, "vican_plan_create: %d camera tiles (more than 64: the host driver vican_amd.tiled runs them)", nt);
    P->tile_width = tile;
    P->tiles.assign(nt, vican_tile_plan{});
    int32_t* cnt = nullptr;
    if (hipMalloc((void**)&cnt, (size_t)nt * T * 4) != hipSuccess) return ferr(VICAN_ERR_LAUNCH, "vican_plan_create: hipMalloc failed");
    hipLaunchKernelGGL(tile_count_kernel, dim3((T + 3) / 4), dim3(256), 0, s, T, nt, tile, row_ptr, col, cnt);
    std::vector<int32_t> h((size_t)nt * T);
    const bool ok = hipMemcpyAsync(h.data(), cnt, h.size() * 4, hipMemcpyDeviceToHost, s) == hipSuccess && hipStreamSynchronize(s) == hipSuccess;
    hipFree(cnt);
    if (!ok) return ferr(VICAN_ERR_LAUNCH, "vican_plan_create: cannot count the tiles' edges");
    int slots = 64 * epl, cap_rows = 64;
    for (int k = 0; k < nt; ++k) {
        vican_tile_plan& t = P->tiles[k];
        t.c0 = k * tile; t.c1 = std::min(C, (k + 1) * tile);
        t.rp.assign((size_t)T + 1, 0);
        for (int r = 0; r < T; ++r) t.rp[r + 1] = t.rp[r] + h[(size_t)k * T + r];
        t.E = t.rp[T];
        if (t.E == 0) return ferr(VICAN_ERR_CAPACITY, "vican_plan_create: camera tile %d (cameras %d..%d) has no edges: per-tile layouts are the host driver's (vican_amd.tiled)", k, t.c0, t.c1 - 1);
        int rows_k = 0;
        CK(wave_params(t.c1 - t.c0, std::max(1.0, (double)t.E / T), t.E, storage, ncu, &slots, &rows_k, &t.n_copy, &t.wg_waves));
        // (the fused launch runs 8 wavefronts per workgroup whatever the tile's own plan says: its LDS must fit too)
        while (rows_k > 1 && vican_tiled_op_lds_bytes(t.c1 - t.c0, rows_k, storage, t.n_copy) > vican_lds_limit_bytes()) --rows_k;
        cap_rows = std::min(cap_rows, rows_k);
    }
    std::vector<const int32_t*> ptrs(nt);
    for (int k = 0; k < nt; ++k) ptrs[k] = P->tiles[k].rp.data();
    P->t_chunks.assign((size_t)T + 2, 0);
    // rows in a better order for the shared chunking (tiled.py: a pool of 512 rows where few rows fill a chunk - the integer effect -
    // 128 up to 32 rows per chunk, none beyond: chunks of many short rows fill well in any order)
    long long e_max = 0;
    for (const vican_tile_plan& t : P->tiles) e_max = std::max(e_max, t.E);
    const double rows_est = std::max(1.0, slots / std::max(1.0, (double)e_max / T));
    const int window = rows_est <= 8 ? 512 : (rows_est <= 32 ? 128 : 1);
    P->t_perm.clear();
    int nchunk = -1;
    if (window > 1 && nt > 1) {
        std::vector<int32_t> perm((size_t)T), c0s((size_t)T + 2);
        const int n = vican_plan_rows_multi(T, nt, ptrs.data(), slots, cap_rows, window, perm.data(), c0s.data(), T + 2);
        bool identity = n >= 0;
        for (int r = 0; identity && r < T; ++r) identity = perm[r] == r;
        if (n >= 0 && identity) { nchunk = n; P->t_chunks = c0s; }
        else if (n >= 0) {
            // the tiles' row_ptr in the new order
            for (int k = 0; k < nt; ++k) {
                vican_tile_plan& t = P->tiles[k];
                std::vector<int32_t> rpn((size_t)T + 1, 0);
                for (int r = 0; r < T; ++r) rpn[r + 1] = rpn[r] + (t.rp[perm[r] + 1] - t.rp[perm[r]]);
                t.rp.swap(rpn);
            }
            nchunk = n; P->t_chunks = c0s; P->t_perm.swap(perm);
        }
    }
    for (int k = 0; k < nt; ++k) ptrs[k] = P->tiles[k].rp.data();
    if (nchunk < 0) nchunk = vican_plan_chunks_multi(T, nt, ptrs.data(), slots, cap_rows, P->t_chunks.data(), T + 2);
    if (nchunk == VICAN_ERR_CAPACITY)
        return ferr(VICAN_ERR_CAPACITY, "vican_plan_create: a timestep row has more than %d edges inside one camera tile: block-layout tiles are the host driver's (vican_amd.tiled)", slots);
    if (nchunk < 0) return nchunk;
    P->t_chunks.resize((size_t)nchunk + 1);
    int rows_max = 1;
    for (int k = 0; k < nchunk; ++k) rows_max = std::max(rows_max, P->t_chunks[k + 1] - P->t_chunks[k]);
    const long long lim = vican_lds_limit_bytes();
    int c_max = 0, copy_max = 1;
    for (vican_tile_plan& t : P->tiles) {
        vican_graph_t& g = t.g;
        const int Ck = t.c1 - t.c0;
        g = vican_graph_t{};
        g.n_cam = Ck; g.n_time = T; g.n_chunk = nchunk; g.slots = slots; g.max_rows = rows_max; g.storage = storage;
        g.block_threads = 64 * t.wg_waves; g.n_copy = t.n_copy; g.layout = VICAN_LAYOUT_WAVE; g.wg_waves = t.wg_waves;
        const long long lds = vican_wsweep_lds_bytes(Ck, rows_max, storage, t.n_copy, t.wg_waves);
        const int occ = (int)std::max(1LL, std::min(lim / std::max(lds, 1LL), (long long)(2048 / g.block_threads)));
        g.n_wg = std::max(1, std::min((nchunk + t.wg_waves - 1) / t.wg_waves, ncu * occ));
        int rpw = 1;
        for (int wg = 0; wg < g.n_wg; ++wg) {
            const long long k0 = (long long)wg * nchunk / g.n_wg, k1 = (long long)(wg + 1) * nchunk / g.n_wg;
            rpw = std::max(rpw, P->t_chunks[k1] - P->t_chunks[k0]);
        }
        t.rows_per_wg_max = rpw;
        const int per = (nchunk + g.n_wg - 1) / g.n_wg;
        g.wg_chunk_cap = ((13 * per + 10 * t.wg_waves - 1) / (10 * t.wg_waves) + 3) * t.wg_waves;
        t.rows_per_wg_sweep = (int)std::max(std::min((long long)T, (long long)g.wg_chunk_cap * rows_max), 1LL);
        g.slot_order = std::max(1.0, (double)t.E / T) < 48 * epl ? 1 : 0;
        const size_t nslot = (size_t)nchunk * slots;
        g.stream_nt = nslot * (9 * (storage == VICAN_STORE_F32 ? 4 : 8) + 4) > STREAM_NT_BYTES ? 1 : 0;
        c_max = std::max(c_max, Ck); copy_max = std::max(copy_max, t.n_copy);
    }
    // the operator as one launch (tiled.py _setup_fused) and the CG product as one launch (2..4 tiles)
    P->nwgt = ncu / nt;
    P->fused_ok = P->nwgt >= 1 && vican_tiled_op_lds_bytes(c_max, rows_max, storage, copy_max) <= lim;
    P->nwgt = std::max(1, P->nwgt);
    P->tcg_ok = P->have_t && nt >= 2 && nt <= 4;
    double n_add = 1, n_add_cg = 1;
    for (vican_tile_plan& t : P->tiles) {
        // adds into one camera accumulator by one workgroup of the fused launch = rows it handles (chunks are handed out per WAVEFRONT
        // with stride nwgt * 8: a workgroup takes up to 8 * ceil(n / (8 nwgt)))
        if (P->fused_ok)
            t.rows_per_wg_sweep = std::max(t.rows_per_wg_sweep, (int)std::min((long long)T, 8LL * ((nchunk + 8 * P->nwgt - 1) / (8 * P->nwgt)) * rows_max));
        t.n_add = (double)std::max(t.rows_per_wg_max, slots) + 1.0;
        n_add = std::max(n_add, t.n_add);
        n_add_cg = std::max(n_add_cg, t.n_add);
        if (P->tcg_ok) n_add_cg = std::max(n_add_cg, (double)((nchunk + P->nwgt - 1) / P->nwgt) * rows_max + 1.0);
    }
    P->n_add = n_add; P->n_add_cg = n_add_cg;
    // what vican_plan_describe reports for a tiled plan: the shared chunking, all tiles' workgroups
    P->g = vican_graph_t{};
    P->g.n_cam = C; P->g.n_time = T; P->g.storage = storage; P->g.n_chunk = nchunk; P->g.slots = slots; P->g.max_rows = rows_max;
    P->g.layout = VICAN_LAYOUT_WAVE; P->g.n_wg = nt * P->nwgt; P->g.block_threads = 512; P->g.wg_waves = 8; P->g.n_copy = copy_max;
    return VICAN_OK;
}

// ---- the tiles' share of the plan's arena (called from carve, both passes) ---------------------------------------------------------
void vican_facade_tiles_carve(vican_plan* P) {
    Arena& A = P->ar;
    const int nt = (int)P->tiles.size(), T1 = std::max(P->T, 1);
    const size_t s = P->storage == VICAN_STORE_F32 ? 4 : 8;
    P->t_chunk_row0 = A.take<int32_t>(P->t_chunks.size());
    for (vican_tile_plan& t : P->tiles) {
        const size_t nslot = (size_t)t.g.n_chunk * t.g.slots, Ck = t.c1 - t.c0;
        t.idx = A.take<int32_t>(nslot); t.idx16 = A.take<uint16_t>(nslot);
        t.blk = A.take<unsigned char>(9 * nslot * s); t.a = A.take<unsigned char>(nslot * s);
        if (P->have_t) { t.w = A.take<double>(nslot); t.u = A.take<double>(3 * nslot); t.v = A.take<double>(3 * nslot); }
        t.fx = A.take<double>(VICAN_FX_DOUBLES);
        t.zpart = A.take<double>((size_t)t.g.n_wg * 9 * Ck);                 // the tile's own plan: two-pass operator, J^T b, per-tile CG product
        t.zpart_f = A.take<double>((size_t)P->nwgt * 9 * Ck);                // the one-launch operator / CG product: nwgt workgroups per tile
    }
    P->t_rows = A.take<double>((size_t)3 * nt * T1);                          // per-tile row sums of a, block norms, row sums of w (pack time)
    P->t_ypart = A.take<double>((size_t)nt * T1 * 9);
    P->t_yp = A.take<double>((size_t)2 * nt * T1 * 9);
    P->t_wrow = A.take<double>((size_t)T1 * 9);
    if (P->have_t) P->t_acc = A.take<double>((size_t)nt * T1 * 3);
    P->t_dev = A.take<vican_tile_t>(nt);
    if (!P->t_perm.empty()) {
        P->t_perm_dev = A.take<int32_t>(P->t_perm.size());
        P->t_rows9 = A.take<double>((size_t)T1 * 9);         // a per-row argument in the plan's row order (Rt in, x_t out)
    }
}

// ---- pack: every tile's edges out of the caller's CSR arrays into its chunked planes; graph constants ----------------------------------
int vican_facade_tiles_pack(vican_plan* P, const int32_t* row_ptr, const int32_t* col, const void* blk, const void* a, const double* w,
                            const double* u, const double* v, double amax, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    const int nt = (int)P->tiles.size(), T = P->T, T1 = std::max(T, 1), storage = P->storage;
    const size_t sb = storage == VICAN_STORE_F32 ? 4 : 8;
    long long e_max = 0;
    for (const vican_tile_plan& t : P->tiles) e_max = std::max(e_max, t.E);
    const size_t nslot = (size_t)P->g.n_chunk * P->g.slots;
    // scratch: one tile's CSR arrays + the slot permutation of vican_pack_edges
    unsigned char* tmp = nullptr;
    Arena S;
    int32_t *rp_k = nullptr, *col_k = nullptr, *perm = nullptr;
    unsigned char *blk_k = nullptr, *a_k = nullptr;
    double *w_k = nullptr, *u_k = nullptr, *v_k = nullptr;
    auto carve = [&]() {
        S.used = 0;
        rp_k = S.take<int32_t>((size_t)T + 1); col_k = S.take<int32_t>((size_t)e_max); perm = S.take<int32_t>(nslot);
        blk_k = S.take<unsigned char>(9 * (size_t)e_max * sb); a_k = S.take<unsigned char>((size_t)e_max * sb);
        if (P->have_t) { w_k = S.take<double>((size_t)e_max); u_k = S.take<double>(3 * (size_t)e_max); v_k = S.take<double>(3 * (size_t)e_max); }
        return S.used + 256;
    };
    const size_t bytes = carve();
    if (hipMalloc((void**)&tmp, bytes) != hipSuccess) return ferr(VICAN_ERR_LAUNCH, "vican_plan_create: hipMalloc of %zu bytes of packing scratch failed", bytes);
    S.base = tmp; S.size = bytes;
    carve();
    int rc = VICAN_OK;
    if (hipMemcpyAsync(P->t_chunk_row0, P->t_chunks.data(), P->t_chunks.size() * 4, hipMemcpyHostToDevice, s) != hipSuccess)
        rc = ferr(VICAN_ERR_LAUNCH, "vican_plan_create: copy failed");
    if (rc >= 0 && P->t_perm_dev && hipMemcpyAsync(P->t_perm_dev, P->t_perm.data(), P->t_perm.size() * 4, hipMemcpyHostToDevice, s) != hipSuccess)
        rc = ferr(VICAN_ERR_LAUNCH, "vican_plan_create: copy failed");
    double* rows_a = P->t_rows; double* rows_n = P->t_rows + (size_t)nt * T1; double* rows_w = P->t_rows + (size_t)2 * nt * T1;
    for (int k = 0; k < nt && rc >= 0; ++k) {
        vican_tile_plan& t = P->tiles[k];
        t.g.blk = t.blk; t.g.idx = (const uint32_t*)t.idx; t.g.chunk_row0 = P->t_chunk_row0;
        if (hipMemcpyAsync(rp_k, t.rp.data(), t.rp.size() * 4, hipMemcpyHostToDevice, s) != hipSuccess) { rc = ferr(VICAN_ERR_LAUNCH, "vican_plan_create: copy failed"); break; }
        if (storage == VICAN_STORE_F32)
            hipLaunchKernelGGL(tile_gather_kernel<float>, dim3((T + 3) / 4), dim3(256), 0, s, T, t.c0, t.c1, P->t_perm_dev, row_ptr, col, (const float*)blk, (const float*)a, w, u, v,
                               rp_k, col_k, (float*)blk_k, (float*)a_k, w_k, u_k, v_k);
        else
            hipLaunchKernelGGL(tile_gather_kernel<double>, dim3((T + 3) / 4), dim3(256), 0, s, T, t.c0, t.c1, P->t_perm_dev, row_ptr, col, (const double*)blk, (const double*)a, w, u, v,
                               rp_k, col_k, (double*)blk_k, (double*)a_k, w_k, u_k, v_k);
        rc = vican_pack_edges(&t.g, rp_k, col_k, blk_k, a_k, w ? w_k : nullptr, w ? u_k : nullptr, w ? v_k : nullptr, t.a, t.w, t.u, t.v, perm, stream);
        if (rc >= 0) rc = vican_pack_idx16(&t.g, t.idx16, stream);
        if (rc < 0) break;
        t.g.idx16 = t.idx16;
        void* cam_ws = t.zpart;                                   // C_k 64-bit words of scratch
        rc = vican_edge_sums(&t.g, t.a, storage == VICAN_STORE_F64, amax > 0 ? amax : 1.0, rows_a + (size_t)k * T1, P->cam_sum_a + t.c0, cam_ws, stream);
        if (rc >= 0) rc = vican_block_norms(&t.g, rows_n + (size_t)k * T1, t.fx, stream);
        if (rc >= 0 && P->have_t) rc = vican_edge_sums(&t.g, t.w, 1, P->wmax, rows_w + (size_t)k * T1, P->cam_sum_w + t.c0, cam_ws, stream);
        // (the scratch is reused by the next tile: the stream orders it)
    }
    // global row constants: sums over the tiles in tile order (tiled.py: row_sum_a, rnorm, row_sum_w)
    if (rc >= 0) rc = vican_sum_apply3(T, 1, nullptr, rows_a, nt, T1, P->row_sum_a, stream);
    if (rc >= 0) rc = vican_sum_apply3(T, 1, nullptr, rows_n, nt, T1, P->rnorm, stream);
    if (rc >= 0 && P->have_t) rc = vican_sum_apply3(T, 1, nullptr, rows_w, nt, T1, P->row_sum_w, stream);
    // descriptors of the one-launch operator, share buffers armed
    if (rc >= 0) {
        P->t_host.assign(nt, vican_tile_t{});
        for (int k = 0; k < nt; ++k) {
            const vican_tile_plan& t = P->tiles[k];
            vican_tile_t& e = P->t_host[k];
            e.g = t.g; e.x = nullptr; e.zpart = t.zpart_f; e.fx = t.fx;
            e.ypart[0] = P->t_yp + (size_t)k * T1 * 9; e.ypart[1] = P->t_yp + (size_t)(nt + k) * T1 * 9;
        }
        if (hipMemcpyAsync(P->t_dev, P->t_host.data(), (size_t)nt * sizeof(vican_tile_t), hipMemcpyHostToDevice, s) != hipSuccess)
            rc = ferr(VICAN_ERR_LAUNCH, "vican_plan_create: copy failed");
        if (rc >= 0) rc = vican_tiled_op_sentinel(P->t_yp, (int64_t)2 * nt * T1 * 9, stream);
        P->t_parity = 0;
    }
    if (hipStreamSynchronize(s) != hipSuccess && rc >= 0) rc = ferr(VICAN_ERR_LAUNCH, "vican_plan_create: packing the camera tiles failed");
    hipFree(tmp);
    for (vican_tile_plan& t : P->tiles) { t.rp.clear(); t.rp.shrink_to_fit(); }
    return rc;
}

// ---- scales: omega = max_t |Lambda_t^-1|_F rnorm[t] over ALL tiles' row norms, every tile's scale buffer finished (tiled.py _refresh_scales)
int vican_facade_tiles_refresh(vican_plan* P, void* stream) {
    const int nt = (int)P->tiles.size();
    CK(vican_duals_bound(P->T, P->lamT, P->rnorm, P->tiles[0].fx, stream));
    double* fxs[64]; double nadd[64];
    for (int k = 0; k < nt; ++k) { fxs[k] = P->tiles[k].fx; nadd[k] = (double)P->tiles[k].rows_per_wg_sweep + 1.0; }
    return vican_fx_finish_multi(fxs, nadd, nt, X_BOUND, P->storage, stream);
}

namespace {
// ypart[k] = sum_{c in tile k} M_ct^T x_c for every tile
int rows_pass(vican_plan* P, const double* x, void* stream) {
    const size_t T1 = std::max(P->T, 1);
    for (size_t k = 0; k < P->tiles.size(); ++k) {
        const vican_tile_plan& t = P->tiles[k];
        CK(vican_tile_rows(&t.g, x + (size_t)9 * t.c0, P->t_ypart + k * T1 * 9, t.fx, stream));
    }
    return VICAN_OK;
}
}  // namespace

// z = P x (this rank's rows): [3C][3] in, [3C][3] out
int vican_facade_tiles_op_z(vican_plan* P, const double* x, double* z, void* stream) {
    const int nt = (int)P->tiles.size();
    const size_t T1 = std::max(P->T, 1);
    if (P->fused_ok) {
        const int rc = vican_tiled_op_z(P->t_host.data(), P->t_dev, nt, P->nwgt, P->lamT, x, z, P->t_parity, stream);
        if (rc != VICAN_ERR_CAPACITY) {
            CK(rc);
            P->t_parity ^= 1;
            return VICAN_OK;
        }
        P->fused_ok = false;                                     // (grid not co-resident: dropped for the rest of the plan's life)
    }
    CK(rows_pass(P, x, stream));
    CK(vican_sum_apply3(P->T, 9, P->lamT, P->t_ypart, nt, (int64_t)T1 * 9, P->t_wrow, stream));
    for (const vican_tile_plan& t : P->tiles) CK(vican_tile_cams(&t.g, P->t_wrow, t.zpart, t.fx, z + (size_t)9 * t.c0, stream));
    return VICAN_OK;
}

// Z_t = sum_c M_ct^T R_c over all tiles, then R_t, Lambda_t^-1 = U S^-1 U^T per row (bipgo.py:318-332); scales by the caller
int vican_facade_tiles_dual_update(vican_plan* P, const double* rc, void* stream) {
    const size_t T1 = std::max(P->T, 1);
    CK(rows_pass(P, rc, stream));
    CK(vican_sum_apply3(P->T, 9, nullptr, P->t_ypart, (int)P->tiles.size(), (int64_t)T1 * 9, P->t_wrow, stream));
    return vican_polar_dual(P->T, P->t_wrow, P->Rt, P->lamT, 2, stream);
}

// J^T b: b_c per tile (complete for the tile's cameras), b_t = sum of the tiles' row parts in tile order
int vican_facade_tiles_rhs(vican_plan* P, const double* rc, const double* Rt, void* stream) {
    const size_t T1 = std::max(P->T, 1);
    for (size_t k = 0; k < P->tiles.size(); ++k) {
        const vican_tile_plan& t = P->tiles[k];
        CK(vican_trans_rhs(&t.g, t.u, t.v, rc + (size_t)9 * t.c0, Rt, P->t_acc + k * T1 * 3, P->b_c + (size_t)3 * t.c0, t.zpart, P->gmax, t.n_add, stream));
    }
    return vican_sum_apply3(P->T, 3, nullptr, P->t_acc, (int)P->tiles.size(), (int64_t)T1 * 3, P->b_t, stream);
}

// First half of a CG iteration (vican_cg_iter_local on tiles - tiled.py cg_iter_local): qcpq = [q_c partial | p_t.q_t]
int vican_facade_tiles_cg_local(vican_plan* P, double rtol, int n_part, void* stream) {
    const int nt = (int)P->tiles.size(), C = P->C;
    const size_t T1 = std::max(P->T, 1);
    CK(vican_cg_begin(C, P->r_c, P->p_c, rtol, P->rr_part, n_part, P->n_add_cg, P->st, stream));
    CK(vican_cg_update_pt(P->T, P->r_t, P->p_t, P->st, stream));
    bool done = false;
    if (P->tcg_ok) {
        vican_cg_tile_t ct[4];
        for (int k = 0; k < nt; ++k) {
            const vican_tile_plan& t = P->tiles[k];
            ct[k].g = t.g; ct[k].w = t.w; ct[k].p_c = P->p_c + (size_t)3 * t.c0; ct[k].acc_t = P->t_acc + (size_t)k * T1 * 3; ct[k].qc_part = t.zpart_f;
        }
        const int rc = vican_cg_sweep_tiles(ct, nt, P->nwgt, P->p_t, P->st, stream);
        if (rc == VICAN_ERR_CAPACITY) P->tcg_ok = false;       // (launch shapes differ / small graphs: per-tile launches)
        else {
            CK(rc);
            const void* parts[4]; int32_t ncams[4];
            for (int k = 0; k < nt; ++k) { parts[k] = P->tiles[k].zpart_f; ncams[k] = P->tiles[k].c1 - P->tiles[k].c0; }
            CK(vican_cg_fold_tiles(parts, ncams, nt, P->nwgt, P->qcpq, P->st, stream));
            done = true;
        }
    }
    if (!done)
        for (int k = 0; k < nt; ++k) {
            const vican_tile_plan& t = P->tiles[k];
            CK(vican_cg_sweep_partial(&t.g, t.w, P->p_c + (size_t)3 * t.c0, P->p_t, P->t_acc + (size_t)k * T1 * 3, t.zpart, P->st, stream));
            CK(vican_cg_fold(t.zpart, t.g.n_wg, t.c1 - t.c0, nullptr, P->qcpq + (size_t)3 * t.c0, P->st, stream));
        }
    const int nb = vican_cg_combine_rows(P->T, nt, (int64_t)T1 * 3, P->row_sum_w, P->p_t, P->t_acc, P->q_t, P->pq_part, 1024, P->st, stream);
    if (nb < 0) return nb;
    return vican_cg_reduce_pq(P->pq_part, nb, P->qcpq + (size_t)3 * C, P->st, stream);
}

// Per-row arrays at the boundary of a plan whose rows are in an order of its own (no-ops otherwise): `to_plan` copies the caller's
// [T][width] array into the plan's order (returns the array to use), `to_caller` writes a plan-order array back in the caller's order
const double* vican_facade_tiles_rows_in(vican_plan* P, const double* src, int width, double* scratch, void* stream) {
    if (!P->t_perm_dev) return src;
    hipLaunchKernelGGL(rows_permute_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, (long long)P->T, width, P->t_perm_dev, src, scratch, 0);
    return scratch;
}
void vican_facade_tiles_rows_out(vican_plan* P, const double* src, int width, double* dst, void* stream) {
    if (!P->t_perm_dev) {
        if (src != dst) hipMemcpyAsync(dst, src, (size_t)P->T * width * 8, hipMemcpyDeviceToDevice, (hipStream_t)stream);
        return;
    }
    hipLaunchKernelGGL(rows_permute_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, (long long)P->T, width, P->t_perm_dev, src, dst, 1);
}
