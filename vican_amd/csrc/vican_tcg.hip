// vican_tcg.hip - the CG Laplacian product of a camera-TILED graph (more cameras than one LDS table holds) in ONE launch.
//
// Per tile the product needs nothing from the other tiles: a tile's sweep yields its share of the row sums sum_{c in tile} w p_c
// (combined afterwards, vican_cg_combine_rows) and the complete camera sums of its own cameras.  Rounds 3-4 launched
// vican_cg_sweep_partial once per tile, each launch on all compute units for a fraction of the edges (42 us per tile on the
// wide workload, four tiles); here workgroup b works for tile b % n_tile as workgroup b / n_tile of that tile's share of the
// grid - the same device function (cg_wsweep_impl, vican_cgw_impl.h) in its `partial` mode, one launch.
// The tiles' descriptors travel BY VALUE in the kernel arguments (up to four): pointers that reach a kernel through memory are
// generic pointers to the compiler and every access through them a FLAT instruction (vican_tsweep.hip).
#include "vican_cgw_impl.h"

struct tcg_tile_t { vican_graph_t g; const double* w; const double* p_c; double* acc_t; u64* qc_part; };

template <int NW, int EPL, int TRIPS, int CP, bool NT>
__global__ __launch_bounds__(NW * 64) void cg_tiles_kernel(const tcg_tile_t t0, const tcg_tile_t t1, const tcg_tile_t t2, const tcg_tile_t t3,
                                                           const int n_tile, const double* __restrict__ p_t,
                                                           const vican_cg_state_t* __restrict__ st) {
    const int tile = (int)blockIdx.x % n_tile, wg = (int)blockIdx.x / n_tile, nwg = (int)gridDim.x / n_tile;
    // (deg_t, r_t and pq_part are not touched in partial mode: any readable pointers)
#define TCG_RUN(T) cg_wsweep_impl<NW, EPL, TRIPS, CP, NT>(T.g, T.w, p_t, T.p_c, p_t, const_cast<double*>(p_t), T.acc_t, T.qc_part, T.acc_t, st, 1, wg, nwg)
    switch (tile) {
        case 0: TCG_RUN(t0); break;
        case 1: TCG_RUN(t1); break;
        case 2: TCG_RUN(t2); break;
        default: TCG_RUN(t3); break;
    }
#undef TCG_RUN
}

// tiles: n_tile (2..4) descriptors - g: the tile (wave layout; all tiles with the same slots, wg_waves, stream_nt, plane stride
// and ceil(3 max_rows / 64)), w: its weights in its slot order, p_c: its slice of the camera vector [C_tile][3], acc_t [T][3]:
// receives its share of the row sums, qc_part: n_wg_tile slabs of 6 C_tile 64-bit words (fold with vican_cg_fold(pq_part = NULL)).
// p_t: already updated (vican_cg_update_pt).  n_tile * n_wg_tile workgroups.
extern "C" int vican_cg_sweep_tiles(const vican_cg_tile_t* tiles, int32_t n_tile, int32_t n_wg_tile, const double* p_t,
                                    const vican_cg_state_t* st, void* stream) {
    if (!tiles || n_tile < 2 || n_tile > 4 || n_wg_tile <= 0 || !p_t || !st) return set_err(VICAN_ERR_ARG, "vican_cg_sweep_tiles: bad argument (2..4 tiles)");
    tcg_tile_t t[4];
    const vican_graph_t& g0 = tiles[0].g;
    const int nw = g0.wg_waves >= 12 ? 12 : (g0.wg_waves >= 8 ? 8 : 4), epl = g0.slots / 64, cp = (int)plane_stride(g0.n_cam);
    const int trips = (3 * g0.max_rows + 63) / 64;
    size_t lds = 0;
    for (int k = 0; k < 4; ++k) {
        const vican_cg_tile_t& s = tiles[k < n_tile ? k : 0];
        if (k < n_tile) {
            if (int rc = vican_check_graph(&s.g, "vican_cg_sweep_tiles")) return rc;
            if (s.g.layout != VICAN_LAYOUT_WAVE || !s.g.idx16 || !s.w || !s.p_c || !s.acc_t || !s.qc_part || s.g.n_chunk == 0)
                return set_err(VICAN_ERR_ARG, "vican_cg_sweep_tiles: tiles must be non-empty wave layouts (2-byte index packed: vican_pack_idx16) with all buffers set");
            const int nwk = s.g.wg_waves >= 12 ? 12 : (s.g.wg_waves >= 8 ? 8 : 4);
            if (nwk != nw || s.g.slots != g0.slots || s.g.stream_nt != g0.stream_nt || (int)plane_stride(s.g.n_cam) != cp ||
                (3 * s.g.max_rows + 63) / 64 != trips)
                return set_err(VICAN_ERR_CAPACITY, "vican_cg_sweep_tiles: the tiles' launch shapes differ");
            const size_t l = (size_t)vican_cg_wsweep_lds_bytes(s.g.n_cam, s.g.max_rows, s.g.n_copy, nw);
            lds = l > lds ? l : lds;
        }
        t[k].g = s.g; t[k].w = s.w; t[k].p_c = s.p_c; t[k].acc_t = s.acc_t; t[k].qc_part = (u64*)s.qc_part;
    }
    if ((int64_t)lds > vican_lds_limit_bytes()) return set_err(VICAN_ERR_CAPACITY, "vican_cg_sweep_tiles: camera tables / row staging do not fit in LDS");
    if (trips > 3) return set_err(VICAN_ERR_CAPACITY, "vican_cg_sweep_tiles: more than 64 rows per chunk");
    const bool nt = g0.stream_nt != 0;
    hipStream_t s_ = (hipStream_t)stream;
    const int grid = n_tile * n_wg_tile;
#define TCG_LAUNCH_(NW_, E_, T_, CP_, NT_)                                                                                 \
    do {                                                                                                                   \
        auto kern = cg_tiles_kernel<NW_, E_, T_, CP_, NT_>;                                                                \
        static size_t conf = 0;                                                                                            \
        if (lds > conf) { hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); conf = lds; } \
        VICAN_LAUNCH_SWEEP(kern, dim3(grid), dim3(NW_ * 64), lds, s_, t[0], t[1], t[2], t[3], (int)n_tile, p_t, st);       \
    } while (0)
#define TCG_LAUNCH(NW_, E_, T_)                                                                                            \
    do {                                                                                                                   \
        if (nt) { if (cp == 256) TCG_LAUNCH_(NW_, E_, T_, 256, true); else if (cp == 512) TCG_LAUNCH_(NW_, E_, T_, 512, true); else TCG_LAUNCH_(NW_, E_, T_, 1024, true); } \
        else    { if (cp == 256) TCG_LAUNCH_(NW_, E_, T_, 256, false); else if (cp == 512) TCG_LAUNCH_(NW_, E_, T_, 512, false); else TCG_LAUNCH_(NW_, E_, T_, 1024, false); } \
    } while (0)
#define TCG_PICK(NW_)                                                                                                      \
    do {                                                                                                                   \
        if (epl == 4) { if (trips <= 1) TCG_LAUNCH(NW_, 4, 1); else if (trips == 2) TCG_LAUNCH(NW_, 4, 2); else TCG_LAUNCH(NW_, 4, 3); } \
        else          { if (trips <= 1) TCG_LAUNCH(NW_, 2, 1); else if (trips == 2) TCG_LAUNCH(NW_, 2, 2); else TCG_LAUNCH(NW_, 2, 3); } \
    } while (0)
    // (a launch per tile costs nothing on graphs too small for 12 wavefronts per workgroup: only that shape is built)
    if (nw != 12) return set_err(VICAN_ERR_CAPACITY, "vican_cg_sweep_tiles: built for tiles planned with 12 wavefronts per workgroup");
    TCG_PICK(12);
#undef TCG_PICK
#undef TCG_LAUNCH
#undef TCG_LAUNCH_
    LAUNCH_CHECK("vican_cg_sweep_tiles");
    return VICAN_OK;
}
