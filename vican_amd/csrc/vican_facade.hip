// vican_facade.hip - the four-call boundary of SURVEY.md 8(b): vican_plan_create / vican_solve_rot / vican_solve_trans /
// vican_plan_destroy.  What a maintainer of the reference binds when he wants the numerics of
// large_bipartite_so3sync (bipgo.py:145-350) and of the normal-equation CG (bipgo.py:445-478) behind ONE handle and has
// no use for the ~70 granular entry points the Python driver (vican_amd/solver.py, device.py) composes.
//
// Everything here is HOST code on top of those entry points - the same kernels, the same fixed-point scales, the same
// stopping rules - with the plain schedule: layout planning as device.py's _Layout (wave layout where every row fits a
// 64-lane chunk, else block layout), block Lanczos with a convergence check (device Ritz step, one blocking 128-byte read) every
// few steps, launch-sequence camera-side step, the fused dual update, CG iterations (vican_cg_iter_fused) in bursts of eight
// with the state polled in between.  No speculation, no HIP graphs.  More than 1024 cameras: the plan cuts the cameras into tiles
// and runs the tiled schedule (vican_facade_tiles.hip; wave-layout tiles with a shared chunking).  vican_plan_set_comm turns the same plan into ONE RANK
// of a timestep-sharded solve (the plan holds this rank's rows; every camera-side quantity is replicated): the sweeps' camera
// partials are all-reduced from C in stream order (vican_block_op_z_comm), the CG runs vican_cg_iter_comm - the schedule
// vican_amd/solver.py takes for sharded runs.  The Python driver remains the fast path; this one is the small stable surface.  The library owns the plan's device memory (one arena); inputs and outputs are the caller's.
#include "vican_facade_impl.h"

namespace {

__global__ void facade_maxima_kernel(long long n, int storage, const void* a, const double* w, const double* u, const double* v,
                                     double* out /* [3]: max|a|, max w, max(|u|+|v|) */) {
    double ma = 0, mw = 0, mg = 0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        ma = fmax(ma, fabs(storage == VICAN_STORE_F32 ? (double)((const float*)a)[i] : ((const double*)a)[i]));
        if (w) {
            mw = fmax(mw, w[i]);
            const double nu = sqrt(u[3 * i] * u[3 * i] + u[3 * i + 1] * u[3 * i + 1] + u[3 * i + 2] * u[3 * i + 2]);
            const double nv = sqrt(v[3 * i] * v[3 * i] + v[3 * i + 1] * v[3 * i + 1] + v[3 * i + 2] * v[3 * i + 2]);
            mg = fmax(mg, nu + nv);
        }
    }
    for (int o = 32; o > 0; o >>= 1) { ma = fmax(ma, __shfl_xor(ma, o, 64)); mw = fmax(mw, __shfl_xor(mw, o, 64)); mg = fmax(mg, __shfl_xor(mg, o, 64)); }
    if ((threadIdx.x & 63) == 0) { atomic_max_pos(out, ma); atomic_max_pos(out + 1, mw); atomic_max_pos(out + 2, mg); }
}

__global__ void facade_seed_kernel(int n, double* x) {         // identity at camera 0, zero elsewhere ([n][3] row-major)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < 3 * n) x[i] = (i < 9 && i / 3 == i % 3) ? 1.0 : 0.0;
}

using vican_facade::n_cu;

// device.py _Layout: sizes of the chunked layout for this graph
int plan_layout(vican_plan* P, const std::vector<int32_t>& rp, std::vector<int32_t>& chunk_row0) {
    const int C = P->C, T = P->T, storage = P->storage, epl = P->epl, ncu = n_cu();
    const long long E = P->E;
    const long long lim = vican_lds_limit_bytes();
    int deg_max = 0;
    for (int t = 0; t < T; ++t) deg_max = std::max(deg_max, rp[t + 1] - rp[t]);
    const double deg_avg = std::max(1.0, T ? (double)E / T : 1.0);
    vican_graph_t& g = P->g;
    g.n_cam = C; g.n_time = T; g.storage = storage;
    const bool wave = deg_max <= 64 * epl && C <= 1024 && E > 0;
    int slots, max_rows, n_copy, wg_waves = 0, block_threads;
    if (wave) {
        slots = 64 * epl;
        int rows_target = std::max(1, std::min(C >= 1024 ? 63 : 64, (int)std::ceil(1.25 * slots / deg_avg) + 1));   // (2-byte index: vican_graph_t.idx16)
        n_copy = 1;
        while (n_copy < 8 && n_copy * epl < deg_avg) n_copy *= 2;
        wg_waves = 12;
        if (E < 12LL * slots * ncu) wg_waves = E >= 8LL * slots * ncu ? 8 : 4;
        auto fits = [&](int rows, int nc, int nw) { return vican_wsweep_lds_bytes(C, rows, storage, nc, nw) <= lim; };
        while (!fits(rows_target, n_copy, wg_waves) && n_copy > 1) n_copy /= 2;
        while (!fits(rows_target, n_copy, wg_waves) && wg_waves > 4) wg_waves -= 4;
        while (!fits(rows_target, n_copy, wg_waves) && rows_target > 1) --rows_target;
        if (!fits(rows_target, n_copy, wg_waves)) return ferr(VICAN_ERR_CAPACITY, "vican_plan_create: camera tables (C=%d) do not fit in LDS", C);
        max_rows = rows_target; block_threads = 64 * wg_waves;
    } else {
        block_threads = E >= 768LL * epl * ncu ? 768 : 256;
        if (deg_max > 256 * epl) block_threads = 768;
        if (deg_max > 768 * epl) block_threads = 1024;
        slots = block_threads * epl;
        if (deg_max > slots) return ferr(VICAN_ERR_CAPACITY, "vican_plan_create: a timestep row has %d edges, a chunk holds %d", deg_max, slots);
        const int rows_target = std::min(65535, (int)std::ceil(1.25 * slots / deg_avg) + 1);
        n_copy = 8;
        while (n_copy > 1 && vican_max_rows_for(C, storage, n_copy) < rows_target) n_copy /= 2;
        max_rows = vican_max_rows_for(C, storage, n_copy);
        if (max_rows < 1) return ferr(VICAN_ERR_CAPACITY, "vican_plan_create: camera tables (C=%d) do not fit in LDS", C);
        max_rows = std::min(max_rows, std::max(rows_target, 1));
    }
    chunk_row0.assign((size_t)T + 2, 0);
    const int nchunk = vican_plan_chunks(T, rp.data(), slots, max_rows, chunk_row0.data(), T + 2);
    if (nchunk < 0) return nchunk;
    chunk_row0.resize((size_t)nchunk + 1);
    int rows_max = 1;
    for (int k = 0; k < nchunk; ++k) rows_max = std::max(rows_max, chunk_row0[k + 1] - chunk_row0[k]);
    g.n_chunk = nchunk; g.slots = slots; g.max_rows = rows_max; g.block_threads = block_threads; g.n_copy = n_copy;
    g.layout = wave ? VICAN_LAYOUT_WAVE : VICAN_LAYOUT_BLOCK; g.wg_waves = wg_waves;
    const long long lds = wave ? vican_wsweep_lds_bytes(C, rows_max, storage, n_copy, wg_waves) : vican_sweep_lds_bytes(C, rows_max, storage, n_copy);
    const int per_wg = wave ? wg_waves : 1;
    const int occ = (int)std::max(1LL, std::min(lim / std::max(lds, 1LL), (long long)(2048 / block_threads)));
    g.n_wg = std::max(1, std::min((nchunk + per_wg - 1) / per_wg, ncu * occ));
    int rpw = 1;
    for (int wg = 0; wg < g.n_wg && nchunk; ++wg) {
        const long long k0 = (long long)wg * nchunk / g.n_wg, k1 = (long long)(wg + 1) * nchunk / g.n_wg;
        rpw = std::max(rpw, chunk_row0[k1] - chunk_row0[k0]);
    }
    P->rows_per_wg_max = rpw;
    const int per = nchunk ? (nchunk + g.n_wg - 1) / g.n_wg : 1;
    if (wave) {
        g.wg_chunk_cap = ((13 * per + 10 * wg_waves - 1) / (10 * wg_waves) + 3) * wg_waves;
        P->rows_per_wg_sweep = std::max(std::min((long long)T, (long long)g.wg_chunk_cap * rows_max), 1LL);
    } else {
        g.wg_chunk_cap = per + std::max(2, (per + 7) / 8);
        P->rows_per_wg_sweep = (int)std::max<long long>(std::max<long long>(rpw, std::min((long long)T, (long long)g.wg_chunk_cap * rows_max)), 1);
    }
    g.slot_order = deg_avg < 48 * epl ? 1 : 0;
    const size_t nslot = (size_t)std::max(1, nchunk) * slots;
    g.stream_nt = nslot * (9 * (storage == VICAN_STORE_F32 ? 4 : 8) + 4) > STREAM_NT_BYTES ? 1 : 0;
    return VICAN_OK;
}

size_t carve(vican_plan* P, size_t n_row0) {
    Arena& A = P->ar;
    A.used = 0;
    const int C = P->C, T1 = std::max(P->T, 1), n = 3 * C;
    const bool tiled = !P->tiles.empty();                    // (the graph arrays of a tiled plan are its tiles': vican_facade_tiles_carve)
    const size_t nslot = tiled ? 0 : (size_t)std::max(1, P->g.n_chunk) * P->g.slots, s = P->storage == VICAN_STORE_F32 ? 4 : 8;
    const int n_wg = tiled ? 1 : std::max(P->g.n_wg, 1);
    P->idx = A.take<int32_t>(nslot); P->chunk_row0 = A.take<int32_t>(n_row0);
    // (wave layout: the 2-byte index the edge sweeps stream, vican_graph_t.idx16)
    P->idx16 = P->g.layout == VICAN_LAYOUT_WAVE && !tiled ? A.take<uint16_t>(nslot) : nullptr;
    P->blk = A.take<unsigned char>(9 * nslot * s); P->a = A.take<unsigned char>(nslot * s);
    if (P->have_t) { P->w = A.take<double>(nslot); P->u = A.take<double>(3 * nslot); P->v = A.take<double>(3 * nslot); }
    P->row_sum_a = A.take<double>(T1); P->cam_sum_a = A.take<double>(C); P->rnorm = A.take<double>(T1); P->fx = A.take<double>(20);
    if (P->have_t) { P->row_sum_w = A.take<double>(T1); P->cam_sum_w = A.take<double>(C); }
    P->zpart = A.take<double>((size_t)n_wg * 9 * C);
    const int m = std::max(1, std::min(M_MAX, n / 3));
    P->ld = n; P->hw = 3 * (m + 1) * 3; P->hb_stride = P->hw + 9;
    P->V = A.take<double>((size_t)3 * (m + 1) * n); P->R = A.take<double>(3 * (size_t)n); P->H = A.take<double>(3 * (m + 1) * 3);
    P->G = A.take<double>(9); P->beta0 = A.take<double>(9); P->HB = A.take<double>((size_t)m * P->hb_stride);
    P->Yd = A.take<double>((size_t)3 * (m + 1) * 3); P->status = A.take<double>(16); P->gate = A.take<int32_t>(4); P->coop_sync = A.take<int32_t>(4);
    P->xrow = A.take<double>(3 * (size_t)n); P->z = A.take<double>(3 * (size_t)n); P->X = A.take<double>(3 * (size_t)n);
    P->Xp = A.take<double>(3 * (size_t)n); P->x0 = A.take<double>(3 * (size_t)n); P->rc = A.take<double>(3 * (size_t)n);
    P->lamC = A.take<double>(9 * (size_t)C); P->cam_deg = A.take<double>(C); P->lamT = A.take<double>(9 * (size_t)T1);
    P->Rt = A.take<double>(9 * (size_t)T1); P->zraw = A.take<double>(3 * (size_t)n);
    P->coop_ws = A.take<double>((size_t)vican_lanczos_coop_ws_doubles(C));
    if (P->have_t) {
        P->cgres_ws = A.take<double>(tiled ? 1 : (size_t)vican_cg_resident_ws_doubles(C, n_wg));
        P->w32 = A.take<float>(nslot); P->w32_flag = A.take<int32_t>(4);
        P->b_c = A.take<double>(3 * (size_t)C); P->b_t = A.take<double>(3 * (size_t)T1); P->r_c = A.take<double>(3 * (size_t)C);
        P->p_c = A.take<double>(3 * (size_t)C); P->r_t = A.take<double>(3 * (size_t)T1); P->p_t = A.take<double>(3 * (size_t)T1);
        P->q_t = A.take<double>(3 * (size_t)T1); P->qcpq = A.take<double>(3 * (size_t)C + 1);
        P->pq_part = A.take<double>(tiled ? 1024 : n_wg); P->rr_part = A.take<double>(1536); P->ws = A.take<double>(1024);
        P->st = (vican_cg_state_t*)A.take<double>(19);
        P->cg_ticket = A.take<uint32_t>(256);        // (vican_cg_iter_fused: the fold's p.q partials; zeroed with the arena)
        P->msg = A.take<double>(3 * (size_t)C + VICAN_CG_PQ_SLICES);      // (vican_cg_iter_comm: [q_c partial | slices of p_t.q_t])
    }
    P->setup_msg = A.take<double>((size_t)C + 4);
    if (tiled) vican_facade_tiles_carve(P);
    return A.used + 256;
}

#define CK(call) do { const int rc_ = (call); if (rc_ < 0) return rc_; } while (0)
#define HIPCK(call, what) do { if ((call) != hipSuccess) return ferr(VICAN_ERR_LAUNCH, "%s: %s failed", what, #call); } while (0)

int fx_finish(vican_plan* P, void* stream) {
    if (!P->tiles.empty()) return vican_facade_tiles_refresh(P, stream);
    return vican_fx_finish(P->fx, X_BOUND, (double)P->rows_per_wg_sweep + 1.0, P->storage, stream);
}

}  // namespace

extern "C" int vican_plan_create(int32_t n_cam, int32_t n_time, int64_t n_edges, int32_t storage, const int32_t* row_ptr,
                                 const int32_t* col, const void* blk, const void* a, const double* w, const double* u,
                                 const double* v, const double* deg_t, const double* deg_c, void* stream, vican_plan_t** plan_out) {
    if (!plan_out) return ferr(VICAN_ERR_ARG, "vican_plan_create: plan_out is NULL");
    *plan_out = nullptr;
    if (n_cam <= 0 || n_time <= 0 || n_edges <= 0 || !row_ptr || !col || !blk || !a || (storage != VICAN_STORE_F32 && storage != VICAN_STORE_F64) ||
        ((w || u || v) && !(w && u && v)))
        return ferr(VICAN_ERR_ARG, "vican_plan_create: bad argument");
    if (n_cam > 65535) return ferr(VICAN_ERR_CAPACITY, "vican_plan_create: more than 65535 cameras are not supported by the packed edge index");
    hipStream_t s = (hipStream_t)stream;
    vican_plan* P = new vican_plan();
    for (double& f : P->floor_level) f = -1.0;
    P->C = n_cam; P->T = n_time; P->E = n_edges; P->storage = storage; P->epl = storage == VICAN_STORE_F32 ? 4 : 2; P->have_t = w != nullptr;
    std::vector<int32_t> rp((size_t)n_time + 1), c0;
    auto fail = [&](int rc) { vican_plan_destroy(P); return rc; };
    if (hipMemcpyAsync(rp.data(), row_ptr, rp.size() * 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
        return fail(ferr(VICAN_ERR_LAUNCH, "vican_plan_create: cannot read row_ptr"));
    if (rp[0] != 0 || rp[n_time] != n_edges) return fail(ferr(VICAN_ERR_ARG, "vican_plan_create: row_ptr does not span n_edges"));
    const bool tiled = n_cam > vican_facade_tile_cams();
    int rc = tiled ? vican_facade_tiles_layout(P, row_ptr, col, stream) : plan_layout(P, rp, c0);
    if (rc < 0) return fail(rc);
    if (tiled) c0.assign(1, 0);
    const size_t bytes = carve(P, c0.size());
    if (hipMalloc((void**)&P->ar.base, bytes) != hipSuccess) return fail(ferr(VICAN_ERR_LAUNCH, "vican_plan_create: hipMalloc of %zu bytes failed", bytes));
    P->ar.size = bytes;
    carve(P, c0.size());
    if (hipHostMalloc((void**)&P->status_host, 2048 * sizeof(double)) != hipSuccess) return fail(ferr(VICAN_ERR_LAUNCH, "vican_plan_create: hipHostMalloc failed"));
    if (hipMemsetAsync(P->ar.base, 0, bytes, s) != hipSuccess) return fail(ferr(VICAN_ERR_LAUNCH, "vican_plan_create: memset failed"));
    if (hipMemcpyAsync(P->chunk_row0, c0.data(), c0.size() * 4, hipMemcpyHostToDevice, s) != hipSuccess) return fail(ferr(VICAN_ERR_LAUNCH, "vican_plan_create: copy failed"));
    if (!tiled) {
        P->g.blk = P->blk; P->g.idx = (const uint32_t*)P->idx; P->g.chunk_row0 = P->chunk_row0;
        P->n_add = (double)std::max(P->rows_per_wg_max, P->g.slots) + 1.0; P->n_add_cg = P->n_add;
    }
    int32_t* perm = nullptr;
    if (tiled) {
        // camera tiles: maxima over the whole edge set (bounds of the fixed-point scales), every tile packed, constants summed
        P->coop_ok = n_cam <= 8192; P->cgres_ok = false;
        double* mx = P->ws ? P->ws : P->G;
        hipLaunchKernelGGL(facade_maxima_kernel, dim3(256), dim3(256), 0, s, (long long)n_edges, storage, a, w, u, v, mx);
        double h[3] = {1, 1, 1};
        if (hipMemcpyAsync(h, mx, 24, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess)
            return fail(ferr(VICAN_ERR_LAUNCH, "vican_plan_create: maxima read failed"));
        hipMemsetAsync(mx, 0, 24, s);
        if (P->have_t) { P->wmax = h[1]; P->gmax = h[2]; }
        rc = vican_facade_tiles_pack(P, row_ptr, col, blk, a, w, u, v, h[0], stream);
        if (rc >= 0 && P->have_t && deg_t) {                 // (into the plan's row order)
            const double* d = vican_facade_tiles_rows_in(P, deg_t, 1, P->t_rows9, stream);
            hipMemcpyAsync(P->row_sum_w, d, (size_t)n_time * 8, hipMemcpyDeviceToDevice, s);
        }
        if (rc >= 0 && P->have_t && deg_c) hipMemcpyAsync(P->cam_sum_w, deg_c, (size_t)n_cam * 8, hipMemcpyDeviceToDevice, s);
        if (rc >= 0 && hipStreamSynchronize(s) != hipSuccess) rc = ferr(VICAN_ERR_LAUNCH, "vican_plan_create: packing failed");
        if (rc < 0) return fail(rc);
    } else {   // the whole CG as one cooperative launch on capture-sized graphs (device.py HipBackend.cg_resident_ok, solver.py small_graph)
        int dev_ = 0, ncu_ = 0;
        hipGetDevice(&dev_); hipDeviceGetAttribute(&ncu_, hipDeviceAttributeMultiprocessorCount, dev_);
        P->cgres_ok = P->have_t && P->g.layout == VICAN_LAYOUT_WAVE && n_edges < 2000000 && P->g.n_chunk > 0 && P->g.n_wg <= std::min(ncu_, 128) &&
                      vican_cg_resident_lds_bytes(n_cam, P->g.max_rows, P->g.n_copy, P->rows_per_wg_max) <= vican_lds_limit_bytes();
        P->coop_ok = n_cam <= 8192;
    // pack: CSR order -> chunked slot order (one scratch array of slot indices)
    const size_t nslot = (size_t)std::max(1, P->g.n_chunk) * P->g.slots;
    if (hipMalloc((void**)&perm, nslot * 4) != hipSuccess) return fail(ferr(VICAN_ERR_LAUNCH, "vican_plan_create: hipMalloc failed"));
    double* mx = P->ws ? P->ws : P->G;           // three doubles of scratch (zeroed by the memset)
    rc = vican_pack_edges(&P->g, row_ptr, col, blk, a, w, u, v, P->a, P->w, P->u, P->v, perm, stream);
    if (rc >= 0 && P->idx16) {
        rc = vican_pack_idx16(&P->g, P->idx16, stream);
        if (rc >= 0) P->g.idx16 = P->idx16;
    }
    // one-row wave graphs with 4 edges per lane: the CG weights as float32 where every one of them is a float32 value
    const bool try_w32 = rc >= 0 && P->have_t && P->g.layout == VICAN_LAYOUT_WAVE && P->g.slots == 256 && P->g.n_chunk == n_time;
    if (try_w32) rc = vican_pack_w32(&P->g, P->w, P->w32, P->w32_flag, stream);
    if (rc >= 0) {
        hipLaunchKernelGGL(facade_maxima_kernel, dim3(256), dim3(256), 0, s, (long long)n_edges, storage, a, w, u, v, mx);
        double h[3] = {1, 1, 1};
        int32_t inexact = 1;
        if (try_w32 && hipMemcpyAsync(&inexact, P->w32_flag, 4, hipMemcpyDeviceToHost, s) != hipSuccess) rc = ferr(VICAN_ERR_LAUNCH, "vican_plan_create: flag read failed");
        if (hipMemcpyAsync(h, mx, 24, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) rc = ferr(VICAN_ERR_LAUNCH, "vican_plan_create: maxima read failed");
        if (rc >= 0 && try_w32 && inexact == 0) { P->g.w32 = P->w32; P->g.w32_src = P->w; }
        if (rc >= 0) {
            hipMemsetAsync(mx, 0, 24, s);
            if (P->have_t) { P->wmax = h[1]; P->gmax = h[2]; }
            void* cam_ws = P->zpart;             // C 64-bit words of scratch
            rc = vican_edge_sums(&P->g, P->a, storage == VICAN_STORE_F64, h[0] > 0 ? h[0] : 1.0, P->row_sum_a, P->cam_sum_a, cam_ws, stream);
            if (rc >= 0) rc = vican_block_norms(&P->g, P->rnorm, P->fx, stream);
            if (rc >= 0 && P->have_t) {
                rc = vican_edge_sums(&P->g, P->w, 1, P->wmax, P->row_sum_w, P->cam_sum_w, cam_ws, stream);
                if (rc >= 0 && deg_t) hipMemcpyAsync(P->row_sum_w, deg_t, (size_t)n_time * 8, hipMemcpyDeviceToDevice, s);
                if (rc >= 0 && deg_c) hipMemcpyAsync(P->cam_sum_w, deg_c, (size_t)n_cam * 8, hipMemcpyDeviceToDevice, s);
            }
        }
    }
    }   // (untiled)
    // |L| <~ 2 max camera degree: sizes the pivot floor of the Cholesky-QR (solver.py: RotationSolver.init)
    std::vector<double> cs((size_t)n_cam);
    if (rc >= 0 && (hipMemcpyAsync(cs.data(), P->cam_sum_a, cs.size() * 8, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess))
        rc = ferr(VICAN_ERR_LAUNCH, "vican_plan_create: degree read failed");
    hipFree(perm);
    if (rc < 0) return fail(rc);
    P->lscale = *std::max_element(cs.begin(), cs.end());
    // propagation sweeps of the start block (solver.py: as many as it takes to reach nearly every camera)
    const double hops1 = ((double)n_edges / n_cam) * std::max((double)n_edges / n_time - 1.0, 0.0) / n_cam;
    P->prop_sweeps = hops1 >= 4.0 ? 1 : (hops1 >= 0.5 ? 2 : 3);
    *plan_out = P;
    return VICAN_OK;
}

extern "C" int vican_plan_destroy(vican_plan_t* P) {
    if (!P) return VICAN_OK;
    if (P->ar.base) hipFree(P->ar.base);
    if (P->lsqr_base) hipFree(P->lsqr_base);
    if (P->status_host) hipHostFree(P->status_host);
    delete P;
    return VICAN_OK;
}

extern "C" int vican_plan_describe(const vican_plan_t* P, vican_graph_t* g_out) {
    if (!P || !g_out) return ferr(VICAN_ERR_ARG, "vican_plan_describe: bad argument");
    *g_out = P->g;
    return VICAN_OK;
}

// One rank of a timestep-sharded solve: the plan was created from THIS rank's rows (row_ptr / col / blk ... of its slice of the
// timesteps, all C cameras; deg_c: this rank's share of the diagonal - the whole vector on one rank, zeros on the others - or NULL:
// the local camera sums).  From here on vican_solve_rot / vican_solve_trans all-reduce the camera-side partials through `comm`
// (vican_comm_*: peer exchange or RCCL) in stream order; every rank must make the same calls.  One collective here: the graph
// constants that must be GLOBAL (camera degrees -> pivot floor, edge and row counts -> propagation sweeps).  comm NULL: back to
// a single-rank plan.
extern "C" int vican_plan_set_comm(vican_plan_t* P, vican_comm_t* comm, void* stream) {
    if (!P) return ferr(VICAN_ERR_ARG, "vican_plan_set_comm: NULL plan");
    P->comm = comm; P->comm_ready = false;
    if (!comm) return VICAN_OK;
    hipStream_t s = (hipStream_t)stream;
    const int C = P->C;
    std::vector<double> h((size_t)C + 2);
    HIPCK(hipMemcpyAsync(P->setup_msg, P->cam_sum_a, (size_t)C * 8, hipMemcpyDeviceToDevice, s), "vican_plan_set_comm");
    const double cnt[2] = {(double)P->E, (double)P->T};
    HIPCK(hipMemcpyAsync(P->setup_msg + C, cnt, 16, hipMemcpyHostToDevice, s), "vican_plan_set_comm");
    HIPCK(hipStreamSynchronize(s), "vican_plan_set_comm");          // (cnt is a stack object)
    CK(vican_comm_allreduce_sum(comm, P->setup_msg, C + 2, stream));
    HIPCK(hipMemcpyAsync(h.data(), P->setup_msg, h.size() * 8, hipMemcpyDeviceToHost, s), "vican_plan_set_comm");
    HIPCK(hipStreamSynchronize(s), "vican_plan_set_comm");
    P->lscale = *std::max_element(h.begin(), h.begin() + C);
    const double n_e = h[C], n_t = std::max(h[C + 1], 1.0);
    P->e_global = n_e;
    const double hops1 = (n_e / C) * std::max(n_e / n_t - 1.0, 0.0) / C;
    P->prop_sweeps = hops1 >= 4.0 ? 1 : (hops1 >= 0.5 ? 2 : 3);
    for (int& v : P->pred_steps) v = 0;
    for (int& v : P->pred_fail) v = 0;
    for (bool& v : P->probe_done) v = false;
    P->comm_ready = true;
    return VICAN_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// rotation stage: reference large_bipartite_so3sync, bipgo.py:279-348 (schedule of vican_amd/solver.py RotationSolver, plain form)
// ---------------------------------------------------------------------------------------------------------------------------
extern "C" int vican_solve_rot(vican_plan_t* P, int32_t maxiter, double eig_tol, double* rc_out, double* Rt_out,
                               vican_solve_info_t* info, void* stream) {
    if (!P || maxiter < 1 || !(eig_tol > 0)) return ferr(VICAN_ERR_ARG, "vican_solve_rot: bad argument");
    hipStream_t s = (hipStream_t)stream;
    const int C = P->C, T = P->T, n = 3 * C, ld = P->ld, m_max = std::max(1, std::min(M_MAX, n / 3));
    const vican_graph_t* g = &P->g;
    vican_set_gate(nullptr);
    // (one rank of a sharded solve: the GLOBAL edge count, the same on every rank - the check schedule decides when collectives run)
    const bool small = (P->comm_ready && P->comm ? P->e_global : (double)P->E) < 2000000.0;
    const int min_steps = small ? 8 : 4, warm_min = small ? 4 : 2, check_every = small ? 4 : 2, max_restarts = 20;
    const double floor_tol = P->storage == VICAN_STORE_F64 ? 1e-13 : 1e-7, pivot_floor = (1e-12 * P->lscale) * (1e-12 * P->lscale);
    vican_solve_info_t inf{};
    vican_comm_t* comm = P->comm_ready ? P->comm : nullptr;       // (one rank of a sharded solve: camera partials all-reduced in stream order)
    const bool tiled = !P->tiles.empty();
    auto op_z = [&](const double* x, double* z) {
        if (!tiled) return vican_block_op_z_comm(g, P->lamT, x, P->zpart, P->fx, z, comm, stream);
        const int rc_ = vican_facade_tiles_op_z(P, x, z, stream);
        return rc_ < 0 || !comm ? rc_ : vican_comm_allreduce_sum(comm, z, 9LL * P->C, stream);
    };
    // duals, Lambda_C = (weighted camera degree) I  (bipgo.py:271-276)
    CK(vican_init_duals(T, P->row_sum_a, P->rnorm, P->lamT, tiled ? P->tiles[0].fx : P->fx, stream));
    CK(fx_finish(P, stream));
    HIPCK(hipMemcpyAsync(P->cam_deg, P->cam_sum_a, (size_t)C * 8, hipMemcpyDeviceToDevice, s), "vican_solve_rot");
    if (comm) CK(vican_comm_allreduce_sum(comm, P->cam_deg, C, stream));
    CK(vican_scaled_identity(C, P->cam_deg, P->lamC, stream));
    // start block: rotations propagated from the gauge camera through the power graph (solver.py: _propagated_start)
    hipLaunchKernelGGL(facade_seed_kernel, dim3((3 * n + 255) / 256), dim3(256), 0, s, n, P->x0);
    for (int k = 0; k < P->prop_sweeps; ++k) {
        CK(op_z(P->x0, P->z));
        CK(vican_polar_dual(C, P->z, P->x0, nullptr, 0, stream));
        ++inf.sweeps;
    }
    bool z_ready = false, have5 = false;
    for (int it = 0; it < maxiter; ++it) {
        // bipgo.py:283: the reference leaves its loop once all five returned eigenvalues are <= 1e-6 in magnitude (>= 5 near-null
        // vectors: graphs with several components) - same exit as solver.py RotationSolver.run takes on `small5`
        if (have5) {
            bool all_small = true;
            for (int q = 0; q < 5; ++q) all_small = all_small && std::isfinite(inf.evals[q]) && std::fabs(inf.evals[q]) <= 1e-6;
            if (all_small) break;
        }
        const int relax = std::max(0, (maxiter - 2) - it);
        const double tol = std::min(std::max(eig_tol * std::pow(100.0, relax), eig_tol), 1e-4);
        const bool last = it == maxiter - 1;
        const double* start = it == 0 ? P->x0 : P->rc;
        bool conv = false;
        int steps = 0;
        double* st = P->status_host;
        for (int restart = 0; restart <= max_restarts && !conv; ++restart) {
            const bool have_z = restart == 0 && z_ready;
            CK(vican_lanczos_seed(n, start, P->V, ld, P->beta0, P->xrow, have_z ? P->zraw : nullptr, have_z ? P->z : nullptr, P->coop_sync, stream));
            z_ready = false;
            steps = 0;
            // (the same graph solved before - time series, repeated solves of one plan: straight to the step count that sufficed
            //  last time instead of paying for checks known to fail; schedule only, every step is executed - solver.py pred_steps)
            const int remembered = restart == 0 && it < 64 ? P->pred_steps[it] : 0;
            int next_check = std::min(remembered > 0 ? remembered : (restart == 0 ? warm_min : min_steps), m_max), prev_steps = 0;
            // capture-sized graphs are checked every four steps, so the remembered count is a multiple of that spacing that sufficed, not
            // the smallest count: later solves of the plan try ONE STEP FEWER each until a first check fails (that solve takes the step
            // back and checks again - one Ritz call more, once); every solve still ends on a passed check (solver.py, spectral)
            bool probing = false;
            if (small && remembered > 1 && it < 64 && !P->probe_done[it]) { --next_check; probing = true; }
            // residual level of the rounding floor: remembered per iteration index, else the one met by the previous iteration (a
            // property of the f32 products, not of the iterate) - lets the FIRST check recognise a stalled residual (solver.py)
            const double level = restart == 0 && it < 64 ? (P->floor_level[it] >= 0.0 ? P->floor_level[it] : (it > 0 ? P->floor_level[it - 1] : -1.0)) : -1.0;
            double prev_res = -1.0;
            int floor_at = 0;
            bool first = true;
            for (;;) {
                const int j = steps;
                // camera side of the step as ONE cooperative launch (vican_lanczos_cam_coop; with few slabs it folds them itself:
                // the sweep's result stays in fixed point) - the launch sequence (7 kernels per step) where that grid is refused
                const bool sweep = !(j == 0 && have_z), from_slabs = sweep && !comm && !tiled && P->coop_ok && g->n_wg <= 64;
                if (sweep) {
                    if (from_slabs) CK(vican_block_op(g, P->lamT, P->xrow, P->zpart, P->fx, stream));
                    else CK(op_z(P->xrow, P->z));
                    ++inf.sweeps;
                }
                bool done_step = false;
                if (P->coop_ok) {
                    const int rc_ = vican_lanczos_cam_coop(C, P->lamC, P->V, ld, j, P->z, P->coop_ws, P->HB + (size_t)j * P->hb_stride,
                                                           P->HB + (size_t)j * P->hb_stride + P->hw, P->xrow, pivot_floor, (uint32_t*)P->coop_sync,
                                                           from_slabs ? P->zpart : nullptr, from_slabs ? g->n_wg : 0, from_slabs ? P->fx + 3 : nullptr,
                                                           from_slabs ? P->fx + 7 : nullptr, !tiled && g->n_wg <= 64 ? 1 : 0, stream);
                    if (rc_ == VICAN_ERR_CAPACITY) {
                        P->coop_ok = false;
                        if (from_slabs) CK(vican_slab_reduce_fx(P->zpart, g->n_wg, C, 9, 1.0, P->fx + 3, P->fx + 7, P->z, stream));
                    } else { CK(rc_); done_step = true; }
                }
                if (!done_step)
                    CK(vican_lanczos_cam_step(C, P->lamC, P->V, ld, j, P->z, P->R, P->H, P->G, P->HB + (size_t)j * P->hb_stride,
                                              P->HB + (size_t)j * P->hb_stride + P->hw, P->xrow, pivot_floor, nullptr, 0, stream));
                ++steps; ++inf.lanczos_steps;
                if (steps < next_check && steps < m_max) continue;
                const int flags = (first ? 1 : 0) | (steps >= m_max ? 2 : 0);
                CK(vican_ritz(P->HB, P->hb_stride, P->hw, steps, flags, tol, floor_tol, level, steps - prev_steps <= 1 ? 0.5 : 0.25, P->Yd,
                              P->status, P->gate, stream));
                first = false;
                HIPCK(hipMemcpyAsync(st, P->status, 16 * 8, hipMemcpyDeviceToHost, s), "vican_solve_rot");
                HIPCK(hipStreamSynchronize(s), "vican_solve_rot");
                conv = st[3] != 0.0;
                const bool floor_hit = st[4] != 0.0;
                if (floor_hit && restart == 0 && it < 64 && !(remembered > 0 && steps > remembered)) {     // where the floor was first reached (the earlier of the two checks), its level
                    P->floor_level[it] = prev_res >= 0.0 ? std::max(st[0], prev_res) : std::max(st[0], P->floor_level[it]);
                    floor_at = (prev_res >= 0.0 && prev_res <= 2.0 * P->floor_level[it]) ? prev_steps : steps;
                }
                if (conv && restart == 0 && it < 64) {
                    // (a larger count than the remembered one is adopted only when the first check at the remembered count failed
                    //  in two solves in a row: about one solve in a few hundred runs into a stalled residual and takes 3-6 extra
                    //  steps - vican_amd/solver.py, RotationSolver.spectral)
                    const int want = floor_hit && floor_at > 0 ? floor_at : steps, have = P->pred_steps[it];
                    if (have <= 0 || want <= have) { P->pred_steps[it] = want; P->pred_fail[it] = 0; }
                    else if (++P->pred_fail[it] >= 2) { P->pred_steps[it] = want; P->pred_fail[it] = 0; P->probe_done[it] = true; }
                }
                prev_res = st[0]; prev_steps = steps;
                if (probing) {                                 // (the first check of this run was the probe)
                    probing = false;
                    if (!(st[2] != 0.0 && conv)) {
                        P->probe_done[it] = true;              // one step fewer does not do: the remembered count is the smallest
                        if (st[2] == 0.0) { next_check = std::min(steps + 1, m_max); continue; }
                    }
                }
                if (st[2] != 0.0) break;                       // stop (converged, noise floor, step budget or exhausted Krylov space)
                const bool near_floor = floor_tol > 1e-12 && st[0] <= floor_tol;
                next_check = std::min(steps + ((steps < 8 && !small) || near_floor || remembered > 0 ? 1 : check_every), m_max);
            }
            CK(vican_tall_combine(n, P->V, ld, 3 * steps, P->Yd, P->X, stream));
            if (!conv) { start = P->X; ++inf.restarts; HIPCK(hipMemcpyAsync(P->Xp, P->X, (size_t)3 * n * 8, hipMemcpyDeviceToDevice, s), "vican_solve_rot"); start = P->Xp; }
        }
        for (int q = 0; q < 3; ++q) inf.evals[q] = st[7 + q];
        inf.evals[3] = st[15]; inf.evals[4] = st[13];
        have5 = true;
        inf.eig_resid = st[0];
        // X = V3 V3[0:3]^-1, per-camera projection, Y = P X, camera duals, timestep duals (bipgo.py:295-334)
        CK(vican_gauge_project(C, P->X, P->Xp, stream));
        CK(op_z(P->Xp, P->z));
        CK(vican_polar_dual(C, P->z, P->rc, P->lamC, 1, stream));
        if (tiled) CK(vican_facade_tiles_dual_update(P, P->rc, stream));      // (no fused dual update on tiles: the next solve sweeps for its z)
        else if (!last) {
            CK(vican_dual_update_op(g, P->rc, P->Rt, P->lamT, P->rnorm, P->fx, P->zpart, P->zraw, stream));
            if (comm) CK(vican_comm_allreduce_sum(comm, P->zraw, 9LL * C, stream));
            z_ready = true;
        }
        else CK(vican_dual_update(g, P->rc, P->Rt, P->lamT, P->rnorm, P->fx, stream));
        CK(fx_finish(P, stream));
        inf.sweeps += 2;
        ++inf.iterations;
    }
    if (rc_out) HIPCK(hipMemcpyAsync(rc_out, P->rc, (size_t)3 * n * 8, hipMemcpyDeviceToDevice, s), "vican_solve_rot");
    if (Rt_out && tiled) vican_facade_tiles_rows_out(P, P->Rt, 9, Rt_out, stream);     // (the caller's row order)
    else if (Rt_out) HIPCK(hipMemcpyAsync(Rt_out, P->Rt, (size_t)9 * T * 8, hipMemcpyDeviceToDevice, s), "vican_solve_rot");
    HIPCK(hipStreamSynchronize(s), "vican_solve_rot");
    if (info) *info = inf;
    return VICAN_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// translation stage: J^T b and scipy's CG on the normal equations (bipgo.py:445-478)
// ---------------------------------------------------------------------------------------------------------------------------
extern "C" int vican_solve_trans(vican_plan_t* P, const double* rc, const double* Rt, double rtol, int64_t maxiter, double* x_c,
                                 double* x_t, vican_solve_info_t* info, void* stream) {
    if (!P || !rc || !Rt || !x_c || !x_t || !(rtol >= 0)) return ferr(VICAN_ERR_ARG, "vican_solve_trans: bad argument");
    if (!P->have_t) return ferr(VICAN_ERR_ARG, "vican_solve_trans: the plan was created without translation arrays (w, u, v)");
    hipStream_t s = (hipStream_t)stream;
    const int C = P->C, T = P->T;
    if (maxiter <= 0) maxiter = 10LL * 3 * (C + T);              // scipy's default
    vican_solve_info_t inf = info ? *info : vican_solve_info_t{};
    vican_comm_t* comm = P->comm_ready ? P->comm : nullptr;
    const bool tiled = !P->tiles.empty();
    // a tiled plan may keep its rows in an order of its own: Rt comes in, x_t goes out through it (the CG runs on the plan's q_t-sized
    // buffer t_rows9, which is free again once J^T b is formed)
    double* const x_t_caller = x_t;
    const bool reorder = tiled && P->t_perm_dev != nullptr;
    if (tiled) {
        CK(vican_facade_tiles_rhs(P, rc, vican_facade_tiles_rows_in(P, Rt, 9, P->t_rows9, stream), stream));
        if (reorder) x_t = P->t_rows9;
    }
    else CK(vican_trans_rhs(&P->g, P->u, P->v, rc, Rt, P->b_t, P->b_c, P->zpart, P->gmax, P->n_add, stream));
    const double* deg_c = P->cam_sum_w;
    if (comm) {
        // the camera side of the system is a sum over all ranks' rows: right-hand side and diagonal (the caller's deg_c was this
        // rank's SHARE); the reduced diagonal in its own buffer - the plan's stays this rank's share for the next solve
        CK(vican_comm_allreduce_sum(comm, P->b_c, 3LL * C, stream));
        HIPCK(hipMemcpyAsync(P->setup_msg, P->cam_sum_w, (size_t)C * 8, hipMemcpyDeviceToDevice, s), "vican_solve_trans");
        CK(vican_comm_allreduce_sum(comm, P->setup_msg, C, stream));
        deg_c = P->setup_msg;
    }
    vican_cg_state_t h{};
    if (P->cgres_ok && !comm) {
        // capture-sized graphs: the whole solve as ONE cooperative launch (vican_cg_resident; an iteration of the launch sequence is
        // three dependent launches of a few microseconds each there)
        const int rc_ = vican_cg_resident(&P->g, P->w, P->row_sum_w, P->cam_sum_w, P->b_c, P->b_t, x_c, x_t, P->zpart, P->cgres_ws, rtol,
                                          (int32_t)std::min<int64_t>(maxiter, 2147483647LL), P->n_add_cg, P->wmax, P->rows_per_wg_max, P->st, stream);
        if (rc_ == VICAN_ERR_CAPACITY) P->cgres_ok = false;
        else {
            CK(rc_);
            HIPCK(hipMemcpyAsync(P->status_host + 16, P->st, sizeof(vican_cg_state_t), hipMemcpyDeviceToHost, s), "vican_solve_trans");
            HIPCK(hipStreamSynchronize(s), "vican_solve_trans");
            std::memcpy(&h, P->status_host + 16, sizeof(h));
            if (h.done == -1) P->cgres_ok = false;          // a grid barrier was not passed in time (shared device): the launch sequence solves it
            else {
                inf.cg_iters = h.iter; inf.cg_converged = h.done == 1;
                inf.cg_relres = h.bnorm2 > 0 ? std::sqrt(h.rho / h.bnorm2) : 0.0;
                if (info) *info = inf;
                if (h.done != 1) return ferr(VICAN_ERR_LAUNCH, "vican_solve_trans: CG did not converge in %lld iterations (scipy exit_code != 0, bipgo.py:478)", (long long)maxiter);
                return VICAN_OK;
            }
        }
    }
    CK(vican_cg_init(C, T, P->b_c, P->b_t, x_c, x_t, P->r_c, P->r_t, P->p_c, P->p_t, P->st, P->ws, P->wmax, stream));
    if (comm) CK(vican_comm_allreduce_sum(comm, &P->st->rr_time, 1, stream));
    long long launched = 0;
    int burst = 8, n_part = 0;
    for (;;) {
        // (scipy: `for iteration in range(maxiter)` - at most maxiter updates of x; no test behind the last one)
        for (int i = 0; i < burst && launched < maxiter; ++i, ++launched) {
            if (tiled) {
                // camera tiles: the product tile by tile, then the step (vican_amd/solver.py, the two-call iteration; sharded: its two
                // messages [q_c | p_t.q_t] and r_t.r_t)
                CK(vican_facade_tiles_cg_local(P, rtol, n_part, stream));
                if (comm) CK(vican_comm_allreduce_sum(comm, P->qcpq, 3LL * C + 1, stream));
                n_part = vican_cg_iter_finish(C, T, deg_c, P->qcpq, P->p_c, x_c, P->r_c, P->p_t, P->q_t, x_t, P->r_t, P->rr_part, 1536, P->st, stream);
                if (n_part < 0) return n_part;
                if (comm) {
                    CK(vican_cg_end(P->rr_part, n_part, P->st, stream));
                    CK(vican_comm_allreduce_sum(comm, &P->st->rr_time, 1, stream));
                    n_part = 0;
                }
            } else if (comm)
                CK(vican_cg_iter_comm(&P->g, P->w, P->row_sum_w, deg_c, P->r_c, P->p_c, x_c, P->r_t, P->p_t, P->q_t, x_t, P->zpart, P->pq_part,
                                      P->msg, rtol, P->rr_part, 1536, P->n_add_cg, launched == 0, P->st, comm, stream));
            else
                CK(vican_cg_iter_fused(&P->g, P->w, P->row_sum_w, P->cam_sum_w, P->r_c, P->p_c, x_c, P->r_t, P->p_t, P->q_t, x_t, P->zpart,
                                       P->pq_part, P->qcpq, rtol, P->rr_part, 1536, P->n_add_cg, launched == 0, P->st, P->cg_ticket, stream));
        }
        HIPCK(hipMemcpyAsync(P->status_host + 16, P->st, sizeof(vican_cg_state_t), hipMemcpyDeviceToHost, s), "vican_solve_trans");
        HIPCK(hipStreamSynchronize(s), "vican_solve_trans");
        std::memcpy(&h, P->status_host + 16, sizeof(h));
        if (h.done || launched >= maxiter) break;
        burst = std::min(2 * burst, 64);
    }
    if (reorder) {                                           // x_t back in the caller's row order
        vican_facade_tiles_rows_out(P, x_t, 3, x_t_caller, stream);
        HIPCK(hipStreamSynchronize(s), "vican_solve_trans");
    }
    inf.cg_iters = h.done == 1 ? h.iter : (int32_t)std::min<long long>(launched, 2147483647LL); inf.cg_converged = h.done == 1;
    inf.cg_relres = h.bnorm2 > 0 ? std::sqrt(h.rho / h.bnorm2) : 0.0;
    if (info) *info = inf;
    if (h.done != 1) return ferr(VICAN_ERR_LAUNCH, "vican_solve_trans: CG did not converge in %lld iterations (scipy exit_code != 0, bipgo.py:478)", (long long)maxiter);
    return VICAN_OK;
}

// ---------------------------------------------------------------------------------------------------------------------------
// translation stage, lsqr_solver="direct": scipy.sparse.linalg.lsqr on the incidence system (bipgo.py:479-480) - the schedule
// of vican_amd/solver.py LsqrTranslationSolver._solve_device on one rank: every scalar of scipy's loop in a device
// vican_lsqr_state_t, ONE fused pass over the edges per iteration (vican_lsqr_step), the host polls `done` between bursts.
// The iteration runs on the MERGED system J~ p = b~ (same normal equations, same iterates); bnorm2_true = |b|^2 of the
// reference's un-merged right-hand side lets the residual norms - and with them scipy's stopping tests - be the reference's
// (<= 0: the merged system's own |b~|^2, exact when no (camera, timestep) pair carries more than one marker).
// ---------------------------------------------------------------------------------------------------------------------------
namespace {
int lsqr_workspace(vican_plan* P) {
    if (P->lsqr_base) return VICAN_OK;
    const bool tiled = !P->tiles.empty();
    const size_t nslot = tiled ? 0 : (size_t)std::max(1, P->g.n_chunk) * P->g.slots, C = P->C, T1 = std::max(P->T, 1);
    Arena A;
    auto carve = [&]() {
        A.used = 0;
        P->lu = A.take<double>(3 * nslot); P->lsw = A.take<double>(nslot); P->lpart = A.take<double>(std::max<size_t>(P->g.n_wg, 1024));
        P->lslab = A.take<double>(tiled ? 1 : (size_t)std::max(P->g.n_wg, 1) * 6 * C);
        for (vican_tile_plan& t : P->tiles) {                 // camera tiles: every tile keeps its own edge vector u~ (tiled.py lsqr_*)
            const size_t ns = (size_t)t.g.n_chunk * t.g.slots;
            t.lu = A.take<double>(3 * ns); t.lsw = A.take<double>(ns); t.lpart = A.take<double>(std::max<size_t>(t.g.n_wg, 1024));
            t.lslab = A.take<double>((size_t)t.g.n_wg * 6 * (t.c1 - t.c0));
        }
        P->ltmp = A.take<double>(P->tiles.size() + 1);
        P->lv_c = A.take<double>(3 * C); P->lw_c = A.take<double>(3 * C); P->lv_t = A.take<double>(3 * T1); P->lw_t = A.take<double>(3 * T1);
        P->lz_t = A.take<double>(3 * T1); P->lacc = A.take<double>(3 * C + 2); P->lpart2 = A.take<double>(1025);
        P->lwp_c = A.take<double>(1024); P->lwp_t = A.take<double>(1024); P->ls2 = A.take<double>(4);
        P->lst = (vican_lsqr_state_t*)A.take<double>((sizeof(vican_lsqr_state_t) + 7) / 8);
        return A.used + 256;
    };
    const size_t bytes = carve();
    if (hipMalloc((void**)&P->lsqr_base, bytes) != hipSuccess) return ferr(VICAN_ERR_LAUNCH, "vican_solve_trans_lsqr: hipMalloc of %zu bytes failed", bytes);
    A.base = P->lsqr_base; A.size = bytes;
    carve();
    return VICAN_OK;
}
}  // namespace

extern "C" int vican_solve_trans_lsqr(vican_plan_t* P, const double* rc, const double* Rt, double bnorm2_true, double atol, double btol,
                                      double conlim, int64_t iter_lim, double* x_c, double* x_t, vican_lsqr_info_t* info, void* stream) {
    if (!P || !rc || !Rt || !x_c || !x_t || !(atol >= 0) || !(btol >= 0)) return ferr(VICAN_ERR_ARG, "vican_solve_trans_lsqr: bad argument");
    if (!P->have_t) return ferr(VICAN_ERR_ARG, "vican_solve_trans_lsqr: the plan was created without translation arrays (w, u, v)");
    CK(lsqr_workspace(P));
    hipStream_t s = (hipStream_t)stream;
    const int C = P->C, T = P->T, C3 = 3 * C;
    if (iter_lim <= 0) iter_lim = 2LL * 3 * (C + T);              // scipy's default: 2 n
    const double ctol = conlim > 0 ? 1.0 / conlim : 0.0;
    const double smax = std::sqrt(P->wmax), n_add = P->n_add;
    const int lo_bits = fix2_lo_bits(n_add);
    vican_lsqr_info_t inf{};
    auto zero = [&](double* p, size_t n) { return hipMemsetAsync(p, 0, n * 8, s) == hipSuccess; };
    if (!zero(P->lv_c, C3) || !zero(P->lw_c, C3) || !zero(x_c, C3) || !zero(P->lv_t, 3 * (size_t)T) || !zero(P->lw_t, 3 * (size_t)T) ||
        !zero(x_t, 3 * (size_t)T) || !zero(P->lz_t, 3 * (size_t)T) || !zero(P->lacc, C3 + 2) || !zero(P->lpart2, 1025) || !zero(P->ls2, 4))
        return ferr(VICAN_ERR_LAUNCH, "vican_solve_trans_lsqr: memset failed");
    auto read = [&](const double* dev, double* host, size_t n) {
        if (hipMemcpyAsync(P->status_host + 32, dev, n * 8, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return false;
        std::memcpy(host, P->status_host + 32, n * 8);
        return true;
    };
    // u~_1 = b~ (unnormalised), beta_1 = |b~|
    // camera tiles: every pass over the edges tile by tile - a tile's sweep yields its row sums (added in tile order), the complete
    // camera sums of its own cameras and its part of |u^|^2 (vican_amd/tiled.py lsqr_init_u / lsqr_step); per-row arguments in the
    // plan's row order
    const bool tiled = !P->tiles.empty(), reorder = tiled && P->t_perm_dev != nullptr;
    const int nt = (int)P->tiles.size();
    const size_t T1 = std::max(T, 1);
    double* const x_t_caller = x_t;
    auto sum_parts = [&](double* out) { return vican_sum_apply3(1, 1, nullptr, P->ltmp, nt, 1, out, stream); };
    auto edge_step = [&]() -> int {
        if (!tiled) return vican_lsqr_step(&P->g, P->lsw, P->lu, P->lv_c, P->lv_t, P->lz_t, P->lslab, P->lpart, P->lacc, P->lst, stream);
        for (int k = 0; k < nt; ++k) {
            const vican_tile_plan& t = P->tiles[k];
            // (the tile writes its camera sums into its slice of lacc and its part of |u^|^2 right behind the slice)
            CK(vican_lsqr_step(&t.g, t.lsw, t.lu, P->lv_c + (size_t)3 * t.c0, P->lv_t, P->t_acc + (size_t)k * T1 * 3, t.lslab, t.lpart,
                               P->lacc + (size_t)3 * t.c0, P->lst, stream));
            HIPCK(hipMemcpyAsync(P->ltmp + k, P->lacc + (size_t)3 * t.c1, 8, hipMemcpyDeviceToDevice, s), "vican_solve_trans_lsqr");
        }
        CK(sum_parts(P->lacc + C3));
        return vican_sum_apply3(T, 3, nullptr, P->t_acc, nt, (int64_t)T1 * 3, P->lz_t, stream);
    };
    if (tiled) {
        const double* Rt_plan = vican_facade_tiles_rows_in(P, Rt, 9, P->t_rows9, stream);
        for (int k = 0; k < nt; ++k) {
            const vican_tile_plan& t = P->tiles[k];
            CK(vican_lsqr_init_u(&t.g, t.w, t.u, t.v, rc + (size_t)9 * t.c0, Rt_plan, t.lu, t.lsw, t.lpart, P->ltmp + k, stream));
        }
        CK(sum_parts(P->ls2));
        if (reorder) {                                          // (t_rows9 is free again: the iteration's x_t in the plan's row order)
            x_t = P->t_rows9;
            if (!zero(x_t, 3 * (size_t)T)) return ferr(VICAN_ERR_LAUNCH, "vican_solve_trans_lsqr: memset failed");
        }
    } else
        CK(vican_lsqr_init_u(&P->g, P->w, P->u, P->v, rc, Rt, P->lu, P->lsw, P->lpart, P->ls2, stream));
    double h4[4];
    if (!read(P->ls2, h4, 1)) return ferr(VICAN_ERR_LAUNCH, "vican_solve_trans_lsqr: read failed");
    const double beta = std::sqrt(h4[0]);
    const double c2 = bnorm2_true > 0 ? std::max(bnorm2_true - beta * beta, 0.0) : 0.0;
    const double bnorm = std::sqrt(beta * beta + c2);
    if (info) *info = inf;
    if (beta == 0.0) return VICAN_OK;                              // x = 0 (scipy returns at once)
    // first bidiagonalisation step alfa_1 v_1 = J~^T u_1: the fused pass with coef = -1 on v = 0
    vican_lsqr_state_t st{};
    double inv;
    st.coef = -1.0; st.smax = smax; st.n_add = n_add; st.lo_bits = lo_bits;
    st.qscale = fix_scale(smax * beta, n_add, &inv, 49); st.qinv = inv;
    HIPCK(hipMemcpyAsync(P->lst, &st, sizeof(st), hipMemcpyHostToDevice, s), "vican_solve_trans_lsqr");
    CK(edge_step());
    int nb = vican_lsqr_nodes(C, T, P->lz_t, P->lacc, P->lv_t, P->lv_c, P->lpart2, P->lst, stream);
    if (nb < 0) return nb;
    std::vector<double> hp(1025);
    if (!read(P->lpart2, hp.data(), 1025)) return ferr(VICAN_ERR_LAUNCH, "vican_solve_trans_lsqr: read failed");
    double nv2 = 0.0;
    for (int i = 0; i < nb; ++i) nv2 += hp[i];
    const double alfa = std::sqrt(nv2 + hp[1024]);
    if (alfa == 0.0) return VICAN_OK;
    // v_1 = v / alfa, w_1 = v_1, x = 0
    CK(vican_lsqr_update(C3, 1.0 / alfa, 0.0, 0.0, P->lv_c, P->lw_c, x_c, P->lpart, P->ls2 + 3, stream));
    CK(vican_lsqr_update(3LL * T, 1.0 / alfa, 0.0, 0.0, P->lv_t, P->lw_t, x_t, P->lpart, P->ls2 + 1, stream));
    st = vican_lsqr_state_t{};
    st.alfa = alfa; st.beta = beta; st.rhobar = alfa; st.phibar = beta; st.cs2 = -1.0; st.c2 = c2; st.bnorm = bnorm; st.atol = atol; st.btol = btol;
    st.ctol = ctol; st.coef = alfa / beta; st.smax = smax; st.n_add = n_add; st.lo_bits = lo_bits;
    st.qscale = fix_scale(smax * (2.0 * smax + alfa), n_add, &inv, 49); st.qinv = inv;
    st.iter_lim = (int32_t)std::min<long long>(iter_lim, 2147483647LL);
    HIPCK(hipMemcpyAsync(P->lst, &st, sizeof(st), hipMemcpyHostToDevice, s), "vican_solve_trans_lsqr");
    HIPCK(hipStreamSynchronize(s), "vican_solve_trans_lsqr");     // (st is a stack object)
    const double* wpart_t = P->ls2 + 1; const double* wpart_c = P->ls2 + 3;
    int n_wt = 1, n_wc = 1, burst = 8;
    long long launched = 0;
    vican_lsqr_state_t hst{};
    for (;;) {
        const long long todo = std::min<long long>(burst, std::max<long long>(iter_lim - launched, 1));
        for (long long i = 0; i < todo; ++i, ++launched) {
            CK(edge_step());
            nb = vican_lsqr_nodes(C, T, P->lz_t, P->lacc, P->lv_t, P->lv_c, P->lpart2, P->lst, stream);
            if (nb < 0) return nb;
            CK(vican_lsqr_scalars(C, P->lacc, P->lpart2, nb, nullptr, wpart_t, n_wt, wpart_c, n_wc, nullptr, P->lst, stream));
            n_wc = vican_lsqr_update_st(C3, P->lv_c, P->lw_c, x_c, P->lwp_c, 0, P->lst, stream);
            if (n_wc < 0) return n_wc;
            n_wt = vican_lsqr_update_st(3LL * T, P->lv_t, P->lw_t, x_t, P->lwp_t, 1, P->lst, stream);
            if (n_wt < 0) return n_wt;
            wpart_t = P->lwp_t; wpart_c = P->lwp_c;
        }
        HIPCK(hipMemcpyAsync(P->status_host + 32, P->lst, sizeof(hst), hipMemcpyDeviceToHost, s), "vican_solve_trans_lsqr");
        HIPCK(hipStreamSynchronize(s), "vican_solve_trans_lsqr");
        std::memcpy(&hst, P->status_host + 32, sizeof(hst));
        if (hst.done || launched >= iter_lim) break;
        burst = std::min(2 * burst, 64);
    }
    if (reorder) {                                           // x_t back in the caller's row order
        vican_facade_tiles_rows_out(P, x_t, 3, x_t_caller, stream);
        HIPCK(hipStreamSynchronize(s), "vican_solve_trans_lsqr");
    }
    inf.itn = hst.itn; inf.istop = hst.istop; inf.rnorm = hst.rnorm; inf.arnorm = hst.arnorm; inf.anorm = hst.anorm; inf.acond = hst.acond;
    inf.xnorm = hst.xnorm;
    if (info) *info = inf;
    return VICAN_OK;
}
