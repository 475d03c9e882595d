// vican_facade_impl.h - the plan object behind the four-call boundary (vican_facade.hip) and its camera-tiled half
// (vican_facade_tiles.hip).  Private to those two translation units.
#pragma once
#include <vector>
#include <algorithm>
#include <cmath>
#include <cstring>
#include <cstdarg>
#include <cstdio>
#include "vican_sweep_common.h"

namespace vican_facade {

inline int ferr(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));
inline int ferr(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_vican_err, sizeof(g_vican_err), fmt, ap);
    va_end(ap);
    return code;
}

constexpr double X_BOUND = 1.7320508075688772;          // |x_c|_F of every sweep input (device.py: X_BOUND)
constexpr size_t STREAM_NT_BYTES = (size_t)192 << 20;   // device.py: STREAM_NT_BYTES
constexpr int M_MAX = 32;                               // VICAN_RITZ_MAX_STEPS

struct Arena {
    unsigned char* base = nullptr;
    size_t size = 0, used = 0;
    template <typename T> T* take(size_t n) {
        used = (used + 255) & ~(size_t)255;
        T* p = base ? (T*)(base + used) : nullptr;
        used += n * sizeof(T);
        return p;
    }
};

inline int n_cu() {
    int dev = 0; hipGetDevice(&dev);
    hipDeviceProp_t p; if (hipGetDeviceProperties(&p, dev) != hipSuccess) return 256;
    return p.multiProcessorCount > 0 ? p.multiProcessorCount : 256;
}

}  // namespace vican_facade
using vican_facade::ferr; using vican_facade::Arena; using vican_facade::X_BOUND; using vican_facade::STREAM_NT_BYTES; using vican_facade::M_MAX;

// one camera tile of a plan with more cameras than one LDS table holds (vican_facade_tiles.hip; vican_amd/tiled.py TiledGraph)
struct vican_tile_plan {
    int c0 = 0, c1 = 0;                 // cameras [c0, c1)
    long long E = 0;
    std::vector<int32_t> rp;            // the tile's row_ptr (host; released after packing)
    vican_graph_t g{};                  // wave layout, chunking shared with the other tiles
    int n_copy = 1, wg_waves = 4, rows_per_wg_max = 1, rows_per_wg_sweep = 1;
    double n_add = 1;
    int32_t* idx = nullptr; uint16_t* idx16 = nullptr; void* blk = nullptr; void* a = nullptr;
    double *w = nullptr, *u = nullptr, *v = nullptr, *fx = nullptr, *zpart = nullptr, *zpart_f = nullptr;
    double *lu = nullptr, *lsw = nullptr, *lpart = nullptr, *lslab = nullptr;       // LSQR workspace of the tile (vican_solve_trans_lsqr)
};

struct vican_plan {
    int C = 0, T = 0, storage = 0, epl = 4;
    long long E = 0;
    double e_global = 0;                                    // merged edges of ALL ranks (vican_plan_set_comm)
    vican_graph_t g{};
    int rows_per_wg_max = 1, rows_per_wg_sweep = 1;
    double n_add = 1, n_add_cg = 1, wmax = 1, gmax = 1, lscale = 1;
    bool have_t = false;
    int prop_sweeps = 3;
    Arena ar;
    // layout arrays
    int32_t* idx = nullptr; int32_t* chunk_row0 = nullptr; void* blk = nullptr; void* a = nullptr; uint16_t* idx16 = nullptr;
    double *w = nullptr, *u = nullptr, *v = nullptr;
    // graph constants
    double *row_sum_a = nullptr, *cam_sum_a = nullptr, *rnorm = nullptr, *fx = nullptr, *row_sum_w = nullptr, *cam_sum_w = nullptr;
    // solver workspace (names as in vican_amd/solver.py)
    double *zpart = nullptr, *V = nullptr, *R = nullptr, *H = nullptr, *G = nullptr, *beta0 = nullptr, *HB = nullptr, *Yd = nullptr,
           *status = nullptr, *xrow = nullptr, *z = nullptr, *X = nullptr, *Xp = nullptr, *x0 = nullptr, *rc = nullptr, *lamC = nullptr,
           *cam_deg = nullptr, *lamT = nullptr, *Rt = nullptr, *zraw = nullptr;
    int32_t* gate = nullptr; int32_t* coop_sync = nullptr;
    double *coop_ws = nullptr, *cgres_ws = nullptr;        // workspaces of the cooperative camera-side step / of the resident CG
    float* w32 = nullptr; int32_t* w32_flag = nullptr;     // float32 copy of the CG weights (vican_graph_t.w32) where they are float32 values
    bool coop_ok = true, cgres_ok = false;                  // (dropped for the rest of the plan's life once a launch is refused)
    int pred_steps[64] = {0};                               // Lanczos steps that sufficed in primal-dual iteration `it` of the previous solve
    int pred_fail[64] = {0};
    bool probe_done[64] = {false};                          // ... and whether the remembered count is known to be the smallest (capture-sized graphs)                                // ... and consecutive solves whose first check at that count failed
    double floor_level[64];                                 // ... and the residual level its f32 rounding floor sat at (< 0: none met)
    int hw = 0, hb_stride = 0, ld = 0;
    // translation workspace
    double *b_c = nullptr, *b_t = nullptr, *r_c = nullptr, *p_c = nullptr, *r_t = nullptr, *p_t = nullptr, *q_t = nullptr, *qcpq = nullptr,
           *pq_part = nullptr, *rr_part = nullptr, *ws = nullptr;
    vican_cg_state_t* st = nullptr;
    uint32_t* cg_ticket = nullptr;
    double* status_host = nullptr;      // pinned
    // LSQR workspace (lsqr_solver="direct"): its own allocation, made by the first vican_solve_trans_lsqr (24 bytes per edge slot)
    // one rank of a timestep-sharded solve (vican_plan_set_comm): the communicator, the CG message, the GLOBAL graph sizes
    vican_comm_t* comm = nullptr;
    double* msg = nullptr;
    double* setup_msg = nullptr;
    bool comm_ready = false;
    // camera tiles (C > the tile width; vican_facade_tiles.hip): empty for an untiled plan
    std::vector<vican_tile_plan> tiles;
    std::vector<int32_t> t_chunks;                          // the shared chunking (host)
    std::vector<int32_t> t_perm;                            // t_perm[new] = old row where the plan keeps its rows in an order of its own (else empty)
    int32_t* t_perm_dev = nullptr;
    double* t_rows9 = nullptr;
    std::vector<vican_tile_t> t_host;                       // descriptors of the one-launch operator
    vican_tile_t* t_dev = nullptr;
    int32_t* t_chunk_row0 = nullptr;
    int tile_width = 0, nwgt = 1, t_parity = 0;
    bool fused_ok = false, tcg_ok = false;                  // (dropped for the rest of the plan's life once a launch is refused)
    double *t_rows = nullptr, *t_ypart = nullptr, *t_yp = nullptr, *t_wrow = nullptr, *t_acc = nullptr;
    unsigned char* lsqr_base = nullptr;
    double *lu = nullptr, *lsw = nullptr, *lpart = nullptr, *lslab = nullptr, *lv_c = nullptr, *lw_c = nullptr, *lv_t = nullptr, *lw_t = nullptr,
           *lz_t = nullptr, *lacc = nullptr, *lpart2 = nullptr, *lwp_c = nullptr, *lwp_t = nullptr, *ls2 = nullptr;
    vican_lsqr_state_t* lst = nullptr;
    double* ltmp = nullptr;                                 // per-tile parts of |u^|^2
};


// camera tiles (vican_facade_tiles.hip)
int vican_facade_tile_cams();
int vican_facade_tiles_layout(vican_plan* P, const int32_t* row_ptr, const int32_t* col, void* stream);
void vican_facade_tiles_carve(vican_plan* P);
int vican_facade_tiles_pack(vican_plan* P, const int32_t* row_ptr, const int32_t* col, const void* blk, const void* a, const double* w,
                            const double* u, const double* v, double amax, void* stream);
int vican_facade_tiles_refresh(vican_plan* P, void* stream);
int vican_facade_tiles_op_z(vican_plan* P, const double* x, double* z, void* stream);
int vican_facade_tiles_dual_update(vican_plan* P, const double* rc, void* stream);
int vican_facade_tiles_rhs(vican_plan* P, const double* rc, const double* Rt, void* stream);
int vican_facade_tiles_cg_local(vican_plan* P, double rtol, int n_part, void* stream);
const double* vican_facade_tiles_rows_in(vican_plan* P, const double* src, int width, double* scratch, void* stream);
void vican_facade_tiles_rows_out(vican_plan* P, const double* src, int width, double* dst, void* stream);
