"""The solver's kernel interface (``vican_amd.device.HipBackend``) in plain NumPy: the GPU-less backend.

BASELINE configs[0] ("cube_calib object_bipartite_se3sync ... on CPU (plumbing, no GPU)") and the reference itself
(vican/bipgo.py:353-490: NumPy / SciPy only) run anywhere; so does this package when - and ONLY when - the caller asks for it:
``bipartite_se3sync(..., device="cpu")`` / ``object_bipartite_se3sync(..., device="cpu")`` / ``solve_problem(..., device="cpu")``.

  * never automatic: without ``device="cpu"`` a missing GPU or library raises ``VicanError`` as before - a GPU box must never
    fall through to this file silently;
  * never the oracle: this is the SAME algorithm the GPU runs (matrix-free block Lanczos on L = Lambda_C - P, device-style
    Ritz verdict, Newton / SVD polar factors, scipy's CG recurrence with alpha / beta in a state vector) driven by the same
    ``vican_amd/solver.py`` - not the reference's explicit-P / ARPACK / scipy.cg formulation that ``oracle/`` restates; nothing
    here imports ``oracle/`` and the oracle does not import this;
  * never on the GPU path: ``vican_amd.bipgo`` imports it inside the ``device="cpu"`` branch only; ``bench.py``'s timed region
    and ``__graft_entry__.smoke()`` never touch it.

Every method restates what one entry point of ``include/vican_hip.h`` computes (the formulas of SURVEY.md section 3.3) on
CPU torch tensors; ``tests/test_kernels_gpu.py`` uses it as the per-kernel cross-check of the HIP kernels, the CPU tests
(``tests/test_solver_cpu.py``, ``tests/test_dist_cpu.py``: world_size-2 gloo) pin it to the goldens of the real reference.
"""
import numpy as np
import torch

from ._lib import CG_F, CG_I


def svd_polar(mats, mode):
    u, s, vt = np.linalg.svd(mats)
    d = np.linalg.det(u @ vt)
    fix = np.ones((len(mats), 3)); fix[:, 2] = d
    R = (u * fix[:, None, :]) @ vt
    lam = None
    if mode & 4:                      # no det fix (bipgo.py:126-127)
        R = u @ vt
        mode &= 3
    if mode == 1:
        lam = (u * s[:, None, :]) @ np.swapaxes(u, 1, 2)
    elif mode == 2:
        lam = (u * (1.0 / s)[:, None, :]) @ np.swapaxes(u, 1, 2)
    return R, lam


class NumpyBackend:
    def __init__(self, n_cam, row_ptr, col, blk, a, w=None, u=None, v=None, storage=np.float64, deg_t=None, deg_c=None,
                 sum_order="scipy"):
        # sum_order (translation-stage sums only): "scipy" = sequential in edge order, the order scipy's CSR product adds a
        # row's terms in (np.add.at); "reversed" = the same terms last to first; "exact" = accumulated in extended precision
        # and rounded once (what the product's exact fixed-point sums do).  tools/standin_orders.py uses the variants to show
        # what sharing scipy's order is worth against the reference's own reproducibility band.
        self.sum_order = sum_order
        self.C = int(n_cam)
        self.storage_f64 = np.dtype(storage) == np.float64
        self.T = len(row_ptr) - 1
        self.row_ptr = np.asarray(row_ptr, dtype=np.int64)
        self.col = np.asarray(col, dtype=np.int64)
        self.row = np.repeat(np.arange(self.T), np.diff(self.row_ptr))
        self.M = np.asarray(blk, dtype=storage).astype(np.float64).reshape(-1, 3, 3)
        self.a = np.asarray(a, dtype=storage).astype(np.float64)
        self.w = None if w is None else np.asarray(w, dtype=np.float64)
        self.u = None if u is None else np.asarray(u, dtype=np.float64).reshape(-1, 3)
        self.v = None if v is None else np.asarray(v, dtype=np.float64).reshape(-1, 3)
        self.deg_t_in, self.deg_c_in = deg_t, deg_c      # the caller's diagonal of the translation system (frontend: J^T J as scipy forms it)
        self.rr_part_n = 1
        self._rr = 0.0
        self._gate = None

    # launch gate (vican_set_gate): the gated entry points do nothing unless gate[0] == 1
    def gated(self, gate):
        import contextlib

        @contextlib.contextmanager
        def cm():
            self._gate = gate
            try:
                yield
            finally:
                self._gate = None
        return cm()

    def post_status(self, status):
        return status

    def wait_status(self, handle):
        return handle.numpy()

    def _closed(self):
        return self._gate is not None and int(self._gate[0]) != 1

    def ritz(self, HB, hw, steps, flags, eig_tol, floor_tol, floor_level, Y, status, gate, stall_ratio=0.25):
        """Mirror of vican_ritz (include/vican_hip.h) with LAPACK instead of the Jacobi iteration."""
        hb = HB[:steps].numpy()
        Hh = hb[:, :hw].reshape(steps, -1, 3)
        Bh = hb[:, hw:hw + 9].reshape(steps, 3, 3)
        eff = steps
        for j in range(steps):
            if np.any(np.diag(Bh[j]) == 0.0):
                eff = j + 1
                break
        ka = 3 * eff
        Tm = np.zeros((ka, ka))
        for j in range(eff):
            kj = 3 * (j + 1)
            Tm[:kj, 3 * j:3 * j + 3] = Hh[j, :kj, :]
        Tm = np.triu(Tm)
        Tm = Tm + Tm.T - np.diag(np.diag(Tm))
        th, Yv = np.linalg.eigh(Tm)
        beta = Bh[eff - 1]
        resmax = float(np.linalg.norm(beta @ Yv[ka - 3:ka, :3], axis=0).max())
        scale = max(abs(th[0]), abs(th[-1]), 1e-300)
        r = resmax / scale
        first, at_max = bool(flags & 1), bool(flags & 2)
        breakdown = bool(np.all(np.diag(beta) == 0.0))
        prev = float(status[12])
        floor_hit = (not first) and r > stall_ratio * prev and r <= floor_tol
        if floor_level >= 0.0 and r <= 2.0 * floor_level:
            floor_hit = True
        stop = eff < steps or breakdown or r <= eig_tol or floor_hit or at_max
        converged = breakdown or floor_hit or resmax <= eig_tol * scale
        y = Y.numpy().reshape(-1, 3)
        y[:3 * steps] = 0.0
        y[:ka] = Yv[:, :3]
        st = status.numpy()
        st[:7] = [r, scale, float(stop), float(converged), float(floor_hit), eff, float(breakdown)]
        st[7:10] = th[:3]
        st[10:12] = th[-2:] if ka >= 5 else np.nan
        st[12], st[13], st[14], st[15] = r, (th[4] if ka >= 5 else np.nan), resmax, (th[3] if ka >= 4 else np.nan)
        gate[0] = 1 if (stop and converged) else 0

    # allocation
    def empty(self, *shape, dtype=torch.float64):
        return torch.zeros(*shape, dtype=dtype)

    zeros = empty

    def from_numpy(self, a):
        return torch.from_numpy(np.ascontiguousarray(a))

    def synchronize(self):
        pass

    # rotation stage
    def init_duals(self, lamT_inv, cam_deg):
        d = np.zeros(self.T); np.add.at(d, self.row, self.a)
        cd = np.zeros(self.C); np.add.at(cd, self.col, self.a)
        if self.T:
            lamT_inv.numpy()[: self.T] = (np.eye(3)[None] / d[:, None, None]).reshape(self.T, 9)
        cam_deg.numpy()[:] = cd

    def scaled_identity(self, scale, out):
        out.numpy()[:] = (np.eye(3)[None] * scale.numpy()[:, None, None]).reshape(-1, 9)

    def _y(self, x):
        xc = x.numpy().reshape(self.C, 3, 3)[self.col]
        y = np.zeros((max(self.T, 1), 3, 3))
        np.add.at(y, self.row, np.swapaxes(self.M, 1, 2) @ xc)
        return y

    def block_op(self, lamT_inv, x, z_out):
        if self._closed():
            return
        y = self._y(x)
        w = lamT_inv.numpy().reshape(-1, 3, 3) @ y
        z = np.zeros((self.C, 3, 3))
        np.add.at(z, self.col, self.M @ w[self.row])
        z_out.numpy()[:] = z.reshape(3 * self.C, 3)

    # non-eliminated solver: both halves of R~ [x_cam; x_time]  (x, z: [3(C+T), 3], cameras first)
    def node_degrees(self, out):
        o = out.numpy(); o[:] = 0.0
        np.add.at(o, self.col, self.a)
        np.add.at(o, self.C + self.row, self.a)

    def bip_scales(self):
        pass

    def bip_apply(self, x, z_out):
        if self._closed():
            return
        C, T = self.C, self.T
        X = x.numpy().reshape(C + T, 3, 3)
        z = np.zeros((C + T, 3, 3))
        np.add.at(z, self.col, self.M @ X[C + self.row])
        np.add.at(z, C + self.row, np.swapaxes(self.M, 1, 2) @ X[self.col])
        z_out.numpy()[:] = z.reshape(3 * (C + T), 3)

    def dual_update(self, rc, Rt, lamT_inv):
        if self._closed():
            return
        if not self.T:
            return
        R, lam = svd_polar(self._y(rc)[: self.T], 2)
        Rt.numpy()[: self.T] = R.reshape(-1, 9)
        lamT_inv.numpy()[: self.T] = lam.reshape(-1, 9)

    def dual_update_op(self, rc, Rt, lamT_inv, z_raw):
        if self._closed():
            return
        z = np.zeros((self.C, 3, 3))
        if self.T:
            u, _, vt = np.linalg.svd(self._y(rc)[: self.T])
            np.add.at(z, self.col, self.M @ (u @ vt)[self.row])            # lamT_new Z_t = U V^T
        z_raw.numpy()[:] = z.reshape(3 * self.C, 3)
        self.dual_update(rc, Rt, lamT_inv)

    def lanczos_seed(self, x0, V, ld, beta0, xrow, zraw=None, z=None):
        n = x0.numel() // 3
        R, G = torch.zeros(3 * n, dtype=torch.float64), torch.zeros(9, dtype=torch.float64)
        self.rows_to_cols(n, x0, R, n, 0)
        self.tall_gram(n, R, n, 3, R, G)
        self.chol_qr3(n, R, G, V, ld, 0, beta0, xrow, 0.0)
        if zraw is not None:
            self.right_solve3(zraw, beta0, z)
        return True

    def right_solve3(self, X, beta, Z):
        b = beta.numpy().reshape(-1)[:9].reshape(3, 3)
        x = X.numpy().reshape(-1, 3)
        z = np.zeros_like(x)
        if b[0, 0] != 0.0:
            z[:, 0] = x[:, 0] / b[0, 0]
        if b[1, 1] != 0.0:
            z[:, 1] = (x[:, 1] - z[:, 0] * b[0, 1]) / b[1, 1]
        if b[2, 2] != 0.0:
            z[:, 2] = (x[:, 2] - z[:, 0] * b[0, 2] - z[:, 1] * b[1, 2]) / b[2, 2]
        Z.numpy().reshape(-1, 3)[:] = z

    def polar_dual(self, mats, R_out, lam_out, mode):
        if self._closed():
            return
        R, lam = svd_polar(mats.numpy().reshape(-1, 3, 3), mode)
        if R_out is not None:
            R_out.numpy().reshape(-1, 9)[:] = R.reshape(-1, 9)
        if lam_out is not None and (mode & 3):
            lam_out.numpy().reshape(-1, 9)[:] = lam.reshape(-1, 9)

    def gauge_project(self, x_in, x_out):
        if self._closed():
            return
        X = x_in.numpy()
        Xg = X @ np.linalg.inv(X[:3, :])
        R, _ = svd_polar(Xg.reshape(-1, 3, 3), 0)
        x_out.numpy()[:] = R.reshape(-1, 3)

    # Lanczos helpers (V column-major: column k = V[k*ld : k*ld+n])
    @staticmethod
    def _cols(V, ld, n, k0, k1):
        return V.numpy().reshape(-1)[k0 * ld: k1 * ld].reshape(k1 - k0, ld)[:, :n]

    def lap_apply(self, lamC, V, ld, col0, z, aq):
        nn = lamC.numel() // 9
        n = 3 * nn
        q = self._cols(V, ld, n, col0, col0 + 3).T.reshape(nn, 3, 3)          # rows 3c+i, cols b
        r = lamC.numpy().reshape(nn, 3, 3) @ q - z.numpy().reshape(nn, 3, 3)
        aq.numpy().reshape(3, n)[:] = r.reshape(n, 3).T

    def tall_gram(self, n, V, ld, ka, R, H):
        H.numpy().reshape(-1)[: ka * 3] = (self._cols(V, ld, n, 0, ka) @ R.numpy().reshape(3, n).T).reshape(-1)

    def tall_update(self, n, V, ld, ka, H, R, H_out, accumulate):
        h = H.numpy().reshape(-1)[: ka * 3].reshape(ka, 3).copy()
        Rv = R.numpy().reshape(3, n)
        Rv -= (self._cols(V, ld, n, 0, ka).T @ h).T
        if H_out is not None:
            ho = H_out.numpy().reshape(-1)
            ho[: ka * 3] = (ho[: ka * 3] if accumulate else 0.0) + h.reshape(-1)

    def chol_qr3(self, n, R, G, V, ld, col0, beta_out, x_out, pivot_floor=0.0):
        g = G.numpy().reshape(-1)[:9].reshape(3, 3)
        floor = max(1e-28 * np.trace(g), pivot_floor)
        b = np.zeros((3, 3)); inv = np.zeros(3)
        if g[0, 0] > floor:
            b[0, 0] = np.sqrt(g[0, 0]); inv[0] = 1 / b[0, 0]; b[0, 1] = g[0, 1] * inv[0]; b[0, 2] = g[0, 2] * inv[0]
        d11 = g[1, 1] - b[0, 1] ** 2
        if d11 > floor:
            b[1, 1] = np.sqrt(d11); inv[1] = 1 / b[1, 1]; b[1, 2] = (g[1, 2] - b[0, 1] * b[0, 2]) * inv[1]
        d22 = g[2, 2] - b[0, 2] ** 2 - b[1, 2] ** 2
        if d22 > floor:
            b[2, 2] = np.sqrt(d22); inv[2] = 1 / b[2, 2]
        Rv = R.numpy().reshape(3, n)
        q0 = Rv[0] * inv[0]
        q1 = (Rv[1] - q0 * b[0, 1]) * inv[1]
        q2 = (Rv[2] - q0 * b[0, 2] - q1 * b[1, 2]) * inv[2]
        self._cols(V, ld, n, col0, col0 + 3)[:] = np.stack([q0, q1, q2])
        if beta_out is not None:
            beta_out.numpy().reshape(-1)[:9] = b.reshape(-1)
        if x_out is not None:
            x_out.numpy()[:] = np.stack([q0, q1, q2], 1)

    def tall_combine(self, n, V, ld, ka, Y, X):
        if self._closed():
            return
        X.numpy()[:] = self._cols(V, ld, n, 0, ka).T @ Y.numpy().reshape(-1)[: ka * 3].reshape(ka, 3)

    def rows_to_cols(self, n, X, V, ld, col0):
        self._cols(V, ld, n, col0, col0 + 3)[:] = X.numpy().T

    # translation stage
    def _segsum(self, n, idx, vals):
        if self.sum_order == "reversed":
            out = np.zeros((n,) + vals.shape[1:]); np.add.at(out, idx[::-1], vals[::-1])
            return out
        if self.sum_order == "exact":
            out = np.zeros((n,) + vals.shape[1:], dtype=np.longdouble); np.add.at(out, idx, vals.astype(np.longdouble))
            return out.astype(np.float64)
        out = np.zeros((n,) + vals.shape[1:]); np.add.at(out, idx, vals)
        return out

    def trans_degrees(self, deg_t, deg_c):
        d = np.zeros(max(self.T, 1)); np.add.at(d, self.row, self.w)
        dc = np.zeros(self.C); np.add.at(dc, self.col, self.w)
        if self.deg_t_in is not None:
            d[: self.T] = self.deg_t_in
        if self.deg_c_in is not None:
            dc = np.asarray(self.deg_c_in, dtype=np.float64)
        deg_t.numpy()[:] = d; deg_c.numpy()[:] = dc

    def jacobi_scale(self, deg, s_out):
        d = deg.numpy()
        s_out.numpy()[:] = np.where(d > 0, 1.0 / np.sqrt(np.where(d > 0, d, 1.0)), 0.0)

    def row_scale(self, s, x):
        x.numpy()[:] = x.numpy() * s.numpy()[:, None]

    def set_cg_scaling(self, s_c, s_t):
        self._w_plain = self.w
        self.w = self.w * s_c.numpy()[self.col] * s_t.numpy()[self.row]

    def clear_cg_scaling(self):
        self.w = self._w_plain

    def trans_rhs(self, rc, rt, rhs_t, rhs_c):
        A = rc.numpy().reshape(self.C, 3, 3)[self.col]
        B = rt.numpy().reshape(-1, 3, 3)[self.row]
        g = np.einsum("eji,ej->ei", A, self.u) + np.einsum("eji,ej->ei", B, self.v)
        bt = self._segsum(max(self.T, 1), self.row, g)
        bc = self._segsum(self.C, self.col, -g)
        rhs_t.numpy()[:] = bt; rhs_c.numpy()[:] = bc

    @staticmethod
    def _st(st):
        return st.numpy(), st.numpy().view(np.int32)

    def cg_init(self, b_c, b_t, x_c, x_t, r_c, r_t, p_c, p_t, st):
        f, i = self._st(st)
        for x in (x_c, x_t):
            x.zero_()
        r_c.copy_(b_c); p_c.copy_(b_c); r_t.copy_(b_t); p_t.copy_(b_t)
        f[:] = 0.0
        f[CG_F["rr_cam"]] = float((b_c.numpy() ** 2).sum())
        f[CG_F["rr_time"]] = float((b_t.numpy()[: self.T] ** 2).sum())
        i[CG_I["first"]] = 1

    def cg_begin(self, r_c, p_c, rtol, st, n_rr_part=0):
        f, i = self._st(st)
        if i[CG_I["done"]]:
            return
        if n_rr_part > 0:
            self.cg_end(n_rr_part, st)
        rho = f[CG_F["rr_cam"]] + f[CG_F["rr_time"]]
        if i[CG_I["iter"]] == 0 and i[CG_I["first"]]:
            f[CG_F["bnorm2"]] = rho; f[CG_F["atol2"]] = rtol * rtol * rho
        f[CG_F["rho"]] = rho
        if np.sqrt(rho) < np.sqrt(f[CG_F["atol2"]]) or rho == 0.0:
            i[CG_I["done"]] = 1
            return
        if not i[CG_I["first"]]:
            beta = rho / f[CG_F["rho_prev"]]
            f[CG_F["beta"]] = beta
            p_c.numpy()[:] = r_c.numpy() + beta * p_c.numpy()

    def cg_sweep(self, deg_t, p_c, r_t, p_t, q_t, qcpq, st):
        f, i = self._st(st)
        if i[CG_I["done"]]:
            return
        T = self.T
        pt = p_t.numpy()
        if not i[CG_I["first"]]:
            pt[:T] = r_t.numpy()[:T] + f[CG_F["beta"]] * pt[:T]
        pc = p_c.numpy()
        acc = self._segsum(max(T, 1), self.row, self.w[:, None] * pc[self.col])
        q = deg_t.numpy()[:, None] * pt - acc
        q_t.numpy()[:] = q
        qc = self._segsum(self.C, self.col, self.w[:, None] * pt[self.row])
        out = qcpq.numpy()
        out[: 3 * self.C] = qc.reshape(-1)
        out[3 * self.C] = float((pt[:T] * q[:T]).sum())

    def cg_cam_step(self, deg_c, qcpq, p_c, x_c, r_c, st):
        f, i = self._st(st)
        if i[CG_I["done"]]:
            return
        pc = p_c.numpy()
        q = deg_c.numpy()[:, None] * pc - qcpq.numpy()[: 3 * self.C].reshape(self.C, 3)
        pq = qcpq.numpy()[3 * self.C] + float((pc * q).sum())
        alpha = f[CG_F["rho"]] / pq
        f[CG_F["pq"]] = pq; f[CG_F["alpha"]] = alpha
        x_c.numpy()[:] += alpha * pc
        r_c.numpy()[:] -= alpha * q
        f[CG_F["rr_cam"]] = float((r_c.numpy() ** 2).sum())

    def cg_time_step(self, p_t, q_t, x_t, r_t, st):
        f, i = self._st(st)
        if i[CG_I["done"]]:
            return 1
        a = f[CG_F["alpha"]]
        T = self.T
        x_t.numpy()[:T] += a * p_t.numpy()[:T]
        r_t.numpy()[:T] -= a * q_t.numpy()[:T]
        self._rr = float((r_t.numpy()[:T] ** 2).sum())
        return 1

    # LSQR on the merged system  s_e (p_t - p_c) = g_e / s_e   (vican_lsqr.hip)
    def lsqr_init_u(self, rc, rt, nrm2_out):
        A = rc.numpy().reshape(self.C, 3, 3)[self.col]
        B = rt.numpy().reshape(-1, 3, 3)[self.row]
        g = np.einsum("eji,ej->ei", A, self.u) + np.einsum("eji,ej->ei", B, self.v)
        self._lu = g / np.sqrt(self.w)[:, None]
        nrm2_out.numpy()[0] = float((self._lu ** 2).sum())

    def lsqr_u_step(self, v_c, v_t, coef, nrm2_out):
        s = np.sqrt(self.w)[:, None]
        self._lu = s * (v_t.numpy()[self.row] - v_c.numpy()[self.col]) - coef * self._lu
        nrm2_out.numpy()[0] = float((self._lu ** 2).sum())

    def lsqr_v_step(self, inv_beta, beta, v_t, acc_c, nrm2_t_out):
        a = np.sqrt(self.w)[:, None] * self._lu * inv_beta
        rows = np.zeros((max(self.T, 1), 3)); np.add.at(rows, self.row, a)
        cams = np.zeros((self.C, 3)); np.add.at(cams, self.col, -a)
        vt = v_t.numpy()
        vt[:] = rows - beta * vt
        acc_c.numpy()[:] = cams.reshape(-1)
        nrm2_t_out.numpy()[0] = float((vt[: self.T] ** 2).sum())

    def lsqr_cam_v(self, acc_c, beta, v_c, nrm2_out):
        vc = v_c.numpy()
        vc[:] = acc_c.numpy().reshape(self.C, 3) - beta * vc
        nrm2_out.numpy()[0] = float((vc ** 2).sum())

    def lsqr_update(self, inv_alfa, t1, t2, v, w, x, nrm2_w_out):
        vn, wn, xn = v.numpy(), w.numpy(), x.numpy()
        vn *= inv_alfa
        xn += t1 * wn
        wn[:] = vn + t2 * wn
        rows = self.T if vn.shape[0] != self.C else self.C
        nrm2_w_out.numpy()[0] = float((wn[:rows] ** 2).sum())

    # composites (same call surface as HipBackend)
    def block_op_slabs(self, lamT_inv, x):
        self._zslab = torch.zeros(3 * self.C, 3, dtype=torch.float64)
        self.block_op(lamT_inv, x, self._zslab)

    def lanczos_cam_step(self, lamC, V, ld, j, z, R, H, G, Hcol, beta, x_out, pivot_floor, from_slabs=False):
        if from_slabs:
            z = self._zslab
        n, ka = 3 * (lamC.numel() // 9), 3 * (j + 1)
        self.lap_apply(lamC, V, ld, 3 * j, z, R)
        self.tall_gram(n, V, ld, ka, R, H); self.tall_update(n, V, ld, ka, H, R, Hcol, 0)
        self.tall_gram(n, V, ld, ka, R, H); self.tall_update(n, V, ld, ka, H, R, Hcol, 1)
        self.tall_gram(n, R, n, 3, R, G)
        self.chol_qr3(n, R, G, V, ld, 3 * (j + 1), beta, x_out, pivot_floor)

    def cg_iter_local(self, deg_t, r_c, p_c, r_t, p_t, q_t, qcpq, rtol, st, n_rr_part):
        self.cg_begin(r_c, p_c, rtol, st, n_rr_part)
        self.cg_sweep(deg_t, p_c, r_t, p_t, q_t, qcpq, st)

    def cg_iter_finish(self, deg_c, qcpq, p_c, x_c, r_c, p_t, q_t, x_t, r_t, st):
        self.cg_cam_step(deg_c, qcpq, p_c, x_c, r_c, st)
        return self.cg_time_step(p_t, q_t, x_t, r_t, st)

    # the sharded iteration behind one call (include/vican_hip.h: vican_cg_iter_comm): scipy's recurrence with the two sums that
    # cross ranks travelling as FIXED slices - msg = [q_c partial | slices of p_t.q_t], then the slices of r_t.r_t - all-reduced
    # element by element and summed in a fixed order behind the collective: the same bits on every rank.  This backend issues
    # the two all-reduces through comm.allreduce itself (comm_iter_host): the HIP backend's are enqueued by the C library.
    comm_iter_host = True
    CG_RR_SLICES = 512

    def cg_iter_comm(self, deg_t, deg_c, r_c, p_c, x_c, r_t, p_t, q_t, x_t, msg, rtol, st, first, comm):
        from ._lib import CG_PQ_SLICES
        f, i = self._st(st)
        T, C3 = self.T, 3 * self.C
        if getattr(self, "_rr_slices", None) is None:
            self._rr_slices = torch.zeros(self.CG_RR_SLICES, dtype=torch.float64)
        rr = self._rr_slices
        if not i[CG_I["done"]]:
            if not first:                                       # close the previous iteration from the REDUCED slices of r.r
                f[CG_F["rr_time"]] = float(np.add.reduce(rr.numpy()))      # (fixed order: sequential over the slices)
                i[CG_I["iter"]] += 1
                f[CG_F["rho_prev"]] = f[CG_F["rho"]]
                i[CG_I["first"]] = 0
            self.cg_begin(r_c, p_c, rtol, st, 0)
        m = msg.numpy()
        if not i[CG_I["done"]]:
            self.cg_sweep(deg_t, p_c, r_t, p_t, q_t, msg[: C3 + 1], st)
            pq = (p_t.numpy()[:T] * q_t.numpy()[:T]).reshape(-1)
            sl = np.zeros(CG_PQ_SLICES)
            n_sl = max(1, min(CG_PQ_SLICES, -(-len(pq) // 8192)))
            for k in range(n_sl):                               # slice k = elements k, k + n_sl, ... blocks of 1024 (the kernel's map)
                sl[k] = float(pq.reshape(-1)[k::n_sl].sum()) if len(pq) else 0.0
            m[C3:] = sl
        comm.allreduce(msg)
        if not i[CG_I["done"]]:
            qcpq = torch.from_numpy(np.concatenate([m[:C3], [float(np.add.reduce(m[C3:]))]]))
            self.cg_cam_step(deg_c, qcpq, p_c, x_c, r_c, st)
            a = f[CG_F["alpha"]]
            x_t.numpy()[:T] += a * p_t.numpy()[:T]
            r_t.numpy()[:T] -= a * q_t.numpy()[:T]
            r2 = (r_t.numpy()[:T] ** 2).reshape(-1)
            nb = max(1, min(self.CG_RR_SLICES, -(-len(r2) // 1024)))
            rr.zero_()
            for k in range(nb):
                rr.numpy()[k] = float(r2[k::nb].sum()) if len(r2) else 0.0
        comm.allreduce(rr)

    # one message per CG iteration (include/vican_hip.h: vican_cg1_iter_local / vican_cg1_iter_finish)
    def cg1_iter_local(self, deg_t, r_c, r_t, s_t, msg, st, n_rr_part):
        f, i = self._st(st)
        if i[CG_I["done"]]:
            return
        T, C3 = self.T, 3 * self.C
        rt, rc = r_t.numpy(), r_c.numpy()
        acc = np.zeros((max(T, 1), 3)); np.add.at(acc, self.row, self.w[:, None] * rc[self.col])
        s = deg_t.numpy()[:, None] * rt - acc
        s_t.numpy()[:] = s
        qc = np.zeros((self.C, 3)); np.add.at(qc, self.col, self.w[:, None] * rt[self.row])
        out = msg.numpy()
        out[:C3] = qc.reshape(-1)
        out[C3] = float((rt[:T] * s[:T]).sum())
        out[C3 + 1] = float((rt[:T] ** 2).sum())

    def cg1_iter_finish(self, k, deg_c, msg, r_c, r_c_new, p_c, q_c, x_c, r_t, s_t, p_t, q_t, x_t, rtol, st):
        f, i = self._st(st)
        if i[CG_I["done"]]:
            return 1
        T, C3 = self.T, 3 * self.C
        m = msg.numpy()
        r_c_new.numpy()[:] = r_c.numpy()
        rc = r_c_new.numpy()
        sc = deg_c.numpy()[:, None] * rc - m[:C3].reshape(self.C, 3)
        gamma = float((rc ** 2).sum()) + m[C3 + 1]
        delta = float((rc * sc).sum()) + m[C3]
        if k == 0:
            f[CG_F["bnorm2"]] = gamma; f[CG_F["atol2"]] = rtol * rtol * gamma
        f[CG_F["rho_prev"]] = f[CG_F["rho"]]; f[CG_F["rho"]] = gamma
        if np.sqrt(gamma) < np.sqrt(f[CG_F["atol2"]]) or gamma == 0.0:
            i[CG_I["done"]] = 1
            return 1
        if k == 0:
            beta, alpha = 0.0, gamma / delta
        else:
            beta = gamma / self._cg1_gamma
            alpha = gamma / (delta - beta * gamma / self._cg1_alpha)
        self._cg1_gamma, self._cg1_alpha = gamma, alpha
        f[CG_F["alpha"]] = alpha; f[CG_F["beta"]] = beta
        i[CG_I["iter"]] = k + 1; i[CG_I["first"]] = 0
        for r, s_, p, q, x, n in ((r_t.numpy(), s_t.numpy(), p_t.numpy(), q_t.numpy(), x_t.numpy(), T),
                                  (rc, sc, p_c.numpy(), q_c.numpy(), x_c.numpy(), self.C)):
            if k:
                p[:n] = r[:n] + beta * p[:n]; q[:n] = s_[:n] + beta * q[:n]
            else:
                p[:n] = r[:n]; q[:n] = s_[:n]
            x[:n] += alpha * p[:n]
            r[:n] -= alpha * q[:n]
        return 1

    def cg_end(self, n_part, st):
        f, i = self._st(st)
        if i[CG_I["done"]]:
            return
        f[CG_F["rr_time"]] = self._rr
        i[CG_I["iter"]] += 1
        f[CG_F["rho_prev"]] = f[CG_F["rho"]]
        i[CG_I["first"]] = 0
