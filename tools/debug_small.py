import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from test_kernels_gpu import make_backends
H, N, g = make_backends(5, 40, 1, 3, 105, np.float32, 256, None, False)
print("n_copy", g.n_copy, "max_rows", g.max_rows, "n_chunk", g.n_chunk, "n_wg", g.n_wg, "fx", g.fx.cpu().numpy())
rng = np.random.default_rng(1)
x = np.linalg.qr(rng.standard_normal((15, 3)))[0]
lamT_h, cd_h = H.empty(40, 9), H.empty(5)
H.init_duals(lamT_h, cd_h)
lam = rng.standard_normal((40, 3, 3)); lam = lam @ np.swapaxes(lam, 1, 2) + np.eye(3)
lam_h, lam_n = H.from_numpy(lam.reshape(40, 9)), N.from_numpy(lam.reshape(40, 9))
H.set_duals(lam_h)
print("fx after set_duals", g.fx.cpu().numpy())
zh, zn = H.empty(15, 3), N.empty(15, 3)
H.block_op(lam_h, H.from_numpy(x), zh); N.block_op(lam_n, N.from_numpy(x), zn)
print("fx after op", g.fx.cpu().numpy())
print(zh.cpu().numpy()[:4]); print(zn.numpy()[:4])
zp = H.zpart.view(torch.int64)[: g.n_wg * 45].cpu().numpy().reshape(g.n_wg, 9, 5)
print("zpart wg0 plane0", zp[0, 0], "max abs", np.abs(zp).max())
