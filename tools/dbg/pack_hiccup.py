import sys, time
sys.path.insert(0, __file__.rsplit("/", 3)[0])
import numpy as np, torch
from vican_amd import frontend, synth
from vican_amd.device import make_backend
from vican_amd.geometry import SE3
from vican_amd.solver import Comm, RotationSolver, TranslationSolver
scene = synth.make_scene(n_cam=340, n_time=10000, n_marker=6, seed=0)
flat = synth.make_camera_edges(scene, cpt=4, mpv=1, sigma_r=1e-3, sigma_t=1e-3, seed=1)
cams = flat["cam_key"].astype(str); tm = np.char.partition(flat["marker_key"].astype(str), "_")
cons = synth.constraints_from_scene(scene, SE3); ones = np.ones(len(cams))
prob = frontend.flatten_arrays(cams, tm[:, 0], tm[:, 2], flat["R"], flat["t"], ones, ones, cons, np.float32)
dev = torch.device("cuda:0")
to = lambda a, dt=torch.float64: torch.from_numpy(np.ascontiguousarray(a)).to(dev, dt)
what = sys.argv[1] if len(sys.argv) > 1 else "all"
for i in range(16):
    t0 = time.perf_counter()
    args = (prob.n_cam, to(prob.row_ptr, torch.int32), to(prob.col, torch.int32), to(prob.blk, torch.float32), to(prob.a, torch.float32), to(prob.w), to(prob.u), to(prob.v))
    torch.cuda.synchronize(); t1 = time.perf_counter()
    g, K = make_backend(*args, deg_t=to(prob.deg_t), deg_c=to(prob.deg_c))
    torch.cuda.synchronize(); t2 = time.perf_counter()
    t3 = t4 = t2
    if what in ("all", "rot"):
        rot = RotationSolver(K, Comm())
        torch.cuda.synchronize(); t3 = time.perf_counter()
        rc, Rt = rot.run(4)
        torch.cuda.synchronize(); t4 = time.perf_counter()
    print("%d: upload %.2f backend %.2f solver-init %.2f run %.2f" % (i, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3), flush=True)
