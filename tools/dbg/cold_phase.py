import sys, time, gc
sys.path.insert(0, __file__.rsplit("/", 3)[0])
import numpy as np, torch
from vican_amd import bipgo, frontend, synth
from vican_amd.geometry import SE3
scene = synth.make_scene(n_cam=340, n_time=10000, n_marker=6, seed=0)
flat = synth.make_camera_edges(scene, cpt=4, mpv=1, sigma_r=1e-3, sigma_t=1e-3, seed=1)
cams = flat["cam_key"].astype(str); tm = np.char.partition(flat["marker_key"].astype(str), "_")
cons = synth.constraints_from_scene(scene, SE3); ones = np.ones(len(cams))
prob = frontend.flatten_arrays(cams, tm[:, 0], tm[:, 2], flat["R"], flat["t"], ones, ones, cons, np.float32)
mode = sys.argv[1] if len(sys.argv) > 1 else ""
if mode == "nogc": gc.disable()
if mode == "freeze": gc.freeze()
for i in range(16):
    info = {}
    t0 = time.perf_counter()
    bipgo.solve_problem(prob, 4, "conjugate_gradient", np.float32, info=info)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    print("%d: %.1f ms (pack %.1f rot %.1f trans %.1f other %.1f) gc %s" % (i, dt, info["t_pack"] * 1e3, info["t_rot"] * 1e3, info["t_trans"] * 1e3,
          dt - 1e3 * (info["t_pack"] + info["t_rot"] + info["t_trans"]), gc.get_count()), flush=True)
