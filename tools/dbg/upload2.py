import sys, time
sys.path.insert(0, __file__.rsplit("/", 3)[0])
import numpy as np, torch
from vican_amd import frontend, synth
from vican_amd.geometry import SE3
if len(sys.argv) > 1 and sys.argv[1] == "1thread":
    torch.set_num_threads(1)
scene = synth.make_scene(n_cam=340, n_time=10000, n_marker=6, seed=0)
flat = synth.make_camera_edges(scene, cpt=4, mpv=1, sigma_r=1e-3, sigma_t=1e-3, seed=1)
cams = flat["cam_key"].astype(str); tm = np.char.partition(flat["marker_key"].astype(str), "_")
cons = synth.constraints_from_scene(scene, SE3); ones = np.ones(len(cams))
prob = frontend.flatten_arrays(cams, tm[:, 0], tm[:, 2], flat["R"], flat["t"], ones, ones, cons, np.float32)
dev = torch.device("cuda:0")
names = [("row_ptr", torch.int32), ("col", torch.int32), ("blk", torch.float32), ("a", torch.float32), ("w", torch.float64), ("u", torch.float64), ("v", torch.float64), ("deg_t", torch.float64), ("deg_c", torch.float64)]
print({n: (getattr(prob, n).dtype, getattr(prob, n).shape) for n, _ in names}, torch.get_num_threads())
torch.zeros(1, device=dev); torch.cuda.synchronize()
for i in range(40):
    ts = []; keep = []
    for n, dt in names:
        t0 = time.perf_counter()
        keep.append(torch.from_numpy(np.ascontiguousarray(getattr(prob, n))).to(dev, dt))
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    if max(ts) > 5 or i < 2:
        print(i, " ".join("%s %.2f" % (n, t) for (n, _), t in zip(names, ts)), flush=True)
print("done")
