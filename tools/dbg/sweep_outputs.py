"""Operator-sweep outputs z = P x of a fixed set of graphs -> npz (run under two builds via VICAN_LIB and compare bit for bit)."""
import sys
sys.path.insert(0, __file__.rsplit("/", 3)[0])
import numpy as np, torch
from vican_amd import synth
from vican_amd.device import HipBackend, LocalGraph
dev = torch.device("cuda:0")
out = {}
cases = [("dense", 1000, 20000, None, 250, torch.float32), ("fixed8", 100, 200000, None, 8, torch.float32), ("ragged2_8", 100, 200000, (2, 8), 0, torch.float32),
         ("ragged1_4", 340, 100000, (1, 4), 0, torch.float32), ("ragged8_40", 300, 50000, (8, 40), 0, torch.float32), ("fixed8_f64", 100, 100000, None, 8, torch.float64),
         ("mid64", 1000, 30000, None, 64, torch.float32), ("tiny", 5, 40, (1, 3), 0, torch.float32), ("ragged2_8_f64", 100, 100000, (2, 8), 0, torch.float64)]
for name, C, T, rag, k, tdt in cases:
    if rag:
        rp, col, blk, a = synth.make_ragged_graph_torch(C, T, rag[0], rag[1], dev)
        blk, a = blk.to(tdt), a.to(tdt)
    else:
        d = synth.make_merged_graph_torch(C, T, k, dev, tdt, seed=0)
        rp, col, blk, a = d["row_ptr"], d["col"], d["blk"], d["a"]
    g = LocalGraph(C, rp, col, blk, a)
    H = HipBackend(g)
    lam, dg, z = H.empty(T, 9), H.empty(C), H.empty(3 * C, 3)
    H.init_duals(lam, dg)
    gen = torch.Generator(device="cpu"); gen.manual_seed(1)
    lam_r = torch.randn(T, 3, 3, generator=gen, dtype=torch.float64)
    lam_r = (lam_r @ lam_r.transpose(1, 2) + torch.eye(3, dtype=torch.float64)).reshape(T, 9).to(dev) * lam[:, :1]
    H.set_duals(lam_r)
    x = torch.linalg.qr(torch.randn(3 * C, 3, generator=gen, dtype=torch.float64))[0].contiguous().to(dev)
    H.block_op(lam_r, x, z)
    torch.cuda.synchronize()
    out[name] = z.cpu().numpy()
    print(name, g.layout, g.wg_waves, g.max_rows, g.n_copy, float(z.abs().max()), flush=True)
np.savez(sys.argv[1], **out)
