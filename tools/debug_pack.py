import sys, torch, numpy as np
sys.path.insert(0, '.')
from vican_amd import synth
from vican_amd.device import LocalGraph
dev = torch.device('cuda:0')
gr = synth.make_merged_graph_torch(1000, 640, 250, dev, torch.float32, seed=0)
g = LocalGraph(1000, gr["row_ptr"], gr["col"], gr["blk"], gr["a"], block_threads=1024)
idx = g.idx.cpu().numpy().astype(np.uint32).reshape(g.n_chunk, 1024, 4)
pad = idx == 0xFFFFFFFF
cam = idx & 0xFFFF; row = idx >> 16
print("chunks", g.n_chunk, "max_rows", g.max_rows, "ncopy", g.n_copy, "pad frac", pad.mean())
lane = np.arange(1024)[None, :, None]
ok = (~pad) & ((cam & 31) == (lane & 31))
print("class-matching frac of valid:", ok.sum() / (~pad).sum())
nrows = np.array([[len(set(row[k, l][~pad[k, l]])) for l in range(1024)] for k in range(min(g.n_chunk, 4))])
print("distinct rows per lane: mean", nrows.mean(), "hist", np.bincount(nrows.reshape(-1)))
print(row[0, :3], cam[0, :3])
