#!/usr/bin/env python3
"""Time graph construction (plan + pack + graph constants) for the stress graph (GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import torch
from vican_amd import synth
from vican_amd.device import LocalGraph
dev = torch.device("cuda:0")
gr = synth.make_merged_graph_torch(1000, 100000, 250, dev, torch.float32, seed=0, t_offset=0)
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    g = LocalGraph(1000, gr["row_ptr"], gr["col"], gr["blk"], gr["a"], gr["w"], gr["u"], gr["v"])
    torch.cuda.synchronize()
    print("LocalGraph build: %.2f ms (%d edges, %d chunks)" % ((time.perf_counter() - t0) * 1e3, g.n_edges, g.n_chunk))
    del g
