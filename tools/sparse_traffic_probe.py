#!/usr/bin/env python3
"""Why does rocprofv3 report LESS HBM traffic than the algorithmic bytes for the operator sweep of the sparse capture (621 MB
against 792 MB), when it reports 0.96-1.00 x for the stress graph?  Hypothesis: the 152 MB of row-side reads - the dual blocks
Lambda_T^-1 (72 B per row, 2 M rows) and the row bounds - are the same bytes in every launch and fit the 256 MB Infinity
Cache, so they are served from it and never reach HBM (FETCH_SIZE counts what the L2 fetches from the memory side... past the
Infinity Cache?).  Probe: the same sweep launched alternately (a) right after the previous one, (b) after 1.5 GB of unrelated
reads that evict the Infinity Cache.  Run under
    rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d <dir> -- python3 tools/sparse_traffic_probe.py
and compare the counter of the even and the odd dispatches of wave_sweep_kernel (tools/pmc_summary.py prints the mean; this
script prints the launch order)."""
import sys
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
from vican_amd import synth                                     # noqa: E402
from vican_amd.device import HipBackend, LocalGraph             # noqa: E402
from vican_amd.solver import RotationSolver                     # noqa: E402

C, T, K = 100, 2000000, 8
dev = torch.device("cuda:0")
d = synth.make_merged_graph_torch(C, T, K, dev, torch.float32, seed=0)
g = LocalGraph(C, d["row_ptr"], d["col"], d["blk"], d["a"])
H = HipBackend(g)
rot = RotationSolver(H, prop_sweeps=0)
rot.init()
x = torch.randn(3 * C, 3, dtype=torch.float64, device=dev)
junk = torch.empty(1_500_000_000 // 4, dtype=torch.float32, device=dev)
for i in range(12):
    if i % 2:
        junk.add_(1.0)                                           # (b): 1.5 GB read + written through the caches
    H.block_op_raw(rot.lamT, x)
torch.cuda.synchronize()
print("launched 12 sweeps: even = back to back, odd = after 3 GB of unrelated traffic; algorithmic bytes per sweep %d" % g.op_bytes())
