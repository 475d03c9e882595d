import time, numpy as np, torch
dev = torch.device("cuda:0")
rng = np.random.default_rng(0)
arrs = [rng.standard_normal(n) for n in (10001, 40000, 360000, 40000, 40000, 120000, 120000, 10000, 340)]
torch.zeros(1, device=dev); torch.cuda.synchronize()
def run(mode):
    out = []
    pin = torch.empty(sum(a.nbytes for a in arrs) + 4096, dtype=torch.uint8).pin_memory() if mode == "pinned" else None
    for i in range(24):
        t0 = time.perf_counter()
        if mode == "pageable":
            ts = [torch.from_numpy(a).to(dev) for a in arrs]
        elif mode == "pinned":
            o = 0; ts = []
            for a in arrs:
                v = pin[o:o + a.nbytes].view(torch.float64)
                v.numpy()[:] = a
                ts.append(v.to(dev, non_blocking=True)); o += (a.nbytes + 255) // 256 * 256
        elif mode == "alloc_only":
            ts = [torch.empty(a.size, dtype=torch.float64, device=dev) for a in arrs]
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) * 1e3)
        del ts
    print(mode, " ".join("%.2f" % x for x in out), flush=True)
for m in ("pageable", "pinned", "alloc_only", "pageable"):
    run(m)
