import time, numpy as np, torch
dev = torch.device("cuda:0")
ts = [torch.randn(n, device=dev, dtype=torch.float64) for n in (90000, 30000, 1020, 3060, 900000, 300000)]
torch.cuda.synchronize()
for mode in ("pageable", "pageable", "pageable"):
    out = []
    for i in range(60):
        t0 = time.perf_counter()
        hs = [t.cpu().numpy() for t in ts]
        out.append((time.perf_counter() - t0) * 1e3)
    print(mode, "max %.2f median %.2f" % (max(out), sorted(out)[len(out) // 2]), " ".join("%.1f" % x for x in out if x > 3), flush=True)
