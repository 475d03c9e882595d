import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    """GPU tests are skipped automatically where no GPU is visible, so a plain
    `pytest tests/` is green on the CPU container too."""
    try:
        import torch
        have = torch.cuda.is_available()
    except Exception:
        have = False
    if have:
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)


# ---- achieved parity, in the driver's record ---------------------------------------------------------------------------------
# The parity tests call record_parity() with what they MEASURED (not only "passed"): golden x dtype x path -> rotation error,
# translation error, the bound asserted, CG iterations against the reference's.  At session end the rows are written to
# profiles/r06_parity_table.json and printed as one compact table at the end of the run, so that the tail of the GPU test log -
# what the round's record keeps - carries the numbers themselves.
_PARITY = []
ROUND_TAG = "r06"


def record_parity(golden, dtype, path, rot_rad, trans_m, bound_m, cg_iters=None, cg_reference=None, note=""):
    _PARITY.append(dict(golden=str(golden), dtype=str(dtype).replace("float", "f"), path=str(path), rot_rad=float(rot_rad), trans_m=float(trans_m),
                        bound_m=None if bound_m is None else float(bound_m), cg_iters=None if cg_iters is None else int(cg_iters),
                        cg_reference=None if cg_reference is None else int(cg_reference), note=str(note)))


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    if not _PARITY:
        return
    import json
    rows = sorted(_PARITY, key=lambda r: (r["golden"], r["dtype"], r["path"]))
    try:
        import torch
        gpu = torch.cuda.is_available()
    except Exception:
        gpu = False
    if gpu:                                         # (the committed table is the GPU run's; a CPU-only session prints, does not overwrite)
        out = os.path.join(ROOT, "profiles", "%s_parity_table.json" % ROUND_TAG)
        try:
            json.dump(dict(north_star="1e-4 rad / 1e-4 m", rows=rows), open(out, "w"), indent=1)
        except OSError:
            pass
    tr = terminalreporter
    tr.write_line("")
    tr.write_line("achieved parity against the REAL reference's goldens (north star 1e-4 rad / 1e-4 m); cg = iterations here / reference")
    # at most ~25 lines: one per golden x dtype, the paths side by side
    byg = {}
    for r in rows:
        byg.setdefault((r["golden"], r["dtype"]), []).append(r)
    for (g, dt), rs in byg.items():
        cells = []
        for r in rs:
            cg = "" if r["cg_iters"] is None else " cg %d/%s" % (r["cg_iters"], "-" if r["cg_reference"] is None else r["cg_reference"])
            cells.append("%s: %.0e rad %.0e m%s%s" % (r["path"], r["rot_rad"], r["trans_m"], "" if r["bound_m"] is None else " (<%.0e)" % r["bound_m"], cg))
        tr.write_line("  %-22s %-3s %s" % (g[:22], dt, " | ".join(cells)))
