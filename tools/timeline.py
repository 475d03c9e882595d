#!/usr/bin/env python3
"""Timeline of one solve from a rocprofv3 --kernel-trace csv: per kernel start offset, duration and the idle gap
before it; totals of busy vs idle time.  usage: tools/timeline.py <dir with *kernel_trace.csv> [step_index_from_end]"""
import csv, glob, os, sys, collections

src = sys.argv[1]
back = int(sys.argv[2]) if len(sys.argv) > 2 else 1
f = glob.glob(os.path.join(src, "**/*kernel_trace.csv"), recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# a solve starts with init_duals_kernel
starts = [i for i, r in enumerate(rows) if r[2].startswith("init_duals_kernel")]
i0 = starts[-back]
i1 = starts[-back + 1] if back > 1 else len(rows)
seg = rows[i0:i1]
t0 = seg[0][0]
busy = 0; prev_end = seg[0][0]
gaps = collections.Counter(); gapn = collections.Counter(); dur = collections.Counter(); cnt = collections.Counter()
verbose = os.environ.get("TL_VERBOSE")
for s, e, n in seg:
    short = n.split("(")[0].replace("void ", "")[:48]
    gap = s - prev_end
    if verbose:
        print("%9.1f us  dur %7.1f  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap / 1e3, short))
    busy += e - s
    gaps[short] += max(gap, 0); gapn[short] += 1; dur[short] += e - s; cnt[short] += 1
    prev_end = max(prev_end, e)
tot = prev_end - t0
print("solve span %.1f us, busy %.1f us (%.1f %%), idle %.1f us, %d kernels" % (tot / 1e3, busy / 1e3, 100.0 * busy / tot, (tot - busy) / 1e3, len(seg)))
print("%-50s %5s %10s %10s %12s" % ("kernel", "n", "busy us", "gap-before", "gap/launch"))
for k, v in sorted(dur.items(), key=lambda kv: -(kv[1] + gaps[kv[0]])):
    print("%-50s %5d %10.1f %10.1f %12.2f" % (k, cnt[k], v / 1e3, gaps[k] / 1e3, gaps[k] / 1e3 / cnt[k]))
