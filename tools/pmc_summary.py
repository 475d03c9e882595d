#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection.csv files per kernel (mean per dispatch)."""
import csv, glob, sys, collections
def main(dirs, pat):
    for d in dirs:
        for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
            acc = collections.defaultdict(lambda: collections.defaultdict(list))
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"]
                if pat in k:
                    acc[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
            for k, cs in acc.items():
                print(f, k)
                for c, v in sorted(cs.items()):
                    print("   %-26s n=%4d mean=%16.1f" % (c, len(v), sum(v) / len(v)))
if __name__ == "__main__":
    main(sys.argv[2:], sys.argv[1])
