#!/usr/bin/env python3
"""Print the rows of a rocprofv3 kernel_stats.csv whose kernel name matches a pattern: calls, average / min / max in us."""
import csv
import re
import sys
pat = re.compile(sys.argv[2] if len(sys.argv) > 2 else ".")
for r in csv.DictReader(open(sys.argv[1])):
    if pat.search(r["Name"]):
        name = re.sub(r"\(.*", "", r["Name"]).replace("void ", "")
        print("%-52s n %5s  avg %9.1f  min %9.1f  max %9.1f us" % (name[:52], r["Calls"], float(r["AverageNs"]) / 1e3,
                                                                  float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
