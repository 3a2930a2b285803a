#!/usr/bin/env python3
"""Average every PMC counter per kernel name over all dispatches found under <dir>/pass*/ (rocprofv3 csv)."""
import csv
import glob
import sys
from collections import defaultdict

root = sys.argv[1]
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
for f in sorted(glob.glob(root + "/pass*/**/*counter_collection.csv", recursive=True)):
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("fftup::", "")
        a = acc[name][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"])
        a[1] += 1
for name in sorted(acc):
    print("== %s" % name)
    for cn in sorted(acc[name]):
        s, n = acc[name][cn]
        print("   %-28s %16.1f  (avg of %d dispatches)" % (cn, s / n, n))
