#!/usr/bin/env python3
"""profiles/kernel_stats_index.json: which committed `rocprofv3 --kernel-trace --stats` summary belongs to which bench
configuration, the average duration of the frame's kernels in it, and the fingerprint of the kernel sources it was measured on
(bench.py prints roofline.frac_rocprof from it and says whether it is current).
    tools/index_kernel_stats.py <key> <profiles/..._kernel_stats_....csv> [<key> <csv> ...]"""
import csv
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

SHORT = (("k_row_r2c", "row_r2c"), ("k_row_c2c_fwd", "row_c2c"), ("k_col", "col_fwd_pad_inv"), ("k_c2r_sharpen", "row_c2r_sharpen"),
         ("k_row_c2r", "row_c2r"), ("k_sharpen", "sharpen"))
out = os.path.join(ROOT, "profiles", "kernel_stats_index.json")
idx = json.load(open(out)) if os.path.exists(out) else {}
for key, path in zip(sys.argv[1::2], sys.argv[2::2]):
    kern = {}
    for row in csv.DictReader(open(path)):
        name = row["Name"].replace("void ", "").replace("fftup::", "")
        for prefix, short in SHORT:
            if name.startswith(prefix) and short not in kern:
                kern[short] = float(row["AverageNs"])
    idx[key] = {"file": os.path.relpath(os.path.abspath(path), ROOT), "kernel_sources_sha256": bench.kernel_sources_sha256(), "kernels_avg_ns": kern}
json.dump(idx, open(out, "w"), indent=1)
print(json.dumps(idx, indent=1))
