#!/usr/bin/env python3
"""profiles/hbm_traffic.json from a tools/gpu_pmc.sh run (rocprofv3 --pmc passes, counters averaged per launch).

Units/corrections (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and WRITE_SIZE are in KiB of 64-byte
fabric requests.  On gfx950 a 128-byte read request is tallied as 64 bytes, so FETCH_SIZE must be doubled for
fully coalesced streaming reads; other patterns have to be calibrated on a known byte count:
  * row_r2c reads whole contiguous input rows (128-byte requests): known 25.17 MB, FETCH_SIZE reports 12.4 MB
    -> factor 2.0 (matches the guide);
  * the column kernel reads whole contiguous 64 KB tiles: known 25.19 MB, FETCH_SIZE reports 12.7 MB -> factor 2.0;
  * the C2R kernels read the blocked spectrum in 32/64-byte pieces: the two-launch C2R kernel of round 1 read a known
    50.38 MB and FETCH_SIZE reported 49.5 MB -> factor 1.0 (k_row_c2r*).
  * the FUSED kernel, calibrated on itself in round 6 (tools/gpu_fetch_calibration.sh, profiles/r06_f_fetch_calibration_*.txt):
    1 GB written in front of every launch, so nothing of the column pass's output is left in the L2s or the Infinity Cache --
    TCC_EA0_RDREQ_sum = 593 538 requests per launch with AND without the fill (the "still L2-resident from the column kernel"
    reading of rounds 2-5 was wrong), TCC_EA0_RDREQ_32B_sum = 0, FETCH_SIZE = requests x 64 B = 37.99 MB.  The kernel needs
    54.6 MB of DISTINCT bytes (50.38 MB of spectrum rows x 13/12 for one halo pair per strip of 12 pairs) plus ~2 MB of corner rows:
    at least 44 % of its requests are 128-byte ones tallied at 64 -- its row pairs sit 64 contiguous bytes per tile, 32 bytes off
    the 64-byte grid, and two consecutive pairs share a 128-byte line.  Factor 1.44 (the lower bound: every distinct byte once;
    2.0 would be every request 128 bytes = 76 MB), applied to every k_c2r_sharpen_g instantiation (same access pattern).
    The same run confirms 2.0 for the row pass (197 206 requests x 128 B = 25.24 MB against 25.17 MB of input) and the column
    pass (198 561 x 128 B = 25.42 MB against 25.26 MB of S1).
Since round 2 the column kernel writes only the odd spectrum rows (25.2 MB, the even rows are the rows of S1).
WRITE_SIZE matched the known output bytes of every kernel within 1 % -> factor 1.0.
"""
import json
import re
import sys

src = sys.argv[1]            # gpurun_out/<tag>/summary.txt
out = sys.argv[2]            # profiles/hbm_traffic.json (merged: one entry per configuration key)
key = sys.argv[3]            # e.g. 2048x1024_p0_planar (bench.py: config_key)
factor = {"k_c2r_sharpen_g": 1.44, "k_row_r2c_t": 2.0, "k_row_r2c": 2.0, "k_col_t": 2.0, "k_col_v": 2.0, "k_row_r2c_m": 2.0, "k_col_m": 2.0, "k_row_r2c_n": 2.0, "k_col_n": 2.0}
names = {"k_row_r2c_t": "row_r2c", "k_row_r2c": "row_r2c", "k_row_r2c_m": "row_r2c", "k_row_r2c_n": "row_r2c", "k_col_t": "col_fwd_pad_inv", "k_col_v": "col_fwd_pad_inv",
         "k_col": "col_fwd_pad_inv", "k_col_m": "col_fwd_pad_inv", "k_col_n": "col_fwd_pad_inv", "k_c2r_sharpen_g": "row_c2r_sharpen", "k_c2r_sharpen_v": "row_c2r_sharpen",
         "k_row_c2r_t": "row_c2r", "k_row_c2r": "row_c2r", "k_sharpen_t": "sharpen", "k_sharpen": "sharpen"}
cur, data = None, {}
for line in open(src):
    m = re.match(r"== (\w+)", line)
    if m:
        cur = m.group(1)
        data[cur] = {}
        continue
    m = re.match(r"\s+(\w+)\s+([0-9.]+)", line)
    if m and cur:
        data[cur][m.group(1)] = float(m.group(2))
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
allres = json.load(open(out)) if os.path.exists(out) else {}
if "_method" not in allres or "row_r2c" in allres:      # (round-1 file: not keyed)
    allres = {}
allres["_method"] = __doc__
res = {"_source": src, "_kernel_sources_sha256": bench.kernel_sources_sha256()}
for k, c in data.items():
    if k not in names or "FETCH_SIZE" not in c:
        continue
    f = factor.get(k, 1.0)
    rd = c["FETCH_SIZE"] * 1024 * f
    wr = c["WRITE_SIZE"] * 1024
    res[names[k]] = {"kernel": k, "fetch_bytes": rd, "write_bytes": wr, "hbm_bytes_per_launch": rd + wr,
                     "fetch_correction": f, "l2_hit_rate": c.get("TCC_HIT_sum", 0) / max(1.0, c.get("TCC_HIT_sum", 0) + c.get("TCC_MISS_sum", 0)),
                     # wave-level instruction counts of one launch (SQ_INSTS_*): what the vector ALUs and the LDS have to issue
                     "valu_insts_per_launch": c.get("SQ_INSTS_VALU"), "lds_insts_per_launch": c.get("SQ_INSTS_LDS")}
allres[key] = res
json.dump(allres, open(out, "w"), indent=1)
print(json.dumps({k: v for k, v in res.items() if not k.startswith("_")}, indent=1))
