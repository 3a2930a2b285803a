#!/usr/bin/env python3
"""Instruction histogram of one kernel in a hipcc -save-temps .s file.
usage: isa_hist.py file.s kernel_substring [--blocks]"""
import collections
import re
import sys

path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = None
for i, l in enumerate(lines):
    if re.match(r"^_Z\w*:", l) and key in l:
        start = i
        break
assert start is not None, "kernel not found"
end = next(i for i in range(start, len(lines)) if lines[i].startswith("\t.end_amdhsa_kernel") or lines[i].startswith(".Lfunc_end"))
body = lines[start:end]
hist = collections.Counter()
cls = collections.Counter()
blocks = []
cur = ["entry", collections.Counter()]
for l in body:
    m = re.match(r"^(\.LBB\w+):", l)
    if m:
        blocks.append(cur)
        cur = [m.group(1), collections.Counter()]
        continue
    m = re.match(r"^\t([a-z_0-9]+)", l)
    if not m or l.startswith("\t."):
        continue
    op = m.group(1)
    hist[op] += 1
    c = ("v_pk" if op.startswith("v_pk_") else "valu" if op.startswith("v_") else "ds" if op.startswith("ds_") else
         "vmem" if op.startswith(("global_", "buffer_", "flat_", "scratch_")) else "salu" if op.startswith("s_") else "other")
    if op in ("s_waitcnt", "s_barrier", "s_nop"):
        c = op
    cls[c] += 1
    cur[1][c] += 1
blocks.append(cur)
print("classes:", dict(cls))
print("top ops:")
for op, n in hist.most_common(45):
    print("  %-28s %d" % (op, n))
if "--blocks" in sys.argv:
    for name, c in blocks:
        tot = sum(c.values())
        if tot >= 40:
            print("%-14s total %4d  %s" % (name, tot, dict(c)))
