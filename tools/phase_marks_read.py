"""Prints the timeline recorded by an instrumented library (tools/phase_marks.py)."""
import sys, numpy as np
d = np.fromfile(sys.argv[1], dtype=np.uint64)
ids = (d >> np.uint64(56)).astype(int); t = (d & np.uint64((1 << 56) - 1)).astype(np.int64)
# last complete strip: find last id==0
starts = [i for i in range(4000) if ids[i] == 0 and t[i] > 0]
i0 = starts[-1] if starts else 0
n = i0 + 1
while n < len(ids) and ids[n] != 0 and t[n] > 0: n += 1
seg = list(zip(ids[i0:n], t[i0:n]))
print("marks in strip:", len(seg)); print("HW_ID words:", [hex(int(x)) for x in d[4000:4008]])
names = {0: "step0", 1: "step", 2: "pre-fft(v formed, prefetch issued)", 3: "fft done", 4: "L rows written", 5: "after L barrier", 6: "sharpen done", 7: "after settle"}
names[8] = "chunk(1,1) done"
names[13] = "  lane transpose done"; names[9] = "taps landed"
for k in range(3): names[50 + k] = "  st%d chunk done" % k
for k in range(3): names[10 + k] = "  st%d bfly done" % k; names[20 + k] = "  st%d scatter issued" % k; names[30 + k] = "  st%d barrier passed" % k; names[40 + k] = "  st%d gather landed" % (k)
# per step table for steps 3..5
stepidx = [i for i, (a, b) in enumerate(seg) if a in (0, 1)]
for si in stepidx[3:6]:
    base = seg[si][1]
    j = si
    prev = base
    while j < len(seg) and (j == si or seg[j][0] not in (0, 1)):
        a, b = seg[j]
        print("%6d (+%5d)  %s" % (b - base, b - prev, names.get(a, str(a))))
        prev = b
        j += 1
    print()
tot = seg[-1][1] - seg[0][1]
print("strip total cycles:", tot, "steps:", len(stepidx))
