#!/usr/bin/env python3
"""LDS bank-conflict model (MI355X_MICROARCH.md, LDS table) for the exchange patterns of the register-resident
Stockham kernels: counts LDS-array cycles per wave-instruction for candidate index maps.
ds_write_b64: groups of 16 contiguous lanes, bank = (byte/4) % 32;  ds_read_b64: groups of 32 lanes, bank = (byte/4) % 64."""
import itertools
import sys

def cycles_write_b64(addrs8):           # addrs8: element (8-byte) index per lane, 64 lanes
    tot = 0
    for g in range(4):
        slots = {}
        for l in range(16 * g, 16 * g + 16):
            a = addrs8[l]
            slots.setdefault(a % 16, set()).add(a)
        tot += max(len(v) for v in slots.values())
    return tot                           # conflict-free = 4

def cycles_read_b64(addrs8):
    tot = 0
    for g in range(2):
        slots = {}
        for l in range(32 * g, 32 * g + 32):
            a = addrs8[l]
            slots.setdefault(a % 32, set()).add(a)
        tot += max(len(v) for v in slots.values())
    return tot                           # conflict-free = 2

def stages(N, E, RMAX=8):
    out, Ns = [], 1
    while Ns < N:
        R = RMAX if N // Ns >= RMAX else N // Ns
        out.append((Ns, R))
        Ns *= R
    return out

def evaluate(phi, N, E, TK=1, RMAX=8, final_to_lds=False, verbose=False):
    Tc = N // E
    T = Tc * TK
    w_cyc = r_cyc = w_n = r_n = 0
    st = stages(N, E, RMAX)
    for si, (Ns, R) in enumerate(st):
        last = Ns * R == N
        if last and not final_to_lds:
            break
        NB = E // R
        for wave in range(max(1, T // 64)):
            lanes = range(64 * wave, 64 * wave + 64)
            for b in range(NB):
                for q in range(R):
                    ad = []
                    for t in lanes:
                        col, p = t % TK, t // TK
                        j = p + b * Tc
                        k = j % Ns
                        j0 = (j - k) * R + k
                        ad.append(phi((j0 + q * Ns) * TK + col))
                    c = cycles_write_b64(ad)
                    w_cyc += c; w_n += 1
            if not last:
                for i in range(E):
                    ad = [phi((t // TK + Tc * i) * TK + t % TK) for t in lanes]
                    c = cycles_read_b64(ad)
                    r_cyc += c; r_n += 1
    return w_cyc / max(w_n, 1), r_cyc / max(r_n, 1)

def pad(a):
    return lambda e: e + (e >> a)
def xr(shifts):                           # e ^ XOR over (src_shift, mask, dst_shift)
    def f(e):
        x = e
        for (s, m, d) in shifts:
            x ^= ((e >> s) & m) << d
        return x
    return f

if __name__ == "__main__":
    N, E, TK = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    RMAX = int(sys.argv[4]) if len(sys.argv) > 4 else 8
    ftl = len(sys.argv) > 5 and sys.argv[5] == "1"
    print("N=%d E=%d TK=%d RMAX=%d stages=%s  (ideal: write 4, read 2)" % (N, E, TK, RMAX, stages(N, E, RMAX)))
    cands = {"none": lambda e: e, "pad16": pad(4), "pad32": pad(5), "pad8": pad(3)}
    for s1 in range(3, 10):
        for m in (1, 3, 7, 15):
            for d in range(0, 4):
                cands["x(%d,%d,%d)" % (s1, m, d)] = xr([(s1, m, d)])
    for (s1, s2) in itertools.combinations(range(3, 10), 2):
        for (m1, d1, m2, d2) in ((7, 0, 1, 3), (3, 0, 3, 2), (1, 3, 7, 0), (3, 2, 3, 0), (15, 0, 15, 0), (7, 0, 7, 0), (3, 0, 3, 0)):
            cands["x(%d,%d,%d)+(%d,%d,%d)" % (s1, m1, d1, s2, m2, d2)] = xr([(s1, m1, d1), (s2, m2, d2)])
    res = []
    for name, phi in cands.items():
        w, r = evaluate(phi, N, E, TK, RMAX, ftl)
        res.append((w * 1.5 + r, w, r, name))      # a write occupies >= 6 cycles of its own: array cycles up to 6 are free
    res.sort()
    for sc, w, r, name in res[:12]:
        print("%-28s write %.2f  read %.2f" % (name, w, r))
    for name in ("none", "pad16", "pad32"):
        w, r = evaluate(cands[name], N, E, TK, RMAX, ftl)
        print("%-28s write %.2f  read %.2f" % (name, w, r))
