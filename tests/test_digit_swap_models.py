"""CPU: the index algebra of the digit-swap transforms of csrc/kernels_dswap.hpp, replayed in numpy -- which element sits in
which (wave, lane, register) after every exchange, and which twiddle it meets there.  These are the models the HIP kernels were
written from (a 4096-point transform with one LDS exchange; k_col_v: forward decimation-in-time, phase, inverse
decimation-in-frequency of the 1024-point column pass); the GPU parity tests check the kernels, these check the derivation."""
import numpy as np

W8 = {+1: np.exp(+2j * np.pi * np.outer(np.arange(8), np.arange(8)) / 8), -1: np.exp(-2j * np.pi * np.outer(np.arange(8), np.arange(8)) / 8)}


def test_vfft4096_digit_swap_model():
    N, DIR = 4096, -1
    rng = np.random.default_rng(1)
    Z = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    ref = np.fft.fft(Z)                                   # X[k] = sum x[n] exp(-2 pi i n k / N)
    tw = np.conj(np.exp(2j * np.pi * np.arange(N) / N))   # twid<-1>(table)
    v = np.zeros((8, 64, 8), complex)                     # [wave][lane][register]
    for w in range(8):
        for l in range(64):
            a, b = l & 7, l >> 3
            v[w, l, :] = Z[b + 8 * a + 64 * w + 512 * np.arange(8)]         # first-stage inputs of thread (w, l)
    v = v @ W8[DIR].T
    nv = np.zeros_like(v)                                 # exchange A: register <-> wave (LDS)
    for w in range(8):
        for k0 in range(8):
            nv[k0, :, w] = v[w, :, k0]
    v = nv
    for w in range(8):
        for m in range(8):
            v[w, :, m] *= tw[(64 * m * w) % N]             # wave-uniform twiddles
    v = v @ W8[DIR].T
    nv = np.zeros_like(v)                                 # exchange B: register <-> lane bits 0-2, inside the wave's LDS block
    for w in range(8):
        blk = np.zeros(512, complex)
        for l in range(64):
            a, b = l & 7, l >> 3
            for k1 in range(8):
                blk[a * 64 + 8 * b + (a ^ k1)] = v[w, l, k1]
        for l in range(64):
            a, b = l & 7, l >> 3
            for m in range(8):
                nv[w, l, m] = blk[m * 64 + 8 * b + (m ^ a)]
    v = nv
    for w in range(8):
        for l in range(64):
            v[w, l, :] *= tw[(8 * np.arange(8) * (w + 8 * (l & 7))) % N]
    v = v @ W8[DIR].T
    nv = np.zeros_like(v)                                 # exchange C: register <-> lane bits 3-5 (permlane swaps, DPP)
    for l in range(64):
        a, b = l & 7, l >> 3
        for k2 in range(8):
            nv[:, a + 8 * k2, b] = v[:, l, k2]
    v = nv
    for w in range(8):
        for l in range(64):
            v[w, l, :] *= tw[(np.arange(8) * (w + 8 * l)) % N]
    v = v @ W8[DIR].T
    for w in range(8):
        for l in range(64):
            assert np.abs(v[w, l, :] - ref[w + 8 * l + 512 * np.arange(8)]).max() <= 1e-9      # X[w + 8 l + 512 q] in register q


def test_k_col_v_digit_swap_model():
    N = 1024
    rng = np.random.default_rng(2)
    x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    tw = lambda j, n=N: np.exp(2j * np.pi * j / n)
    F = np.fft.ifft(x) * N                                # forward: exp(+2 pi i n k / N)
    k = np.arange(N)
    t = np.exp(-2j * np.pi * k / (2 * N)) * np.where(k < N // 2, 1, -1)
    ref = np.fft.fft(F * t)                               # the odd rows of the zero-padded inverse (polyphase column pass)
    v = np.zeros((8, 16, 8), complex)                     # [wave][h = lane >> 2][register]; pp = 16 w + h
    for w in range(8):
        for h in range(16):
            v[w, h, :] = x[16 * w + h + 128 * np.arange(8)]
    v = v @ W8[+1].T
    nv = np.zeros_like(v)                                 # A: register <-> wave
    for w in range(8):
        for k0 in range(8):
            nv[k0, :, w] = v[w, :, k0]
    v = nv
    for k0 in range(8):
        for R in range(8):
            v[k0, :, R] *= tw(16 * R * k0)
    v = v @ W8[+1].T
    nv = np.zeros_like(v)                                 # C: register <-> lane bits 5-3 (= h >> 1)
    for h in range(16):
        g, h0 = h >> 1, h & 1
        for k1 in range(8):
            nv[:, (k1 << 1) | h0, g] = v[:, h, k1]
    v = nv
    for w in range(8):
        for h in range(16):
            v[w, h, :] *= tw(2 * np.arange(8) * (w + 8 * (h >> 1)))
    v = v @ W8[+1].T
    nv = np.zeros_like(v)                                 # register bit 2 <-> lane bit 2 (= h & 1)
    for h in range(16):
        k1, h0 = h >> 1, h & 1
        for k2 in range(8):
            nv[:, (k1 << 1) | (k2 >> 2), (h0 << 2) | (k2 & 3)] = v[:, h, k2]
    v = nv
    for w in range(8):
        for h in range(16):
            k1, lb2 = h >> 1, h & 1
            for r in range(4):
                a, b = v[w, h, r], v[w, h, r + 4] * tw(w + 8 * k1 + 256 * lb2) * tw(64 * r)
                v[w, h, r], v[w, h, r + 4] = a + b, a - b
    for w in range(8):                                    # F[k] at k = w + 8 k1 + 256 lb2 + 64 r + 512 k3 in register r + 4 k3
        for h in range(16):
            for r in range(4):
                for k3 in range(2):
                    assert abs(v[w, h, r + 4 * k3] - F[w + 8 * (h >> 1) + 256 * (h & 1) + 64 * r + 512 * k3]) <= 1e-9
    for w in range(8):                                    # the phase, where the elements are
        for h in range(16):
            base = np.conj(tw(w + 8 * (h >> 1) + 256 * (h & 1), 2 * N))
            for r in range(4):
                for k3 in range(2):
                    v[w, h, r + 4 * k3] *= base * np.exp(-2j * np.pi * r / 32) * (1j if k3 else 1)
    for w in range(8):                                    # inverse, decimation in frequency
        for h in range(16):
            for r in range(4):
                a, b = v[w, h, r], v[w, h, r + 4]
                v[w, h, r], v[w, h, r + 4] = a + b, a - b
    nv = np.zeros_like(v)
    for h in range(16):
        k1, lb2 = h >> 1, h & 1
        for R in range(8):
            nv[:, (k1 << 1) | (R >> 2), (lb2 << 2) | (R & 3)] = v[:, h, R]
    v = nv
    for h in range(16):
        v[:, h, :] *= np.exp(-2j * np.pi * np.arange(8) * (h & 1) / 16)
    v = v @ W8[-1].T
    nv = np.zeros_like(v)
    for h in range(16):
        k1, h0 = h >> 1, h & 1
        for g in range(8):
            nv[:, (g << 1) | h0, k1] = v[:, h, g]
    v = nv
    for h in range(16):
        v[:, h, :] *= np.conj(tw(8 * h * np.arange(8)))
    v = v @ W8[-1].T
    nv = np.zeros_like(v)
    for k0 in range(8):
        for wp in range(8):
            nv[wp, :, k0] = v[k0, :, wp]
    v = nv
    for w in range(8):
        for h in range(16):
            v[w, h, :] *= np.conj(tw((16 * w + h) * np.arange(8)))
    v = v @ W8[-1].T
    for w in range(8):
        for h in range(16):
            assert np.abs(v[w, h, :] - ref[16 * w + h + 128 * np.arange(8)]).max() <= 1e-8      # row pp + 128 i in register i: the load layout


def test_row_r2c_v_digit_swap_model():
    """k_row_r2c_v<2048>: forward transform of 2048 = 8 (registers) x 4 (waves) x 8 (lane bits 5-3) x 8 (lane bits 2-0) points on
    256 threads, decimation in time.  Exchanges: registers <-> wave through LDS (8 register values against 4 waves: a thread
    comes out with 4 values of the wave digit x 2 of the 8 first-stage outputs, two radix-4 butterflies), registers <-> lane
    bits 5-3 by permlane swaps, registers <-> lane bits 2-0 inside the wave's own LDS block."""
    N = 2048
    rng = np.random.default_rng(3)
    x = rng.standard_normal(N) + 1j * rng.standard_normal(N)
    tw = lambda j: np.exp(2j * np.pi * (j % N) / N)       # the table (sign +)
    F = np.fft.ifft(x) * N                                # forward: exp(+2 pi i n k / N)
    W4 = np.exp(2j * np.pi * np.outer(np.arange(4), np.arange(4)) / 4)
    v = np.zeros((4, 64, 8), complex)                     # [wave][lane][register]
    for w in range(4):
        for l in range(64):
            v[w, l, :] = x[64 * w + l + 256 * np.arange(8)]                  # the load layout of k_row_r2c_t: x[tid + 256 i]
    v = v @ W8[+1].T                                      # registers: k0
    nv = np.zeros_like(v)                                 # A: element (k0, n2 = wave) -> thread wave k0 >> 1, register 4 (k0 & 1) + n2
    for n2 in range(4):
        for k0 in range(8):
            nv[k0 >> 1, :, 4 * (k0 & 1) + n2] = v[n2, :, k0]
    v = nv
    for wp in range(4):
        for b in range(2):
            for n2 in range(4):
                v[wp, :, 4 * b + n2] *= tw(64 * n2 * (2 * wp + b))          # exp(2 pi i n2 k0 / 32): wave-uniform
            v[wp, :, 4 * b:4 * b + 4] = v[wp, :, 4 * b:4 * b + 4] @ W4.T     # registers: 4 b + k1
    nv = np.zeros_like(v)                                 # C: register <-> lane bits 5-3
    for l in range(64):
        lo, hi = l & 7, l >> 3
        for m in range(8):
            nv[:, lo + 8 * m, hi] = v[:, l, m]
    v = nv
    for wp in range(4):
        for l in range(64):
            m = l >> 3
            kk = 2 * wp + (m >> 2) + 8 * (m & 3)         # k0 + 8 k1
            v[wp, l, :] *= tw(8 * np.arange(8) * kk)      # exp(2 pi i n1 (k0 + 8 k1) / 256)
    v = v @ W8[+1].T                                      # registers: k2
    nv = np.zeros_like(v)                                 # B: register <-> lane bits 2-0
    for l in range(64):
        lo, hi = l & 7, l >> 3
        for k2 in range(8):
            nv[:, 8 * hi + k2, lo] = v[:, l, k2]
    v = nv
    for wp in range(4):
        for l in range(64):
            m, k2 = l >> 3, l & 7
            kk = 2 * wp + (m >> 2) + 8 * (m & 3) + 32 * k2
            v[wp, l, :] *= tw(np.arange(8) * kk)
            v[wp, l, :] = W8[+1] @ v[wp, l, :]
            assert np.abs(v[wp, l, :] - F[kk + 256 * np.arange(8)]).max() <= 1e-8          # X[k'' + 256 k3] in register k3


import pytest


@pytest.mark.parametrize("WV", [8, 4, 2])
def test_k_col_v_general_wave_digit_model(WV):
    """k_col_v for H = 128 * WV (1024, 512, 256): the wave digit has WV values against 8 registers, so the wave exchange hands
    every thread WV values of the wave digit x Q = 8 / WV of the eight first-stage outputs and the second stage is Q butterflies of
    radix WV; everything else as in the H = 1024 kernel.  Forward decimation in time, phase, inverse decimation in frequency."""
    H, Q = 128 * WV, 8 // WV
    rng = np.random.default_rng(10 + WV)
    x = rng.standard_normal(H) + 1j * rng.standard_normal(H)
    tw = lambda j, n=H: np.exp(2j * np.pi * j / n)
    F = np.fft.ifft(x) * H
    k = np.arange(H)
    t = np.exp(-2j * np.pi * k / (2 * H)) * np.where(k < H // 2, 1, -1)
    ref = np.fft.fft(F * t)
    WR = {+1: np.exp(+2j * np.pi * np.outer(np.arange(WV), np.arange(WV)) / WV), -1: np.exp(-2j * np.pi * np.outer(np.arange(WV), np.arange(WV)) / WV)}
    v = np.zeros((WV, 16, 8), complex)                    # [wave][h = lane >> 2][register]; pp = 16 w + h
    for w in range(WV):
        for h in range(16):
            v[w, h, :] = x[16 * w + h + (H // 8) * np.arange(8)]
    v = v @ W8[+1].T                                      # registers: k0
    nv = np.zeros_like(v)                                 # A: element (k0, wave n_w) -> thread wave k0 // Q, register WV (k0 % Q) + n_w
    for nw in range(WV):
        for k0 in range(8):
            nv[k0 // Q, :, WV * (k0 % Q) + nw] = v[nw, :, k0]
    v = nv
    for wp in range(WV):
        for b in range(Q):
            for nw in range(WV):
                v[wp, :, WV * b + nw] *= tw(16 * nw * (Q * wp + b))           # exp(2 pi i n_w k0 / 8 WV): wave-uniform
            v[wp, :, WV * b:WV * b + WV] = v[wp, :, WV * b:WV * b + WV] @ WR[+1].T      # registers: WV b + k1
    kk = lambda wp, s: Q * wp + s // WV + 8 * (s % WV)   # k0 + 8 k1 of a thread whose lane bits 5-3 hold s = WV b + k1
    nv = np.zeros_like(v)                                 # C: register <-> lane bits 5-3 (= h >> 1)
    for h in range(16):
        g, h0 = h >> 1, h & 1
        for s in range(8):
            nv[:, (s << 1) | h0, g] = v[:, h, s]
    v = nv
    for wp in range(WV):
        for h in range(16):
            v[wp, h, :] *= tw(2 * np.arange(8) * kk(wp, h >> 1))               # exp(2 pi i g (k0 + 8 k1) / 64 WV)
    v = v @ W8[+1].T                                      # registers: k2
    nv = np.zeros_like(v)                                 # register bit 2 <-> lane bit 2 (= h & 1)
    for h in range(16):
        s, h0 = h >> 1, h & 1
        for k2 in range(8):
            nv[:, (s << 1) | (k2 >> 2), (h0 << 2) | (k2 & 3)] = v[:, h, k2]
    v = nv
    kt = lambda wp, h: kk(wp, h >> 1) + 32 * WV * (h & 1)
    for wp in range(WV):
        for h in range(16):
            for r in range(4):
                a, b_ = v[wp, h, r], v[wp, h, r + 4] * tw(kt(wp, h)) * tw(8 * WV * r)
                v[wp, h, r], v[wp, h, r + 4] = a + b_, a - b_
            for r in range(4):
                for k3 in range(2):
                    assert abs(v[wp, h, r + 4 * k3] - F[kt(wp, h) + 8 * WV * r + 64 * WV * k3]) <= 1e-9
            base = np.conj(tw(kt(wp, h), 2 * H))          # the phase, where the elements are
            for r in range(4):
                for k3 in range(2):
                    v[wp, h, r + 4 * k3] *= base * np.exp(-2j * np.pi * r / 32) * (1j if k3 else 1)
            for r in range(4):                            # inverse, decimation in frequency: radix 2 over k3
                a, b_ = v[wp, h, r], v[wp, h, r + 4]
                v[wp, h, r], v[wp, h, r + 4] = a + b_, a - b_
    nv = np.zeros_like(v)                                 # bit-2 swap back: registers k2, lane bit 2 = h0
    for h in range(16):
        s, lb2 = h >> 1, h & 1
        for R in range(8):
            nv[:, (s << 1) | (R >> 2), (lb2 << 2) | (R & 3)] = v[:, h, R]
    v = nv
    for h in range(16):
        v[:, h, :] *= np.exp(-2j * np.pi * np.arange(8) * (h & 1) / 16)
    v = v @ W8[-1].T                                      # registers: g
    nv = np.zeros_like(v)                                 # C back: lane bits 5-3 = g, registers s = WV b + k1
    for h in range(16):
        s, h0 = h >> 1, h & 1
        for g in range(8):
            nv[:, (g << 1) | h0, s] = v[:, h, g]
    v = nv
    for wp in range(WV):
        for h in range(16):
            for b in range(Q):
                v[wp, h, WV * b:WV * b + WV] *= np.conj(tw(8 * h * np.arange(WV)))       # exp(-2 pi i k1 h / 16 WV)
                v[wp, h, WV * b:WV * b + WV] = WR[-1] @ v[wp, h, WV * b:WV * b + WV]      # registers: WV b + n_w
    nv = np.zeros_like(v)                                 # A back: thread wave n_w, register k0 = Q wp + b
    for wp in range(WV):
        for b in range(Q):
            for nw in range(WV):
                nv[nw, :, Q * wp + b] = v[wp, :, WV * b + nw]
    v = nv
    for w in range(WV):
        for h in range(16):
            v[w, h, :] *= np.conj(tw((16 * w + h) * np.arange(8)))               # exp(-2 pi i k0 pp / H)
    v = v @ W8[-1].T
    for w in range(WV):
        for h in range(16):
            assert np.abs(v[w, h, :] - ref[16 * w + h + (H // 8) * np.arange(8)]).max() <= 1e-8
