"""Index-faithful numpy emulation of the reference's device buffers.  TEST INFRASTRUCTURE ONLY.

Parity pin and its limits: see oracle/fftup_oracle.c (border quirks remain "PARITY UNPINNED"): the reference cannot be built or run here and
ships no golden data.  This module re-creates the three device buffers of VkResample
(`inputBuffer`, `buffer`, `tempBuffer`) as flat arrays and replays, dispatch by dispatch, what
each of the reference's eight kernels reads and writes -- with the reference's strides, the
trailing "DC column", the in-place shift and the zero-padding read guards -- using numpy.fft only
for the 1-D transforms themselves.  It exists to pin the *closed-form* C oracle
(oracle/fftup_oracle.c), which uses its own simple arrays, against the reference's actual
memory layout; never-written elements are NaN so that any stale read shows up.

VR = /root/reference/VkResample.cpp, VF = /root/reference/vkFFT/vkFFT.h.
Small sizes only (pure numpy, float64).
"""
import numpy as np


def _dft_plus(x):
    """sum_n x[n] exp(+2 pi i nk/N): the reference's *forward* kernel (VF:4545, VF:751)."""
    return np.fft.ifft(x) * len(x)


def _dft_minus(x):
    """sum_n x[n] exp(-2 pi i nk/N): the reference's inverse kernel, before the 1/N of VF:2921-2923."""
    return np.fft.fft(x)


def out_dims(W, H, upscale):
    u = np.float32(upscale)
    return int(np.uint32(u * np.float32(W))), int(np.uint32(u * np.float32(H)))   # VR:1417-1418


def emulate(planes, upscale):
    """planes: float64 [3][H][W] (already load-converted).  Returns pre-sharpen R as [3][uH][uW]
    exactly as the reference's C2R kernel leaves it in `tempBuffer` (plus the raw tempBuffer)."""
    planes = np.asarray(planes, dtype=np.float64)
    C, H, W = planes.shape
    uW, uH = out_dims(W, H, upscale)
    u = np.float32(upscale)

    # ---- inputBuffer: real, row stride W, plane stride (W+2)*H   (VR:1644, VF:6337-6338)
    in_ps = (W + 2) * H
    inputBuffer = np.full(C * in_ps, np.nan)
    for c in range(C):
        for y in range(H):
            inputBuffer[c * in_ps + y * W: c * in_ps + y * W + W] = planes[c, y]

    # ---- buffer: complex, row stride uW/2, plane stride (uW/2+1)*uH, DC column at uW*uH/2
    rs = uW // 2
    ps = (uW // 2 + 1) * uH                     # VR:1553
    dc = uW * uH // 2                           # VF:4299, VF:5446
    buffer = np.full(C * ps, np.nan + 1j * np.nan, dtype=np.complex128)

    # F0: R2C rows (type 5).  WG y = j handles input rows 2j, 2j+1 (VF:1945-2058); writes
    #     A[k] -> (col k-1, row 2j), B[k] -> (col k-1, row 2j+1), k = 1..W/2 (VF:4320-4350);
    #     DC of both rows -> DC column rows 2j, 2j+1, imaginary part 0 (VF:4292-4312).
    for c in range(C):
        for j in range(H // 2):
            a = inputBuffer[c * in_ps + (2 * j) * W: c * in_ps + (2 * j) * W + W]
            b = inputBuffer[c * in_ps + (2 * j + 1) * W: c * in_ps + (2 * j + 1) * W + W]
            Z = _dft_plus(a + 1j * b)
            for k in range(1, W // 2 + 1):
                zk, zn = Z[k], Z[W - k]
                A = 0.5 * complex(zk.real + zn.real, zk.imag - zn.imag)
                B = 0.5 * complex(zk.imag + zn.imag, -zk.real + zn.real)
                buffer[c * ps + (k - 1) + (2 * j) * rs] = A
                buffer[c * ps + (k - 1) + (2 * j + 1) * rs] = B
            buffer[c * ps + dc + 2 * j] = complex(Z[0].real, 0.0)
            buffer[c * ps + dc + 2 * j + 1] = complex(Z[0].imag, 0.0)

    # F1: DC-column FFT, contiguous, length H (VF:5190-6040, offset VF:5446-5447)
    for c in range(C):
        s = c * ps + dc
        buffer[s: s + H] = _dft_plus(buffer[s: s + H])
    # F2: column FFT over columns 0..W/2-1, stride uW/2 (VF:1656-1717)
    for c in range(C):
        for col in range(W // 2):
            idx = c * ps + col + np.arange(H) * rs
            buffer[idx] = _dft_plus(buffer[idx])

    # S: shift (VR:514-526) with size = (W/2, H), inputStride = (uW/2, uH, ps) (VR:1515-1553).
    #    In place; for uH >= 1.5 H source and destination rows are disjoint.
    sz0, sz1 = W // 2, H
    st0, st1 = rs, uH
    snapshot = buffer.copy()      # all invocations read "inputs" = same buffer; disjoint for u>=1.5
    for c in range(C):
        for t in range(sz1 // 2):                         # first branch: gx + gy*size0 < size1/2
            buffer[c * ps + (st1 - 1 - t) + st1 * st0] = snapshot[c * ps + (sz1 - 1 - t) + st1 * st0]
        for gy in range(sz1 // 2):                        # second branch
            for gx in range(sz0):
                buffer[c * ps + gx + (st1 - 1 - gy) * st0] = snapshot[c * ps + gx + (sz1 - 1 - gy) * st0]

    # zero-padding ranges of the inverse plan (VR:1491-1495), float arithmetic then uint32
    zlx, zrx = W // 2, uW // 2
    zly = int(np.uint32(np.float32(uH) / (np.float32(2) * u)))
    zry = int(np.uint32((np.float32(2) * u - np.float32(1)) * np.float32(uH) / (np.float32(2) * u)))

    def guarded(col_vals):
        v = col_vals.copy()
        v[zly:zry] = 0.0                                   # read guard VF:1670-1695 / VF:1536-1576
        return v

    # I0: DC-column inverse FFT of length uH with the read guard (VF:5751-5758), /uH (VF:2921-2923)
    for c in range(C):
        s = c * ps + dc
        buffer[s: s + uH] = _dft_minus(guarded(buffer[s: s + uH])) / uH
    # I1: columns [0, W/2) only -- columns in [zlx, zrx) are skipped entirely (VF:1284-1295)
    for c in range(C):
        for col in range(uW // 2):
            if zlx <= col < zrx:
                continue
            idx = c * ps + col + np.arange(uH) * rs
            buffer[idx] = _dft_minus(guarded(buffer[idx])) / uH

    # I2: C2R rows (type 6, VF:2059-2201) -> tempBuffer real, row stride uW, plane stride (uW+2)*uH
    t_ps = (uW + 2) * uH
    tempBuffer = np.full(C * t_ps, np.nan)
    for c in range(C):
        for j in range(uH // 2):
            sd = np.zeros(uW, dtype=np.complex128)
            for col in range(uW // 2):
                if col < zlx or col >= zrx:
                    t0 = buffer[c * ps + col + (2 * j) * rs]
                    t1 = buffer[c * ps + col + (2 * j + 1) * rs]
                else:
                    t0 = t1 = 0j
                sd[col + 1] = complex(t0.real - t1.imag, t0.imag + t1.real)
                sd[uW - col - 1] = complex(t0.real + t1.imag, -t0.imag + t1.real)
            t0 = buffer[c * ps + dc + 2 * j]                 # VF:2110-2131
            t1 = buffer[c * ps + dc + 2 * j + 1]
            sd[0] = complex(t0.real - t1.imag, t0.imag + t1.real)
            z = _dft_minus(sd) / uW
            tempBuffer[c * t_ps + (2 * j) * uW: c * t_ps + (2 * j) * uW + uW] = z.real
            tempBuffer[c * t_ps + (2 * j + 1) * uW: c * t_ps + (2 * j + 1) * uW + uW] = z.imag

    R = np.empty((C, uH, uW))
    for c in range(C):
        R[c] = tempBuffer[c * t_ps: c * t_ps + uW * uH].reshape(uH, uW)
    return R, tempBuffer


def closed_form(planes, upscale):
    """SURVEY App. A.3 closed form in numpy convention (rfft2 / pad / irfft2 + row-pair DC leak)."""
    planes = np.asarray(planes, dtype=np.float64)
    C, H, W = planes.shape
    uW, uH = out_dims(W, H, upscale)
    u = float(np.float32(upscale))
    out = np.empty((C, uH, uW))
    y = np.arange(uH)
    for c in range(C):
        S = np.fft.rfft2(planes[c])
        G = np.zeros((uH, uW // 2 + 1), dtype=np.complex128)
        G[:H // 2, :W // 2 + 1] = S[:H // 2]
        G[uH - H // 2:, :W // 2 + 1] = S[H // 2:]
        g = np.fft.irfft2(G, s=(uH, uW))
        q = S[H // 2, 0].real * np.sin(np.pi * y / u) / (uH * uW)
        g[0::2, :] -= q[1::2, None]
        g[1::2, :] += q[0::2, None]
        out[c] = g
    return out


# ------------------------------------------------------------------ the non-R2C path (VR:1424 false), SURVEY 8 f4
def emulate_complex(planes, upscale):
    """The reference's buffers when performR2C is false (uW > 8192, or > 4096 for -p 1): `inputBuffer` complex with
    strides (W, H) whose imaginary parts the host never writes (VR:1620, 1647 -- DEFINED as 0 here), `buffer` complex
    with strides (uW, uH); forward C2C, the four-quadrant shift shader replayed invocation by invocation with its own
    index arithmetic (VR:527-546), inverse C2C with the read guards of VR:1497-1502.  Returns the complex image the
    sharpen shader reads, [3][uH][uW]."""
    planes = np.asarray(planes, dtype=np.float64)
    C, H, W = planes.shape
    uW, uH = out_dims(W, H, upscale)
    u = np.float32(upscale)
    ps = uW * uH                                         # VR:1556: inputStride[2] = bufferStride[0] * bufferStride[1]
    buffer = np.full(C * ps, np.nan + 1j * np.nan, dtype=np.complex128)
    for c in range(C):
        F = np.empty((H, W), dtype=np.complex128)
        for y in range(H):
            F[y] = _dft_plus(planes[c, y] + 0j)
        for x in range(W):
            F[:, x] = _dft_plus(F[:, x])
        for y in range(H):
            buffer[c * ps + y * uW: c * ps + y * uW + W] = F[y]
    snapshot = buffer.copy()                             # every invocation reads "inputs" of the same buffer

    def index(c, ix, iy):
        return ix + iy * uW + c * ps
    s0, s1 = W, H                                        # appShift.size (VR:1518-1519)
    for c in range(C):
        for gy in range(s1):
            for gx in range(s0):
                if not (gx >= s0 // 2 or gy >= s1 // 2):
                    continue
                if gx >= s0 // 2 and gy < s1 // 2:
                    i_in, i_out = index(c, 3 * s0 // 2 - 1 - gx, gy), index(c, uW + s0 // 2 - 1 - gx, gy)
                if gx >= s0 // 2 and gy >= s1 // 2:
                    i_in = index(c, 3 * s0 // 2 - 1 - gx, 3 * s1 // 2 - 1 - gy)
                    i_out = index(c, uW + s0 // 2 - 1 - gx, uH + s1 // 2 - 1 - gy)
                if gx < s0 // 2 and gy >= s1 // 2:
                    i_in, i_out = index(c, gx, 3 * s1 // 2 - 1 - gy), index(c, gx, uH + s1 // 2 - 1 - gy)
                buffer[i_out] = snapshot[i_in]
    zlx = W // 2
    zrx = int(np.uint32((np.float32(2) * u - np.float32(1)) * np.float32(uW) / (np.float32(2) * u)))
    zly = int(np.uint32(np.float32(uH) / (np.float32(2) * u)))
    zry = int(np.uint32((np.float32(2) * u - np.float32(1)) * np.float32(uH) / (np.float32(2) * u)))
    out = np.empty((C, uH, uW), dtype=np.complex128)
    for c in range(C):
        B = buffer[c * ps:(c + 1) * ps].reshape(uH, uW).copy()
        for x in range(uW):
            if zlx <= x < zrx:
                continue                                 # sequences consisting of zeros only are skipped
            col = B[:, x].copy()
            col[zly:zry] = 0.0
            B[:, x] = _dft_minus(col) / uH
        for y in range(uH):
            row = B[y].copy()
            row[zlx:zrx] = 0.0
            B[y] = _dft_minus(row) / uW
        out[c] = B
    return out


def closed_form_complex(planes, upscale):
    """the same as plain numpy: fft2 (numpy's sign is the reference's inverse: conjugate), quadrant placement, ifft2"""
    planes = np.asarray(planes, dtype=np.float64)
    C, H, W = planes.shape
    uW, uH = out_dims(W, H, upscale)
    out = np.empty((C, uH, uW), dtype=np.complex128)
    for c in range(C):
        F = np.conj(np.fft.fft2(planes[c]))              # sum x exp(+2 pi i ..) of a real image
        G = np.zeros((uH, uW), dtype=np.complex128)
        G[:H // 2, :W // 2] = F[:H // 2, :W // 2]
        G[:H // 2, uW - W // 2:] = F[:H // 2, W // 2:]
        G[uH - H // 2:, :W // 2] = F[H // 2:, :W // 2]
        G[uH - H // 2:, uW - W // 2:] = F[H // 2:, W // 2:]
        out[c] = np.fft.fft2(G) / (uW * uH)              # exp(-2 pi i ..), 1/N
    return out
