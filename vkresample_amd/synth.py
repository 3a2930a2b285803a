"""Deterministic synthetic frames: frame k is uint8 RGB [H][W][3], seed 0x5EED0000 + k.
'U' = uniform 0..255 (stress), 'N' = natural-like (low-frequency cosines + noise)."""
import numpy as np

SEED_BASE = 0x5EED0000


def frame(k, width, height, dist="U"):
    rng = np.random.Generator(np.random.PCG64(SEED_BASE + int(k)))
    if dist == "U":
        return rng.integers(0, 256, size=(height, width, 3), dtype=np.uint8)
    y = np.arange(height)[:, None, None] / height
    x = np.arange(width)[None, :, None] / width
    img = np.full((height, width, 3), 128.0)
    for _ in range(8):
        fx, fy = rng.integers(0, 6, size=2)
        ph = rng.uniform(0, 2 * np.pi, size=(1, 1, 3))
        amp = rng.uniform(0.3, 1.0, size=(1, 1, 3))
        img = img + 60.0 / 8 ** 0.5 * amp * np.cos(2 * np.pi * (fx * x + fy * y) + ph)
    img = img + rng.normal(0.0, 4.0, size=img.shape)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8)
