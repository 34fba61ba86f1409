"""Deterministic synthetic inputs of the benchmark configurations (host, numpy)."""
import numpy as np


def synthetic_frame(w, h, seed, nrect=200, sigma=4.0):
    """Random axis-aligned grey rectangles on a mid-grey field, 3x3 box blur,
    Gaussian noise, clipped to u8: yields well over 1000 FAST candidates and
    exercises both FAST thresholds."""
    rng = np.random.default_rng(seed)
    img = np.full((h, w), 128.0, np.float32)
    for _ in range(nrect):
        rw = int(rng.integers(8, max(9, w // 4)))
        rh = int(rng.integers(8, max(9, h // 4)))
        x0 = int(rng.integers(0, max(1, w - rw)))
        y0 = int(rng.integers(0, max(1, h - rh)))
        img[y0:y0 + rh, x0:x0 + rw] = float(rng.integers(0, 256))
    p = np.pad(img, 1, mode="edge")
    blur = sum(p[dy:dy + h, dx:dx + w] for dy in range(3) for dx in range(3)) / 9.0
    blur = blur + rng.normal(0.0, sigma, size=(h, w)).astype(np.float32)
    return np.clip(np.rint(blur), 0, 255).astype(np.uint8)


def synthetic_batch(n, w, h, seed0=1234, **kw):
    return np.stack([synthetic_frame(w, h, seed0 + i, **kw) for i in range(n)])


def synthetic_stereo_pair(w, h, seed, sigma=2.0):
    """Right frame = left frame shifted by a per-row-constant disparity of 5..60 px, plus noise."""
    rng = np.random.default_rng(seed)
    left = synthetic_frame(w, h, seed, sigma=sigma)
    wide = synthetic_frame(w + 64, h, seed + 100000, sigma=sigma)
    wide[:, :w] = left
    right = np.empty_like(left)
    # disparity constant over bands of rows
    band = 40
    for y0 in range(0, h, band):
        d = int(rng.integers(5, 61))
        right[y0:y0 + band] = wide[y0:y0 + band, d:d + w]
    noise = rng.normal(0.0, sigma, size=(h, w))
    right = np.clip(np.rint(right.astype(np.float32) + noise), 0, 255).astype(np.uint8)
    return left, right


def random_descriptors(n, seed):
    rng = np.random.default_rng(seed)
    return rng.integers(0, 256, size=(n, 32), dtype=np.uint8)


def map_descriptors(n, frame_desc, seed=99, planted=0.10, max_flips=30):
    """Uniform random map descriptors with a share of near-duplicates of frame descriptors."""
    rng = np.random.default_rng(seed)
    m = rng.integers(0, 256, size=(n, 32), dtype=np.uint8)
    if frame_desc is not None and len(frame_desc):
        k = int(n * planted)
        where = rng.choice(n, size=k, replace=False)
        src = rng.integers(0, len(frame_desc), size=k)
        cp = frame_desc[src].copy()
        bits = np.unpackbits(cp, axis=1)
        for i in range(k):
            nf = int(rng.integers(0, max_flips + 1))
            if nf:
                pos = rng.choice(256, size=nf, replace=False)
                bits[i, pos] ^= 1
        m[where] = np.packbits(bits, axis=1)
    return m
