import numpy as np


def rel_rms(a, b):
    """Relative RMS over pixels: ||a-b||_2 / ||b||_2 (the image-level tolerance of DESIGN.md)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.sqrt(np.mean(b ** 2))
    num = np.sqrt(np.mean((a - b) ** 2))
    return num / den if den > 0 else num


def random_rays(rng, n, center, radius):
    """Rays from points on a sphere around `center` aimed at jittered points inside it."""
    d = rng.normal(size=(n, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    o = center + radius * d
    tgt = center + rng.uniform(-0.6, 0.6, size=(n, 3)) * radius
    dd = tgt - o
    dd /= np.linalg.norm(dd, axis=1, keepdims=True)
    return o, dd
