"""Seeded ray batches shared by the CPU and GPU parity tests (numpy only)."""
import numpy as np


def camera_rays(cam12, width, height, n, seed):
    """n jittered pinhole rays (float32) over the whole image, plus a block of exact-pixel-centre rays."""
    rng = np.random.default_rng(seed)
    lf, ul, hz, vt = cam12[0:3], cam12[3:6], cam12[6:9], cam12[9:12]
    x = rng.uniform(0, 1, n).astype(np.float32)[:, None]
    y = rng.uniform(0, 1, n).astype(np.float32)[:, None]
    d = (ul + x * hz + y * vt - lf).astype(np.float32)
    d = (d / np.linalg.norm(d.astype(np.float64), axis=1, keepdims=True)).astype(np.float32)
    o = np.tile(lf, (n, 1)).astype(np.float32)
    return o, d


def bounce_rays(o, d, t, hit, seed, eps=1e-4):
    """Incoherent secondary rays leaving the hit points of a primary batch in random directions."""
    rng = np.random.default_rng(seed)
    p = (o[hit] + t[hit, None] * d[hit]).astype(np.float32)
    w = rng.normal(size=p.shape)
    w = (w / np.linalg.norm(w, axis=1, keepdims=True)).astype(np.float32)
    return (p + np.float32(eps) * w).astype(np.float32), w


def axis_aligned_rays(n, seed):
    """Rays with exactly-zero / tiny direction components and origins on box faces (slab-test edge cases)."""
    rng = np.random.default_rng(seed)
    o = rng.uniform(0.05, 0.95, (n, 3)).astype(np.float32)
    o[:, 2] = -o[:, 2]
    d = np.zeros((n, 3), np.float32)
    axis = rng.integers(0, 3, n)
    sign = rng.choice([-1.0, 1.0], n).astype(np.float32)
    d[np.arange(n), axis] = sign
    tiny = rng.choice([0.0, 1e-8, -1e-8, 1e-6], n).astype(np.float32)
    d[np.arange(n), (axis + 1) % 3] = tiny
    d = (d / np.linalg.norm(d.astype(np.float64), axis=1, keepdims=True)).astype(np.float32)
    return o, d
