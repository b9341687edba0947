"""Deterministic synthetic video clips for fixtures, in pure integer arithmetic (numpy int64): the same bytes on every platform and
library version, so a fixture only has to store what the REFERENCE made of them.  Data generation, nothing of the reference."""
import numpy as np


def _lcg(n, seed):
    """n pseudo-random 31-bit integers (a 64-bit linear congruential generator, top bits)."""
    out = np.empty(n, np.int64)
    x = np.uint64(seed * 2654435761 % (1 << 63) + 88172645463325252)
    a, c = np.uint64(6364136223846793005), np.uint64(1442695040888963407)
    with np.errstate(over="ignore"):
        for i in range(n):
            x = x * a + c
            out[i] = int(x >> np.uint64(33))
    return out


def moving_texture(T, h, w, seed, drift=(3, -2), cell=8, amp=96, noise=8):
    """(T, 3, h, w) int8: a smooth random texture (a coarse grid of values in [-amp, amp], bilinearly interpolated in integers) that moves
    by `drift` = (dx, dy) pixels per frame, plus per-pixel integer noise in [-noise, noise].  Frame values / 32 are the floats fed to a
    network (a 1/32 grid in [-4, 4))."""
    pad = (abs(drift[0]) + abs(drift[1])) * T + cell
    H, W = h + 2 * pad, w + 2 * pad
    gh, gw = H // cell + 2, W // cell + 2
    grid = (_lcg(3 * gh * gw, seed) % (2 * amp + 1) - amp).reshape(3, gh, gw)
    yy, xx = np.arange(H), np.arange(W)
    y0, fy = yy // cell, yy % cell
    x0, fx = xx // cell, xx % cell
    g00 = grid[:, y0][:, :, x0]
    g01 = grid[:, y0][:, :, x0 + 1]
    g10 = grid[:, y0 + 1][:, :, x0]
    g11 = grid[:, y0 + 1][:, :, x0 + 1]
    wy, wx = fy[None, :, None], fx[None, None, :]
    canvas = ((cell - wy) * ((cell - wx) * g00 + wx * g01) + wy * ((cell - wx) * g10 + wx * g11)) // (cell * cell)      # (3, H, W)
    out = np.empty((T, 3, h, w), np.int64)
    ys, xs = np.arange(h)[:, None], np.arange(w)[None, :]
    for t in range(T):
        oy, ox = pad - drift[1] * t, pad - drift[0] * t          # the content moves by +drift per frame
        hsh = (ys * 73856093) ^ (xs * 19349663) ^ ((t + 1) * 83492791) ^ (seed * 2971215073)
        for c in range(3):
            nz = ((hsh * (c + 1) * 2246822519) >> 7) % (2 * noise + 1) - noise
            out[t, c] = canvas[c, oy:oy + h, ox:ox + w] + nz
    return np.clip(out, -128, 127).astype(np.int8)


def lab_like(T, h, w, seed, drift=(3, -2)):
    """(T, 3, h, w) int8 on a 1/32 grid inside the range of the reference's input pipeline (configs/eval/base_data.py:1-7: RGB -> Lab ->
    (x - [50, 0, 0]) / [50, 127, 127]): channel 0 (L) within [-1, 1], channels 1, 2 (a, b) within [-0.45, 0.45]; the moving texture above
    at a smaller amplitude."""
    x = moving_texture(T, h, w, seed, drift=drift, amp=29, noise=3).astype(np.int64)
    x[:, 1:] = (x[:, 1:] * 7) >> 4
    return x.astype(np.int8)
