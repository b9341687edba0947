"""Write a small pickle in the TAP-Vid layout (for exercising tools/test.py --data-root without the real files): a textured
image translated by a constant velocity, tracks = grid points moving with it."""
import pickle, sys
import numpy as np
out = sys.argv[1] if len(sys.argv) > 1 else "/tmp/fake_tapvid.pkl"
rng = np.random.default_rng(0)
vids = {}
for v in range(3):
    T, H, W = 10, 96, 128
    base = rng.integers(0, 255, (H // 8 + 8, W // 8 + 8, 3)).astype(np.float32)
    big = np.kron(base, np.ones((8, 8, 1), dtype=np.float32))
    vx, vy = int(rng.integers(-2, 3)), int(rng.integers(-2, 3))
    frames = np.stack([big[32 - vy * t: 32 - vy * t + H, 32 - vx * t: 32 - vx * t + W] for t in range(T)]).astype(np.uint8)
    px, py = np.meshgrid(np.linspace(40, W - 40, 4), np.linspace(30, H - 30, 3))
    p0 = np.stack([px.ravel(), py.ravel()], -1)                               # (P,2) pixels
    pts = p0[:, None, :] + np.arange(T)[None, :, None] * np.array([vx, vy])[None, None, :]
    vids[f"v{v}"] = dict(video=frames, points=(pts / np.array([W, H])).astype(np.float32), occluded=np.zeros((p0.shape[0], T), dtype=bool))
with open(out, "wb") as f:
    pickle.dump(vids, f)
print(out, len(vids))
