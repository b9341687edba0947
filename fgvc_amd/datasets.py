"""TAP-Vid sample format (mmpt/datasets/tapvid.py:122-174) on synthetic clips, plus the strided per-rank sampler
(mmpt/datasets/samplers/distributed_sampler.py:53).  The real TAP-Vid / JHMDB files are not available offline;
`SyntheticTapVid` produces tensors with exactly the shapes, dtypes and conventions the model consumes:
    rgbs (1,T,3,h,w) float32 (stands for Lab-normalised frames), query_points (1,P,3) = (t,x,y) float32,
    trajectories (1,T,P,2) float32, visibilities (1,T,P) float32.
"""
from __future__ import annotations

import torch


class SyntheticTapVid:
    """Moving-texture clips with known point tracks: every frame is the first frame shifted by an integer
    (dx,dy) per frame, so ground-truth trajectories are exact and a tracker's accuracy is measurable."""

    def __init__(self, n_videos=4, frames=8, size=(256, 256), points=8, query_mode="first", seed=0, device="cpu"):
        self.n, self.T, self.h, self.w, self.P = n_videos, frames, size[0], size[1], points
        self.query_mode, self.seed, self.device = query_mode, seed, device

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 1000 + i)
        T, h, w, P = self.T, self.h, self.w, self.P
        pad = 2 * T
        base = torch.nn.functional.interpolate(torch.randn(1, 3, (h + 2 * pad) // 8 + 1, (w + 2 * pad) // 8 + 1, generator=g),
                                               size=(h + 2 * pad, w + 2 * pad), mode="bilinear", align_corners=False)[0]
        base = base + 0.25 * torch.randn(base.shape, generator=g)
        vx, vy = int(torch.randint(-2, 3, (1,), generator=g)), int(torch.randint(-2, 3, (1,), generator=g))
        rgbs = torch.stack([base[:, pad - vy * t: pad - vy * t + h, pad - vx * t: pad - vx * t + w] for t in range(T)], 0)
        t0 = torch.zeros(P) if self.query_mode == "first" else torch.randint(0, max(1, T // 2), (P,), generator=g).float()
        margin = 2 * T + 8
        x0 = torch.rand(P, generator=g) * (w - 2 * margin) + margin
        y0 = torch.rand(P, generator=g) * (h - 2 * margin) + margin
        ts = torch.arange(T).view(T, 1).float()
        traj = torch.stack([x0.view(1, P) + vx * (ts - t0.view(1, P)) + 0 * ts, y0.view(1, P) + vy * (ts - t0.view(1, P))], -1)
        # express the query at its own time: position at t0 is (x0,y0)
        qp = torch.stack([t0, x0, y0], -1)
        vis = (ts >= t0.view(1, P)).float()
        d = self.device
        return dict(rgbs=rgbs.unsqueeze(0).to(d), query_points=qp.unsqueeze(0).to(d),
                    trajectories=traj.unsqueeze(0).to(d), visibilities=vis.unsqueeze(0).to(d))


class StridedLoader:
    """indices[rank::world], one video per step (samples_per_gpu=1, tools/test.py:127)."""

    def __init__(self, dataset, rank=0, world=1):
        self.dataset, self.rank, self.world = dataset, rank, world
        self.total = len(dataset)

    def __iter__(self):
        for i in range(self.rank, len(self.dataset), self.world):
            yield self.dataset[i]

    def __len__(self):
        return len(range(self.rank, len(self.dataset), self.world))
