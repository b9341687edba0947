"""Discrete-event model of the key-block ring of fgvc_pair_topk_f16x3: four consumers (the 2 x 2 query blocks of a super-tile) step
through the list of key blocks the super-tile reaches; a consumer spends one tile time on a block its query block reaches and ~0 on
one it does not; block e may be read once it has landed, and its ring slot is refilled (with block e + D) once all four consumers
have left block e - D.  Prints the loop length in tile times for several list orders and ring depths; a workgroup barrier per
block costs one tile time per list entry.   python tools/sim_pair_ring.py [radius]"""
import sys

R = int(sys.argv[1]) if len(sys.argv) > 1 else 15
R2, QBH, QBW = R * R, 4, 8


def reach(qy, qx, ky, kx):
    dy = max(0, ky - (qy + QBH - 1), qy - (ky + QBH - 1))
    dx = max(0, kx - (qx + QBW - 1), qx - (kx + QBW - 1))
    return dy * dy + dx * dx <= R2


qbs = [(0, 0), (4, 0), (0, 8), (4, 8)]
ents = []
for by in range(-R // QBH - 2, R // QBH + 4):
    for bx in range(-R // QBW - 2, R // QBW + 4):
        m = [reach(qy, qx, by * QBH, bx * QBW) for (qy, qx) in qbs]
        if any(m):
            ents.append(((by, bx), m))


def sim(order, D, latency=0.3, skip=0.03):
    n = len(order)
    fin = [[0.0] * n for _ in range(4)]
    for e in range(n):
        ready = (max(fin[w][e - D] for w in range(4)) + latency) if e >= D else latency
        for w in range(4):
            prev = fin[w][e - 1] if e else 0.0
            fin[w][e] = max(prev, ready) + (1.0 if order[e][1][w] else skip)
    return max(fin[w][n - 1] for w in range(4))


row_major = ents
half = (len(ents) + 1) // 2
alternating = [None] * len(ents)
for r, e in enumerate(ents):
    alternating[2 * r if r < half else 2 * (len(ents) - 1 - r) + 1] = e
col_major = sorted(ents, key=lambda e: (e[0][1], e[0][0]))
print(f"radius {R}: {len(ents)} key blocks in the union, {[sum(e[1][i] for e in ents) for i in range(4)]} reached per query block")
print(f"a barrier per block: {len(ents)} tile times; lower bound {max(sum(e[1][i] for e in ents) for i in range(4))}")
for D in (2, 4, 6, 8):
    print(f"ring depth {D}: row-major {sim(row_major, D):.1f}  column-major {sim(col_major, D):.1f}  alternating (first, last, second, ...) {sim(alternating, D):.1f}")
