#!/usr/bin/env python3
"""How often do the sort keys of a BVH node tie? (VERDICT round 5, item 6 proposed a fast path for nodes WITHOUT ties: any correct sort
gives libstdc++'s order there.) For the top nodes of the 871 200-triangle bench mesh, and of the same mesh with every vertex jittered by a
random ~1e-4 (an "irregular" mesh: no two triangles share a box extent by construction), along the axis the build would choose: the number
of adjacent equal keys after a stable sort. CPU only.   python3 tools/bvh_key_ties.py"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from gpuart_amd import synth_scenes as S  # noqa: E402


def top_nodes(tri, name, depth_max=3):
    lo, hi = tri.min(1), tri.max(1)

    def rec(idx, depth):
        n = len(idx)
        l, h = lo[idx].min(0), hi[idx].max(0)
        r = h - l
        ax = 0 if (r[0] >= r[1] and r[0] >= r[2]) else (1 if (r[1] >= r[0] and r[1] >= r[2]) else 2)
        key = (lo[idx, ax] + hi[idx, ax]).astype(np.float32)
        order = np.argsort(key, kind="stable")
        ks = key[order]
        ties = int((ks[1:] == ks[:-1]).sum())
        print("%-10s level %d  %7d triangles  axis %d  adjacent equal keys %7d (%.2f %%)" % (name, depth, n, ax, ties, 100.0 * ties / n), flush=True)
        if depth < depth_max:
            mid = np.float64(l[ax]) + 0.5 * np.float64(r[ax])
            split = int(np.searchsorted(0.5 * ks.astype(np.float64), mid, side="right"))
            split = min(max(split, 1), n - 1)
            rec(idx[order[:split]], depth + 1)
            rec(idx[order[split:]], depth + 1)
    rec(np.arange(len(tri)), 0)


d = S.scene_d(660, 660)
tri = np.array([x[1] for x in d if x[0] == 2], np.float32).reshape(-1, 3, 3)
top_nodes(tri, "bench mesh")
rng = np.random.RandomState(1)
top_nodes((tri + rng.uniform(-1e-4, 1e-4, tri.shape)).astype(np.float32), "jittered", 2)
