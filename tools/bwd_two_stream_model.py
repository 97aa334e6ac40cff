"""Ad-hoc model: pipeline steps of the backward blend if a wave ran TWO 32-lane pipelines (two pixel streams through the bucket's first
32 Gaussians, then — only for the pixels whose last contributor lies deeper — through the other 32), against the current one 64-lane
pipeline with four injection classes.   python tools/bwd_two_stream_model.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from conftest import make_scene
from gpu_helpers import hip_forward, npy
P, W, H = 2_000_000, 1920, 1080
raw, sc, camd, cam = make_scene("random", P, W, H, 3, 0)
f = hip_forward(raw, cam, export=("ranges", "n_contrib", "max_contrib"))
rg = npy(f["dbg"]["ranges"]).astype(np.int64); nc = npy(f["dbg"]["n_contrib"]).astype(np.int64); mc = npy(f["dbg"]["max_contrib"]).astype(np.int64)
gx, gy = (W + 15) // 16, (H + 15) // 16
pad = np.zeros((gy * 16, gx * 16), np.int64); pad[:H, :W] = nc
tiles = pad.reshape(gy, 16, gx, 16).transpose(0, 2, 1, 3).reshape(gy * gx, 256)
n = rg[:, 1] - rg[:, 0]; nb = (n + 63) // 64
cur = two = two_sorted = 0; run = 0; n2_tot = 0; ninj_tot = 0
for t in range(len(n)):
    for b in range(nb[t]):
        if b * 64 >= mc[t]: continue
        rel = np.clip(tiles[t] - b * 64, 0, 64); rel = rel[rel > 0]
        if rel.size == 0: continue
        run += 1
        order = np.argsort(-((rel - 1) // 16), kind="stable")
        r = rel[order]; i = np.arange(r.size)
        cur += int((i + r).max())
        s1 = int((i // 2 + np.minimum(r, 32)).max())
        deep = r > 32                       # (the deep pixels are the first injected: classes 3 and 2)
        r2 = r[deep] - 32; j = np.arange(r2.size)
        s2 = int((j // 2 + r2).max()) if r2.size else 0
        two += s1 + s2
        n2_tot += int(r2.size); ninj_tot += int(r.size)
print(f"running buckets {run}; injected pixels {ninj_tot/1e6:.2f}M of which {100*n2_tot/ninj_tot:.0f}% reach the second half")
print(f"steps: one 64-lane pipeline (current) {cur/1e6:.2f}M = {cur/run:.0f} per bucket; two 32-lane pipelines, two phases {two/1e6:.2f}M = {two/run:.0f} per bucket ({100*two/cur:.0f}%)")
