"""Ad-hoc: utilisation of the backward systolic pipeline on the bench scene (from n_contrib / ranges exported by the HIP path)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import make_scene
from gpu_helpers import hip_forward, npy
P = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
W, H = 1920, 1080
raw, sc, camd, cam = make_scene("random", P, W, H, 3, 0)
f = hip_forward(raw, cam, export=("ranges", "n_contrib", "max_contrib"))
rg = npy(f["dbg"]["ranges"]).astype(np.int64); nc = npy(f["dbg"]["n_contrib"]).astype(np.int64); mc = npy(f["dbg"]["max_contrib"]).astype(np.int64)
gx, gy = (W + 15) // 16, (H + 15) // 16
pad = np.zeros((gy * 16, gx * 16), np.int64); pad[:H, :W] = nc
tiles = pad.reshape(gy, 16, gx, 16).transpose(0, 2, 1, 3).reshape(gy * gx, 256)   # [T, 256] in thread_rank order
n = rg[:, 1] - rg[:, 0]
nb = (n + 63) // 64
tot_buckets = run_buckets = 0; steps_full = 0; steps_compact = 0; pairs_useful = 0; active_steps_now = 0
for t in range(len(n)):
    for b in range(nb[t]):
        tot_buckets += 1
        if b * 64 >= mc[t]: continue
        run_buckets += 1
        act = tiles[t] > b * 64                      # pixel reaches this bucket
        k = int(act.sum())
        lanes = min(64, n[t] - b * 64)
        steps_full += 256 + 63
        steps_compact += k + lanes - 1
        pairs_useful += int(np.clip(tiles[t] - b * 64, 0, lanes).sum())
        idx = np.nonzero(act)[0]
        cover = np.zeros(256 + 64, bool)
        for p in idx: cover[p:p + 64] = True
        active_steps_now += int(cover.sum())
print(f"P={P} R={f['R']} B={f['B']} buckets run {run_buckets}/{tot_buckets}")
print(f"steps: full {steps_full/1e6:.2f}M  steps-with-any-active-lane {active_steps_now/1e6:.2f}M  compacted {steps_compact/1e6:.2f}M")
print(f"useful (pixel,gaussian) pairs {pairs_useful/1e6:.1f}M = {100*pairs_useful/(steps_full*64):.1f}% of full slots, {100*pairs_useful/(steps_compact*64):.1f}% of compacted slots")
