"""Ad-hoc: utilisation of the backward systolic pipeline on the bench scene (from n_contrib / ranges exported by the HIP path), and a
step-count model of CHAINED buckets: a wave takes K consecutive global buckets and keeps the pipeline full across their boundaries
(one marker slot per boundary) instead of draining 63 steps after each.    python tools/bwd_util.py [P] [scene]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import make_scene
from gpu_helpers import hip_forward, npy
P = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
kind = sys.argv[2] if len(sys.argv) > 2 else "random"
W, H = 1920, 1080
raw, sc, camd, cam = make_scene(kind, P, W, H, 3, 0)
f = hip_forward(raw, cam, export=("ranges", "n_contrib", "max_contrib"))
rg = npy(f["dbg"]["ranges"]).astype(np.int64); nc = npy(f["dbg"]["n_contrib"]).astype(np.int64); mc = npy(f["dbg"]["max_contrib"]).astype(np.int64)
gx, gy = (W + 15) // 16, (H + 15) // 16
pad = np.zeros((gy * 16, gx * 16), np.int64); pad[:H, :W] = nc
tiles = pad.reshape(gy, 16, gx, 16).transpose(0, 2, 1, 3).reshape(gy * gx, 256)   # [T, 256] in thread_rank order
n = rg[:, 1] - rg[:, 0]
nb = (n + 63) // 64
# per global bucket: tile, running?, injected pixels k, valid lanes
b_tile, b_run, b_k, b_lanes = [], [], [], []
tot_buckets = run_buckets = 0; steps_full = 0; steps_compact = 0; pairs_useful = 0
for t in range(len(n)):
    for b in range(nb[t]):
        tot_buckets += 1
        run = b * 64 < mc[t]
        k = int((tiles[t] > b * 64).sum()) if run else 0
        lanes = int(min(64, n[t] - b * 64))
        b_tile.append(t); b_run.append(run); b_k.append(k); b_lanes.append(lanes)
        if not run: continue
        run_buckets += 1
        steps_full += 256 + 63
        steps_compact += k + lanes - 1
        pairs_useful += int(np.clip(tiles[t] - b * 64, 0, lanes).sum())
print(f"P={P} R={f['R']} B={f['B']} buckets run {run_buckets}/{tot_buckets}; tiles with a running bucket {int((mc > 0).sum())}/{len(n)}")
print(f"steps: full {steps_full/1e6:.2f}M  compacted (current kernel) {steps_compact/1e6:.2f}M = {steps_compact/max(run_buckets,1):.0f} per running bucket")
print(f"useful (pixel,gaussian) pairs {pairs_useful/1e6:.1f}M = {100*pairs_useful/(steps_full*64):.1f}% of full slots, {100*pairs_useful/(steps_compact*64):.1f}% of compacted slots")
kk = np.array([k for k, r in zip(b_k, b_run) if r])
print("injected pixels per running bucket: mean %.0f  p10 %d  p50 %d  p90 %d;  buckets with < 32 px: %.1f%%, < 64 px: %.1f%%" % (
    kk.mean(), np.percentile(kk, 10), np.percentile(kk, 50), np.percentile(kk, 90), 100 * (kk < 32).mean(), 100 * (kk < 64).mean()))
# what ordering the injected pixels by descending remaining depth (rel) would save: a pixel injected at position i needs i + rel_i steps
sorted_steps = 0
for t in range(len(n)):
    for b in range(nb[t]):
        if b * 64 >= mc[t]: continue
        rel = np.clip(tiles[t] - b * 64, 0, 64)
        rel = np.sort(rel[rel > 0])[::-1]
        sorted_steps += int((np.arange(rel.size) + rel).max())
print(f"steps with the pixels of a bucket injected in descending order of their remaining depth: {sorted_steps/1e6:.2f}M ({100*sorted_steps/steps_compact:.0f}% of current)")
for width in (4, 8, 16, 32):   # coarse classes of `width` consecutive rel values, pixel-index order inside a class
    cs = 0
    for t in range(len(n)):
        for b in range(nb[t]):
            if b * 64 >= mc[t]: continue
            rel = np.clip(tiles[t] - b * 64, 0, 64)
            rel = rel[rel > 0]
            order = np.argsort(-((rel - 1) // width), kind="stable")
            cs += int((np.arange(rel.size) + rel[order]).max())
    print(f"  ... in {64 // width} classes of {width} rel values: {cs/1e6:.2f}M ({100*cs/steps_compact:.0f}% of current)")
B = tot_buckets
for K in (1, 2, 4, 8, 16, 32):
    steps = 0; waves = 0; longest = 0; pads = 0; boundaries = 0
    for w0 in range(0, B, K):
        s = 0; inflight = 0; cur_tile = -1; prev_k = None; started = False
        for gb in range(w0, min(B, w0 + K)):
            if not b_run[gb]:
                continue
            if b_tile[gb] != cur_tile and inflight > 0:
                s += inflight; inflight = 0; prev_k = None       # tile switch with pixels in flight: drain
            if started and inflight > 0:
                s += 1; boundaries += 1                          # marker slot
                if prev_k is not None and prev_k + b_k[gb] < 64: pads += 1
            cur_tile = b_tile[gb]
            s += b_k[gb]
            inflight = b_lanes[gb] - 1 if b_k[gb] > 0 else max(inflight - b_k[gb], 0)
            prev_k = b_k[gb]; started = True
        s += inflight
        if s > 0: waves += 1
        steps += s; longest = max(longest, s)
    print(f"chain K={K:2d}: steps {steps/1e6:.2f}M ({100*steps/steps_compact:.0f}% of current)  working waves {waves}  longest wave {longest} steps  "
          f"boundaries {boundaries} (short pairs < 64 px: {pads})")
