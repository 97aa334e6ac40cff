"""Bytes on the links per optimiser step, dense slab vs visible rows only vs the rank-1 exchange (11 floats all-reduced + 3 floats per
view all-gathered), for the 8 synthetic views of BASELINE config 4 (2M Gaussians,
1080p) and config 5 (5M, 4K): renders each view once on the GPU and counts the visibility masks.   python tools/exchange_bytes.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from conftest import make_scene
from gpu_helpers import hip_forward
for P, W, H in ((2_000_000, 1920, 1080), (5_000_000, 3840, 2160)):
    masks = []
    for view in range(8):
        raw, sc, camd, cam = make_scene("random", P, W, H, 3, 0, view=view)
        f = hip_forward(raw, cam)
        masks.append((f["radii"] > 0).clone())
        del f
    per_view = [int(m.sum()) for m in masks]
    for n in (2, 4, 8):
        union = torch.zeros_like(masks[0])
        for m in masks[:n]:
            union |= m
        u = int(union.sum())
        ring = 2.0 * (n - 1) / n     # bytes a rank sends (and receives) per byte all-reduced on a ring; an all-gather moves (n - 1) shards
        print(f"P={P} {W}x{H} N={n}: visible per view {min(per_view[:n])}..{max(per_view[:n])}, union {u} = {100.0 * u / P:.1f}% of P; "
              f"payload: dense slab {4 * 59 * P / 1e6:.0f} MB, visible rows {4 * 59 * u / 1e6:.0f} MB (+ {P / 1e6:.1f} MB mask bytes); "
              f"sent per rank and step: dense {ring * 4 * 59 * P / 1e6:.0f} MB, visible rows {ring * 4 * 59 * u / 1e6:.0f} MB, "
              f"rank-1 {(ring * 4 * 11 * P + (n - 1) * 12 * P) / 1e6:.0f} MB")
