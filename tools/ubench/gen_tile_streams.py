"""Writes the REAL emission-order tile streams of the 2M / 1080p bench scene for tools/ubench/tile_binning.hip: tiles_morton.bin (the map's rows in
Morton order) and tiles_insertion.bin (rows as generated), uint32 tile id per (Gaussian, tile) instance, from the CPU oracle's preprocess + binning
(test infrastructure; runs without a GPU in about a minute; the .bin files are git-ignored, 24 MB each).
    python tools/ubench/gen_tile_streams.py"""
import sys, os, numpy as np, torch, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import make_scene
import importlib
pkg = importlib.import_module("gaussian-lic_amd")
from oracle.oracle import Oracle
P, W, H = 2_000_000, 1920, 1080
raw, sc, camd, cam = make_scene("random", P, W, H, 3, 0)
o = Oracle(np.float32)
t = time.time()
pre = o.preprocess(sc, camd)
bins = o.binning(pre, W, H)
print("oracle", time.time() - t, "R", bins["R"])
keys = bins["keys"]; gid = bins["point_list"].astype(np.int64)
tile = (keys >> np.uint64(32)).astype(np.uint32)
trainer = importlib.import_module("gaussian-lic_amd.trainer")
xyz = raw["xyz"] if "xyz" in raw else raw["means"]
order = trainer.morton_order(torch.as_tensor(xyz)).numpy()      # row r of the stored map = original row order[r]
rank = np.empty(P, np.int64); rank[order] = np.arange(P)
for name, r in (("morton", rank[gid]), ("insertion", gid)):
    idx = np.lexsort((tile, r))
    tile[idx].astype(np.uint32).tofile(os.path.join(ROOT, "tools", "ubench", f"tiles_{name}.bin"))
    print(name, "written")
