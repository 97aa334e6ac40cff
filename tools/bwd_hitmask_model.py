"""Model of the blend backward's pipeline steps when injection and residency are derived from the strict forward's recorded decision masks
(SampleState::hit) instead of from n_contrib.  Runs the HIP forward on the bench scene, reads the masks out of the sample
buffer and counts, per live bucket:
  current : a pixel is injected when n_contrib > bucket start, rel = min(n_contrib - bstart, 64); four descending classes of 16 rel values
  masks   : injected when its 64-bit mask != 0, rel = 64 - clz(mask) (index of the last entry it blended + 1); same classes
  masks16 : the same with 16 classes of 4
plus what bounds any lane-per-Gaussian pipeline: the share of (pixel, Gaussian) slots that blend at all (popcount), and a model of
QUADRANT-SEGMENTED pipelines (per 8x8 quadrant only the entries some pixel of the quadrant blended; segments packed into rounds of <= 64 lanes).
    python tools/bwd_hitmask_model.py [density] [opacity_shift] [P]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from conftest import make_scene
from gpu_helpers import hip_forward, npy

density = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
oshift = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
P = int(sys.argv[3]) if len(sys.argv) > 3 else 2_000_000
W, H = 1920, 1080
raw, sc, camd, cam = make_scene("random", P, W, H, 3, 0)
if density != 1.0: raw["scaling"] = (raw["scaling"] + float(np.log(density))).contiguous()
if oshift != 0.0: raw["opacity"] = (raw["opacity"] + oshift).contiguous()
f = hip_forward(raw, cam, export=("ranges", "n_contrib", "max_contrib"))
B = int(f["B"])
rg = npy(f["dbg"]["ranges"]).astype(np.int64); nc = npy(f["dbg"]["n_contrib"]).astype(np.int64); mc = npy(f["dbg"]["max_contrib"]).astype(np.int64)
sample = f["bufs"][3]
a256 = lambda x: (x + 255) & ~255
base = a256(sample.data_ptr()) - sample.data_ptr()
o_b2t = base; o_ck = a256(o_b2t + 4 * B); o_hit = a256(o_ck + 16 * 256 * B)
b2t = npy(sample[o_b2t:o_b2t + 4 * B].view(torch.int32)).astype(np.int64)
hit = npy(sample[o_hit:o_hit + 8 * 256 * B].view(torch.int64)).view(np.uint64).reshape(B, 256)
gx, gy = (W + 15) // 16, (H + 15) // 16
T = gx * gy
# n_contrib tile-major in the kernel's element order: element i = quadrant Q = i >> 6, l = i & 63 -> pixel (8 (Q & 1) + (l & 7), 8 (Q >> 1) + (l >> 3))
pad = np.zeros((gy * 16, gx * 16), np.int64); pad[:H, :W] = nc
i = np.arange(256); Q = i >> 6; l = i & 63
ex = 8 * (Q & 1) + (l & 7); ey = 8 * (Q >> 1) + (l >> 3)
tiles = pad.reshape(gy, 16, gx, 16).transpose(0, 2, 1, 3)[:, :, ey, ex].reshape(T, 256)
n = rg[:, 1] - rg[:, 0]; nb = (n + 63) // 64
boff = np.concatenate([[0], np.cumsum(nb)])
assert boff[-1] == B, (boff[-1], B)
bidx = np.arange(B); bstart = (bidx - boff[b2t]) * 64
live = bstart < mc[b2t]
Lb = np.nonzero(live)[0]
print(f"scene sigma x{density:g} logit opacity {oshift:+g}: P={P} R={f['R']} B={B} live buckets {Lb.size}")
ncb = tiles[b2t[Lb]]                                   # [Bl, 256]
rel_cur = np.clip(ncb - bstart[Lb, None], 0, 64)
m = np.where(rel_cur > 0, hit[Lb], np.uint64(0))   # (a forward wave stops writing masks once its pixels are finished: beyond n_contrib they are stale)
# highest set bit + 1 (0 for an empty mask), popcount
hi = (m >> np.uint64(32)).astype(np.uint32); lo = (m & np.uint64(0xffffffff)).astype(np.uint32)
def top32(x):
    r = np.zeros(x.shape, np.int64); y = x.copy()
    for s in (16, 8, 4, 2, 1):
        t = (y >> np.uint32(s)) != 0
        r += np.where(t, s, 0); y = np.where(t, y >> np.uint32(s), y)
    return np.where(x != 0, r + 1, 0)
def pop32(x):
    x = x - ((x >> np.uint32(1)) & np.uint32(0x55555555)); x = (x & np.uint32(0x33333333)) + ((x >> np.uint32(2)) & np.uint32(0x33333333))
    x = (x + (x >> np.uint32(4))) & np.uint32(0x0f0f0f0f); return ((x * np.uint32(0x01010101)) >> np.uint32(24)).astype(np.int64)
rel_msk = np.where(hi != 0, 32 + top32(hi), top32(lo))
pc = pop32(hi) + pop32(lo)
assert (rel_msk <= rel_cur).all(), "a mask bit beyond n_contrib"
def steps(rel, width):
    key = np.where(rel > 0, (rel - 1) // width, -1)
    order = np.argsort(-key, axis=1, kind="stable")
    r = np.take_along_axis(rel, order, 1)
    e = np.where(r > 0, np.arange(256)[None, :] + r, 0).max(1)
    return (e + 1) // 2 * 2          # the kernel runs an even number of steps
s_cur = steps(rel_cur, 16); s_m = steps(rel_msk, 16); s_m16 = steps(rel_msk, 4); s_c16 = steps(rel_cur, 4)
inj_cur = (rel_cur > 0).sum(1); inj_m = (rel_msk > 0).sum(1)
tot = lambda x: x.sum() / 1e6
print(f"injected pixels: current {tot(inj_cur):.2f}M ({inj_cur.mean():.0f} per live bucket)   masks {tot(inj_m):.2f}M ({inj_m.mean():.0f})   "
      f"pixels injected today whose mask is empty: {100 * (1 - inj_m.sum() / inj_cur.sum()):.1f}%")
print(f"pipeline steps: current (4 classes) {tot(s_cur):.2f}M = {s_cur.mean():.0f}/bucket | current rel, 16 classes {tot(s_c16):.2f}M ({100 * s_c16.sum() / s_cur.sum():.1f}%) | "
      f"masks, 4 classes {tot(s_m):.2f}M ({100 * s_m.sum() / s_cur.sum():.1f}%) | masks, 16 classes {tot(s_m16):.2f}M ({100 * s_m16.sum() / s_cur.sum():.1f}%)")
d = rel_cur - rel_msk
inj = rel_cur > 0
print("rel(current) - rel(mask) over the injected pixels: mean %.1f; share with 0: %.1f%%, >= 16: %.1f%%, >= 32: %.1f%%" % (
    d[inj].mean(), 100 * (d[inj] == 0).mean(), 100 * (d[inj] >= 16).mean(), 100 * (d[inj] >= 32).mean()))
h = np.bincount(pc[inj], minlength=65)
print("popcount(mask) of the injected pixels: mean %.1f of 64; histogram by eighths: %s" % (pc[inj].mean(), [int(h[k:k + 8].sum()) for k in range(0, 64, 8)] + [int(h[64])]))
print(f"blended (pixel, Gaussian) pairs {tot(pc):.1f}M = {100 * pc.sum() / (s_cur.sum() * 64):.1f}% of the pipeline's slots (steps x 64), {100 * pc.sum() / (inj_cur.sum() * 64):.1f}% of injected x 64")
# ---- quadrant-segmented pipelines: per quadrant q the entries S_q some pixel of q blended; pipeline q = (pixels of q with a mask) through |S_q| lanes
mq = m.reshape(-1, 4, 64)
Sq = np.bitwise_or.reduce(mq, axis=2)                                  # [Bl, 4] u64
nS = pop32((Sq >> np.uint64(32)).astype(np.uint32)) + pop32((Sq & np.uint64(0xffffffff)).astype(np.uint32))
nq = (mq != 0).sum(2)                                                   # pixels per quadrant with work
print("entries reached per quadrant |S_q|: mean %.1f; pixels with work per quadrant: mean %.1f; sum_q |S_q| per bucket: mean %.1f, <= 64 in %.1f%% of the buckets" % (
    nS.mean(), nq.mean(), nS.sum(1).mean(), 100 * (nS.sum(1) <= 64).mean()))
# greedy packing of the four segments into rounds of <= 64 lanes (order q = 0..3); a round runs max(n_q + |S_q|) steps (no descending-rel order modelled)
seg_steps = np.zeros(Lb.size, np.int64)
lanes = np.zeros(Lb.size, np.int64); cur = np.zeros(Lb.size, np.int64); rounds = np.zeros(Lb.size, np.int64)
for q in range(4):
    need = nS[:, q]; st = np.where(need > 0, nq[:, q] + need, 0)
    newround = (lanes + need > 64)
    seg_steps += np.where(newround, cur, 0); rounds += newround
    cur = np.where(newround, st, np.maximum(cur, st)); lanes = np.where(newround, need, lanes + need)
seg_steps += cur; rounds += 1
print(f"quadrant-segmented pipelines: {tot(seg_steps):.2f}M steps ({100 * seg_steps.sum() / s_cur.sum():.1f}% of current), {rounds.mean():.2f} rounds per bucket; "
      f"perfect packing (sum_q (n_q + |S_q|) |S_q| / 64): {tot(((nq + nS) * nS).sum(1) / 64):.2f}M")
# entries of live buckets that NO pixel of the tile blended: their nine partial sums are exact zeros (a flag byte could stand for the 36-byte row)
S_tile = np.bitwise_or.reduce(m, axis=1)
hit_e = pop32((S_tile >> np.uint64(32)).astype(np.uint32)) + pop32((S_tile & np.uint64(0xffffffff)).astype(np.uint32))
valid_e = np.clip(n[b2t[Lb]] - bstart[Lb], 0, 64)
print(f"entries of live buckets: {valid_e.sum() / 1e6:.2f}M, blended by >= 1 pixel of their tile: {hit_e.sum() / 1e6:.2f}M ({100 * hit_e.sum() / valid_e.sum():.1f}%); "
      f"all-zero rows {100 - 100 * hit_e.sum() / valid_e.sum():.1f}%")
if os.environ.get("HITMASK_ONLY_ROWS") == "1":
    sys.exit(0)
# lane-per-pixel alternative: (entry, quadrant) combinations with at least one blending pixel (each would cost one evaluation + a 64-lane reduction)
print(f"(entry, quadrant) combinations with >= 1 blending pixel: {tot(nS):.2f}M = {100 * nS.sum() / (4 * 64 * Lb.size):.1f}% of all in live buckets; blended pairs per combination: {pc.sum() / max(nS.sum(), 1):.1f} of 64")
# ---- ROW-SCAN decomposition: a wave works on (G entries x 64/G pixels) per step — lane = (entry slot, pixel row); the T / A recurrences over the G
# entries of a row are log-step DPP scans, no fill / drain.  Per pixel block (quadrant or finer) only the entries some pixel of the block blended are
# visited (compacted into groups of G) and only the pixels with a non-empty mask: steps = sum_blocks ceil(|S_blk| / G) * ceil(n_blk / (64 / G)).
def pop64(x):
    return pop32((x >> np.uint64(32)).astype(np.uint32)) + pop32((x & np.uint64(0xffffffff)).astype(np.uint32))
lq = np.arange(64)
blocks = {"quadrant 8x8": np.zeros(64, np.int64), "half quadrant 8x4": lq >> 5, "4x4": ((lq >> 3) >> 2) * 2 + ((lq & 7) >> 2)}
for name, sub in blocks.items():
    nsub = int(sub.max()) + 1
    S = np.zeros((Lb.size, 4, nsub), np.uint64); npx = np.zeros((Lb.size, 4, nsub), np.int64)
    for k in range(nsub):
        sel = mq[:, :, sub == k]
        S[:, :, k] = np.bitwise_or.reduce(sel, axis=2); npx[:, :, k] = (sel != 0).sum(2)
    nSb = pop64(S)
    for G in (8, 16, 32):
        rows = 64 // G
        st = (-(-nSb // G)) * (-(-npx // rows))
        print(f"row-scan, blocks = {name}, {G} entries x {rows} pixels per step: {tot(st):.2f}M steps = {st.sum() / Lb.size:.0f} per live bucket "
              f"({100 * st.sum() / s_cur.sum():.1f}% of the pipeline's steps); entries per block {nSb.mean():.1f}, group passes per bucket {(-(-nSb // G)).sum() / Lb.size:.1f}")
# ---- per-bucket choice between the two kernels (cycle model: pipeline 110 cycles per step; row scan 150 per step + ~250 per quadrant pass + 300 fixed)
nS16 = -(-nS // 16); nq4 = -(-nq // 4)
rs_cycles = (nS16 * nq4).sum(1) * 150 + nS16.sum(1) * 120 + (nq > 0).sum(1) * 130 + 300
pl_cycles = s_cur * 110 + 400
best = np.minimum(rs_cycles, pl_cycles)
print(f"cycle model per live bucket: pipeline {pl_cycles.mean():.0f}, row scan {rs_cycles.mean():.0f}, per-bucket minimum {best.mean():.0f} "
      f"({100 * best.sum() / pl_cycles.sum():.1f}% of the pipeline; row scan wins in {100 * (rs_cycles < pl_cycles).mean():.1f}% of the buckets)")
kb = (bstart[Lb] // 64)
for k in range(0, 8):
    sel = kb == k
    if sel.any():
        print(f"  bucket {k} of its tile: {int(sel.sum())} live buckets, pipeline {pl_cycles[sel].mean():.0f}, row scan {rs_cycles[sel].mean():.0f}, row scan wins in {100 * (rs_cycles[sel] < pl_cycles[sel]).mean():.0f}%, "
              f"injected {inj_cur[sel].mean():.0f}, entries per quadrant {nS[sel].mean():.1f}")
