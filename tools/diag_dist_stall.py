"""Round 6: what is the one-off 36-39 ms stall at the ~79th step of a process group (profiles/r04_dist_warmup_stall.txt) tied to?
A ONE-rank RCCL group (collectives are copies), 2M Gaussians / 1080p.  Three experiments in one process, chosen by argv[1]:

  count   N tiny all-reduces first (argv[2], default 400; 4 bytes each, every one synchronised and timed), THEN 160 steps of the N > 1 path.
          If the stall belongs to the number of collectives a process group has issued (c10d / RCCL / HIP event or signal pools growing at
          ~240), it shows among the tiny collectives and the steps behind them run clean: priming = a few hundred empty collectives at start-up
          (milliseconds) instead of 90 full steps.
  steps   160 steps as they are (the control: the stall at step ~78).
  async   the tiny all-reduces issued with async_op=True and waited for (the step's own pattern), then the steps.
  snap    torch.cuda.memory_stats() around every step (allocator counters) — this run showed NO stall, which pointed away from HIP / RCCL:
  gcfreeze / gcoff   gc.collect() + gc.freeze() before the loop / the cyclic collector off: every step reports the collector passes that ran inside it.

    GSLIC_FORCE_DIST=1 MASTER_ADDR=127.0.0.1 MASTER_PORT=29571 RANK=0 WORLD_SIZE=1 python tools/diag_dist_stall.py count 400"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

mode = sys.argv[1] if len(sys.argv) > 1 else "count"
n_tiny = int(sys.argv[2]) if len(sys.argv) > 2 else 400
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
torch.distributed.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
import gaussian_lic_amd  # noqa: E402,F401
from gaussian_lic_amd import trainer  # noqa: E402
from gaussian_lic_amd.camera import synthetic_camera  # noqa: E402
from gaussian_lic_amd.synthetic import gt_image, random_scene  # noqa: E402

W, H, P = 1920, 1080, 2_000_000
model = trainer.GaussianModel(random_scene(P, W, H, 3, 0), dev)
model.training_setup({k: v * 0.01 for k, v in trainer.DEFAULT_LRS.items()})
cam = synthetic_camera(W, H).to_device(dev)
gt = gt_image(H, W).to(dev)
bg = torch.zeros(3, device=dev)


def slow(ts):
    med = sorted(ts)[len(ts) // 2]
    return [(i, round(t, 3)) for i, t in enumerate(ts) if t > max(3.0 * med, med + 1.0)], round(med, 4)


if mode in ("count", "async"):
    x = torch.ones(1, device=dev)
    ts = []
    for i in range(n_tiny):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if mode == "async":
            w = torch.distributed.all_reduce(x, async_op=True)
            w.wait()
        else:
            torch.distributed.all_reduce(x)
        torch.cuda.synchronize()
        ts.append(1e3 * (time.perf_counter() - t0))
    s, med = slow(ts)
    print(f"{n_tiny} tiny all-reduces ({mode}): median {med} ms, total {sum(ts):.1f} ms, slow ones (index, ms): {s}")
import gc
if mode == "gcfreeze":     # hypothesis (round 6): the stall is ONE full (generation-2) pass of Python's cyclic garbage collector over the torch process's heap
    gc.collect()
    gc.freeze()
elif mode == "gcoff":
    gc.disable()
gc_events = []
gc.callbacks.append(lambda phase, info: gc_events.append((phase, info.get("generation"), time.perf_counter())))
SNAP = mode == "snap"
ts, mem = [], []
keys = ("num_device_alloc", "num_device_free", "num_alloc_retries", "num_sync_all_streams", "reserved_bytes.all.current", "segment.all.current")


def snap():
    st = torch.cuda.memory_stats()
    return tuple(int(st.get(k, 0)) for k in keys)


for i in range(160):
    torch.cuda.synchronize()
    m0 = snap() if SNAP else None
    n_ev = len(gc_events)
    t0 = time.perf_counter()
    trainer.training_step_fused(model, cam, gt, bg)
    torch.cuda.synchronize()
    ts.append(1e3 * (time.perf_counter() - t0))
    mem.append((m0, snap() if SNAP else None, [(p_, g_) for p_, g_, _ in gc_events[n_ev:]]))
s, med = slow(ts[1:])
print(f"160 steps of the N > 1 path behind them: median {med} ms, slow ones (index after step 0, ms): {s}; step 0 {ts[0]:.2f} ms")
print("allocator counters", keys)
for i, (a, b, ev) in enumerate(mem):
    if (SNAP and (a != b or i in (0, 1, 2))) or any(g_ == 2 for _, g_ in ev) or ts[i] > 3.0 * med:
        print(f"  step {i}: {a} -> {b}  ({ts[i]:.2f} ms)  garbage-collector passes inside the step (phase, generation): {ev}")
print("gc passes during the 160 steps by generation:", {g_: sum(1 for p_, gg, _ in gc_events if p_ == 'start' and gg == g_) for g_ in (0, 1, 2)}, " gc thresholds", gc.get_threshold())
print("TORCH_NCCL_AVOID_RECORD_STREAMS =", os.environ.get("TORCH_NCCL_AVOID_RECORD_STREAMS"), " PYTORCH_HIP_ALLOC_CONF =", os.environ.get("PYTORCH_HIP_ALLOC_CONF"))
torch.distributed.destroy_process_group()
