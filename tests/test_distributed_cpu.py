"""CPU, world_size 2, gloo: the one exchange step of the N-GPU path — SUM of the gradient slab and
MAX (= OR) of the visibility mask — and the view sharding rule of bench.py."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    import gaussian_lic_amd  # noqa: F401
    from gaussian_lic_amd import trainer
    P = 257
    g = torch.Generator().manual_seed(100 + rank)
    shapes = [(P, 3), (P, 1, 3), (P, 15, 3), (P, 1), (P, 3), (P, 4)]   # the six groups of gaussian.cpp:399-418
    grads = [torch.randn(*s, generator=g) for s in shapes]
    vis = torch.rand(P, generator=g) < 0.3
    red, rvis = trainer.allreduce_gradients(grads, vis)
    # the zero-copy variant used by the fused step: gradients already live in one slab
    class _M:
        NAMES = ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation")
        device = torch.device("cpu")
        P = 257
        def parameters(self): return [torch.empty(*s) for s in shapes]
    slab = trainer.GradSlab(_M())
    for v, x in zip(slab.grads(_M()), grads):
        v.copy_(x)
    svis = trainer.allreduce_slab(slab, vis)
    for a, b in zip(slab.grads(_M()), red):
        assert torch.equal(a, b)
    assert torch.equal(svis, rvis)
    # GSLIC_EXCHANGE=single: slab AND mask in ONE all-reduce (north_star's wording): same sums bit for bit (two addends), mask = OR
    for v, x in zip(slab.grads(_M()), grads):
        v.copy_(x)
    calls = []
    real_all_reduce = torch.distributed.all_reduce
    torch.distributed.all_reduce = lambda *a, **k: (calls.append(1), real_all_reduce(*a, **k))[1]
    try:
        onevis = trainer.allreduce_slab_single(slab, vis)
    finally:
        torch.distributed.all_reduce = real_all_reduce
    assert len(calls) == 1                      # exactly one collective
    for a, b in zip(slab.grads(_M()), red):
        assert torch.equal(a, b)
    assert onevis.dtype == torch.bool and torch.equal(onevis, rvis)
    # the pipelined variant of the fused step: three asynchronous segment all-reduces, consumed in order
    for v, x in zip(slab.grads(_M()), grads):
        v.copy_(x)
    pvis, works = trainer.allreduce_slab_async(slab, vis, _M())
    seen = []
    for work, idx in works:
        work.wait()
        seen += idx
        for i in idx:
            assert torch.equal(slab.grads(_M())[i], red[i])
    assert sorted(seen) == [0, 1, 2, 3, 4, 5] and torch.equal(pvis, rvis)
    # the visible-rows-only variant: a view's gradient rows are exact zeros where it does not see the Gaussian (what the backward writes)
    for v, x in zip(slab.grads(_M()), grads):
        v.copy_(x * vis.view(-1, *([1] * (x.dim() - 1))))
    dense = [v.clone() for v in slab.grads(_M())]
    dvis = trainer.allreduce_slab(slab, vis)
    dense_red = [v.clone() for v in slab.grads(_M())]
    for v, x in zip(slab.grads(_M()), dense):
        v.copy_(x)
    svis2, rows = trainer.allreduce_slab_sparse(slab, vis, _M())
    assert torch.equal(svis2, dvis) and rows == int(dvis.sum())
    for a, b in zip(slab.grads(_M()), dense_red):
        assert torch.equal(a, b)      # bit for bit the dense result (two addends)
    # the rank-1 exchange (default at N > 1): xyz / opacity / scaling / rotation all-reduced, the views' masked colour gradients and
    # camera centres all-gathered, dL_ddc / dL_dsh of all views rebuilt locally.  The HIP rebuild kernel is replaced by its plain-torch
    # restatement here (tests/sh_rank1_ref.py; the kernel itself is checked on the GPU): this covers the collectives, the slab
    # layout and the order in which the groups become ready
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import sh_rank1_ref as ref
    from gaussian_lic_amd import rasterizer as rz
    means = torch.randn(P, 3, generator=torch.Generator().manual_seed(7)) * 3.0     # replicated map: same on every rank
    campos = torch.tensor([0.25 * rank - 0.1, 0.05 * rank, -0.3])
    rgb = torch.randn(P, 3, generator=g) * vis.view(-1, 1)                          # zeros where this view does not see the Gaussian
    dc_v, sh_v = ref.rows_one_view(means, campos, rgb, 3, 15)
    for v, x in zip(slab.grads(_M()), [grads[0] * vis.view(-1, 1), dc_v, sh_v, grads[3] * vis.view(-1, 1), grads[4] * vis.view(-1, 1), grads[5] * vis.view(-1, 1)]):
        v.copy_(x)
    keep = [v.clone() for v in slab.grads(_M())]
    trainer.allreduce_slab(slab, vis)
    dense_r1 = [v.clone() for v in slab.grads(_M())]
    for v, x in zip(slab.grads(_M()), keep):
        v.copy_(x)
    slab.views["features_dc"].fill_(float("nan")); slab.views["features_rest"].fill_(float("nan"))   # not shipped: rebuilt after the exchange
    def _rebuild(means3D, campos_all, rgb_all, degree, dL_ddc, dL_dsh, input_is_ddc=False, n_views=None, view_stride=0):
        # the kernel reads the all-gathered payload in place: view v's colour gradients / camera centre start view_stride floats further
        rgb_v = rgb_all.as_strided((n_views, means3D.shape[0], 3), (view_stride, 3, 1))
        cam_v = campos_all.as_strided((n_views, 3), (view_stride, 1))
        a, b = ref.rows_from_rgb(means3D, cam_v, rgb_v, degree, dL_dsh.shape[1])
        dL_ddc.copy_(a); dL_dsh.copy_(b)
    rz.sh_grad_from_rgb = _rebuild
    class _Model(_M):
        xyz = means
        sh_degree = 3
    slab.rgb.copy_(rgb)    # (the backward writes the colour gradient straight into the slab's all-gather payload)
    r1vis, works = trainer.exchange_rank1(slab, slab.rgb, vis, _Model(), campos)
    order = []
    for work, idx in works:
        work.wait()
        order += idx
    assert order == [1, 2, 0, 3, 4, 5] and torch.equal(r1vis, dvis)
    for i, (a, b) in enumerate(zip(slab.grads(_M()), dense_r1)):
        assert torch.equal(a, b), i      # two addends, the restatement on both sides: bit for bit the dense all-reduce
    q.put((rank, [r.clone().numpy() for r in red], rvis.numpy(), [x.numpy() for x in grads], vis.numpy()))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_gradient_allreduce_two_ranks_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, red0, vis0, g0, v0), (_, red1, vis1, g1, v1) = res
    for a, b, x, y in zip(red0, red1, g0, g1):
        np.testing.assert_array_equal(a, b)                 # every rank holds the identical reduced slab
        np.testing.assert_allclose(a, x + y, rtol=0, atol=0)  # = sum of the per-view gradients (2 addends: exact)
        assert a.shape == x.shape
    np.testing.assert_array_equal(vis0, vis1)
    np.testing.assert_array_equal(vis0, v0 | v1)              # mask = OR of the views


def _worker_n(rank, world, port, q):
    """World size > 2: the reduction order is the backend's, so the variants are held to each other by tolerance (1e-6 of the group's max-abs)
    instead of bit for bit; what stays exact is the mask (OR) and that every rank ends with the same bits (checked by the parent)."""
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.distributed.init_process_group("gloo", rank=rank, world_size=world)
    import gaussian_lic_amd  # noqa: F401
    import sh_rank1_ref as ref
    from gaussian_lic_amd import rasterizer as rz
    from gaussian_lic_amd import trainer
    P = 321
    g = torch.Generator().manual_seed(200 + rank)
    shapes = [(P, 3), (P, 1, 3), (P, 15, 3), (P, 1), (P, 3), (P, 4)]
    vis = torch.rand(P, generator=g) < 0.4

    class _M:
        NAMES = ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation")
        device = torch.device("cpu")
        P = 321
        xyz = torch.randn(321, 3, generator=torch.Generator().manual_seed(7)) * 3.0    # replicated map
        sh_degree = 3
        def parameters(self): return [torch.empty(*s) for s in shapes]
    m = _M()
    campos = torch.tensor([0.25 * rank - 0.1, 0.05 * rank, -0.3])
    rgb = torch.randn(P, 3, generator=g) * vis.view(-1, 1)
    dc_v, sh_v = ref.rows_one_view(m.xyz, campos, rgb, 3, 15)
    mask = vis.view(-1, 1).float()
    local = [torch.randn(P, 3, generator=g) * mask, dc_v, sh_v, torch.randn(P, 1, generator=g) * mask, torch.randn(P, 3, generator=g) * mask,
             torch.randn(P, 4, generator=g) * mask]
    slab = trainer.GradSlab(m)
    def load():
        for v, x in zip(slab.grads(m), local):
            v.copy_(x)
    load()
    dvis = trainer.allreduce_slab(slab, vis)
    dense = [v.clone() for v in slab.grads(m)]
    load()
    avis, works = trainer.allreduce_slab_async(slab, vis, m)
    for work, _idx in works:
        work.wait()
    asyncr = [v.clone() for v in slab.grads(m)]
    load()
    slab.views["features_dc"].fill_(float("nan")); slab.views["features_rest"].fill_(float("nan"))
    def _rebuild(means3D, campos_all, rgb_all, degree, dL_ddc, dL_dsh, input_is_ddc=False, n_views=None, view_stride=0):
        rgb_v = rgb_all.as_strided((n_views, means3D.shape[0], 3), (view_stride, 3, 1))
        cam_v = campos_all.as_strided((n_views, 3), (view_stride, 1))
        a, b = ref.rows_from_rgb(means3D, cam_v, rgb_v, degree, dL_dsh.shape[1])
        dL_ddc.copy_(a); dL_dsh.copy_(b)
    rz.sh_grad_from_rgb = _rebuild
    slab.rgb.copy_(rgb)
    rvis, works = trainer.exchange_rank1(slab, slab.rgb, vis, m, campos)
    for work, _idx in works:
        work.wait()
    rank1 = [v.clone() for v in slab.grads(m)]
    assert torch.equal(avis, dvis) and torch.equal(rvis, dvis)
    for i in range(6):
        scale = float(dense[i].abs().max()) + 1e-30
        assert float((asyncr[i] - dense[i]).abs().max()) / scale < 1e-6, ("async", i)
        assert float((rank1[i] - dense[i]).abs().max()) / scale < 2e-6, ("rank1", i)
    q.put((rank, [x.numpy() for x in dense], [x.numpy() for x in rank1], dvis.numpy(), [x.numpy() for x in local], vis.numpy()))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_gradient_exchange_four_ranks_gloo():
    """World size 4 on the CPU: dense, pipelined and rank-1 exchange agree (tolerance: the backend's reduction order), every rank holds
    the identical result of each, it equals the sum over the ranks' local gradients, and the mask is the OR of the views' masks."""
    world, port = 4, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_n, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in res[1:]:
        for a, b in zip(res[0][1], r[1]):
            np.testing.assert_array_equal(a, b)            # dense: identical on every rank
        for a, b in zip(res[0][2], r[2]):
            np.testing.assert_array_equal(a, b)            # rank-1: identical on every rank
        np.testing.assert_array_equal(res[0][3], r[3])
    for i in range(6):
        total = sum(r[4][i].astype(np.float64) for r in res)
        scale = np.abs(total).max() + 1e-30
        assert np.abs(res[0][1][i] - total).max() / scale < 1e-6
    np.testing.assert_array_equal(res[0][3], np.logical_or.reduce([r[5] for r in res]))


def test_view_sharding_rule():
    """bench.py: rank k of an N-rank job renders synthetic view k % 8; a single rank renders the identity pose."""
    import gaussian_lic_amd  # noqa: F401
    from gaussian_lic_amd.camera import synthetic_camera
    c0 = synthetic_camera(64, 48, None)
    assert np.allclose(c0.world_view_transform, np.eye(4))
    views = [synthetic_camera(64, 48, k) for k in range(8)]
    cens = np.array([v.camera_center for v in views])
    assert np.allclose(cens[:, 0], (np.arange(8) - 3.5) * 0.25, atol=1e-6) and np.allclose(cens[:, 1:], 0, atol=1e-6)
    for v in views:   # full_proj = view x proj, stored transposed (camera.h:60,86,109)
        assert np.allclose(v.full_proj_transform, v.world_view_transform @ v.projection_matrix, atol=1e-6)


def test_bench_self_launch_command(monkeypatch):
    """bench.py --gpus N without WORLD_SIZE re-launches itself under torch.distributed.run on 127.0.0.1 with N ranks and its own flags (the GPU
    run of it is tests/test_dist_gpu.py::test_bench_gpus_2_launches_itself)."""
    import subprocess
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "20", "--warmup", "5"])
    assert bench.self_launch(4) == 7
    cmd = seen["cmd"]
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-6:] == ["--gpus", "4", "--steps", "20", "--warmup", "5"] and cmd[-7].endswith("bench.py")
    assert seen["env"]["GSLIC_BENCH_SELF_LAUNCHED"] == "1"
