"""Training step of the hot path — mirrors optimize() (src/gaussian.cpp:640-719) and the parts of GaussianModel it
touches (getters :147-175, trainingSetup :399-418), plus the one exchange step the north star adds for N > 1 GPUs:
a single all-reduce of the Gaussian-gradient slab (+ a max-reduce of the visibility mask) per optimiser step.

One process per GPU (torch.distributed, backend "nccl" = RCCL over xGMI; "gloo" on CPU for tests).  Every rank
holds a full replica of the parameters and Adam state, renders a different camera view, and applies the identical
sparse-Adam update, so replicas stay bit-identical without a broadcast.
"""
import os

import torch

from . import loss as loss_utils
from .optim import SparseGaussianAdam
from .rasterizer import render

# config/fastlivo.yaml:18-22 (position, feature, opacity, scaling, rotation) and lambda_dssim
DEFAULT_LRS = dict(position_lr=1.6e-4, feature_lr=2.5e-3, opacity_lr=5e-2, scaling_lr=5e-3, rotation_lr=1e-3)
LAMBDA_DSSIM = 0.2


def morton_order(xyz, bits=10):
    """Permutation (LongTensor [P], CPU) that sorts the rows by the Morton code of their position: `bits` per axis over the robust bounding box
    (0.1 .. 99.9 percentile per axis, outliers clamped), ties in row order (stable).  View-independent: spatial neighbours become memory
    neighbours, which is what a SLAM map's insertion order gives frame by frame and a synthetic scene in random order does not."""
    import numpy as np
    q = xyz.detach().cpu().double().numpy()
    lo, hi = np.percentile(q, 0.1, axis=0), np.percentile(q, 99.9, axis=0)
    u = (np.clip((q - lo) / np.maximum(hi - lo, 1e-30), 0.0, 1.0) * ((1 << bits) - 1)).astype(np.uint64)

    def spread(v):   # 10 bits -> every third bit
        v = (v | (v << np.uint64(16))) & np.uint64(0x030000FF)
        v = (v | (v << np.uint64(8))) & np.uint64(0x0300F00F)
        v = (v | (v << np.uint64(4))) & np.uint64(0x030C30C3)
        v = (v | (v << np.uint64(2))) & np.uint64(0x09249249)
        return v
    assert bits <= 10
    code = spread(u[:, 0]) | (spread(u[:, 1]) << np.uint64(1)) | (spread(u[:, 2]) << np.uint64(2))
    return torch.from_numpy(np.argsort(code, kind="stable").astype(np.int64))


def morton_order_device(xyz, bits=10):
    """morton_order with torch ops on xyz's own device (a few ms for 2M rows on the GPU: what GaussianModel.resort() runs between keyframes).
    Same construction — `bits` per axis over a robust bounding box (the 0.1 % / 99.9 % order statistics per axis), outliers clamped, ties in row
    order (stable sort) — but not necessarily the same permutation as the numpy version to the last row: any permutation renders and trains
    bit-identically (tie_rank), the order only decides how coherent memory is."""
    assert bits <= 10
    q = xyz.detach().float()
    P = int(q.shape[0])
    # the robust box from a strided sample of at most 65 536 rows (sorted per axis): the bounds only place the quantisation grid, and
    # torch.kthvalue over all rows cost 100 ms of the 103 a re-sort of 2M rows took (profiles/r06d: resort_on_device_ms)
    smp = q[::max(1, P // 65536)]
    srt = torch.sort(smp, dim=0).values
    n_s = int(srt.shape[0])
    lo, hi = srt[min(n_s - 1, int(0.001 * n_s))], srt[max(0, min(n_s - 1, int(0.999 * n_s)))]
    u = (((q - lo) / (hi - lo).clamp_min(1e-30)).clamp(0.0, 1.0) * ((1 << bits) - 1)).to(torch.int64)

    def spread(v):   # 10 bits -> every third bit
        v = (v | (v << 16)) & 0x030000FF
        v = (v | (v << 8)) & 0x0300F00F
        v = (v | (v << 4)) & 0x030C30C3
        v = (v | (v << 2)) & 0x09249249
        return v
    code = spread(u[:, 0]) | (spread(u[:, 1]) << 1) | (spread(u[:, 2]) << 2)
    return torch.sort(code, stable=True).indices


class GaussianModel:
    """Parameters + activations of src/gaussian.{h,cpp} that the hot path touches, with capacity-doubling storage so that
    extend() appends rows in place instead of six torch::cat reallocations of every parameter and Adam moment per keyframe
    (densificationPostfix, gaussian.cpp:426-497)."""

    NAMES = ("xyz", "features_dc", "features_rest", "opacity", "scaling", "rotation")   # group order of gaussian.cpp:399-418
    raw_parameter_leaves = True   # the attributes of those names are the PRE-activation leaves: rasterizer.render() may feed them to the raw-parameter node

    def __init__(self, raw, device, lambda_erank=0.0, capacity=None, scaling_scale=1.0, order="insertion", resort_fraction=0.1):
        """order = "insertion": rows in the order given (the reference's: initialize() then extend() appends).
        order = "morton": the rows are stored sorted by the Morton code of their position (morton_order): what a camera sees is then contiguous
        in memory, whole waves of the per-Gaussian kernels are invisible and skip, and the 128-byte lines of parameters and Adam moments are
        no longer shared between visible and invisible rows.  self.tie_rank[s] = original index of storage row s: the forward breaks depth
        ties by it (gslic_raster_params.tie_rank), so rendering and training are bit-identical to the insertion order; original_order() /
        io_ply.save_map() give the rows back in the original order.  Rows appended by extend() keep their insertion order behind the sorted block;
        once they make up more than `resort_fraction` of the map, extend() re-sorts the whole map on the device (resort(): a few ms between
        keyframes; None = never) — the coherence the layout is for would otherwise decay as a SLAM map grows.
        self.sort_ms: what the sorts cost (the construction's CPU sort, then the device re-sorts), for the record next to a throughput number."""
        self.sh_degree = int(raw["sh_degree"])
        self._tie = None
        self.order = order
        self.resort_fraction = resort_fraction if order == "morton" else None
        self.sort_ms = []
        import time as _time
        _t0 = _time.perf_counter()
        if order == "morton":
            perm = morton_order(raw["xyz"])
            raw = {k: (v[perm].contiguous() if (torch.is_tensor(v) and k in self.NAMES) else v) for k, v in raw.items()}
        else:
            assert order == "insertion", order
            perm = None
        self.lambda_erank = float(lambda_erank)
        self.scaling_scale = float(scaling_scale)
        self.device = device
        self.P = int(raw["xyz"].shape[0])
        cap = max(int(capacity or self.P), self.P)
        self._buf, self._m, self._v = {}, {}, {}
        for n in self.NAMES:
            src = raw[n].to(device).float()
            self._buf[n] = torch.empty((cap,) + tuple(src.shape[1:]), device=device)
            self._buf[n][:self.P].copy_(src)
            self._m[n] = torch.zeros_like(self._buf[n])
            self._v[n] = torch.zeros_like(self._buf[n])
        if perm is not None:
            self._tie = torch.empty(cap, dtype=torch.int32, device=device)
            self._tie[:self.P].copy_(perm.to(torch.int32))
            self.sort_ms.append(round(1e3 * (_time.perf_counter() - _t0), 2))   # (CPU sort + the permuted upload, once, outside any timed region)
        self._sorted_P = self.P      # rows [0, _sorted_P) are in Morton order (order == "morton")
        self.optimizer = None
        self._rebind()

    @torch.no_grad()
    def resort(self):
        """Puts ALL rows back into Morton order (order == "morton" only): parameters, both Adam moments and tie_rank are permuted together on the
        device, so the map renders, trains and exports exactly as before (tie_rank still holds every row's ORIGINAL index).  Invalidates hipGraph
        captures of nothing (addresses do not change) but any per-row state a host keeps outside the model must be permuted by the same index:
        returns the permutation applied (LongTensor [P] on the device; new row s = old row perm[s])."""
        assert self._tie is not None, "resort(): the model keeps its rows in insertion order"
        import time as _time
        torch.cuda.synchronize(self.device) if self.device.type == "cuda" else None
        t0 = _time.perf_counter()
        P = self.P
        perm = morton_order_device(self._buf["xyz"][:P])
        for d in (self._buf, self._m, self._v):
            for n in self.NAMES:
                d[n][:P] = d[n][:P][perm]
        self._tie[:P] = self._tie[:P][perm]
        self._sorted_P = P
        self._rebind()
        torch.cuda.synchronize(self.device) if self.device.type == "cuda" else None
        self.sort_ms.append(round(1e3 * (_time.perf_counter() - t0), 2))
        return perm

    @property
    def tie_rank(self):
        """int32 [P] original index of every storage row, or None when the rows are in their original order."""
        return None if self._tie is None else self._tie[:self.P]

    def original_order(self):
        """LongTensor [P]: storage rows listed in the ORIGINAL order (x[model.original_order()] un-permutes a per-row tensor x); None = identity."""
        return None if self._tie is None else torch.argsort(self._tie[:self.P].long())

    @property
    def capacity(self):
        return self._buf["xyz"].shape[0]

    def _rebind(self):
        for n in self.NAMES:
            setattr(self, n, self._buf[n][:self.P].detach().requires_grad_(True))
        if self.optimizer is not None:
            self.optimizer.rebind(self.parameters(), [self._m[n][:self.P] for n in self.NAMES], [self._v[n][:self.P] for n in self.NAMES])

    def _reserve(self, newP):
        if newP <= self.capacity:
            return
        cap = max(2 * self.capacity, newP)
        for d in (self._buf, self._m, self._v):
            for n in self.NAMES:
                old = d[n]
                new = torch.zeros((cap,) + tuple(old.shape[1:]), device=self.device) if d is not self._buf else \
                    torch.empty((cap,) + tuple(old.shape[1:]), device=self.device)
                new[:self.P].copy_(old[:self.P])
                d[n] = new
        if self._tie is not None:
            tie = torch.empty(cap, dtype=torch.int32, device=self.device)
            tie[:self.P].copy_(self._tie[:self.P])
            self._tie = tie

    # gaussian.cpp:147-175
    def get_xyz(self): return self.xyz
    def get_features_dc(self): return self.features_dc
    def get_features_rest(self): return self.features_rest
    def get_opacity(self): return torch.sigmoid(self.opacity)
    def get_scaling(self): return torch.exp(self.scaling)
    def get_rotation(self): return torch.nn.functional.normalize(self.rotation)

    def parameters(self):
        """Group order of trainingSetup (gaussian.cpp:399-418)."""
        return [getattr(self, n) for n in self.NAMES]

    def training_setup(self, lrs=None):
        c = dict(DEFAULT_LRS)
        c.update(lrs or {})
        group_lrs = [c["position_lr"], c["feature_lr"], c["feature_lr"] / 20.0, c["opacity_lr"], c["scaling_lr"], c["rotation_lr"]]
        self.optimizer = SparseGaussianAdam(self.parameters(), group_lrs, eps=1e-15)
        self._rebind()
        return self.optimizer

    @torch.no_grad()
    def extend(self, camera, points, colors, depths_rsp, R_cw, t_cw, intrinsics, bg=None):
        """extend() of gaussian.cpp:499-638 for one new frame: transmittance-only render of the latest camera, GPU selection of
        the LiDAR points that land on not-yet-opaque pixels (nearest per pixel), append of the new Gaussians (+ zero Adam moments)
        in place.  points/colors [n,3], depths_rsp [n] on the device; R_cw [3,3], t_cw [3]; intrinsics = (fx, fy, cx, cy).
        Returns the number of Gaussians inserted."""
        import ctypes
        from . import _lib
        from .rasterizer import render
        L = _lib.lib()
        dev = self.device
        bg = torch.zeros(3, device=dev) if bg is None else bg
        _img, final_T, _pts, _vis, _radii = render(camera, self, bg, no_color=True)
        n = int(points.shape[0])
        fx, fy, cx, cy = (float(v) for v in intrinsics)
        points, colors, depths_rsp = points.contiguous().float(), colors.contiguous().float(), depths_rsp.contiguous().float()
        Rc, tc = R_cw.to(dev).float().contiguous(), t_cw.to(dev).float().contiguous()
        scratch = _lib.TensorAllocator(dev)
        flags, pos, count = ctypes.c_void_p(), ctypes.c_void_p(), ctypes.c_int32(0)
        p = _lib.ptr
        _lib.check(L.gslic_extend_select(n, p(points), p(depths_rsp), p(Rc), p(tc), fx, fy, cx, cy, camera.image_width, camera.image_height,
                                         p(final_T.contiguous()), scratch.cb, None, ctypes.byref(flags), ctypes.byref(pos),
                                         ctypes.byref(count), _lib.current_stream_ptr()))
        k = count.value
        if k == 0:
            return 0
        P0 = self.P
        self._reserve(P0 + k)
        M = self._buf["features_rest"].shape[1]
        row = lambda name: ctypes.c_void_p(self._buf[name][P0:].data_ptr()) if self._buf[name][P0:].numel() else None
        focal = (fx + fy) / 2.0
        _lib.check(L.gslic_extend_emit(n, flags, pos, p(points), p(colors), p(depths_rsp), self.scaling_scale, focal, M, row("xyz"),
                                       row("features_dc"), row("features_rest"), row("opacity"), row("scaling"), row("rotation"),
                                       _lib.current_stream_ptr()))
        for d in (self._m, self._v):       # new rows start with zero moments (gaussian.cpp:458-459)
            for name in self.NAMES:
                d[name][P0:P0 + k].zero_()
        if self._tie is not None:          # appended rows are in insertion order behind the sorted block: their original index is their row
            self._tie[P0:P0 + k].copy_(torch.arange(P0, P0 + k, dtype=torch.int32, device=dev))
        self.P = P0 + k
        self._rebind()
        torch.cuda.current_stream().synchronize()  # scratch (flags/pos) is released on return
        if self._tie is not None and self.resort_fraction is not None and (self.P - self._sorted_P) > self.resort_fraction * self.P:
            self.resort()    # the appended tail has grown past resort_fraction of the map: Morton order again (a few ms, between keyframes)
        return k


def exchange_mode():
    """How the N > 1 step exchanges gradients: "rank1" (default: 11 floats all-reduced + the 3-float colour gradient all-gathered,
    exchange_rank1), "dense" (the whole [P x 59] slab all-reduced in three pipelined segments), "sparse" (only the rows some view sees).
    GSLIC_EXCHANGE selects; GSLIC_SPARSE_EXCHANGE=1 is the older spelling of "sparse"."""
    m = os.environ.get("GSLIC_EXCHANGE", "").lower()
    if m in ("rank1", "dense", "sparse", "single"):   # "single": north_star's wording to the letter — ONE all-reduce per step (allreduce_slab_single)
        return m
    return "sparse" if os.environ.get("GSLIC_SPARSE_EXCHANGE") == "1" else "rank1"


DIST_TIMING = None   # bench.py sets this to a dict: the N > 1 step then records CUDA events at its three phase boundaries


def _dist_mark(name):
    if DIST_TIMING is not None:
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        DIST_TIMING.setdefault(name, []).append(ev)


def exchange_chunks():
    """GSLIC_EXCHANGE_CHUNKS = C > 1: the rank-1 exchange runs the per-Gaussian backward in C row chunks and puts chunk c on the wire (one
    all-gather + one all-reduce per chunk) while chunk c + 1 is computed; the Adam updates of a chunk follow its collectives.  Default 1."""
    try:
        return max(1, int(os.environ.get("GSLIC_EXCHANGE_CHUNKS", "1")))
    except ValueError:
        return 1


def _dist_on():
    """True when the per-step gradient exchange has to run.  GSLIC_FORCE_DIST=1 also takes the exchange path in a process group of
    ONE rank (the RCCL smoke test on a single-GPU box: same code path as N > 1, the all-reduce degenerates to a copy)."""
    if not (torch.distributed.is_available() and torch.distributed.is_initialized()):
        return False
    return torch.distributed.get_world_size() > 1 or os.environ.get("GSLIC_FORCE_DIST") == "1"


class GradSlab:
    """One contiguous fp32 buffer [P x (11+3K)] holding the six gradient tensors back to back (group order of
    gaussian.cpp:399-418).  The fused backward writes straight into its views, so the per-step exchange is ONE in-place
    all-reduce of `flat` with no concatenation copy (472 MB at 2M Gaussians, SH degree 3)."""

    def __init__(self, model):
        self.P = model.P
        shapes = [tuple(p.shape) for p in model.parameters()]
        sizes = [int(torch.tensor(s).prod()) for s in shapes]
        # (the slab is followed by P more floats: GSLIC_EXCHANGE=single ships the visibility mask in the SAME all-reduce, allreduce_slab_single)
        self.flat_and_mask = torch.empty(sum(sizes) + self.P, device=model.device)
        self.flat = self.flat_and_mask[:sum(sizes)]
        self.mask_f = self.flat_and_mask[sum(sizes):]
        # exchange_rank1's all-gather payload of THIS rank, one contiguous block: {colour gradient [P,3] fp32, camera centre [3] fp32,
        # visibility [P] bytes, padding to 4 bytes}; `rgb` (what the backward writes), `pay_campos` and `pay_vis` are views into it
        self.pay_bytes = (12 * self.P + 12 + self.P + 3) // 4 * 4
        self.payload = torch.zeros(self.pay_bytes, dtype=torch.uint8, device=model.device)
        self.rgb = self.payload[:12 * self.P].view(torch.float32).view(self.P, 3)
        self.pay_campos = self.payload[12 * self.P:12 * self.P + 12].view(torch.float32)
        self.pay_vis = self.payload[12 * self.P + 12:12 * self.P + 12 + self.P]
        self.payload_all = None
        self.rgb_all, self.campos_all = None, None
        self.views, off = {}, 0
        for name, shp, n in zip(model.NAMES, shapes, sizes):
            self.views[name] = self.flat[off:off + n].view(shp)
            off += n

    def grads(self, model):
        return [self.views[n] for n in model.NAMES]

    def run(self, model, g0, g1):
        """(flat slice covering the whole groups g0..g1-1, their indices): consecutive groups are contiguous in the slab."""
        sizes = [self.views[n].numel() for n in model.NAMES]
        offs = [0]
        for n in sizes:
            offs.append(offs[-1] + n)
        return self.flat[offs[g0]:offs[g1]], list(range(g0, g1))

    def segments(self, model):
        """The slab as three contiguous runs of whole groups, largest first: features_rest (76 % of the bytes), then
        xyz + features_dc, then opacity + scaling + rotation.  Returns [(flat slice, group indices)]."""
        sizes = [self.views[n].numel() for n in model.NAMES]
        offs = [0]
        for n in sizes:
            offs.append(offs[-1] + n)
        return [(self.flat[offs[2]:offs[3]], [2]), (self.flat[offs[0]:offs[2]], [0, 1]), (self.flat[offs[3]:offs[6]], [3, 4, 5])]


def allreduce_slab(slab, visible):
    """The per-step exchange on a GradSlab: SUM all-reduce of the slab in place + MAX (= OR) of the visibility bytes."""
    vis = visible.to(torch.uint8)
    torch.distributed.all_reduce(slab.flat, op=torch.distributed.ReduceOp.SUM)
    torch.distributed.all_reduce(vis, op=torch.distributed.ReduceOp.MAX)
    return vis.bool()


def allreduce_slab_single(slab, visible):
    """GSLIC_EXCHANGE=single — the exchange exactly as north_star words it: "a single RCCL all-reduce of Gaussian gradients over xGMI per optimiser
    step".  ONE SUM all-reduce of [P x 59] gradient floats + P floats carrying the visibility mask (0 / 1 per view: the sum is > 0 where some view
    sees the Gaussian — the OR), one collective on the wire instead of the default's three, at 4 (59 + 1) P bytes per rank and direction against
    rank-1's 44 P + (N - 1) 13 P (480 vs 114-336 MB at 2M Gaussians, N = 2-8).  Kept selectable so that the first run on real links measures both
    designs in one command (bench.py --gpus N reports `exchange.mode`)."""
    slab.mask_f.copy_(visible)
    torch.distributed.all_reduce(slab.flat_and_mask, op=torch.distributed.ReduceOp.SUM)
    return slab.mask_f > 0


def allreduce_slab_async(slab, visible, model):
    """The same exchange, pipelined against the optimiser: the visibility mask first (2 MB, blocking), then the three segments of
    the slab as asynchronous SUM all-reduces issued back to back.  Returns (reduced visibility, [(work, group indices)]): the
    caller waits for a segment and runs Adam on ITS groups while the following segments are still on the links, so all but the
    last, small Adam launch hide under the transfer (at 2M Gaussians the transfer is the longer leg of an N > 1 step)."""
    vis = visible.to(torch.uint8)
    torch.distributed.all_reduce(vis, op=torch.distributed.ReduceOp.MAX)
    works = [(torch.distributed.all_reduce(seg, op=torch.distributed.ReduceOp.SUM, async_op=True), idx) for seg, idx in slab.segments(model)]
    return vis.bool(), works


def allreduce_slab_sparse(slab, visible, model):
    """The exchange restricted to the rows some view actually sees: MAX (= OR) of the visibility bytes first, then ONE SUM all-reduce
    of a compacted slab holding only the rows of the OR-ed mask (every rank derives the same row list from the same mask), scattered
    back into the full slab.  Rows outside the mask are invisible on every rank: their gradients are exact zeros everywhere and Adam
    skips them.  Bytes on the links: 4 * 59 * |mask| instead of 4 * 59 * P — a SLAM view sees a fraction of the map, the synthetic
    8-view rig ~85 % (tools/exchange_bytes.py).  Result identical to allreduce_slab (bit for bit at N = 2; for N > 2 the rank order of
    the sum may differ per element).  Returns (reduced visibility, rows exchanged)."""
    vis = visible.to(torch.uint8)
    torch.distributed.all_reduce(vis, op=torch.distributed.ReduceOp.MAX)
    mask = vis.bool()
    idx = mask.nonzero(as_tuple=False).squeeze(1)          # (one host sync: the row count sizes the buffer; the MAX-reduce above already blocked)
    P, V = slab.P, int(idx.numel())
    if V == 0:
        return mask, 0
    rows = [slab.views[n].view(P, -1) for n in model.NAMES]
    widths = [r.shape[1] for r in rows]
    compact = torch.empty(V * sum(widths), device=slab.flat.device, dtype=slab.flat.dtype)
    parts, off = [], 0
    for r, w in zip(rows, widths):                          # group-major, like the slab itself
        part = compact[off:off + V * w].view(V, w)
        torch.index_select(r, 0, idx, out=part)
        parts.append(part)
        off += V * w
    torch.distributed.all_reduce(compact, op=torch.distributed.ReduceOp.SUM)
    for r, part in zip(rows, parts):
        r.index_copy_(0, idx, part)
    return mask, V


def exchange_rank1(slab, rgb_local, visible, model, campos):
    """The N > 1 exchange with the SH gradients shipped as what they are — rank-1.  computeColorFromSH's backward (backward.cu:27-136) is
    linear in the clamp-masked colour gradient dRGB: dL_ddc = SH_C0 dRGB, dL_dsh[k] = c_k(dir) dRGB with c_k a function of the view
    direction only.  So of the 59 gradient floats per Gaussian only 11 (xyz, opacity, scaling, rotation) are all-reduced; each rank
    all-gathers the views' 3-float dRGB and rebuilds the summed dL_ddc / dL_dsh itself (gslic_sh_grad_from_rgb: the backward's own
    products, summed in view order — at N = 2 bit-identical to the dense all-reduce).  Bytes on a rank's links per step:
    2 (N-1)/N 44 P + (N-1) 13 P instead of 2 (N-1)/N 236 P — 4.1x less at N = 2, 2.5x less at N = 8 (xGMI is point-to-point: the dense
    slab is per-link bound).

    THREE collectives per step, all asynchronous, none in front of the others: ONE all-gather of a per-rank payload {dRGB, camera centre,
    visibility bytes} (the rebuild kernel reads the gathered blocks in place through a view stride; the masks are OR-ed locally, so no
    separate MAX-reduce blocks the step), and the all-reduces of the two contiguous runs of the gradient slab that hold the 11 small
    floats (xyz | ... | opacity, scaling, rotation).  Returns (OR-ed visibility, works) like allreduce_slab_async: wait on a work, then run
    Adam on its groups; the rebuild of the SH rows runs behind the all-gather while the all-reduces are still on the links.

    ALIASING (folded path, the default): the returned mask is a bool VIEW of the persistent buffer slab.vis_or.  It is filled by the launch
    inside the returned work's wait() — not before — and overwritten by the next step: a caller that keeps `visible` across steps
    (densification statistics, logging) must clone it after wait()."""
    from . import rasterizer as rz
    dist = torch.distributed
    n = dist.get_world_size()
    assert rgb_local.data_ptr() == slab.rgb.data_ptr(), "the colour gradient must have been written into the slab's payload"
    if campos is not None:   # (None: the backward kernel has filled the payload's camera centre and mask itself: gslic_rasterize_backward_rgb_payload)
        slab.pay_campos.copy_(campos.reshape(3))
        slab.pay_vis.copy_(visible)
    if slab.payload_all is None or slab.payload_all.size(0) != n:
        slab.payload_all = torch.empty(n, slab.pay_bytes, dtype=torch.uint8, device=slab.flat.device)
    w_pay = dist.all_gather_into_tensor(slab.payload_all, slab.payload.view(1, slab.pay_bytes), async_op=True)
    works = [(dist.all_reduce(seg, op=dist.ReduceOp.SUM, async_op=True), idx) for seg, idx in (slab.run(model, 0, 1), slab.run(model, 3, 6))]
    P = slab.P
    rgb0 = slab.payload_all[0, :12 * P].view(torch.float32)            # view 0's block; view v sits pay_bytes / 4 floats further
    cam0 = slab.payload_all[0, 12 * P:12 * P + 12].view(torch.float32)
    stride = slab.pay_bytes // 4
    fused = getattr(model, "optimizer", None) is not None and os.environ.get("GSLIC_RANK1_SPLIT_ADAM") != "1"
    if fused and os.environ.get("GSLIC_RANK1_FOLD_ADAM", "1") != "0":
        # ONE launch behind the three collectives: OR of the gathered masks, SH rows rebuilt and consumed, Adam of all six groups
        # (gslic_sh_grad_from_rgb_adam_all) — no MAX-reduce, no mask copies, no separate Adam launches on the step's critical path
        if getattr(slab, "vis_or", None) is None:
            slab.vis_or = torch.zeros(P, dtype=torch.uint8, device=slab.flat.device)
        vis_all0 = slab.payload_all[0, 12 * P + 12:12 * P + 12 + P]

        class _All:
            def wait(self_inner):
                w_pay.wait()
                for w, _ in works:
                    w.wait()
                v = slab.views
                model.optimizer.step_all_from_exchange(model.xyz.detach(), cam0, rgb0, model.sh_degree, n, stride, vis_all0, slab.pay_bytes, slab.vis_or,
                                                       (v["xyz"], v["opacity"], v["scaling"], v["rotation"]))
        return slab.vis_or.view(torch.bool), [(_All(), [])]
    w_pay.wait()   # (orders the current stream behind the gather; the host does not block on RCCL)
    vis = slab.payload_all[:, 12 * P + 12:12 * P + 12 + P].max(0).values.bool()   # OR of the views' masks

    class _Rebuild:   # rebuilds dL_ddc / dL_dsh of all views from the gathered colour gradients (the mask has to be set first: see the caller)
        def wait(self_inner):
            if fused:   # ... and applies the masked Adam to features_dc / features_rest in the same kernel: the rows are never materialised
                model.optimizer.step_sh_from_rgb(model.xyz.detach(), cam0, rgb0, model.sh_degree, n_views=n, view_stride=stride)
            else:       # ... into the slab, for a following optimizer.step(only=[1, 2])
                rz.sh_grad_from_rgb(model.xyz.detach(), cam0, rgb0, model.sh_degree, slab.views["features_dc"], slab.views["features_rest"],
                                    n_views=n, view_stride=stride)
    return vis, [(_Rebuild(), [] if fused else [1, 2])] + works


class _ChunkedExchange:
    """Buffers of the chunked rank-1 exchange for one (P, world, chunks): per chunk c = rows [p0, p1) a block of the 11 small gradient floats
    {xyz [n,3] | opacity [n] | scaling [n,3] | rotation [n,4]} (ONE all-reduce) and an all-gather payload {dRGB [n,3], camera centre [3],
    visibility [n] bytes}.  The backward kernels index their outputs by the absolute Gaussian index: they get the blocks' addresses moved back
    by p0 rows."""

    def __init__(self, P, world, chunks, device):
        self.P, self.world, self.chunks = P, world, chunks
        step = -(-P // chunks)
        step = -(-step // 256) * 256
        self.bounds = [(p0, min(p0 + step, P)) for p0 in range(0, P, step)]
        self.small = torch.empty(11 * P, device=device)
        self.vis = torch.zeros(P, dtype=torch.uint8, device=device)
        self.pay, self.pay_all, self.small_off = [], [], []
        off = 0
        for p0, p1 in self.bounds:
            n = p1 - p0
            nbytes = (12 * n + 12 + n + 3) // 4 * 4
            self.pay.append(torch.zeros(nbytes, dtype=torch.uint8, device=device))
            self.pay_all.append(torch.empty(world, nbytes, dtype=torch.uint8, device=device))
            self.small_off.append(off)
            off += 11 * n


def training_step_rank1_chunked(model, cam, bg, dL_dimage, fwd, chunks):
    """The N > 1 step behind the forward and the loss, with the exchange overlapped chunk by chunk (exchange_chunks()).  Per chunk: the
    per-Gaussian backward of its rows (the blend backward runs once, with the first chunk), then — asynchronously, on RCCL's streams — ONE
    all-gather of the chunk's {dRGB, camera centre, mask} and ONE all-reduce of its 11 small gradient floats, while the next chunk is being
    computed.  Then, chunk by chunk as the collectives land: OR of the masks, SH rows rebuilt and consumed by the masked Adam of
    features_dc / features_rest, Adam of the four small groups.  Same arithmetic per Gaussian as the unchunked step."""
    from . import rasterizer as rz
    dist = torch.distributed
    R, B, radii, geom, binning, img, sample = fwd
    dev = model.device
    e = torch.empty(0, device=dev)
    n_world = dist.get_world_size()
    cx = getattr(model, "_chunk_xchg", None)
    if cx is None or cx.P != model.P or cx.world != n_world or cx.chunks != chunks:
        cx = model._chunk_xchg = _ChunkedExchange(model.P, n_world, chunks, dev)
    xyz, dc, rest = model.xyz.detach(), model.features_dc.detach(), model.features_rest.detach()
    sc, rot = model.scaling.detach(), model.rotation.detach()
    slab = model._grad_slab
    visible_local = (radii > 0).to(torch.uint8)
    works = []
    for c, (p0, p1) in enumerate(cx.bounds):
        n = p1 - p0
        sm = cx.small.data_ptr() + 4 * cx.small_off[c]
        pay = cx.pay[c]
        addr = {"xyz": sm - 12 * p0, "opacity": sm + 12 * n - 4 * p0, "scaling": sm + 16 * n - 12 * p0, "rotation": sm + 28 * n - 16 * p0,
                "rgb": pay.data_ptr() - 12 * p0}
        rz.rasterize_gaussians_backward(
            bg, xyz, radii, e, sc, rot, 1.0, e, cam.d_world_view_transform, cam.d_full_proj_transform, float(cam.tanfovx), float(cam.tanfovy),
            float(cam.limx_neg), float(cam.limx_pos), float(cam.limy_neg), float(cam.limy_pos), dL_dimage, dc, rest, model.sh_degree,
            cam.d_camera_center, geom, R, binning, img, B, sample, model.lambda_erank, False, raw_params=True, out=slab.views, rgb_out=slab.rgb,
            rows=(p0, p1), skip_blend=(c > 0), out_addr=addr)
        pay[12 * n:12 * n + 12].view(torch.float32).copy_(cam.d_camera_center.reshape(3))
        pay[12 * n + 12:12 * n + 12 + n].copy_(visible_local[p0:p1])
        if c == 0:
            _dist_mark("bwd_done")   # (first chunk on the wire: everything behind it overlaps with the remaining chunks' compute)
        w_pay = dist.all_gather_into_tensor(cx.pay_all[c], pay.view(1, -1), async_op=True)
        w_red = dist.all_reduce(cx.small[cx.small_off[c]:cx.small_off[c] + 11 * n], op=dist.ReduceOp.SUM, async_op=True)
        works.append((w_pay, w_red))
    model.optimizer.set_visibility_and_N(cx.vis, model.P)
    for c, (p0, p1) in enumerate(cx.bounds):
        n = p1 - p0
        w_pay, w_red = works[c]
        w_pay.wait()
        pa = cx.pay_all[c]
        cx.vis[p0:p1] = pa[:, 12 * n + 12:12 * n + 12 + n].max(0).values
        rgb0 = pa[0, :12 * n].view(torch.float32)
        cam0 = pa[0, 12 * n:12 * n + 12].view(torch.float32)
        model.optimizer.step_sh_from_rgb(xyz, cam0, rgb0, model.sh_degree, n_views=n_world, view_stride=pa.shape[1] // 4, rows=(p0, p1))
        w_red.wait()
        sm = cx.small.data_ptr() + 4 * cx.small_off[c]
        model.optimizer.step(None, only=[0, 3, 4, 5], rows=(p0, p1), grad_addr={0: sm, 3: sm + 12 * n, 4: sm + 16 * n, 5: sm + 28 * n})
    _dist_mark("end")
    return cx.vis.bool()


def allreduce_gradients(grads, visible):
    """The per-step exchange: SUM of the concatenated gradient slab [P x (11+3K)] and MAX (= OR) of the visibility
    bytes.  One collective each; returns (list of reduced gradient views, reduced visibility)."""
    flat = torch.cat([g.reshape(-1) for g in grads])
    vis = visible.to(torch.uint8)
    torch.distributed.all_reduce(flat, op=torch.distributed.ReduceOp.SUM)
    torch.distributed.all_reduce(vis, op=torch.distributed.ReduceOp.MAX)
    out, off = [], 0
    for g in grads:
        n = g.numel()
        out.append(flat[off:off + n].view_as(g))
        off += n
    return out, vis.bool()


def training_step(model, camera, gt_image, bg, lambda_dssim=LAMBDA_DSSIM, do_step=True, raw_render=None, one_node_loss=False):
    """One iteration of optimize()'s loop (gaussian.cpp:674-716) the way the reference's host writes it: render -> 0.8*L1 + 0.2*(1-SSIM) ->
    loss.backward() -> sparse Adam, on LibTorch autograd.  Returns (loss tensor, visible mask).
    raw_render: passed to render() — None = its default (the drop-in renderer: activations inside the kernels), False = renderer.cpp as written
    (getOpacity / getScaling / getRotation as LibTorch ops, the pure operator path).  one_node_loss: the optional loss_utils.l1_ssim_loss node
    instead of the reference's l1_loss + fused_ssim lines (gaussian.cpp:685-691)."""
    image, _final_T, _pts, visible, _radii = render(camera, model, bg, raw=raw_render)
    if one_node_loss:
        loss = loss_utils.l1_ssim_loss(image, gt_image, lambda_dssim)
    else:
        Ll1 = loss_utils.l1_loss(image, gt_image)
        ssim_value = loss_utils.fused_ssim(image.unsqueeze(0), gt_image.unsqueeze(0))
        loss = (1.0 - lambda_dssim) * Ll1 + lambda_dssim * (1.0 - ssim_value)
    loss.backward()
    if do_step:
        params = model.parameters()
        grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in params]
        if _dist_on():
            grads, visible = allreduce_gradients(grads, visible)
        model.optimizer.set_visibility_and_N(visible, model.xyz.size(0))
        model.optimizer.step(grads)
        model.optimizer.zero_grad(True)
    return loss.detach(), visible


def pose_gradient(model, camera, gt_image, bg, fused_loss=None):
    """dL/dxi of the training loss w.r.t. a left se(3) increment of the camera pose (Camera.pose_gradient) for the current map: forward -> loss
    kernels -> gslic_rasterize_backward_camera -> the chain.  The map is not touched.  Returns (float64 [6] = (d/drho, d/dphi), terms).
    One step of pose refinement is `camera.apply_pose_increment(-lr * g); camera.to_device(dev)`."""
    from . import rasterizer as rz
    fl = fused_loss or _default_fused_loss()
    e = torch.empty(0, device=model.device)
    cam = camera
    with torch.no_grad():
        xyz, dc, rest = model.xyz.detach(), model.features_dc.detach(), model.features_rest.detach()
        op, sc, rot = model.opacity.detach(), model.scaling.detach(), model.rotation.detach()
        (R, B, image, _final_T, radii, geom, binning, img, sample) = rz.rasterize_gaussians(
            bg, xyz, e, op, sc, rot, 1.0, e, cam.d_world_view_transform, cam.d_full_proj_transform, float(cam.tanfovx), float(cam.tanfovy),
            cam.image_height, cam.image_width, float(cam.limx_neg), float(cam.limx_pos), float(cam.limy_neg), float(cam.limy_pos), dc, rest,
            model.sh_degree, cam.d_camera_center, False, False, False, raw_params=True, tie_rank=getattr(model, "tie_rank", None))
        dL_dimage, terms = fl.forward_backward(image, gt_image)
        out = rz.rasterize_gaussians_backward(
            bg, xyz, radii, e, sc, rot, 1.0, e, cam.d_world_view_transform, cam.d_full_proj_transform, float(cam.tanfovx), float(cam.tanfovy),
            float(cam.limx_neg), float(cam.limx_pos), float(cam.limy_neg), float(cam.limy_pos), dL_dimage, dc, rest, model.sh_degree,
            cam.d_camera_center, geom, R, binning, img, B, sample, model.lambda_erank, False, raw_params=True, camera_grads=True)
    return cam.pose_gradient(out[9], out[10], out[11]), terms


def training_step_with_pose(model, camera, gt_image, bg, pose_lr=0.0, fused_loss=None):
    """One joint map + pose iteration: forward -> loss kernels -> gslic_rasterize_backward_camera (parameter gradients AND the camera gradient in
    one pass) -> masked Adam on the map -> `camera` moved by -pose_lr * dL/dxi (left se(3) increment; 35 floats cross to the host, which is the
    step's only synchronisation).  Returns (terms, visible, dL/dxi)."""
    from . import rasterizer as rz
    fl = fused_loss or _default_fused_loss()
    dev = model.device
    e = torch.empty(0, device=dev)
    cam = camera
    with torch.no_grad():
        xyz, dc, rest = model.xyz.detach(), model.features_dc.detach(), model.features_rest.detach()
        op, sc, rot = model.opacity.detach(), model.scaling.detach(), model.rotation.detach()
        (R, B, image, _final_T, radii, geom, binning, img, sample) = rz.rasterize_gaussians(
            bg, xyz, e, op, sc, rot, 1.0, e, cam.d_world_view_transform, cam.d_full_proj_transform, float(cam.tanfovx), float(cam.tanfovy),
            cam.image_height, cam.image_width, float(cam.limx_neg), float(cam.limx_pos), float(cam.limy_neg), float(cam.limy_pos), dc, rest,
            model.sh_degree, cam.d_camera_center, False, False, False, raw_params=True, tie_rank=getattr(model, "tie_rank", None))
        dL_dimage, terms = fl.forward_backward(image, gt_image)
        slab = getattr(model, "_grad_slab", None)
        if slab is None or slab.P != model.P:
            slab = model._grad_slab = GradSlab(model)
        out = rz.rasterize_gaussians_backward(
            bg, xyz, radii, e, sc, rot, 1.0, e, cam.d_world_view_transform, cam.d_full_proj_transform, float(cam.tanfovx), float(cam.tanfovy),
            float(cam.limx_neg), float(cam.limx_pos), float(cam.limy_neg), float(cam.limy_pos), dL_dimage, dc, rest, model.sh_degree,
            cam.d_camera_center, geom, R, binning, img, B, sample, model.lambda_erank, False, raw_params=True, out=slab.views, camera_grads=True)
        visible = radii > 0
        model.optimizer.set_visibility_and_N(visible, model.P)
        model.optimizer.step(slab.grads(model))
        g = cam.pose_gradient(out[9], out[10], out[11])
        if pose_lr:
            cam.apply_pose_increment(-float(pose_lr) * g).to_device(dev)
    return terms, visible, g


def training_step_fused(model, camera, gt_image, bg, fused_loss=None, do_step=True, adam_in_backward=True):
    """The same iteration as training_step — identical arithmetic up to fp32 rounding — on the fused entry points
    (SURVEY.md §8f row 2): sigmoid / exp / normalize and their backward run inside preprocess / preprocess_bwd (raw_params),
    the loss and dL/dimage come from two loss kernels, and there is no autograd graph: 0 LibTorch elementwise launches per step
    (the drop-in path issues ~35).  Returns (terms [mean L1, mean SSIM] device tensor, visible mask)."""
    from . import rasterizer as rz
    fl = fused_loss or _default_fused_loss()
    dev = model.device
    e = torch.empty(0, device=dev)
    if DIST_TIMING is not None and do_step and _dist_on():
        _dist_mark("start")
    with torch.no_grad():
        xyz, dc, rest = model.xyz.detach(), model.features_dc.detach(), model.features_rest.detach()
        op, sc, rot = model.opacity.detach(), model.scaling.detach(), model.rotation.detach()
        cam = camera
        (R, B, image, _final_T, radii, geom, binning, img, sample) = rz.rasterize_gaussians(
            bg, xyz, e, op, sc, rot, 1.0, e, cam.d_world_view_transform, cam.d_full_proj_transform, float(cam.tanfovx), float(cam.tanfovy),
            cam.image_height, cam.image_width, float(cam.limx_neg), float(cam.limx_pos), float(cam.limy_neg), float(cam.limy_pos), dc, rest,
            model.sh_degree, cam.d_camera_center, False, False, False, raw_params=True, tie_rank=getattr(model, "tie_rank", None))
        dL_dimage, terms = fl.forward_backward(image, gt_image)
        if do_step and adam_in_backward and not _dist_on():
            # single GPU: nothing to exchange, so the Adam update runs inside the per-Gaussian backward kernel while the
            # gradients are still in registers / LDS (bit-identical to backward + SparseGaussianAdam.step, ~2 GB less HBM traffic)
            vis_u8 = torch.empty(model.P, dtype=torch.uint8, device=dev)
            rz.rasterize_gaussians_backward(
                bg, xyz, radii, e, sc, rot, 1.0, e, cam.d_world_view_transform, cam.d_full_proj_transform, float(cam.tanfovx),
                float(cam.tanfovy), float(cam.limx_neg), float(cam.limx_pos), float(cam.limy_neg), float(cam.limy_pos), dL_dimage, dc, rest,
                model.sh_degree, cam.d_camera_center, geom, R, binning, img, B, sample, model.lambda_erank, False, raw_params=True,
                adam=model.optimizer.fused_descriptor(visible_out=vis_u8))
            model.optimizer.count_step()
            return terms, vis_u8.view(torch.bool)   # `radii > 0` (renderer.cpp:85), written by the backward kernel: no compare launch
        slab = getattr(model, "_grad_slab", None)
        if slab is None or slab.P != model.P:
            slab = model._grad_slab = GradSlab(model)
        mode = exchange_mode() if (do_step and _dist_on()) else None
        if mode == "rank1" and exchange_chunks() > 1 and os.environ.get("GSLIC_RANK1_SPLIT_ADAM") != "1":
            visible = training_step_rank1_chunked(model, cam, bg, dL_dimage, (R, B, radii, geom, binning, img, sample), exchange_chunks())
            return terms, visible
        if mode == "rank1":
            # the SH gradients travel as the 3-float colour gradient they are the outer product of (exchange_rank1)
            rz.rasterize_gaussians_backward(
                bg, xyz, radii, e, sc, rot, 1.0, e, cam.d_world_view_transform, cam.d_full_proj_transform, float(cam.tanfovx), float(cam.tanfovy),
                float(cam.limx_neg), float(cam.limx_pos), float(cam.limy_neg), float(cam.limy_pos), dL_dimage, dc, rest, model.sh_degree,
                cam.d_camera_center, geom, R, binning, img, B, sample, model.lambda_erank, False, raw_params=True, out=slab.views, rgb_out=slab.rgb,
                payload=(slab.pay_vis, slab.pay_campos))   # mask and camera centre of the all-gather payload come out of the same kernel
            _dist_mark("bwd_done")
            visible, works = exchange_rank1(slab, slab.rgb, None, model, None)
            model.optimizer.set_visibility_and_N(visible, model.P)
            grads = slab.grads(model)
            for work, idx in works:
                work.wait()
                model.optimizer.step(grads, only=idx)
            _dist_mark("end")
            return terms, visible
        rz.rasterize_gaussians_backward(
            bg, xyz, radii, e, sc, rot, 1.0, e, cam.d_world_view_transform, cam.d_full_proj_transform, float(cam.tanfovx), float(cam.tanfovy),
            float(cam.limx_neg), float(cam.limx_pos), float(cam.limy_neg), float(cam.limy_pos), dL_dimage, dc, rest, model.sh_degree,
            cam.d_camera_center, geom, R, binning, img, B, sample, model.lambda_erank, False, raw_params=True, out=slab.views)
        visible = radii > 0
        if do_step:
            if _dist_on():
                _dist_mark("bwd_done")
            if mode == "sparse":
                visible, _rows = allreduce_slab_sparse(slab, visible, model)   # visible rows only: fewer bytes on the links, one gather / scatter pass
                model.optimizer.set_visibility_and_N(visible, model.P)
                model.optimizer.step(slab.grads(model))
            elif mode == "single":
                visible = allreduce_slab_single(slab, visible)                 # ONE collective: slab + mask (north_star's wording)
                model.optimizer.set_visibility_and_N(visible, model.P)
                model.optimizer.step(slab.grads(model))
            elif _dist_on():
                visible, works = allreduce_slab_async(slab, visible, model)
                model.optimizer.set_visibility_and_N(visible, model.P)
                grads = slab.grads(model)
                for work, idx in works:       # Adam on a segment's groups as soon as ITS all-reduce is done
                    work.wait()
                    model.optimizer.step(grads, only=idx)
            else:
                model.optimizer.set_visibility_and_N(visible, model.P)
                model.optimizer.step(slab.grads(model))       # group order of gaussian.cpp:399-418
            if _dist_on():
                _dist_mark("end")
    return terms, visible


_FUSED_LOSS = None


def _default_fused_loss():
    global _FUSED_LOSS
    if _FUSED_LOSS is None:
        _FUSED_LOSS = loss_utils.FusedLoss(LAMBDA_DSSIM)
    return _FUSED_LOSS


def render_fwd_bwd(model, camera, dL_dimage, bg):
    """Bare rendered view forward + backward (the metric's "(fwd+bwd)"), no loss kernels, no optimiser."""
    image, _final_T, _pts, visible, _radii = render(camera, model, bg)
    image.backward(dL_dimage)
    for p in model.parameters():
        p.grad = None
    return visible


class GraphedStep:
    """training_step_fused captured in a hipGraph: forward in capacity mode (gslic_rasterize_forward_capacity: no host round trip, the
    reference blocks twice per forward, rasterizer_impl.cu:398,442) -> loss kernels -> backward with the Adam update inside, replayed
    with ONE launch per step.  Camera matrices and the ground-truth image are read from static device buffers that `step()` refreshes
    before the replay.  Single GPU (the N > 1 path has an exchange between backward and Adam and runs eagerly).

    The scratch capacities come from one eager forward (R, B) with `headroom`; the instance / bucket counts of a step stay on the
    device.  If they outgrow the buffers the step turns itself into a no-op (status bits; no gradients, no Adam) and `check()` —
    called every `check_every` steps (at most 32: the device keeps one did-not-fit bit per issued step), and by the caller at the
    end — grows the buffers from the largest counts seen, re-captures and repeats exactly the steps that did not fit, each with ITS
    pose (snapshotted by value) and ground truth (kept by reference with its version counter: a target that was modified in place, or
    replaced while steps were issued with gt_image=None, makes the repeat fail loudly instead of training on the wrong data), after the
    ones that did.  extend() changes P: build a new GraphedStep afterwards."""

    def __init__(self, model, camera, gt_image, bg, headroom=1.25, check_every=16, cap_R=None, cap_B=None, use_graph=False):
        """use_graph=False (default since round 5): the capacity-mode step (caller-owned scratch, no host round trip, overflow checks and repeats as
        below) issued as eager launches; use_graph=True: captured in a hipGraph and replayed with one launch per step.  The two are equally fast at
        2M Gaussians (bench.py: `capacity_eager` / `graphed`; the host runs ahead of the device either way), and the eager form has no hazard:
        on ROCm 7.2 a replay FAULTS (GPU memory access fault) when the host has called torch.cuda.synchronize() and then enqueued ANY other device
        work — a copy, a fill — before it (tools/experiments/graph_check_repro.py, modes B / C / E; a synchronise alone, or a blocking .cpu() copy
        alone, is harmless: modes A / D / F).  A host that uses the graph must keep its own device work off that pattern."""
        from . import rasterizer as rz
        assert not _dist_on(), "GraphedStep is the single-GPU path"
        self.use_graph = bool(use_graph)
        self.model, self.bg, self.headroom, self.check_every = model, bg, float(headroom), int(check_every)
        dev = model.device
        self.H, self.W = int(camera.image_height), int(camera.image_width)
        self.cam = camera
        self.view = camera.d_world_view_transform.clone()
        self.proj = camera.d_full_proj_transform.clone()
        self.campos = camera.d_camera_center.clone()
        self.gt = gt_image.clone()
        self.fl = loss_utils.FusedLoss(LAMBDA_DSSIM)
        self.e = torch.empty(0, device=dev)
        # sizes from one eager forward of the current state
        with torch.no_grad():
            R, B = rz.rasterize_gaussians(bg, model.xyz.detach(), self.e, model.opacity.detach(), model.scaling.detach(), model.rotation.detach(),
                                          1.0, self.e, self.view, self.proj, float(camera.tanfovx), float(camera.tanfovy), self.H, self.W,
                                          float(camera.limx_neg), float(camera.limx_pos), float(camera.limy_neg), float(camera.limy_pos),
                                          model.features_dc.detach(), model.features_rest.detach(), model.sh_degree, self.campos, False, False,
                                          False, raw_params=True, tie_rank=getattr(model, "tie_rank", None))[:2]
        self.cap_R, self.cap_B = int(R * self.headroom) + 65536, int(B * self.headroom) + 1024
        if cap_R is not None:
            self.cap_R = int(cap_R)
        if cap_B is not None:
            self.cap_B = int(cap_B)
        assert self.check_every <= 32, "the device keeps one did-not-fit bit per step of a window of 32"
        self.graph, self.bufs = None, None
        self.steps_issued = 0      # replays since the last check
        self.window = []           # (camera, ground truth) of those replays, in issue order
        self.recaptures = 0
        self._capture()

    def _eager(self):
        from . import rasterizer as rz
        m, c = self.model, self.cam
        xyz, dc, rest = m.xyz.detach(), m.features_dc.detach(), m.features_rest.detach()
        op, sc, rot = m.opacity.detach(), m.scaling.detach(), m.rotation.detach()
        scal = (float(c.tanfovx), float(c.tanfovy), float(c.limx_neg), float(c.limx_pos), float(c.limy_neg), float(c.limy_pos))
        (R, B, image, _T, radii, geom, binning, img, sample) = rz.rasterize_gaussians_capacity(
            self.bufs, self.bg, xyz, op, sc, rot, 1.0, self.view, self.proj, *scal, dc, rest, m.sh_degree, self.campos, raw_params=True,
            tie_rank=getattr(m, "tie_rank", None))
        dL_dimage, self.terms = self.fl.forward_backward(image, self.gt)
        rz.rasterize_gaussians_backward(self.bg, xyz, radii, self.e, sc, rot, 1.0, self.e, self.view, self.proj, scal[0], scal[1], *scal[2:],
                                        dL_dimage, dc, rest, m.sh_degree, self.campos, geom, R, binning, img, B, sample, m.lambda_erank, False,
                                        raw_params=True, adam=self._adam)

    def _capture(self):
        from . import rasterizer as rz
        dev = self.model.device
        self.bufs = rz.CapacityBuffers(self.model.P, self.W, self.H, self.cap_R, self.cap_B, dev)
        self._adam = self.model.optimizer.fused_descriptor()
        # warm-up on a side stream (allocations of the loss scratch, library one-offs), then capture.  The warm-up and the capture
        # pass both execute the step, so the parameters are saved and restored around them.
        names = self.model.NAMES
        saved = [(self.model._buf[n][:self.model.P].clone(), self.model._m[n][:self.model.P].clone(), self.model._v[n][:self.model.P].clone())
                 for n in names]
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s), torch.no_grad():
            self._eager()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        g = None
        if self.use_graph:
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g), torch.no_grad():
                self._eager()
            torch.cuda.synchronize()
        for n, (p, m1, m2) in zip(names, saved):
            self.model._buf[n][:self.model.P].copy_(p); self.model._m[n][:self.model.P].copy_(m1); self.model._v[n][:self.model.P].copy_(m2)
        self.bufs.status.zero_()
        self.graph = g
        self.steps_issued = 0
        self.window = []

    _BAKED = ("tanfovx", "tanfovy", "limx_neg", "limx_pos", "limy_neg", "limy_pos")   # scalars of the camera that are constants of the captured graph

    @staticmethod
    def _pose_of(camera):
        """The three pose arrays of a Camera by VALUE, on the host (they are numpy members of the object: no device work, no allocation on the
        device between two replays)."""
        return (camera.world_view_transform.copy(), camera.full_proj_transform.copy(), camera.camera_center.copy())

    def _load(self, camera, gt_image):
        """Refresh the static buffers the captured step reads.  The pose is copied on every call that names a camera (the same Camera object
        may have been moved in place since it was loaded last)."""
        if camera is not None:
            for k in self._BAKED:
                assert float(getattr(camera, k)) == float(getattr(self.cam, k)), f"{k} is baked into the graph"
            self.view.copy_(camera.d_world_view_transform); self.proj.copy_(camera.d_full_proj_transform); self.campos.copy_(camera.d_camera_center)
            self.cam = camera
            self._pose_host = self._pose_of(camera)
        if gt_image is not None and gt_image.data_ptr() != self.gt.data_ptr():
            self.gt.copy_(gt_image)
            self._gt_serial = getattr(self, "_gt_serial", 0) + 1

    def _snapshot(self, gt_image):
        """What a repeat of this step needs, independent of what the caller does to its objects afterwards: the pose by VALUE (host copies of the
        35 floats that are loaded now) and the target by reference together with the evidence that it is still the same data (the tensor's
        version counter; for gt_image=None — "the target that is loaded" — the serial number of the load)."""
        if getattr(self, "_pose_host", None) is None:
            self._pose_host = self._pose_of(self.cam)
        return dict(pose=self._pose_host, gt=gt_image, gt_version=None if gt_image is None else gt_image._version, gt_serial=getattr(self, "_gt_serial", 0))

    def _load_snapshot(self, snap):
        dev = self.view.device
        v, p, c = snap["pose"]
        self.view.copy_(torch.from_numpy(v).to(dev)); self.proj.copy_(torch.from_numpy(p).to(dev)); self.campos.copy_(torch.from_numpy(c).to(dev))
        self._pose_host = snap["pose"]
        gt = snap["gt"]
        if gt is None:
            if snap["gt_serial"] != getattr(self, "_gt_serial", 0):
                raise RuntimeError("GraphedStep: a step that has to be repeated ran on a target that has been replaced since (it was issued with "
                                   "gt_image=None); pass the target explicitly to step() when targets change between checks")
        else:
            if gt._version != snap["gt_version"]:
                raise RuntimeError("GraphedStep: the ground-truth tensor of a step that has to be repeated was modified in place after the step was issued")
            if gt.data_ptr() != self.gt.data_ptr():
                self.gt.copy_(gt)
                self._gt_serial = getattr(self, "_gt_serial", 0) + 1

    def _issue(self):
        if self.graph is not None:
            self.graph.replay()
        else:
            with torch.no_grad():
                self._eager()

    def step(self, camera=None, gt_image=None):
        """One optimiser step (replay).  Returns the device tensor [mean L1, mean SSIM] of this step's loss terms."""
        self._load(camera, gt_image)
        self._issue()
        self.steps_issued += 1
        if len(self.window) < 32:
            self.window.append(self._snapshot(gt_image))
        self.model.optimizer.count_step()
        if self.check_every and self.steps_issued >= self.check_every:
            self.check()
        return self.terms

    def check(self):
        """Synchronise and read the status words.  Steps that did not fit were no-ops: grow the buffers from the largest counts seen,
        re-capture and repeat THOSE steps — each with the camera and ground truth it was issued with — until all of them fitted.
        Returns the number of steps that had to be repeated."""
        repeated = 0
        for _round in range(8):
            issued, failed_mask, max_R, max_B = self.bufs.read_window()
            _R, _B, bits, good = self.bufs.read_status()
            missed = issued - good
            if missed <= 0:
                break
            # which steps: one bit per issue index while the window is at most 32 steps long; a longer unchecked run (check_every = 0)
            # can only be repeated on the view that is loaded now
            todo = [self.window[i] for i in range(min(issued, len(self.window))) if (failed_mask >> i) & 1] if issued <= 32 else [self._snapshot(None)] * missed
            self.cap_R = max(self.cap_R, int(max_R * self.headroom) + 65536)
            self.cap_B = max(self.cap_B, int(max_B * self.headroom) + 1024)
            if max_R > 0x7fffffff // 2 or bits & 16:
                raise RuntimeError("GraphedStep: the instance count does not fit 31 bits")
            self.recaptures += 1
            self._capture()                     # (zeroes the status words, empties the window)
            for snap in todo:
                self._load_snapshot(snap)
                self._issue()
                self.steps_issued += 1
                self.window.append(snap)
            repeated += len(todo)
        else:
            raise RuntimeError("GraphedStep: steps keep overflowing their capacity buffers")
        self.bufs.status.zero_()
        self.steps_issued = 0
        self.window = []
        return repeated

    @property
    def visible(self):
        return self.bufs.radii > 0
