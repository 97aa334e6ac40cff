// gslic_dist.h — the ONE exchange step the N-GPU path adds to the reference's C++ host: every rank holds a full
// replica of the map, renders a different camera view, and between `loss.backward()` and `sparse_optimizer_->step()`
// (src/gaussian.cpp:697-707) the ranks SUM their parameter gradients and OR their visibility masks over RCCL, so that every replica
// applies the identical masked Adam update and no broadcast is ever needed.  Header-only, LibTorch + c10d only (backend "nccl" IS RCCL
// on ROCm).  Build flags the host needs for ProcessGroupNCCL.hpp: -DUSE_C10D_NCCL -DUSE_ROCM -D__HIP_PLATFORM_AMD__, the ROCm include
// directory, and an `nccl.h` on the include path that forwards to <rccl/rccl.h> (gaussian-lic_amd/shim/include/nccl_fwd/nccl.h).
//
//   auto pg = gslic::make_rccl_group("127.0.0.1", port, rank, world);          // once, after cudaSetDevice(local_rank)
//   ...
//   loss.backward();
//   visible = gslic::exchange_gradients(*pg, {xyz_, features_dc_, features_rest_, opacity_, scaling_, rotation_}, visible);
//   sparse_optimizer_->set_visibility_and_N(visible, xyz_.size(0));
//   sparse_optimizer_->step();
#pragma once
#include <torch/torch.h>
#include <torch/csrc/distributed/c10d/ProcessGroupNCCL.hpp>
#include <torch/csrc/distributed/c10d/TCPStore.hpp>

#include <string>
#include <vector>

#include "gslic_stream.h"
#include "../../../include/gslic_hip.h"   // gslic_sh_grad_from_rgb (exchange_gradients_rank1)

namespace gslic {

inline c10::intrusive_ptr<c10d::ProcessGroupNCCL> make_rccl_group(const std::string& master_addr, int port, int rank, int world)
{
    c10d::TCPStoreOptions so;
    so.port = static_cast<uint16_t>(port);
    so.isServer = rank == 0;
    so.numWorkers = world;
    auto store = c10::make_intrusive<c10d::TCPStore>(master_addr, so);
    return c10::make_intrusive<c10d::ProcessGroupNCCL>(store, rank, world, c10d::ProcessGroupNCCL::Options::create());
}

// In place on the .grad() of every parameter (a parameter without a gradient contributes zeros); returns the OR of the masks.
// sparse = true exchanges only the rows of the OR-ed mask (a view sees a fraction of the map): same result, fewer bytes on the links.
inline torch::Tensor exchange_gradients(c10d::Backend& pg, const std::vector<torch::Tensor>& params, const torch::Tensor& visible, bool sparse = false)
{
    torch::NoGradGuard no_grad;
    const int64_t P = visible.size(0);
    std::vector<at::Tensor> vis{visible.to(torch::kUInt8).contiguous()};
    c10d::AllreduceOptions mx;
    mx.reduceOp = c10d::ReduceOp::MAX;
    pg.allreduce(vis, mx)->wait();
    torch::Tensor mask = vis[0].to(torch::kBool);
    std::vector<torch::Tensor> grads;
    for (const auto& p : params) grads.push_back(p.grad().defined() ? p.grad() : torch::zeros_like(p));
    torch::Tensor idx;
    if (sparse) idx = mask.nonzero().squeeze(1);
    std::vector<torch::Tensor> rows;
    for (const auto& g : grads) {
        torch::Tensor r = g.reshape({P, -1});
        rows.push_back(sparse ? r.index_select(0, idx).reshape({-1}) : r.reshape({-1}));
    }
    std::vector<at::Tensor> slab{torch::cat(rows)};   // ONE collective for the six groups
    pg.allreduce(slab)->wait();
    int64_t off = 0;
    for (size_t i = 0; i < grads.size(); i++) {
        const int64_t n = rows[i].numel();
        torch::Tensor part = slab[0].narrow(0, off, n);
        if (sparse) grads[i].reshape({P, -1}).index_copy_(0, idx, part.reshape({idx.size(0), -1}));
        else grads[i].reshape({-1}).copy_(part);
        if (!params[i].grad().defined()) params[i].mutable_grad() = grads[i];
        off += n;
    }
    return mask;
}

// The same exchange with the SH gradients shipped as what they are — rank-1 (include/gslic_hip.h, gslic_sh_grad_from_rgb): the SH backward
// is linear in the clamp-masked colour gradient, dL_ddc = SH_C0 dRGB and dL_dsh[k] = c_k(view direction) dRGB, so only xyz / opacity /
// scaling / rotation (11 of 59 floats) are all-reduced; the ranks all-gather their dL_ddc (3 floats) and every rank rebuilds the summed
// dL_ddc / dL_dsh of all views itself.  A rank sends 2 (N-1)/N 44 P + (N-1) 13 P bytes per step instead of 2 (N-1)/N 236 P: 4.1x less at
// N = 2, 2.5x at N = 8 — on point-to-point xGMI links that is the exposed part of the step.
// TWO collectives, both asynchronous, neither in front of the other: ONE all-gather of a per-rank payload {dL_ddc [P,3], camera centre [3],
// visibility [P] bytes} — the rebuild kernel reads the gathered blocks in place through view_stride, the masks are OR-ed locally (no
// separate MAX-reduce) — and ONE all-reduce of the four small groups as one slab.
// This host only has the reference's gradient tensors, so dRGB is recovered from dL_ddc (exact to 1 ulp); all replicas rebuild the
// identical rows, so they stay bit-identical to each other.
//   params: {xyz, features_dc, features_rest, opacity, scaling, rotation};  camera_center: [3] device (Camera::camera_center_).
inline torch::Tensor exchange_gradients_rank1(c10d::Backend& pg, const std::vector<torch::Tensor>& params, const torch::Tensor& visible,
                                              const torch::Tensor& camera_center, int sh_degree)
{
    torch::NoGradGuard no_grad;
    TORCH_CHECK(params.size() == 6, "exchange_gradients_rank1: six parameter groups expected");
    const int64_t P = visible.size(0), world = pg.getSize();
    std::vector<torch::Tensor> grads;
    for (const auto& p : params) grads.push_back(p.grad().defined() ? p.grad().contiguous() : torch::zeros_like(p));
    const int small_groups[4] = {0, 3, 4, 5};
    std::vector<torch::Tensor> rows;
    for (int g : small_groups) rows.push_back(grads[g].reshape({-1}));
    std::vector<at::Tensor> slab{torch::cat(rows)};
    auto w_small = pg.allreduce(slab);
    // the all-gather payload of this rank
    const int64_t pay_bytes = (12 * P + 12 + P + 3) / 4 * 4;
    auto bo = grads[1].options().dtype(torch::kUInt8);
    at::Tensor pay = torch::zeros({pay_bytes}, bo), pay_all = torch::empty({world, pay_bytes}, bo);
    pay.narrow(0, 0, 12 * P).view(torch::kFloat32).copy_(grads[1].reshape({-1}));
    pay.narrow(0, 12 * P, 12).view(torch::kFloat32).copy_(camera_center.to(grads[1].options()).reshape({3}));
    pay.narrow(0, 12 * P + 12, P).copy_(visible.to(torch::kUInt8));
    at::Tensor pay_in = pay.view({1, pay_bytes});
    pg._allgather_base(pay_all, pay_in)->wait();
    torch::Tensor mask = std::get<0>(pay_all.narrow(1, 12 * P + 12, P).max(0)).to(torch::kBool);
    const int64_t M = params[2].numel() ? params[2].size(1) : 0;
    at::Tensor xyz = params[0].detach().contiguous();
    const uint8_t* base = pay_all.data_ptr<uint8_t>();
    int rc = gslic_sh_grad_from_rgb((int32_t)P, sh_degree, (int32_t)M, (int32_t)world, xyz.data_ptr<float>(), reinterpret_cast<const float*>(base + 12 * P),
                                    reinterpret_cast<const float*>(base), /*input_is_ddc=*/1, grads[1].data_ptr<float>(),
                                    M ? grads[2].data_ptr<float>() : nullptr, /*view_stride=*/pay_bytes / 4, /*stream=*/current_stream());
    TORCH_CHECK(rc == GSLIC_OK, "gslic_sh_grad_from_rgb failed: ", gslic_last_error());
    w_small->wait();
    int64_t off = 0;
    for (size_t i = 0; i < 4; i++) {
        const int64_t n = rows[i].numel();
        grads[small_groups[i]].reshape({-1}).copy_(slab[0].narrow(0, off, n));
        off += n;
    }
    for (size_t i = 0; i < 6; i++)
        if (!params[i].grad().defined() || !params[i].grad().is_contiguous()) params[i].mutable_grad() = grads[i];
    return mask;
}

}  // namespace gslic
