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

}  // namespace gslic
