// Optional replacement for the reference's src/optim_utils.h (SparseGaussianAdam, :26-142) — same class names, constructor,
// param_groups()/add_param_group() (inherited), set_visibility_and_N(), step(), get_state() and State fields, so gaussian.cpp
// (trainingSetup :399-418, densificationPostfix :444-487, optimize :705-707) compiles against it unchanged — but one step() is ONE
// kernel launch over all parameter groups (adamUpdateGroups -> gslic_adam_update_groups) instead of one launch plus one
// grad.clone() per group (reference :130-133).  Put gaussian-lic_amd/shim/include ahead of the reference's src/ on the include
// path to select it; with the reference's own header the host still works, through adamUpdate, six launches per step.
// The arithmetic is adam.cu:26-37 either way: parameters after a step are bit-identical (tests/test_shim_gpu.py).
#pragma once

#include "rasterizer/adam_groups.h"

#include <torch/torch.h>

#include <memory>
#include <unordered_map>
#include <vector>

// per-parameter moments, keyed by TensorImpl* like the reference (densificationPostfix re-keys them)
struct State {
    int64_t step = 0;
    torch::Tensor exp_avg;
    torch::Tensor exp_avg_sq;
    bool initialized = false;
};

struct SparseGaussianAdamOptions : public torch::optim::OptimizerOptions {
    double lr_;
    double eps_;
    explicit SparseGaussianAdamOptions(double lr = 1e-3, double eps = 1e-8) : lr_(lr), eps_(eps) {}
    std::unique_ptr<torch::optim::OptimizerOptions> clone() const override { return std::make_unique<SparseGaussianAdamOptions>(*this); }
    double get_lr() const override { return lr_; }
    void set_lr(const double lr) override { lr_ = lr; }
    double get_eps() const { return eps_; }
    void set_eps(const double eps) { eps_ = eps; }
};

class SparseGaussianAdam : public torch::optim::Optimizer {
public:
    SparseGaussianAdam(const std::vector<torch::Tensor>& params, double lr, double eps)
        : torch::optim::Optimizer({torch::optim::OptimizerParamGroup(params)}, std::make_unique<SparseGaussianAdamOptions>(lr, eps))
    {
    }

    void set_visibility_and_N(const torch::Tensor& visibility, int64_t N)
    {
        visibility_ = visibility;
        N_ = N;
    }

    std::unordered_map<torch::TensorImpl*, State>& get_state() { return state_; }

    torch::Tensor step(LossClosure closure = nullptr) override
    {
        torch::Tensor loss;
        if (closure != nullptr) loss = closure();
        // gather every group that has a gradient, then update them all in one launch
        std::vector<torch::Tensor> params, grads, m1, m2;
        std::vector<double> lrs;
        double eps = 1e-15;
        bool have_eps = false;
        for (auto& group : param_groups_) {
            TORCH_CHECK(group.params().size() == 1, "More than one tensor in group");
            auto& opt = static_cast<SparseGaussianAdamOptions&>(group.options());
            torch::Tensor& p = group.params()[0];
            if (!p.grad().defined()) continue;
            State& st = state_[p.unsafeGetTensorImpl()];
            if (!st.initialized) {
                st.exp_avg = torch::zeros_like(p, torch::MemoryFormat::Preserve);
                st.exp_avg_sq = torch::zeros_like(p, torch::MemoryFormat::Preserve);
                st.step = 0;
                st.initialized = true;
            }
            st.step += 1;
            if (p.numel() == 0) continue;
            params.push_back(p);
            grads.push_back(p.grad());   // read in place: the kernel never writes the gradient, so no clone
            m1.push_back(st.exp_avg);
            m2.push_back(st.exp_avg_sq);
            lrs.push_back(opt.get_lr());
            // one launch carries ONE epsilon: the reference passes each group's own eps_ to adamUpdate (optim_utils.h:126-131), which is the
            // same value for every group of trainingSetup (gaussian.cpp:399-418); a host that sets them apart must not be folded silently
            TORCH_CHECK(!have_eps || opt.get_eps() == eps, "SparseGaussianAdam: the parameter groups use different eps values; the one-launch step applies a single one");
            eps = opt.get_eps();
            have_eps = true;
        }
        if (!params.empty()) adamUpdateGroups(params, grads, m1, m2, visibility_, lrs, 0.9f, 0.999f, (float)eps, (uint32_t)N_);
        return loss;
    }

private:
    torch::Tensor visibility_;
    int64_t N_ = 0;
    std::unordered_map<torch::TensorImpl*, State> state_;
};
