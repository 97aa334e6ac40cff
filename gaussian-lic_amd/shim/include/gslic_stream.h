// gslic_stream.h — which HIP stream the LibTorch hosts of this directory hand to the C-ABI.
//
// Default: NULL, the legacy default stream — what the reference's bare <<<grid, block>>> launches use (rasterizer_impl.cu, ssim.cu,
// simple_knn.cu), and LibTorch's current stream as long as the host installs no stream guard.  A host that runs under a
// c10 stream guard (a non-default current stream: ProcessGroupNCCL work->wait() and every LibTorch op order against THAT stream)
// compiles with -DGSLIC_SHIM_CURRENT_STREAM (HIP headers on the include path, linked with c10_hip): the kernels then go to LibTorch's
// current stream, so they stay ordered with the tensors' producers and consumers.  The in-tree shim and check programs are built that way.
#pragma once
#if defined(GSLIC_SHIM_CURRENT_STREAM)
#include <c10/hip/HIPStream.h>
namespace gslic {
inline void* current_stream() { return static_cast<void*>(c10::hip::getCurrentHIPStream().stream()); }
}  // namespace gslic
#else
namespace gslic {
inline void* current_stream() { return nullptr; }
}  // namespace gslic
#endif
