// translation unit = the reference's rasterizer_impl.cu, unmodified, with the CUDA shift semantics of ref_warp_size.h
#include "forward.h"
#include "ref_warp_size.h"
#undef WARP_SIZE
#define WARP_SIZE (RefWarpSize{})
#include "rasterizer_impl.cu"
