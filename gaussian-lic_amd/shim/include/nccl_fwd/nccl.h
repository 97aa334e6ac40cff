// c10d's ProcessGroupNCCL.hpp includes <nccl.h>; on ROCm that library is RCCL.
#pragma once
#include <rccl/rccl.h>
