#pragma once
#include "ref_prelude.h"
#include <hip/hip_cooperative_groups.h>
