#pragma once
#include <cub/cub.cuh>
