#pragma once
#include "ref_prelude.h"
#include <hipcub/hipcub.hpp>
namespace cub = hipcub;
