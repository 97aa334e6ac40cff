#pragma once
#include "ref_prelude.h"
