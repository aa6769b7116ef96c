// grid_common.hpp -- what the density and the Vxc translation units share (f64 MFMA wrapper, fragment layout).
//
// f64 MFMA fragment layout (gfx950): A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15],
// C[row = (lane>>4) + 4*reg][col = lane&15].
#pragma once
#include <algorithm>
#include <cstdlib>

#include "common.hpp"
#include "xc_funcs.hpp"

namespace dqc {

typedef double v4d __attribute__((ext_vector_type(4)));

DQC_DEV v4d mfma_f64(double a, double b, v4d c) { return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0); }

}  // namespace dqc
