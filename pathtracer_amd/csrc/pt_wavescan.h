// Inclusive prefix sum over the 64 lanes of a wavefront in registers (DPP row shifts and row
// broadcasts; no LDS, no memory).  All 64 lanes must be active.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ptd {

// DPP controls (gfx9): row_shr:n = 0x110 + n, row_bcast:15 = 0x142, row_bcast:31 = 0x143.
// update_dpp(old, src, ctrl, row_mask, bank_mask, bound_ctrl): lanes whose source is out of bounds or
// masked off keep `old` (0 here), so the add below leaves them unchanged.
__device__ __forceinline__ uint32_t wave_inclusive_sum(uint32_t v) {
	int x = (int)v;
	x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xf, 0xf, false);   // row_shr:1
	x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xf, 0xf, false);   // row_shr:2
	x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xf, 0xf, false);   // row_shr:4
	x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xf, 0xf, false);   // row_shr:8   -> inclusive within each row of 16
	x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xa, 0xf, false);   // row_bcast:15 into rows 1 and 3
	x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xc, 0xf, false);   // row_bcast:31 into rows 2 and 3
	return (uint32_t)x;
}

}  // namespace ptd
