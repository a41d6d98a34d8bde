// rr_ols_dev.hpp — device helpers shared by the overlap-save kernels (rr_ols.hip: k_ols_wave; rr_ols_frame.hip: k_ols_frame).
#pragma once
#include "rr_wave_math.hpp"

namespace rr {

// Read side of the three exchanges of the wave-local inverse DFT_256 (butterfly l takes in[l + 64 c], c < 4, at rd + st c):
//   pass 1  the image of pass 0 (element i at 2 (i >> 2) + (i & 1) + 144 ((i >> 1) & 1)), st = 32
//   pass 2  element i at i + 4 (i >> 4), st = 80 (its stores - 20 (l >> 2) + (l & 3) + 4 c - are conflict-free, these reads
//           2-way: groups of 16 lanes on the store side and halves of 32 on the read side cannot both be served by a padding)
//   pass 3  element i at i + 16 (i >> 6), st = 80 (stores 80 g + q + 16 c)
__device__ __forceinline__ const f2 *inv256_rd(const f2 *lds, int l, int pass) {
    if (pass == 1) return lds + (2 * (l >> 2) + (l & 1) + 144 * ((l >> 1) & 1));
    if (pass == 2) return lds + (l + 4 * (l >> 4));
    return lds + l;
}

// a read whose address is the same for the whole wave, from memory no kernel of the launch writes: through the constant address
// space it becomes an s_load (a plain global pointer gives a vector load per lane once the kernel has stored anything)
__device__ __forceinline__ float2 ld_uniform(const float2 *p) {
    typedef float __attribute__((ext_vector_type(2))) v2;
    const v2 v = *(const v2 __attribute__((address_space(4))) *)(unsigned long long)p;
    return float2{v.x, v.y};
}

}  // namespace rr
