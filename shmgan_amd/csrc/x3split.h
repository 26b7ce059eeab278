// The exact three-plane bf16 split of fp32 operands shared by the opt-in fp32-from-bf16-products kernels (conv_wgrad_x3.hip, conv_fwd_x3.hip).
#pragma once
#include "common.h"

// (a, b) -> the three dwords (plane 0, 1, 2) holding the bf16 planes of a in the low half and of b in the high half:
// x0 = x & 0xffff0000, r = x - x0 (exact), x1 = r & 0xffff0000, x2 = r - x1 (exact, at most eight significant bits: a bf16 value as it stands)
__device__ __forceinline__ void x3_split_pair(float a, float b, unsigned& p0, unsigned& p1, unsigned& p2) {
    const unsigned ua = __float_as_uint(a), ub = __float_as_uint(b);
    p0 = __builtin_amdgcn_perm(ub, ua, 0x07060302u);                                   // hi16(b) : hi16(a)
    const float ra = a - __uint_as_float(ua & 0xffff0000u), rb = b - __uint_as_float(ub & 0xffff0000u);
    const unsigned va = __float_as_uint(ra), vb = __float_as_uint(rb);
    p1 = __builtin_amdgcn_perm(vb, va, 0x07060302u);
    const float sa = ra - __uint_as_float(va & 0xffff0000u), sb = rb - __uint_as_float(vb & 0xffff0000u);
    p2 = __builtin_amdgcn_perm(__float_as_uint(sb), __float_as_uint(sa), 0x07060302u);
}
