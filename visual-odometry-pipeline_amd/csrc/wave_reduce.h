// Wave-wide fp64 sum for gfx950 wave64 without touching LDS: four DPP steps (quad swaps, half-row
// mirror, row mirror) leave every row of 16 lanes holding its row sum, then the four row sums are
// read to scalar registers and added in a fixed order.  Every lane receives the same bits, and
// the summation order is fixed, so results are reproducible.  About 10x shorter dependent latency
// than six ds_bpermute-based __shfl_xor steps.
#pragma once

#include <hip/hip_runtime.h>

namespace vohip_dev {

// (only used with controls that give every lane a source lane -- quad permutations and mirrors -- so the
// bound-control bit changes nothing but spares the compiler a zero-initialised destination per move)
template <int CTRL>
__device__ __forceinline__ double dpp_move_f64(double v)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(0, lo, CTRL, 0xf, 0xf, true);
    hi = __builtin_amdgcn_update_dpp(0, hi, CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double readlane_f64(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

template <int CTRL>
__device__ __forceinline__ float dpp_add_f32(float v)
{
    const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false);
    return v + __int_as_float(t);
}

// Wave-wide fp32 sum in a fixed, balanced tree: fours, pairs of fours, pairs of eights (DPP within
// each row of 16 lanes), then the four row sums as (r0 + r1) + (r2 + r3) by two row broadcasts
// (row 1 += row 0 and row 3 += row 2, then row 3 += row 1); the total is read from the last lane,
// so every lane receives the same bits.
__device__ __forceinline__ float wave_sum_f32(float v)
{
    v = dpp_add_f32<0xB1>(v);    // quad_perm [1,0,3,2]
    v = dpp_add_f32<0x4E>(v);    // quad_perm [2,3,0,1]
    v = dpp_add_f32<0x141>(v);   // row_half_mirror
    v = dpp_add_f32<0x140>(v);   // row_mirror
    // row_bcast15 into rows 1 and 3, row_bcast31 into rows 2 and 3 (other rows add zero)
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xA, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xC, 0xF, false));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// Two independent sums at once, step by step in turn: each DPP step has to wait for the result of the
// one before it, and the other sum's step fills that wait.  Same operations per sum as wave_sum_f32.
__device__ __forceinline__ void wave_sum2_f32(float &a, float &b)
{
    a = dpp_add_f32<0xB1>(a);
    b = dpp_add_f32<0xB1>(b);
    a = dpp_add_f32<0x4E>(a);
    b = dpp_add_f32<0x4E>(b);
    a = dpp_add_f32<0x141>(a);
    b = dpp_add_f32<0x141>(b);
    a = dpp_add_f32<0x140>(a);
    b = dpp_add_f32<0x140>(b);
    // The two row broadcasts as adds that write only the rows they concern (rows 1 and 3 += lane 15 of the
    // row before, then rows 2 and 3 += lane 31): one instruction per step and sum.  Written out because the
    // compiler may not fold "other rows add zero" into the row mask (x + 0 is not x for x = -0); the rows
    // that are masked off never reach the total, which is read from the last lane.  A DPP operand must
    // have been written two instructions earlier: hence the idle slots.
    asm volatile("s_nop 1\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "v_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                 "s_nop 0\n\t"
                 "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                 "v_add_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                 "s_nop 1"
                 : "+v"(a), "+v"(b));
    a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(a), 63));
    b = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(b), 63));
}

// Sum over each aligned group of 8 lanes (every lane of the group receives it): three DPP steps.
__device__ __forceinline__ double group8_sum_f64(double v)
{
    v += dpp_move_f64<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_move_f64<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_move_f64<0x141>(v);  // row_half_mirror
    return v;
}

__device__ __forceinline__ double wave_sum_f64(double v)
{
    v += dpp_move_f64<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_move_f64<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_move_f64<0x141>(v);  // row_half_mirror
    v += dpp_move_f64<0x140>(v);  // row_mirror
    const double r0 = readlane_f64(v, 0), r1 = readlane_f64(v, 16), r2 = readlane_f64(v, 32),
                 r3 = readlane_f64(v, 48);
    return ((r0 + r1) + r2) + r3;
}

// Three independent fp64 sums at once, step by step in turn (same operations per sum as wave_sum_f64).
__device__ __forceinline__ void wave_sum3_f64(double &a, double &b, double &c)
{
    a += dpp_move_f64<0xB1>(a);
    b += dpp_move_f64<0xB1>(b);
    c += dpp_move_f64<0xB1>(c);
    a += dpp_move_f64<0x4E>(a);
    b += dpp_move_f64<0x4E>(b);
    c += dpp_move_f64<0x4E>(c);
    a += dpp_move_f64<0x141>(a);
    b += dpp_move_f64<0x141>(b);
    c += dpp_move_f64<0x141>(c);
    a += dpp_move_f64<0x140>(a);
    b += dpp_move_f64<0x140>(b);
    c += dpp_move_f64<0x140>(c);
    const double a0 = readlane_f64(a, 0), a1 = readlane_f64(a, 16), a2 = readlane_f64(a, 32), a3 = readlane_f64(a, 48);
    const double b0 = readlane_f64(b, 0), b1 = readlane_f64(b, 16), b2 = readlane_f64(b, 32), b3 = readlane_f64(b, 48);
    const double c0 = readlane_f64(c, 0), c1 = readlane_f64(c, 16), c2 = readlane_f64(c, 32), c3 = readlane_f64(c, 48);
    a = ((a0 + a1) + a2) + a3;
    b = ((b0 + b1) + b2) + b3;
    c = ((c0 + c1) + c2) + c3;
}

}  // namespace vohip_dev
