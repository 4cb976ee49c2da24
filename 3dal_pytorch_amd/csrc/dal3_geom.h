// dal3_geom.h — the reference's points-in-rotated-box test on the device.
// det3d/core/bbox/box_np_ops.py:641-647 (points_in_rbbox) -> det3d/core/bbox/geometry.py:240-275
// (_points_in_convex_polygon_3d_jit): a point is OUTSIDE as soon as one of the six faces gives
//     ((px*nx + py*ny) + pz*nz) + d >= 0
// evaluated left to right with every product and sum rounded on its own (NumPy scalar arithmetic / numba without
// fastmath): no FMA contraction here either, hence the explicit *_rn intrinsics. A NaN never compares >= 0, so a
// NaN point stays inside — kept. The face equations [nx,ny,nz,d] come from the host (O(#boxes) NumPy, the same
// calls the reference makes: sin/cos of a float32 yaw are NumPy's own SIMD routines and cannot be reproduced bit
// for bit by device libm); they are stored as float64. The expression is evaluated in float32 when points AND
// boxes are float32 (the sweep / detector case), otherwise in float64 (the Datasets: float64 points, float32 box).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DAL3_PLANE_DOUBLES 24            // 6 faces x [nx, ny, nz, d]

// `v >= 0` on the bit pattern: the reference's NaN behaviour is part of the contract here (a NaN point is inside
// every box), so the compare is done in integers and does not depend on the floating-point flags of the build
// (only dal3_pointmlp*.hip are compiled with -fno-honor-nans; this header's users keep IEEE NaN semantics).
__device__ __forceinline__ bool ge_zero(float v) {
    const uint32_t b = __float_as_uint(v);
    return (b & 0x7fffffffu) <= 0x7f800000u && ((b >> 31) == 0 || (b << 1) == 0);
}
__device__ __forceinline__ bool ge_zero(double v) {
    const uint64_t b = (uint64_t)__double_as_longlong(v);
    return (b & 0x7fffffffffffffffull) <= 0x7ff0000000000000ull && ((b >> 63) == 0 || (b << 1) == 0);
}

__device__ __forceinline__ bool inside_box_f32(const double* __restrict__ pl, float x, float y, float z) {
    bool in = true;
#pragma unroll
    for (int f = 0; f < 6; ++f) {
        const float nx = (float)pl[f * 4 + 0], ny = (float)pl[f * 4 + 1], nz = (float)pl[f * 4 + 2], d = (float)pl[f * 4 + 3];
        const float sgn = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(x, nx), __fmul_rn(y, ny)), __fmul_rn(z, nz)), d);
        in = in && !ge_zero(sgn);
    }
    return in;
}

__device__ __forceinline__ bool inside_box_f64(const double* __restrict__ pl, double x, double y, double z) {
    bool in = true;
#pragma unroll
    for (int f = 0; f < 6; ++f) {
        const double sgn = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(x, pl[f * 4 + 0]), __dmul_rn(y, pl[f * 4 + 1])),
                                               __dmul_rn(z, pl[f * 4 + 2])), pl[f * 4 + 3]);
        in = in && !ge_zero(sgn);
    }
    return in;
}
