// dal3_device.h — device-side building blocks shared by the gfx950 kernels.
//
// Orientation used everywhere ("Kernel-design notes"): a shared-MLP layer is
//   Y(Cout x P) = W'(Cout x Cin) . X(Cin x P)
// with CHANNELS on the MFMA rows and POINTS on the MFMA columns (= lanes). With the 32x32 C/D
// layout of gfx950 (col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)) the accumulator of
// layer k is then directly the B operand of layer k+1 for v_mfma_f32_32x32x2_f32 (lane half h
// supplies k = h): accumulator register r of channel-tile t is k-step 16t+r, and the A fragment
// of that k-step holds the two weight columns 32t + chan(r,0) and 32t + chan(r,1). A whole
// per-point MLP therefore stays in registers: no LDS round trip, no lane movement.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// logical (B,C,N) view with element strides (mirrors dal3_bcn of include/dal3.h)
struct BCN {
    const float* data;
    int64_t sb, sc, sn;
};

// channel (within a 32-channel tile) held by accumulator register r in lane half h
__host__ __device__ constexpr int tile_chan(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// accumulator tile initialised with the per-channel vector v[0..31] (bias, or a per-crop term):
// register 4q+e of half h is channel 8q + 4h + e, i.e. four contiguous floats per q.
__device__ __forceinline__ f32x16 tile_from_channels(const float* __restrict__ v, int h) {
    f32x16 t;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(v + 8 * q + 4 * h);
        t[4 * q + 0] = x[0];
        t[4 * q + 1] = x[1];
        t[4 * q + 2] = x[2];
        t[4 * q + 3] = x[3];
    }
    return t;
}

__device__ __forceinline__ f32x16 relu16(f32x16 a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = fmaxf(a[r], 0.0f);
    return a;
}

// acc[j] += W'(32 x 32*KT) . X[j]   for the T point tiles of this wave.
// wblk: fragment-packed weights of ONE 32-row output tile: [KT][4 q][64 lanes] float4, where
// element e of (kt,q,lane) is W'[row = lane&31][col = 32kt + tile_chan(4q+e, lane>>5)].
template <int KT, int T>
__device__ __forceinline__ void mma_block(const f32x4* __restrict__ wblk, const f32x16 (&X)[T][KT],
                                          f32x16 (&acc)[T], int lane) {
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 a = wblk[(kt * 4 + q) * 64 + lane];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int j = 0; j < T; ++j) acc[j] = mfma32(a[e], X[j][kt][4 * q + e], acc[j]);
            }
        }
    }
}

// Y = relu(W' X + b') for a Cin=32*KT -> Cout=32*MT layer; w is [MT][KT][4][64] float4.
template <int KT, int MT, int T>
__device__ __forceinline__ void mlp_layer(const f32x4* __restrict__ w, const float* __restrict__ b,
                                          const f32x16 (&X)[T][KT], f32x16 (&Y)[T][MT], int lane) {
    const int h = lane >> 5;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        f32x16 acc[T];
        const f32x16 bias = tile_from_channels(b + 32 * mt, h);
#pragma unroll
        for (int j = 0; j < T; ++j) acc[j] = bias;
        mma_block<KT, T>(w + mt * KT * 256, X, acc, lane);
#pragma unroll
        for (int j = 0; j < T; ++j) Y[j][mt] = relu16(acc[j]);
    }
}

// First layer (raw input, Cin <= 2*KS channels in natural order): in[j][s] is this lane's B
// operand of k-step s (channel 2s + h of point 32j + (lane&31)); w1 is [MT][KS][64] floats.
template <int KS, int MT, int T>
__device__ __forceinline__ void first_layer(const float* __restrict__ w1, const float* __restrict__ b,
                                            const float (&in)[T][KS], f32x16 (&Y)[T][MT], int lane) {
    const int h = lane >> 5;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        f32x16 acc[T];
        const f32x16 bias = tile_from_channels(b + 32 * mt, h);
#pragma unroll
        for (int j = 0; j < T; ++j) acc[j] = bias;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const float a = w1[(mt * KS + s) * 64 + lane];
#pragma unroll
            for (int j = 0; j < T; ++j) acc[j] = mfma32(a, in[j][s], acc[j]);
        }
#pragma unroll
        for (int j = 0; j < T; ++j) Y[j][mt] = relu16(acc[j]);
    }
}

// B operands of the first layer for T point tiles starting at point n0 of item b. Points past
// the end replicate the last point: every consumer is per-point work followed by a max over
// points (or a guarded store), so a duplicate never changes a result.
template <int KS, int T>
__device__ __forceinline__ void load_points(const BCN& x, int64_t b, int n0, int n_pts, int c_in,
                                            float (&in)[T][KS], int lane) {
    const int h = lane >> 5;
#pragma unroll
    for (int j = 0; j < T; ++j) {
        int n = n0 + 32 * j + (lane & 31);
        n = n < n_pts ? n : n_pts - 1;
        const float* p = x.data + b * x.sb + (int64_t)n * x.sn;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int c = 2 * s + h;
            in[j][s] = c < c_in ? p[c * x.sc] : 0.0f;
        }
    }
}

// xor-exchange inside each 32-lane half without an address register (ds_swizzle bit-mask mode)
template <int XOR>
__device__ __forceinline__ float swz_xor(float v) {
    return __int_as_float(__builtin_amdgcn_ds_swizzle(__float_as_int(v), (XOR << 10) | 0x1F));
}

// Channel-wise max over the 32 points of a tile (and the T tiles of the wave), post-ReLU,
// combined across waves / workgroups by an integer atomic max on the bit pattern (values are
// >= +0, so the order of unsigned bit patterns is the order of the floats; dst is zero-filled
// before the launch). Transposing butterfly: each step halves the live registers, 16 swizzles
// instead of 80. dst points at the 32 channels of this output tile for this item.
template <int T>
__device__ __forceinline__ void tile_max_atomic(const f32x16 (&acc)[T], float* __restrict__ dst, int lane) {
    f32x16 m = acc[0];
#pragma unroll
    for (int j = 1; j < T; ++j) {
#pragma unroll
        for (int r = 0; r < 16; ++r) m[r] = fmaxf(m[r], acc[j][r]);
    }
    float v8[8], v4[4], v2[2], v1;
    {
        const bool up = lane & 1;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float send = up ? m[i] : m[i + 8];
            const float keep = up ? m[i + 8] : m[i];
            v8[i] = fmaxf(keep, swz_xor<1>(send));
        }
    }
    {
        const bool up = lane & 2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float send = up ? v8[i] : v8[i + 4];
            const float keep = up ? v8[i + 4] : v8[i];
            v4[i] = fmaxf(keep, swz_xor<2>(send));
        }
    }
    {
        const bool up = lane & 4;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float send = up ? v4[i] : v4[i + 2];
            const float keep = up ? v4[i + 2] : v4[i];
            v2[i] = fmaxf(keep, swz_xor<4>(send));
        }
    }
    {
        const bool up = lane & 8;
        const float send = up ? v2[0] : v2[1];
        const float keep = up ? v2[1] : v2[0];
        v1 = fmaxf(keep, swz_xor<8>(send));
    }
    v1 = fmaxf(v1, swz_xor<16>(v1));
    // lane bits 0..3 selected register bits 3..0
    const int r = ((lane & 1) << 3) | ((lane & 2) << 1) | ((lane & 4) >> 1) | ((lane & 8) >> 3);
    int bits = __float_as_int(v1);
    bits = bits > 0 ? bits : 0;                       // ReLU on the bit pattern (-0.0 and negatives -> +0)
    if ((lane & 16) == 0) atomicMax(reinterpret_cast<int*>(dst) + tile_chan(r, lane >> 5), bits);
}
