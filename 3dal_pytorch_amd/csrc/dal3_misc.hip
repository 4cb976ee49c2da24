// dal3_misc.hip — weight packing (BN fold), batched FC heads, the standalone N-axis max-pool,
// device-side mask compaction + object-point sampling, the two-stage re-centring, box decode.
#include "dal3_device.h"
#include "dal3_kernels.h"

#define BN_EPS 1e-5f

// ================================================================================== packing
__device__ __forceinline__ float bn_scale(const dal3_layer& L, int row) {
    return L.bn_weight ? L.bn_weight[row] / sqrtf(L.bn_var[row] + BN_EPS) : 1.0f;
}

// W'[row][col] = W[row][col_off + col] * gamma/sqrt(var+eps), zero outside the layer
__device__ __forceinline__ float folded_w(const dal3_layer& L, int row, int col, int col_off, int n_cols) {
    if (row >= L.c_out || col >= n_cols) return 0.0f;
    return L.weight[(int64_t)row * L.c_in + col_off + col] * bn_scale(L, row);
}

__global__ void pack_weight_kernel(dal3_layer L, int mode, int col_off, int n_cols, int mt_n, int kt_n,
                                   float* __restrict__ out, int64_t total, int grp_blocks, int64_t grp_a0,
                                   int64_t grp_a1, int64_t grp_stride) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    if (mode == PACK_ROWMAJOR) {
        const int row = (int)(i / n_cols), col = (int)(i % n_cols);
        out[i] = folded_w(L, row, col, col_off, n_cols);
    } else if (mode == PACK_FIRST) {                      // [mt][ks][64]: row = lane&31, k = 2s + (lane>>5)
        const int lane = (int)(i & 63);
        const int s = (int)((i >> 6) % kt_n), mt = (int)((i >> 6) / kt_n);
        out[i] = folded_w(L, 32 * mt + (lane & 31), 2 * s + (lane >> 5), col_off, n_cols);
    } else {                                              // fragment order, see mma_block
        const int e = (int)(i & 3), lane = (int)((i >> 2) & 63), q = (int)((i >> 8) & 3);
        const int blk = (int)(i >> 10);
        const int mt = mode == PACK_FRAG_MT_MAJOR ? blk / kt_n : blk % mt_n;
        const int kt = mode == PACK_FRAG_MT_MAJOR ? blk % kt_n : blk / mt_n;
        int64_t o = i;
        if (grp_blocks > 0) {
            const int g = blk / grp_blocks;
            o = (g == 0 ? grp_a0 : grp_a1 + (int64_t)(g - 1) * grp_stride) + (int64_t)(blk % grp_blocks) * 1024 + (i & 1023);
        }
        out[o] = folded_w(L, 32 * mt + (lane & 31), 32 * kt + tile_chan(4 * q + e, lane >> 5), col_off, n_cols);
    }
}

// 16-bit fragments for v_mfma_f32_32x32x16_{bf16,f16}: a block (mt,kt) is two 1-KiB fragments (k-steps s = 0,1)
// of [64 lanes][8 elements]; the k order inside a step matches the C/D layout of the producing MFMA (dal3_lp.h).
__global__ void pack_weight_lp_kernel(dal3_layer L, int dtype, int kt_major, int col_off, int n_cols, int mt_n, int kt_n,
                                      uint16_t* __restrict__ out, int64_t total, int grp_blocks, int64_t grp_a0,
                                      int64_t grp_a1, int64_t grp_stride) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int j = (int)(i & 7), lane = (int)((i >> 3) & 63), s = (int)((i >> 9) & 1);
    const int blk = (int)(i >> 10);
    const int mt = kt_major ? blk % mt_n : blk / kt_n;
    const int kt = kt_major ? blk / mt_n : blk % kt_n;
    const float v = folded_w(L, 32 * mt + (lane & 31), 32 * kt + 16 * s + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3), col_off,
                             n_cols);
    uint16_t bits;
    if (dtype == DAL3_BF16) {
        const __bf16 h = (__bf16)v;
        bits = __builtin_bit_cast(uint16_t, h);
    } else {
        const _Float16 h = (_Float16)v;
        bits = __builtin_bit_cast(uint16_t, h);
    }
    int64_t o = i;
    if (grp_blocks > 0) {
        const int g = blk / grp_blocks;
        o = (g == 0 ? grp_a0 : grp_a1 + (int64_t)(g - 1) * grp_stride) + (int64_t)(blk % grp_blocks) * 1024 + (i & 1023);
    }
    out[o] = bits;
}

// "f16x3" fragments: every k-step of a block as a PAIR of fp16 fragments, hi = fp16(w), lo = fp16(w - hi). A block (mt, kt)
// is four 1-KiB fragments [k-step 0: hi, lo][k-step 1: hi, lo]; element order inside a fragment as above.
__global__ void pack_weight_x3_kernel(dal3_layer L, int kt_major, int col_off, int n_cols, int mt_n, int kt_n,
                                      uint16_t* __restrict__ out, int64_t total, int grp_blocks, int64_t grp_a0,
                                      int64_t grp_a1, int64_t grp_stride, int64_t grp_last) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int j = (int)(i & 7), lane = (int)((i >> 3) & 63), half = (int)((i >> 9) & 1), s = (int)((i >> 10) & 1);
    const int blk = (int)(i >> 11);
    const int mt = kt_major ? blk % mt_n : blk / kt_n;
    const int kt = kt_major ? blk / mt_n : blk % kt_n;
    const float v = folded_w(L, 32 * mt + (lane & 31), 32 * kt + 16 * s + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3), col_off,
                             n_cols);
    // a folded weight beyond fp16's range (dal3.h, DAL3_F16X3) is packed as NaN: the head's outputs are NaN, not silently wrong
    const _Float16 hi = fabsf(v) < 65504.0f ? (_Float16)v : (_Float16)__builtin_nanf("");
    const _Float16 lo = (_Float16)(v - (float)hi);
    int64_t o = i;
    if (grp_blocks > 0) {
        const int g = blk / grp_blocks;
        // (grp_last >= 0: the last group sits there instead — the decoder's D_15, which has no A block in front of it)
        const bool last = grp_last >= 0 && g == mt_n * kt_n / grp_blocks - 1;
        o = (last ? grp_last : g == 0 ? grp_a0 : grp_a1 + (int64_t)(g - 1) * grp_stride) + (int64_t)(blk % grp_blocks) * 2048 + (i & 2047);
    }
    out[o] = __builtin_bit_cast(uint16_t, half ? lo : hi);
}
hipError_t launch_pack_weight_x3(const dal3_layer& L, int kt_major, int col_off, int n_cols, int mt_n, int kt_n, uint16_t* out,
                                 hipStream_t s, int grp_blocks, int64_t grp_a0, int64_t grp_a1, int64_t grp_stride,
                                 int64_t grp_last) {
    const int64_t total = (int64_t)mt_n * kt_n * 2048;
    hipLaunchKernelGGL(pack_weight_x3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, L, kt_major, col_off, n_cols,
                       mt_n, kt_n, out, total, grp_blocks, grp_a0, grp_a1, grp_stride, grp_last);
    return hipGetLastError();
}

hipError_t launch_pack_weight_lp(const dal3_layer& L, int dtype, int kt_major, int col_off, int n_cols, int mt_n, int kt_n,
                                 uint16_t* out, hipStream_t s, int grp_blocks, int64_t grp_a0, int64_t grp_a1,
                                 int64_t grp_stride) {
    const int64_t total = (int64_t)mt_n * kt_n * 1024;
    hipLaunchKernelGGL(pack_weight_lp_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, L, dtype, kt_major,
                       col_off, n_cols, mt_n, kt_n, out, total, grp_blocks, grp_a0, grp_a1, grp_stride);
    return hipGetLastError();
}

// b' = (b - mean) * s + beta, zero padded to a multiple of 32
__global__ void pack_bias_kernel(dal3_layer L, float* __restrict__ out, int padded) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= padded) return;
    float v = 0.0f;
    if (i < L.c_out) {
        v = L.bias[i];
        if (L.bn_weight) v = (v - L.bn_mean[i]) * bn_scale(L, i) + L.bn_bias[i];
    }
    out[i] = v;
}

hipError_t launch_pack_weight(const dal3_layer& L, int mode, int col_off, int n_cols, int mt_n, int kt_n, float* out,
                              hipStream_t s, int grp_blocks, int64_t grp_a0, int64_t grp_a1, int64_t grp_stride) {
    int64_t total;
    if (mode == PACK_ROWMAJOR) total = (int64_t)L.c_out * n_cols;
    else if (mode == PACK_FIRST) total = (int64_t)mt_n * kt_n * 64;
    else total = (int64_t)mt_n * kt_n * 1024;
    hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, L, mode, col_off,
                       n_cols, mt_n, kt_n, out, total, grp_blocks, grp_a0, grp_a1, grp_stride);
    return hipGetLastError();
}

hipError_t launch_pack_bias(const dal3_layer& L, float* out, hipStream_t s) {
    const int padded = (L.c_out + 31) / 32 * 32;
    hipLaunchKernelGGL(pack_bias_kernel, dim3((padded + 255) / 256), dim3(256), 0, s, L, out, padded);
    return hipGetLastError();
}

// ================================================================================== FC heads
// One workgroup = one 32(out) x 32(items) tile; out channels on MFMA rows, items on columns. The contraction is
// split over the 4 waves (each takes a contiguous quarter of K, in steps of 8 with the same k <-> (lane half,
// element) map on both operands, k = k0 + 4h + e) and summed through LDS in a fixed order, so the result does not
// depend on the batch size or on timing. Two k-steps are loaded before their MFMAs: at small batches this kernel
// is a chain of dependent L2 round trips, not arithmetic.
__global__ __launch_bounds__(256) void fc_kernel(const float* __restrict__ W, const float* __restrict__ bias,
                                                 const float* __restrict__ x, int64_t xs, float* __restrict__ y,
                                                 int64_t ys, int B, int c_in, int c_out, int relu, int n_mt) {
    __shared__ float red[3][16][64];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int h = lane >> 5;
    const int mt = blockIdx.x % n_mt, bt = blockIdx.x / n_mt;
    const int row = 32 * mt + (lane & 31);
    const int item = 32 * bt + (lane & 31);
    const bool row_ok = row < c_out, item_ok = item < B;
    const float* wp = W + (int64_t)(row_ok ? row : 0) * c_in + 4 * h;
    const float* xp = x + (int64_t)(item_ok ? item : 0) * xs + 4 * h;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ch = 32 * mt + tile_chan(r, h);
        acc[r] = (wave == 0 && ch < c_out) ? bias[ch] : 0.0f;
    }
    // (rows past c_out and items past B read row 0 / item 0 — wp and xp are clamped above — and are never stored: a
    // conditional load is a branch per load to hipcc, four per trip of this loop)
    const int chunk = ((c_in + 31) / 32) * 8;              // per-wave share of K, a multiple of 8
    const int k_end = min(c_in, (wave + 1) * chunk);
    int k0 = wave * chunk;
    for (; k0 + 16 <= k_end; k0 += 16) {
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(wp + k0);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(wp + k0 + 8);
        const f32x4 b0 = *reinterpret_cast<const f32x4*>(xp + k0);
        const f32x4 b1 = *reinterpret_cast<const f32x4*>(xp + k0 + 8);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = mfma32(a0[e], b0[e], acc);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = mfma32(a1[e], b1[e], acc);
    }
    for (; k0 < k_end; k0 += 8) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(wp + k0);
        const f32x4 bv = *reinterpret_cast<const f32x4*>(xp + k0);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = mfma32(a[e], bv[e], acc);
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[wave - 1][r][lane] = acc[r];
    }
    __syncthreads();
    if (wave > 0 || !item_ok) return;
    float* yp = y + (int64_t)item * ys;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ch = 32 * mt + tile_chan(r, h);
        const float v = ((acc[r] + red[0][r][lane]) + red[1][r][lane]) + red[2][r][lane];
        if (ch < c_out) yp[ch] = relu ? (v < 0.0f ? 0.0f : v) : v;   // (NaN stays NaN, as torch's relu: a non-finite item, dal3.h)
    }
}

hipError_t launch_fc(const float* W, const float* bias, const float* x, int64_t xs, float* y, int64_t ys, int B,
                     int c_in, int c_out, int relu, hipStream_t s) {
    if (c_in % 8 != 0 || xs % 4 != 0) return hipErrorInvalidValue;
    const int n_mt = (c_out + 31) / 32, n_bt = (B + 31) / 32;
    hipLaunchKernelGGL(fc_kernel, dim3((unsigned)(n_mt * n_bt)), dim3(256), 0, s, W, bias, x, xs, y, ys, B, c_in, c_out,
                       relu, n_mt);
    return hipGetLastError();
}

// ================================================================================== max-pool over N
// torch.max(x, 2)[0] for contiguous (rows, n): one wave per row, 16-byte loads coalesced along N (1 KiB per
// wave-instruction: 4 fp32 or 8 bf16 / fp16 values per lane), per-lane running max, then a wave64 butterfly. HBM-bound.
// A row that holds a NaN gives NaN, as torch.max does (fmaxf alone would drop it): every lane keeps a sticky flag.
// 16-bit rows: the values are widened exactly (bf16: a 16-bit shift; fp16: v_cvt_f32_f16), compared as fp32 and the
// maximum — one of the inputs — is narrowed back exactly.
template <int DT> struct MpElem;                            // DT: DAL3_F32 / DAL3_BF16 / DAL3_F16
template <> struct MpElem<DAL3_F32> {
    typedef float T;
    static constexpr int PER16 = 4;
    static __device__ __forceinline__ void unpack(const u32x4& v, float (&f)[8]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = __uint_as_float(v[i]);
    }
    static __device__ __forceinline__ float load1(const T* p) { return *p; }
    static __device__ __forceinline__ void store1(T* p, float v) { *p = v; }
};
template <> struct MpElem<DAL3_BF16> {
    typedef uint16_t T;
    static constexpr int PER16 = 8;
    static __device__ __forceinline__ void unpack(const u32x4& v, float (&f)[8]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = __uint_as_float(v[i] << 16);
            f[2 * i + 1] = __uint_as_float(v[i] & 0xFFFF0000u);
        }
    }
    static __device__ __forceinline__ float load1(const T* p) { return __uint_as_float((uint32_t)*p << 16); }
    static __device__ __forceinline__ void store1(T* p, float v) { *p = (uint16_t)(__float_as_uint(v) >> 16); }
};
template <> struct MpElem<DAL3_F16> {
    typedef uint16_t T;
    static constexpr int PER16 = 8;
    static __device__ __forceinline__ float widen(uint32_t h) {
        return (float)__builtin_bit_cast(_Float16, (uint16_t)h);
    }
    static __device__ __forceinline__ void unpack(const u32x4& v, float (&f)[8]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = widen(v[i] & 0xFFFFu);
            f[2 * i + 1] = widen(v[i] >> 16);
        }
    }
    static __device__ __forceinline__ float load1(const T* p) { return widen(*p); }
    static __device__ __forceinline__ void store1(T* p, float v) { *p = __builtin_bit_cast(uint16_t, (_Float16)v); }
};

// R rows per wave, NL 16-byte loads per lane and row, all R * NL loads of a wave issued before the first is used
// (NL = 0: any row length, loads in batches of four, the last batch's indices clamped — a repeated element cannot
// change a maximum — so that no load is conditional: to hipcc a conditional load is a branch).
template <int DT, int R, int NL, bool VEC>
__global__ __launch_bounds__(256) void maxpool_rows_kernel(const void* __restrict__ xv, int64_t rows, int64_t n,
                                                           void* __restrict__ outv) {
    typedef MpElem<DT> E;
    typedef typename E::T T;
    const T* x = static_cast<const T*>(xv);
    T* out = static_cast<T*>(outv);
    const int lane = threadIdx.x & 63;
    const int64_t wave0 = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t n_waves = (int64_t)gridDim.x * 4;
    for (int64_t row0 = wave0 * R; row0 < rows; row0 += n_waves * R) {
        float m[R];
        bool nan = false;
#pragma unroll
        for (int r = 0; r < R; ++r) m[r] = -INFINITY;
        if (VEC) {                                          // n % PER16 == 0 and 16-byte aligned rows
            const int64_t nv = n / E::PER16;
            auto take = [&](const u32x4& v, int r) {
                float f[8];
                E::unpack(v, f);
#pragma unroll
                for (int k = 0; k < E::PER16; ++k) {
                    nan |= f[k] != f[k];
                    m[r] = fmaxf(m[r], f[k]);
                }
            };
            if (NL > 0) {                                   // nv == 64 * NL: straight-line
                u32x4 v[R][NL];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int64_t row = row0 + r < rows ? row0 + r : rows - 1;     // (a repeated row is stored once)
                    const u32x4* p = reinterpret_cast<const u32x4*>(x + row * n);
#pragma unroll
                    for (int u = 0; u < NL; ++u) v[r][u] = __builtin_nontemporal_load(p + lane + u * 64);
                }
#pragma unroll
                for (int r = 0; r < R; ++r)
#pragma unroll
                    for (int u = 0; u < NL; ++u) take(v[r][u], r);
            } else {
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const int64_t row = row0 + r < rows ? row0 + r : rows - 1;
                    const u32x4* p = reinterpret_cast<const u32x4*>(x + row * n);
                    for (int64_t i = lane; i < nv; i += 4 * 64) {
                        u32x4 v[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const int64_t j = i + u * 64;
                            v[u] = __builtin_nontemporal_load(p + (j < nv ? j : nv - 1));
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) take(v[u], r);
                    }
                }
            }
        } else {                                            // any n, any alignment: one element per lane and trip
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int64_t row = row0 + r < rows ? row0 + r : rows - 1;
                for (int64_t i = lane; i < n; i += 64) {
                    const float f = E::load1(x + row * n + i);
                    nan |= f != f;
                    m[r] = fmaxf(m[r], f);
                }
            }
        }
        // (the NaN flag is per lane and row-blind for R > 1: redo the flag per row only in that rare case)
        const bool any_nan = __any(nan);
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float v = m[r];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
            m[r] = v;
        }
        if (any_nan) {                                      // rare: which of the wave's rows hold the NaN
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const int64_t row = row0 + r < rows ? row0 + r : rows - 1;
                bool rn = false;
                for (int64_t i = lane; i < n; i += 64) {
                    const float f = E::load1(x + row * n + i);
                    rn |= f != f;
                }
                if (__any(rn)) m[r] = __uint_as_float(0x7FC00000u);
            }
        }
        if (lane < R && row0 + lane < rows) {
            float v = m[0];
#pragma unroll
            for (int r = 1; r < R; ++r) v = lane == r ? m[r] : v;
            E::store1(out + row0 + lane, v);
        }
    }
}

hipError_t launch_maxpool_n(const void* x, int dtype, int64_t rows, int64_t n, void* out, hipStream_t s) {
    if (rows <= 0 || n <= 0) return hipErrorInvalidValue;
    // one wave per row (fp32 rows) / per two rows (16-bit rows: a 1024-value row is 2 KiB, two loads per lane — too
    // little per wave to hide its butterfly and store behind), as many workgroups as that takes: with the grid capped
    // at 16 workgroups per CU (waves walking 256 rows each) the fp32 kernel read 6.5 TB/s, uncapped 7.1 (one box,
    // (4096, 1024, 1024) fp32) — a wave's next row starts behind its butterfly and store, a fresh wave's loads do not.
    // (Two fp32 rows per wave: 6.9 in the same process; plain instead of non-temporal loads: 6.5; a one-off A/B script of round 3, since deleted)
    const int per16 = dtype == DAL3_F32 ? 4 : 8;
    const bool vec = (n % per16 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15) == 0);
    const int R = dtype == DAL3_F32 ? 1 : 2;
    int64_t blocks = ((rows + R - 1) / R + 3) / 4;
    if (blocks > 0x7fffffff) blocks = 0x7fffffff;
    const dim3 grid((unsigned)blocks), block(256);
    const int64_t nl = vec && (n / per16) % 64 == 0 ? n / per16 / 64 : 0;      // 16-byte loads per lane and row
#define MP_LAUNCH(DT, RR)                                                                                              \
    do {                                                                                                               \
        if (!vec) hipLaunchKernelGGL((maxpool_rows_kernel<DT, RR, 0, false>), grid, block, 0, s, x, rows, n, out);     \
        else if (nl == 4) hipLaunchKernelGGL((maxpool_rows_kernel<DT, RR, 4, true>), grid, block, 0, s, x, rows, n, out); \
        else if (nl == 2) hipLaunchKernelGGL((maxpool_rows_kernel<DT, RR, 2, true>), grid, block, 0, s, x, rows, n, out); \
        else hipLaunchKernelGGL((maxpool_rows_kernel<DT, RR, 0, true>), grid, block, 0, s, x, rows, n, out);          \
    } while (0)
    switch (dtype) {
        case DAL3_F32: MP_LAUNCH(DAL3_F32, 1); break;
        case DAL3_BF16: MP_LAUNCH(DAL3_BF16, 2); break;
        case DAL3_F16: MP_LAUNCH(DAL3_F16, 2); break;
        default: return hipErrorInvalidValue;
    }
#undef MP_LAUNCH
    return hipGetLastError();
}

// ================================================================================== mask -> object points
__device__ __forceinline__ uint32_t hash_key(uint64_t seed, uint64_t item, uint32_t i) {
    uint64_t z = seed ^ (item * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)i << 32 | i);
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (uint32_t)(z >> 32);
}

// exclusive prefix of `flag` over the 256 threads of the block (thread order); *total = block sum
__device__ __forceinline__ int block_scan_flag(bool flag, int* total, int* lds_wave /*[8]*/) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long bal = __ballot(flag);
    const int within = __popcll(bal & ((1ull << lane) - 1ull));
    __syncthreads();                                        // lds_wave reuse
    if (lane == 0) lds_wave[wave] = __popcll(bal);
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int wv = 0; wv < 4; ++wv) {
        const int c = lds_wave[wv];
        base += wv < wave ? c : 0;
        tot += c;
    }
    *total = tot;
    return base + within;
}

__device__ __forceinline__ int block_sum(int v, int* lds_wave) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) lds_wave[threadIdx.x >> 6] = v;
    __syncthreads();
    return lds_wave[0] + lds_wave[1] + lds_wave[2] + lds_wave[3];
}

__global__ __launch_bounds__(256) void segment_counts_kernel(const uint8_t* __restrict__ mask, int N,
                                                             int32_t* __restrict__ counts) {
    __shared__ int lds_wave[8];
    const int64_t b = blockIdx.x;
    int c = 0;
    for (int i = threadIdx.x; i < N; i += 256) c += mask[b * N + i] ? 1 : 0;
    c = block_sum(c, lds_wave);
    if (threadIdx.x == 0) counts[b] = c;
}

// One 256-thread workgroup per item:
//  1. ordered compaction of the segmented point indices (== torch.nonzero order) into pos[b,:count]
//  2. choose M of them: count == 0 -> zero row; CHOICE sampler -> pos[choice[k]];
//     count < M -> every segmented point plus duplicates (the reference tops up with random
//     duplicates; every consumer is a max over points, so which duplicates is immaterial);
//     count >= M -> the M smallest 32-bit hash keys (uniform subset without replacement), found by
//     a radix select on the key value, ties taken in index order.
//  3. gather obj_pts[b,k,:C] = pts[b,:C,idx[k]]
__global__ __launch_bounds__(256) void compact_sample_kernel(const uint8_t* __restrict__ mask, BCN pts, int N, int C,
                                                             int M, int sampler, const int32_t* __restrict__ choice,
                                                             uint64_t seed, int64_t item_offset, const int64_t* __restrict__ step,
                                                             int32_t* __restrict__ counts, int32_t* __restrict__ pos,
                                                             int32_t* __restrict__ obj_idx, float* __restrict__ obj_pts) {
    // `step` (device, optional): a draw counter kept in device memory, so that a training step captured into a hipGraph
    // draws fresh object points on every replay (a kernel ARGUMENT would be frozen in the graph)
    if (step) seed += (uint64_t)(*step) * 0xD6E8FEB86659FD93ull;
    __shared__ int lds_wave[8];
    __shared__ int lds_hist[256];
    const int64_t b = blockIdx.x;
    const int tid = threadIdx.x;
    int32_t* pos_b = pos + b * N;
    int32_t* idx_b = obj_idx + b * M;
    int count = 0;
    for (int base = 0; base < N; base += 256) {
        const int i = base + tid;
        const bool f = i < N && mask[b * N + i];
        int tot;
        const int p = block_scan_flag(f, &tot, lds_wave);
        if (f) pos_b[count + p] = i;
        count += tot;
    }
    if (tid == 0) counts[b] = count;
    __syncthreads();                                        // pos_b visible to the whole block (same CU)
    __threadfence_block();

    if (count == 0) {
        for (int k = tid; k < M; k += 256) idx_b[k] = 0;
        for (int k = tid; k < M * C; k += 256) obj_pts[b * M * C + k] = 0.0f;
        return;
    }
    if (sampler == DAL3_SAMPLER_CHOICE) {
        for (int k = tid; k < M; k += 256) {
            int c = choice[b * M + k];
            c = c < 0 ? 0 : (c >= count ? count - 1 : c);
            idx_b[k] = pos_b[c];
        }
    } else if (count < M) {
        for (int k = tid; k < M; k += 256) idx_b[k] = pos_b[k % count];
    } else {
        const uint64_t item = (uint64_t)(item_offset + b);
        // thr = the M-th smallest key, by a radix select: four rounds of one byte each (a histogram of the byte over the
        // keys that match the prefix found so far, then the bin holding the M-th) instead of 32 bisection steps with a
        // block-wide count each
        uint32_t prefix = 0;
        int remaining = M;
        for (int shift = 24; shift >= 0; shift -= 8) {
            __syncthreads();                                // lds_hist / lds_wave reuse
            lds_hist[tid] = 0;
            __syncthreads();
            for (int i = tid; i < count; i += 256) {
                const uint32_t key = hash_key(seed, item, (uint32_t)i);
                if (shift == 24 || (key >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&lds_hist[(key >> shift) & 255u], 1);
            }
            __syncthreads();
            // inclusive prefix sum of the 256 bins (bin = thread); the wanted bin is the first with cum >= remaining
            const int lane = tid & 63, wave = tid >> 6;
            int cum = lds_hist[tid];
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int up = __shfl_up(cum, off);
                if (lane >= off) cum += up;
            }
            if (lane == 63) lds_wave[wave] = cum;
            __syncthreads();
            int base = 0;
#pragma unroll
            for (int wv = 0; wv < 4; ++wv) base += wv < wave ? lds_wave[wv] : 0;
            cum += base;
            const int before = cum - lds_hist[tid];
            if (before < remaining && cum >= remaining) {   // exactly one thread
                lds_wave[4] = tid;
                lds_wave[5] = before;
            }
            __syncthreads();
            prefix |= (uint32_t)lds_wave[4] << shift;
            remaining -= lds_wave[5];
        }
        const uint32_t thr = prefix;
        int n_less = 0;
        for (int i = tid; i < count; i += 256) n_less += hash_key(seed, item, (uint32_t)i) < thr ? 1 : 0;
        n_less = block_sum(n_less, lds_wave);
        const int need = M - n_less;                        // taken from the ties, in index order
        int out_less = 0, out_tie = 0;
        for (int base = 0; base < count; base += 256) {
            const int i = base + tid;
            const uint32_t key = i < count ? hash_key(seed, item, (uint32_t)i) : 0xFFFFFFFFu;
            const bool fl = i < count && key < thr;
            const bool ft = i < count && key == thr;
            int tl, tt;
            const int pl = block_scan_flag(fl, &tl, lds_wave);
            const int pt = block_scan_flag(ft, &tt, lds_wave);
            if (fl) idx_b[out_less + pl] = pos_b[i];
            if (ft && out_tie + pt < need) idx_b[n_less + out_tie + pt] = pos_b[i];
            out_less += tl;
            out_tie += tt;
        }
    }
    __syncthreads();
    __threadfence_block();
    for (int k = tid; k < M; k += 256) {
        const int n = idx_b[k];
        const int64_t o = b * pts.sb + (int64_t)n * pts.sn;
        for (int c = 0; c < C; ++c) obj_pts[(b * M + k) * C + c] = bcn_value(pts, o + c * pts.sc);   // (16-bit points: widened here)
    }
}

// The same, for items whose positions and hash keys fit the LDS (N <= CS_LDS_MAX_N: every BASELINE shape — 1024, 4096,
// 5 x 1024 points). Round 6: at small batches the kernel above is a chain of ~150 barriers per item (a block scan per
// 256 mask bytes, six passes that each recompute the 64-bit hash of every segmented point, a two-scan output pass per
// 256 candidates, positions bounced through global memory): 34-42 us for 64-128 dynamic items, 7-9 % of a 16-bit step.
// Here a thread owns a CONTIGUOUS piece of the mask (its flags in one register), ONE block scan orders the positions
// into LDS, the keys are hashed once into LDS, the radix select reads them there, and the output pass is one more
// block scan over contiguous pieces of the candidates. Same definition, same order of obj_idx: bit-identical outputs
// (tests/test_gpu_parity.py runs both kernels on the same masks: DAL3_BCN_NO_LDS_SAMPLER in the view's flags selects
// the kernel above).
#define CS_LDS_MAX_N 7936

// exclusive prefix of v over the 256 threads (thread order) and the block total; lds_wave: 4 ints
__device__ __forceinline__ int block_scan_int(int v, int* total, int* lds_wave) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int cum = v;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int up = __shfl_up(cum, off);
        if (lane >= off) cum += up;
    }
    __syncthreads();                                        // lds_wave reuse
    if (lane == 63) lds_wave[wave] = cum;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int wv = 0; wv < 4; ++wv) {
        const int c = lds_wave[wv];
        base += wv < wave ? c : 0;
        tot += c;
    }
    *total = tot;
    return base + cum - v;
}

template <int CT>                                           // CT: channels per point at compile time (3, 4, 8), 0 = any
__global__ __launch_bounds__(256) void compact_sample_lds_kernel(const uint8_t* __restrict__ mask, BCN pts, int N, int C, int M,
                                                                 int sampler, const int32_t* __restrict__ choice, uint64_t seed,
                                                                 int64_t item_offset, const int64_t* __restrict__ step,
                                                                 int32_t* __restrict__ counts, int32_t* __restrict__ obj_idx,
                                                                 float* __restrict__ obj_pts) {
    extern __shared__ int cs_smem[];                        // pos[N] | key[N] | idx[M]
    int* s_pos = cs_smem;
    uint32_t* s_key = reinterpret_cast<uint32_t*>(cs_smem + N);
    int* s_idx = cs_smem + 2 * N;                           // the chosen indices stay here for the gather (no global round trip)
    __shared__ int lds_wave[8];
    __shared__ int lds_hist[256];
    if (step) seed += (uint64_t)(*step) * 0xD6E8FEB86659FD93ull;
    const int64_t b = blockIdx.x;
    const int tid = threadIdx.x;
    int32_t* idx_b = obj_idx + b * M;
    // 1. ordered compaction: thread t owns the mask bytes [t * chunk, (t + 1) * chunk), chunk <= 31
    const int chunk = (N + 255) / 256;
    const int i0 = tid * chunk;
    uint32_t bits = 0;
    {
        const uint8_t* mb = mask + b * N;
        for (int j = 0; j < chunk; ++j) {
            const int i = i0 + j;
            bits |= (i < N && mb[i]) ? (1u << j) : 0u;
        }
    }
    int count;
    int p = block_scan_int(__popc(bits), &count, lds_wave);
    for (uint32_t m = bits; m; m &= m - 1) s_pos[p++] = i0 + (__ffs((int)m) - 1);
    if (tid == 0) counts[b] = count;
    __syncthreads();                                        // s_pos complete

    if (count == 0) {
        for (int k = tid; k < M; k += 256) idx_b[k] = 0;
        for (int k = tid; k < M * C; k += 256) obj_pts[b * M * C + k] = 0.0f;
        return;
    }
    if (sampler == DAL3_SAMPLER_CHOICE) {
        for (int k = tid; k < M; k += 256) {
            int c = choice[b * M + k];
            c = c < 0 ? 0 : (c >= count ? count - 1 : c);
            s_idx[k] = s_pos[c];
        }
    } else if (count < M) {
        for (int k = tid; k < M; k += 256) s_idx[k] = s_pos[k % count];
    } else {
        const uint64_t item = (uint64_t)(item_offset + b);
        for (int i = tid; i < count; i += 256) s_key[i] = hash_key(seed, item, (uint32_t)i);
        // thr = the M-th smallest key: four radix rounds of one byte each over the keys that match the prefix so far
        uint32_t prefix = 0;
        int remaining = M;
        for (int shift = 24; shift >= 0; shift -= 8) {
            __syncthreads();                                // s_key complete (first round) / lds_hist, lds_wave reuse
            lds_hist[tid] = 0;
            __syncthreads();
            for (int i = tid; i < count; i += 256) {
                const uint32_t key = s_key[i];
                if (shift == 24 || (key >> (shift + 8)) == (prefix >> (shift + 8))) atomicAdd(&lds_hist[(key >> shift) & 255u], 1);
            }
            __syncthreads();
            const int h = lds_hist[tid];
            int tot;
            const int before = block_scan_int(h, &tot, lds_wave);
            if (before < remaining && before + h >= remaining) {   // exactly one thread
                lds_wave[4] = tid;
                lds_wave[5] = before;
            }
            __syncthreads();
            prefix |= (uint32_t)lds_wave[4] << shift;
            remaining -= lds_wave[5];
        }
        const uint32_t thr = prefix;
        // output: the keys below thr in index order, then the first (M - n_less) ties in index order. Thread t owns the
        // candidates [t * cc, (t + 1) * cc); one scan carries both counts (count <= 7936 < 2^16)
        const int cc = (count + 255) / 256;
        const int j0 = tid * cc, j1 = min(count, j0 + cc);
        int nl = 0, nt = 0;
        for (int i = j0; i < j1; ++i) {
            const uint32_t key = s_key[i];
            nl += key < thr ? 1 : 0;
            nt += key == thr ? 1 : 0;
        }
        int tot;
        const int pre = block_scan_int(nl | (nt << 16), &tot, lds_wave);
        int pl = pre & 0xffff, pt = pre >> 16;
        const int n_less = tot & 0xffff, need = M - n_less;
        for (int i = j0; i < j1; ++i) {
            const uint32_t key = s_key[i];
            if (key < thr) s_idx[pl++] = s_pos[i];
            else if (key == thr) {
                if (pt < need) s_idx[n_less + pt] = s_pos[i];
                ++pt;
            }
        }
    }
    __syncthreads();                                        // s_idx complete
    // gather, four points per thread in flight (a dependent index -> point chain each: one at a time this loop was most of
    // the kernel at M = 2560)
    constexpr int U = 4;
    const int Cn = CT ? CT : C;
    for (int k0 = tid; k0 < M; k0 += 256 * U) {
        int n[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int k = k0 + 256 * u;
            n[u] = s_idx[k < M ? k : 0];               // (past the end: a valid index, its point is loaded and dropped)
            if (k < M) idx_b[k] = n[u];
        }
        if (CT) {
            float v[U][CT ? CT : 1];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int64_t o = b * pts.sb + (int64_t)n[u] * pts.sn;
#pragma unroll
                for (int c = 0; c < CT; ++c) v[u][c] = bcn_value(pts, o + c * pts.sc);   // (16-bit points: widened here)
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = k0 + 256 * u;
                if (k < M) {
#pragma unroll
                    for (int c = 0; c < CT; ++c) obj_pts[(b * M + k) * CT + c] = v[u][c];
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = k0 + 256 * u;
                if (k >= M) continue;
                const int64_t o = b * pts.sb + (int64_t)n[u] * pts.sn;
                for (int c = 0; c < Cn; ++c) obj_pts[(b * M + k) * Cn + c] = bcn_value(pts, o + c * pts.sc);
            }
        }
    }
}

// Zero-fill as a KERNEL, not hipMemsetAsync: on ROCm 7.0 a memset node of a captured hipGraph does not keep its
// fill value reliably — after other work on the process (a larger eager launch, a big allocation) replays of the graph
// filled the max-pool accumulators with an arbitrary 32-bit pattern instead of 0 (a one-off reproducer of round 2, since deleted: about half of
// the processes, persistent until the graph is recorded again; torch-only graphs and eager calls were never affected).
// A kernel node carries its arguments by value. n_words 32-bit words, 16-byte aligned pointers take the wide path.
__global__ __launch_bounds__(256) void fill_words_kernel(uint32_t* __restrict__ p, size_t n_words, uint32_t value) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
        const size_t n4 = n_words / 4;
        uint4 v;
        v.x = v.y = v.z = v.w = value;
        for (size_t k = i; k < n4; k += stride) reinterpret_cast<uint4*>(p)[k] = v;
        for (size_t k = 4 * n4 + i; k < n_words; k += stride) p[k] = value;
    } else {
        for (size_t k = i; k < n_words; k += stride) p[k] = value;
    }
}
// The zero-fill of a pooled-feature accumulator (B, C) that also applies the library's contract for NON-FINITE inputs
// (include/dal3.h): one workgroup per item scans the item's n_pts x c_in input values; an item with a NaN / +-Inf gets
// the quiet-NaN pattern in all C channels instead of zeros. The pooling kernels combine with an integer atomicMax on the
// bit pattern, and that pattern is above every finite value's, so the NaN survives the pooling whatever the kernels
// compute for the item — as torch.max returns NaN for a row that holds one (static_model.py:284,334).
//
// With `list` it also builds the point heads' WORKLIST (dal3_pointmlp.hip, point_head_pers_kernel): entry
// {item, tile, n_eff, 0} for every 32-point tile that holds distinct points (n_eff = min(max(distinct[item], 1),
// n_pts), or n_pts without `distinct`), items in order; ctl[0] = number of entries, ctl[1] = 0 (the cursor). Every
// workgroup sums the tile counts of the items in front of its own from `distinct` itself (B ints, L2-resident: 16 per
// thread at B = 4096) — deterministic, no atomics, nothing to zero beforehand, and no launch of its own.
__global__ __launch_bounds__(256) void nonfinite_rows_kernel(BCN x, int n_pts, int c_in, uint32_t* __restrict__ dst, int C,
                                                             const int32_t* __restrict__ distinct,
                                                             uint32_t* __restrict__ ctl, u32x4* __restrict__ list) {
    __shared__ int lds_wave[8];
    const int64_t b = blockIdx.x;
    if (list) {
        auto eff = [&](int i) {
            if (!distinct) return n_pts;
            const int d = distinct[i];
            return d <= 0 ? 1 : (d < n_pts ? d : n_pts);
        };
        int before = 0;
        for (int i = threadIdx.x; i < (int)b; i += 256) before += (eff(i) + 31) >> 5;
        before = block_sum(before, lds_wave);
        const int n_eff = eff((int)b), n_t = (n_eff + 31) >> 5;
        for (int t = threadIdx.x; t < n_t; t += 256) {
            u32x4 e;
            e[0] = (uint32_t)b;
            e[1] = (uint32_t)t;
            e[2] = (uint32_t)n_eff;
            e[3] = 0u;
            list[before + t] = e;
        }
        if (b == gridDim.x - 1 && threadIdx.x == 0) {
            ctl[0] = (uint32_t)(before + n_t);
            ctl[1] = 0u;
        }
    }
    const int64_t base = b * x.sb;
    bool bad = false;
    const int total = n_pts * c_in;
    if (x.sc == 1 && x.sn == c_in) {                        // point-major storage: the item is one contiguous run
        // round 6: 16 bytes per thread and trip where the run allows (an exponent of all ones on the raw bits: NaN or
        // +-Inf in fp32 / bf16 / fp16 alike), four trips' loads in flight; one element per trip made this fill 12-26 us
        // for a 5120-point item — 80 dependent-free but un-unrolled scalar loads per thread
        const int es = x.dtype == 0 ? 4 : 2;               // bytes per stored element
        const char* p0 = reinterpret_cast<const char*>(x.data) + base * es;
        const int bytes = total * es;
        int done = 0;
        if ((reinterpret_cast<uintptr_t>(p0) & 15) == 0) {
            const uint4* v4 = reinterpret_cast<const uint4*>(p0);
            const int n16 = bytes >> 4;
            const uint32_t m32 = x.dtype == 0 ? 0x7F800000u : (x.dtype == 1 ? 0x7F807F80u : 0x7C007C00u);
            auto word_bad = [&](uint32_t w) {
                if (x.dtype == 0) return (w & m32) == m32;
                const uint32_t lo = m32 & 0xffffu, a = w & m32;
                return (a & 0xffffu) == lo || (a >> 16) == lo;
            };
            int i = threadIdx.x;
            for (; i + 768 < n16; i += 1024) {
                uint4 q[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) q[u] = v4[i + 256 * u];
#pragma unroll
                for (int u = 0; u < 4; ++u) bad |= word_bad(q[u].x) | word_bad(q[u].y) | word_bad(q[u].z) | word_bad(q[u].w);
            }
            for (; i < n16; i += 256) {
                const uint4 q = v4[i];
                bad |= word_bad(q.x) | word_bad(q.y) | word_bad(q.z) | word_bad(q.w);
            }
            done = (n16 << 4) / es;
        }
        for (int i = done + threadIdx.x; i < total; i += 256) bad |= bits_nonfinite(bcn_value(x, base + i));
    } else {
        for (int i = threadIdx.x; i < total; i += 256) {
            const int n = i / c_in, c = i - n * c_in;
            bad |= bits_nonfinite(bcn_value(x, base + (int64_t)n * x.sn + (int64_t)c * x.sc));
        }
    }
    const uint32_t v = __syncthreads_or(bad) ? (uint32_t)DAL3_QNAN_BITS : 0u;
    for (int c = threadIdx.x; c < C; c += 256) dst[b * C + c] = v;
}
hipError_t launch_nonfinite_rows(BCN x, int B, int n_pts, int c_in, float* dst, int C, hipStream_t s, const int32_t* distinct,
                                 void* worklist) {
    uint32_t* ctl = static_cast<uint32_t*>(worklist);
    u32x4* list = worklist ? reinterpret_cast<u32x4*>(static_cast<char*>(worklist) + 256) : nullptr;
    hipLaunchKernelGGL(nonfinite_rows_kernel, dim3((unsigned)B), dim3(256), 0, s, x, n_pts, c_in,
                       reinterpret_cast<uint32_t*>(dst), C, distinct, ctl, list);
    return hipGetLastError();
}

hipError_t launch_fill_words(void* p, size_t n_words, uint32_t value, hipStream_t s) {
    if (n_words == 0) return hipSuccess;
    const size_t want = (n_words / 4 + 255) / 256;
    const unsigned grid = (unsigned)(want < 1 ? 1 : (want > 2048 ? 2048 : want));
    hipLaunchKernelGGL(fill_words_kernel, dim3(grid), dim3(256), 0, s, static_cast<uint32_t*>(p), n_words, value);
    return hipGetLastError();
}

hipError_t launch_segment_counts(const uint8_t* mask, int B, int N, int32_t* counts, hipStream_t s) {
    hipLaunchKernelGGL(segment_counts_kernel, dim3(B), dim3(256), 0, s, mask, N, counts);
    return hipGetLastError();
}

hipError_t launch_compact_sample(const uint8_t* mask, BCN pts, int B, int N, int C, int M, int sampler,
                                 const int32_t* choice, uint64_t seed, int64_t item_offset, int32_t* counts,
                                 int32_t* pos, int32_t* obj_idx, float* obj_pts, hipStream_t s, const int64_t* step) {
#ifndef DAL3_CS_LDS
#define DAL3_CS_LDS 1
#endif
    const size_t lds = (size_t)N * 8 + (size_t)M * 4;      // positions + keys + chosen indices (<= 62 KiB: no attribute needed)
    if (DAL3_CS_LDS && N <= CS_LDS_MAX_N && lds <= 62 * 1024 && !(pts.flags & DAL3_BCN_NO_LDS_SAMPLER)) {
#define CS_GO(CT) hipLaunchKernelGGL(compact_sample_lds_kernel<CT>, dim3(B), dim3(256), lds, s, mask, pts, N, C, M, sampler, choice, \
                                     seed, item_offset, step, counts, obj_idx, obj_pts)
        if (C == 3) CS_GO(3);
        else if (C == 4) CS_GO(4);
        else if (C == 8) CS_GO(8);
        else CS_GO(0);
#undef CS_GO
        return hipGetLastError();
    }
    hipLaunchKernelGGL(compact_sample_kernel, dim3(B), dim3(256), 0, s, mask, pts, N, C, M, sampler, choice, seed,
                       item_offset, step, counts, pos, obj_idx, obj_pts);
    return hipGetLastError();
}

// ================================================================================== decode / parse
__constant__ float c_mean_size[9] = {DAL3_MEAN_SIZE_VALUES};
__constant__ double c_mean_size_d[9] = {DAL3_MEAN_SIZE_VALUES};

// One thread per crop. fp32 where the reference computes in fp32 tensors (parse_output_to_tensors),
// fp64 where its eval driver computes in NumPy float64 (class2angle / class2size), rounded to fp32
// on store.
__device__ __forceinline__ void decode_one(const DecodeArgs& d, int b) {
    float* bp = d.box_pred + (int64_t)b * 39;
    float c[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        c[k] = bp[k] + (d.center_add ? d.center_add[b * d.ca_stride + k] : 0.0f);
        if (d.center_inplace) bp[k] = c[k];
        if (d.center) d.center[b * 3 + k] = c[k];
    }
    const float hr_scale = (float)(3.14159265358979323846 / 12.0);
    int hc = 0;
    float hbest = bp[3];
    float hr_sel = 0.0f;
#pragma unroll
    for (int i = 0; i < 12; ++i) {
        const float hr = bp[15 + i] * hr_scale;
        if (d.heading_residuals) d.heading_residuals[b * 12 + i] = hr;
        if (i == 0) hr_sel = hr;
        if (bp[3 + i] > hbest) {                                // first maximum, as np.argmax
            hbest = bp[3 + i];
            hc = i;
            hr_sel = hr;
        }
    }
    int sc = 0;
    float sbest = bp[27];
#pragma unroll
    for (int i = 1; i < 3; ++i) {
        if (bp[27 + i] > sbest) {
            sbest = bp[27 + i];
            sc = i;
        }
    }
    float sr[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        sr[i] = bp[30 + i] * c_mean_size[i];
        if (d.size_residuals) d.size_residuals[b * 9 + i] = sr[i];
    }
    if (!d.boxes7) return;
    double ang = (double)hc * (2.0 * 3.14159265358979323846 / 12.0) + (double)hr_sel;
    if (ang > 3.14159265358979323846) ang -= 2.0 * 3.14159265358979323846;
    if (d.yaw_base) ang += (double)d.yaw_base[b * d.yaw_stride];
    float* o = d.boxes7 + (int64_t)b * 7;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        o[k] = c[k] + (d.boxes_center_add ? d.boxes_center_add[b * d.bca_stride + k] : 0.0f);
        o[3 + k] = (float)(c_mean_size_d[sc * 3 + k] + (double)sr[sc * 3 + k]);
    }
    o[6] = (float)ang;
}

__global__ void decode_boxes_kernel(DecodeArgs d, int B) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b < B) decode_one(d, b);
}

hipError_t launch_decode_boxes(float* box_pred, int B, const float* center_add, int64_t center_add_stride,
                               int center_inplace, const float* boxes_center_add, int64_t boxes_center_add_stride,
                               const float* yaw_base, int64_t yaw_stride, float* heading_residuals,
                               float* size_residuals, float* center, float* boxes7, hipStream_t s) {
    const DecodeArgs d{box_pred, center_add, center_add_stride, center_inplace, boxes_center_add, boxes_center_add_stride,
                       yaw_base, yaw_stride, heading_residuals, size_residuals, center, boxes7};
    hipLaunchKernelGGL(decode_boxes_kernel, dim3((B + 127) / 128), dim3(128), 0, s, d, B);
    return hipGetLastError();
}

// The last FC layer of a box estimator (c_in -> 39, no ReLU) and the decode of its output in ONE launch (round 3: one
// launch boundary less per call, 7 -> 6 kernels behind the point head of a static call). One workgroup per tile of 32
// items computes BOTH output tiles (rows 0..31, 32..38) with fc_kernel's own split of the contraction — wave w takes
// the w-th quarter of K, the partial sums meet in LDS in the same fixed order — so box_pred has the bits fc_kernel
// gives it; then the tile's first 32 threads decode their items from the rows just written.
__global__ __launch_bounds__(256) void fc39_decode_kernel(const float* __restrict__ W, const float* __restrict__ bias,
                                                          const float* __restrict__ x, int64_t xs, int B, int c_in,
                                                          DecodeArgs d) {
    __shared__ float red[3][2][16][64];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int h = lane >> 5;
    const int bt = blockIdx.x;
    const int item = 32 * bt + (lane & 31);
    const bool item_ok = item < B;
    const int row1 = 32 + (lane & 31);
    const float* wp0 = W + (int64_t)(lane & 31) * c_in + 4 * h;
    const float* wp1 = W + (int64_t)(row1 < 39 ? row1 : 0) * c_in + 4 * h;     // (rows past 38: row 0, never stored)
    const float* xp = x + (int64_t)(item_ok ? item : 0) * xs + 4 * h;
    f32x16 acc0, acc1;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int ch = tile_chan(r, h);
        acc0[r] = wave == 0 ? bias[ch] : 0.0f;
        acc1[r] = (wave == 0 && 32 + ch < 39) ? bias[32 + ch] : 0.0f;
    }
    const int chunk = ((c_in + 31) / 32) * 8;               // fc_kernel's per-wave share of K
    const int k_end = min(c_in, (wave + 1) * chunk);
    for (int k0 = wave * chunk; k0 < k_end; k0 += 8) {
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(wp0 + k0);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(wp1 + k0);
        const f32x4 bv = *reinterpret_cast<const f32x4*>(xp + k0);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc0 = mfma32(a0[e], bv[e], acc0);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc1 = mfma32(a1[e], bv[e], acc1);
    }
    if (wave > 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            red[wave - 1][0][r][lane] = acc0[r];
            red[wave - 1][1][r][lane] = acc1[r];
        }
    }
    __syncthreads();
    if (wave == 0 && item_ok) {
        float* yp = d.box_pred + (int64_t)item * 39;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ch = tile_chan(r, h);
            yp[ch] = ((acc0[r] + red[0][0][r][lane]) + red[1][0][r][lane]) + red[2][0][r][lane];
            if (32 + ch < 39) yp[32 + ch] = ((acc1[r] + red[0][1][r][lane]) + red[1][1][r][lane]) + red[2][1][r][lane];
        }
    }
    __threadfence_block();
    __syncthreads();                                        // the tile's 32 rows of box_pred are visible to the workgroup
    if (threadIdx.x < 32 && 32 * bt + (int)threadIdx.x < B) decode_one(d, 32 * bt + (int)threadIdx.x);
}

hipError_t launch_fc39_decode(const float* W, const float* bias, const float* x, int64_t xs, int B, int c_in,
                              const DecodeArgs& d, hipStream_t s) {
    if (c_in % 8 != 0 || xs % 4 != 0) return hipErrorInvalidValue;
    hipLaunchKernelGGL(fc39_decode_kernel, dim3((unsigned)((B + 31) / 32)), dim3(256), 0, s, W, bias, x, xs, B, c_in, d);
    return hipGetLastError();
}

// ================================================================================== two-stage re-centring
__device__ __forceinline__ float torch_remainder(float a, float m) {   // aten remainder for floats
    float r = fmodf(a, m);
    if (r != 0.0f && ((r < 0.0f) != (m < 0.0f))) r += m;
    return r;
}

__global__ void recenter_kernel(const float* __restrict__ obj, int B, int M, const float* __restrict__ init_box,
                                const float* __restrict__ box_one, const float* __restrict__ bbox_gt,
                                float* __restrict__ obj2, int64_t* __restrict__ hcl, float* __restrict__ hrl) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * M) return;
    const int b = (int)(i / M);
    const float* ib = init_box + b * 7;
    const float* bo = box_one + b * 7;
    const float c0 = cosf(ib[6]), s0 = sinf(ib[6]);
    const float c1 = cosf(-bo[6]), s1 = sinf(-bo[6]);
    const float x = obj[i * 3 + 0], y = obj[i * 3 + 1], z = obj[i * 3 + 2];
    float px = c0 * x - s0 * y, py = s0 * x + c0 * y, pz = z;
    px = px + ib[0];
    py = py + ib[1];
    pz = pz + ib[2];
    px = px - bo[0];
    py = py - bo[1];
    pz = pz - bo[2];
    obj2[i * 3 + 0] = c1 * px - s1 * py;
    obj2[i * 3 + 1] = s1 * px + c1 * py;
    obj2[i * 3 + 2] = pz;
    if (bbox_gt && hcl && (i % M) == 0) {
        // angle2class (utils.py:53-60) evaluated on fp32 tensors as the reference does
        const float two_pi = (float)(2.0 * 3.14159265358979323846);
        const double per_d = 2.0 * 3.14159265358979323846 / 12.0;
        float a = bbox_gt[b * 7 + 6] - bo[6];
        a = torch_remainder(a, two_pi);
        const float shifted = torch_remainder(a + (float)(per_d / 2.0), two_pi);
        const int cid = (int)(shifted / (float)per_d);
        hcl[b] = cid;
        hrl[b] = shifted - (float)((double)cid * per_d + per_d / 2.0);
    }
}

hipError_t launch_recenter(const float* obj_pts, int B, int M, const float* init_box7, const float* box_one7,
                           const float* bbox_gt7, float* obj_pts_two, int64_t* hcl, float* hrl, hipStream_t s) {
    const int64_t total = (int64_t)B * M;
    hipLaunchKernelGGL(recenter_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, obj_pts, B, M,
                       init_box7, box_one7, bbox_gt7, obj_pts_two, hcl, hrl);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------- parse_output_to_tensors
// tools/static_model.py:64-92 (dynamic_model.py: the same slices): the (B, 39) output of a box estimator cut into centre (3),
// heading scores (12), normalised heading residuals (12), size scores (3), normalised size residuals (3 x 3), plus the two
// scaled copies heading_residuals = hrn * (pi / 12) and size_residuals = srn * MEAN_SIZE — seven tensors from one launch
// (train mode, where they have to be tensors of their own: five strided copies and two multiplications before), and their
// gradients put back side by side by one launch (a concatenation, two multiplications and two additions before).
__global__ void parse_box_pred_kernel(const float* __restrict__ bp, int64_t ldb, int B, float* __restrict__ c, float* __restrict__ hs,
                                      float* __restrict__ hrn, float* __restrict__ hr, float* __restrict__ ss,
                                      float* __restrict__ srn, float* __restrict__ sr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * 39) return;
    const int b = i / 39, k = i % 39;
    const float v = bp[(int64_t)b * ldb + k];
    if (k < 3) {
        c[b * 3 + k] = v;
    } else if (k < 15) {
        hs[b * 12 + k - 3] = v;
    } else if (k < 27) {
        hrn[b * 12 + k - 15] = v;
        hr[b * 12 + k - 15] = v * (float)(3.14159265358979323846 / 12.0);
    } else if (k < 30) {
        ss[b * 3 + k - 27] = v;
    } else {
        srn[b * 9 + k - 30] = v;
        sr[b * 9 + k - 30] = v * c_mean_size[k - 30];
    }
}
// any of the seven gradients may be NULL (= zeros)
__global__ void parse_box_pred_bwd_kernel(const float* __restrict__ gc, const float* __restrict__ ghs, const float* __restrict__ ghrn,
                                          const float* __restrict__ ghr, const float* __restrict__ gss,
                                          const float* __restrict__ gsrn, const float* __restrict__ gsr, int B,
                                          float* __restrict__ g) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * 39) return;
    const int b = i / 39, k = i % 39;
    float v = 0.0f;
    if (k < 3) {
        v = gc ? gc[b * 3 + k] : 0.0f;
    } else if (k < 15) {
        v = ghs ? ghs[b * 12 + k - 3] : 0.0f;
    } else if (k < 27) {
        const float t = ghr ? __fmul_rn(ghr[b * 12 + k - 15], (float)(3.14159265358979323846 / 12.0)) : 0.0f;
        v = ghrn ? (ghr ? __fadd_rn(ghrn[b * 12 + k - 15], t) : ghrn[b * 12 + k - 15]) : t;
    } else if (k < 30) {
        v = gss ? gss[b * 3 + k - 27] : 0.0f;
    } else {
        const float t = gsr ? __fmul_rn(gsr[b * 9 + k - 30], c_mean_size[k - 30]) : 0.0f;
        v = gsrn ? (gsr ? __fadd_rn(gsrn[b * 9 + k - 30], t) : gsrn[b * 9 + k - 30]) : t;
    }
    g[i] = v;
}
hipError_t launch_parse_box_pred(const float* bp, int64_t ldb, int B, float* c, float* hs, float* hrn, float* hr, float* ss, float* srn,
                                 float* sr, hipStream_t s) {
    hipLaunchKernelGGL(parse_box_pred_kernel, dim3((unsigned)((B * 39 + 255) / 256)), dim3(256), 0, s, bp, ldb, B, c, hs, hrn, hr, ss, srn, sr);
    return hipGetLastError();
}
hipError_t launch_parse_box_pred_backward(const float* gc, const float* ghs, const float* ghrn, const float* ghr, const float* gss,
                                          const float* gsrn, const float* gsr, int B, float* g, hipStream_t s) {
    hipLaunchKernelGGL(parse_box_pred_bwd_kernel, dim3((unsigned)((B * 39 + 255) / 256)), dim3(256), 0, s, gc, ghs, ghrn, ghr, gss, gsrn, gsr,
                       B, g);
    return hipGetLastError();
}
