// dal3_latency.hip — the three shared-MLP kernels for SMALL jobs (a handful of crops): one workgroup of 16 waves
// per 32-point tile, the layers' output channels dealt out over the waves, activations exchanged through LDS.
//
// The throughput kernels (dal3_pointmlp.hip) carry a tile through a whole network inside ONE wave: no LDS, no
// barrier, perfect for thousands of tiles — but a tile's latency is the serial chain of all its MFMAs (encode 72 us,
// decode 102 us, point head 84 us at 2.4 GHz) however empty the chip is. The reference's online use (one track
// at a time, a few crops) sits exactly there: B = 1..16 crops all took ~315 us. Here a tile's layer is split by
// OUTPUT TILE over the 16 waves of a workgroup (4 per SIMD, so the CU's four matrix pipes work on one tile), the
// 32 x C activations live in LDS point-major ([point][channel], rows padded by 4 floats: a lane's four consecutive
// channels are one ds_read_b128 / ds_write_b128, conflict-free), one barrier per layer.
//
// Same packed weights (every wave streams the fragments of ITS output tiles through the 8-deep ring), same folded
// biases, and — per output element — the same k-ordered FMA chain as the throughput kernels: the results are
// bit-identical (tests/test_gpu_latency.py), so the dispatch between the two families is invisible to the caller.
#include "dal3_device.h"
#include "dal3_kernels.h"

#define LAT_WAVES 16
#define LAT_LDA 516                       // bufA: up to 512 channels + 4 floats of padding per point row
#define LAT_LDB 260                       // bufB: up to 256 channels
#define LAT_BUF_FLOATS (32 * LAT_LDA + 32 * LAT_LDB)

namespace {

// this lane's four consecutive channels [c0 + 4h, +4) of its point, as the B operand registers 4q..4q+3 expect them
__device__ __forceinline__ f32x4 lat_x(const float* __restrict__ X, int ld, int m, int h, int c0) {
    return *reinterpret_cast<const f32x4*>(X + m * ld + c0 + 4 * h);
}

// store an accumulator tile (channels on rows) point-major: registers 4g..4g+3 are channels 8g+4h..+3 of point m
__device__ __forceinline__ void lat_store(float* __restrict__ Y, int ld, int m, int h, int mt, const f32x16& acc) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        f32x4 o = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
        *reinterpret_cast<f32x4*>(Y + m * ld + 32 * mt + 8 * g + 4 * h) = o;
    }
}

// acc += W'(32 x 32 KT) . X  (SWAP: X^T . W'^T) with the out-tile's KT*4 fragments read through the ring and X from
// LDS. KT is even: eight fragments (two k-tiles) per unrolled round keep the ring's slots at constant indices.
template <bool SWAP>
__device__ __forceinline__ void lat_block(WRing<8>& ring, int KT, const float* __restrict__ X, int ld, int m, int h,
                                          f32x16& acc) {
    for (int kt = 0; kt < KT; kt += 2) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const f32x4 a = ring.slot[i];
            ring.slot[i] = ring.fetch();
            const f32x4 x = lat_x(X, ld, m, h, 32 * (kt + i / 4) + 8 * (i % 4));
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = SWAP ? mfma32(x[e], a[e], acc) : mfma32(a[e], x[e], acc);
        }
    }
}

// one layer: Y[:, 32 mt ..] = relu(W' X + init) for the out-tiles mt = wave, wave + 16, ...; `frags` = the layer's
// fragments [MT][KT][4][64] f32x4; init (LDS) = the folded bias, or the crop's dconv1 term
__device__ __forceinline__ void lat_layer(const f32x4* __restrict__ frags, int KT, int MT, const float* __restrict__ init,
                                          const float* __restrict__ X, int ldx, float* __restrict__ Y, int ldy, int lane,
                                          int wave) {
    const int m = lane & 31, h = lane >> 5;
    for (int mt = wave; mt < MT; mt += LAT_WAVES) {
        WRing<8> ring;
        ring.init(frags + (size_t)mt * KT * 4 * 64, lane);
        f32x16 acc = tile_from_channels(init + 32 * mt, h);
        lat_block<false>(ring, KT, X, ldx, m, h, acc);
        lat_store(Y, ldy, m, h, mt, relu16(acc));
    }
}

// the max-pooled last layer, computed transposed; dst: the item's 32 n_tiles channel maxima (global, zero-filled)
__device__ __forceinline__ void lat_max_layer(const f32x4* __restrict__ frags, int KT, int n_tiles,
                                              const float* __restrict__ s_bias, const float* __restrict__ X, int ldx,
                                              float* __restrict__ dst, int lane, int wave) {
    const int m = lane & 31, h = lane >> 5;
    for (int mt = wave; mt < n_tiles; mt += LAT_WAVES) {
        WRing<8> ring;
        ring.init(frags + (size_t)mt * KT * 4 * 64, lane);
        f32x16 acc[1] = {f32x16{}};
        lat_block<true>(ring, KT, X, ldx, m, h, acc[0]);
        MaxEpilogueT<1> ep;
        ep.all(acc, s_bias + 32 * mt, dst + 32 * mt, lane);
    }
}

// first layer (raw coordinates, K = 2 KS): out-tiles dealt over the waves like every other layer
template <int KS>
__device__ __forceinline__ void lat_first(const float* __restrict__ w1, const float* __restrict__ s_b1, int MT, const BCN& x,
                                          int64_t b, int n0, int n_pts, int c_in, float* __restrict__ Y, int ldy, int lane,
                                          int wave) {
    const int m = lane & 31, h = lane >> 5;
    if (wave >= MT) return;
    float in[1][KS];
    load_points<KS, 1>(x, b, n0, n_pts, c_in, in, lane);
    for (int mt = wave; mt < MT; mt += LAT_WAVES) {
        f32x16 acc = tile_from_channels(s_b1 + 32 * mt, h);
#pragma unroll
        for (int s = 0; s < KS; ++s) acc = mfma32(w1[(mt * KS + s) * 64 + lane], in[0][s], acc);
        lat_store(Y, ldy, m, h, mt, relu16(acc));
    }
}

__device__ __forceinline__ void lat_copy(float* __restrict__ dst, const float* __restrict__ src, int n) {
    for (int i = threadIdx.x; i < n; i += 64 * LAT_WAVES) dst[i] = src[i];
}

}   // namespace

// ------------------------------------------------------------------------------------------------ encode
__global__ __launch_bounds__(64 * LAT_WAVES) void ins_seg_encode_lat_kernel(InsSegW w, BCN pts, int c_in, int n_pts,
                                                                          int tiles_per_item, float* __restrict__ g) {
    extern __shared__ __attribute__((aligned(16))) float lat_smem[];
    float* bufA = lat_smem;
    float* bufB = bufA + 32 * LAT_LDA;
    float* s_b = bufB + 32 * LAT_LDB;                   // b1 64 | b2 64 | b3 64 | b4 128 | b5 1024
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t b = blockIdx.x / tiles_per_item;
    const int n0 = (blockIdx.x % tiles_per_item) * 32;
    lat_copy(s_b, w.b1, 64);
    lat_copy(s_b + 64, w.b2, 64);
    lat_copy(s_b + 128, w.b3, 64);
    lat_copy(s_b + 192, w.b4, 128);
    lat_copy(s_b + 320, w.b5, 1024);
    __syncthreads();
    lat_first<2>(w.w1, s_b, 2, pts, b, n0, n_pts, c_in, bufA, LAT_LDA, lane, wave);
    __syncthreads();
    lat_layer(w.enc_stream + ENC_W2 * 64, 2, 2, s_b + 64, bufA, LAT_LDA, bufB, LAT_LDB, lane, wave);
    __syncthreads();
    lat_layer(w.enc_stream + ENC_W3 * 64, 2, 2, s_b + 128, bufB, LAT_LDB, bufA, LAT_LDA, lane, wave);
    __syncthreads();
    lat_layer(w.enc_stream + ENC_W4 * 64, 2, 4, s_b + 192, bufA, LAT_LDA, bufB, LAT_LDB, lane, wave);
    __syncthreads();
    lat_max_layer(w.enc_stream + ENC_W5 * 64, 4, 32, s_b + 320, bufB, LAT_LDB, g + b * 1024, lane, wave);
}

// ------------------------------------------------------------------------------------------------ decode
__global__ __launch_bounds__(64 * LAT_WAVES) void ins_seg_decode_lat_kernel(InsSegW w, BCN pts, int c_in, int n_pts,
                                                                          int tiles_per_item,
                                                                          const float* __restrict__ gbias,
                                                                          float* __restrict__ logits,
                                                                          uint8_t* __restrict__ mask) {
    extern __shared__ __attribute__((aligned(16))) float lat_smem[];
    float* bufA = lat_smem;
    float* bufB = bufA + 32 * LAT_LDA;
    float* s_b = bufB + 32 * LAT_LDB;                   // b1 64 | b2 64 | gb 512 | db2 256 | db3 128 | db4 128 | dw5 256 | db5 2
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int m = lane & 31, h = lane >> 5;
    const int64_t b = blockIdx.x / tiles_per_item;
    const int n0 = (blockIdx.x % tiles_per_item) * 32;
    lat_copy(s_b, w.b1, 64);
    lat_copy(s_b + 64, w.b2, 64);
    lat_copy(s_b + 128, gbias + b * 512, 512);
    lat_copy(s_b + 640, w.db2, 256);
    lat_copy(s_b + 896, w.db3, 128);
    lat_copy(s_b + 1024, w.db4, 128);
    lat_copy(s_b + 1152, w.dw5, 256);
    lat_copy(s_b + 1408, w.db5, 2);
    __syncthreads();
    lat_first<2>(w.w1, s_b, 2, pts, b, n0, n_pts, c_in, bufA, LAT_LDA, lane, wave);
    __syncthreads();
    lat_layer(w.dec_stream + DEC_W2 * 64, 2, 2, s_b + 64, bufA, LAT_LDA, bufB, LAT_LDB, lane, wave);       // conv2 -> out2
    __syncthreads();
    lat_layer(w.lat_stream, 2, 16, s_b + 128, bufB, LAT_LDB, bufA, LAT_LDA, lane, wave);                   // dconv1 (per-point part, init = crop term)
    __syncthreads();
    lat_layer(w.lat_stream + 128 * 64, 16, 8, s_b + 640, bufA, LAT_LDA, bufB, LAT_LDB, lane, wave);        // dconv2
    __syncthreads();
    lat_layer(w.dec_stream + DEC_W3 * 64, 8, 4, s_b + 896, bufB, LAT_LDB, bufA, LAT_LDA, lane, wave);      // dconv3
    __syncthreads();
    lat_layer(w.dec_stream + DEC_W4 * 64, 4, 4, s_b + 1024, bufA, LAT_LDA, bufB, LAT_LDB, lane, wave);     // dconv4
    __syncthreads();
    if (wave == 0) {                                     // dconv5 + mask: the throughput kernel's exact summation order
        const float* s_dw5 = s_b + 1152;
        float l0 = 0.0f, l1 = 0.0f;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 wa = *reinterpret_cast<const f32x4*>(s_dw5 + 32 * kt + 8 * q + 4 * h);
                const f32x4 wb = *reinterpret_cast<const f32x4*>(s_dw5 + 128 + 32 * kt + 8 * q + 4 * h);
                const f32x4 y = lat_x(bufB, LAT_LDB, m, h, 32 * kt + 8 * q);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    l0 = fmaf(wa[e], y[e], l0);
                    l1 = fmaf(wb[e], y[e], l1);
                }
            }
        }
        float s0 = l0 + __shfl_xor(l0, 32) + s_b[1408];
        float s1 = l1 + __shfl_xor(l1, 32) + s_b[1409];
        const bool crop_bad = bits_nonfinite(s_b[128]);     // the crop's dconv1 term is NaN: a non-finite coordinate (dal3_device.h)
        if (crop_bad) s0 = s1 = __int_as_float(DAL3_QNAN_BITS);
        const int n = n0 + m;
        if (h == 0 && n < n_pts) {
            f32x2 o;
            o[0] = s0;
            o[1] = s1;
            *reinterpret_cast<f32x2*>(logits + (b * n_pts + n) * 2) = o;
            mask[b * n_pts + n] = (!crop_bad && s0 < s1) ? 1 : 0;
        }
    }
}

// ------------------------------------------------------------------------------------------------ point heads
template <int KS, int C1, int C2, int C3>
__global__ __launch_bounds__(64 * LAT_WAVES) void point_head_lat_kernel(PointHeadW w, BCN x, int c_in, int n_pts,
                                                                      int tiles_per_item, float* __restrict__ feat,
                                                                      const int32_t* __restrict__ distinct) {
    extern __shared__ __attribute__((aligned(16))) float lat_smem[];
    float* bufA = lat_smem;
    float* bufB = bufA + 32 * LAT_LDA;
    float* s_b = bufB + 32 * LAT_LDB;                   // b1 C1 | b2 C2 | b3 C3 | b4 512
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t b = blockIdx.x / tiles_per_item;
    const int n0 = (blockIdx.x % tiles_per_item) * 32;
    if (distinct) {                                     // copies beyond the first distinct[b] points: see point_head_kernel
        const int d = distinct[b];
        n_pts = d <= 0 ? 1 : (d < n_pts ? d : n_pts);
    }
    if (n0 >= n_pts) return;                            // (uniform for the workgroup)
    lat_copy(s_b, w.b1, C1);
    lat_copy(s_b + C1, w.b2, C2);
    lat_copy(s_b + C1 + C2, w.b3, C3);
    lat_copy(s_b + C1 + C2 + C3, w.b4, 512);
    __syncthreads();
    lat_first<KS>(w.w1, s_b, C1 / 32, x, b, n0, n_pts, c_in, bufA, LAT_LDA, lane, wave);
    __syncthreads();
    const f32x4* f2 = w.stream;
    const f32x4* f3 = f2 + (size_t)(C2 / 32) * (C1 / 32) * 4 * 64;
    const f32x4* f4 = f3 + (size_t)(C3 / 32) * (C2 / 32) * 4 * 64;
    lat_layer(f2, C1 / 32, C2 / 32, s_b + C1, bufA, LAT_LDA, bufB, LAT_LDB, lane, wave);
    __syncthreads();
    lat_layer(f3, C2 / 32, C3 / 32, s_b + C1 + C2, bufB, LAT_LDB, bufA, LAT_LDA, lane, wave);
    __syncthreads();
    lat_max_layer(f4, C3 / 32, 16, s_b + C1 + C2 + C3, bufA, LAT_LDA, feat + b * 512, lane, wave);
}

// ------------------------------------------------------------------------------------------------ launchers
template <class K>
static hipError_t lat_launch_prep(K kernel, size_t lds) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
}

hipError_t launch_ins_seg_encode_lat(const InsSegW& w, BCN pts, int c_in, int B, int N, float* g, hipStream_t s) {
    const size_t lds = (LAT_BUF_FLOATS + 1344) * sizeof(float);
    hipError_t e = lat_launch_prep(ins_seg_encode_lat_kernel, lds);
    if (e != hipSuccess) return e;
    const int tpi = (N + 31) / 32;
    hipLaunchKernelGGL(ins_seg_encode_lat_kernel, dim3((unsigned)((int64_t)B * tpi)), dim3(64 * LAT_WAVES), lds, s, w, pts, c_in, N,
                       tpi, g);
    return hipGetLastError();
}

hipError_t launch_ins_seg_decode_lat(const InsSegW& w, BCN pts, int c_in, int B, int N, const float* gbias, float* logits,
                                     uint8_t* mask, hipStream_t s) {
    const size_t lds = (LAT_BUF_FLOATS + 1412) * sizeof(float);
    hipError_t e = lat_launch_prep(ins_seg_decode_lat_kernel, lds);
    if (e != hipSuccess) return e;
    const int tpi = (N + 31) / 32;
    hipLaunchKernelGGL(ins_seg_decode_lat_kernel, dim3((unsigned)((int64_t)B * tpi)), dim3(64 * LAT_WAVES), lds, s, w, pts, c_in, N,
                       tpi, gbias, logits, mask);
    return hipGetLastError();
}

template <int KS, int C1, int C2, int C3>
static hipError_t head_lat(const PointHeadW& w, BCN x, int c_in, int B, int M, float* feat, const int32_t* distinct,
                           hipStream_t s) {
    static_assert(C1 <= 512 && C2 <= 256 && C3 <= 512, "activation widths must fit the LDS buffers (A: 512, B: 256)");
    const size_t lds = (LAT_BUF_FLOATS + C1 + C2 + C3 + 512) * sizeof(float);
    auto k = point_head_lat_kernel<KS, C1, C2, C3>;
    hipError_t e = lat_launch_prep(k, lds);
    if (e != hipSuccess) return e;
    const int tpi = (M + 31) / 32;
    hipLaunchKernelGGL(k, dim3((unsigned)((int64_t)B * tpi)), dim3(64 * LAT_WAVES), lds, s, w, x, c_in, M, tpi, feat, distinct);
    return hipGetLastError();
}

hipError_t launch_point_head_lat(int head_kind, const PointHeadW& w, BCN x, int c_in, int B, int M, float* feat,
                                 const int32_t* distinct, hipStream_t s) {
    switch (head_kind) {
        case 1: return head_lat<2, 128, 128, 256>(w, x, c_in, B, M, feat, distinct, s);
        case 2: return head_lat<2, 64, 128, 256>(w, x, c_in, B, M, feat, distinct, s);
        case 3: return head_lat<4, 64, 64, 128>(w, x, c_in, B, M, feat, distinct, s);
        default: return hipErrorInvalidValue;
    }
}
