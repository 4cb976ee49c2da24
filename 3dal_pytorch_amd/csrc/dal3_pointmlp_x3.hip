// dal3_pointmlp_x3.hip — the shared-MLP kernels on fp16 MFMAs at fp32 ACCURACY (round 3): "f16x3".
//
// gfx950 has no TF32-like mode: the exact-f32 MFMA runs at 1/16 of the fp16 rate (157 vs 2,500 TFLOP/s dense), and
// rounding operands to 16 bits once costs three digits (logits 3e-3 / 3e-2 relative for fp16 / bf16, tests/
// f16x3_accuracy.py). Splitting BOTH operands of every product into two fp16 halves,
//     x = x_hi + x_lo,  w = w_hi + w_lo      (hi = fp16(v), lo = fp16(v - hi): 22 bits of significand together)
//     w x ~= w_hi x_hi + w_hi x_lo + w_lo x_hi           (the dropped w_lo x_lo is 2^-22 relative)
// gives the fp32 kernels' accuracy (logits 1.0e-6 vs 2.0e-6 against float64, same script) from THREE fp16 MFMAs with
// fp32 accumulation — 16/3 of the fp32 MFMA rate on paper. Everything else is the 16-bit family's design (dal3_lp.h):
// channels on MFMA rows, points on columns, the fp32 accumulator of layer k split in registers into the B operands of
// layer k+1 (activations never leave registers), weights as one stream of 1-KiB fragments — a (hi, lo) PAIR per k-step —
// shared by the four waves of a workgroup through the three-slot LDS ring filled by LDS-DMA, persistent workgroups.
// ONE generic block (x3_block) walks the fragment stream with a cursor that is a compile-time constant everywhere; the
// next k-step's pair is always in registers, a segment is opened a k-step before it is needed, the refill's LDS-DMA
// instructions go out one per k-step, and the VALU work between layers (bias, ReLU, the (hi, lo) split; the max
// epilogues) is dealt out under the NEXT block's MFMAs (profiles/LEDGER_r01_r03.md 5.4).
//
// The first layer (raw coordinates, K = 3/4/8) stays on the fp32 MFMA, biases and the per-crop dconv1 term are fp32,
// logits / mask / pooled features and all I/O are fp32, as in the other two families.
#include <stdlib.h>

#include "dal3_kernels.h"
#include "dal3_lp.h"
#include "dal3_x3.h"

// This file is compiled twice (Makefile), like dal3_pointmlp_lp.hip: X3_PART=1 -> point heads + encode with `-mllvm
// -amdgpu-mfma-vgpr-form` (their accumulators fit the VGPR file: the splits and max epilogues read them in place instead
// of through one v_accvgpr_read per value), X3_PART=2 -> decode without it (its resident dconv2 accumulators ARE the
// AccVGPR file).
#ifndef X3_PART
#define X3_PART 3
#endif
static int x3_cu_count() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
            v = 256;
        n = v;
    }
    return n;
}

#if X3_PART & 1
// ------------------------------------------------------------------------------------------------ point heads
// LDS: b2 C2 | b3 C3 | b4 512 | s_max 512 | b1 C1 | w1 (C1/32) KS 64 floats, then the ring (3 x 32 KiB)
__host__ __device__ constexpr int x3_head_small_bytes(int c1, int c2, int c3, int ks) {
    return ((c2 + c3 + 512 + 512 + c1 + (c1 / 32) * ks * 64) * 4 + 1023) / 1024 * 1024;
}
__host__ __device__ constexpr int x3_head_frags(int c1, int c2, int c3) {      // per group, before padding to segments
    return (c2 / 32) * (c1 / 32) * 4 + (c3 / 32) * (c2 / 32) * 4 + 16 * (c3 / 32) * 4;
}

template <int KS, int C1, int C2, int C3, int T>
__global__ __launch_bounds__(256) void point_head_x3_kernel(PointHeadX3W w, BCN x, int c_in, int n_pts_all, int n_items,
                                                            int n_groups, float* __restrict__ feat,
                                                            const int32_t* __restrict__ distinct) {
    constexpr int K2 = C1 / 32, M2 = C2 / 32, K3 = C2 / 32, M3 = C3 / 32, K4 = C3 / 32;
    constexpr int NB = C2 + C3 + 512, NW1 = K2 * KS * 64;
    constexpr int N_SEGS = (x3_head_frags(C1, C2, C3) + X3_SEG - 1) / X3_SEG;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_bias = reinterpret_cast<float*>(smem);
    int* s_max = reinterpret_cast<int*>(s_bias + NB);
    float* s_b1 = s_bias + NB + 512;
    float* s_w1 = s_b1 + C1;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5;
    for (int i = threadIdx.x; i < NB; i += 256) s_bias[i] = w.bias[i];
    for (int i = threadIdx.x; i < 512; i += 256) s_max[i] = 0;
    for (int i = threadIdx.x; i < C1; i += 256) s_b1[i] = w.b1[i];
    for (int i = threadIdx.x; i < NW1; i += 256) s_w1[i] = w.w1[i];
    __syncthreads();
    X3Stream st;
    st.init(w.stream, smem + x3_head_small_bytes(C1, C2, C3, KS), N_SEGS, wave, lane);

    for (int id = blockIdx.x; id < n_groups; id += gridDim.x) {
        const int64_t b = id % n_items;
        const int wg_tile = id / n_items;
        int np = n_pts_all;
        if (distinct) {                                    // copies beyond the first distinct[b] points: see point_head_kernel
            const int d = distinct[b];
            np = d <= 0 ? 1 : (d < np ? d : np);
        }
        if (wg_tile * X3_WAVES * 32 * T >= np) continue;   // only copies in this group: uniform skip (the stream stays put)
        st.pin(st.START, 0);
        float in[T][KS];
        load_points<KS, T>(x, b, (wg_tile * X3_WAVES + wave) * (32 * T), np, c_in, in, lane);
        X3Tile x1[T][K2], x2[T][M2], x3[T][M3];
#pragma unroll
        for (int mt = 0; mt < K2; ++mt) {                  // first layer in fp32 (operands from LDS)
            const f32x16 bv = tile_from_channels(s_b1 + 32 * mt, h);
#pragma unroll
            for (int j = 0; j < T; ++j) {
                f32x16 acc = bv;
#pragma unroll
                for (int k = 0; k < KS; ++k) acc = mfma32(s_w1[(mt * KS + k) * 64 + lane], in[j][k], acc);
                x1[j][mt] = x3_split_relu(acc);
            }
        }
        X3Carry<T> carry;
        x3_layer<K2, M2, T, 0>(st, s_bias, x1, x2, h, carry, X3NoSide());
        x3_layer<K3, M3, T, x3_vpg(x3_carry_per(K3, T), T)>(st, s_bias + C2, x2, x3, h, carry,
                                                           [&](int s) { x3_carry_steps<K3, T>(carry, x2, s); });
        x3_max_layer<K4, T>(st, s_bias + C2 + C3, x3, s_max, 16, lane, carry);
        st.end_group();
        __syncthreads();
        int* fi = reinterpret_cast<int*>(feat + b * 512);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int i = threadIdx.x + 256 * r;
            const int v = s_max[i];
            if (v > 0) atomicMax(fi + i, v);
            s_max[i] = 0;                                  // for the next group: its LDS atomics come after >= 1 barrier
        }
    }
}

template <int KS, int C1, int C2, int C3>
static hipError_t head_x3(const PointHeadX3W& w, BCN x, int c_in, int B, int M, float* feat, const int32_t* distinct,
                          hipStream_t s) {
    constexpr int T = DAL3_X3_HEAD_T;
    const size_t lds = 3 * X3_SEG * 1024 + x3_head_small_bytes(C1, C2, C3, KS);
    auto k = point_head_x3_kernel<KS, C1, C2, C3, T>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const int tiles = (M + 32 * X3_WAVES * T - 1) / (32 * X3_WAVES * T);
    const int64_t n_groups = (int64_t)B * tiles;
    if (n_groups > 0x7fffffff) return hipErrorInvalidValue;
    const int64_t grid = n_groups < x3_cu_count() ? n_groups : x3_cu_count();
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(256), lds, s, w, x, c_in, M, B, (int)n_groups, feat, distinct);
    return hipGetLastError();
}

int point_head_x3_segments(int head_kind) {
    int c_in, ks, c[4], n_fc, fi[3], fo[3];
    point_head_dims(head_kind, &c_in, &ks, c, &n_fc, fi, fo);
    return (x3_head_frags(c[0], c[1], c[2]) + X3_SEG - 1) / X3_SEG;
}

hipError_t launch_point_head_x3(int head_kind, const PointHeadX3W& w, BCN x, int c_in, int B, int M, float* feat,
                                const int32_t* distinct, hipStream_t s) {
    hipError_t e0 = launch_nonfinite_rows(x, B, M, c_in, feat, 512, s);     // feat = 0 (NaN rows: include/dal3.h)
    if (e0 != hipSuccess) return e0;
    switch (head_kind) {
        case DAL3_HEAD_STATIC_BOX_EST: return head_x3<2, 128, 128, 256>(w, x, c_in, B, M, feat, distinct, s);
        case DAL3_HEAD_POINT_EMB: return head_x3<2, 64, 128, 256>(w, x, c_in, B, M, feat, distinct, s);
        case DAL3_HEAD_BOX_EMB: return head_x3<4, 64, 64, 128>(w, x, c_in, B, M, feat, distinct, s);
        default: return hipErrorInvalidValue;
    }
}

// ------------------------------------------------------------------------------------------------ encode
// conv1 (fp32) .. conv5 + max over the points -> g (B,1024). Stream per group: conv2 16 | conv3 16 | conv4 32 | conv5 512
// fragments = 18 segments. LDS small arrays: b2 64 | b3 64 | b4 128 | b5 1024 | s_max 1024 | b1 64 | w1 256 floats.
#define X3_ENC_SMALL_BYTES 11264
#define X3_ENC_SEGS 18
template <int T>
__global__ __launch_bounds__(256) void ins_seg_encode_x3_kernel(InsSegX3W w, BCN pts, int c_in, int n_pts, int tiles_per_item,
                                                                int n_groups, float* __restrict__ g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_bias = reinterpret_cast<float*>(smem);
    int* s_max = reinterpret_cast<int*>(s_bias + 1280);
    float* s_b1 = s_bias + 1280 + 1024;
    float* s_w1 = s_b1 + 64;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5;
    for (int i = threadIdx.x; i < 1280; i += 256) s_bias[i] = w.bias_enc[i];
    for (int i = threadIdx.x; i < 1024; i += 256) s_max[i] = 0;
    if (threadIdx.x < 64) s_b1[threadIdx.x] = w.b1[threadIdx.x];
    s_w1[threadIdx.x] = w.w1[threadIdx.x];
    __syncthreads();
    X3Stream st;
    st.init(w.enc_stream, smem + X3_ENC_SMALL_BYTES, X3_ENC_SEGS, wave, lane);

    float in_nx[T][2];                                     // the next group's points, fetched a group ahead
    auto prefetch = [&](int gq) {
        load_points<2, T>(pts, gq / tiles_per_item, ((gq % tiles_per_item) * X3_WAVES + wave) * (32 * T), n_pts, c_in, in_nx, lane);
    };
    prefetch(blockIdx.x);
    for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        st.pin(st.START, 0);
        const int64_t b = grp / tiles_per_item;
        X3Tile x1[T][2], x2[T][2], x3[T][2], x4[T][4];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {                   // conv1 in fp32
            const f32x16 bv = tile_from_channels(s_b1 + 32 * mt, h);
#pragma unroll
            for (int j = 0; j < T; ++j) {
                f32x16 acc = bv;
#pragma unroll
                for (int k = 0; k < 2; ++k) acc = mfma32(s_w1[(mt * 2 + k) * 64 + lane], in_nx[j][k], acc);
                x1[j][mt] = x3_split_relu(acc);
            }
        }
        {
            const int nx = grp + (int)gridDim.x;
            prefetch(nx < n_groups ? nx : grp);            // (the last group re-reads itself: uniform control flow)
        }
        X3Carry<T> carry;
        constexpr int VC = x3_vpg(x3_carry_per(2, T), T);
        x3_layer<2, 2, T, 0>(st, s_bias, x1, x2, h, carry, X3NoSide());
        x3_layer<2, 2, T, VC>(st, s_bias + 64, x2, x3, h, carry, [&](int s) { x3_carry_steps<2, T>(carry, x2, s); });
        x3_layer<2, 4, T, VC>(st, s_bias + 128, x3, x4, h, carry, [&](int s) { x3_carry_steps<2, T>(carry, x3, s); });
        x3_max_layer<4, T>(st, s_bias + 256, x4, s_max, 32, lane, carry);
        st.end_group();
        __syncthreads();
        int* gi = reinterpret_cast<int*>(g + b * 1024);
        for (int i = threadIdx.x; i < 1024; i += 256) {
            const int v = s_max[i];
            if (v > 0) atomicMax(gi + i, v);
            s_max[i] = 0;
        }
    }
}

hipError_t launch_ins_seg_encode_x3(const InsSegX3W& w, BCN pts, int c_in, int B, int N, float* g, hipStream_t s) {
    constexpr int T = DAL3_X3_ENC_T;
    const size_t lds = 3 * X3_SEG * 1024 + X3_ENC_SMALL_BYTES;
    auto k = ins_seg_encode_x3_kernel<T>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const int tpi = (N + 32 * X3_WAVES * T - 1) / (32 * X3_WAVES * T);
    const int64_t n_groups = (int64_t)B * tpi;
    if (n_groups > 0x7fffffff) return hipErrorInvalidValue;
    const int64_t grid = n_groups < x3_cu_count() ? n_groups : x3_cu_count();
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(256), lds, s, w, pts, c_in, N, tpi, (int)n_groups, g);
    return hipGetLastError();
}

#endif  // X3_PART & 1
#if X3_PART & 2
// ------------------------------------------------------------------------------------------------ decode
// conv1 (fp32), conv2, then dconv1's per-point part in 16 chunks of 32 channels (A_c: 64 -> 32, bias: the crop's dconv1
// term, fp32) each split and fed straight into dconv2's eight resident accumulator tiles (D_c: 32 -> 256), dconv3,
// dconv4, and dconv5 as one more out-tile (rows 0, 1) -> logits, mask. The chunks are software-pipelined: A_{c+1} runs
// BEFORE D_c and its output is split under D_c's MFMAs, so the stream order per group is
//   conv2 16 | A_0 8 | 15 x { A_{c+1} 8, D_c 32 } | D_15 32 | dconv3 128 | dconv4 64 | dconv5 16 fragments = 27 segments.
// LDS small arrays: b2 64 | db2 256 | db3 128 | db4 128 | db5 32 | gb 512 | b1 64 | w1 256 floats.
#define X3_DEC_SMALL_BYTES 6144
#define X3_DEC_SEGS 27

// one chunk pair: A_{c+1} into t (side: aside, VA slots), then D_c from tc[P] (chunk c, split already) into a2, with
// chunk c + 1 split into tc[P ^ 1] under its 16 k-steps
template <int T, bool FIRST, int P, int VA, class ASide>
__device__ __forceinline__ void x3_dec_pair(X3Stream& st, const X3Tile (&x2)[T][2], f32x16 (&t)[T], X3Tile (&tc)[2][T][1],
                                            f32x16 (&a2)[8][T], const float* s_gb, int c, int h, ASide&& aside) {
    constexpr int PER = (8 * T + 15) / 16;
    f32x16 bv;
    x3_block<2, T, false, true, VA, true>(st, x2, t, [&](int s) {
        aside(s);
        if (s == 3) bv = tile_from_channels(s_gb + 32 * (c + 1), h);
    });
#pragma unroll
    for (int mt = 0; mt < 8; ++mt)
        x3_block<1, T, false, FIRST, x3_vpg(PER, T)>(st, tc[P], a2[mt],
                                                     [&](int s) { x3_split_steps<PER>(t, bv, tc[P ^ 1], 0, 2 * mt + s); });
}

// units [step PER, (step + 1) PER) of the split of dconv2's out-tiles TILE0 .. TILE0 + NT - 1 (accumulators a2, bias
// db2) into dconv3's input xd; bv / bvn: the current and the next tile's bias vectors (read a tile ahead)
template <int PER, int TILE0, int NT, int T>
__device__ __forceinline__ void x3_xd_steps(const f32x16 (&a2)[8][T], X3Tile (&xd)[T][8], const float* db2, int h, f32x16& bv,
                                            f32x16& bvn, int step) {
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int id = step * PER + k;
        if (id >= 0 && id < NT * 8 * T) {
            const int tile = TILE0 + id / (8 * T), u = id % (8 * T);
            if (u == 0) {
                bv = bvn;
                if (tile + 1 < 8) bvn = tile_from_channels(db2 + 32 * (tile + 1), h);
            }
            x3_split_unit<true>(a2[tile][u >> 3], bv, xd[u >> 3][tile], u & 7);
        }
    }
}

template <int T>
__global__ __launch_bounds__(256) void ins_seg_decode_x3_kernel(InsSegX3W w, BCN pts, int c_in, int n_pts, int tiles_per_item,
                                                                int n_groups, const float* __restrict__ gbias,
                                                                float* __restrict__ logits, uint8_t* __restrict__ mask) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_bias = reinterpret_cast<float*>(smem);        // b2 64 | db2 256 | db3 128 | db4 128 | db5 32 = 608
    float* s_gb = s_bias + 608;
    float* s_b1 = s_gb + 512;
    float* s_w1 = s_b1 + 64;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5;
    for (int i = threadIdx.x; i < 608; i += 256) s_bias[i] = w.bias_dec[i];
    if (threadIdx.x < 64) s_b1[threadIdx.x] = w.b1[threadIdx.x];
    s_w1[threadIdx.x] = w.w1[threadIdx.x];
    const float* s_db2 = s_bias + 64;
    const float* s_db3 = s_bias + 320;
    const float* s_db4 = s_bias + 448;
    const float* s_db5 = s_bias + 576;
    __syncthreads();
    X3Stream st;
    st.init(w.dec_stream, smem + X3_DEC_SMALL_BYTES, X3_DEC_SEGS, wave, lane);

    float in_nx[T][2], gb_nx[2];                           // the next group's points and its crop's dconv1 term
    auto prefetch = [&](int gq) {
        const int64_t bb = gq / tiles_per_item;
        load_points<2, T>(pts, bb, ((gq % tiles_per_item) * X3_WAVES + wave) * (32 * T), n_pts, c_in, in_nx, lane);
        gb_nx[0] = gbias[bb * 512 + threadIdx.x];
        gb_nx[1] = gbias[bb * 512 + 256 + threadIdx.x];
    };
    const int vblk = xcd_contiguous_block();               // (dal3_kernels.h: a crop's groups on one L2)
    prefetch(vblk);
    for (int grp = vblk; grp < n_groups; grp += gridDim.x) {
        st.pin(st.START, 0);
        const int64_t b = grp / tiles_per_item;
        const int n0 = ((grp % tiles_per_item) * X3_WAVES + wave) * (32 * T);
        __syncthreads();                                   // everyone is done with the previous group's s_gb
        s_gb[threadIdx.x] = gb_nx[0];
        s_gb[256 + threadIdx.x] = gb_nx[1];
        __syncthreads();
        // a crop with a non-finite coordinate: its dconv1 term is NaN (dal3_device.h) -> NaN logits, empty mask
        const bool crop_bad = (__builtin_amdgcn_readfirstlane(__float_as_int(s_gb[0])) & 0x7F800000) == 0x7F800000;

        X3Tile x1[T][2], x2[T][2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {                   // conv1 in fp32
            const f32x16 bv = tile_from_channels(s_b1 + 32 * mt, h);
#pragma unroll
            for (int j = 0; j < T; ++j) {
                f32x16 acc = bv;
#pragma unroll
                for (int k = 0; k < 2; ++k) acc = mfma32(s_w1[(mt * 2 + k) * 64 + lane], in_nx[j][k], acc);
                x1[j][mt] = x3_split_relu(acc);
            }
        }
        {
            const int nx = grp + (int)gridDim.x;
            prefetch(nx < n_groups ? nx : grp);
        }
        X3Carry<T> carry;
        constexpr int VC2 = x3_vpg(x3_carry_per(2, T), T), VC4 = x3_vpg(x3_carry_per(4, T), T);
        x3_layer<2, 2, T, 0>(st, s_bias, x1, x2, h, carry, X3NoSide());

        f32x16 a2[8][T];                                   // dconv2's accumulators (zero C in the first chunk)
        X3Tile tc[2][T][1];
        f32x16 t[T];
        {
            f32x16 t0[T], bv0;                             // A_0 (side: conv2's last tile), split under A_1
            x3_block<2, T, false, true, VC2, true>(st, x2, t0, [&](int s) {
                x3_carry_steps<2, T>(carry, x2, s);
                if (s == 3) bv0 = tile_from_channels(s_gb, h);
            });
            constexpr int PA = (8 * T + 3) / 4;
            x3_dec_pair<T, true, 0, x3_vpg(PA, T)>(st, x2, t, tc, a2, s_gb, 0, h,
                                                   [&](int s) { x3_split_steps<PA>(t0, bv0, tc[0], 0, s); });
        }
        x3_dec_pair<T, false, 1, 0>(st, x2, t, tc, a2, s_gb, 1, h, X3NoSide());
        x3_dec_pair<T, false, 0, 0>(st, x2, t, tc, a2, s_gb, 2, h, X3NoSide());
        {
            const int c0 = st.cur, p0 = st.pending;        // four pairs = 160 fragments = 5 segments: the state repeats
            for (int it = 0; it < 3; ++it) {
                st.pin(c0, p0);
                const int c = 3 + 4 * it;
                x3_dec_pair<T, false, 1, 0>(st, x2, t, tc, a2, s_gb, c, h, X3NoSide());
                x3_dec_pair<T, false, 0, 0>(st, x2, t, tc, a2, s_gb, c + 1, h, X3NoSide());
                x3_dec_pair<T, false, 1, 0>(st, x2, t, tc, a2, s_gb, c + 2, h, X3NoSide());
                x3_dec_pair<T, false, 0, 0>(st, x2, t, tc, a2, s_gb, c + 3, h, X3NoSide());
            }
        }
        // D_15 (chunk 15 is tc[1]); dconv2's out-tiles 0..3 are split under it, 4..7 under dconv3's first block
        X3Tile xd[T][8], y3[T][4], y4[T][4];
        f32x16 bvx, bvn;
        constexpr int PX = (4 * 8 * T + 13) / 14, VX = x3_vpg(PX, T);
#pragma unroll
        for (int mt = 0; mt < 8; ++mt)
            x3_block<1, T, false, false, VX>(st, tc[1], a2[mt], [&](int s) {
                if (mt == 0) {
                    if (s == 0) bvn = tile_from_channels(s_db2, h);
                } else {
                    x3_xd_steps<PX, 0, 4>(a2, xd, s_db2, h, bvx, bvn, 2 * (mt - 1) + s);
                }
            });
        x3_layer<8, 4, T, VX>(st, s_db3, xd, y3, h, carry, [&](int s) { x3_xd_steps<PX, 4, 4>(a2, xd, s_db2, h, bvx, bvn, s); });
        x3_layer<4, 4, T, VC4>(st, s_db4, y3, y4, h, carry, [&](int s) { x3_carry_steps<4, T>(carry, y3, s); });
        f32x16 lg[T];                                      // dconv5 (128 -> 2, no ReLU): rows 0, 1 of one more out-tile
        x3_block<4, T, false, true, VC4>(st, y4, lg, [&](int s) { x3_carry_steps<4, T>(carry, y4, s); });
        st.end_group();
        const float b50 = s_db5[0], b51 = s_db5[1];
#pragma unroll
        for (int j = 0; j < T; ++j) {
            float s0 = lg[j][0] + b50, s1 = lg[j][1] + b51;    // rows 0, 1: registers 0, 1 of lanes 0..31, one point each
            if (crop_bad) s0 = s1 = __int_as_float(DAL3_QNAN_BITS);
            const int n = n0 + 32 * j + (lane & 31);
            if (h == 0 && n < n_pts) {
                f32x2 o;
                o[0] = s0;
                o[1] = s1;
                *reinterpret_cast<f32x2*>(logits + (b * n_pts + n) * 2) = o;
                mask[b * n_pts + n] = (!crop_bad && s0 < s1) ? 1 : 0;
            }
        }
    }
}

hipError_t launch_ins_seg_decode_x3(const InsSegX3W& w, BCN pts, int c_in, int B, int N, const float* gbias, float* logits,
                                    uint8_t* mask, hipStream_t s) {
    constexpr int T = DAL3_X3_DEC_T;
    const size_t lds = 3 * X3_SEG * 1024 + X3_DEC_SMALL_BYTES;
    auto k = ins_seg_decode_x3_kernel<T>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const int tpi = (N + 32 * X3_WAVES * T - 1) / (32 * X3_WAVES * T);
    const int64_t n_groups = (int64_t)B * tpi;
    if (n_groups > 0x7fffffff) return hipErrorInvalidValue;
    const int64_t grid = n_groups < x3_cu_count() ? n_groups : x3_cu_count();
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(256), lds, s, w, pts, c_in, N, tpi, (int)n_groups, gbias, logits, mask);
    return hipGetLastError();
}
#endif  // X3_PART & 2
