// dal3_pointmlp_x3.hip — the shared-MLP kernels on fp16 MFMAs at fp32 ACCURACY (round 3): "f16x3".
//
// gfx950 has no TF32-like mode: the exact-f32 MFMA runs at 1/16 of the fp16 rate (157 vs 2,500 TFLOP/s dense), and
// rounding operands to 16 bits once costs three digits (logits 3e-3 / 3e-2 relative for fp16 / bf16, tools/
// f16x3_accuracy.py). Splitting BOTH operands of every product into two fp16 halves,
//     x = x_hi + x_lo,  w = w_hi + w_lo      (hi = fp16(v), lo = fp16(v - hi): 22 bits of significand together)
//     w x ~= w_hi x_hi + w_hi x_lo + w_lo x_hi           (the dropped w_lo x_lo is 2^-22 relative)
// gives the fp32 kernels' accuracy (logits 1.0e-6 vs 2.0e-6 against float64, same script) from THREE fp16 MFMAs with
// fp32 accumulation — 16/3 of the fp32 MFMA rate on paper. Everything else is the 16-bit family's design (dal3_lp.h):
// channels on MFMA rows, points on columns, the fp32 accumulator of layer k split in registers into the B operands of
// layer k+1 (activations never leave registers), weights as one stream of 1-KiB fragments — a (hi, lo) PAIR per k-step —
// shared by the four waves of a workgroup through the three-slot LDS ring filled by LDS-DMA, persistent workgroups.
// What is deliberately simpler than the 16-bit kernels: ONE generic block (x3_block) walks the fragment stream with a
// cursor, opens a segment when the cursor reaches its end and deals the refill's LDS-DMA instructions out one per
// k-step behind it; no per-kernel hand schedule yet.
//
// The first layer (raw coordinates, K = 3/4/8) stays on the fp32 MFMA, biases and the per-crop dconv1 term are fp32,
// logits / mask / pooled features and all I/O are fp32, as in the other two families.
#include <stdlib.h>

#include "dal3_kernels.h"
#include "dal3_lp.h"

#define X3_WAVES 4
#define X3_SEG 32                       // fragments (1 KiB) per ring segment: 16 (hi, lo) pairs

typedef f16x8_t x3v8;

// one 32-channel x 32-point activation tile as the B operands of its two k-steps (16 channels each), split in two halves
struct X3Tile {
    x3v8 hi[2], lo[2];
};

__device__ __forceinline__ f32x16 x3_mfma(x3v8 a, x3v8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// relu, then hi = fp16(x), lo = fp16(x - hi) pair by pair (register pair i of k-step s holds channels 16s + 8(i>>1)... as
// in pack_relu of dal3_lp.h: the A operands' k order is permuted to match by the packer)
__device__ __forceinline__ X3Tile x3_split_relu(const f32x16& acc) {
    X3Tile t;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        int4_t wh, wl;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x2 p = {relu1(acc[8 * s + 2 * i]), relu1(acc[8 * s + 2 * i + 1])};
            const f16x2_t h = __builtin_convertvector(p, f16x2_t);
            const f32x2 back = __builtin_convertvector(h, f32x2);
            const f32x2 rest = {p[0] - back[0], p[1] - back[1]};
            const f16x2_t l = __builtin_convertvector(rest, f16x2_t);
            wh[i] = __builtin_bit_cast(int, h);
            wl[i] = __builtin_bit_cast(int, l);
        }
        t.hi[s] = __builtin_bit_cast(x3v8, wh);
        t.lo[s] = __builtin_bit_cast(x3v8, wl);
    }
    return t;
}

// The weight stream of one kernel, walked by a wave-uniform cursor. Stream order = consumption order; a k-step is a
// (hi, lo) fragment pair; the stream is padded to whole segments per group of points, so every group starts at the
// beginning of a segment. next(): the segment in use is finished (every fragment of it is in registers or used) ->
// counted wait + barrier (LdsRing::acquire_wait), then the refill of the slot just freed is PENDING: pump() issues one
// of its LDS-DMA instructions and is called once per k-step from the blocks that follow, i.e. the refill is spread over
// the next segment's MFMAs (the third slot gives it a further segment to land). All parts are out before the next
// acquire_wait (flush()), whose counted vmcnt relies on it.
struct X3Stream {
    typedef LdsRing<X3_SEG, 3> Ring;
    Ring ring;
    int cur, pending;
    __device__ __forceinline__ void init(const void* stream, char* lds, int n_segs, int wave, int lane) {
        ring.init(stream, lds, n_segs, wave, lane, true);  // segments 0 and 1 in flight
        cur = X3_SEG;                                      // the first pair() opens segment 0: every group of points then
        pending = 0;                                       // starts in the same cursor state (cur == X3_SEG)
    }
    __device__ __forceinline__ void pump() {
        if (pending > 0) {
            ring.issue_part(Ring::MY_LOADS - pending);
            if (--pending == 0) ring.issue_done();
        }
    }
    __device__ __forceinline__ void next() {
        while (pending > 0) pump();
        ring.acquire_wait();
        pending = Ring::MY_LOADS;
        cur = 0;
    }
    // the (hi, lo) pair of the next k-step
    __device__ __forceinline__ void pair(x3v8& wh, x3v8& wl) {
        if (cur == X3_SEG) next();
        wh = ring.template frag<FP16>(cur);
        wl = ring.template frag<FP16>(cur + 1);
        cur += 2;
    }
    // end of a group of points: whatever is left of the open segment is padding, the next pair() opens the following one
    __device__ __forceinline__ void skip_padding() { cur = X3_SEG; }
};

// acc[j] += W'(32 x 32 KT) . X[j] for the wave's T point tiles: 2 KT k-steps, three MFMAs per k-step and tile.
// SWAP: operands exchanged, acc[j] += X[j]^T . W'^T — the transposed tile of the max-pooled layers (points on the
// accumulator's registers, channels on its lanes: the max over points is a max over registers).
template <int KT, int T, bool SWAP = false>
__device__ __forceinline__ void x3_block(X3Stream& st, const X3Tile (&X)[T][KT], f32x16 (&acc)[T]) {
    x3v8 wh, wl, nh, nl;
    st.pair(wh, wl);
#pragma unroll
    for (int s = 0; s < 2 * KT; ++s) {
        // the next k-step's pair is read before this one's MFMAs (an LDS round trip hides under 3 T MFMAs) — unless it
        // lies in the next segment, which the barrier in pair() has to open first
        const bool ahead = s + 1 < 2 * KT && st.cur != X3_SEG;
        if (ahead) st.pair(nh, nl);
        DAL3_SCHED_FENCE();
#pragma unroll
        for (int j = 0; j < T; ++j) {
            const x3v8 xh = X[j][s >> 1].hi[s & 1], xl = X[j][s >> 1].lo[s & 1];
            if (SWAP) {
                acc[j] = x3_mfma(xh, wh, acc[j]);
                acc[j] = x3_mfma(xl, wh, acc[j]);
                acc[j] = x3_mfma(xh, wl, acc[j]);
            } else {
                acc[j] = x3_mfma(wh, xh, acc[j]);
                acc[j] = x3_mfma(wh, xl, acc[j]);
                acc[j] = x3_mfma(wl, xh, acc[j]);
            }
        }
        st.pump();
        DAL3_SCHED_FENCE();
        if (s + 1 < 2 * KT) {
            if (!ahead) st.pair(nh, nl);
            wh = nh;
            wl = nl;
        }
    }
}

// Y = split(relu(W' X + b')) for a 32 KT -> 32 MT layer; bias: LDS pointer to the layer's folded bias
template <int KT, int MT, int T>
__device__ __forceinline__ void x3_layer(X3Stream& st, const float* bias, const X3Tile (&X)[T][KT], X3Tile (&Y)[T][MT], int h) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        f32x16 acc[T];
        const f32x16 b = tile_from_channels(bias + 32 * m, h);
#pragma unroll
        for (int j = 0; j < T; ++j) acc[j] = b;
        x3_block<KT, T>(st, X, acc);
#pragma unroll
        for (int j = 0; j < T; ++j) Y[j][m] = x3_split_relu(acc[j]);
    }
}

// the max-pooled last layer: n_tiles out-tiles computed transposed, each tile's maxima over the wave's 32 T points joined
// into the workgroup's LDS array (lp_tile_max_t of dal3_lp.h: bias added after the max, ReLU on the bit pattern, LDS
// integer atomicMax). Two accumulator sets: a tile's epilogue is issued behind the next tile's first MFMAs.
template <int KT, int T>
__device__ __forceinline__ void x3_max_layer(X3Stream& st, const float* bias, const X3Tile (&X)[T][KT], int* smax, int n_tiles,
                                             int lane) {
    f32x16 acc[2][T];
    for (int m = 0; m < n_tiles; m += 2) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
#pragma unroll
            for (int j = 0; j < T; ++j) acc[c][j] = f32x16{};
            x3_block<KT, T, true>(st, X, acc[c]);
            lp_tile_max_t<T>(acc[c], bias + 32 * (m + c), smax + 32 * (m + c), lane);
        }
    }
}

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
        x3_layer<K2, M2, T>(st, s_bias, x1, x2, h);
        x3_layer<K3, M3, T>(st, s_bias + C2, x2, x3, h);
        x3_max_layer<K4, T>(st, s_bias + C2 + C3, x3, s_max, 16, lane);
        st.skip_padding();
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
        x3_layer<2, 2, T>(st, s_bias, x1, x2, h);
        x3_layer<2, 2, T>(st, s_bias + 64, x2, x3, h);
        x3_layer<2, 4, T>(st, s_bias + 128, x3, x4, h);
        x3_max_layer<4, T>(st, s_bias + 256, x4, s_max, 32, lane);
        st.skip_padding();
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

// ------------------------------------------------------------------------------------------------ decode
// conv1 (fp32), conv2, then dconv1's per-point part in 16 chunks of 32 channels (initial value: the crop's dconv1 term,
// fp32) each split and fed straight into dconv2's eight resident accumulator tiles, dconv3, dconv4, and dconv5 as one
// more out-tile (rows 0, 1) -> logits, mask. Stream per group: conv2 16 | 16 x { dconv1a chunk 8, dconv2 chunk 32 } |
// dconv3 128 | dconv4 64 | dconv5 16 fragments = 27 segments. LDS small arrays: b2 64 | db2 256 | db3 128 | db4 128 |
// db5 32 | gb 512 | b1 64 | w1 256 floats.
#define X3_DEC_SMALL_BYTES 6144
#define X3_DEC_SEGS 27
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
    prefetch(blockIdx.x);
    for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
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
        x3_layer<2, 2, T>(st, s_bias, x1, x2, h);

        f32x16 a2[T][8];
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) {
            const f32x16 bv = tile_from_channels(s_db2 + 32 * mt, h);
#pragma unroll
            for (int j = 0; j < T; ++j) a2[j][mt] = bv;
        }
        for (int c = 0; c < 16; ++c) {                     // dconv1 chunk c -> dconv2
            f32x16 t[T];
            const f32x16 gv = tile_from_channels(s_gb + 32 * c, h);
#pragma unroll
            for (int j = 0; j < T; ++j) t[j] = gv;
            x3_block<2, T>(st, x2, t);
            X3Tile tc[T][1];
#pragma unroll
            for (int j = 0; j < T; ++j) tc[j][0] = x3_split_relu(t[j]);
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
                f32x16 acc[T];
#pragma unroll
                for (int j = 0; j < T; ++j) acc[j] = a2[j][mt];
                x3_block<1, T>(st, tc, acc);
#pragma unroll
                for (int j = 0; j < T; ++j) a2[j][mt] = acc[j];
            }
        }
        X3Tile xd[T][8], y3[T][4], y4[T][4];
#pragma unroll
        for (int j = 0; j < T; ++j) {
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) xd[j][mt] = x3_split_relu(a2[j][mt]);
        }
        x3_layer<8, 4, T>(st, s_db3, xd, y3, h);
        x3_layer<4, 4, T>(st, s_db4, y3, y4, h);
        f32x16 lg[T];                                      // dconv5 (128 -> 2, no ReLU): rows 0, 1 of one more out-tile
        {
            const f32x16 bv = tile_from_channels(s_db5, h);
#pragma unroll
            for (int j = 0; j < T; ++j) lg[j] = bv;
            x3_block<4, T>(st, y4, lg);
        }
        st.skip_padding();
#pragma unroll
        for (int j = 0; j < T; ++j) {
            float s0 = lg[j][0], s1 = lg[j][1];            // rows 0, 1: registers 0, 1 of lanes 0..31, one point each
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
