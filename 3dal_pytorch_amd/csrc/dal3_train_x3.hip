// dal3_train_x3.hip — the training step's big linear layers on the "f16x3" engine (dal3_x3.h): z = act(a) W^T + b with every
// product formed as w_hi x_hi + w_hi x_lo + w_lo x_hi on the fp16 MFMA, fp32 accumulate — the accuracy of the fp32-MFMA
// kernels of dal3_train.hip (1e-6 of the output's range) at a third of their MFMA time. The forward's operands
// (post-BatchNorm activations, weights) are O(1), inside fp16's exponent range. A dgrad's operand, dz, is 1e-6 and smaller:
// its producer (tr_bnbwd_apply_kernel) leaves the bit pattern of its largest |value| in device words, and the kernel
// multiplies the operand by the power of two that brings that value to 2^14 and the result by the inverse (in_amax).
// wgrad — both operands transposed with respect to memory — is the second kernel of this file (tr_wgrad_x3_kernel, below).
//
// Unlike the eval kernels the activations come from HBM, not from the previous layer's accumulators: a wave owns T = 2
// tiles of 32 points and MTB output tiles; per 32-channel k-tile it loads its 2 x 32 x 32 fp32 inputs (16 B per lane and
// instruction, two k-tiles ahead of their use), applies the input activation (BatchNorm affine + ReLU of the producing
// layer) and splits them into (hi, lo) fp16 operands UNDER the MFMAs of the k-tile in front (x3_block's side work), and
// walks the layer's weight stream — [k-tile][out-tile][k-step][hi | lo] fragments, shared by the workgroup's four waves
// through the three-slot LDS ring — with the compile-time cursor of X3Stream: a loop iteration is two k-tiles = one or
// two whole ring segments. Workgroups are persistent (one per CU) over groups of 256 points of ONE output block.
#include <stdlib.h>

#include <type_traits>

#include "dal3_kernels.h"
#include "dal3_lp.h"
#include "dal3_x3.h"

#define TRX_T 2

// Two units (two pairs of values: registers 4 q .. 4 q + 3 of a point tile, i.e. channels 8 q + 4 h .. + 3 of the k-tile) of
// a staged k-tile: activation, then the (hi, lo) split of x3_split_unit. xq / scq / shq: the quad, its scale and shift.
template <bool ACT>
__device__ __forceinline__ void trx_split_quad(const f32x4& xq, const f32x4& scq, const f32x4& shq, float floor_v, X3Tile& t, int q) {
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int s = q >> 1, i = 2 * (q & 1) + u;          // register pair 4 q + 2 u = 8 s + 2 i
        f32x2 p = {xq[2 * u], xq[2 * u + 1]};
        if (ACT) {
            const f32x2 a = {scq[2 * u], scq[2 * u + 1]}, b = {shq[2 * u], shq[2 * u + 1]};
            p = p * a + b;                                  // (contracted to one packed fma)
            p[0] = __builtin_fmaxf(p[0], floor_v);          // floor_v: 0 with ReLU, -FLT_MAX without
            p[1] = __builtin_fmaxf(p[1], floor_v);
        }
        const f16x2_t hh = __builtin_convertvector(p, f16x2_t);
        const int hi = __builtin_bit_cast(int, hh);
        int lo;
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
            : "=&v"(lo)
            : "v"(hi), "v"(p[0]), "v"(p[1]));
        t.hi[s][i] = hi;
        t.lo[s][i] = lo;
    }
}

// Every VMEM load of a wave is an LDS-DMA instruction (inline asm, invisible to hipcc's wait-count pass: a compiler-visible
// load in the same queue would be waited for with a count that is too small by the DMA loads issued behind it, i.e. together
// with loads issued moments ago): the weight ring's refills, the wave's own input k-tiles (staged in LDS, 8 KiB per k-tile:
// lane (m, h) of instruction (j, q) fetches the 16 bytes it will read back, so staging is conflict-free), and — wave 0 —
// the next group's bias row. The order of issue is this source's, so a segment's open waits with an exact count
// (X3StreamT<true, 2, 1, 8>: vmcnt(8) + barrier): the 8 instructions of the input request issued under k-step 8 stay in
// flight over the open at k-step 14 and have until the NEXT open to land (22 k-steps, ~2.2 us); in a group's last k-tile
// the stores / atomics issued behind the request are counted with them (st.keep).
// One k-tile = 8 out-tiles x 4 fragments = one ring segment (MTB = 8: c_out % 256 == 0). Pipeline, by k-tile n:
//   input(n + 3) requested under k-step 8 of k-tile n  |  input(n + 1) read from its stage and split under the first 8 k-steps of
//   k-tile n  |  MFMAs on split(n)  |  in a group's last k-tile: an out-tile's bias add + stores under the next tile's MFMAs.
// LDS: ring 2 x 32 KiB | input stages 4 waves x 2 x 8 KiB | scale | shift (c_in floats each) | bias rows 2 x 256 floats.
#ifdef TRX_ABL_NOSTORE_X3                                    // timing experiment only: the output is not written
#define TRX_STORE(p, o) do { if ((o)[0] == 12345.678f) *reinterpret_cast<f32x4*>(p) = (o); } while (0)
#else
#define TRX_STORE(p, o) (*reinterpret_cast<f32x4*>(p) = (o))
#endif
#define TRX_STAGE_BYTES 8192
#ifndef TRX_EARLY_REQUEST
#define TRX_EARLY_REQUEST 1             // 0: request a stage's next k-tile at the open (A/B)
#endif
// POOL: the pooled layer's forward without its output tensor (tr_linear_pool_kernel of dal3_train.hip): the blocks run
// with the MFMA operands swapped (points on the accumulator's registers, channels on its lanes), an out-tile's epilogue is
// conv bias -> BN affine -> ReLU -> max over the tile's 32 points with the point index (first maximum wins) -> the packed
// 64-bit atomicMax of tr_segmax_kernel; bias / out_scale / out_shift are per channel (LDS: 3 x 256 floats), z is unused.
template <bool ACT, bool POOL>
__global__ __launch_bounds__(256) void tr_linear_x3_kernel(const float* __restrict__ a, int64_t lda, int c_in,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           int relu_in, const uint16_t* __restrict__ wpk,
                                                           const float* __restrict__ bias, int64_t seg, int c_out,
                                                           float* __restrict__ z, int64_t ldz, int n_mblk, int n_groups,
                                                           const float* __restrict__ out_scale, const float* __restrict__ out_shift,
                                                           unsigned long long* __restrict__ packed,
                                                           const uint32_t* __restrict__ in_amax) {
    constexpr int T = TRX_T, MTB = 8;
    constexpr int PER = 2, VS = x3_vpg(PER, T) + 1;         // a k-tile's 8 T split units done within its first 8 k-steps
    static_assert(T == 2, "one point tile per k-step of a block in the epilogue; stage layout");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const stage0 = smem + 2 * X3_SEG * 1024;
    float* s_sc = reinterpret_cast<float*>(stage0 + X3_WAVES * 2 * TRX_STAGE_BYTES);
    float* s_sh = s_sc + c_in;
    float* s_bias = s_sh + c_in;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5, m = lane & 31;
    const float floor_v = relu_in ? 0.0f : -3.0e38f;
    // an operand below fp16's range (a dgrad's dz; ACT instantiation, scale == NULL): in_amax holds the bits of its largest
    // |value|; the operand is multiplied by the power of two that brings that value to [2^14, 2^15) — inside fp16's range
    // with the full 22 bits of the split for everything down to 2^-28 of it — and the result by the inverse. Both exact.
    float s_in = 1.0f, s_out = 1.0f;
    if (in_amax) {
        uint32_t mxb = in_amax[lane];                           // 64 words (dal3_tr_bnbwd_apply_amax spreads its atomics): their maximum
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const uint32_t other = (uint32_t)__shfl_xor((int)mxb, off);
            mxb = other > mxb ? other : mxb;
        }
        const int ex = (int)((__builtin_amdgcn_readfirstlane((int)mxb) >> 23) & 0xff);           // biased exponent of the max
        int sh_e = ex == 0 ? 0 : 14 - (ex - 127);
        sh_e = sh_e < -100 ? -100 : (sh_e > 100 ? 100 : sh_e);
        s_in = __int_as_float((127 + sh_e) << 23);
        s_out = __int_as_float((127 - sh_e) << 23);
    }
    const int mblk = (int)(blockIdx.x % (unsigned)n_mblk);
    const int g0 = (int)(blockIdx.x / (unsigned)n_mblk), gstride = (int)(gridDim.x / (unsigned)n_mblk);
    if (g0 >= n_groups) return;
    const int KT = c_in / 32, mt0 = mblk * MTB;
    const auto bias_row = [&](int g) {                          // (scalar; a group of 256 points lies in one segment)
        return bias + (seg > 0 ? (int64_t)((uint32_t)((int64_t)g * 256) / (uint32_t)seg) * c_out : 0) + 32 * mt0;
    };
    if (ACT) {
        for (int i = threadIdx.x; i < c_in; i += 256) {
            s_sc[i] = scale ? scale[i] : s_in;
            s_sh[i] = scale ? shift[i] : 0.0f;
        }
    }
    if (POOL) {                                                 // per-channel constants of this output block
        s_bias[threadIdx.x] = bias ? bias[32 * mt0 + threadIdx.x] : 0.0f;
        s_bias[256 + threadIdx.x] = out_scale[32 * mt0 + threadIdx.x];
        s_bias[512 + threadIdx.x] = out_shift[32 * mt0 + threadIdx.x];
    } else {
        s_bias[threadIdx.x] = bias ? bias_row(g0)[threadIdx.x] : 0.0f;   // 256 = 32 MTB floats: the first group's row, buffer 0
        s_bias[256 + threadIdx.x] = 0.0f;
    }
    __syncthreads();
    X3StreamT<true, 2, 1, TRX_EARLY_REQUEST ? 8 : 0> st;      // (8: the input request's LDS-DMA instructions stay in flight over an open)
    st.init(wpk + (size_t)mblk * KT * MTB * 4 * 512, smem, KT, wave, lane);

    char* const stage = stage0 + wave * (2 * TRX_STAGE_BYTES);  // this wave's two input stages (k-tile n in stage n & 1)
    const uint32_t lane_off = (uint32_t)(((int64_t)m * lda + 4 * h) * sizeof(float));
    // k-tile tau of group g's sequence (tau >= KT: a following group's; past the last group: a harmless re-read of g's own)
    const auto request = [&](int buf, int g, int tau) {
        const int gq = tau >= 2 * KT ? 2 : tau >= KT ? 1 : 0;
#ifdef TRX_ABL_SAMEROWS                                       // timing experiment only: every group reads the first group's rows (L2 hits)
        const int gt = (int)(blockIdx.x / (unsigned)n_mblk) + 0 * (g + gq);
#else
        const int gt = g + gq * gstride < n_groups ? g + gq * gstride : g;
#endif
        const char* src = reinterpret_cast<const char*>(a + ((int64_t)gt * 256 + wave * 64) * lda + 32 * (tau - gq * KT));
#pragma unroll
        for (int j = 0; j < T; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                lds_dma16(src + ((int64_t)32 * j * lda + 8 * q) * sizeof(float), lane_off, stage + buf * TRX_STAGE_BYTES + (4 * j + q) * 1024);
    };
    // quad q of point tile j of a staged k-tile / of a k-tile's scale and shift: this lane's four values
    const auto staged = [&](int buf, int j, int q) {
        return *reinterpret_cast<const f32x4*>(stage + buf * TRX_STAGE_BYTES + (4 * j + q) * 1024 + lane * 16);
    };
    const auto act_quad = [&](const float* v, int k, int q) { return *reinterpret_cast<const f32x4*>(v + 32 * k + 8 * q + 4 * h); };
    X3Tile xs[2][T][1];                                     // split k-tiles
    f32x4 xq, scq = f32x4{}, shq = f32x4{};                 // the quad split next: read one k-step ahead of its split
    // prologue: k-tiles 0 and 1 of the first group staged, 0 split in the open, 2 requested into its stage
    request(0, g0, 0);
    request(1, g0, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int j = 0; j < T; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            xq = staged(0, j, q);
            if (ACT) {
                scq = act_quad(s_sc, 0, q);
                shq = act_quad(s_sh, 0, q);
            }
            trx_split_quad<ACT>(xq, scq, shq, floor_v, xs[0][j][0], q);
        }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (stage 0 has been read: it may be overwritten)
    request(0, g0, 2);
    xq = staged(1, 0, 0);
    if (ACT) {
        scq = act_quad(s_sc, 1 % KT, 0);
        shq = act_quad(s_sh, 1 % KT, 0);
    }

    // POOL: the epilogue of out-tile t, point tile j of the group whose first row (this wave's) is `row`
    const auto pool_tile = [&](const f32x16& acc_tj, int t, int j, int64_t row) {
        const int c = 32 * t + m;                               // this lane's channel within the output block
        const float bch = s_bias[c], osc = s_bias[256 + c], osh = s_bias[512 + c];
        float bv = -1.0f;
        int bi = 0;
#pragma unroll
        for (int r = 0; r < 16; ++r) {                          // increasing point index: '>' keeps the first maximum
            const float y = fmaxf(__builtin_fmaf(acc_tj[r] + bch, osc, osh), 0.0f);
            if (y > bv) {
                bv = y;
                bi = (r & 3) + 8 * (r >> 2) + 4 * h;
            }
        }
        const float ov = __shfl_xor(bv, 32);
        const int oi = __shfl_xor(bi, 32);
        if (ov > bv || (ov == bv && oi < bi)) {
            bv = ov;
            bi = oi;
        }
        if (h == 0) {
            const int64_t p0 = row + 32 * j;
            const int64_t s_idx = (int64_t)((uint32_t)p0 / (uint32_t)seg);
            const uint32_t in_seg = (uint32_t)(p0 - s_idx * seg) + (uint32_t)bi;
            const unsigned long long key = ((unsigned long long)__float_as_uint(bv) << 32) | (0xffffffffu - in_seg);
            atomicMax(packed + s_idx * c_out + 32 * mt0 + c, key);
        }
    };
    int par = 0;                                            // which bias buffer this group reads
    for (int g = g0; g < n_groups; g += gstride, par ^= 1) {
        const int gn = g + gstride < n_groups ? g + gstride : g;          // (the last group re-reads itself: uniform control flow)
        const int64_t row = (int64_t)g * 256 + wave * 64;
        const float* sb = s_bias + 256 * par;
        f32x16 acc[MTB][T];
        f32x16 bprev = f32x16{};
        // one pair of k-tiles = two segments (kt even)
        auto pair = [&](int kt, auto first_c, auto last_c) {
            constexpr bool FIRST = decltype(first_c)::value, LAST = decltype(last_c)::value;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int k = kt + kk;                              // this k-tile; its split is xs[kk], k + 1 is in stage kk ^ 1
                const int k1 = k + 1 < KT ? k + 1 : 0;              // the k-tile split under this one (index of its activation)
                const int k2 = k + 2 < KT ? k + 2 : k + 2 - KT;     // the one after it
#pragma unroll
                for (int t = 0; t < MTB; ++t) {
                    const auto side = [&](int s) {
                        const int sg = 2 * t + s;                   // k-step of the segment
                        // k-tile k + 1 (stage kk ^ 1): point tile 0 split under k-steps 0..3, tile 1 under 4..7, a quad per
                        // k-step, read (with its scale and shift) a k-step ahead; k-step 15 — behind the open — reads the
                        // first quad of k-tile k + 2, which has landed in stage kk
                        if (sg < 8) trx_split_quad<ACT>(xq, scq, shq, floor_v, xs[kk ^ 1][sg >> 2][0], sg & 3);
                        // stage kk ^ 1 has been read (its last quad at k-step 6, split at 7): it takes k-tile k + 3 NOW, six
                        // k-steps before the open — 22 k-steps instead of 16 to land before the open after this one drains
                        if (TRX_EARLY_REQUEST && sg == 8) {
                            // In a group's last k-tile the stores (POOL: the atomics) of the out-tiles that finish behind this
                            // point are issued behind the request too: 7 sides x 4 stores (x 1 atomic) until the open at k-step
                            // 14. They stay in flight with it — counted exactly, see X3StreamT::open.
                            if (LAST && kk == 1) st.keep = 8 + (POOL ? 7 : 28);
                            // (the next group's bias row first: at the open the request's 8 instructions must be the youngest)
                            if (!POOL && LAST && kk == 1 && bias && wave == 0)
                                lds_dma16(reinterpret_cast<const char*>(bias_row(gn)), (uint32_t)(lane * 16),
                                          reinterpret_cast<char*>(s_bias + 256 * (par ^ 1)));
                            request(kk ^ 1, g, k + 3);
                        }
                        if (sg < 7 || sg == 15) {
                            const int nb = sg == 15 ? kk : kk ^ 1, ns = sg == 15 ? 0 : sg + 1, nk = sg == 15 ? k2 : k1;
                            xq = staged(nb, ns >> 2, ns & 3);
                            if (ACT) {
                                scq = act_quad(s_sc, nk, ns & 3);
                                shq = act_quad(s_sh, nk, ns & 3);
                            }
                        }
                        if (POOL && LAST && kk == 1) {
                            if (t > 0) pool_tile(acc[t - 1][s], t - 1, s, row);
                        } else if (LAST && kk == 1) {
                            // an output tile is final behind its last block: bias, store — under the next tile's MFMAs
                            if (t > 0) {
                                const int j = s;                    // (T == 2: one point tile per k-step)
                                float* zp = z + (row + 32 * j + m) * ldz + 32 * (mt0 + t - 1) + 4 * h;
#pragma unroll
                                for (int q = 0; q < 4; ++q) {
                                    f32x4 o = {acc[t - 1][j][4 * q], acc[t - 1][j][4 * q + 1], acc[t - 1][j][4 * q + 2], acc[t - 1][j][4 * q + 3]};
                                    o[0] = __builtin_fmaf(o[0], s_out, bprev[4 * q]);
                                    o[1] = __builtin_fmaf(o[1], s_out, bprev[4 * q + 1]);
                                    o[2] = __builtin_fmaf(o[2], s_out, bprev[4 * q + 2]);
                                    o[3] = __builtin_fmaf(o[3], s_out, bprev[4 * q + 3]);
                                    TRX_STORE(zp + 8 * q, o);
                                }
                            }
                            if (s == 1) bprev = tile_from_channels(sb + 32 * t, h);      // (behind its last use: tile t's, for the next block)
                        }
                    };
                    // the segment's open (in its last out-tile's block): k-tile k + 1's stage has been read (k-steps 0..3 of
                    // this segment) and takes k-tile k + 3
                    const auto opened = [&](int) {
                        st.keep = 8;
                        if (!TRX_EARLY_REQUEST) {
                            request(kk ^ 1, g, k + 3);
                            if (!POOL && LAST && kk == 1 && bias && wave == 0)
                                lds_dma16(reinterpret_cast<const char*>(bias_row(gn)), (uint32_t)(lane * 16),
                                          reinterpret_cast<char*>(s_bias + 256 * (par ^ 1)));
                        }
                    };
                    if (FIRST && kk == 0)
                        x3_block<1, T, POOL, true, VS>(st, xs[kk], acc[t], side, opened);
                    else
                        x3_block<1, T, POOL, false, VS>(st, xs[kk], acc[t], side, opened);
                }
            }
        };
        // FIRST / LAST are compile-time (the first block takes C = 0, the last k-tile carries the stores); in between a
        // runtime loop whose body is whole segments (the cursor state repeats: pin)
        if (KT == 2) {
            st.pin(st.START, 0);
            pair(0, std::true_type{}, std::true_type{});
        } else {
            st.pin(st.START, 0);
            pair(0, std::true_type{}, std::false_type{});
            const int c0 = st.cur, p0 = st.pending;
            for (int kt = 2; kt + 2 < KT; kt += 2) {
                st.pin(c0, p0);
                pair(kt, std::false_type{}, std::false_type{});
            }
            st.pin(c0, p0);
            pair(KT - 2, std::false_type{}, std::true_type{});
        }
        st.end_group();
        if (POOL) {
#pragma unroll
            for (int j = 0; j < T; ++j) pool_tile(acc[MTB - 1][j], MTB - 1, j, row);
        } else {                                                // the last output tile
            const int t = MTB - 1;
#pragma unroll
            for (int j = 0; j < T; ++j) {
                float* zp = z + (row + 32 * j + m) * ldz + 32 * (mt0 + t) + 4 * h;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 o = {acc[t][j][4 * q], acc[t][j][4 * q + 1], acc[t][j][4 * q + 2], acc[t][j][4 * q + 3]};
                    o[0] = __builtin_fmaf(o[0], s_out, bprev[4 * q]);
                    o[1] = __builtin_fmaf(o[1], s_out, bprev[4 * q + 1]);
                    o[2] = __builtin_fmaf(o[2], s_out, bprev[4 * q + 2]);
                    o[3] = __builtin_fmaf(o[3], s_out, bprev[4 * q + 3]);
                    TRX_STORE(zp + 8 * q, o);
                }
            }
        }
    }
}

static int trx_cu_count() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
            v = 256;
        n = v;
    }
    return n;
}

// 0: this call does not take the f16x3 kernel; otherwise its packed-image layout code, 0x100 | MTB
int tr_linear_x3_layout(int64_t M, int c_in, int64_t seg, int c_out, int accumulate, int has_act) {
    (void)has_act;
    if (accumulate || M < 4096 || M % 256 != 0 || M >= ((int64_t)1 << 31) || c_in % 64 != 0 || c_in < 64 || c_in > 2048 ||    // (LDS: ring 64 KiB + stages 64 KiB + 8 c_in + 3 KiB <= 160 KiB)
        c_out % 256 != 0 || (seg != 0 && seg % 256 != 0))
        return 0;
    return 0x100 | 8;
}

static hipError_t trx_launch(bool pool, const float* a, int64_t M, int c_in, int64_t lda, const float* scale, const float* shift,
                             int relu_in, const uint16_t* wpk, const float* bias, int64_t seg, int c_out, float* z, int64_t ldz,
                             const float* out_scale, const float* out_shift, unsigned long long* packed, hipStream_t s,
                             const uint32_t* in_amax = nullptr) {
    const int n_mblk = c_out / 256;
    const int n_groups = (int)(M / 256);
    int per = trx_cu_count() / n_mblk;                               // workgroups per output block: one workgroup per CU in all
    if (per < 1) per = 1;
    if (per > n_groups) per = n_groups;
    const unsigned grid = (unsigned)(per * n_mblk);
    const size_t lds = 2 * X3_SEG * 1024 + X3_WAVES * 2 * TRX_STAGE_BYTES + 2 * (size_t)c_in * sizeof(float) + 3 * 256 * sizeof(float);
    const auto go = [&](auto kern) -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, s, a, lda, c_in, scale, shift, relu_in, wpk, bias, seg, c_out, z, ldz,
                           n_mblk, n_groups, out_scale, out_shift, packed, in_amax);
        return hipGetLastError();
    };
    if (pool) return scale ? go(tr_linear_x3_kernel<true, true>) : go(tr_linear_x3_kernel<false, true>);
    return (scale || in_amax) ? go(tr_linear_x3_kernel<true, false>) : go(tr_linear_x3_kernel<false, false>);
}

hipError_t launch_tr_linear_x3(const float* a, int64_t M, int c_in, int64_t lda, const float* scale, const float* shift, int relu_in,
                               const uint16_t* wpk, int layout, const float* bias, int64_t seg, int c_out, float* z, int64_t ldz,
                               hipStream_t s, const uint32_t* in_amax) {
    if ((layout & 0xff) != 8) return hipErrorInvalidValue;
    return trx_launch(false, a, M, c_in, lda, scale, shift, in_amax ? 0 : relu_in, wpk, bias, seg, c_out, z, ldz, nullptr, nullptr,
                      nullptr, s, in_amax);
}

// the pooled layer (launch_tr_linear_pool of dal3_train.hip on the f16x3 engine); ok: tr_linear_pool_x3_ok()
bool tr_linear_pool_x3_ok(int64_t M, int c_in, int64_t seg, int c_out) {
    return tr_linear_x3_layout(M, c_in, seg, c_out, 0, 1) != 0 && seg > 0;
}
hipError_t launch_tr_linear_pool_x3(const float* a, int64_t M, int c_in, int64_t lda, const float* scale, const float* shift,
                                    int relu_in, const float* W, int64_t ldw, const float* bias, const float* out_scale,
                                    const float* out_shift, int64_t seg, int c_out, float* g, int32_t* arg, float* ws,
                                    unsigned long long* packed, hipStream_t s) {
    const int64_t n_seg = M / seg;
    hipError_t e = launch_fill_words(packed, (size_t)n_seg * c_out * 2, 0u, s);
    if (e != hipSuccess) return e;
    dal3_tr_pack_item it{W, ldw, 0, c_out, c_in, 0x108, ws};
    e = launch_tr_pack_many(&it, 1, s);
    if (e != hipSuccess) return e;
    e = trx_launch(true, a, M, c_in, lda, scale, shift, relu_in, reinterpret_cast<const uint16_t*>(ws), bias, seg, c_out, nullptr, 0,
                   out_scale, out_shift, packed, s);
    if (e != hipSuccess) return e;
    return launch_tr_segmax_unpack(packed, n_seg * c_out, g, arg, s);
}

// ================================================================================================ wgrad
// dW[co][ci] = sum over the points p of dz[p][co] * act(a[p][ci]) on the same arithmetic. Both operands are "transposed" with
// respect to memory (a lane of an MFMA operand holds 8 consecutive POINTS of one channel; memory holds a point's channels
// together), so a block of 16 points is staged through LDS: loaded row-wise (16 B per lane, coalesced), scaled (dz: the
// power of two from its amax words, as the dgrad) or sent through the producing layer's activation (a), split into
// (hi, lo) fp16 and written as [point][channel] images — 128-channel panels of 256-byte rows, 16-byte chunks
// XOR-swizzled by the row (cdna guide T10 (b): conflict-free for the transposed reads) — and read back with gfx950's
// ds_read_b64_tr_b16, which hands every lane 4 points of its channel. No LDS-DMA here: every load is the compiler's, its
// wait counts are exact. A workgroup of 4 waves (WM x WN) owns a (WM MT 32) x (WN KT 32) block of dW and a slice of the
// points; three register sets of raw blocks (a block is loaded two iterations before it is converted), two LDS stages,
// one barrier per 16 points. The slices' partial sums go to part[slice][co][ci] and are added in slice order by
// tr_wgrad_final_kernel (dal3_train.hip).
typedef __fp16 trx_fh4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
typedef int int2_t __attribute__((ext_vector_type(2)));

__device__ __forceinline__ int wgx_off(int row, int chan) {
    const int c = chan & 127, ch = c >> 3;                  // ch: 16-byte chunk of the panel's 256-byte row
    return (chan >> 7) * 4096 + 256 * row + 16 * (ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) + 2 * (c & 7);
}
__host__ __device__ constexpr int wgx_image_bytes(int c_blk) { return (c_blk + 127) / 128 * 4096; }

// four consecutive channels of one point -> 8 bytes of the hi image and 8 of the lo image
__device__ __forceinline__ void wgx_put(const f32x4& v, char* hi_img, char* lo_img, int off) {
    int2_t hi, lo;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const f32x2 p = {v[2 * u], v[2 * u + 1]};
        const f16x2_t hh = __builtin_convertvector(p, f16x2_t);
        const int h = __builtin_bit_cast(int, hh);
        int l;
        asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
            : "=&v"(l)
            : "v"(h), "v"(p[0]), "v"(p[1]));
        hi[u] = h;
        lo[u] = l;
    }
    *reinterpret_cast<int2_t*>(hi_img + off) = hi;
    *reinterpret_cast<int2_t*>(lo_img + off) = lo;
}
// the MFMA operand of a 32-channel tile (first channel chan0) for the 16 points of a stage image: two transposed reads
__device__ __forceinline__ x3v8 wgx_operand(const char* img, int chan0, int lane) {
    const int g = lane >> 4, j = lane & 15, q = j >> 2, p = j & 3;
    const int chan = chan0 + 16 * (g & 1) + 4 * p;
    typedef __attribute__((address_space(3))) trx_fh4* lds_fh4;
    const trx_fh4 r0 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fh4)(img + wgx_off(8 * (g >> 1) + q, chan)));
    const trx_fh4 r1 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_fh4)(img + wgx_off(8 * (g >> 1) + 4 + q, chan)));
    const int2_t a = __builtin_bit_cast(int2_t, r0), b = __builtin_bit_cast(int2_t, r1);
    const int4_t v = {a[0], a[1], b[0], b[1]};
    return __builtin_bit_cast(x3v8, v);
}

#ifndef WGX_VPG
#define WGX_VPG 6
#endif
template <int MT, int KT, int WM, int WN, int NG = 3>     // NG: register sets of raw blocks (a block is loaded NG - 1 iterations before its conversion)
__global__ __launch_bounds__(64 * WM * WN) void tr_wgrad_x3_kernel(const float* __restrict__ dz, int64_t lddz, const float* __restrict__ a,
                                                          int64_t lda, const float* __restrict__ scale,
                                                          const float* __restrict__ shift, int relu_in,
                                                          const uint32_t* __restrict__ dz_amax, int64_t M, int64_t slice_pts,
                                                          int n_cb, int c_out, int c_in, float* __restrict__ part) {
    static_assert(WM * WN == 4 || WM * WN == 8, "four waves, or eight (two per SIMD: 256 registers each)");
    constexpr int CO = WM * MT * 32, CI = WN * KT * 32, NT = 64 * WM * WN;
    constexpr int NA = 4 * CO / NT, NB = 4 * CI / NT;       // 16-byte loads per thread and 16-point block
    static_assert(NA >= 1 && NB >= 1 && (4 * CO) % NT == 0 && (4 * CI) % NT == 0, "whole loads per thread");
    constexpr int IMG_A = wgx_image_bytes(CO), IMG_B = wgx_image_bytes(CI), STAGE = 2 * IMG_A + 2 * IMG_B;
    constexpr int PERIOD = NG == 3 ? 6 : 2;                 // the loop is unrolled over a common period of stages and register sets
    static_assert(NG == 2 || NG == 3, "register sets");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_sc = reinterpret_cast<float*>(smem + 2 * STAGE);   // the a operand's activation of this block's CI channels
    float* s_sh = s_sc + CI;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int blk = (int)blockIdx.x, slice = blk / n_cb / (c_out / CO), cb = blk % n_cb, mb = (blk / n_cb) % (c_out / CO);
    const int64_t p_lo = (int64_t)slice * slice_pts, p_hi = p_lo + slice_pts < M ? p_lo + slice_pts : M;
    const int n_it = (int)((p_hi - p_lo) / 16);
    float s_in = 1.0f, s_out = 1.0f;                        // dz's power-of-two scale (see tr_linear_x3_kernel)
    if (dz_amax) {
        uint32_t mxb = dz_amax[lane];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const uint32_t other = (uint32_t)__shfl_xor((int)mxb, off);
            mxb = other > mxb ? other : mxb;
        }
        const int ex = (int)((__builtin_amdgcn_readfirstlane((int)mxb) >> 23) & 0xff);
        int sh_e = ex == 0 ? 0 : 14 - (ex - 127);
        sh_e = sh_e < -100 ? -100 : (sh_e > 100 ? 100 : sh_e);
        s_in = __int_as_float((127 + sh_e) << 23);
        s_out = __int_as_float((127 - sh_e) << 23);
    }
    const float floor_v = relu_in ? 0.0f : -3.0e38f;
    // this thread's pieces of a block: piece i of the dz block is 16-byte load number threadIdx.x + 256 i of its 16 x CO floats
    const float* ga[NA];
    int oa[NA];
    const float* gb[NB];
    int ob[NB], cb4[NB];
#pragma unroll
    for (int i = 0; i < NA; ++i) {
        const int v = threadIdx.x + NT * i, row = v / (CO / 4), c4 = v % (CO / 4);
        ga[i] = dz + (p_lo + row) * lddz + (int64_t)mb * CO + 4 * c4;
        oa[i] = wgx_off(row, 4 * c4);
    }
#pragma unroll
    for (int i = 0; i < NB; ++i) {
        const int v = threadIdx.x + NT * i, row = v / (CI / 4), c4 = v % (CI / 4);
        gb[i] = a + (p_lo + row) * lda + (int64_t)cb * CI + 4 * c4;
        ob[i] = wgx_off(row, 4 * c4);
        cb4[i] = 4 * c4;
    }
    if (scale) {
        for (int i = threadIdx.x; i < CI; i += NT) {
            s_sc[i] = scale[(int64_t)cb * CI + i];
            s_sh[i] = shift[(int64_t)cb * CI + i];
        }
    }
    struct Raw {
        f32x4 a[NA], b[NB];
    };
    Raw G[NG];
    const auto load = [&](Raw& g, int it) {
#pragma unroll
        for (int i = 0; i < NA; ++i) g.a[i] = *reinterpret_cast<const f32x4*>(ga[i] + (int64_t)it * 16 * lddz);
#pragma unroll
        for (int i = 0; i < NB; ++i) g.b[i] = *reinterpret_cast<const f32x4*>(gb[i] + (int64_t)it * 16 * lda);
    };
    const auto convert = [&](const Raw& g, char* st) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            f32x4 v = g.a[i];
            v[0] *= s_in;
            v[1] *= s_in;
            v[2] *= s_in;
            v[3] *= s_in;
            wgx_put(v, st, st + IMG_A, oa[i]);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            f32x4 v = g.b[i];
            if (scale) {
                const f32x4 sc4 = *reinterpret_cast<const f32x4*>(s_sc + cb4[i]), sh4 = *reinterpret_cast<const f32x4*>(s_sh + cb4[i]);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = __builtin_fmaxf(__builtin_fmaf(v[e], sc4[e], sh4[e]), floor_v);
            }
            wgx_put(v, st + 2 * IMG_A, st + 2 * IMG_A + IMG_B, ob[i]);
        }
    };
    f32x16 acc[MT][KT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int k = 0; k < KT; ++k) acc[m][k] = f32x16{};
    if (n_it > 0) {
#pragma unroll
        for (int i = 0; i < NG - 1; ++i) load(G[i], i < n_it ? i : n_it - 1);
        __syncthreads();                                        // (s_sc / s_sh)
        convert(G[0], smem);
        __syncthreads();
        for (int base = 0; base < n_it; base += PERIOD) {
#pragma unroll
            for (int u = 0; u < PERIOD; ++u) {
                const int it = base + u;
                if (it >= n_it) break;
                // (past the slice's end the last block is loaded and converted again, into a stage nobody reads: no branch
                // between the MFMAs and the conversion they are interleaved with)
                const int nx = it + NG - 1;
                load(G[(u + NG - 1) % NG], nx < n_it ? nx : n_it - 1);
                const char* st = smem + (u & 1) * STAGE;
                x3v8 ah[MT], al[MT];
#pragma unroll
                for (int m = 0; m < MT; ++m) {
                    ah[m] = wgx_operand(st, 32 * (wm * MT + m), lane);
                    al[m] = wgx_operand(st + IMG_A, 32 * (wm * MT + m), lane);
                }
#pragma unroll
                for (int k = 0; k < KT; ++k) {
                    const x3v8 bh = wgx_operand(st + 2 * IMG_A, 32 * (wn * KT + k), lane);
                    const x3v8 bl = wgx_operand(st + 2 * IMG_A + IMG_B, 32 * (wn * KT + k), lane);
#pragma unroll
                    for (int m = 0; m < MT; ++m) {
                        acc[m][k] = x3_mfma(ah[m], bh, acc[m][k]);
                        acc[m][k] = x3_mfma(ah[m], bl, acc[m][k]);
                        acc[m][k] = x3_mfma(al[m], bh, acc[m][k]);
                    }
                }
                convert(G[(u + 1) % NG], smem + ((u + 1) & 1) * STAGE);
                // the next block's conversion (6 VALU + two 8-byte LDS writes per 16 bytes loaded) rides under this block's MFMAs
#pragma unroll
                for (int n = 0; n < 3 * MT * KT; ++n) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, WGX_VPG, 0);
                    __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                }
                __syncthreads();
            }
        }
    }
    // part[slice][co][ci]: register r of half h is row 8 (r >> 2) + 4 h + (r & 3) of the tile, the lane its column
    const int h = lane >> 5, col = lane & 31;
    float* out = part + ((int64_t)slice * c_out + (int64_t)mb * CO + 32 * wm * MT) * c_in + (int64_t)cb * CI + 32 * wn * KT + col;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int k = 0; k < KT; ++k)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                out[(int64_t)(32 * m + 8 * (r >> 2) + 4 * h + (r & 3)) * c_in + 32 * k] = acc[m][k][r] * s_out;
}

// which instantiation a layer takes (0: none — the fp32 kernel): the block shape is the largest that divides the layer
static int wgx_shape(int c_out, int c_in) {
    if (c_out % 256 == 0 && c_in % 256 == 0) return 1;      // 256 x 256
    if (c_out % 128 == 0 && c_in % 256 == 0) return 2;      // 128 x 256
    if (c_out % 128 == 0 && c_in % 128 == 0) return 3;      // 128 x 128
    if (c_out % 512 == 0 && c_in % 64 == 0) return 4;       // 512 x 64
    return 0;
}
static void wgx_dims(int shape, int* co, int* ci) {
    static const int d[5][2] = {{0, 0}, {256, 256}, {128, 256}, {128, 128}, {512, 64}};
    *co = d[shape][0];
    *ci = d[shape][1];
}
static int64_t wgx_slice_pts(int64_t M, int c_out, int c_in, int shape) {
    int co, ci;
    wgx_dims(shape, &co, &ci);
    const int64_t blocks = (int64_t)(c_out / co) * (c_in / ci);
    int64_t want = (2 * trx_cu_count() + blocks - 1) / blocks;   // ~two workgroups per CU in all
    if (want < 1) want = 1;
    int64_t pts = (M + want - 1) / want;
    pts = (pts + 95) / 96 * 96;                              // (whole 16-point blocks; a multiple of the loop's unroll)
    return pts < 96 ? 96 : pts;
}
bool tr_wgrad_x3_ok(int64_t M, int c_out, int c_in) {
    return M >= 8192 && M % 16 == 0 && M < ((int64_t)1 << 31) && wgx_shape(c_out, c_in) != 0;
}
size_t tr_wgrad_x3_workspace_bytes(int64_t M, int c_out, int c_in) {
    const int shape = wgx_shape(c_out, c_in);
    if (!shape) return 0;
    const int64_t pts = wgx_slice_pts(M, c_out, c_in, shape);
    return (size_t)((M + pts - 1) / pts) * c_out * c_in * sizeof(float);
}
hipError_t launch_tr_wgrad_final(const float* part, int n_slices, int64_t n, float* dW, hipStream_t s);
hipError_t launch_tr_wgrad_x3(const float* dz, int64_t lddz, const float* a, int64_t lda, const float* scale, const float* shift,
                              int relu_in, const uint32_t* dz_amax, int64_t M, int c_out, int c_in, float* part, float* dW,
                              hipStream_t s) {
    const int shape = wgx_shape(c_out, c_in);
    int co, ci;
    wgx_dims(shape, &co, &ci);
    const int64_t pts = wgx_slice_pts(M, c_out, c_in, shape);
    const int n_slices = (int)((M + pts - 1) / pts), n_cb = c_in / ci, n_mb = c_out / co;
    const unsigned grid = (unsigned)(n_slices * n_cb * n_mb);
    const auto go = [&](auto kern, size_t lds, unsigned threads = 256) -> hipError_t {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, s, dz, lddz, a, lda, scale, shift, relu_in, dz_amax, M, pts, n_cb,
                           c_out, c_in, part);
        return hipGetLastError();
    };
    const size_t lds = 2 * (2 * (size_t)wgx_image_bytes(co) + 2 * (size_t)wgx_image_bytes(ci)) + 2 * (size_t)ci * sizeof(float);
    hipError_t e = hipSuccess;
    switch (shape) {
        case 1: e = go(tr_wgrad_x3_kernel<2, 4, 4, 2>, lds, 512); break;     // 256 x 256: eight waves of 64 x 128
#ifdef TRX_WGRAD_S2_4WAVES
        case 2: e = go(tr_wgrad_x3_kernel<2, 4, 2, 2>, lds); break;
#else
        case 2: e = go(tr_wgrad_x3_kernel<2, 2, 2, 4>, lds, 512); break;     // 128 x 256: eight waves of 64 x 64
#endif
        case 3: e = go(tr_wgrad_x3_kernel<2, 2, 2, 2>, lds); break;
        case 4: e = go(tr_wgrad_x3_kernel<4, 2, 4, 1>, lds); break;
        default: return hipErrorInvalidValue;
    }
    if (e != hipSuccess || !dW) return e;                       // (dW NULL: the caller adds the slices later)
    return launch_tr_wgrad_final(part, n_slices, (int64_t)c_out * c_in, dW, s);
}
