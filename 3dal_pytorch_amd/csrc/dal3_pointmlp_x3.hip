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
// ONE generic block (x3_block) walks the fragment stream with a cursor that is a compile-time constant everywhere; the
// next k-step's pair is always in registers, a segment is opened a k-step before it is needed, the refill's LDS-DMA
// instructions go out one per k-step, and the VALU work between layers (bias, ReLU, the (hi, lo) split; the max
// epilogues) is dealt out under the NEXT block's MFMAs.
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
    int hi[2][4], lo[2][4];             // (32-bit pieces, written one split unit at a time)
};
__device__ __forceinline__ x3v8 x3_operand(const int (&p)[4]) {
    const int4_t v = {p[0], p[1], p[2], p[3]};
    return __builtin_bit_cast(x3v8, v);
}

__device__ __forceinline__ f32x16 x3_mfma(x3v8 a, x3v8 b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// One UNIT of a tile's split (q = 0..7: registers 8 (q >> 2) + 2 (q & 3), + 1 of the accumulator): bias, relu, then
// hi = fp16(x), lo = fp16(x - hi) for the pair -> one 32-bit element of the tile's hi and lo operands (register pair i of
// k-step s holds channels 16 s + 8 (i >> 1) ... as in pack_relu of dal3_lp.h: the A operands' k order is permuted to
// match by the packer). Ten issue slots (2 accumulator reads, pk_add, 2 max, cvt_pk, 2 cvt, pk_add, cvt_pk): a tile is
// 80, i.e. as long as 10 of its 6 KT MFMAs — which is why the units are dealt out under the NEXT block's MFMAs
// (x3_layer) instead of standing between two blocks with the matrix pipe idle.
template <bool BIAS>
__device__ __forceinline__ void x3_split_unit(const f32x16& acc, const f32x16& bv, X3Tile& t, int q) {
    const int s = q >> 2, i = q & 3, r = 8 * s + 2 * i;
#ifdef DAL3_X3_ABL_SPLIT                                   // timing experiment only: what the split's VALU work costs
    t.hi[s][i] = __float_as_int(acc[r]);
    t.lo[s][i] = __float_as_int(acc[r + 1]);
    return;
#endif
    f32x2 p = {acc[r], acc[r + 1]};
    if (BIAS) {
        const f32x2 b = {bv[r], bv[r + 1]};
        p = p + b;
    }
    p[0] = relu1(p[0]);
    p[1] = relu1(p[1]);
    const f16x2_t hh = __builtin_convertvector(p, f16x2_t);
    const int hi = __builtin_bit_cast(int, hh);
    // lo = fp16(x - float(hi)) straight from the packed hi and the two fp32 values: v_fma_mixlo/mixhi_f16 read an fp16
    // half as a source of an fp32 fma and round the result to fp16 (the fma is exact, one rounding) — two instructions
    // instead of two conversions back, a subtraction and a packing conversion. (hipcc does not select them from C.)
    int lo;
    asm("v_fma_mixlo_f16 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]\n\tv_fma_mixhi_f16 %0, %1, -1.0, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]"
        : "=&v"(lo)
        : "v"(hi), "v"(p[0]), "v"(p[1]));
    t.hi[s][i] = hi;
    t.lo[s][i] = lo;
}
// a whole tile at once (the fp32 first layer's output: bias already in the accumulator)
__device__ __forceinline__ X3Tile x3_split_relu(const f32x16& acc) {
    X3Tile t;
#pragma unroll
    for (int q = 0; q < 8; ++q) x3_split_unit<false>(acc, acc, t, q);
    return t;
}
// units [step PER, (step + 1) PER) of the split of T accumulator tiles a[.] into Y[.][m] (unit u: tile u / 8, piece u % 8)
template <int PER, int T, int MT>
__device__ __forceinline__ void x3_split_steps(const f32x16 (&a)[T], const f32x16& bv, X3Tile (&Y)[T][MT], int m, int step) {
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int u = step * PER + k;
        if (u >= 0 && u < 8 * T) x3_split_unit<true>(a[u >> 3], bv, Y[u >> 3][m], u & 7);
    }
}

// The weight stream of one kernel, walked by a wave-uniform cursor that is a compile-time constant everywhere (all loops
// over the stream are unrolled; a runtime loop's body consumes whole segments and re-pins the cursor at its top, pin()).
// Stream order = consumption order; a k-step is a (hi, lo) fragment pair. The pair of the NEXT k-step is always in
// registers already (nh, nl): take() hands it out and requests the one after it, so its LDS round trip runs under the
// k-step's own 3 T MFMAs — across blocks, layers and groups of points (the stream is cyclic, a group ends where the
// next one starts). Behind the k-step's MFMAs (x3_block): one instruction of the pending refill (pump()), and, when
// the pair just requested was the last of its segment, the next segment is opened right there — counted wait + barrier
// (LdsRing::acquire_wait) with the matrix pipe still busy, a k-step before anything of that segment is needed. The
// refill of the slot this frees is then PENDING: its LDS-DMA instructions go out one per k-step (the third slot gives it
// a further segment to land); all of them are out before the next acquire_wait, whose counted vmcnt relies on it.
struct X3Stream {
    typedef LdsRing<X3_SEG, 3> Ring;
    Ring ring;
    int cur, pending;
    x3v8 nh, nl;
    __device__ __forceinline__ void pump() {
        if (pending > 0) {
            ring.issue_part(Ring::MY_LOADS - pending);
            if (--pending == 0) ring.issue_done();
        }
    }
    __device__ __forceinline__ void flush() {
        while (pending > 0) pump();
    }
    __device__ __forceinline__ void open() {
        flush();
        ring.acquire_wait();
        pending = Ring::MY_LOADS;
        cur = 0;
    }
    __device__ __forceinline__ void fetch() {
        nh = ring.template frag<FP16>(cur);
        nl = ring.template frag<FP16>(cur + 1);
        cur += 2;
    }
    // state at the start of every group of points: first pair in (nh, nl), cur == 2, nothing pending
    __device__ __forceinline__ void init(const void* stream, char* lds, int n_segs, int wave, int lane) {
        ring.init(stream, lds, n_segs, wave, lane, true);  // segments 0 and 1 in flight
        pending = 0;
        open();
        flush();
        fetch();
    }
    __device__ __forceinline__ void take(x3v8& wh, x3v8& wl) {
        wh = nh;
        wl = nl;
        fetch();
    }
    // end of a group of points: a stream of whole segments needs nothing (the last take() fetched the next group's first
    // pair); otherwise what is left of the open segment is padding
    __device__ __forceinline__ void end_group() {
        if (cur != 2) {
            open();
            fetch();
        }
        flush();
    }
    // top of a runtime loop whose body leaves the cursor where it found it: tell the compiler
    __device__ __forceinline__ void pin(int c, int p) {
#ifdef DAL3_X3_CHECK
        if (cur != c || pending != p) __builtin_trap();
#endif
        cur = c;
        pending = p;
    }
};

struct X3NoSide {
    __device__ __forceinline__ void operator()(int) const {}
};

// The same MFMA with its accumulator in ARCHITECTURAL VGPRs, written as inline asm: hipcc selects one form per function
// (accumulators in AccVGPRs as soon as a kernel needs more than fit the VGPR file) and the decoder's resident dconv2
// accumulators alone are all 256 AccVGPRs — 32 more for the dconv1 chunk were spilled to scratch, and every reload
// waited with vmcnt(0) for the weight ring's loads in flight. Hazards the compiler does not see inside asm, by hand:
// a chain alternates its T >= 2 accumulators (exact-overlap SrcC, an independent MFMA in between); the VALU that reads
// the result comes behind x3_mfma_v_settle() (8-pass MFMA: 11 wait states); operands written by VALU / LDS are tracked
// by the compiler's own waitcnt and hazard passes through the asm's register operands.
template <bool ZEROC>
__device__ __forceinline__ void x3_mfma_v(f32x16& acc, x3v8 a, x3v8 b) {
    if (ZEROC)
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
    else
        asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
}
template <int T>
__device__ __forceinline__ void x3_mfma_v_settle(f32x16 (&acc)[T]) {
    static_assert(T == 2, "one asm statement naming every accumulator");
    asm volatile("s_nop 7\n\ts_nop 7" : "+v"(acc[0]), "+v"(acc[1]));
}

// acc[j] (+)= W'(32 x 32 KT) . X[j] for the wave's T point tiles: 2 KT k-steps, three MFMAs per k-step and tile
// (hi hi, hi lo, lo hi; tiles innermost, so consecutive MFMAs never share an accumulator). ZERO: the first MFMA of every
// tile takes the constant 0 as C — accumulators are never initialised, biases are added by the split (x3_split_unit).
// SWAP: operands exchanged, acc[j] = X[j]^T . W'^T — the transposed tile of the max-pooled layers (points on the
// accumulator's registers, channels on its lanes: the max over points is a max over registers). VG: accumulators in
// VGPRs (x3_mfma_v).
// side(s): work for the shadow of k-step s's MFMAs (a slice of the previous tile's split, of a max epilogue ...), dealt
// out between them VPG VALU instructions per MFMA (sched_group_barrier); s is a constant after unrolling.
template <int KT, int T, bool SWAP, bool ZERO, int VPG, bool VG = false, class Side>
__device__ __forceinline__ void x3_block(X3Stream& st, const X3Tile (&X)[T][KT], f32x16 (&acc)[T], Side&& side) {
    static_assert(!(VG && SWAP) && !(VG && T < 2), "x3_mfma_v");
#pragma unroll
    for (int s = 0; s < 2 * KT; ++s) {
        x3v8 wh, wl;
        st.take(wh, wl);
        DAL3_SCHED_FENCE();
#pragma unroll
        for (int r = 0; r < 3; ++r) {
#pragma unroll
            for (int j = 0; j < T; ++j) {
                const x3v8 xh = x3_operand(X[j][s >> 1].hi[s & 1]), xl = x3_operand(X[j][s >> 1].lo[s & 1]);
                const x3v8 a = r == 2 ? wl : wh, b = r == 1 ? xl : xh;
                if (VG) {
                    if (ZERO && s == 0 && r == 0) x3_mfma_v<true>(acc[j], a, b);
                    else x3_mfma_v<false>(acc[j], a, b);
                } else {
                    const f32x16 c = (ZERO && s == 0 && r == 0) ? f32x16{} : acc[j];
                    acc[j] = SWAP ? x3_mfma(b, a, c) : x3_mfma(a, b, c);
                }
            }
        }
        if (VG && s == 2 * KT - 1) x3_mfma_v_settle(acc);
        side(s);
        st.pump();
        if (VPG > 0) {
#pragma unroll
            for (int n = 0; n < 3 * T; ++n) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);         // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, VPG, 0);       // up to VPG VALU
            }
        }
        DAL3_SCHED_FENCE();
        if (st.cur == X3_SEG) st.open();
    }
}

// what a layer leaves behind: its LAST out-tile still as accumulators, with that tile's bias — whoever consumes the
// layer's output splits it into its own X[.][KT - 1] under the first k-steps of its first block (x3_carry_steps)
template <int T>
struct X3Carry {
    f32x16 acc[T];
    f32x16 bv;
};
// VALU slots per MFMA for a side of `per` split units per k-step
__host__ __device__ constexpr int x3_vpg(int per, int t) { return (per * 7 + 3 * t - 1) / (3 * t); }
// units per k-step so that 8 T units are done within the first 2 KT - 2 k-steps (the last two read X[.][KT - 1])
__host__ __device__ constexpr int x3_carry_per(int kt, int t) { return (8 * t + 2 * kt - 3) / (2 * kt - 2); }
template <int KT, int T>
__device__ __forceinline__ void x3_carry_steps(const X3Carry<T>& c, X3Tile (&X)[T][KT], int s) {
    static_assert(KT >= 2, "the carried tile is the block's last k-tile");
    x3_split_steps<x3_carry_per(KT, T)>(c.acc, c.bv, X, KT - 1, s);
}

// Y = split(relu(W' X + b')) for a 32 KT -> 32 MT layer; bias: LDS pointer to the layer's folded bias. Two accumulator
// sets: out-tile m - 1 is split under the MFMAs of out-tile m (PER units per k-step), its bias vector read from LDS a
// k-step before the first unit needs it. The last out-tile is left in `carry`. first(s): the side of the first block
// (the previous layer's carry, normally), VPG0 its VALU slots per MFMA.
template <int KT, int MT, int T, int VPG0, class First>
__device__ __forceinline__ void x3_layer(X3Stream& st, const float* bias, const X3Tile (&X)[T][KT], X3Tile (&Y)[T][MT], int h,
                                         X3Carry<T>& carry, First&& first) {
    constexpr int PER = (8 * T + 2 * KT - 1) / (2 * KT);
    f32x16 acc[2][T];
    f32x16 bv;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
        if (m == 0) {
            x3_block<KT, T, false, true, VPG0>(st, X, acc[0], [&](int s) {
                first(s);
                if (s == 2 * KT - 1) bv = tile_from_channels(bias, h);
            });
        } else {
            x3_block<KT, T, false, true, x3_vpg(PER, T)>(st, X, acc[m & 1], [&](int s) {
                x3_split_steps<PER>(acc[(m - 1) & 1], bv, Y, m - 1, s);
                if (s == 2 * KT - 1) bv = tile_from_channels(bias + 32 * m, h);
            });
        }
    }
#pragma unroll
    for (int j = 0; j < T; ++j) carry.acc[j] = acc[(MT - 1) & 1][j];
    carry.bv = bv;
}

// slice s of NS of a transposed tile's max epilogue (lp_tile_max_t of dal3_lp.h: max over the wave's 32 T points = over
// registers, bias added after the max, ReLU on the bit pattern, LDS integer atomicMax), mx carried between the slices
template <int NS, int T>
__device__ __forceinline__ void x3_max_step(const f32x16 (&a)[T], float& mx, int s, const float* bias, int* smax, int lane) {
    constexpr int R = (16 * T + NS - 1) / NS;
#pragma unroll
    for (int k = 0; k < R; ++k) {
        const int u = s * R + k;
        if (u < 16 * T) mx = u == 0 ? a[0][0] : __builtin_fmaxf(mx, a[u >> 4][u & 15]);
    }
    if (s == NS - 1) {
        const int ch = lane & 31;
        int bits = __float_as_int(mx + bias[ch]);
        bits = bits > 0 ? bits : 0;
        atomicMax(smax + ch, bits);
    }
}

// the max-pooled last layer: n_tiles (even) out-tiles computed transposed, each tile's maxima joined into the
// workgroup's LDS array. Two accumulator sets: a tile's epilogue runs under the next tile's MFMAs; tile 0 runs with the
// previous layer's carry as its side, the last tile's epilogue stands alone. The loop body is two tiles = a whole
// number of segments.
template <int KT, int T>
__device__ __forceinline__ void x3_max_layer(X3Stream& st, const float* bias, X3Tile (&X)[T][KT], int* smax, int n_tiles,
                                             int lane, const X3Carry<T>& carry) {
    static_assert((2 * KT * 4) % X3_SEG == 0, "two tiles = whole segments");
    constexpr int NS = 2 * KT, VPG = ((16 * T + NS - 1) / NS * 2 + 6 + 3 * T - 1) / (3 * T);
    f32x16 acc[2][T];
    float mx = 0.0f;
    x3_block<KT, T, true, true, x3_vpg(x3_carry_per(KT, T), T)>(st, X, acc[0], [&](int s) { x3_carry_steps<KT, T>(carry, X, s); });
    const int c0 = st.cur, p0 = st.pending;
    for (int m = 1; m + 1 < n_tiles; m += 2) {
        st.pin(c0, p0);
        x3_block<KT, T, true, true, VPG>(st, X, acc[1], [&](int s) { x3_max_step<NS>(acc[0], mx, s, bias + 32 * (m - 1), smax + 32 * (m - 1), lane); });
        x3_block<KT, T, true, true, VPG>(st, X, acc[0], [&](int s) { x3_max_step<NS>(acc[1], mx, s, bias + 32 * m, smax + 32 * m, lane); });
    }
    const int m = n_tiles - 1;
    x3_block<KT, T, true, true, VPG>(st, X, acc[1], [&](int s) { x3_max_step<NS>(acc[0], mx, s, bias + 32 * (m - 1), smax + 32 * (m - 1), lane); });
#pragma unroll
    for (int s = 0; s < NS; ++s) x3_max_step<NS>(acc[1], mx, s, bias + 32 * m, smax + 32 * m, lane);
}

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
        st.pin(2, 0);
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
        st.pin(2, 0);
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
    prefetch(blockIdx.x);
    for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
        st.pin(2, 0);
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
