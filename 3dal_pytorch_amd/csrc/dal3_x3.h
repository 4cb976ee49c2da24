// dal3_x3.h — the "f16x3" engine shared by the eval kernels (dal3_pointmlp_x3.hip) and the training forward
// (dal3_train_x3.hip): fp16 MFMAs on (hi, lo) split operands with fp32 accumulation. See dal3_pointmlp_x3.hip's header.
#pragma once
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
#ifndef DAL3_X3_DEPTH
#define DAL3_X3_DEPTH 1                 // fragment pairs in registers ahead of the one in use (A/B: 2)
#endif
// DRAIN: open() waits for the wave's VMEM traffic except its KEEP youngest operations (LdsRing::acquire_wait_all / _keep)
template <bool DRAIN = false, int SLOTS = 3, int DEPTH = DAL3_X3_DEPTH, int KEEP = 0>
struct X3StreamT {
    typedef LdsRing<X3_SEG, SLOTS> Ring;
    static constexpr int START = 2 * DEPTH;                // the cursor at the start of a group of points
    x3v8 n2h, n2l;                                         // (DEPTH == 2: the pair after (nh, nl))
    Ring ring;
    int cur, pending;
    int keep = KEEP;
    x3v8 nh, nl;
    __device__ __forceinline__ void pump1() {
        if (pending > 0) {
            ring.issue_part(Ring::MY_LOADS - pending);
            if (--pending == 0) ring.issue_done();
        }
    }
    __device__ __forceinline__ void pump() {               // (two slots: the refill has one segment to land — out in four k-steps)
        pump1();
        if (SLOTS == 2) pump1();
    }
    __device__ __forceinline__ void flush() {
        while (pending > 0) pump1();
    }
    __device__ __forceinline__ void open() {
        flush();
        if (DRAIN && KEEP > 0) {
            // `keep` (a constant after unrolling): how many VMEM operations the caller has issued behind the last thing
            // this open must wait for — they stay in flight. Too small only waits longer; too large would be a race.
            if (keep == KEEP)
                ring.template acquire_wait_keep<KEEP>();
            else if (keep == KEEP + 7)
                ring.template acquire_wait_keep<KEEP + 7>();
            else if (keep == KEEP + 28)
                ring.template acquire_wait_keep<KEEP + 28>();
            else
                ring.acquire_wait_all();
        } else if (DRAIN)
            ring.acquire_wait_all();
        else
            ring.acquire_wait();
        pending = Ring::MY_LOADS;
        cur = 0;
    }
    __device__ __forceinline__ void fetch() {
        if (DEPTH == 2) {
            nh = n2h;
            nl = n2l;
            n2h = ring.template frag<FP16>(cur);
            n2l = ring.template frag<FP16>(cur + 1);
        } else {
#ifdef DAL3_X3_ABL_HALFREADS                               // timing experiment only: every other pair is not read (what LDS weight reads cost)
            if (cur & 2) {
                cur += 2;
                return;
            }
#endif
            nh = ring.template frag<FP16>(cur);
            nl = ring.template frag<FP16>(cur + 1);
        }
        cur += 2;
    }
    // state at the start of every group of points: first pair in (nh, nl), cur == 2, nothing pending
    __device__ __forceinline__ void init(const void* stream, char* lds, int n_segs, int wave, int lane) {
        ring.init(stream, lds, n_segs, wave, lane, true);  // segments 0 and (three slots) 1 in flight
        pending = 0;
        if (DRAIN) {                                       // (nothing of the caller's is in flight yet: a full drain)
            ring.acquire_wait_all();
            pending = Ring::MY_LOADS;
            cur = 0;
        } else {
            open();
        }
        flush();
        fetch();
        if (DEPTH == 2) fetch();
    }
    __device__ __forceinline__ void take(x3v8& wh, x3v8& wl) {
        wh = nh;
        wl = nl;
        fetch();
    }
    // end of a group of points: a stream of whole segments needs nothing (the last take() fetched the next group's first
    // pair); otherwise what is left of the open segment is padding
    __device__ __forceinline__ void end_group() {
        if (cur != START) {
            open();
            fetch();
            if (DEPTH == 2) fetch();
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

typedef X3StreamT<> X3Stream;

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
template <int KT, int T, bool SWAP, bool ZERO, int VPG, bool VG = false, class St, class Side, class Hook = X3NoSide>
__device__ __forceinline__ void x3_block(St& st, const X3Tile (&X)[T][KT], f32x16 (&acc)[T], Side&& side, Hook&& opened = Hook()) {
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
        if (st.cur == X3_SEG) {
            st.open();
            opened(s);                                     // (right behind the barrier: the k-step that follows is the segment's last)
        }
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

