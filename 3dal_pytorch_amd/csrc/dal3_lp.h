// dal3_lp.h — building blocks of the bf16 / fp16 MFMA path (v_mfma_f32_32x32x16_{bf16,f16}, fp32
// accumulate). Same orientation as the fp32 path (channels on MFMA rows, points on columns): the
// fp32 accumulator of layer k, ReLU'd and rounded to 16 bits pairwise, IS the B operand of layer k+1
// (cdna guide "An accumulator tile as the next MFMA's operand": registers 8s..8s+7 are the fragment
// of k-step s, element j of lane half h being channel 16s + 8(j>>2) + 4h + (j&3); the weight packer
// permutes the A operand's k order to match), so activations never leave registers.
//
// Weights: at 32 cycles per MFMA a wave would need 1 KiB of A operand every 32*T cycles — far more
// than per-wave L2 streams can deliver — so the four waves of a workgroup share them through LDS.
// Each kernel's fragments form one stream in consumption order, cut into fixed segments of SEG
// fragments (1 KiB each); a 3-slot LDS ring is filled by LDS-DMA (global_load_lds, 16 B per lane,
// lane-linear = exactly one fragment per wave-instruction) two segments ahead of the MFMAs. Per
// segment: every wave waits for its own DMA share with a COUNTED vmcnt (the next segment's loads stay
// in flight), one raw s_barrier publishes the slot and frees the oldest one, the next fill is issued,
// and the fragments are read back with ds_read_b128. Ordinary global loads are kept out of the loops
// (biases and the per-crop dconv1 term are staged in LDS once) because hipcc drains vmcnt(0) for them.
#pragma once
#include "dal3_device.h"

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));

struct BF16 {
    typedef bf16x8_t v8;
    typedef bf16x2_t v2;
    typedef __bf16 elem;
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    }
};
struct FP16 {
    typedef f16x8_t v8;
    typedef f16x2_t v2;
    typedef _Float16 elem;
    static __device__ __forceinline__ f32x16 mfma(v8 a, v8 b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    }
};

// The four k-steps of a 64-channel -> 32-channel block for TWO point tiles with the accumulators in ARCH VGPRs, as one
// asm statement. hipcc picks the accumulator register class per FUNCTION: the decode kernel's eight resident dconv2
// tiles fill the AGPR half exactly (2 x 8 x 16), so with AGPR-form MFMAs everywhere the short-lived dconv1 chunk
// accumulators overflow it and the compiler shuttles 64 registers between the two halves in every iteration (PMC:
// 5.3 VALU instructions per MFMA in the main loop, 320 of 420 of them v_accvgpr_read/write). Written out, these eight
// MFMAs take VGPR C/D — legal for any single instruction — and their results are read by the pack/ReLU directly.
// Wait states are inside the string (cdna guide 5.7 item 2: hipcc pads nothing inside asm): `s_nop 1` covers a VALU
// write of an operand just before, `s_nop 11` the 8-pass XDL write -> any reader after the statement.
template <class DT>
struct MfmaAsm;
// acc0 comes in holding the INITIAL value of both tiles (the crop's dconv1 term: the same for every point of a crop, so it
// is read from LDS once, not once per tile); tile 1's first MFMA takes it as its C operand and goes first, before tile
// 0's first MFMA overwrites it. Per tile the k order is unchanged.
#define DAL3_LP_MFMA4X2(MN)                                                                                          \
    asm volatile("s_nop 1\n\t" MN " %1, %2, %10, %0\n\t" MN " %0, %2, %6, %0\n\t" MN " %0, %3, %7, %0\n\t" MN           \
                 " %1, %3, %11, %1\n\t" MN " %0, %4, %8, %0\n\t" MN " %1, %4, %12, %1\n\t" MN " %0, %5, %9, %0\n\t" MN  \
                 " %1, %5, %13, %1\n\ts_nop 11"                                                                       \
                 : "+v"(acc0), "=&v"(acc1)                                                                           \
                 : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b00), "v"(b01), "v"(b02), "v"(b03), "v"(b10), "v"(b11),      \
                   "v"(b12), "v"(b13))
template <>
struct MfmaAsm<BF16> {
    typedef BF16::v8 v8;
    static __device__ __forceinline__ void block4x2(f32x16& acc0, f32x16& acc1, v8 a0, v8 a1, v8 a2, v8 a3, v8 b00, v8 b01,
                                                    v8 b02, v8 b03, v8 b10, v8 b11, v8 b12, v8 b13) {
        DAL3_LP_MFMA4X2("v_mfma_f32_32x32x16_bf16");
    }
};
template <>
struct MfmaAsm<FP16> {
    typedef FP16::v8 v8;
    static __device__ __forceinline__ void block4x2(f32x16& acc0, f32x16& acc1, v8 a0, v8 a1, v8 a2, v8 a3, v8 b00, v8 b01,
                                                    v8 b02, v8 b03, v8 b10, v8 b11, v8 b12, v8 b13) {
        DAL3_LP_MFMA4X2("v_mfma_f32_32x32x16_f16");
    }
};

// B-operand fragments of one 32-channel x 32-point activation tile (two k-steps of 16 channels)
template <class DT>
struct ActTile {
    typename DT::v8 k[2];
};

// round to 16 bits first, then ReLU on the packed pairs with ONE integer v_pk_max_i16 per dword (the sign bit
// of a bf16 / fp16 is the sign bit of its int16 pattern; rounding never changes the sign, so
// relu(round(x)) == round(relu(x))): 8 cvt_pk + 8 pk_max per tile instead of 16 max + 8 cvt_pk
// (written pair by pair: converting the whole vector and then taking the max makes hipcc split and re-merge
// the halves with v_perm_b32, 16 instructions instead of 8)
typedef short short2_t __attribute__((ext_vector_type(2)));
typedef int int4_t __attribute__((ext_vector_type(4)));
template <class DT>
__device__ __forceinline__ ActTile<DT> pack_relu(const f32x16& acc) {
    ActTile<DT> t;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        int4_t w;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x2 p = {acc[8 * s + 2 * i], acc[8 * s + 2 * i + 1]};
            const typename DT::v2 q = __builtin_convertvector(p, typename DT::v2);
            short2_t b = __builtin_bit_cast(short2_t, q);
            const short2_t zero = {0, 0};
            b = __builtin_elementwise_max(b, zero);
            w[i] = __builtin_bit_cast(int, b);
        }
        t.k[s] = __builtin_bit_cast(typename DT::v8, w);
    }
    return t;
}

// one dword of pack_relu: two accumulator registers -> a ReLU'd 16-bit pair
template <class DT>
__device__ __forceinline__ int pack_relu_pair(float a, float b) {
    const f32x2 p = {a, b};
    const typename DT::v2 q = __builtin_convertvector(p, typename DT::v2);
    short2_t s = __builtin_bit_cast(short2_t, q);
    const short2_t zero = {0, 0};
    s = __builtin_elementwise_max(s, zero);
    return __builtin_bit_cast(int, s);
}

#define LP_SLOTS 3

// One LDS-DMA instruction: 16 B per lane from gbase + lane_off (uniform 64-bit base, 32-bit per-lane offset) to the
// LDS address dst + 16 * lane. Written as inline asm on purpose: with __builtin_amdgcn_global_load_lds anywhere in
// a loop hipcc's waitcnt pass degrades EVERY wait on an LDS read in that loop to `s_waitcnt lgkmcnt(0)` (LDS written
// by a VMEM instruction: it gives up counting), so each group of fragment reads issued a group ahead was waited for
// right away, together with the group actually needed — the LDS latency the read-ahead was there to hide. Hidden in
// asm, the compiler counts LDS reads exactly again (lgkmcnt(4) etc.). Its own vmcnt waits stay safe: loads it does
// not know of only make the real count higher than its model, and loads return in order. m0 is not used otherwise
// in these kernels (it cannot be declared as a clobber: reserved).
__device__ __forceinline__ void lds_dma16(const char* gbase, uint32_t lane_off, char* dst) {
    const uint32_t lds_addr = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) char*)dst;
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(lane_off), "s"(gbase), "s"(lds_addr) : "memory");
}

// LDS ring of weight segments. All state is wave-uniform.
template <int SEG, int SLOTS = LP_SLOTS>
struct LdsRing {
    static_assert(SLOTS == 2 || SLOTS == 3, "SLOTS - 1 segments are in flight ahead of the one in use");
    static constexpr int N_SLOTS = SLOTS;
    static constexpr int SLOT_BYTES = SEG * 1024;
    static constexpr int MY_LOADS = SEG / 4;              // LDS-DMA instructions per wave per segment (4 waves)
    static_assert(SEG % 4 == 0, "segment must split evenly over the 4 waves");
    const char* stream;                                   // global: this kernel's fragment stream
    char* lds;                                            // LDS: SLOTS * SLOT_BYTES
    int n_segs, seg_issue, slot_issue, slot_use, wave, lane;
#ifdef DAL3_STAMP
    long long bar_ticks = 0, wait_ticks = 0;              // diagnostic: time spent in the counted wait / in s_barrier
#endif
    bool cyclic;                                          // persistent kernels: past the stream's end comes its start

    __device__ __forceinline__ void init(const void* stream_, char* lds_, int n_segs_, int wave_, int lane_,
                                         bool cyclic_ = false) {
        stream = static_cast<const char*>(stream_);
        lds = lds_;
        n_segs = n_segs_;
        cyclic = cyclic_;
        wave = __builtin_amdgcn_readfirstlane(wave_);
        lane = lane_;
        seg_issue = 0;
        slot_issue = 0;
        slot_use = SLOTS - 1;
        issue();                                          // segments 0 and (three slots) 1 in flight before the first acquire
        if (SLOTS == 3) issue();
    }
    // this wave's quarter of the next segment -> slot_issue (past the stream's end: re-read the last one)
    __device__ __forceinline__ void issue() {
        const int seg = seg_issue < n_segs ? seg_issue : n_segs - 1;
        // uniform 64-bit base + unsigned 32-bit lane offset: the scalar-base form of the load (one VGPR of address per
        // lane instead of a 64-bit pair that has to be built, kept or spilled)
        const char* src = stream + (size_t)seg * SLOT_BYTES;
        char* dst = lds + slot_issue * SLOT_BYTES;
#pragma unroll
        for (int k = 0; k < MY_LOADS; ++k) {
            const int f = wave + 4 * k;
            lds_dma16(src + f * 1024, (uint32_t)(lane * 16), dst + f * 1024);
        }
        ++seg_issue;
        if (cyclic && seg_issue == n_segs) seg_issue = 0;
        slot_issue = slot_issue + 1 == SLOTS ? 0 : slot_issue + 1;
    }
    // make the next segment readable: my share landed (the following segment's MY_LOADS loads may stay in
    // flight), everyone's share landed and everyone is done with the slot about to be refilled (barrier)
    __device__ __forceinline__ void acquire() {
        acquire_wait();
        issue();
        __builtin_amdgcn_sched_barrier(0);
    }
    // The two halves of acquire() for callers that hide the refill's instructions under MFMAs: acquire_wait() as soon as
    // every fragment of the current segment is in registers (it also waits for this wave's outstanding LDS reads:
    // the barrier tells the others the slot may be overwritten), then issue_part(0..MY_LOADS-1) one at a time
    // between MFMAs, then issue_done().
    __device__ __forceinline__ void acquire_wait() {
        __builtin_amdgcn_sched_barrier(0);
#ifdef DAL3_STAMP
        unsigned long long t0_, t1_, t2_;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0_)::"memory");
#endif
#ifndef DAL3_ABL_WAIT                                     // timing experiment only
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(SLOTS == 3 ? MY_LOADS : 0) : "memory");   // (two slots: no younger segment in flight)
#endif
#ifdef DAL3_STAMP
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1_)::"memory");
#endif
#ifndef DAL3_ABL_BAR                                      // timing experiment only
        __builtin_amdgcn_s_barrier();
#endif
#ifdef DAL3_STAMP
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t2_)::"memory");
        wait_ticks += (long long)(t1_ - t0_);
        bar_ticks += (long long)(t2_ - t1_);
#endif
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        slot_use = slot_use + 1 == SLOTS ? 0 : slot_use + 1;
    }
    // acquire_wait() for a kernel that also has compiler-visible VMEM traffic of its own (dal3_train_x3.hip): EVERYTHING
    // this wave has outstanding has landed or left (vmcnt(0)) before the barrier. hipcc cannot see the LDS-DMA loads
    // (inline asm), so its own vmcnt(N) in front of the first use of a loaded register is N too small by the number of
    // DMA loads issued after that load: it then waits for a load issued moments ago. With every such use placed right
    // behind this call its waits find the counter at zero.
    __device__ __forceinline__ void acquire_wait_all() {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        slot_use = slot_use + 1 == SLOTS ? 0 : slot_use + 1;
    }
    // ... and the same with the KEEP youngest operations left in flight (a request issued in front of this call that has
    // until the NEXT call to land; the caller's issue order makes them the youngest)
    template <int KEEP>
    __device__ __forceinline__ void acquire_wait_keep() {
        __builtin_amdgcn_sched_barrier(0);
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(KEEP) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        slot_use = slot_use + 1 == SLOTS ? 0 : slot_use + 1;
    }
    __device__ __forceinline__ void issue_part(int k) {
        const int seg = seg_issue < n_segs ? seg_issue : n_segs - 1;
        const int f = wave + 4 * k;
        const char* src = stream + (size_t)seg * SLOT_BYTES + f * 1024;
        char* dst = lds + slot_issue * SLOT_BYTES + f * 1024;
        lds_dma16(src, (uint32_t)(lane * 16), dst);
    }
    __device__ __forceinline__ void issue_done() {
        ++seg_issue;
        if (cyclic && seg_issue == n_segs) seg_issue = 0;
        slot_issue = slot_issue + 1 == SLOTS ? 0 : slot_issue + 1;
    }
    template <class DT>
    __device__ __forceinline__ typename DT::v8 frag(int f) const {
        return *reinterpret_cast<const typename DT::v8*>(lds + slot_use * SLOT_BYTES + f * 1024 + lane * 16);
    }
};

// acc[j] += W'(32 x 32*KT) . X[j] with the block's KT*2 fragments starting at fragment f0 of the slot; the
// fragments are read in groups of four, a group ahead of their MFMAs (see lp_layer)
// EARLY: the block's last four fragments are the last reads of the slot — the next segment is opened before their
// MFMAs (acquire_wait) and its refill dealt out between them; whoever uses the ring next must not acquire again.
template <int SEG, int GAPS, class Ring>
__device__ __forceinline__ void lp_refill_gap(Ring& ring, int n) {
    constexpr int ML = Ring::MY_LOADS;
    static_assert(ML <= 2 * GAPS, "two refill parts per MFMA gap at most");
    if (GAPS >= ML) {
        if (n < ML) ring.issue_part(n);
    } else {
        if (2 * n < ML) ring.issue_part(2 * n);
        if (2 * n + 1 < ML) ring.issue_part(2 * n + 1);
    }
}
// gap(n): work placed (between scheduling fences) after MFMA n = k T + j of the block — a few VALU instructions per
// gap ride in the shadow of the 32-cycle MFMAs (the previous tile's packing, say); n is a constant after unrolling.
struct LpNoGap {
    __device__ __forceinline__ void operator()(int) const {}
};
// On entry acc[0] holds the INITIAL value of every tile (the folded bias: one value per channel, the same for all the
// wave's points, so it is read from LDS once): the first k-step runs the tiles in reverse order, tiles 1.. take acc[0]
// as their C operand, tile 0 overwrites it last. Per tile the k order is what it always was.
// g: the caller's two fragment groups. CARRY_IN: g[0] already holds this block's first four fragments (read by the block
// before, under ITS MFMAs). F_NEXT >= 0: this block in turn reads the four fragments at F_NEXT for the block that follows —
// of the same slot a group ahead as usual, or (EARLY) of the slot it opens, in the first gap behind the barrier — so no
// block of a layer chain starts by waiting out an LDS round trip with the matrix pipe idle (stamps: 700-1,000 ticks
// in front of the first MFMA of dconv3 and of dconv4, ~200 in front of every other block).
template <class DT, int KT, int T, int SEG, bool EARLY = false, bool CARRY_IN = false, int F_NEXT = -1, class Gap = LpNoGap,
          class Ring>
__device__ __forceinline__ void lp_block(Ring& ring, int f0, const ActTile<DT> (&X)[T][KT], f32x16 (&acc)[T],
                                         typename DT::v8 (&g)[2][4], Gap gap = Gap()) {
    constexpr int NG = KT * 2 / 4;
    constexpr bool GAPS = !__is_same(Gap, LpNoGap);
    static_assert((KT * 2) % 4 == 0, "whole groups of four fragments");
    static_assert(NG % 2 == 0, "the carried group is g[0] on the way in and on the way out");
    if (!CARRY_IN) {
#pragma unroll
        for (int i = 0; i < 4; ++i) g[0][i] = ring.template frag<DT>(f0 + i);
    }
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
        const bool last = gi == NG - 1;
        if (!last) {
#pragma unroll
            for (int i = 0; i < 4; ++i) g[(gi + 1) & 1][i] = ring.template frag<DT>(f0 + 4 * (gi + 1) + i);
        } else if (F_NEXT >= 0 && !EARLY) {
#pragma unroll
            for (int i = 0; i < 4; ++i) g[0][i] = ring.template frag<DT>(F_NEXT + i);
        }
        DAL3_SCHED_FENCE();
        if (last && EARLY) ring.acquire_wait();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int k = 4 * gi + i;
#pragma unroll
            for (int jj = 0; jj < T; ++jj) {
                const int j = k == 0 ? T - 1 - jj : jj;     // (first k-step: tile 0 last, it overwrites the shared initial value)
                acc[j] = DT::mfma(g[gi & 1][i], X[j][k >> 1].k[k & 1], k == 0 ? acc[0] : acc[j]);
                if ((last && EARLY) || GAPS) {
                    DAL3_SCHED_FENCE();
                    if (last && EARLY) {
                        if (F_NEXT >= 0 && i == 0 && jj == 0) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) g[0][q] = ring.template frag<DT>(F_NEXT + q);
                        }
                        lp_refill_gap<SEG, 4 * T>(ring, i * T + jj);
                    }
                    gap(k * T + jj);
                    DAL3_SCHED_FENCE();
                }
            }
        }
        if (last && EARLY) ring.issue_done();
        DAL3_SCHED_FENCE();
    }
}

// Y[.][MT0 .. MT0+MTN) = relu(W' X + b') for the MTN out-tiles whose blocks start at fragment f0 of the
// current slot (bias: LDS pointer to the layer's folded bias); output rounded to 16 bits for the next layer.
// The MTN*KT*2 fragments are walked as ONE stream in groups of four, each group's ds_reads issued before the
// previous group's MFMAs (also across out-tile boundaries): left to itself hipcc puts every read right in front
// of its MFMAs and each fragment exposes the LDS latency (T = 2: only 64 MFMA cycles per fragment to hide it).
template <class DT, int KT, int MT, int T, int SEG, int MT0, int MTN, bool EARLY = false, class Ring>
__device__ __forceinline__ void lp_layer(Ring& ring, int f0, const float* bias,
                                         const ActTile<DT> (&X)[T][KT], ActTile<DT> (&Y)[T][MT], int lane) {
    typedef typename DT::v8 frag_t;
    constexpr int FPT = KT * 2;                            // fragments per out-tile
    constexpr int NG = MTN * FPT / 4;                      // groups of four
    static_assert(FPT % 4 == 0, "a tile's fragments must be whole groups of four");
    const int h = lane >> 5;
    frag_t g[2][4];
    f32x16 acc[T];
#pragma unroll
    for (int i = 0; i < 4; ++i) g[0][i] = ring.template frag<DT>(f0 + i);
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
        const bool last = EARLY && gi == NG - 1;           // (EARLY: see lp_block)
        if (gi + 1 < NG) {
#pragma unroll
            for (int i = 0; i < 4; ++i) g[(gi + 1) & 1][i] = ring.template frag<DT>(f0 + 4 * (gi + 1) + i);
        }
        DAL3_SCHED_FENCE();
        if (last) ring.acquire_wait();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = 4 * gi + i, m = MT0 + f / FPT, k = f % FPT;
            if (k == 0) {
                const f32x16 b = tile_from_channels(bias + 32 * m, h);
#pragma unroll
                for (int j = 0; j < T; ++j) acc[j] = b;
            }
#pragma unroll
            for (int j = 0; j < T; ++j) {
                acc[j] = DT::mfma(g[gi & 1][i], X[j][k >> 1].k[k & 1], acc[j]);
                if (last) {
                    DAL3_SCHED_FENCE();
                    lp_refill_gap<SEG, 4 * T>(ring, i * T + j);
                    DAL3_SCHED_FENCE();
                }
            }
            if (k == FPT - 1) {
#pragma unroll
                for (int j = 0; j < T; ++j) Y[j][m] = pack_relu<DT>(acc[j]);
            }
        }
        if (last) ring.issue_done();
        DAL3_SCHED_FENCE();
    }
}

// tiles of one layer per ring segment: as many whole out-tiles (KT*2 fragments each) as fit 32 fragments
__host__ __device__ constexpr int lp_tiles_per_seg(int kt, int mt) {
    return (32 / (kt * 2) < 1 ? 1 : 32 / (kt * 2)) < mt ? (32 / (kt * 2) < 1 ? 1 : 32 / (kt * 2)) : mt;
}

// a whole layer, one ring segment per TPS out-tiles
// CHAIN = false: every segment is acquired in front of its out-tiles. CHAIN = true: the layer's first segment is open
// already (by whoever used the ring before) and every segment opens the next one itself under its last MFMAs (lp_layer
// EARLY) — after the layer the following layer's (or group's) first segment is open.
template <class DT, int KT, int MT, int T, int SEG, int TPS, int M0, bool CHAIN = false, class Ring>
__device__ __forceinline__ void lp_seg_layers(Ring& ring, const float* bias, const ActTile<DT> (&X)[T][KT],
                                              ActTile<DT> (&Y)[T][MT], int lane) {
    if constexpr (M0 < MT) {
        if (!CHAIN) ring.acquire();
        lp_layer<DT, KT, MT, T, SEG, M0, TPS, CHAIN>(ring, 0, bias, X, Y, lane);
        lp_seg_layers<DT, KT, MT, T, SEG, TPS, M0 + TPS, CHAIN>(ring, bias, X, Y, lane);
    }
}

// (the maxima go into an LDS int array, zero-initialised, flushed to global once per group)
// Max epilogue of a TRANSPOSED tile (points on the MFMA rows = the 16 registers, channels on the columns = the lanes):
// the max over the points of the wave's T tiles is a max over registers — elementwise VALU, no lane exchange; the two
// lane halves hold the same 32 channels and simply both take part in the LDS atomic (two lanes per address).
template <int T>
__device__ __forceinline__ void lp_tile_max_t(const f32x16 (&acc)[T], const float* bias, int* smax, int lane) {
    float m = acc[0][0];
#pragma unroll
    for (int j = 0; j < T; ++j) {
#pragma unroll
        for (int r = (j == 0 ? 1 : 0); r < 16; ++r) m = __builtin_fmaxf(m, acc[j][r]);
    }
    const int ch = lane & 31;
    int bits = __float_as_int(m + bias[ch]);
    bits = bits > 0 ? bits : 0;
    atomicMax(smax + ch, bits);
}

// The caller acquires the first segment of the layer and instantiates its first call with FIRST; every call then opens the next segment
// itself, early: as soon as the slot's last fragments have been read the barrier is taken, and the refill's LDS-DMA
// instructions and the next segment's first four fragment reads (into g, carried to the next call) go between the last
// group's MFMAs. (Taken in one piece in front of a segment, wait + barrier + refill + first reads cost ~1,100 cycles
// with the matrix pipe idle.) After the layer's last segment the ring is one segment ahead, which is what a following
// layer's acquire() or the next group of a persistent kernel expects... so the caller must not acquire again.
// Round 3 (three-slot rings only): PEND_IN / DEFER_OUT spread a refill over a whole segment instead of the 8 consecutive
// gaps behind the barrier. With DEFER_OUT the call opens the next segment as before (wait + barrier) but does NOT issue
// the refill of the slot it frees; the following call, instantiated with PEND_IN, issues that refill's ML parts one per
// group of four fragments, in the gap behind the group's second MFMA (groups 0 .. NG-2: all of them before its own
// acquire_wait, whose counted vmcnt takes exactly ML younger loads to be in flight). The third slot gives the refill a
// whole further segment to land. A layer's last call is not DEFER_OUT, so the ring is in its usual state when the layer
// is left.
template <class DT, int KT, int T, int SEG, int TPS, bool FIRST, bool PEND_IN = false, bool DEFER_OUT = false, class Ring>
__device__ __forceinline__ void lp_max_tiles(Ring& ring, const ActTile<DT> (&X)[T][KT], const float* bias,
                                             int* smax, int lane, typename DT::v8 (&g)[2][4], f32x16 (&acc)[2][T]) {
    constexpr int FPT = KT * 2, NG = TPS * FPT / 4, ML = Ring::MY_LOADS;
    static_assert(!(PEND_IN || DEFER_OUT) || Ring::N_SLOTS == 3, "a deferred refill needs the third slot");
    static_assert(!PEND_IN || (NG >= 4 && ML <= 4 * (NG / 2 - 1)), "the pending refill's parts must fit the odd groups in front of the last one");
    static_assert(FPT % 8 == 0, "at least two groups of four fragments per tile");
    static_assert(NG % 2 == 0 && TPS % 2 == 0, "the carried group must be g[0], the carried tile acc[1]");
    static_assert(ML <= 4 * T * 2, "two refill parts per MFMA gap at most");
    static_assert(T <= 4, "two accumulator sets");
    // TWO accumulator sets (acc, the caller's: they live across the calls of a layer): a tile's max epilogue is placed
    // with the first 4 T MFMAs of the NEXT tile and dealt out between them (sched_group_barrier) instead of standing
    // between two tiles with the matrix pipe idle — also across segments: a segment's last tile is finished under the
    // next call's first MFMAs (bias / smax of consecutive segments are contiguous), the layer's very last tile by
    // lp_max_tiles_finish().
    if (FIRST) {
#pragma unroll
        for (int i = 0; i < 4; ++i) g[0][i] = ring.template frag<DT>(i);
    }
#pragma unroll
    for (int gi = 0; gi < NG; ++gi) {
        const bool last = gi == NG - 1;
        if (!last) {
#pragma unroll
            for (int i = 0; i < 4; ++i) g[(gi + 1) & 1][i] = ring.template frag<DT>(4 * (gi + 1) + i);
        }
        DAL3_SCHED_FENCE();
        if (last) ring.acquire_wait();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = 4 * gi + i, k = f % FPT, c = (f / FPT) & 1;
            if (k == 0) {
#pragma unroll
                for (int j = 0; j < T; ++j) acc[c][j] = f32x16{};
            }
#pragma unroll
            for (int j = 0; j < T; ++j) {
                acc[c][j] = DT::mfma(X[j][k >> 1].k[k & 1], g[gi & 1][i], acc[c][j]);      // operands swapped: D^T
                if (last) {
                    const int n = i * T + j;               // gap n of the 4 T after the barrier
                    DAL3_SCHED_FENCE();
                    if (n == 0) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) g[0][q] = ring.template frag<DT>(q);
                    }
                    if (!DEFER_OUT) lp_refill_gap<SEG, 4 * T>(ring, n);
                    DAL3_SCHED_FENCE();
                } else if (PEND_IN && (gi & 1) && j == 1) {
                    // the previous call's refill: the ODD groups (an even group's MFMAs are the ones a tile's max
                    // epilogue is interleaved with) take PER parts each, one behind the second MFMA of a fragment
                    constexpr int ODD = NG / 2 - 1;        // odd groups in front of the last one
                    constexpr int PER = (ML + ODD - 1) / ODD;
                    static_assert(PER <= 4, "one part per fragment of a group");
                    const int part = (gi / 2) * PER + i;
                    if (i < PER && part < ML) {
                        DAL3_SCHED_FENCE();
                        ring.issue_part(part);
                        if (part == ML - 1) ring.issue_done();
                        DAL3_SCHED_FENCE();
                    }
                }
            }
        }
        if (last && !DEFER_OUT) ring.issue_done();
        if ((4 * gi) % FPT == 0 && !last && (gi > 0 || !FIRST)) {
            const int t = 4 * gi / FPT - 1;                // the tile that finished with the previous group (-1: the
            lp_tile_max_t<T>(acc[t & 1], bias + 32 * t, smax + 32 * t, lane);              // previous segment's last)
#pragma unroll
            for (int n = 0; n < 4 * T; ++n) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);     // up to four VALU
                __builtin_amdgcn_sched_group_barrier(0x080, 2, 0);     // up to two DS (the bias read, the atomic)
            }
        }
        DAL3_SCHED_FENCE();
    }
}
// the epilogue of the layer's very last tile (bias / smax: that tile's 32 channels)
template <int T>
__device__ __forceinline__ void lp_max_tiles_finish(const f32x16 (&acc)[2][T], const float* bias, int* smax, int lane) {
    lp_tile_max_t<T>(acc[1], bias, smax, lane);
}

