// dal3_pointmlp_lp.hip — bf16 / fp16 MFMA versions of the three shared-MLP kernels (configs C3 / C5 of
// BASELINE.json). Same math and fusion as dal3_pointmlp.hip; see dal3_lp.h for the operand layout and
// the LDS-DMA weight ring. The K=3/4/8 first layer stays on the fp32 MFMA (raw coordinates are not
// rounded to 16 bits); accumulation is fp32 throughout; dconv5 is a 17th 16-bit out-tile of the decode kernel (rows 0, 1);
// logits, mask and all I/O are fp32.
#include "dal3_kernels.h"
#include "dal3_lp.h"

#define LP_WAVES 4

// ------------------------------------------------------------------------------------------------
#ifndef LP_PART
#define LP_PART 3
#endif
// Diagnostic build only (-DDAL3_STAMP): s_memtime stamps of the decode kernel's phases (tools/stamps_lp.py).
#if defined(DAL3_STAMP)
__device__ long long* g_stamps_lp = nullptr;               // one per translation unit (LP_PART 1: encode, 2: decode)
#if LP_PART & 2
extern "C" int dal3_debug_set_stamps_lp(void* p) {
#else
extern "C" int dal3_debug_set_stamps_lp_enc(void* p) {
#endif
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps_lp), &p, sizeof(p));
}
#define LP_STAMP(k)                                                                               \
    do {                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        unsigned long long t_;                                                                    \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        if (lane == 0 && grp < 4096) g_stamps_lp[(grp * 4 + wave) * 8 + (k)] = t_;                \
    } while (0)
// sub-phase stamps (16 per wave and group) behind the 4096 x 4 x 8 phase stamps
#define LP_SUB(k)                                                                                 \
    do {                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        unsigned long long t_;                                                                    \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        if (lane == 0 && grp < 4096) g_stamps_lp[4096 * 4 * 8 + (grp * 4 + wave) * 32 + (k)] = t_; \
    } while (0)
#else
#define LP_STAMP(k)
#define LP_SUB(k)
#endif

// bytes of the point heads' small LDS arrays (biases, maxima, first layer), rounded up to 1 KiB
__host__ __device__ constexpr int lp_head_small_bytes(int c1, int c2, int c3, int ks) {
    return ((c2 + c3 + 512 + 512 + c1 + (c1 / 32) * ks * 64) * 4 + 1023) / 1024 * 1024;
}
#define LP_ENC_SMALL_BYTES 11264                         // (1280 + 1024 + 64 + 256) floats, rounded up to 1 KiB
// Persistent like the decode kernel below (one workgroup per CU walking the 128 T-point groups, cyclic weight ring,
// the next group's points fetched a layer ahead). T = 4: with eight tiles per wave conv4's output (256 registers) no
// longer fits the arch VGPRs next to the accumulators and every conv5 MFMA paid v_accvgpr_read for its B operand
// (54 against 40 cycles per MFMA); what made T = 4 slower before — twice the workgroups, each with its start-up, and
// twice the ring refills per point — is gone with persistence and with the refills hidden under MFMAs.
#ifndef DAL3_LP_ENC_T
#define DAL3_LP_ENC_T 4
#endif
#ifndef DAL3_LP_ENC_SLOTS
#define DAL3_LP_ENC_SLOTS 3                               // 2 (with T = 2): two workgroups per CU (A/B switch)
#endif
template <class DT, int T>
__global__ __launch_bounds__(256, DAL3_LP_ENC_SLOTS == 2 ? 2 : 1) void ins_seg_encode_lp_kernel(InsSegLpW w, BCN pts, int c_in, int n_pts,
                                                                int tiles_per_item, int n_groups, float* __restrict__ g) {
    constexpr int SEG = LP_ENC_SEG;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // small arrays first (one address register + immediates, see the decode kernel), the ring behind them
    float* s_bias = reinterpret_cast<float*>(smem);        // b2 64 | b3 64 | b4 128 | b5 1024
    int* s_max = reinterpret_cast<int*>(s_bias + 1280);    // 1024 channel maxima of the current group
    float* s_b1 = s_bias + 1280 + 1024;                    // conv1: bias (64) and its four A fragments (4 x 64)
    float* s_w1 = s_b1 + 64;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5;

    for (int i = threadIdx.x; i < 1280; i += 256) s_bias[i] = w.bias_enc[i];
    for (int i = threadIdx.x; i < 1024; i += 256) s_max[i] = 0;
    if (threadIdx.x < 64) s_b1[threadIdx.x] = w.b1[threadIdx.x];
    s_w1[threadIdx.x] = w.w1[threadIdx.x];
    __syncthreads();
    LdsRing<SEG, DAL3_LP_ENC_SLOTS> ring;
    ring.init(w.enc_stream, smem + LP_ENC_SMALL_BYTES, LP_ENC_SEGS, wave, lane, true);

    float in_nx[T][2];
    auto prefetch = [&](int gq) {                          // (addresses rebuilt from a fresh lane id: see the decode kernel)
        unsigned z = 0;
        asm volatile("" : "+v"(z));
        const int l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z));
        load_points<2, T>(pts, gq / tiles_per_item, ((gq % tiles_per_item) * LP_WAVES + wave) * (32 * T), n_pts, c_in, in_nx, l);
    };
    prefetch(blockIdx.x);
    ring.acquire();                                        // segment 0 of the first group: conv2 | conv3 | conv4

  for (int grp = blockIdx.x; grp < n_groups; grp += gridDim.x) {
    const int64_t b = grp / tiles_per_item;
    LP_STAMP(0);
    ActTile<DT> x1[T][2], x2[T][2], x3[T][2], x4[T][4];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {                       // conv1 in fp32 (first_layer of dal3_device.h, operands from LDS)
        const f32x16 bv = tile_from_channels(s_b1 + 32 * mt, h);
#pragma unroll
        for (int j = 0; j < T; ++j) {
            f32x16 acc = bv;
#pragma unroll
            for (int k = 0; k < 2; ++k) acc = mfma32(s_w1[(mt * 2 + k) * 64 + lane], in_nx[j][k], acc);
            x1[j][mt] = pack_relu<DT>(acc);
        }
    }
    LP_STAMP(1);
    lp_layer<DT, 2, 2, T, SEG, 0, 2>(ring, 0, s_bias, x1, x2, lane);
    lp_layer<DT, 2, 2, T, SEG, 0, 2>(ring, 8, s_bias + 64, x2, x3, lane);
    lp_layer<DT, 2, 4, T, SEG, 0, 4, true>(ring, 16, s_bias + 128, x3, x4, lane);    // opens conv5's first segment
    {
        const int nx = grp + (int)gridDim.x;
        prefetch(nx < n_groups ? nx : grp);               // (the last group re-reads itself: uniform control flow)
        __builtin_amdgcn_sched_barrier(0);
    }
    LP_STAMP(2);
    typename DT::v8 g5[2][4];
    f32x16 acc5[2][T];                                     // conv5: 4 out-tiles (32 fragments) per segment; the last
#ifndef DAL3_LP_ENC_DEFER
#define DAL3_LP_ENC_DEFER 1                                // conv5's ring refills spread over the following segment (dal3_lp.h)
#endif
    constexpr bool DEFER = DAL3_LP_ENC_DEFER && DAL3_LP_ENC_SLOTS == 3;
    lp_max_tiles<DT, 4, T, SEG, 4, true, false, DEFER>(ring, x4, s_bias + 256, s_max, lane, g5, acc5);
    for (int seg = 1; seg < 7; ++seg)
        lp_max_tiles<DT, 4, T, SEG, 4, false, DEFER, DEFER>(ring, x4, s_bias + 256 + 128 * seg, s_max + 128 * seg, lane, g5, acc5);
    // the last call: issues its own refill at once, it has already opened the next group's segment 0
    lp_max_tiles<DT, 4, T, SEG, 4, false, DEFER, false>(ring, x4, s_bias + 256 + 128 * 7, s_max + 128 * 7, lane, g5, acc5);
    lp_max_tiles_finish<T>(acc5, s_bias + 256 + 1024 - 32, s_max + 1024 - 32, lane);
    LP_STAMP(3);
    __syncthreads();
    int* gi = reinterpret_cast<int*>(g + b * 1024);
    for (int i = threadIdx.x; i < 1024; i += 256) {
        const int v = s_max[i];
        if (v > 0) atomicMax(gi + i, v);
        s_max[i] = 0;                                      // for the next group: its LDS atomics come after >= 1 barrier
    }
    LP_STAMP(4);
  }
}

#define LP_DEC_SMALL_BYTES 7168                          // (864 + 512 + 64 + 256) floats, rounded up to 1 KiB
// ------------------------------------------------------------------------------------------------
// Persistent: 256 workgroups (one per CU: the ring takes 120 of the 160 KiB of LDS) each walk the 256-point groups
// g = blockIdx.x, blockIdx.x + gridDim.x, ... The weight stream is the same for every group, so the ring simply
// runs on cyclically: while a group's last segments are consumed the next group's first two are already on their
// way, and only the first group of a workgroup pays the cold start of the ring (with one workgroup per group it was
// paid 64 times per CU at 4096 x 1024, behind nothing: one wave per SIMD leaves no one to hide it).
template <class DT, int T>
__global__ __launch_bounds__(256) void ins_seg_decode_lp_kernel(InsSegLpW w, BCN pts, int c_in, int n_pts,
                                                                int tiles_per_item, int n_groups,
                                                                const float* __restrict__ gbias,
                                                                float* __restrict__ logits, uint8_t* __restrict__ mask) {
    constexpr int SEG = LP_DEC_SEG;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // LDS: the small arrays FIRST, the ring behind them: every array is then reached from one per-lane register
    // (16 h) plus an immediate offset (< 64 KiB); placed behind the 120-KiB ring each array needed an address
    // register of its own, hipcc spilled them, and every reload sat behind an s_waitcnt vmcnt(0) that drained the
    // ring's LDS-DMA right after it had been issued.
    // b2 64 | db2 256 | db3 128 | db4 128 | dw5 256 | db5 32  (= 864 floats), then the crop's dconv1 term (512)
    float* s_bias = reinterpret_cast<float*>(smem);
    float* s_gb = s_bias + 864;
    float* s_b1 = s_gb + 512;                              // conv1's bias (64) and its four A fragments (4 x 64): read
    float* s_w1 = s_b1 + 64;                               // per group, so from LDS, not from global
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);      // scalar: everything derived from it too
    const int h = lane >> 5;

    for (int i = threadIdx.x; i < 864; i += 256) s_bias[i] = w.bias_dec[i];
    if (threadIdx.x < 64) s_b1[threadIdx.x] = w.b1[threadIdx.x];
    s_w1[threadIdx.x] = w.w1[threadIdx.x];
    const float* s_db2 = s_bias + 64;
    const float* s_db3 = s_bias + 320;
    const float* s_db4 = s_bias + 448;
    const float* s_db5 = s_bias + 832;
    LdsRing<SEG> ring;
    ring.init(w.dec_stream, smem + LP_DEC_SMALL_BYTES, LP_DEC_SEGS, wave, lane, true);

    // A lane id the compiler cannot hoist out of the group loop: addresses derived from the kernel's `lane` are loop
    // invariants, hipcc keeps them across the main loop where registers are scarcest and spills them, and every reload is
    // a scratch read behind an `s_waitcnt vmcnt(0)` that also waits for whatever LDS-DMA the ring has in flight (stamps:
    // 2,500 ticks per group in front of the logit stores, 700 in front of publish_gb). Rebuilt where it is used, the
    // id costs two instructions and no register across the loop.
    auto fresh_lane = [&]() {
        unsigned z = 0;
        asm volatile("" : "+v"(z));
        return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z));
    };
    // the NEXT group's points and dconv1 term are fetched into registers in the middle of the current group (after
    // its main loop), a good 15 us before they are needed: read at the top of a group they cost an exposed HBM round
    // trip per group (stamps: 15 % of the kernel sat in front of the first MFMA)
    float in_nx[T][2], gb_nx[2];
    auto prefetch = [&](int g) {
        // the lane id goes through an opaque asm: hipcc otherwise computes the per-lane address parts once, keeps them
        // (64-bit, per tile and channel) across the main loop where registers are scarcest, spills them, and each
        // reload is a scratch read behind an s_waitcnt vmcnt(0) that drains the ring's LDS-DMA (2,400 ticks per group)
        const int l = fresh_lane();                        // (not `lane`, and not hoistable: either would be spilled too)
        const int64_t bb = g / tiles_per_item;
        load_points<2, T>(pts, bb, ((g % tiles_per_item) * LP_WAVES + wave) * (32 * T), n_pts, c_in, in_nx, l);
        gb_nx[0] = gbias[bb * 512 + wave * 64 + l];
        gb_nx[1] = gbias[bb * 512 + 256 + wave * 64 + l];
    };
#ifndef DAL3_LP_DEC_XCD
#define DAL3_LP_DEC_XCD 1
#endif
    const int vblk = DAL3_LP_DEC_XCD ? xcd_contiguous_block() : (int)blockIdx.x;    // (dal3_kernels.h: a crop's groups on one L2)
    prefetch(vblk);
    // The crop's dconv1 term goes to LDS one group ahead: here for the first group, in front of dconv4 for the others
    // (nobody reads the old one after the main loop; dconv4's barrier, here acquire(), publishes the new one).
    // (the compiler's wait for gb_nx is an `s_waitcnt vmcnt(0)`: it does not know of the ring's LDS-DMA, which it
    // therefore also waits for — so this is called where the ring's last refill is oldest, and the points fetched with
    // gb_nx are declared used here too, so that the same wait serves them)
    auto publish_gb = [&]() {
        const int t = wave * 64 + fresh_lane();
        s_gb[t] = gb_nx[0];
        s_gb[256 + t] = gb_nx[1];
#pragma unroll
        for (int j = 0; j < T; ++j) asm volatile("" : "+v"(in_nx[j][0]), "+v"(in_nx[j][1]));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };
    publish_gb();
    ring.acquire();                                        // segment 0 of the first group (publishes s_bias ... too)

  for (int grp = vblk; grp < n_groups; grp += gridDim.x) {
    const int64_t b = grp / tiles_per_item;
    const int n0 = ((grp % tiles_per_item) * LP_WAVES + wave) * (32 * T);
    LP_STAMP(0);
    // a crop with a non-finite coordinate: its dconv1 term is NaN (dal3_device.h) -> NaN logits, empty mask below. Read here,
    // while s_gb holds THIS group's term (the next group's is published in front of dconv4), into a scalar register.
    const bool crop_bad = (__builtin_amdgcn_readfirstlane(__float_as_int(s_gb[0])) & 0x7F800000) == 0x7F800000;

    ActTile<DT> x1[T][2], x2[T][2];
    const int lnA = fresh_lane(), hA = lnA >> 5;          // for the prologue (see fresh_lane)
    {
        // conv1 in fp32 (first_layer of dal3_device.h, operands from LDS). Segment 0 (conv2 | dconv1a chunk 0) is open:
        // by the acquire in front of the loop for the first group, by the previous group's dconv4 for the others.
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const f32x16 bv = tile_from_channels(s_b1 + 32 * mt, hA);
#pragma unroll
            for (int j = 0; j < T; ++j) {
                f32x16 acc = bv;
#pragma unroll
                for (int k = 0; k < 2; ++k) acc = mfma32(s_w1[(mt * 2 + k) * 64 + lnA], in_nx[j][k], acc);
                x1[j][mt] = pack_relu<DT>(acc);
            }
        }
    }
    LP_SUB(8);
    lp_layer<DT, 2, 2, T, SEG, 0, 2>(ring, 0, s_bias, x1, x2, lnA);
    LP_SUB(9);

    f32x16 a2[T][8];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
        const f32x16 bv = tile_from_channels(s_db2 + 32 * mt, hA);
#pragma unroll
        for (int j = 0; j < T; ++j) a2[j][mt] = bv;
    }
    LP_SUB(10);
    f32x16 tA[T], tB[T];
    typedef typename DT::v8 frag_t;
    // LDS fragment reads are issued a whole group of four (= 8 MFMAs, 256 cycles) before their MFMAs: with the read
    // right in front of its MFMA (what the compiler does on its own) every fragment exposed the LDS latency
#ifndef DAL3_ABL
#define DAL3_ABL 0                                        // timing experiments only (tools/README.md): bit mask of main-loop parts left out
#endif
    auto load4 = [&](frag_t (&d)[4], int f0) {
        if (DAL3_ABL & 8) {
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("" : "+v"(d[i]));
            return;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) d[i] = ring.template frag<DT>(f0 + i);
    };
    // t <- the crop's dconv1 term of a chunk (the accumulators' initial value), read from LDS straight into both
    // tiles' registers a half-iteration before the chunk's MFMAs
    auto init_chunk = [&](f32x16 (&t)[T], int chunk) {
        if (DAL3_ABL & 16) {
            asm volatile("" : "+v"(t[0]));
            return;
        }
        // tile 0's registers only: the term is the same for both tiles (one crop), MfmaAsm::block4x2 feeds it to both
        t[0] = tile_from_channels(s_gb + 32 * (chunk & 15), h);
    };
    auto dconv1_chunk = [&](f32x16 (&t)[T], const frag_t (&q)[4]) {
        static_assert(T == 2, "MfmaAsm::block4x2 is written for two point tiles");
        if (DAL3_ABL & 64) return;
        MfmaAsm<DT>::block4x2(t[0], t[1], q[0], q[1], q[2], q[3], x2[0][0].k[0], x2[0][0].k[1], x2[0][1].k[0],
                              x2[0][1].k[1], x2[1][0].k[0], x2[1][0].k[1], x2[1][1].k[0], x2[1][1].k[1]);
    };
    frag_t q[4], ga[4], gb[4];
    ActTile<DT> pA[T], pB[T];
    {
        load4(q, 8);
        init_chunk(tA, 0);
        DAL3_SCHED_FENCE();
        dconv1_chunk(tA, q);
        init_chunk(tB, 1);
#pragma unroll
        for (int j = 0; j < T; ++j) pA[j] = pack_relu<DT>(tA[j]);
        init_chunk(tA, 2);
    }
    LP_SUB(11);
    ring.acquire();                                        // segment 1
    load4(q, 0);
    load4(ga, 4);
    LP_STAMP(1);
    // Segment 1+i: 1a(2i+1) | 2(2i) | 1a(2i+2) | 2(2i+1). At the top of an iteration the segment's first eight fragments
    // are already in registers; the barrier that opens the NEXT segment is taken as soon as the current one's last
    // fragments have been read. pA / pB: the packed chunk the first / second half's dconv2 MFMAs consume; each is
    // rewritten (from tA / tB, which is then re-initialised for the chunk after next) under the other half's MFMAs.
    // WHERE the non-MFMA work sits (round 1 had the refill's ten pieces in the last eight gaps and the packing two
    // pairs per gap): tools/ubench/lp_loop.hip prices it: four VALU in one MFMA
    // gap stretch that gap by ~13 cycles, two by ~2.5; ten LDS-DMA pieces dealt out one per gap in consecutive gaps
    // (plus the counted wait and barrier behind them) cost ~330 cycles per 80 MFMAs, one piece every eighth gap ~110.
    // So: the ring refill that the barrier at the end of iteration i-1 made room for is issued one piece after every
    // eighth MFMA of iteration i (it has until the barrier at the end of iteration i+1 to land: the ring's third slot
    // gives that distance for free), and a chunk's 16-bit packing takes ONE register pair per gap over the sixteen
    // gaps of the two MFMA groups in front of its consumer instead of two pairs in each of eight gaps. Same
    // instructions, same operands, same results.
    constexpr int ML = LdsRing<SEG>::MY_LOADS;
    static_assert(ML == 10, "ten refill points per iteration below");
    auto mma4_gap = [&](const frag_t (&a)[4], const ActTile<DT> (&p)[T], int mt0, auto&& gap) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < T; ++j) {
                a2[j][mt0 + (i >> 1)] = DT::mfma(a[i], p[j].k[i & 1], a2[j][mt0 + (i >> 1)]);
                DAL3_SCHED_FENCE();
                gap(2 * i + j);
                DAL3_SCHED_FENCE();
            }
        }
    };
    auto pack_pair = [&](int4_t (&w)[T][2], const f32x16 (&t)[T], int k) {   // register pair k of the 16 (T x 2 x 4) of a chunk
        const int tj = k >> 3, ts = (k >> 2) & 1, ti = k & 3;
        if (DAL3_ABL & 32) {
            asm volatile("" : "=v"(w[tj][ts][ti]));
            return;
        }
        w[tj][ts][ti] = pack_relu_pair<DT>(t[tj][8 * ts + 2 * ti], t[tj][8 * ts + 2 * ti + 1]);
    };
    auto packed_into = [&](ActTile<DT> (&pn)[T], const int4_t (&w)[T][2]) {
#pragma unroll
        for (int j = 0; j < T; ++j) {
            pn[j].k[0] = __builtin_bit_cast(frag_t, w[j][0]);
            pn[j].k[1] = __builtin_bit_cast(frag_t, w[j][1]);
        }
    };
    for (int i = 0; i < 8; ++i) {
        const bool refill = i > 0;                         // (iteration 0's slot was refilled by the acquire() above)
        auto part = [&](int k) {
            if (refill && !(DAL3_ABL & 1)) ring.issue_part(k);
        };
        DAL3_SCHED_FENCE();
        dconv1_chunk(tB, q);                               // chunk 2i+1
        load4(gb, 8);
        part(0);
        DAL3_SCHED_FENCE();
        mma4_gap(ga, pA, 0, [&](int g) { if (g == 7) part(1); });
        DAL3_SCHED_FENCE();
        load4(ga, 12);
        DAL3_SCHED_FENCE();
        mma4_gap(gb, pA, 2, [&](int g) { if (g == 7) part(2); });
        DAL3_SCHED_FENCE();
        load4(gb, 16);
        DAL3_SCHED_FENCE();
        int4_t wB[T][2];
        mma4_gap(ga, pA, 4, [&](int g) { pack_pair(wB, tB, g); if (g == 7) part(3); });
        DAL3_SCHED_FENCE();
        load4(q, 20);
        load4(ga, 24);
        DAL3_SCHED_FENCE();
        mma4_gap(gb, pA, 6, [&](int g) { pack_pair(wB, tB, 8 + g); if (g == 7) part(4); });
        packed_into(pB, wB);
        init_chunk(tB, 2 * i + 3);
        DAL3_SCHED_FENCE();
        dconv1_chunk(tA, q);                               // chunk 2i+2 (16 = zero filler weights, result unused)
        load4(gb, 28);
        part(5);
        DAL3_SCHED_FENCE();
        mma4_gap(ga, pB, 0, [&](int g) { if (g == 7) part(6); });
        DAL3_SCHED_FENCE();
        load4(ga, 32);
        DAL3_SCHED_FENCE();
        int4_t wA[T][2];
        mma4_gap(gb, pB, 2, [&](int g) { pack_pair(wA, tA, g); if (g == 7) part(7); });
        DAL3_SCHED_FENCE();
        load4(gb, 36);
        DAL3_SCHED_FENCE();
        mma4_gap(ga, pB, 4, [&](int g) { pack_pair(wA, tA, 8 + g); if (g == 3) part(8); if (g == 7) part(9); });
        packed_into(pA, wA);
        init_chunk(tA, 2 * i + 4);
        if (refill) ring.issue_done();
        DAL3_SCHED_FENCE();
        ring.acquire_wait();                               // segment 2+i (after the loop: dconv3's first)
        mma4_gap(gb, pB, 6, [&](int g) {                   // (the slot's refill: dealt out over the next iteration)
            if (g == 0) {
                load4(q, 0);
                load4(ga, 4);
            }
        });
        DAL3_SCHED_FENCE();
        if (i == 0) LP_SUB(12);
        if (i == 1) LP_SUB(13);
    }
    // the refill the loop's last barrier made room for has no next iteration to ride on: in one piece, once per group
#pragma unroll
    for (int k = 0; k < ML; ++k) {
        if (!(DAL3_ABL & 1)) ring.issue_part(k);
    }
    ring.issue_done();
    DAL3_SCHED_FENCE();
    LP_STAMP(2);
    {
        const int nx = grp + (int)gridDim.x;
        prefetch(nx < n_groups ? nx : grp);               // (the last group re-reads itself: uniform control flow)
        __builtin_amdgcn_sched_barrier(0);                 // pinned here: left alone hipcc sinks loads to their first use
    }
    frag_t gq[2][4];                                        // dconv3's first four fragments: its segment has been open
    load4(gq[0], 0);                                       // since the main loop's last barrier — read in front of the a2
    DAL3_SCHED_FENCE();                                    // pack, not behind it with the matrix pipe waiting
    ActTile<DT> xd[T][8], y3[T][4];
#ifndef DAL3_LP_A2_UNDER
#define DAL3_LP_A2_UNDER 1
#endif
    // one register pair of dconv2's output chunk mt -> its 16-bit place in dconv3's input
    static_assert(!DAL3_LP_A2_UNDER || T == 2, "pack_a2_pair's p -> (tile, half, word) map and the gap schedule below (16 MFMAs per chunk pair) are written for T == 2");
    auto pack_a2_pair = [&](int mt, int p) {
        const int tj = p >> 3, ts = (p >> 2) & 1, ti = p & 3;
        int4_t w = __builtin_bit_cast(int4_t, xd[tj][mt].k[ts]);
        w[ti] = pack_relu_pair<DT>(a2[tj][mt][8 * ts + 2 * ti], a2[tj][mt][8 * ts + 2 * ti + 1]);
        xd[tj][mt].k[ts] = __builtin_bit_cast(frag_t, w);
    };
    if (DAL3_LP_A2_UNDER) {
        // round 5 (0.6 % at both shapes, bitwise-equal: tools/ab_kernels.py): only the first two chunks are packed with the matrix pipe idle (they also free 64 AGPRs for
        // dconv3's first accumulators); chunks 2..7 are packed under dconv3 tile 0's MFMAs, four pairs per gap, each a
        // whole chunk (four gaps) ahead of its first use
#pragma unroll
        for (int j = 0; j < T; ++j) {
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
                const int4_t z = {0, 0, 0, 0};
                xd[j][mt].k[0] = __builtin_bit_cast(frag_t, z);
                xd[j][mt].k[1] = __builtin_bit_cast(frag_t, z);
            }
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int pp = 0; pp < 16; ++pp) pack_a2_pair(mt, pp);
        }
    } else {
#pragma unroll
        for (int j = 0; j < T; ++j) {
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) xd[j][mt] = pack_relu<DT>(a2[j][mt]);
        }
    }
    const int lnB = fresh_lane(), hB = lnB >> 5;         // for the rest of the group (see fresh_lane)
    LP_STAMP(3);
    // dconv3 (two segments of two out-tiles) and dconv4 (one segment): each segment's last block opens the next
    // segment itself, under its last eight MFMAs. Two accumulator sets alternate: a finished tile is read out of its
    // AGPRs, ReLU'd and (dconv3) rounded in pieces of a few instructions between the MFMAs of the NEXT tile — taken in
    // one piece after each tile (64 VALU per 32 MFMAs) these layers ran at 64 / 71 cycles per MFMA.
    static_assert(T == 2, "gap schedules below are written for two point tiles");
    f32x16 accA[T], accB[T];
    int4_t pw[T][2];
    auto bias_tile = [&](f32x16 (&acc)[T], const float* bias) {  // tile 0's registers: lp_block feeds it to every tile
        acc[0] = tile_from_channels(bias, hB);
    };
    auto pack_piece = [&](const f32x16 (&acc)[T], int p) {       // register pair p of the 16 (T x 2 x 4) of a tile
        const int tj = p >> 3, ts = (p >> 2) & 1, ti = p & 3;
        pw[tj][ts][ti] = pack_relu_pair<DT>(acc[tj][8 * ts + 2 * ti], acc[tj][8 * ts + 2 * ti + 1]);
    };
    auto packed_to = [&](int m) {
#pragma unroll
        for (int j = 0; j < T; ++j) {
            y3[j][m].k[0] = __builtin_bit_cast(frag_t, pw[j][0]);
            y3[j][m].k[1] = __builtin_bit_cast(frag_t, pw[j][1]);
        }
    };
    // dconv3 | dconv4 | dconv5 as ONE chain of blocks: every block reads the first four fragments of the next one under
    // its own MFMAs (lp_block F_NEXT / CARRY_IN), the chain's first four were read in front of the a2 pack above.
    bias_tile(accA, s_db3);
    if (DAL3_LP_A2_UNDER) {
        lp_block<DT, 8, T, SEG, false, true, 16>(ring, 0, xd, accA, gq, [&](int n) {      // dconv3 tile 0 | pack a2 chunks 2..7
            const int c = 2 + n / 4;
            if (c < 8) {
#pragma unroll
                for (int q = 0; q < 4; ++q) pack_a2_pair(c, (n & 3) * 4 + q);
            }
        });
    } else {
        lp_block<DT, 8, T, SEG, false, true, 16>(ring, 0, xd, accA, gq);                 // dconv3 tile 0
    }
    LP_SUB(0);
    bias_tile(accB, s_db3 + 32);
    lp_block<DT, 8, T, SEG, true, true, 0>(ring, 16, xd, accB, gq, [&](int n) {          // tile 1 | pack tile 0
        if (n % 2 == 0) pack_piece(accA, n / 2);
    });
    packed_to(0);
    LP_SUB(1);
    bias_tile(accA, s_db3 + 64);
    lp_block<DT, 8, T, SEG, false, true, 16>(ring, 0, xd, accA, gq, [&](int n) {         // tile 2 | pack tile 1
        if (n % 2 == 0) pack_piece(accB, n / 2);
    });
    packed_to(1);
    LP_SUB(2);
    publish_gb();                                          // the next group's (fetched after the main loop); nobody reads
    bias_tile(accB, s_db3 + 96);                           // s_gb between the main loop's last barrier and the next group
    lp_block<DT, 8, T, SEG, true, true, 0>(ring, 16, xd, accB, gq, [&](int n) {          // tile 3 | pack tile 2
        if (n % 2 == 0) pack_piece(accA, n / 2);
    });
    packed_to(2);
    LP_SUB(3);
    LP_STAMP(4);
    // dconv4's output is rounded to 16 bits like every other hidden layer: dconv5 (128 -> 2) then is ONE more out-tile
    // on the matrix pipe — eight fragments at the end of dconv4's ring segment, rows 0 and 1 real, the rest zero; 16
    // MFMAs per wave and group — instead of 256 fp32 FMAs per lane on the VALU with the matrix pipe idle (stamps: 3,800
    // ticks per group, 7 % of the kernel). Rows 0 / 1 of the result are registers 0 / 1 of lanes 0..31: no lane exchange.
    ActTile<DT> y4p[T][4];
    auto packed_to4 = [&](int m) {
#pragma unroll
        for (int j = 0; j < T; ++j) {
            y4p[j][m].k[0] = __builtin_bit_cast(frag_t, pw[j][0]);
            y4p[j][m].k[1] = __builtin_bit_cast(frag_t, pw[j][1]);
        }
    };
    bias_tile(accA, s_db4);
    lp_block<DT, 4, T, SEG, false, true, 8>(ring, 0, y3, accA, gq, [&](int n) {          // dconv4 tile 0 | pack dconv3 tile 3:
        if (n < 8) {                                                                     // its k-steps 6, 7 (n >= 12) are
            pack_piece(accB, 2 * n);                                                     // the first to need it
            pack_piece(accB, 2 * n + 1);
        }
        if (n == 7) packed_to(3);
    });
    LP_SUB(4);
    bias_tile(accB, s_db4 + 32);
    lp_block<DT, 4, T, SEG, false, true, 16>(ring, 8, y3, accB, gq, [&](int n) { pack_piece(accA, n); });    // tile 1 | pack tile 0
    packed_to4(0);
    LP_SUB(5);
    bias_tile(accA, s_db4 + 64);
    lp_block<DT, 4, T, SEG, false, true, 24>(ring, 16, y3, accA, gq, [&](int n) { pack_piece(accB, n); });   // tile 2 | pack tile 1
    packed_to4(1);
    LP_SUB(6);
    bias_tile(accB, s_db4 + 96);
    lp_block<DT, 4, T, SEG, false, true, 32>(ring, 24, y3, accB, gq, [&](int n) { pack_piece(accA, n); });   // tile 3 | pack tile 2
    packed_to4(2);
    LP_STAMP(5);
    bias_tile(accA, s_db5);                                                              // (db5 padded to 32 rows)
    lp_block<DT, 4, T, SEG, true, true>(ring, 32, y4p, accA, gq, [&](int n) {            // dconv5 | pack dconv4 tile 3 (its
        if (n < 8) {                                                                     // k-steps 6, 7 need it); the second
            pack_piece(accB, 2 * n);                                                     // half opens the next group's
            pack_piece(accB, 2 * n + 1);                                                 // segment 0
        }
        if (n == 7) packed_to4(3);
    });
    LP_SUB(7);
#pragma unroll
    for (int j = 0; j < T; ++j) {
        float s0 = accA[j][0], s1 = accA[j][1];            // rows 0, 1 of the tile: lanes 0..31, one point each
        if (crop_bad) s0 = s1 = __int_as_float(DAL3_QNAN_BITS);
        const int n = n0 + 32 * j + (lnB & 31);
        if (hB == 0 && n < n_pts) {
            f32x2 o;
            o[0] = s0;
            o[1] = s1;
            *reinterpret_cast<f32x2*>(logits + (b * n_pts + n) * 2) = o;
            mask[b * n_pts + n] = (!crop_bad && s0 < s1) ? 1 : 0;
        }
    }
    LP_STAMP(6);
#if defined(DAL3_STAMP)
    if (lane == 0 && grp < 4096) {                         // slot 7: ticks inside s_barrier (low 32 bits) and inside the
        g_stamps_lp[(grp * 4 + wave) * 8 + 7] = (ring.bar_ticks & 0xffffffffll) | (ring.wait_ticks << 32);   // counted wait
    }
    ring.bar_ticks = ring.wait_ticks = 0;
#endif
  }   // groups
}

// ------------------------------------------------------------------------------------------------
// conv1..4 + max over the points. Per layer the out-tiles are grouped into segments of <= 32 fragments.
// Persistent like the two kernels above: one workgroup per CU walks the (point tile, item) pairs in item-minor order
// (XCD balance: see point_head_kernel), skipping the tiles that hold only duplicates.
// The point heads fit 256 registers per wave, so with a TWO-slot ring (64 + 8 KiB of LDS) two workgroups share a CU:
// two waves per SIMD, one computing while the other sits at a barrier or in an epilogue (heads 0.63 -> 0.61 ms; with
// two slots one segment is in flight ahead of the one in use instead of two). 3 restores one workgroup per CU.
#ifndef DAL3_LP_HEAD_SLOTS
#define DAL3_LP_HEAD_SLOTS 2
#endif
template <class DT, int KS, int C1, int C2, int C3, int T>
__global__ __launch_bounds__(256, DAL3_LP_HEAD_SLOTS == 2 ? 2 : 1) void point_head_lp_kernel(PointHeadLpW w, BCN x, int c_in, int n_pts_all,
                                                            int n_items, int n_groups, float* __restrict__ feat,
                                                            const int32_t* __restrict__ distinct) {
    constexpr int SEG = LP_HEAD_SEG;
    constexpr int K2 = C1 / 32, M2 = C2 / 32, K3 = C2 / 32, M3 = C3 / 32, K4 = C3 / 32, M4 = 16;
    constexpr int TPS2 = lp_tiles_per_seg(K2, M2), TPS3 = lp_tiles_per_seg(K3, M3), TPS4 = lp_tiles_per_seg(K4, M4);
    constexpr int NB = C2 + C3 + 512, NW1 = K2 * KS * 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* s_bias = reinterpret_cast<float*>(smem);        // b2 C2 | b3 C3 | b4 512 ; small arrays first, ring behind
    int* s_max = reinterpret_cast<int*>(s_bias + NB);      // 512 channel maxima of the current group
    float* s_b1 = s_bias + NB + 512;                       // first layer: bias (C1) and A fragments [K2][KS][64]
    float* s_w1 = s_b1 + C1;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int h = lane >> 5;

    for (int i = threadIdx.x; i < NB; i += 256) s_bias[i] = w.bias[i];
    for (int i = threadIdx.x; i < 512; i += 256) s_max[i] = 0;
    for (int i = threadIdx.x; i < C1; i += 256) s_b1[i] = w.b1[i];
    for (int i = threadIdx.x; i < NW1; i += 256) s_w1[i] = w.w1[i];
    __syncthreads();
    LdsRing<SEG, DAL3_LP_HEAD_SLOTS> ring;
    ring.init(w.stream, smem + lp_head_small_bytes(C1, C2, C3, KS), M2 / TPS2 + M3 / TPS3 + M4 / TPS4, wave, lane, true);

    // the next group's points and its item's count of distinct points, fetched a layer ahead (see the decode kernel)
    float in_nx[T][KS];
    int np_nx;
    auto prefetch = [&](int id) {
        unsigned z = 0;
        asm volatile("" : "+v"(z));
        const int l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z));
        const int bb = id % n_items;
        int np = n_pts_all;
        if (distinct) {                                    // duplicates beyond the first distinct[b] points: see point_head_kernel
            const int d = distinct[bb];
            np = d <= 0 ? 1 : (d < np ? d : np);
        }
        np_nx = np;
        load_points<KS, T>(x, bb, ((id / n_items) * LP_WAVES + wave) * (32 * T), np, c_in, in_nx, l);
    };
    prefetch(blockIdx.x);
    ring.acquire();                                        // the first segment of the first group's conv2

  for (int id = blockIdx.x; id < n_groups; id += gridDim.x) {
    const int64_t b = id % n_items;
    const int wg_tile = id / n_items;
    const int nx = id + (int)gridDim.x < n_groups ? id + (int)gridDim.x : id;
    if (wg_tile * LP_WAVES * 32 * T >= np_nx) {            // only duplicates in this tile: uniform skip (the ring stays put)
        prefetch(nx);
        continue;
    }
    ActTile<DT> x1[T][K2], x2[T][M2], x3[T][M3];
#pragma unroll
    for (int mt = 0; mt < K2; ++mt) {                      // first layer in fp32 (first_layer of dal3_device.h, operands from LDS)
        const f32x16 bv = tile_from_channels(s_b1 + 32 * mt, h);
#pragma unroll
        for (int j = 0; j < T; ++j) {
            f32x16 acc = bv;
#pragma unroll
            for (int k = 0; k < KS; ++k) acc = mfma32(s_w1[(mt * KS + k) * 64 + lane], in_nx[j][k], acc);
            x1[j][mt] = pack_relu<DT>(acc);
        }
    }
    lp_seg_layers<DT, K2, M2, T, SEG, TPS2, 0, true>(ring, s_bias, x1, x2, lane);     // every segment opens the next
    prefetch(nx);
    __builtin_amdgcn_sched_barrier(0);
    lp_seg_layers<DT, K3, M3, T, SEG, TPS3, 0, true>(ring, s_bias + C2, x2, x3, lane);
    const float* s_b4 = s_bias + C2 + C3;
    typename DT::v8 g4[2][4];
    f32x16 acc4[2][T];                                     // (the last call opens the next group's first segment)
    lp_max_tiles<DT, K4, T, SEG, TPS4, true>(ring, x3, s_b4, s_max, lane, g4, acc4);
    for (int seg = 1; seg < M4 / TPS4; ++seg)
        lp_max_tiles<DT, K4, T, SEG, TPS4, false>(ring, x3, s_b4 + 32 * TPS4 * seg, s_max + 32 * TPS4 * seg, lane, g4, acc4);
    lp_max_tiles_finish<T>(acc4, s_b4 + 512 - 32, s_max + 512 - 32, lane);
    __syncthreads();
    int* fi = reinterpret_cast<int*>(feat + b * 512);
    {
        // the thread index rebuilt from an opaque lane id: derived from threadIdx.x the per-lane address is a loop
        // invariant that hipcc hoists, spills, and reloads here behind an s_waitcnt vmcnt(0) — which also waits for the
        // ring refill issued a moment ago (see the decode kernel's fresh_lane)
        unsigned z = 0;
        asm volatile("" : "+v"(z));
        const int tix = wave * 64 + (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, z));
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int i = tix + 256 * r;
            const int v = s_max[i];
            if (v > 0) atomicMax(fi + i, v);
            s_max[i] = 0;                                  // for the next group: its LDS atomics come after >= 1 barrier
        }
    }
  }
}

// ------------------------------------------------------------------------------------------------
static inline int lp_tiles(int n_pts, int T) { return (n_pts + 32 * LP_WAVES * T - 1) / (32 * LP_WAVES * T); }

static int lp_cu_count() {                                  // one persistent workgroup per CU
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
            v = 256;
        n = v;
    }
    return n;
}
template <class DT>
static hipError_t enc_lp(const InsSegLpW& w, BCN pts, int c_in, int B, int N, float* g, hipStream_t s) {
    constexpr int T = DAL3_LP_ENC_T;
    const size_t lds = DAL3_LP_ENC_SLOTS * LP_ENC_SEG * 1024 + LP_ENC_SMALL_BYTES;
    auto k = ins_seg_encode_lp_kernel<DT, T>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const int tpi = lp_tiles(N, T);
    const int64_t n_groups = (int64_t)B * tpi;
    if (n_groups > 0x7fffffff) return hipErrorInvalidValue;
    const int64_t wgs = (int64_t)lp_cu_count() * (DAL3_LP_ENC_SLOTS == 2 ? 2 : 1);
    const int64_t grid = n_groups < wgs ? n_groups : wgs;
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(256), lds, s, w, pts, c_in, N, tpi, (int)n_groups, g);
    return hipGetLastError();
}
template <class DT>
static hipError_t dec_lp(const InsSegLpW& w, BCN pts, int c_in, int B, int N, const float* gbias, float* logits,
                         uint8_t* mask, hipStream_t s) {
    constexpr int T = DAL3_LP_DEC_T;
    const size_t lds = LP_SLOTS * LP_DEC_SEG * 1024 + LP_DEC_SMALL_BYTES;
    auto k = ins_seg_decode_lp_kernel<DT, T>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const int tpi = lp_tiles(N, T);
    const int64_t n_groups = (int64_t)B * tpi;
    if (n_groups > 0x7fffffff) return hipErrorInvalidValue;
    const int64_t grid = n_groups < lp_cu_count() ? n_groups : lp_cu_count();
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(256), lds, s, w, pts, c_in, N, tpi, (int)n_groups, gbias, logits, mask);
    return hipGetLastError();
}
template <class DT, int KS, int C1, int C2, int C3>
static hipError_t head_lp(const PointHeadLpW& w, BCN x, int c_in, int B, int M, float* feat, const int32_t* distinct,
                          hipStream_t s) {
    constexpr int T = DAL3_LP_HEAD_T;
    const size_t lds = DAL3_LP_HEAD_SLOTS * LP_HEAD_SEG * 1024 + lp_head_small_bytes(C1, C2, C3, KS);
    auto k = point_head_lp_kernel<DT, KS, C1, C2, C3, T>;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    const int64_t n_groups = (int64_t)B * lp_tiles(M, T);
    if (n_groups > 0x7fffffff) return hipErrorInvalidValue;
    const int64_t wgs = (int64_t)lp_cu_count() * (DAL3_LP_HEAD_SLOTS == 2 ? 2 : 1);
    const int64_t grid = n_groups < wgs ? n_groups : wgs;
    hipLaunchKernelGGL(k, dim3((unsigned)grid), dim3(256), lds, s, w, x, c_in, M, B, (int)n_groups, feat, distinct);
    return hipGetLastError();
}

// This file is compiled twice (Makefile): LP_PART=1 -> encode + point heads, with `-mllvm
// -amdgpu-mfma-vgpr-form` (accumulators in arch VGPRs: the pack/ReLU/max epilogues then need no
// v_accvgpr_read; encode 1.18 -> 1.00 ms, heads 0.90 -> 0.86 ms), LP_PART=2 -> decode without it (its 256
// accumulator registers only fit beside the rest in the AGPR half; with the flag it slows 1.92 -> 2.27 ms).
#if LP_PART & 1
hipError_t launch_ins_seg_encode_lp(int dtype, const InsSegLpW& w, BCN pts, int c_in, int B, int N, float* g, hipStream_t s) {
    return dtype == DAL3_BF16 ? enc_lp<BF16>(w, pts, c_in, B, N, g, s) : enc_lp<FP16>(w, pts, c_in, B, N, g, s);
}
#endif
#if LP_PART & 2
hipError_t launch_ins_seg_decode_lp(int dtype, const InsSegLpW& w, BCN pts, int c_in, int B, int N, const float* gbias,
                                    float* logits, uint8_t* mask, hipStream_t s) {
    return dtype == DAL3_BF16 ? dec_lp<BF16>(w, pts, c_in, B, N, gbias, logits, mask, s)
                              : dec_lp<FP16>(w, pts, c_in, B, N, gbias, logits, mask, s);
}
#endif
#if LP_PART & 1
hipError_t launch_point_head_lp(int dtype, int head_kind, const PointHeadLpW& w, BCN x, int c_in, int B, int M,
                                float* feat, const int32_t* distinct, hipStream_t s) {
    const bool bf = dtype == DAL3_BF16;
    // feat = 0 (NaN rows for items with a non-finite input, include/dal3.h)
    hipError_t e0 = launch_nonfinite_rows(x, B, M, c_in, feat, 512, s);
    if (e0 != hipSuccess) return e0;
    switch (head_kind) {
        case DAL3_HEAD_STATIC_BOX_EST:
            return bf ? head_lp<BF16, 2, 128, 128, 256>(w, x, c_in, B, M, feat, distinct, s)
                      : head_lp<FP16, 2, 128, 128, 256>(w, x, c_in, B, M, feat, distinct, s);
        case DAL3_HEAD_POINT_EMB:
            return bf ? head_lp<BF16, 2, 64, 128, 256>(w, x, c_in, B, M, feat, distinct, s)
                      : head_lp<FP16, 2, 64, 128, 256>(w, x, c_in, B, M, feat, distinct, s);
        case DAL3_HEAD_BOX_EMB:
            return bf ? head_lp<BF16, 4, 64, 64, 128>(w, x, c_in, B, M, feat, distinct, s)
                      : head_lp<FP16, 4, 64, 64, 128>(w, x, c_in, B, M, feat, distinct, s);
        default:
            return hipErrorInvalidValue;
    }
}
#endif
