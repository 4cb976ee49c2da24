// dal3_pointmlp.hip — the per-point shared-MLP kernels (fp32 MFMA, activations in registers).
//
//   ins_seg_encode : conv1..conv5 + BN + ReLU, channel-wise max over N fused into conv5's epilogue
//                    (static_model.py:279-284) -> g (B,1024). The (B,1024,N) tensor never exists.
//   ins_seg_decode : recomputes conv1-2 from the points, then dconv1 (per-point 64->512 part; the
//                    global-feature part arrives as the per-crop vector gb) streamed in 32-channel
//                    chunks straight into dconv2's K loop, dconv3, dconv4, dconv5 + the '<' mask
//                    (static_model.py:286-295, :59) -> logits (B,N,2), mask (B,N).
//   point_head     : conv1..conv4 + max over M (static_model.py:330-334, dynamic_model.py:241-245,
//                    278-282) -> feat (B,512).
//
// One wave owns T tiles of 32 points; waves never talk to each other (no LDS, no barrier): the
// cross-wave max is an atomic on g / feat. Weights stream from L2 as 1-KiB coalesced dwordx4
// fragments shared by all waves of the chip.
#include <stdlib.h>

#include "dal3_device.h"
#include "dal3_kernels.h"

// Diagnostic build only (-DDAL3_STAMP): s_memtime stamps of the decode kernel's phases, written to a
// buffer of their own (dal3_debug_set_stamps). No stamp executes in the shipped library.
#ifdef DAL3_STAMP
__device__ long long* g_stamps = nullptr;
extern "C" int dal3_debug_set_stamps(void* p) {
    return (int)hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &p, sizeof(p));
}
#define STAMP(k)                                                                                  \
    do {                                                                                          \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        unsigned long long t_;                                                                    \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");               \
        __builtin_amdgcn_sched_barrier(0);                                                        \
        if (lane == 0 && blockIdx.x < 2048) g_stamps[(blockIdx.x * DAL3_WG_WAVES + wave) * 8 + (k)] = t_; \
    } while (0)
#else
#define STAMP(k)
#endif

// ------------------------------------------------------------------------------------------------
// Every kernel reads its MFMA weights as ONE fragment stream in consumption order (see InsSegW /
// PointHeadW) through one prefetch ring that never restarts between layers.
#ifndef DAL3_ENC_LDS_MAX
#define DAL3_ENC_LDS_MAX 1              // 0: every wave's per-tile maxima straight to global memory (A/B builds)
#endif
template <int T>
__global__ __launch_bounds__(64 * DAL3_WG_WAVES) void ins_seg_encode_kernel(InsSegW w, BCN pts, int c_in, int n_pts,
                                                             int tiles_per_item, float* __restrict__ g) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int h = lane >> 5;
    const int64_t b = blockIdx.x / tiles_per_item;
    const int n0 = ((blockIdx.x % tiles_per_item) * DAL3_WG_WAVES + wave) * (32 * T);
    __shared__ float s_b5[1024];                       // conv5's folded bias, read by the max epilogue
    // The workgroup's maxima meet in LDS first (round 4): every wave used to send its 1024 per-channel maxima to the
    // crop's row of g with global atomics — 4 KiB of atomic traffic per 64 points, 268 MB per 4096 x 1024 launch for a
    // 16.8 MB result (16 x). The bit patterns are >= 0 after the ReLU, so the table starts at 0 and an
    // integer max is the float max; one global atomicMax per channel and WORKGROUP follows (none where the value is 0:
    // the caller zero-fills g).
    __shared__ int s_max[DAL3_ENC_LDS_MAX ? 1024 : 1];
    for (int i = threadIdx.x; i < 1024; i += 64 * DAL3_WG_WAVES) {
        s_b5[i] = w.b5[i];
        if (DAL3_ENC_LDS_MAX) s_max[i] = 0;
    }
    __syncthreads();
    if (n0 < n_pts) {                                  // (no early return: every wave meets the barrier below)
        WRing<DAL3_PF> ring;
        ring.init(w.enc_stream, lane);                 // conv2 | conv3 | conv4 | conv5
        f32x16 bias = tile_from_channels(w.b2, h);
        float in[T][2];
        load_points<2, T>(pts, b, n0, n_pts, c_in, in, lane);
        f32x16 x1[T][2], x2[T][2], x3[T][2], x4[T][4];
        first_layer<2, 2, T>(w.w1, w.b1, in, x1, lane);
        mlp_layer_ring<2, 2, T>(ring, w.b2, w.b3, bias, x1, x2, lane);
        mlp_layer_ring<2, 2, T>(ring, w.b3, w.b4, bias, x2, x3, lane);
        mlp_layer_ring<2, 4, T>(ring, w.b4, w.b4, bias, x3, x4, lane);
        conv_max_layer<4, T>(ring, s_b5, x4, DAL3_ENC_LDS_MAX ? reinterpret_cast<float*>(s_max) : g + b * 1024, 32, lane);
    }
    if (DAL3_ENC_LDS_MAX) {
        __syncthreads();
        int* gi = reinterpret_cast<int*>(g + b * 1024);
        for (int c = threadIdx.x; c < 1024; c += 64 * DAL3_WG_WAVES) {
            const int v = s_max[c];
            if (v > 0) atomicMax(gi + c, v);
        }
    }
}

// ------------------------------------------------------------------------------------------------
template <int T>
__global__ __launch_bounds__(64 * DAL3_WG_WAVES) void ins_seg_decode_kernel(InsSegW w, BCN pts, int c_in, int n_pts,
                                                             int tiles_per_item, const float* __restrict__ gbias,
                                                             float* __restrict__ logits, uint8_t* __restrict__ mask) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int h = lane >> 5;
    // Which (crop, tile) this workgroup takes. The hardware deals consecutive workgroup ids round-robin over the 8 XCDs,
    // each with an L2 of its own; with id -> (id / tiles_per_item, id % tiles_per_item) the 8 workgroups of a 1024-point
    // crop land on 8 different XCDs and each XCD's L2 fetches the crop's 2-KiB dconv1 term `gbias` from memory for itself:
    // 8 x 8.4 MB of the 130.7 MB this kernel read per 4096 x 1024 launch against 58.7 MB of points + terms (VERDICT r5 #6;
    // profiles/r06_pmc_decode_traffic.txt). Round 6: XCD x (ids = x mod 8) takes a CONTIGUOUS range of logical blocks, so a
    // crop's tiles share one L2. Pure index arithmetic; per-tile work is independent: bit-identical results.
#ifndef DAL3_DEC_XCD
#define DAL3_DEC_XCD 1
#endif
    unsigned blk = blockIdx.x;
    if (DAL3_DEC_XCD) {
        const unsigned per = gridDim.x >> 3, rem = gridDim.x & 7u, x = blockIdx.x & 7u;
        blk = x * per + (x < rem ? x : rem) + (blockIdx.x >> 3);
    }
    const int64_t b = blk / tiles_per_item;
    const int n0 = ((blk % tiles_per_item) * DAL3_WG_WAVES + wave) * (32 * T);
    if (n0 >= n_pts) return;

    STAMP(0);
    // stream: conv2 | dconv1a(0) | { dconv1a(c+1), dconv2(c) } c = 0..15 | dconv3 | dconv4
    WRing<DAL3_PF> ring;
    ring.init(w.dec_stream, lane);
    f32x16 bias = tile_from_channels(w.b2, h);
    const float* gb = gbias + b * 512;                 // W1g . g + b1' of this crop
    // a crop with a non-finite coordinate: its pooled feature, hence this term, is NaN (dal3_device.h); as in the
    // reference every logit of the crop is then NaN and its mask empty (static_model.py:284-289, :59)
    const bool crop_bad = bits_nonfinite(gb[0]);

    f32x16 x2[T][2];
    {
        float in[T][2];
        load_points<2, T>(pts, b, n0, n_pts, c_in, in, lane);
        f32x16 x1[T][2];
        first_layer<2, 2, T>(w.w1, w.b1, in, x1, lane);
        mlp_layer_ring<2, 2, T>(ring, w.b2, gb, bias, x1, x2, lane);       // leaves bias = gb chunk 0
    }

    // dconv1 (512 outputs, 16 chunks of 32) streamed into dconv2 (256 outputs = 8 accumulator tiles).
    // dconv1 chunk c+1 is computed (into the other t buffer) BEFORE dconv2 consumes chunk c, and the
    // ReLU of chunk c rides under those MFMAs: no MFMA -> VALU -> MFMA bubble at the chunk seam.
    f32x16 a2[T][8];
#pragma unroll
    for (int mt = 0; mt < 8; ++mt) {
        const f32x16 b2v = tile_from_channels(w.db2 + 32 * mt, h);
#pragma unroll
        for (int j = 0; j < T; ++j) a2[j][mt] = b2v;
    }
    STAMP(1);
    f32x16 tA[T], tB[T];
    auto dconv2_part = [&](const f32x16 (&t)[T]) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {                 // fragment i of the chunk: out-tile i/4, q = i%4
            ring_batch_wait<DAL3_PF>(ring, i);
            const f32x4 a = ring.slot[i % DAL3_PF];
            ring.slot[i % DAL3_PF] = ring.fetch();
#ifdef DAL3_ABLATE_WINDOW
            ring.soff &= ~0x2000u;
#endif
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int j = 0; j < T; ++j) a2[j][i / 4] = mfma32(a[e], t[j][4 * (i % 4) + e], a2[j][i / 4]);
            }
            DAL3_SCHED_FENCE();
        }
    };
#pragma unroll
    for (int j = 0; j < T; ++j) tA[j] = bias;
    mma_block_ring<2, T>(ring, x2, tA, [&](int i) {
        if (i == 0) bias = tile_from_channels(gb + 32, h);
    });
    for (int c = 0; c < 16; c += 2) {
#pragma unroll
        for (int j = 0; j < T; ++j) tB[j] = bias;
        mma_block_ring<2, T>(ring, x2, tB, [&](int i) {                // chunk c+1; ReLU(chunk c) in its shadow
            if (i == 0) bias = tile_from_channels(gb + 32 * ((c + 2) & 15), h);
#pragma unroll
            for (int j = 0; j < T; ++j) {              // two registers per fragment group
                tA[j][2 * i] = relu1(tA[j][2 * i]);
                tA[j][2 * i + 1] = relu1(tA[j][2 * i + 1]);
            }
        });
        dconv2_part(tA);
#pragma unroll
        for (int j = 0; j < T; ++j) tA[j] = bias;
        // chunk c+2 (on the last round its 8 fragments are the stream's zero filler; the result is unused)
        mma_block_ring<2, T>(ring, x2, tA, [&](int i) {
            if (i == 0)
                bias = c + 2 < 16 ? tile_from_channels(gb + 32 * ((c + 3) & 15), h) : tile_from_channels(w.db3, h);
#pragma unroll
            for (int j = 0; j < T; ++j) {
                tB[j][2 * i] = relu1(tB[j][2 * i]);
                tB[j][2 * i + 1] = relu1(tB[j][2 * i + 1]);
            }
        });
        dconv2_part(tB);
    }
    STAMP(2);
#pragma unroll
    for (int j = 0; j < T; ++j) {
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) a2[j][mt] = relu16(a2[j][mt]);
    }

    f32x16 y3[T][4], y4[T][4];
    mlp_layer_ring<8, 4, T>(ring, w.db3, w.db4, bias, a2, y3, lane);
    mlp_layer_ring<4, 4, T>(ring, w.db4, w.db4, bias, y3, y4, lane);

    STAMP(3);
    // dconv5 (128 -> 2, no BN/ReLU) on the VALU: each lane holds 64 of its point's 128 channels
    float l0[T], l1[T];
#pragma unroll
    for (int j = 0; j < T; ++j) l0[j] = l1[j] = 0.0f;
#pragma unroll
    for (int kt = 0; kt < 4; ++kt) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 wa = *reinterpret_cast<const f32x4*>(w.dw5 + 32 * kt + 8 * q + 4 * h);
            const f32x4 wb = *reinterpret_cast<const f32x4*>(w.dw5 + 128 + 32 * kt + 8 * q + 4 * h);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int j = 0; j < T; ++j) {
                    l0[j] = fmaf(wa[e], y4[j][kt][4 * q + e], l0[j]);
                    l1[j] = fmaf(wb[e], y4[j][kt][4 * q + e], l1[j]);
                }
            }
        }
    }
    const float bias0 = w.db5[0], bias1 = w.db5[1];
#pragma unroll
    for (int j = 0; j < T; ++j) {
        float s0 = l0[j] + __shfl_xor(l0[j], 32) + bias0;
        float s1 = l1[j] + __shfl_xor(l1[j], 32) + bias1;
        if (crop_bad) s0 = s1 = __int_as_float(DAL3_QNAN_BITS);
        const int n = n0 + 32 * j + (lane & 31);
        if (h == 0 && n < n_pts) {
            f32x2 o;
            o[0] = s0;
            o[1] = s1;
            *reinterpret_cast<f32x2*>(logits + (b * n_pts + n) * 2) = o;
            mask[b * n_pts + n] = (!crop_bad && s0 < s1) ? 1 : 0;      // strict '<': ties are background
        }
    }
    STAMP(4);
}

// ------------------------------------------------------------------------------------------------
template <int KS, int C1, int C2, int C3, int T>
__global__ __launch_bounds__(64 * DAL3_WG_WAVES) void point_head_kernel(PointHeadW w, BCN x, int c_in, int n_pts,
                                                         int tiles_per_item, float* __restrict__ feat,
                                                         const int32_t* __restrict__ distinct) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int h = lane >> 5;
    // item-minor block order: workgroups are dealt round-robin over the 8 XCDs, and with duplicate skipping only
    // an item's FIRST tiles carry work; (item, tile) = (id % B, id / B) spreads those over every XCD (with
    // item-major order and 4 tiles per item they all landed on 2 of the 8 XCDs: 4.6x slower)
    // ONE wave per workgroup: a wave that has nothing to do (below) exits at once, and inside a 4-wave workgroup its
    // SIMD would then idle until the slowest sibling is done (the workgroup keeps the CU: one wave per SIMD at this
    // register count) — 12 % of this kernel's time on the bench mix of crops.
    const int wg_waves = blockDim.x >> 6;
    const int n_items = gridDim.x / tiles_per_item;
    const int64_t b = blockIdx.x % n_items;
    const int n0 = ((blockIdx.x / n_items) * wg_waves + wave) * (32 * T);
    __shared__ float s_b4[512];                        // conv4's folded bias, read by the max epilogue
    for (int i = threadIdx.x; i < 512; i += blockDim.x) s_b4[i] = w.b4[i];
    __syncthreads();
    // distinct[b] (optional): only the first distinct[b] points of the item differ, the rest are duplicates of
    // them (gather: count < M tops up with copies; count == 0 is an all-zero row). The head is per-point work
    // followed by a max over points, so the duplicates cannot change the result and are not computed.
    if (distinct) {
        const int d = distinct[b];
        n_pts = d <= 0 ? 1 : (d < n_pts ? d : n_pts);
    }
    if (n0 >= n_pts) return;

    WRing<DAL3_PF> ring;
    ring.init(w.stream, lane);                         // conv2 | conv3 | conv4
    f32x16 bias = tile_from_channels(w.b2, h);
    f32x16 x3[T][C3 / 32];
    {
        float in[T][KS];
        load_points<KS, T>(x, b, n0, n_pts, c_in, in, lane);
        f32x16 x1[T][C1 / 32], x2[T][C2 / 32];
        first_layer<KS, C1 / 32, T>(w.w1, w.b1, in, x1, lane);
        mlp_layer_ring<C1 / 32, C2 / 32, T>(ring, w.b2, w.b3, bias, x1, x2, lane);
        mlp_layer_ring<C2 / 32, C3 / 32, T>(ring, w.b3, w.b3, bias, x2, x3, lane);
    }
    conv_max_layer<C3 / 32, T>(ring, s_b4, x3, feat + b * 512, 16, lane);
}

// ------------------------------------------------------------------------------------------------
// The point heads on a WORKLIST of live tiles (round 3). With the device sampler only the first min(count, M) object
// points of an item are distinct, so most items have tiles that hold nothing but copies; point_head_kernel above
// launches a one-wave workgroup for every (item, tile) and lets the dead ones exit — 26 % of the grid on the bench's mix
// of crops — and every live one pays its own launch, bias preload and cold weight ring for ONE 32-point tile (~90 us).
// Here the zero-fill of `feat` also turns `distinct` into the compacted list of (item, tile, points that count) entries, and
// persistent one-wave workgroups (as many as the chip holds at this register count) take entries from a device-side
// cursor: no dead workgroups, no relaunch, the weight ring runs on cyclically from tile to tile, and the next entry
// and its points are fetched while the current tile is computed. Per-point arithmetic and the atomicMax combine are
// those of point_head_kernel: bit-identical results (tests/test_gpu_parity.py). (The 16-bit heads were given the
// same list, in groups of 256 points, and timed both ways: 3.52 ms per C3 step either way — their kernel is persistent
// already and a group of copies is a uniform 1-us skip there — so they keep walking (item, group) pairs.)
//   ctl[0] = n_live, ctl[1] = cursor (zero), list[i] = {item, tile, n_eff, 0}: written by nonfinite_rows_kernel
//   (dal3_misc.hip), the launch that zero-fills `feat` in front of this kernel anyway
// Round 5: a wave takes RUNS of consecutive entries (the list is item-major, so a run is mostly one item's tiles) and
// keeps the item's 512 channel maxima in a private LDS row across the run; the global atomicMax is issued once per
// channel when the run leaves the item, not once per channel and TILE (rocprofv3 WRITE_SIZE: 108 MB per launch for an
// 8.4-MB result at 4096 x 512 object points, profiles/r04_pmc.json). Runs are HEAD_RUN entries long while plenty of
// work is left and shrink towards single entries at the end of the list (a run is the unit of the dynamic schedule:
// long runs at the end would leave waves idle for up to a run's time; guided self-scheduling: half of an even share of
// what is left). max is order-independent: bit-identical results.
#ifndef HEAD_RUN
#define HEAD_RUN 6
#endif
template <int KS, int C1, int C2, int C3>
__global__ __launch_bounds__(64) void point_head_pers_kernel(PointHeadW w, BCN x, int c_in, float* __restrict__ feat,
                                                             uint32_t* __restrict__ ctl, const u32x4* __restrict__ list) {
    constexpr int T = 1;
    const int lane = threadIdx.x & 63;
    const int h = lane >> 5;
    __shared__ float s_b4[512];                        // conv4's folded bias, read by the max epilogue
    __shared__ int s_run[512];                         // the current item's channel maxima (bit patterns, >= 0)
    for (int i = threadIdx.x; i < 512; i += 64) {
        s_b4[i] = w.b4[i];
        s_run[i] = 0;
    }
    __syncthreads();
    const uint32_t n_live = ctl[0];
    // the schedule's cursor counts ENTRIES; the first gridDim.x * run0 entries are dealt out by block index. The first
    // run follows the same guided rule as every later one (round 6): half of an even share of the list, at most
    // HEAD_RUN, at least 1 — so a job of about one tile per wave (the reference's eval batch: 64 crops x 512 object
    // points = 1024 tiles) gives every wave ONE tile instead of six to a sixth of the chip (0.48 ms -> 0.1 ms there).
    uint32_t run0 = n_live / (2u * gridDim.x);
    run0 = run0 < 1u ? 1u : (run0 > HEAD_RUN ? HEAD_RUN : run0);
    const uint32_t first_free = gridDim.x * run0;
    uint32_t cur = blockIdx.x * run0, run_end = cur + run0;             // this wave's run: entries [cur, run_end)
    if (cur >= n_live) return;

    constexpr uint32_t STREAM_BYTES =
        ((C2 / 32) * (C1 / 32) + (C3 / 32) * (C2 / 32) + 16 * (C3 / 32)) * 4u * 1024u;   // conv2 | conv3 | conv4 fragments
    WRing<DAL3_PF, true> ring;
    ring.init(w.stream, lane, STREAM_BYTES);
    f32x16 bias = tile_from_channels(w.b2, h);
    u32x4 e = list[cur];
    float in[T][KS];
    load_points<KS, T>(x, (int64_t)e[0], (int)e[1] * 32, (int)e[2], c_in, in, lane);
    auto flush = [&](int64_t b) {                      // the finished item's maxima -> feat (nothing where the value is 0)
        int* gi = reinterpret_cast<int*>(feat + b * 512);
#pragma unroll
        for (int c = lane; c < 512; c += 64) {
            const int v = s_run[c];
            if (v > 0) atomicMax(gi + c, v);
            s_run[c] = 0;
        }
    };
    for (;;) {
        const int64_t b = (int64_t)e[0];
        // the run's last entry asks for the next run: one returning atomic, issued a whole layer before its result is
        // looked at (by then it is older than everything the ring has in flight, so the wait for it costs the ring
        // nothing). Its length: HEAD_RUN while more than 8 runs per wave are left behind the cursor, else 1.
        const bool last_of_run = cur + 1 >= run_end || cur + 1 >= n_live;
        uint32_t nxt = 0, take = 1;
        if (last_of_run) {
            const uint32_t left = n_live > run_end ? n_live - run_end : 0u;           // (behind this wave's own run: a proxy for the cursor)
            take = left / (2u * gridDim.x);                                           // guided: half of an even share of what is left
            take = take < 1u ? 1u : (take > HEAD_RUN ? HEAD_RUN : take);
            if (lane == 0) nxt = atomicAdd(&ctl[1], take);
        }
        f32x16 x3[T][C3 / 32];
        float in_n[T][KS];
        uint32_t ncur, nend;
        {
            f32x16 x1[T][C1 / 32], x2[T][C2 / 32];
            first_layer<KS, C1 / 32, T>(w.w1, w.b1, in, x1, lane);
            mlp_layer_ring<C1 / 32, C2 / 32, T>(ring, w.b2, w.b3, bias, x1, x2, lane);
            if (last_of_run) {
                ncur = (uint32_t)__builtin_amdgcn_readfirstlane((int)nxt) + first_free;
                nend = ncur + take;
            } else {
                ncur = cur + 1;
                nend = run_end;
            }
            const uint32_t nclamp = ncur < n_live ? ncur : cur;        // (past the end: re-read this tile's entry, unused)
            e = list[nclamp];
            mlp_layer_ring<C2 / 32, C3 / 32, T>(ring, w.b3, w.b2, bias, x2, x3, lane);   // leaves bias = conv2's tile 0
            load_points<KS, T>(x, (int64_t)e[0], (int)e[1] * 32, (int)e[2], c_in, in_n, lane);
        }
        conv_max_layer<C3 / 32, T>(ring, s_b4, x3, reinterpret_cast<float*>(s_run), 16, lane);
        const bool more = ncur < n_live;
        if (!more || (int64_t)e[0] != b) flush(b);     // (the wave's own LDS operations are in order: no barrier)
        if (!more) break;
        cur = ncur;
        run_end = nend;
#pragma unroll
        for (int k = 0; k < KS; ++k) in[0][k] = in_n[0][k];
    }
}

// ------------------------------------------------------------------------------------------------
// One layer, any Cin = 32*KT (or a raw first layer when kt == 0 && ks > 0) -> y (B,N,Cout)
// point-major. Layer-wise test entry (dal3_shared_mlp_layer); not on the production path.
__global__ __launch_bounds__(256) void generic_layer_kernel(const f32x4* __restrict__ wf, const float* __restrict__ w1,
                                                            const float* __restrict__ bias, int kt_n, int ks_n, int mt_n,
                                                            int relu, BCN x, int c_in, int n_pts, int tiles_per_item,
                                                            float* __restrict__ y) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int h = lane >> 5;
    const int64_t b = blockIdx.x / tiles_per_item;
    const int n0 = ((blockIdx.x % tiles_per_item) * 4 + wave) * 32;
    if (n0 >= n_pts) return;
    int n = n0 + (lane & 31);
    const bool valid = n < n_pts;
    n = valid ? n : n_pts - 1;
    const float* p = x.data + b * x.sb + (int64_t)n * x.sn;
    const int c_out = 32 * mt_n;
    for (int mt = 0; mt < mt_n; ++mt) {
        f32x16 acc = tile_from_channels(bias + 32 * mt, h);
        if (ks_n > 0) {
            for (int s = 0; s < ks_n; ++s) {
                const int c = 2 * s + h;
                const float bv = c < c_in ? p[c * x.sc] : 0.0f;
                acc = mfma32(w1[(mt * ks_n + s) * 64 + lane], bv, acc);
            }
        } else {
            for (int kt = 0; kt < kt_n; ++kt) {
                for (int q = 0; q < 4; ++q) {
                    const f32x4 a = wf[((mt * kt_n + kt) * 4 + q) * 64 + lane];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int c = 32 * kt + 8 * q + 4 * h + e;
                        acc = mfma32(a[e], p[c * x.sc], acc);
                    }
                }
            }
        }
        if (relu) acc = relu16(acc);
        if (valid) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 o;
                o[0] = acc[4 * q + 0];
                o[1] = acc[4 * q + 1];
                o[2] = acc[4 * q + 2];
                o[3] = acc[4 * q + 3];
                *reinterpret_cast<f32x4*>(y + (b * n_pts + n) * c_out + 32 * mt + 8 * q + 4 * h) = o;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
static inline int tiles_per_item(int n_pts, int T) {
    return (n_pts + 32 * DAL3_WG_WAVES * T - 1) / (32 * DAL3_WG_WAVES * T);
}

// Which family runs: the latency kernels (dal3_latency.hip: a 16-wave workgroup per tile, one per CU) while the job has
// at most DAL3_LAT_MAX_TILES tiles (512 = two rounds of workgroups on the 256 CUs), the throughput kernels above that.
// Both give the same bits. A pure function of the job and of the caller's flags (dal3.h: DAL3_BCN_NO_SMALL_JOB_KERNELS):
// no environment variable, no process-wide switch.
#ifndef DAL3_LAT_MAX_TILES
#define DAL3_LAT_MAX_TILES 512
#endif
#ifndef DAL3_LAT_MAX_TILES_ENC
#define DAL3_LAT_MAX_TILES_ENC DAL3_LAT_MAX_TILES
#endif
#ifndef DAL3_LAT_MAX_TILES_DEC
#define DAL3_LAT_MAX_TILES_DEC DAL3_LAT_MAX_TILES
#endif
#ifndef DAL3_LAT_MAX_TILES_HEAD
#define DAL3_LAT_MAX_TILES_HEAD DAL3_LAT_MAX_TILES
#endif
bool lat_use(int64_t tiles, int flags, int64_t max_tiles) { return !(flags & DAL3_BCN_NO_SMALL_JOB_KERNELS) && tiles <= max_tiles; }

hipError_t launch_ins_seg_encode(const InsSegW& w, BCN pts, int c_in, int B, int N, float* g, hipStream_t s) {
    if (lat_use((int64_t)B * ((N + 31) / 32), pts.flags, DAL3_LAT_MAX_TILES_ENC)) return launch_ins_seg_encode_lat(w, pts, c_in, B, N, g, s);
    constexpr int T = DAL3_ENC_T;
    // A wave runs its T tiles through the whole encoder one after the other (~150 us per tile at 2.4 GHz). When the
    // job cannot fill the chip's 1024 SIMDs anyway (small eval batches), one tile per wave halves that serial
    // chain; per-point arithmetic and the atomicMax combine are the same, so the result is bit-identical.
    if (T > 1 && (int64_t)B * N <= 1024 * 64) {
        const int tpi1 = tiles_per_item(N, 1);
        hipLaunchKernelGGL(ins_seg_encode_kernel<1>, dim3((unsigned)((int64_t)B * tpi1)), dim3(64 * DAL3_WG_WAVES), 0, s, w, pts, c_in, N, tpi1, g);
        return hipGetLastError();
    }
    const int tpi = tiles_per_item(N, T);
    hipLaunchKernelGGL(ins_seg_encode_kernel<T>, dim3((unsigned)((int64_t)B * tpi)), dim3(64 * DAL3_WG_WAVES), 0, s, w, pts, c_in, N, tpi, g);
    return hipGetLastError();
}

hipError_t launch_ins_seg_decode(const InsSegW& w, BCN pts, int c_in, int B, int N, const float* gbias,
                                 float* logits, uint8_t* mask, hipStream_t s) {
    if (lat_use((int64_t)B * ((N + 31) / 32), pts.flags, DAL3_LAT_MAX_TILES_DEC)) return launch_ins_seg_decode_lat(w, pts, c_in, B, N, gbias, logits, mask, s);
    constexpr int T = DAL3_DEC_T;
    const int tpi = tiles_per_item(N, T);
    hipLaunchKernelGGL(ins_seg_decode_kernel<T>, dim3((unsigned)((int64_t)B * tpi)), dim3(64 * DAL3_WG_WAVES), 0, s, w, pts, c_in, N, tpi,
                       gbias, logits, mask);
    return hipGetLastError();
}

// bytes of the worklist a launch_point_head of (B items, M points) may need: ctl (2 words, padded) + one entry per tile
size_t point_head_worklist_bytes(int B, int M) { return 256 + (size_t)B * ((M + 31) / 32) * sizeof(u32x4); }

// persistent one-wave workgroups the CURRENT device holds at this kernel's register count (one wave per SIMD): 4 per
// CU. Asked per call: hipGetDevice + hipDeviceGetAttribute are host-side table lookups (~0.1 us), and a process that
// drives several devices gets each one's own count.
static int head_slots() {
    int dev = 0, v = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0)
        v = 256;
    return 4 * v;
}

hipError_t launch_point_head(int head_kind, const PointHeadW& w, BCN x, int c_in, int B, int M, float* feat,
                             const int32_t* distinct, hipStream_t s, void* worklist, size_t worklist_bytes) {
    constexpr int T = DAL3_HEAD_T;
    const int tpi = (M + 32 * T - 1) / (32 * T);       // one-wave workgroups
    const bool lat = lat_use((int64_t)B * ((M + 31) / 32), x.flags, DAL3_LAT_MAX_TILES_HEAD);
    const bool pers = !lat && worklist && worklist_bytes >= point_head_worklist_bytes(B, M) && !(x.flags & DAL3_BCN_NO_WORKLIST) && T == 1;
    // feat = 0 (NaN rows for items with a non-finite input, dal3.h) and, for the persistent kernel, the worklist
    hipError_t e0 = launch_nonfinite_rows(x, B, M, c_in, feat, 512, s, distinct, pers ? worklist : nullptr);
    if (e0 != hipSuccess) return e0;
    if (lat) return launch_point_head_lat(head_kind, w, x, c_in, B, M, feat, distinct, s);
    if (pers) {
        uint32_t* ctl = static_cast<uint32_t*>(worklist);
        u32x4* list = reinterpret_cast<u32x4*>(static_cast<char*>(worklist) + 256);
        const int64_t tiles = (int64_t)B * tpi;            // (an upper bound: the list holds the live tiles only; a wave beyond it exits)
        const int64_t slots = head_slots();
        const dim3 grid((unsigned)(tiles < slots ? tiles : slots)), block(64);
        switch (head_kind) {
            case 1:
                hipLaunchKernelGGL((point_head_pers_kernel<2, 128, 128, 256>), grid, block, 0, s, w, x, c_in, feat, ctl, list);
                break;
            case 2:
                hipLaunchKernelGGL((point_head_pers_kernel<2, 64, 128, 256>), grid, block, 0, s, w, x, c_in, feat, ctl, list);
                break;
            case 3:
                hipLaunchKernelGGL((point_head_pers_kernel<4, 64, 64, 128>), grid, block, 0, s, w, x, c_in, feat, ctl, list);
                break;
            default:
                return hipErrorInvalidValue;
        }
        return hipGetLastError();
    }
    const dim3 grid((unsigned)((int64_t)B * tpi)), block(64);
    switch (head_kind) {
        case 1:  // static box_est 3 -> 128 -> 128 -> 256 -> 512
            hipLaunchKernelGGL((point_head_kernel<2, 128, 128, 256, T>), grid, block, 0, s, w, x, c_in, M, tpi, feat, distinct);
            break;
        case 2:  // point_emb 4 -> 64 -> 128 -> 256 -> 512
            hipLaunchKernelGGL((point_head_kernel<2, 64, 128, 256, T>), grid, block, 0, s, w, x, c_in, M, tpi, feat, distinct);
            break;
        case 3:  // box_emb 8 -> 64 -> 64 -> 128 -> 512
            hipLaunchKernelGGL((point_head_kernel<4, 64, 64, 128, T>), grid, block, 0, s, w, x, c_in, M, tpi, feat, distinct);
            break;
        default:
            return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_generic_layer(const f32x4* wf, const float* w1, const float* bias, int kt_n, int ks_n, int mt_n,
                                int relu, BCN x, int c_in, int B, int N, float* y, hipStream_t s) {
    const int tpi = (N + 127) / 128;
    hipLaunchKernelGGL(generic_layer_kernel, dim3((unsigned)((int64_t)B * tpi)), dim3(256), 0, s, wf, w1, bias, kt_n,
                       ks_n, mt_n, relu, x, c_in, N, tpi, y);
    return hipGetLastError();
}
