// dal3_device.h — device-side building blocks shared by the gfx950 kernels.
//
// Orientation used everywhere ("Kernel-design notes"): a shared-MLP layer is
//   Y(Cout x P) = W'(Cout x Cin) . X(Cin x P)
// with CHANNELS on the MFMA rows and POINTS on the MFMA columns (= lanes). With the 32x32 C/D
// layout of gfx950 (col = lane&31, row = (reg&3) + 8*(reg>>2) + 4*(lane>>5)) the accumulator of
// layer k is then directly the B operand of layer k+1 for v_mfma_f32_32x32x2_f32 (lane half h
// supplies k = h): accumulator register r of channel-tile t is k-step 16t+r, and the A fragment
// of that k-step holds the two weight columns 32t + chan(r,0) and 32t + chan(r,1). A whole
// per-point MLP therefore stays in registers: no LDS round trip, no lane movement.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// logical (B,C,N) view with element strides (mirrors dal3_bcn of include/dal3.h). dtype: how the values are STORED
// (0 = fp32, 1 = bf16, 2 = fp16, the DAL3_* codes); 16-bit storage is widened exactly on load, in place — no fp32 copy
// of the points is made (BASELINE.json configs C3 / C5: "bf16 storage"). `data` is typed float* for the common case.
struct BCN {
    const float* data;
    int64_t sb, sc, sn;
    int dtype;
    int flags;      // DAL3_BCN_* dispatch hints (host side only; the kernels never read it)
};
__device__ __forceinline__ float widen_bf16(uint16_t v) { return __uint_as_float((uint32_t)v << 16); }
__device__ __forceinline__ float widen_f16(uint16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
// element `idx` (in elements of the stored type) of a BCN view, as fp32
__device__ __forceinline__ float bcn_value(const BCN& x, int64_t idx) {
    if (x.dtype == 0) return x.data[idx];
    const uint16_t v = reinterpret_cast<const uint16_t*>(x.data)[idx];
    return x.dtype == 1 ? widen_bf16(v) : widen_f16(v);
}

// channel (within a 32-channel tile) held by accumulator register r in lane half h
__host__ __device__ constexpr int tile_chan(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// accumulator tile initialised with the per-channel vector v[0..31] (bias, or a per-crop term):
// register 4q+e of half h is channel 8q + 4h + e, i.e. four contiguous floats per q.
__device__ __forceinline__ f32x16 tile_from_channels(const float* __restrict__ v, int h) {
    f32x16 t;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(v + 8 * q + 4 * h);
        t[4 * q + 0] = x[0];
        t[4 * q + 1] = x[1];
        t[4 * q + 2] = x[2];
        t[4 * q + 3] = x[3];
    }
    return t;
}

// Non-finite inputs (include/dal3.h "Non-finite coordinates"): a crop with a NaN / Inf coordinate gets the quiet-NaN
// pattern in every channel of its pooled feature BEFORE the encoder runs (nonfinite_rows_kernel, dal3_misc.hip) —
// as a signed integer that pattern lies above every finite value and +inf, so the encoder's atomicMax leaves it in
// place; the per-crop FC carries it into the crop's dconv1 term, where the decode kernels look for it.
#define DAL3_QNAN_BITS 0x7FC00000
__device__ __forceinline__ bool bits_nonfinite(float v) {   // NaN or +-Inf, on the bit pattern (the shared-MLP kernels
    return (__float_as_uint(v) & 0x7F800000u) == 0x7F800000u;   // are built -fno-honor-nans: no float compare here)
}

// max(x, 0) as ONE integer v_max_i32 on the bit pattern (negatives and -0.0 have the sign bit set ->
// +0; positives unchanged). fmaxf() on an MFMA result makes hipcc emit a canonicalising v_max_f32 x,x
// in front of it, doubling the ReLU's VALU cost; an inline-asm v_max_f32 would read the MFMA result
// without the hazard wait states hipcc only inserts for instructions it models (cdna guide 5.7).
__device__ __forceinline__ float relu1(float x) {
    const int b = __float_as_int(x);
    return __int_as_float(b > 0 ? b : 0);
}
__device__ __forceinline__ f32x16 relu16(f32x16 a) {
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = relu1(a[r]);
    return a;
}

// acc[j] += W'(32 x 32*KT) . X[j]   for the T point tiles of this wave.
// wblk: fragment-packed weights of ONE 32-row output tile: [KT][4 q][64 lanes] float4, where
// element e of (kt,q,lane) is W'[row = lane&31][col = 32kt + tile_chan(4q+e, lane>>5)].
template <int KT, int T>
__device__ __forceinline__ void mma_block(const f32x4* __restrict__ wblk, const f32x16 (&X)[T][KT],
                                          f32x16 (&acc)[T], int lane) {
#pragma unroll
    for (int kt = 0; kt < KT; ++kt) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 a = wblk[(kt * 4 + q) * 64 + lane];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int j = 0; j < T; ++j) acc[j] = mfma32(a[e], X[j][kt][4 * q + e], acc[j]);
            }
        }
    }
}

// Rolling prefetch of a fragment stream. The packed layouts put the fragments of the big layers in
// exactly the order they are consumed (conv5 / conv4: [out-tile][k-tile][q]; dconv1a: [chunk][kt][q];
// dconv2: [chunk][out-tile][q]), so a layer is ONE contiguous stream of 1-KiB wave-fragments. The
// ring keeps D of them in flight ahead of the MFMAs, across the runtime loop over output chunks;
// without it every chunk starts with an exposed L2 round trip. The last D fetches run past the
// stream's end into the next section of the blob (blobs carry tail padding); their values are unused.
#ifndef DAL3_SCHED_FENCE
#define DAL3_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

// The stream is read through a buffer descriptor: address = base (4 SGPRs) + lane*16 (one VGPR, never changes) +
// a SCALAR byte offset that advances 1 KiB per fragment with one s_add_u32. With a flat pointer per lane, every
// fragment cost two VALU adds on a 64-bit VGPR pair (hipcc folds the lane into the pointer and strength-reduces it,
// whatever the source says) — on this chip VALU slots are MFMA slots (see below), and those adds were the largest
// single group of VALU instructions in the fp32 kernels. Reads past the stream's end stay inside the 2 GiB window
// of the descriptor (blobs carry tail padding); their values are unused.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
// CYC: the stream is read CYCLICALLY (a persistent wave runs it once per tile: the ring's last D fetches of a tile
// are the first D fragments of the next one, so no tile starts with an exposed L2 round trip); `len` = its bytes.
template <int D, bool CYC = false>
struct WRing {
    __amdgpu_buffer_rsrc_t rsrc;
    uint32_t voff, soff, len;
    f32x4 slot[D];
    __device__ __forceinline__ f32x4 fetch() {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
        soff += 1024;
        if (CYC && soff == len) soff = 0;                  // scalar compare + select
        return __builtin_bit_cast(f32x4, v);
    }
    __device__ __forceinline__ void init(const f32x4* stream, int lane, uint32_t bytes = 0) {
        rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<f32x4*>(stream), 0, 0x7fffffff, 0x00020000);
        voff = (uint32_t)lane * 16u;
        soff = 0;
        len = bytes;
#pragma unroll
        for (int i = 0; i < D; ++i) slot[i] = fetch();
    }
};

// On gfx950 the f32 MFMA shares the vector ALU's issue (tools/ubench/mfma_bubble.hip: every VALU op
// between two MFMAs adds ~6 cycles, every s_waitcnt ~17, a global_load ~1), so the loops keep the
// non-MFMA instruction count down. A lever that used to pay and no longer does: hipcc puts an s_waitcnt vmcnt(N)
// in front of the first use of EVERY fragment; touching the youngest fragment of a batch of DAL3_WAIT_BATCH first
// makes it emit one wait per batch (loads retire in order). With flat loads a batch of 4 was worth ~1 %; since the
// ring reads through a buffer descriptor, waiting for a younger fragment than needed costs more than the waits
// saved (decode 13.15 -> 12.89 ms with the batch at 1), so the default is 1 (= off).
#ifndef DAL3_WAIT_BATCH
#define DAL3_WAIT_BATCH 1
#endif
template <int D, bool CYC = false>
__device__ __forceinline__ void ring_batch_wait(WRing<D, CYC>& ring, int i) {
    static_assert(D % DAL3_WAIT_BATCH == 0, "wait batch must divide the ring depth");
    if (DAL3_WAIT_BATCH > 1 && i % DAL3_WAIT_BATCH == 0)
        asm volatile("" : "+v"(ring.slot[(i + DAL3_WAIT_BATCH - 1) % D][0]));
}

// acc[j] += W'(32 x 32*KT) . X[j], fragments taken from the ring (KT*4 of them, in stream order).
// side(i) runs after fragment i's MFMAs have been issued: VALU work placed there (the previous
// chunk's epilogue, a ReLU of the other buffer) executes in the shadow of those 64-cycle MFMAs
// instead of leaving the matrix pipe idle at a chunk seam. The sched_barriers pin both the fetch
// (D fragments ahead of its use; left alone hipcc sinks every load to just before its first use and
// each group of MFMAs waits out an L2 round trip) and the side jobs.
struct NoSide {
    __device__ __forceinline__ void operator()(int) const {}
};
// SWAP: the operands change places, acc[j] += X[j]^T . W'^T — the TRANSPOSED tile (points on the rows = registers,
// channels on the columns = lanes); the A and B fragment layouts are mirror images, so the same registers serve.
template <int KT, int T, int D, typename Side = NoSide, bool SWAP = false, bool CYC = false>
__device__ __forceinline__ void mma_block_ring(WRing<D, CYC>& ring, const f32x16 (&X)[T][KT], f32x16 (&acc)[T],
                                               Side side = Side()) {
    static_assert((KT * 4) % D == 0, "ring depth must divide the fragments per block");
#pragma unroll
    for (int i = 0; i < KT * 4; ++i) {
        ring_batch_wait<D, CYC>(ring, i);
        const f32x4 a = ring.slot[i % D];
        ring.slot[i % D] = ring.fetch();
#ifdef DAL3_ABLATE_WINDOW   // timing experiment only: every fetch hits the same 8 KiB (L1-resident)
        ring.soff &= ~0x2000u;
#endif
#pragma unroll
        for (int e = 0; e < 4; ++e) {
#pragma unroll
            for (int j = 0; j < T; ++j)
                acc[j] = SWAP ? mfma32(X[j][i / 4][4 * (i % 4) + e], a[e], acc[j]) : mfma32(a[e], X[j][i / 4][4 * (i % 4) + e], acc[j]);
        }
        DAL3_SCHED_FENCE();
        side(i);
        DAL3_SCHED_FENCE();
    }
}

// Y = relu(W' X + b') for a Cin=32*KT -> Cout=32*MT layer whose fragments are next on the kernel's
// weight stream. Two accumulator sets alternate so that the ReLU of tile mt-1 and the bias fetch of
// tile mt+1 ride under the MFMAs of tile mt. bias_cur arrives holding this layer's tile-0 bias and
// leaves holding the first tile's bias of the NEXT layer (b_next), fetched a whole tile ahead.
template <int KT, int MT, int T, int D, bool CYC = false>
__device__ __forceinline__ void mlp_layer_ring(WRing<D, CYC>& ring, const float* __restrict__ b,
                                               const float* __restrict__ b_next, f32x16& bias_cur,
                                               const f32x16 (&X)[T][KT], f32x16 (&Y)[T][MT], int lane) {
    const int h = lane >> 5;
    f32x16 acc[2][T];
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
        for (int j = 0; j < T; ++j) acc[mt & 1][j] = bias_cur;
        mma_block_ring(ring, X, acc[mt & 1], [&](int i) {
            if (i == 0) bias_cur = tile_from_channels(mt + 1 < MT ? b + 32 * (mt + 1) : b_next, h);
            if (mt > 0 && i < 8) {                     // ReLU of the previous tile, two registers per fragment group
#pragma unroll
                for (int j = 0; j < T; ++j) {
                    Y[j][mt - 1][2 * i] = relu1(acc[(mt - 1) & 1][j][2 * i]);
                    Y[j][mt - 1][2 * i + 1] = relu1(acc[(mt - 1) & 1][j][2 * i + 1]);
                }
            }
        });
    }
#pragma unroll
    for (int j = 0; j < T; ++j) Y[j][MT - 1] = relu16(acc[(MT - 1) & 1][j]);
}

// First layer (raw input, Cin <= 2*KS channels in natural order): in[j][s] is this lane's B
// operand of k-step s (channel 2s + h of point 32j + (lane&31)); w1 is [MT][KS][64] floats.
template <int KS, int MT, int T>
__device__ __forceinline__ void first_layer(const float* __restrict__ w1, const float* __restrict__ b,
                                            const float (&in)[T][KS], f32x16 (&Y)[T][MT], int lane) {
    const int h = lane >> 5;
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        f32x16 acc[T];
        const f32x16 bias = tile_from_channels(b + 32 * mt, h);
#pragma unroll
        for (int j = 0; j < T; ++j) acc[j] = bias;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const float a = w1[(mt * KS + s) * 64 + lane];
#pragma unroll
            for (int j = 0; j < T; ++j) acc[j] = mfma32(a, in[j][s], acc[j]);
        }
#pragma unroll
        for (int j = 0; j < T; ++j) Y[j][mt] = relu16(acc[j]);
    }
}

// B operands of the first layer for T point tiles starting at point n0 of item b. Points past
// the end replicate the last point: every consumer is per-point work followed by a max over
// points (or a guarded store), so a duplicate never changes a result.
template <int KS, int T>
__device__ __forceinline__ void load_points(const BCN& x, int64_t b, int n0, int n_pts, int c_in,
                                            float (&in)[T][KS], int lane) {
    const int h = lane >> 5;
    if (x.dtype == 0) {                                    // fp32 storage (one uniform branch per call)
#pragma unroll
        for (int j = 0; j < T; ++j) {
            int n = n0 + 32 * j + (lane & 31);
            n = n < n_pts ? n : n_pts - 1;
            const float* p = x.data + b * x.sb + (int64_t)n * x.sn;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const int c = 2 * s + h;
                in[j][s] = c < c_in ? p[c * x.sc] : 0.0f;
            }
        }
        return;
    }
    const bool bf = x.dtype == 1;                          // bf16 / fp16 storage, widened exactly
#pragma unroll
    for (int j = 0; j < T; ++j) {
        int n = n0 + 32 * j + (lane & 31);
        n = n < n_pts ? n : n_pts - 1;
        const uint16_t* p = reinterpret_cast<const uint16_t*>(x.data) + b * x.sb + (int64_t)n * x.sn;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const int c = 2 * s + h;
            const uint16_t v = c < c_in ? p[c * x.sc] : (uint16_t)0;
            in[j][s] = bf ? widen_bf16(v) : widen_f16(v);
        }
    }
}

// Channel-wise max over the points of a tile (and the T tiles of the wave), post-ReLU, combined across waves /
// workgroups by an integer atomic max on the bit pattern (values are >= +0, so the order of the bit patterns is the
// order of the floats; dst is zero-filled before the launch). The folded bias is added AFTER the max (x -> fl(x + b)
// is monotone, so max_p fl(acc_p + b) == fl(max_p acc_p + b)): accumulators start at zero and no load sits in front
// of a chunk's MFMAs. The work is cut into 10 micro-steps; step k is issued right after fragment group k of the NEXT
// tile (mma_block_ring's side job), i.e. in the shadow of that group's last 64-cycle MFMA.
//
// The tile is TRANSPOSED (conv_max_layer computes the max-pooled layer with the MFMA operands swapped):
// the 32 points of a tile are the 16 registers of the two lane halves, the 32 channels are the lanes, so the max over
// the wave's points is a v_max3 chain over registers (8 T instructions instead of ~110 plus 16 swizzles) and one
// exchange between the halves. On this chip every VALU instruction between two f32 MFMAs costs matrix-pipe time
// (see above): without any epilogue the encode kernel ran 8.5 % faster, the heads 5 %.
template <int T>
struct MaxEpilogueT {
    static constexpr int STEPS = 10;
    float m, recv, bias_v;

    __device__ __forceinline__ void step(int k, const f32x16 (&acc)[T], const float* __restrict__ bias,
                                         float* __restrict__ dst, int lane) {
        if (k < 8) {                                   // registers 2k, 2k+1 of every tile
#pragma unroll
            for (int j = 0; j < T; ++j) {
                if (k == 0 && j == 0)
                    m = __builtin_fmaxf(acc[0][0], acc[0][1]);
                else
                    m = __builtin_fmaxf(__builtin_fmaxf(m, acc[j][2 * k]), acc[j][2 * k + 1]);
            }
            if (k == 0) bias_v = bias[lane & 31];      // LDS copy (an lgkmcnt wait, never a vmcnt drain of the ring)
        } else if (k == 8) {
            recv = __shfl_xor(m, 32);                  // the other half's 16 points of the same channel (v_permlane32_swap
        } else if (k == 9) {                           // instead of the LDS crossbar measured 0.5 % slower: copy + swap)
            int bits = __float_as_int(__builtin_fmaxf(m, recv) + bias_v);
            bits = bits > 0 ? bits : 0;                // ReLU on the bit pattern (-0.0 and negatives -> +0)
            if (lane < 32) atomicMax(reinterpret_cast<int*>(dst) + lane, bits);
        }
    }
    __device__ __forceinline__ void all(const f32x16 (&acc)[T], const float* bias, float* dst, int lane) {
#pragma unroll
        for (int k = 0; k < STEPS; ++k) step(k, acc, bias, dst, lane);
    }
};

// The last layer of a shared MLP with the max over points fused: n_tiles (even) output tiles of 32
// channels, swept with two accumulator sets; the epilogue of tile mt-1 rides under the MFMAs of tile mt.
// The fragments [n_tiles][KT][4][64] are next on the kernel's weight stream; bias/dst: the n_tiles*32 channels.
// Computed transposed (SWAP): see MaxEpilogueT.
template <int KT, int T, int D, bool CYC = false>
__device__ __forceinline__ void conv_max_layer(WRing<D, CYC>& ring, const float* __restrict__ bias,
                                               const f32x16 (&X)[T][KT], float* __restrict__ dst, int n_tiles,
                                               int lane) {
    f32x16 accA[T], accB[T];
    MaxEpilogueT<T> ep;
#ifdef DAL3_ABLATE_EPILOGUE
    for (int mt = 0; mt < n_tiles; ++mt) {
#pragma unroll
        for (int j = 0; j < T; ++j) accA[j] = f32x16{};
        mma_block_ring<KT, T, D, NoSide, true, CYC>(ring, X, accA);
#pragma unroll
        for (int j = 0; j < T; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) asm volatile("" ::"v"(accA[j][r]));
        }
    }
    return;
#endif
#pragma unroll
    for (int j = 0; j < T; ++j) accA[j] = f32x16{};
    mma_block_ring<KT, T, D, NoSide, true, CYC>(ring, X, accA);                // tile 0
    // The loop body is two whole tiles and nothing else. With the last tile's un-hidden epilogue inside it (an
    // `if (mt + 1 < n_tiles) ... else ep.all(...)`) the two paths left the ring's eight fragment registers and the two
    // accumulator sets in different places, and hipcc reconciled them on the back edge: s_waitcnt vmcnt(7) ... vmcnt(0)
    // around sixteen register copies — the ring drained, an L2 round trip with the matrix pipe idle, once per pair of tiles.
    auto tile_b = [&](int mt) {                                            // tile mt, epilogue of tile mt-1
#pragma unroll
        for (int j = 0; j < T; ++j) accB[j] = f32x16{};
        auto epA = [&](int i) {
            if (i < MaxEpilogueT<T>::STEPS) ep.step(i, accA, bias + 32 * (mt - 1), dst + 32 * (mt - 1), lane);
        };
        mma_block_ring<KT, T, D, decltype(epA), true, CYC>(ring, X, accB, epA);
    };
    int mt = 1;
    for (; mt + 1 < n_tiles; mt += 2) {
        tile_b(mt);
#pragma unroll
        for (int j = 0; j < T; ++j) accA[j] = f32x16{};
        auto epB = [&](int i) {                                            // tile mt+1, epilogue of tile mt
            if (i < MaxEpilogueT<T>::STEPS) ep.step(i, accB, bias + 32 * mt, dst + 32 * mt, lane);
        };
        mma_block_ring<KT, T, D, decltype(epB), true, CYC>(ring, X, accA, epB);
    }
    tile_b(mt);                                                            // n_tiles is even: mt == n_tiles - 1
    ep.all(accB, bias + 32 * mt, dst + 32 * mt, lane);                    // last tile: nothing left to hide under
}
