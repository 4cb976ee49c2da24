// Microbenchmark: what limits a v_mfma_f32_32x32x2_f32 stream on gfx950?
//   mode 0/1/2: 1, 2, 4 independent accumulators, operands in registers
//   mode 3/4/5: same, plus one 1-KiB global_load_dwordx4 per 4 MFMAs supplying the A operand from an
//               L2-resident 512-KiB stream through an 8-deep ring pinned with sched_barrier
//   mode 6:     1 accumulator + loads + a ReLU'd hand-off every 32 MFMAs (the decode kernel's chunk seam)
// Prints shader cycles per MFMA per wave (s_memtime) at 1 and 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define FENCE() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ f32x16 mf(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

template <int NACC, bool LOADS, bool SEAM>
__global__ __launch_bounds__(256) void k(const f32x4* __restrict__ w, float* __restrict__ out, long long* __restrict__ cyc, int iters, int wrap_frags) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j) acc[j] = f32x16{};
    float b[16];
    for (int r = 0; r < 16; ++r) b[r] = 1.0f + 0.001f * (lane + r);
    f32x4 ring[8];
    const f32x4* next = w + lane;
    if (LOADS) for (int i = 0; i < 8; ++i) { ring[i] = *next; next += 64; }
    else for (int i = 0; i < 8; ++i) ring[i] = f32x4{0.5f, 0.25f, 0.125f, 1.0f};
    long long t0 = __builtin_amdgcn_s_memtime();
    int frag = 8;
    for (int it = 0; it < iters; ++it) {
        const f32x4* base = w + lane;
#pragma unroll
        for (int i = 0; i < 16; ++i) {                    // 16 fragments = 64 MFMAs per accumulator set
            const f32x4 a = ring[i % 8];
            if (LOADS) { ring[i % 8] = *next; next += 64; if (++frag >= wrap_frags) { frag = 0; next = base; } }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
#pragma unroll
                for (int j = 0; j < NACC; ++j) acc[j] = mf(a[e], b[(4 * (i % 4) + e)], acc[j]);
            }
            FENCE();
            if (SEAM && (i % 8) == 7) {                   // consume the accumulator on the VALU, feed it back as B
#pragma unroll
                for (int r = 0; r < 16; ++r) b[r] = fmaxf(acc[0][r], 0.0f) * 1e-3f + 1.0f;
                acc[0] = f32x16{};
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s + b[3];
    if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int NACC, bool LOADS, bool SEAM>
void run(const char* name, const f32x4* w, float* out, long long* cyc, int blocks, int wrap_frags = 16) {
    const int iters = 200;
    hipLaunchKernelGGL((k<NACC, LOADS, SEAM>), dim3(blocks), dim3(256), 0, 0, w, out, cyc, iters, wrap_frags);
    hipDeviceSynchronize();
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<NACC, LOADS, SEAM>), dim3(blocks), dim3(256), 0, 0, w, out, cyc, iters, wrap_frags);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    std::vector<long long> h(blocks * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= h.size();
    const double n_mfma = (double)iters * 64 * NACC;
    const double waves_per_simd = blocks / 256.0;
    printf("%-34s blocks %4d  memtime ticks/MFMA/wave %7.2f   wall: %.3f ms -> %.1f TF, %.2f ns per MFMA per SIMD\n", name, blocks,
           avg / n_mfma, ms, n_mfma * blocks * 4 * 4096.0 / (ms * 1e-3) / 1e12, ms * 1e6 / (n_mfma * waves_per_simd));
}

int main() {
    f32x4* w; float* out; long long* cyc;
    hipMalloc(&w, 8 << 20); hipMemset(w, 0, 8 << 20);
    hipMalloc(&out, 2048 * 256 * 4); hipMalloc(&cyc, 2048 * 4 * 8);
    for (int frags : {16, 64, 512, 1024, 4096}) {
        char nm[64];
        snprintf(nm, sizeof nm, "1 acc +load/4, stream %d KiB", frags);
        run<1, true, false>(nm, w, out, cyc, 256, frags);
        snprintf(nm, sizeof nm, "2 acc +load/8, stream %d KiB", frags);
        run<2, true, false>(nm, w, out, cyc, 256, frags);
    }
    for (int blocks : {256}) {
        run<1, false, false>("1 acc, regs", w, out, cyc, blocks);
        run<2, false, false>("2 acc, regs", w, out, cyc, blocks);
        run<4, false, false>("4 acc, regs", w, out, cyc, blocks);
        run<1, true, false>("1 acc, +load/4 MFMA", w, out, cyc, blocks);
        run<2, true, false>("2 acc, +load/8 MFMA", w, out, cyc, blocks);
        run<4, true, false>("4 acc, +load/16 MFMA", w, out, cyc, blocks);
        run<1, true, true>("1 acc, +load, VALU seam/32 MFMA", w, out, cyc, blocks);
    }
    return 0;
}
