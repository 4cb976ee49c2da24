// Microbenchmark 3: where does the ~20-cycle bubble per {global_load, s_waitcnt, 4 dependent MFMA} group come from?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define FENCE() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ f32x16 mf(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

// MODE 0: load at group start, consumed 8 groups later (the kernel's shape)
// MODE 1: load issued after the 2nd MFMA of the group
// MODE 2: load issued but never consumed (A operand constant): no s_waitcnt in the loop
// MODE 3: no load, but two VALU per group
// MODE 4: as 0 with two independent accumulators alternating per MFMA (T=2 shape, 8 MFMAs per load)
// MODE 5: as 0 but the accumulator alternates per GROUP between two tiles (dependency across the load is broken)
// MODE 11: as 0 but the fragments come from LDS (ds_read_b128, ring of D), no global loads in the loop
// MODE 12: as 11 with the read issued after the 2nd MFMA of the group
template <int MODE, int D = 8>
__global__ __launch_bounds__(256) void k(const f32x4* __restrict__ w, float* __restrict__ out, long long* __restrict__ cyc, int iters) {
    const int lane = threadIdx.x & 63;
    f32x16 acc0 = {}, acc1 = {};
    float t[16];
    for (int r = 0; r < 16; ++r) t[r] = 1.0f + 0.001f * (lane + r);
    f32x4 ring[D];
    const f32x4* next = w + lane;
    for (int i = 0; i < D; ++i) { ring[i] = *next; next += 64; }
    const f32x4 cst = {0.5f, 0.25f, 0.125f, 1.0f};
    __shared__ f32x4 lds[64 * 64];                                  // 64 KiB: 64 fragments
    if (MODE == 11 || MODE == 12) {
        for (int i = threadIdx.x; i < 64 * 64; i += 256) lds[i] = w[i];
        __syncthreads();
    }
    int lf = D;
    float junk = 0;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            if (MODE == 7 && i % 4 == 0) asm volatile("" : "+v"(ring[(i + 3) % D][0]));
            f32x4 a = (MODE == 2 || MODE == 3) ? cst : ring[i % D];
            if (MODE == 0 || MODE == 4 || MODE == 5 || MODE == 7 || MODE == 10) { ring[i % D] = *next; next += 64; }
            if (MODE == 11) { ring[i % D] = lds[(lf & 63) * 64 + lane]; ++lf; }
            if (MODE == 8) { asm volatile("s_nop 0"); }
            if (MODE == 9) { asm volatile("s_waitcnt vmcnt(0)"); }
            if (MODE == 2) { ring[i % D] = *next; next += 64; }
            if (MODE == 3) { junk += 1.0f; junk *= 1.5f; }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (MODE == 4) { acc0 = mf(a[e], t[4 * (i % 4) + e], acc0); acc1 = mf(a[e], t[(4 * (i % 4) + e + 5) & 15], acc1); }
                else if (MODE == 5) { if (i & 1) acc1 = mf(a[e], t[4 * (i % 4) + e], acc1); else acc0 = mf(a[e], t[4 * (i % 4) + e], acc0); }
                else acc0 = mf(a[e], t[4 * (i % 4) + e], acc0);
                if (MODE == 1 && e == 1) { FENCE(); ring[i % D] = *next; next += 64; FENCE(); }
                if (MODE == 12 && e == 1) { FENCE(); ring[i % D] = lds[(lf & 63) * 64 + lane]; ++lf; FENCE(); }
            }
            FENCE();
        }
        if (MODE == 10) next = w + lane; else if ((it & 15) == 15) next = w + lane + D * 64;
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = junk;
    for (int r = 0; r < 16; ++r) s += acc0[r] + acc1[r];
    if (MODE == 2) for (int i = 0; i < D; ++i) s += ring[i][0];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int MODE, int D = 8>
void run(const char* name, const f32x4* w, float* out, long long* cyc) {
    const int iters = 64, blocks = 256;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<MODE, D>), dim3(blocks), dim3(256), 0, 0, w, out, cyc, iters);
        hipDeviceSynchronize();
    }
    std::vector<long long> h(blocks * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= h.size();
    const double n = iters * 32.0 * (MODE == 4 ? 8 : 4);
    printf("%-64s ticks/MFMA/wave %7.2f\n", name, avg / n);
}

int main() {
    f32x4* w; float* out; long long* cyc;
    hipMalloc(&w, 2 << 20); hipMemset(w, 0, 2 << 20);
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 4 * 8);
    run<0>("0: load at group start, consumed 8 groups later", w, out, cyc);
    run<1>("1: load after the 2nd MFMA of the group", w, out, cyc);
    run<2>("2: load never consumed (no s_waitcnt)", w, out, cyc);
    run<3>("3: no load, two VALU per group", w, out, cyc);
    run<4>("4: two accumulators alternating (8 MFMA per load)", w, out, cyc);
    run<5>("5: accumulator alternates per group", w, out, cyc);
    run<10>("10: as 0 but the stream is 32 KiB (L1-resident)", w, out, cyc);
    run<0, 16>("0/D16: ring depth 16", w, out, cyc);
    run<0, 4>("0/D4: ring depth 4", w, out, cyc);
    run<4, 16>("4/D16: two accumulators (8 MFMA per load), depth 16", w, out, cyc);
    run<6>("6: no loads, A operand cycles through 32 preloaded registers", w, out, cyc);
    run<7>("7: as 0 with one s_waitcnt per 4 groups", w, out, cyc);
    run<8>("8: as 6 plus one s_nop 0 per group", w, out, cyc);
    run<9>("9: as 6 plus one (trivially satisfied) s_waitcnt vmcnt(0) per group", w, out, cyc);
    run<11>("11: fragments from LDS (ds_read_b128 ring of 8)", w, out, cyc);
    run<12>("12: as 11, read issued after the 2nd MFMA", w, out, cyc);
    run<11, 4>("11/D4: LDS ring of 4", w, out, cyc);
    return 0;
}
