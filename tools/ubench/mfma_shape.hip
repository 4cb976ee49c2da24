// Microbenchmark 2: the decode kernel's dconv2 loop shape in isolation (T=1): 32 fragments per chunk,
// 4 MFMAs each into one of 8 resident accumulator tiles, B operands from a 16-register tile t, A
// operands from an 8-deep prefetch ring over a 512-KiB L2-resident stream. Variants toggle features.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define FENCE() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ f32x16 mf(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

// NTILE: accumulator tiles cycled through (8 = real kernel, 1 = single chain); RELU: refresh t each chunk
template <int NTILE, bool RELU, bool LOADS>
__global__ __launch_bounds__(256) void k(const f32x4* __restrict__ w, float* __restrict__ out, long long* __restrict__ cyc, int chunks) {
    const int lane = threadIdx.x & 63;
    f32x16 a2[NTILE];
    for (int j = 0; j < NTILE; ++j) a2[j] = f32x16{};
    f32x16 t;
    for (int r = 0; r < 16; ++r) t[r] = 1.0f + 0.001f * (lane + r);
    f32x4 ring[8];
    const f32x4* next = w + lane;
    for (int i = 0; i < 8; ++i) { ring[i] = LOADS ? *next : f32x4{0.5f, 0.25f, 0.125f, 1.0f}; next += 64; }
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int c = 0; c < chunks; ++c) {
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            const f32x4 a = ring[i % 8];
            if (LOADS) { ring[i % 8] = *next; next += 64; }
#pragma unroll
            for (int e = 0; e < 4; ++e) a2[(i / 4) % NTILE] = mf(a[e], t[4 * (i % 4) + e], a2[(i / 4) % NTILE]);
            FENCE();
            if (RELU && i < 8) {   // two registers of the next t per group, from another accumulator
                const int bb = __float_as_int(a2[NTILE - 1][2 * i]);
                t[2 * i] = __int_as_float(bb > 0 ? bb : 0) * 1e-6f + 1.0f;
                const int b2 = __float_as_int(a2[NTILE - 1][2 * i + 1]);
                t[2 * i + 1] = __int_as_float(b2 > 0 ? b2 : 0) * 1e-6f + 1.0f;
            }
            FENCE();
        }
        if ((c & 15) == 15) next = w + lane + 8 * 64;   // restart the 512 KiB stream
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int j = 0; j < NTILE; ++j) for (int r = 0; r < 16; ++r) s += a2[j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s + t[3];
    if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int NTILE, bool RELU, bool LOADS>
void run(const char* name, const f32x4* w, float* out, long long* cyc) {
    const int chunks = 64, blocks = 256;
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((k<NTILE, RELU, LOADS>), dim3(blocks), dim3(256), 0, 0, w, out, cyc, chunks);
        hipDeviceSynchronize();
    }
    std::vector<long long> h(blocks * 4);
    hipMemcpy(h.data(), cyc, h.size() * 8, hipMemcpyDeviceToHost);
    double avg = 0; for (auto v : h) avg += v; avg /= h.size();
    printf("%-44s ticks/MFMA/wave %7.2f\n", name, avg / (chunks * 128.0));
}

int main() {
    f32x4* w; float* out; long long* cyc;
    hipMalloc(&w, 2 << 20); hipMemset(w, 0, 2 << 20);
    hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 4 * 8);
    run<1, false, false>("1 tile, no loads", w, out, cyc);
    run<8, false, false>("8 tiles, no loads", w, out, cyc);
    run<1, false, true>("1 tile, loads", w, out, cyc);
    run<8, false, true>("8 tiles, loads", w, out, cyc);
    run<8, true, true>("8 tiles, loads, relu side job", w, out, cyc);
    run<8, true, false>("8 tiles, no loads, relu side job", w, out, cyc);
    return 0;
}
