// tools/ubench/store_pattern.hip — what the shape of a wave's stores costs when a kernel is bound by its output:
// 262,144 rows x 512 fp32 channels (537 MB), every wave writes a 64-row x 64-channel tile (row stride 2 KB), as
//   A: the MFMA C/D layout's natural stores (lane = (h, m): 16 B at row m, channel 8q + 4h: 32 rows x 32 B per instruction)
//   B: after a transposition (lane l: row l / 16 (+4 per step), channel 4 (l % 16): 4 rows x 256 B per instruction)
//   C: A with non-temporal stores        D: B with non-temporal stores
// hipcc -O3 --offload-arch=gfx950 store_pattern.hip -o store_pattern && ./store_pattern
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float* __restrict__ z, int64_t ldz, int n_mblk) {
    const int lane = threadIdx.x & 63;
    const uint32_t unit = blockIdx.x * 4u + (threadIdx.x >> 6);
    const int mblk = unit % n_mblk;
    const int64_t pt0 = (int64_t)(unit / n_mblk) * 64;
    const f32x4 v = {1.0f * lane, 2.0f, 3.0f, (float)unit};
    if (MODE == 0 || MODE == 2) {
        const int h = lane >> 5, m = lane & 31;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4* dst = reinterpret_cast<f32x4*>(z + (pt0 + 32 * j + m) * ldz + 64 * mblk + 32 * t + 8 * q + 4 * h);
                    if (MODE == 2) __builtin_nontemporal_store(v, dst); else *dst = v;
                }
    } else {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            f32x4* dst = reinterpret_cast<f32x4*>(z + (pt0 + 4 * i + (lane >> 4)) * ldz + 64 * mblk + 4 * (lane & 15));
            if (MODE == 3) __builtin_nontemporal_store(v, dst); else *dst = v;
        }
    }
}
int main() {
    const int64_t M = 262144; const int C = 512;
    float* z; hipMalloc(&z, M * C * 4);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    const int n_mblk = C / 64; const unsigned grid = (unsigned)(M / 64 * n_mblk / 4);
    for (int mode = 0; mode < 4; ++mode) {
        float best = 1e9f;
        for (int r = 0; r < 6; ++r) {
            hipEventRecord(a);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, z, C, n_mblk);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, z, C, n_mblk);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, z, C, n_mblk);
            if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, 0, z, C, n_mblk);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (r && ms < best) best = ms;
        }
        printf("mode %c: %.1f us  %.2f TB/s\n", 'A' + mode, best * 1e3, M * C * 4 / (best * 1e-3) / 1e12);
    }
    return 0;
}
