// dal3_train.hip — training-mode building blocks for the shared-MLP stacks: what
// `loss.backward()` drives through Conv1d(k=1) + BatchNorm1d(batch statistics) + ReLU + max over points in
// tools/static_model.py:271-295,326-339 (and the dynamic twins), as fp32 MFMA kernels over POINT-MAJOR row-major
// activations  a[M x C]  (M = B*N points; what `(B,C,N).transpose(2,1)` is in memory).
//
// Why row-major point-major: the same buffer feeds all three GEMM shapes of a layer without a transpose —
//   forward   z[M x Co]  = act(a)[M x Ci] . W^T         channels on MFMA rows, points on lanes: a lane reads four
//   dgrad     da[M x Ci] = dz[M x Co] . W               consecutive channels of ITS point (16 B), the accumulator
//                                                       tile stores the same way;
//   wgrad     dW[Co x Ci] = dz^T . act(a)               the contraction runs over POINTS: the A operand wants lane =
//                                                       out-channel, k = point, i.e. for each point 32 consecutive
//                                                       floats of a row: a 128-B coalesced row segment per half-wave.
// Train-mode BN needs the statistics of a layer's whole output before the next layer can run, so — unlike the
// eval kernels, which chain a tile through a whole network in registers — every layer's pre-BN output z is
// materialised once; normalisation + ReLU are applied on load by the consumer ("act": y = max(z*scale+shift, 0)).
// All channel counts are multiples of 32 here (the host pads 3 -> 32 input channels and 2 -> 32 logits; that is
// < 1 % of the work) and M is a multiple of 32.
#include "dal3_device.h"
#include "dal3_kernels.h"

#define TR_T 2                          // point tiles (32 points each) per wave in the linear kernel
#define TR_MTB 4                        // output tiles (32 channels each) per wave: 4*2*16 = 128 accumulator registers

// ---------------------------------------------------------------------------------------------- linear
// z[p][co] (+)= sum_ci act(a[p][ci]) * Wop[co][ci] + bias
//   Wop = W (row-major (c_out, c_in), ldw = row stride)           when !transpose_w   (forward)
//   Wop[co][ci] = W[ci][co] with W row-major (c_in, c_out)         when  transpose_w   (dgrad: W is the layer's
//                                                                   (c_out_fwd, c_in_fwd) matrix, co here = ci_fwd)
//   act: scale == NULL -> identity; else y = z*scale[ci] + shift[ci], then max(y,0) if relu_in
//   bias: NULL, per-channel (seg == 0) or per-segment (bias[(p / seg) * c_out + co], the decoder's per-crop term)
// grid: (M/32/TR_T, c_out/32/TR_MTB rounded up); one wave per block of 64 threads x 4 waves? -> 1 wave = 1 unit.
__global__ __launch_bounds__(256) void tr_linear_kernel(const float* __restrict__ a, int64_t M, int c_in, int64_t lda,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        int relu_in, const float* __restrict__ W, int64_t ldw,
                                                        int transpose_w, const float* __restrict__ bias, int64_t seg,
                                                        int c_out, float* __restrict__ z, int64_t ldz, int accumulate,
                                                        int n_mblk) {
    const int lane = threadIdx.x & 63, h = lane >> 5, m = lane & 31;
    const int64_t unit = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int mblk = (int)(unit % n_mblk);
    const int64_t pt0 = (unit / n_mblk) * (32 * TR_T);
    if (pt0 >= M) return;
    const int mt0 = mblk * TR_MTB;
    const int n_mt = min(TR_MTB, c_out / 32 - mt0);
    const int KT = c_in / 32;

    f32x16 acc[TR_T][TR_MTB];
#pragma unroll
    for (int j = 0; j < TR_T; ++j) {
        const int64_t p = min(pt0 + 32 * j, M - 32) + m;
#pragma unroll
        for (int t = 0; t < TR_MTB; ++t) {
            if (t < n_mt && bias) {
                const float* bp = bias + (seg > 0 ? (p / seg) * c_out : 0) + 32 * (mt0 + t);
                acc[j][t] = tile_from_channels(bp, h);
            } else {
                acc[j][t] = f32x16{};
            }
        }
    }
    for (int kt = 0; kt < KT; ++kt) {
        f32x16 X[TR_T];
#pragma unroll
        for (int j = 0; j < TR_T; ++j) {
            const int64_t p = min(pt0 + 32 * j, M - 32) + m;
            const float* ap = a + p * lda + 32 * kt + 4 * h;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4 v = *reinterpret_cast<const f32x4*>(ap + 8 * q);
                if (scale) {
                    const f32x4 sc = *reinterpret_cast<const f32x4*>(scale + 32 * kt + 8 * q + 4 * h);
                    const f32x4 sh = *reinterpret_cast<const f32x4*>(shift + 32 * kt + 8 * q + 4 * h);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        v[e] = v[e] * sc[e] + sh[e];
                        if (relu_in) v[e] = fmaxf(v[e], 0.0f);
                    }
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) X[j][4 * q + e] = v[e];
            }
        }
#pragma unroll
        for (int t = 0; t < TR_MTB; ++t) {
            if (t >= n_mt) break;
            const int row = 32 * (mt0 + t) + m;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int col = 32 * kt + 8 * q + 4 * h;
                f32x4 w;
                if (!transpose_w) {
                    w = *reinterpret_cast<const f32x4*>(W + (int64_t)row * ldw + col);
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) w[e] = W[(int64_t)(col + e) * ldw + row];
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
#pragma unroll
                    for (int j = 0; j < TR_T; ++j) acc[j][t] = mfma32(w[e], X[j][4 * q + e], acc[j][t]);
                }
            }
        }
    }
#pragma unroll
    for (int j = 0; j < TR_T; ++j) {
        if (pt0 + 32 * j >= M) break;
        float* zp = z + (pt0 + 32 * j + m) * ldz + 4 * h;
#pragma unroll
        for (int t = 0; t < TR_MTB; ++t) {
            if (t >= n_mt) break;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                f32x4* dst = reinterpret_cast<f32x4*>(zp + 32 * (mt0 + t) + 8 * q);
                f32x4 o = {acc[j][t][4 * q], acc[j][t][4 * q + 1], acc[j][t][4 * q + 2], acc[j][t][4 * q + 3]};
                if (accumulate) {
                    const f32x4 old = *dst;
                    o[0] += old[0];
                    o[1] += old[1];
                    o[2] += old[2];
                    o[3] += old[3];
                }
                *dst = o;
            }
        }
    }
}

hipError_t launch_tr_linear(const float* a, int64_t M, int c_in, int64_t lda, const float* scale, const float* shift,
                            int relu_in, const float* W, int64_t ldw, int transpose_w, const float* bias, int64_t seg,
                            int c_out, float* z, int64_t ldz, int accumulate, hipStream_t s) {
    const int n_mblk = (c_out / 32 + TR_MTB - 1) / TR_MTB;
    const int64_t units = ((M + 32 * TR_T - 1) / (32 * TR_T)) * n_mblk;
    hipLaunchKernelGGL(tr_linear_kernel, dim3((unsigned)((units + 3) / 4)), dim3(256), 0, s, a, M, c_in, lda, scale, shift,
                       relu_in, W, ldw, transpose_w, bias, seg, c_out, z, ldz, accumulate, n_mblk);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------- column sums
// Two-stage, fixed-order reductions over the points (deterministic): stage 1 writes one double partial per
// (row block, channel), stage 2 adds the partials of a channel in index order.
// mode 0: s0 = sum z, s1 = sum z^2                                   (batch statistics)
// mode 1: dy = da * [z*scale+shift > 0];  s0 = sum dy (= dbeta), s1 = sum dy * xhat (= dgamma), xhat = (z - mu)*rstd
// da may be NULL with (dg, arg): da[p][c] = (p == arg[seg_of_p][c]) ? dg[seg_of_p][c] : 0   (the max over points)
#define TR_RED_ROWS 256
struct DaSrc {
    const float* da;       // dense (M x C) or NULL
    int64_t ldda;
    const float* dg;       // (M/seg x C) when da == NULL
    const int32_t* arg;    // (M/seg x C) point index within the segment
    int64_t seg;
    __device__ __forceinline__ float at(int64_t p, int c, int C) const {
        if (da) return da[p * ldda + c];
        const int64_t s = p / seg;
        return arg[s * C + c] == (int32_t)(p - s * seg) ? dg[s * C + c] : 0.0f;
    }
};

__global__ __launch_bounds__(256) void tr_colred_kernel(const float* __restrict__ z, int64_t M, int C, int64_t ldz, int mode,
                                                        DaSrc src, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, const float* __restrict__ mu,
                                                        const float* __restrict__ rstd, double* __restrict__ part) {
    // block: 64 channels x 4 row-lanes; rows [blockIdx.y*TR_RED_ROWS, +TR_RED_ROWS)
    __shared__ double sm[2][4][64];
    const int cl = threadIdx.x & 63, rl = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    const int64_t r0 = (int64_t)blockIdx.y * TR_RED_ROWS;
    double s0 = 0.0, s1 = 0.0;
    if (c < C) {
        float sc = 0.f, sh = 0.f, mean = 0.f, rs = 0.f;
        if (mode == 1) {
            sc = scale[c];
            sh = shift[c];
            mean = mu[c];
            rs = rstd[c];
        }
        for (int64_t p = r0 + rl; p < min(M, r0 + TR_RED_ROWS); p += 4) {
            const float v = z[p * ldz + c];
            if (mode == 0) {
                s0 += v;
                s1 += (double)v * v;
            } else {
                const float y = v * sc + sh;
                const float dy = y > 0.0f ? src.at(p, c, C) : 0.0f;
                s0 += dy;
                s1 += (double)dy * ((v - mean) * rs);
            }
        }
    }
    sm[0][rl][cl] = s0;
    sm[1][rl][cl] = s1;
    __syncthreads();
    if (rl == 0 && c < C) {
        const int64_t o = ((int64_t)blockIdx.y * C + c) * 2;
        part[o] = ((sm[0][0][cl] + sm[0][1][cl]) + sm[0][2][cl]) + sm[0][3][cl];
        part[o + 1] = ((sm[1][0][cl] + sm[1][1][cl]) + sm[1][2][cl]) + sm[1][3][cl];
    }
}

__global__ void tr_colred_final_kernel(const double* __restrict__ part, int n_blocks, int C, double* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    double s0 = 0.0, s1 = 0.0;
    for (int b = 0; b < n_blocks; ++b) {
        s0 += part[((int64_t)b * C + c) * 2];
        s1 += part[((int64_t)b * C + c) * 2 + 1];
    }
    out[c] = s0;
    out[C + c] = s1;
}

size_t tr_colred_workspace_bytes(int64_t M, int C) {
    return (size_t)((M + TR_RED_ROWS - 1) / TR_RED_ROWS) * C * 2 * sizeof(double);
}

hipError_t launch_tr_colred(const float* z, int64_t M, int C, int64_t ldz, int mode, const float* da, int64_t ldda,
                            const float* dg, const int32_t* arg, int64_t seg, const float* scale, const float* shift,
                            const float* mu, const float* rstd, double* part, double* out, hipStream_t s) {
    const int nb = (int)((M + TR_RED_ROWS - 1) / TR_RED_ROWS);
    DaSrc src{da, ldda, dg, arg, seg};
    hipLaunchKernelGGL(tr_colred_kernel, dim3((C + 63) / 64, nb), dim3(256), 0, s, z, M, C, ldz, mode, src, scale, shift, mu,
                       rstd, part);
    hipLaunchKernelGGL(tr_colred_final_kernel, dim3((C + 127) / 128), dim3(128), 0, s, part, nb, C, out);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------- BN backward, applied
// dz = k1[c] * (dy - k2[c] - xhat * k3[c]),  k1 = gamma*rstd, k2 = mean(dy), k3 = mean(dy*xhat)
__global__ __launch_bounds__(256) void tr_bnbwd_apply_kernel(const float* __restrict__ z, int64_t M, int C, int64_t ldz,
                                                             DaSrc src, const float* __restrict__ scale,
                                                             const float* __restrict__ shift, const float* __restrict__ mu,
                                                             const float* __restrict__ rstd, const float* __restrict__ k1,
                                                             const float* __restrict__ k2, const float* __restrict__ k3,
                                                             float* __restrict__ dz, int64_t lddz) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * C) return;
    const int64_t p = i / C;
    const int c = (int)(i - p * C);
    const float v = z[p * ldz + c];
    const float y = v * scale[c] + shift[c];
    const float dy = y > 0.0f ? src.at(p, c, C) : 0.0f;
    const float xhat = (v - mu[c]) * rstd[c];
    dz[p * lddz + c] = k1[c] * (dy - k2[c] - xhat * k3[c]);
}

hipError_t launch_tr_bnbwd_apply(const float* z, int64_t M, int C, int64_t ldz, const float* da, int64_t ldda,
                                 const float* dg, const int32_t* arg, int64_t seg, const float* scale, const float* shift,
                                 const float* mu, const float* rstd, const float* k1, const float* k2, const float* k3,
                                 float* dz, int64_t lddz, hipStream_t s) {
    DaSrc src{da, ldda, dg, arg, seg};
    const int64_t total = M * C;
    hipLaunchKernelGGL(tr_bnbwd_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, z, M, C, ldz, src,
                       scale, shift, mu, rstd, k1, k2, k3, dz, lddz);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------- wgrad
// dW[co][ci] = sum_p dz[p][co] * act(a[p][ci]): D(32 co x 32 ci) += A(co x 2 points) . B(2 points x ci) per MFMA.
// A lane (co = lane&31, h) reads dz[p0 + 2s + h][co], B lane (ci, h) reads a[p0 + 2s + h][ci]: each a 128-B row
// segment per half-wave. A wave owns a WG_MT x WG_KT block of tiles and a slice of the points; the slices' partial
// sums go to part[slice][co][ci] and are added in slice order by tr_wgrad_final_kernel (deterministic).
#define WG_MT 4
#define WG_KT 4
#define WG_SLICE 2048                    // points per wave

__global__ __launch_bounds__(256) void tr_wgrad_kernel(const float* __restrict__ dz, int64_t lddz, const float* __restrict__ a,
                                                       int64_t lda, const float* __restrict__ scale,
                                                       const float* __restrict__ shift, int relu_in, int64_t M, int c_out,
                                                       int c_in, float* __restrict__ part, int n_mb, int n_kb) {
    const int lane = threadIdx.x & 63, h = lane >> 5, m = lane & 31;
    const int64_t unit = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int blk = (int)(unit % (n_mb * n_kb));
    const int64_t slice = unit / (n_mb * n_kb);
    const int64_t p_begin = slice * WG_SLICE;
    if (p_begin >= M) return;
    const int64_t p_end = min(M, p_begin + WG_SLICE);
    const int mt0 = (blk / n_kb) * WG_MT, kt0 = (blk % n_kb) * WG_KT;
    const int n_mt = min(WG_MT, c_out / 32 - mt0), n_kt = min(WG_KT, c_in / 32 - kt0);
    float sc[WG_KT], sh[WG_KT];
#pragma unroll
    for (int k = 0; k < WG_KT; ++k) {
        sc[k] = (scale && k < n_kt) ? scale[32 * (kt0 + k) + m] : 1.0f;
        sh[k] = (scale && k < n_kt) ? shift[32 * (kt0 + k) + m] : 0.0f;
    }
    f32x16 acc[WG_MT][WG_KT];
#pragma unroll
    for (int t = 0; t < WG_MT; ++t)
#pragma unroll
        for (int k = 0; k < WG_KT; ++k) acc[t][k] = f32x16{};
    for (int64_t p = p_begin + h; p < p_end; p += 2) {
        float av[WG_MT], bv[WG_KT];
#pragma unroll
        for (int t = 0; t < WG_MT; ++t) av[t] = t < n_mt ? dz[p * lddz + 32 * (mt0 + t) + m] : 0.0f;
#pragma unroll
        for (int k = 0; k < WG_KT; ++k) {
            float v = k < n_kt ? a[p * lda + 32 * (kt0 + k) + m] : 0.0f;
            if (scale) {
                v = v * sc[k] + sh[k];
                if (relu_in) v = fmaxf(v, 0.0f);
            }
            bv[k] = v;
        }
#pragma unroll
        for (int t = 0; t < WG_MT; ++t)
#pragma unroll
            for (int k = 0; k < WG_KT; ++k) acc[t][k] = mfma32(av[t], bv[k], acc[t][k]);
    }
    // D tile: row (co) = tile_chan(r, h), col (ci) = lane & 31
    float* out = part + slice * (int64_t)c_out * c_in;
#pragma unroll
    for (int t = 0; t < WG_MT; ++t) {
        if (t >= n_mt) break;
#pragma unroll
        for (int k = 0; k < WG_KT; ++k) {
            if (k >= n_kt) break;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                out[(int64_t)(32 * (mt0 + t) + tile_chan(r, h)) * c_in + 32 * (kt0 + k) + m] = acc[t][k][r];
        }
    }
}

__global__ void tr_wgrad_final_kernel(const float* __restrict__ part, int n_slices, int64_t n, float* __restrict__ dW) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.0f;
    for (int k = 0; k < n_slices; ++k) s += part[(int64_t)k * n + i];
    dW[i] = s;
}

size_t tr_wgrad_workspace_bytes(int64_t M, int c_out, int c_in) {
    return (size_t)((M + WG_SLICE - 1) / WG_SLICE) * c_out * c_in * sizeof(float);
}

hipError_t launch_tr_wgrad(const float* dz, int64_t lddz, const float* a, int64_t lda, const float* scale,
                           const float* shift, int relu_in, int64_t M, int c_out, int c_in, float* part, float* dW,
                           hipStream_t s) {
    const int n_mb = (c_out / 32 + WG_MT - 1) / WG_MT, n_kb = (c_in / 32 + WG_KT - 1) / WG_KT;
    const int64_t n_slices = (M + WG_SLICE - 1) / WG_SLICE;
    const int64_t units = n_slices * n_mb * n_kb;
    hipLaunchKernelGGL(tr_wgrad_kernel, dim3((unsigned)((units + 3) / 4)), dim3(256), 0, s, dz, lddz, a, lda, scale, shift,
                       relu_in, M, c_out, c_in, part, n_mb, n_kb);
    const int64_t n = (int64_t)c_out * c_in;
    hipLaunchKernelGGL(tr_wgrad_final_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, part, (int)n_slices, n, dW);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------- max over points
// g[s][c] = max_p act(z[p][c]) over the `seg` points of segment s, with the index of the (first) maximum for the
// backward pass. One thread per (segment, channel) walking the segment: consecutive threads read consecutive
// channels of one row, so a warp-row of 64 channels is one 256-B segment per step.
__global__ __launch_bounds__(256) void tr_segmax_kernel(const float* __restrict__ z, int64_t ldz, int64_t seg, int C,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        float* __restrict__ g, int32_t* __restrict__ arg, int64_t n_seg) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_seg * C) return;
    const int64_t s = i / C;
    const int c = (int)(i - s * C);
    const float sc = scale[c], sh = shift[c];
    float best = -INFINITY;
    int32_t bi = 0;
    const float* zp = z + s * seg * ldz + c;
    for (int64_t p = 0; p < seg; ++p) {
        const float y = fmaxf(zp[p * ldz] * sc + sh, 0.0f);
        if (y > best) {
            best = y;
            bi = (int32_t)p;
        }
    }
    g[i] = best;
    arg[i] = bi;
}

hipError_t launch_tr_segmax(const float* z, int64_t ldz, int64_t seg, int C, const float* scale, const float* shift,
                            float* g, int32_t* arg, int64_t n_seg, hipStream_t s) {
    const int64_t total = n_seg * C;
    hipLaunchKernelGGL(tr_segmax_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, z, ldz, seg, C, scale, shift,
                       g, arg, n_seg);
    return hipGetLastError();
}

// per-segment column sums: out[s][c] = sum_{p in segment s} x[p][c]  (gradient of the decoder's per-crop term)
__global__ __launch_bounds__(256) void tr_segsum_kernel(const float* __restrict__ x, int64_t ldx, int64_t seg, int C,
                                                        float* __restrict__ out, int64_t n_seg) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_seg * C) return;
    const int64_t s = i / C;
    const int c = (int)(i - s * C);
    const float* xp = x + s * seg * ldx + c;
    double acc = 0.0;
    for (int64_t p = 0; p < seg; ++p) acc += xp[p * ldx];
    out[i] = (float)acc;
}

hipError_t launch_tr_segsum(const float* x, int64_t ldx, int64_t seg, int C, float* out, int64_t n_seg, hipStream_t s) {
    const int64_t total = n_seg * C;
    hipLaunchKernelGGL(tr_segsum_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, ldx, seg, C, out, n_seg);
    return hipGetLastError();
}
