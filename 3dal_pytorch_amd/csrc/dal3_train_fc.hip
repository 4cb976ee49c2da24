// dal3_train_fc.hip — the per-item FC tails of the heads in train mode (static_model.py:336-338, dynamic_model.py:247-248,
// :284-285, :306-311): Linear -> BatchNorm1d (batch statistics over the B items) -> ReLU with rows = ITEMS, 2–256 of
// them. On the per-point kernels (dal3_train.hip) such a layer is a 64-row GEMM walked by eight waves one k-tile after the
// other, followed by a statistics pass, its second stage, and in the backward a sums pass, its second stage, the apply pass,
// a wgrad, its second stage and a dgrad: ≈ 30 dependent dispatches and 230 us a step for ≈ 100 MFLOP (round 4 timeline).
// Here a layer is ONE launch forward and TWO backward, on the vector ALUs:
//   tr_fc_fwd_kernel    z = act(a) W^T + b for FC_CB output channels x all rows per workgroup; the channel's batch statistics
//                       (float64, rows in order) and everything BatchNorm derives from them in the same workgroup
//   tr_fc_bwd_w_kernel  per FC_CB channels: dy = da [bn(z) > 0], the BatchNorm-backward sums, dz = k1 (dy - k2 - xhat k3),
//                       dgamma, dbeta, db, and dW[c][:] = sum_r dz[r][c] act(a_prev[r][:])
//   tr_fc_fwd_kernel<TRANS>  da_prev = dz W (the same product with the weight read transposed)
// These launches are latency, not arithmetic (17 MFLOP a layer): a workgroup walks the reduction dimension in chunks of
// 8192 / B-rounded-up values per row (128 for B <= 64), every load of a chunk — 16 bytes per lane where the shapes allow —
// issued before the first is used, and the next chunk's loads issued before the current chunk is computed from LDS.
// Deterministic: every sum runs over the rows (or k) in index order. No padding, no packed weight image.
#include <stdint.h>

#include <type_traits>

#include <hip/hip_runtime.h>

#include "dal3_kernels.h"

#define FC_CB 4                          // output channels per workgroup
#define FC_RS 64                         // row slots per workgroup (256 threads / FC_CB)
#define FC_MAX_ROWS 256                  // rows per thread <= 4
#define FC_RPT (FC_MAX_ROWS / FC_RS)
#define FC_STAGE 8192                    // floats of input rows staged at a time: rows x chunk
#define FC_NLD (FC_STAGE / 4 / 256)      // 16-byte loads per thread and chunk

typedef float fc_f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int fc_chunk(int B) { return B <= 64 ? 128 : B <= 128 ? 64 : 32; }

// One chunk [k0, k0 + KC) of the rows of `a` (B x c_in, ld lda) on its way to LDS: load() puts this thread's share into
// registers (everything in flight at once), store() applies the input activation and writes a_s[r * (KC + 1) + k].
// VEC: c_in, lda multiples of 4 and a 16-byte aligned — 16 bytes per load; else element by element.
#define FC_MAX_ACT_CIN 2048                // most input channels of a layer whose input carries an activation (LDS copy)
template <bool VEC>
struct FcStage {
    fc_f4 v[FC_NLD];
    // lk = log2(KC): KC is 32, 64 or 128
    __device__ __forceinline__ void load(const float* __restrict__ a, int B, int c_in, int64_t lda, int k0, int lk) {
        if (VEC) {
#pragma unroll
            for (int j = 0; j < FC_NLD; ++j) {
                const int i = threadIdx.x + 256 * j, r = i >> (lk - 2), k = k0 + 4 * (i & ((1 << (lk - 2)) - 1));
                v[j] = (r < B && k < c_in) ? *reinterpret_cast<const fc_f4*>(a + (int64_t)r * lda + k) : fc_f4{0.0f, 0.0f, 0.0f, 0.0f};
            }
        } else {
#pragma unroll
            for (int j = 0; j < FC_NLD; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int i = threadIdx.x + 256 * (4 * j + e), r = i >> lk, k = k0 + (i & ((1 << lk) - 1));
                    v[j][e] = (r < B && k < c_in) ? a[(int64_t)r * lda + k] : 0.0f;
                }
        }
    }
    // s_sc / s_sh: the input activation's scale and shift in LDS (the whole reduction dimension), or act == false
    __device__ __forceinline__ void store(float* __restrict__ a_s, int B, int c_in, int k0, int lk, bool act,
                                          const float* __restrict__ s_sc, const float* __restrict__ s_sh, int relu_in) const {
        const int KC = 1 << lk;
#pragma unroll
        for (int j = 0; j < FC_NLD; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int r, kk;
                if (VEC) {
                    const int i = threadIdx.x + 256 * j;
                    r = i >> (lk - 2);
                    kk = 4 * (i & ((1 << (lk - 2)) - 1)) + e;
                } else {
                    const int i = threadIdx.x + 256 * (4 * j + e);
                    r = i >> lk;
                    kk = i & (KC - 1);
                }
                if (r >= B) continue;
                float x = v[j][e];
                if (act && k0 + kk < c_in) {                    // (columns past c_in stay zero)
                    x = __builtin_fmaf(x, s_sc[k0 + kk], s_sh[k0 + kk]);
                    x = relu_in ? fmaxf(x, 0.0f) : x;
                }
                a_s[r * (KC + 1) + kk] = x;
            }
    }
};

// the input activation's per-channel constants into LDS (c_in <= FC_MAX_ACT_CIN: the host checks)
__device__ __forceinline__ void fc_act_to_lds(float* s_sc, float* s_sh, const float* __restrict__ in_scale,
                                              const float* __restrict__ in_shift, int c_in) {
    if (!in_scale) return;
    for (int i = threadIdx.x; i < c_in; i += 256) {
        s_sc[i] = in_scale[i];
        s_sh[i] = in_shift[i];
    }
}

// sum over the 64 row slots of a workgroup for each of its FC_CB channels (thread = (channel tid % 4, slot tid / 4)): a fixed
// tree inside the wave (lanes 4 apart, then 8, 16, 32), then the four waves in order; valid in threads 0 .. FC_CB - 1
__device__ __forceinline__ void fc_sum_slots(double& s0, double& s1, double (*red)[4][FC_CB]) {
#pragma unroll
    for (int off = 4; off < 64; off <<= 1) {
        s0 += __shfl_xor(s0, off, 64);
        s1 += __shfl_xor(s1, off, 64);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane < FC_CB) {
        red[0][wave][lane] = s0;
        red[1][wave][lane] = s1;
    }
    __syncthreads();
    if (threadIdx.x < FC_CB) {
        s0 = ((red[0][0][threadIdx.x] + red[0][1][threadIdx.x]) + red[0][2][threadIdx.x]) + red[0][3][threadIdx.x];
        s1 = ((red[1][0][threadIdx.x] + red[1][1][threadIdx.x]) + red[1][2][threadIdx.x]) + red[1][3][threadIdx.x];
    }
}

struct FcBn {                            // BatchNorm of the layer being computed (gamma == NULL: none)
    const float* gamma;
    const float* beta;
    float* running_mean;
    float* running_var;
    float momentum, eps;
    float* mu;
    float* rstd;
    float* scale;
    float* shift;
};

// z[r][c] = sum_k act(a[r][k]) Wop[c][k] + bias[c];  Wop[c][k] = TRANS ? W[k][c] : W[c][k]  (W row-major, ldw)
// grid: ceil(c_out / FC_CB); thread = (channel tid % FC_CB, row slot tid / FC_CB); rows slot, slot + 64, ...
template <bool TRANS, bool VEC>
__global__ __launch_bounds__(256) void tr_fc_fwd_kernel(const float* __restrict__ a, int B, int c_in, int64_t lda,
                                                        const float* __restrict__ in_scale, const float* __restrict__ in_shift,
                                                        int relu_in, const float* __restrict__ W, int64_t ldw,
                                                        const float* __restrict__ bias, int c_out, float* __restrict__ z,
                                                        int64_t ldz, FcBn bn) {
    __shared__ float a_s[FC_STAGE + FC_MAX_ROWS];
    __shared__ float w_s[FC_CB][128 + 1];
    __shared__ float s_sc[FC_MAX_ACT_CIN], s_sh[FC_MAX_ACT_CIN];
    __shared__ double red[2][4][FC_CB];
    const int cl = threadIdx.x % FC_CB, slot = threadIdx.x / FC_CB;
    const int c0 = blockIdx.x * FC_CB, c = c0 + cl;
    const int KC = fc_chunk(B), lk = KC == 128 ? 7 : KC == 64 ? 6 : 5, nrow = (B + FC_RS - 1) / FC_RS;
    const bool act = in_scale != nullptr;
    float acc[FC_RPT] = {0.0f, 0.0f, 0.0f, 0.0f};
    FcStage<VEC> st;
    float wv[2];                                                // this thread's share of the weight chunk (FC_CB x KC <= 512 values)
    const auto load_w = [&](int k0) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = threadIdx.x + 256 * u;
            const int wc = TRANS ? i % FC_CB : i / KC, k = TRANS ? i / FC_CB : i % KC;
            wv[u] = 0.0f;
            if (i < FC_CB * KC && k0 + k < c_in && c0 + wc < c_out)
                wv[u] = TRANS ? W[(int64_t)(k0 + k) * ldw + c0 + wc] : W[(int64_t)(c0 + wc) * ldw + k0 + k];
        }
    };
    const auto store_w = [&]() {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int i = threadIdx.x + 256 * u;
            const int wc = TRANS ? i % FC_CB : i / KC, k = TRANS ? i / FC_CB : i % KC;
            if (i < FC_CB * KC) w_s[wc][k] = wv[u];
        }
    };
    st.load(a, B, c_in, lda, 0, lk);
    load_w(0);
    fc_act_to_lds(s_sc, s_sh, in_scale, in_shift, c_in);
    if (act) __syncthreads();
    for (int k0 = 0; k0 < c_in; k0 += KC) {
        st.store(a_s, B, c_in, k0, lk, act, s_sc, s_sh, relu_in);
        store_w();
        __syncthreads();
        if (k0 + KC < c_in) {                                   // the next chunk's loads fly while this one is computed
            st.load(a, B, c_in, lda, k0 + KC, lk);
            load_w(k0 + KC);
        }
        const auto compute = [&](auto nr_c) {
            constexpr int NR = decltype(nr_c)::value;
            const float* ar = a_s + slot * (KC + 1);
#pragma unroll 8
            for (int k = 0; k < KC; ++k) {
                const float w = w_s[cl][k];
#pragma unroll
                for (int i = 0; i < NR; ++i) acc[i] = __builtin_fmaf(ar[FC_RS * i * (KC + 1) + k], w, acc[i]);
            }
        };
        if (nrow == 1) compute(std::integral_constant<int, 1>{});
        else if (nrow == 2) compute(std::integral_constant<int, 2>{});
        else compute(std::integral_constant<int, 4>{});
        __syncthreads();
    }
    const float b = (bias && c < c_out) ? bias[c] : 0.0f;
    double s0 = 0.0, s1 = 0.0;
#pragma unroll
    for (int i = 0; i < FC_RPT; ++i) {
        const int r = slot + FC_RS * i;
        if (r < B && c < c_out) {
            const float v = acc[i] + b;
            z[(int64_t)r * ldz + c] = v;
            s0 += (double)v;
            s1 = __builtin_fma((double)v, (double)v, s1);
        }
    }
    if (!bn.gamma) return;
    fc_sum_slots(s0, s1, red);
    if (threadIdx.x < FC_CB && c0 + (int)threadIdx.x < c_out) {
        const int cc = c0 + threadIdx.x;
        const double t0 = s0, t1 = s1;
        const double mean = t0 / (double)B;
        double var = t1 / (double)B - mean * mean;
        var = var > 0.0 ? var : 0.0;
        const double rs = 1.0 / sqrt(var + (double)bn.eps);
        const double sc = (double)bn.gamma[cc] * rs;
        bn.mu[cc] = (float)mean;
        bn.rstd[cc] = (float)rs;
        bn.scale[cc] = (float)sc;
        bn.shift[cc] = (float)((double)bn.beta[cc] - mean * sc);
        if (bn.running_mean) {
            bn.running_mean[cc] = (1.0f - bn.momentum) * bn.running_mean[cc] + bn.momentum * (float)mean;
            bn.running_var[cc] = (1.0f - bn.momentum) * bn.running_var[cc] + bn.momentum * (float)(var * ((double)B / (double)(B - 1)));
        }
    }
}

struct FcBwd {                           // the layer's own BatchNorm in the backward (scale == NULL: none, dz = da)
    const float* z;
    int64_t ldz;
    const float* scale;
    const float* shift;
    const float* mu;
    const float* rstd;
    const float* gamma;
    float* dgamma;
    float* dbeta;
};

// per FC_CB channels of the layer's output: dz (written, ld lddz), dgamma / dbeta, db, dW[c][0..c_in)
template <bool VEC>
__global__ __launch_bounds__(256) void tr_fc_bwd_w_kernel(const float* __restrict__ da, int64_t ldda, int B, int c_out, FcBwd bn,
                                                          const float* __restrict__ a_prev, int c_in, int64_t lda,
                                                          const float* __restrict__ in_scale, const float* __restrict__ in_shift,
                                                          int relu_in, float* __restrict__ dz, int64_t lddz,
                                                          float* __restrict__ dW, int64_t lddw, float* __restrict__ db) {
    __shared__ float a_s[FC_STAGE + FC_MAX_ROWS];
    __shared__ float dz_s[FC_MAX_ROWS][FC_CB];
    __shared__ float s_sc[FC_MAX_ACT_CIN], s_sh[FC_MAX_ACT_CIN];
    __shared__ double red[2][4][FC_CB];
    __shared__ float coef[3][FC_CB];
    const int cl = threadIdx.x % FC_CB, slot = threadIdx.x / FC_CB;
    const int c0 = blockIdx.x * FC_CB, c = c0 + cl;
    const bool live = c < c_out;
    const int KC = fc_chunk(B), lk = KC == 128 ? 7 : KC == 64 ? 6 : 5;
    const bool act = in_scale != nullptr;
    FcStage<VEC> st;
    if (dW) {                                                   // (under the BatchNorm-backward phase)
        st.load(a_prev, B, c_in, lda, 0, lk);
        fc_act_to_lds(s_sc, s_sh, in_scale, in_shift, c_in);
    }
    float dy[FC_RPT], xh[FC_RPT];
    double s0 = 0.0, s1 = 0.0;
    const float sc = (bn.scale && live) ? bn.scale[c] : 1.0f, sh = (bn.scale && live) ? bn.shift[c] : 0.0f;
    const float mu = (bn.scale && live) ? bn.mu[c] : 0.0f, rs = (bn.scale && live) ? bn.rstd[c] : 1.0f;
#pragma unroll
    for (int i = 0; i < FC_RPT; ++i) {
        const int r = slot + FC_RS * i;
        dy[i] = 0.0f;
        xh[i] = 0.0f;
        if (r < B && live) {
            const float g = da[(int64_t)r * ldda + c];
            if (bn.scale) {
                const float zz = bn.z[(int64_t)r * bn.ldz + c];
                dy[i] = __builtin_fmaf(zz, sc, sh) > 0.0f ? g : 0.0f;
                xh[i] = (zz - mu) * rs;
            } else {
                dy[i] = g;
            }
            s0 += (double)dy[i];
            s1 = __builtin_fma((double)dy[i], (double)xh[i], s1);
        }
    }
    fc_sum_slots(s0, s1, red);
    if (threadIdx.x < FC_CB) {
        const double t0 = s0, t1 = s1;
        const int cc = c0 + threadIdx.x;
        if (cc < c_out) {
            if (bn.scale) {
                bn.dbeta[cc] = (float)t0;
                bn.dgamma[cc] = (float)t1;
                coef[0][threadIdx.x] = bn.gamma[cc] * bn.rstd[cc];
                coef[1][threadIdx.x] = (float)(t0 / (double)B);
                coef[2][threadIdx.x] = (float)(t1 / (double)B);
                if (db) db[cc] = 0.0f;                          // (a bias in front of a train-mode BatchNorm: sum dz = 0)
            } else {
                coef[0][threadIdx.x] = 1.0f;
                coef[1][threadIdx.x] = 0.0f;
                coef[2][threadIdx.x] = 0.0f;
                if (db) db[cc] = (float)t0;
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < FC_RPT; ++i) {
        const int r = slot + FC_RS * i;
        if (r < B) {
            const float v = live ? coef[0][cl] * (dy[i] - coef[1][cl] - xh[i] * coef[2][cl]) : 0.0f;
            dz_s[r][cl] = v;
            if (live && dz) dz[(int64_t)r * lddz + c] = v;
        }
    }
    // dW[c][k] = sum_r dz[r][c] act(a_prev[r][k]), a chunk of the input channels at a time through LDS: thread = (channel,
    // k = slot, slot + 64 of the chunk), rows in order
    if (!dW) return;
    for (int k0 = 0; k0 < c_in; k0 += KC) {
        st.store(a_s, B, c_in, k0, lk, act, s_sc, s_sh, relu_in);
        __syncthreads();                                        // (the first pass: dz_s too)
        if (k0 + KC < c_in) st.load(a_prev, B, c_in, lda, k0 + KC, lk);
        for (int kk = slot; kk < KC; kk += FC_RS) {
            float t = 0.0f;
#pragma unroll 4
            for (int r = 0; r < B; ++r) t = __builtin_fmaf(dz_s[r][cl], a_s[r * (KC + 1) + kk], t);
            if (live && k0 + kk < c_in) dW[(int64_t)c * lddw + k0 + kk] = t;
        }
        __syncthreads();
    }
}

static bool fc_vec_ok(const float* a, int c_in, int64_t lda) {
    return c_in % 4 == 0 && lda % 4 == 0 && (reinterpret_cast<uintptr_t>(a) & 15) == 0;
}

hipError_t launch_tr_fc_forward(const float* a, int B, int c_in, int64_t lda, const float* in_scale, const float* in_shift, int relu_in,
                                const float* W, int64_t ldw, int transpose_w, const float* bias, int c_out, float* z, int64_t ldz,
                                const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum,
                                float eps, float* mu, float* rstd, float* scale, float* shift, hipStream_t s) {
    const FcBn bn{gamma, beta, running_mean, running_var, momentum, eps, mu, rstd, scale, shift};
    const dim3 grid((unsigned)((c_out + FC_CB - 1) / FC_CB));
    const auto go = [&](auto kern) {
        hipLaunchKernelGGL(kern, grid, dim3(256), 0, s, a, B, c_in, lda, in_scale, in_shift, relu_in, W, ldw, bias, c_out, z, ldz, bn);
    };
    if (fc_vec_ok(a, c_in, lda)) {
        if (transpose_w) go(tr_fc_fwd_kernel<true, true>); else go(tr_fc_fwd_kernel<false, true>);
    } else {
        if (transpose_w) go(tr_fc_fwd_kernel<true, false>); else go(tr_fc_fwd_kernel<false, false>);
    }
    return hipGetLastError();
}

hipError_t launch_tr_fc_backward_w(const float* da, int64_t ldda, int B, int c_out, const float* z, int64_t ldz, const float* scale,
                                   const float* shift, const float* mu, const float* rstd, const float* gamma, float* dgamma,
                                   float* dbeta, const float* a_prev, int c_in, int64_t lda, const float* in_scale,
                                   const float* in_shift, int relu_in, float* dz, int64_t lddz, float* dW, int64_t lddw, float* db,
                                   hipStream_t s) {
    const FcBwd bn{z, ldz, scale, shift, mu, rstd, gamma, dgamma, dbeta};
    const dim3 grid((unsigned)((c_out + FC_CB - 1) / FC_CB));
    if (dW && fc_vec_ok(a_prev, c_in, lda))
        hipLaunchKernelGGL(tr_fc_bwd_w_kernel<true>, grid, dim3(256), 0, s, da, ldda, B, c_out, bn, a_prev, c_in, lda, in_scale, in_shift,
                           relu_in, dz, lddz, dW, lddw, db);
    else
        hipLaunchKernelGGL(tr_fc_bwd_w_kernel<false>, grid, dim3(256), 0, s, da, ldda, B, c_out, bn, a_prev, c_in, lda, in_scale, in_shift,
                           relu_in, dz, lddz, dW, lddw, db);
    return hipGetLastError();
}

int tr_fc_max_rows() { return FC_MAX_ROWS; }
int tr_fc_max_act_cin() { return FC_MAX_ACT_CIN; }
