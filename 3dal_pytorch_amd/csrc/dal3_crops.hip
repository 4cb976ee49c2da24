// dal3_crops.hip — crop extraction from full sweeps: the per-detection loop of
// _create_pd_detection (det3d/datasets/waymo/waymo_common.py:166-171) — for every tracked detection of a frame, the
// sweep's points inside its rotated box, moved to the global frame with the frame's veh_to_global — for many frames
// in one pass. HBM-bound integer/compare work: every sweep point is read once per pass (12 B), tested against the
// frame's boxes whose face equations sit in scalar registers, and only the members are written.
//
// Ordered output without a sort: a wavefront owns a chunk of CROP_CHUNK consecutive points of one frame.
//   pass 1 (count):  cc[box][chunk] = members of the chunk                       (ballot + popcount)
//   scan:            per box, exclusive prefix over its frame's chunks; counts[box] = total
//   (caller: exclusive prefix of counts over boxes -> box_start, and the grand total to size the output)
//   pass 2 (fill):   recompute membership; member rank inside the chunk = running popcount + lanes-below mask
// so each detection's points come out in sweep order, exactly as `lidars[indices]` gives them.
#include "dal3_geom.h"
#include "dal3_kernels.h"

#define CROP_CHUNK 1024                  // points per wavefront: 16 rounds of 64
#define CROP_ROUNDS (CROP_CHUNK / 64)
#define CROP_WAVES 4

__device__ __forceinline__ int lanes_below(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
}

// Conservative cull before the exact test: a sphere around the detection (centre = box centre, radius = half
// diagonal + a margin far above fp32 rounding of the face equations). One subtract-square-add chain per point and
// box instead of six plane evaluations; a round of 64 points goes on to the exact test only when some lane is a
// candidate. Non-finite distances (NaN / inf coordinates) are always candidates, so the exact test alone decides
// membership, as in the reference. d2 >= 0, so its bit pattern orders like the value.
__device__ __forceinline__ bool sphere_candidate(float x, float y, float z, float cx, float cy, float cz, uint32_t r2_bits) {
    const float dx = x - cx, dy = y - cy, dz = z - cz;
    const uint32_t b = __float_as_uint(dx * dx + dy * dy + dz * dz) & 0x7fffffffu;
    return b <= r2_bits || b >= 0x7f800000u;
}

template <bool FILL>
__global__ __launch_bounds__(64 * CROP_WAVES) void crop_pass_kernel(
    const float* __restrict__ points, const int64_t* __restrict__ point_offsets, const double* __restrict__ planes,
    const float* __restrict__ spheres, const int64_t* __restrict__ box_offsets, int max_chunks, int32_t* __restrict__ cc,
    const int64_t* __restrict__ counts, const double* __restrict__ pose_all, const int64_t* __restrict__ box_start,
    double* __restrict__ out_points, int32_t* __restrict__ out_index) {
    const int frame = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int chunk = blockIdx.x * CROP_WAVES + (threadIdx.x >> 6);
    const int64_t p0 = point_offsets[frame], n_pts = point_offsets[frame + 1] - p0;
    if ((int64_t)chunk * CROP_CHUNK >= n_pts) return;
    const bool last_chunk = (int64_t)(chunk + 1) * CROP_CHUNK >= n_pts;
    const int64_t k0 = box_offsets[frame], k1 = box_offsets[frame + 1];
    float x[CROP_ROUNDS], y[CROP_ROUNDS], z[CROP_ROUNDS];
    uint32_t valid = 0;
#pragma unroll
    for (int r = 0; r < CROP_ROUNDS; ++r) {
        const int64_t i = (int64_t)chunk * CROP_CHUNK + r * 64 + lane;
        const bool ok = i < n_pts;
        const float* p = points + (p0 + (ok ? i : 0)) * 3;
        x[r] = p[0];
        y[r] = p[1];
        z[r] = p[2];
        valid |= (uint32_t)ok << r;
    }
    const double* pose = FILL ? pose_all + (int64_t)frame * 16 : nullptr;
    for (int64_t k = k0; k < k1; ++k) {
        const double* pl = planes + k * DAL3_PLANE_DOUBLES;              // uniform address: scalar loads
        int64_t base = 0;
        if (FILL) {                                                      // nothing of this detection in this chunk: skip
            const int32_t excl = cc[k * max_chunks + chunk];
            const int64_t next = last_chunk ? counts[k] : (int64_t)cc[k * max_chunks + chunk + 1];
            if (next == excl) continue;
            base = box_start[k] + excl;
        }
        const float cx = spheres[k * 4 + 0], cy = spheres[k * 4 + 1], cz = spheres[k * 4 + 2];
        const uint32_t r2 = __float_as_uint(spheres[k * 4 + 3]);
        int cnt = 0;
#pragma unroll
        for (int r = 0; r < CROP_ROUNDS; ++r) {
            const bool cand = ((valid >> r) & 1u) && sphere_candidate(x[r], y[r], z[r], cx, cy, cz, r2);
            if (__ballot(cand) == 0) continue;                           // wave-uniform
            const bool in = cand && inside_box_f32(pl, x[r], y[r], z[r]);
            const uint64_t m = __ballot(in);
            if (FILL) {
                if (in) {
                    const int64_t o = base + lanes_below(m);
                    const double px = x[r], py = y[r], pz = z[r];        // concatenate([lidars_o, ones]) is float64
                    out_points[o * 3 + 0] = pose[0] * px + pose[1] * py + pose[2] * pz + pose[3];
                    out_points[o * 3 + 1] = pose[4] * px + pose[5] * py + pose[6] * pz + pose[7];
                    out_points[o * 3 + 2] = pose[8] * px + pose[9] * py + pose[10] * pz + pose[11];
                    if (out_index) out_index[o] = (int32_t)((int64_t)chunk * CROP_CHUNK + r * 64 + lane);
                }
                base += __popcll(m);
            } else {
                cnt += __popcll(m);
            }
        }
        if (!FILL && lane == 0) cc[k * max_chunks + chunk] = cnt;
    }
}

// one wavefront per box: exclusive scan of its chunk counts in place, total to counts[box]
__global__ void crop_scan_kernel(const int64_t* __restrict__ point_offsets, const int64_t* __restrict__ box_offsets, int F,
                                 int max_chunks, int32_t* __restrict__ cc, int64_t* __restrict__ counts) {
    const int64_t k = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (k >= box_offsets[F]) return;
    int lo = 0, hi = F;                                                  // frame of box k: last f with box_offsets[f] <= k
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (box_offsets[mid] <= k) lo = mid; else hi = mid;
    }
    const int64_t n_pts = point_offsets[lo + 1] - point_offsets[lo];
    const int n_chunks = (int)((n_pts + CROP_CHUNK - 1) / CROP_CHUNK);
    int32_t* row = cc + k * max_chunks;
    int64_t run = 0;
    for (int c0 = 0; c0 < n_chunks; c0 += 64) {
        const int c = c0 + lane;
        const int v = c < n_chunks ? row[c] : 0;
        int incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int up = __shfl_up(incl, d);
            if (lane >= d) incl += up;
        }
        if (c < n_chunks) row[c] = (int32_t)(run + incl - v);
        run += __shfl(incl, 63);
    }
    if (lane == 0) counts[k] = run;
}

size_t crop_workspace_bytes(int64_t K_total, int64_t max_points_per_frame) {
    const int64_t max_chunks = (max_points_per_frame + CROP_CHUNK - 1) / CROP_CHUNK;
    return (size_t)(K_total * (max_chunks > 0 ? max_chunks : 1)) * sizeof(int32_t);
}

hipError_t launch_crop_count(const float* points, const int64_t* point_offsets, const double* planes,
                             const float* spheres, const int64_t* box_offsets, int F, int64_t K_total, int64_t max_points_per_frame,
                             int64_t* counts, int32_t* cc, hipStream_t s) {
    const int max_chunks = (int)((max_points_per_frame + CROP_CHUNK - 1) / CROP_CHUNK);
    if (F > 0 && max_chunks > 0 && K_total > 0)
        hipLaunchKernelGGL(crop_pass_kernel<false>, dim3((max_chunks + CROP_WAVES - 1) / CROP_WAVES, F), dim3(64 * CROP_WAVES),
                           0, s, points, point_offsets, planes, spheres, box_offsets, max_chunks, cc, (const int64_t*)nullptr,
                           (const double*)nullptr, (const int64_t*)nullptr, (double*)nullptr, (int32_t*)nullptr);
    if (K_total > 0)
        hipLaunchKernelGGL(crop_scan_kernel, dim3((unsigned)((K_total + 3) / 4)), dim3(256), 0, s, point_offsets, box_offsets,
                           F, max_chunks > 0 ? max_chunks : 1, cc, counts);
    return hipGetLastError();
}

hipError_t launch_crop_fill(const float* points, const int64_t* point_offsets, const double* planes,
                            const float* spheres, const int64_t* box_offsets, int F, int64_t K_total,
                            int64_t max_points_per_frame, const double* pose, const int64_t* counts,
                            const int64_t* box_start, const int32_t* cc, double* out_points, int32_t* out_index,
                            hipStream_t s) {
    const int max_chunks = (int)((max_points_per_frame + CROP_CHUNK - 1) / CROP_CHUNK);
    if (F > 0 && max_chunks > 0 && K_total > 0)
        hipLaunchKernelGGL(crop_pass_kernel<true>, dim3((max_chunks + CROP_WAVES - 1) / CROP_WAVES, F), dim3(64 * CROP_WAVES),
                           0, s, points, point_offsets, planes, spheres, box_offsets, max_chunks, const_cast<int32_t*>(cc),
                           counts, pose, box_start, out_points, out_index);
    return hipGetLastError();
}

// ---------------------------------------------------------------- plain membership table (P,K), any dtype mix
template <typename T>
__global__ void points_in_boxes_kernel(const T* __restrict__ points, int64_t P, int64_t stride,
                                       const double* __restrict__ planes, int K, int f32_math,
                                       uint8_t* __restrict__ inside) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= P) return;
    const T* p = points + i * stride;
    const T x = p[0], y = p[1], z = p[2];
    for (int k = 0; k < K; ++k) {
        const double* pl = planes + (int64_t)k * DAL3_PLANE_DOUBLES;
        const bool in = f32_math ? inside_box_f32(pl, (float)x, (float)y, (float)z)
                                 : inside_box_f64(pl, (double)x, (double)y, (double)z);
        inside[i * K + k] = in ? 1 : 0;
    }
}

hipError_t launch_points_in_boxes(const void* points, int points_f64, int64_t P, int64_t stride, const double* planes,
                                  int K, int f32_math, uint8_t* inside, hipStream_t s) {
    if (P <= 0 || K <= 0) return hipSuccess;
    const dim3 grid((unsigned)((P + 255) / 256));
    if (points_f64)
        hipLaunchKernelGGL(points_in_boxes_kernel<double>, grid, dim3(256), 0, s, (const double*)points, P, stride, planes, K,
                           0, inside);
    else
        hipLaunchKernelGGL(points_in_boxes_kernel<float>, grid, dim3(256), 0, s, (const float*)points, P, stride, planes, K,
                           f32_math, inside);
    return hipGetLastError();
}
