// dal3_prep.hip — crop preparation on the device: what STATICTRACK.__getitem__
// (tools/static_model.py:529-572) and DYNAMICTRACK.__getitem__ (tools/dynamic_model.py:419-509) do per item on
// the host — resample with replacement, global -> vehicle frame, re-centre on the box, rotate by -yaw, assemble
// the 5-frame point window and the 101-box window — as two HBM-bound gather kernels. float64 arithmetic as in the
// reference (global coordinates are kilometres; fp32 would lose centimetres), fp32 on store (the drivers' .float()).
#include "dal3_device.h"
#include "dal3_geom.h"
#include "dal3_kernels.h"

__device__ __forceinline__ uint32_t prep_hash(uint64_t seed, uint64_t item, uint32_t i) {
    uint64_t z = seed ^ (item * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)i << 32 | i);
    z += 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (uint32_t)(z >> 32);
}

// p' = Rz(-yaw) ( pose[:3,:3] p + pose[:3,3] - c ), pose row-major 4x4 (vehicle <- global)
__device__ __forceinline__ void to_box_frame(const double* __restrict__ pose, double x, double y, double z, double cx,
                                             double cy, double cz, double cosn, double sinn, float* __restrict__ o) {
    const double vx = pose[0] * x + pose[1] * y + pose[2] * z + pose[3] - cx;
    const double vy = pose[4] * x + pose[5] * y + pose[6] * z + pose[7] - cy;
    const double vz = pose[8] * x + pose[9] * y + pose[10] * z + pose[11] - cz;
    o[0] = (float)(cosn * vx - sinn * vy);             // rotz(-yaw) = [[c, s],[-s, c]] with c = cos(yaw), s = sin(yaw):
    o[1] = (float)(sinn * vx + cosn * vy);             //   passed in as cosn = cos(-yaw), sinn = sin(-yaw)
    o[2] = (float)vz;
}

// static: one thread per output point. box = best detection already in the vehicle frame (host: transform_box).
__global__ void static_crop_prep_kernel(const double* __restrict__ points, const int64_t* __restrict__ offsets,
                                        const int32_t* __restrict__ choice, const double* __restrict__ pose,
                                        const double* __restrict__ box, int B, int N, uint64_t seed, int64_t item_offset,
                                        float* __restrict__ pts_out, float* __restrict__ init_box_out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * N) return;
    const int b = (int)(i / N), n = (int)(i % N);
    const int64_t p0 = offsets[b], cnt = offsets[b + 1] - p0;
    const double* bx = box + (int64_t)b * 7;
    if (n < 7) init_box_out[b * 7 + n] = (float)bx[n];
    float* o = pts_out + i * 3;
    if (cnt <= 0) {
        o[0] = o[1] = o[2] = 0.0f;
        return;
    }
    int64_t k = choice ? choice[i] : (int64_t)(((uint64_t)prep_hash(seed, (uint64_t)(item_offset + b), (uint32_t)n) * (uint64_t)cnt) >> 32);
    k = k < 0 ? 0 : (k >= cnt ? cnt - 1 : k);
    const double* p = points + (p0 + k) * 3;
    const double yaw = -bx[6];
    to_box_frame(pose + (int64_t)b * 16, p[0], p[1], p[2], bx[0], bx[1], bx[2], cos(yaw), sin(yaw), o);
}

// dynamic: blocks [0, B*(2r+1)*n_per) -> points, then B*(2s+1) threads -> boxes
__global__ void dynamic_item_prep_kernel(const double* __restrict__ points, const int64_t* __restrict__ frame_offsets,
                                         const double* __restrict__ boxes, const int64_t* __restrict__ track_first,
                                         const int32_t* __restrict__ item_track, const int32_t* __restrict__ item_frame,
                                         const int32_t* __restrict__ choice, const double* __restrict__ pose_all, int B,
                                         int n_per, int r, int s, uint64_t seed, int64_t item_offset,
                                         float* __restrict__ pts_out, float* __restrict__ box_out,
                                         float* __restrict__ init_box_out) {
    const int64_t n_pts_total = (int64_t)B * (2 * r + 1) * n_per;
    const int64_t n_box_total = (int64_t)B * (2 * s + 1);
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pts_total + n_box_total) return;
    const bool is_pt = i < n_pts_total;
    const int b = is_pt ? (int)(i / ((int64_t)(2 * r + 1) * n_per)) : (int)((i - n_pts_total) / (2 * s + 1));
    const double* pose = pose_all + (int64_t)b * 16;
    const int trk = item_track[b], it = item_frame[b];
    const int64_t f0 = track_first[trk];
    const int n_frames = (int)(track_first[trk + 1] - f0);
    // centre box in the vehicle frame (transform_box of the item's own box; out-of-range cannot happen for it)
    const double* cb = boxes + (f0 + it) * 7;
    const double ccx = pose[0] * cb[0] + pose[1] * cb[1] + pose[2] * cb[2] + pose[3];
    const double ccy = pose[4] * cb[0] + pose[5] * cb[1] + pose[6] * cb[2] + pose[7];
    const double ccz = pose[8] * cb[0] + pose[9] * cb[1] + pose[10] * cb[2] + pose[11];
    const double dyaw = atan2(pose[4], pose[0]);
    const double cyaw = cb[6] + dyaw;
    if (is_pt) {
        const int64_t w = i - (int64_t)b * (2 * r + 1) * n_per;
        const int j = (int)(w / n_per);
        const int fr = it - r + j;
        double x = 0.0, y = 0.0, z = 0.0;                 // missing / empty frame: zero points, still transformed
        if (fr >= 0 && fr < n_frames) {
            const int64_t p0 = frame_offsets[f0 + fr], cnt = frame_offsets[f0 + fr + 1] - p0;
            if (cnt > 0) {
                int64_t k = choice ? choice[i] : (int64_t)(((uint64_t)prep_hash(seed, (uint64_t)(item_offset + b), (uint32_t)w) * (uint64_t)cnt) >> 32);
                k = k < 0 ? 0 : (k >= cnt ? cnt - 1 : k);
                const double* p = points + (p0 + k) * 3;
                x = p[0];
                y = p[1];
                z = p[2];
            }
        }
        float* o = pts_out + i * 4;
        to_box_frame(pose, x, y, z, ccx, ccy, ccz, cos(-cyaw), sin(-cyaw), o);
        o[3] = (float)(0.1 * (j - r));
    } else {
        const int j = (int)((i - n_pts_total) % (2 * s + 1));
        const int fr = it - s + j;
        double bx[7] = {0, 0, 0, 0, 0, 0, 0};            // missing box: a zero row, still pose-transformed
        if (fr >= 0 && fr < n_frames) {
            const double* q = boxes + (f0 + fr) * 7;
#pragma unroll
            for (int k = 0; k < 7; ++k) bx[k] = q[k];
        }
        const double tx = pose[0] * bx[0] + pose[1] * bx[1] + pose[2] * bx[2] + pose[3];
        const double ty = pose[4] * bx[0] + pose[5] * bx[1] + pose[6] * bx[2] + pose[7];
        const double tz = pose[8] * bx[0] + pose[9] * bx[1] + pose[10] * bx[2] + pose[11];
        const double tyaw = bx[6] + dyaw;
        float* o = box_out + ((int64_t)b * (2 * s + 1) + j) * 8;
        o[0] = (float)(tx - ccx);
        o[1] = (float)(ty - ccy);
        o[2] = (float)(tz - ccz);
        o[3] = (float)bx[3];
        o[4] = (float)bx[4];
        o[5] = (float)bx[5];
        o[6] = (float)(tyaw - cyaw);
        o[7] = (float)(0.1 * (j - s));
        if (j == s) {                                     // init_box = the centre box BEFORE re-centring
            float* ib = init_box_out + (int64_t)b * 8;
            ib[0] = (float)tx;
            ib[1] = (float)ty;
            ib[2] = (float)tz;
            ib[3] = (float)bx[3];
            ib[4] = (float)bx[4];
            ib[5] = (float)bx[5];
            ib[6] = (float)tyaw;
            ib[7] = 0.0f;
        }
    }
}

// ---- the mask labels of the same items (training only): static_model.py:548-556, dynamic_model.py:455-487.
// The SAME draws as the prep kernels (choice, or the device hash of (seed, item, n)), the same float64 vehicle-frame
// point, tested against the matched annotation's box with the reference's points_in_rbbox (dal3_geom.h; float64
// points against float32-valued face equations -> float64 arithmetic).
__global__ void static_crop_labels_kernel(const double* __restrict__ points, const int64_t* __restrict__ offsets,
                                          const int32_t* __restrict__ choice, const double* __restrict__ pose_all,
                                          int B, int N, uint64_t seed, int64_t item_offset,
                                          const double* __restrict__ gt_planes, uint8_t* __restrict__ mask_label) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)B * N) return;
    const int b = (int)(i / N), n = (int)(i % N);
    const int64_t p0 = offsets[b], cnt = offsets[b + 1] - p0;
    if (cnt <= 0) {
        mask_label[i] = 0;
        return;
    }
    int64_t k = choice ? choice[i] : (int64_t)(((uint64_t)prep_hash(seed, (uint64_t)(item_offset + b), (uint32_t)n) * (uint64_t)cnt) >> 32);
    k = k < 0 ? 0 : (k >= cnt ? cnt - 1 : k);
    const double* p = points + (p0 + k) * 3;
    const double* pose = pose_all + (int64_t)b * 16;
    const double vx = pose[0] * p[0] + pose[1] * p[1] + pose[2] * p[2] + pose[3];
    const double vy = pose[4] * p[0] + pose[5] * p[1] + pose[6] * p[2] + pose[7];
    const double vz = pose[8] * p[0] + pose[9] * p[1] + pose[10] * p[2] + pose[11];
    mask_label[i] = inside_box_f64(gt_planes + (int64_t)b * DAL3_PLANE_DOUBLES, vx, vy, vz) ? 1 : 0;
}

// dynamic: window frame j of item b is labelled in THAT frame's vehicle frame: q = xform[b][j] (pose p), with
// xform = inv(veh_to_global_j) @ inv(pose) from the host (dynamic_model.py:481); frames without a matched
// annotation (valid == 0) or outside the track give zeros. Empty frames' zero points are moved and tested too.
__global__ void dynamic_item_labels_kernel(const double* __restrict__ points, const int64_t* __restrict__ frame_offsets,
                                           const int64_t* __restrict__ track_first, const int32_t* __restrict__ item_track,
                                           const int32_t* __restrict__ item_frame, const int32_t* __restrict__ choice,
                                           const double* __restrict__ pose_all, int B, int n_per, int r, uint64_t seed,
                                           int64_t item_offset, const double* __restrict__ xform,
                                           const double* __restrict__ planes, const uint8_t* __restrict__ valid,
                                           uint8_t* __restrict__ mask_label) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t per_item = (int64_t)(2 * r + 1) * n_per;
    if (i >= (int64_t)B * per_item) return;
    const int b = (int)(i / per_item);
    const int64_t w = i - (int64_t)b * per_item;
    const int j = (int)(w / n_per);
    const int64_t seg = (int64_t)b * (2 * r + 1) + j;
    const int trk = item_track[b], it = item_frame[b];
    const int64_t f0 = track_first[trk];
    const int n_frames = (int)(track_first[trk + 1] - f0);
    const int fr = it - r + j;
    if (fr < 0 || fr >= n_frames || !valid[seg]) {
        mask_label[i] = 0;
        return;
    }
    double x = 0.0, y = 0.0, z = 0.0;
    const int64_t p0 = frame_offsets[f0 + fr], cnt = frame_offsets[f0 + fr + 1] - p0;
    if (cnt > 0) {
        int64_t k = choice ? choice[i] : (int64_t)(((uint64_t)prep_hash(seed, (uint64_t)(item_offset + b), (uint32_t)w) * (uint64_t)cnt) >> 32);
        k = k < 0 ? 0 : (k >= cnt ? cnt - 1 : k);
        const double* p = points + (p0 + k) * 3;
        x = p[0];
        y = p[1];
        z = p[2];
    }
    const double* pose = pose_all + (int64_t)b * 16;
    const double vx = pose[0] * x + pose[1] * y + pose[2] * z + pose[3];
    const double vy = pose[4] * x + pose[5] * y + pose[6] * z + pose[7];
    const double vz = pose[8] * x + pose[9] * y + pose[10] * z + pose[11];
    const double* t = xform + seg * 16;
    const double qx = t[0] * vx + t[1] * vy + t[2] * vz + t[3];
    const double qy = t[4] * vx + t[5] * vy + t[6] * vz + t[7];
    const double qz = t[8] * vx + t[9] * vy + t[10] * vz + t[11];
    mask_label[i] = inside_box_f64(planes + seg * DAL3_PLANE_DOUBLES, qx, qy, qz) ? 1 : 0;
}

hipError_t launch_static_crop_labels(const double* points, const int64_t* offsets, const int32_t* choice, const double* pose,
                                     int B, int N, uint64_t seed, int64_t item_offset, const double* gt_planes,
                                     uint8_t* mask_label, hipStream_t s) {
    const int64_t total = (int64_t)B * N;
    hipLaunchKernelGGL(static_crop_labels_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, points, offsets,
                       choice, pose, B, N, seed, item_offset, gt_planes, mask_label);
    return hipGetLastError();
}

hipError_t launch_dynamic_item_labels(const double* points, const int64_t* frame_offsets, const int64_t* track_first,
                                      const int32_t* item_track, const int32_t* item_frame, const int32_t* choice,
                                      const double* pose, int B, int n_per, int r, uint64_t seed, int64_t item_offset,
                                      const double* xform, const double* planes, const uint8_t* valid, uint8_t* mask_label,
                                      hipStream_t st) {
    const int64_t total = (int64_t)B * (2 * r + 1) * n_per;
    hipLaunchKernelGGL(dynamic_item_labels_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, points,
                       frame_offsets, track_first, item_track, item_frame, choice, pose, B, n_per, r, seed, item_offset, xform,
                       planes, valid, mask_label);
    return hipGetLastError();
}

hipError_t launch_static_crop_prep(const double* points, const int64_t* offsets, const int32_t* choice, const double* pose,
                                   const double* box, int B, int N, uint64_t seed, int64_t item_offset, float* pts_out,
                                   float* init_box_out, hipStream_t s) {
    const int64_t total = (int64_t)B * N;
    hipLaunchKernelGGL(static_crop_prep_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, points, offsets,
                       choice, pose, box, B, N, seed, item_offset, pts_out, init_box_out);
    return hipGetLastError();
}

hipError_t launch_dynamic_item_prep(const double* points, const int64_t* frame_offsets, const double* boxes,
                                    const int64_t* track_first, const int32_t* item_track, const int32_t* item_frame,
                                    const int32_t* choice, const double* pose, int B, int n_per, int r, int s_,
                                    uint64_t seed, int64_t item_offset, float* pts_out, float* box_out,
                                    float* init_box_out, hipStream_t st) {
    const int64_t total = (int64_t)B * (2 * r + 1) * n_per + (int64_t)B * (2 * s_ + 1);
    hipLaunchKernelGGL(dynamic_item_prep_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, points,
                       frame_offsets, boxes, track_first, item_track, item_frame, choice, pose, B, n_per, r, s_, seed,
                       item_offset, pts_out, box_out, init_box_out);
    return hipGetLastError();
}

// ================================================================================== write-back (N3)
// The step right after the heads (static_eval.py:71-87,148-155; dynamic_eval.py:53-64,121-129): carry each refined
// box into a frame of its track and overwrite the first detection of that frame whose centre lies within 0.1 m of
// the track's own box there. One thread per (track, frame) pair; float64 like the reference.
__device__ __forceinline__ void tbox(const double* __restrict__ m, const double* in, double* out) {   // transform_box
    out[0] = m[0] * in[0] + m[1] * in[1] + m[2] * in[2] + m[3];
    out[1] = m[4] * in[0] + m[5] * in[1] + m[6] * in[2] + m[7];
    out[2] = m[8] * in[0] + m[9] * in[1] + m[10] * in[2] + m[11];
    out[3] = in[3];
    out[4] = in[4];
    out[5] = in[5];
    out[6] = in[6] + atan2(m[4], m[0]);
}

__global__ void writeback_match_kernel(const double* __restrict__ track_box, const double* __restrict__ pose_inv,
                                       const float* __restrict__ det, const int64_t* __restrict__ det_start,
                                       const int32_t* __restrict__ det_count, const uint8_t* __restrict__ active, int P,
                                       int32_t* __restrict__ match, int32_t* __restrict__ owner) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P) return;
    int found = -1;
    if (active[p]) {
        double c[7];
        tbox(pose_inv + (int64_t)p * 16, track_box + (int64_t)p * 7, c);
        const float* d = det + det_start[p] * 7;
        for (int k = 0; k < det_count[p]; ++k) {
            const double dx = (double)d[k * 7 + 0] - c[0], dy = (double)d[k * 7 + 1] - c[1], dz = (double)d[k * 7 + 2] - c[2];
            if (sqrt(dx * dx + dy * dy + dz * dz) < 0.1) {
                found = k;
                break;
            }
        }
        if (found >= 0) atomicMax(owner + det_start[p] + found, p);      // a later pair wins, as in the sequential loop
    }
    match[p] = found;
}

__global__ void writeback_apply_kernel(const double* __restrict__ final_boxes, const int32_t* __restrict__ final_idx,
                                       const double* __restrict__ pose_best, const double* __restrict__ pose_inv,
                                       const int64_t* __restrict__ det_start, const int32_t* __restrict__ match,
                                       const int32_t* __restrict__ owner, int P, float* __restrict__ det) {
    const int p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= P || match[p] < 0) return;
    const int64_t row = det_start[p] + match[p];
    if (owner[row] != p) return;
    double a[7], b[7];
    const double* f = final_boxes + (int64_t)final_idx[p] * 7;
    if (pose_best) {                                       // static: best-frame vehicle -> global -> this frame's vehicle
        tbox(pose_best + (int64_t)p * 16, f, a);
        tbox(pose_inv + (int64_t)p * 16, a, b);
    } else {
#pragma unroll
        for (int k = 0; k < 7; ++k) b[k] = f[k];
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) det[row * 7 + k] = (float)b[k];
}

hipError_t launch_writeback(const double* final_boxes, const int32_t* final_idx, const double* pose_best,
                            const double* pose_inv, const double* track_box, float* det, const int64_t* det_start,
                            const int32_t* det_count, const uint8_t* active, int P, int64_t n_det, int32_t* match,
                            int32_t* owner, hipStream_t s) {
    hipError_t e = launch_fill_words(owner, (size_t)n_det, 0xFFFFFFFFu, s);             // -1
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(writeback_match_kernel, dim3((P + 127) / 128), dim3(128), 0, s, track_box, pose_inv, det, det_start,
                       det_count, active, P, match, owner);
    hipLaunchKernelGGL(writeback_apply_kernel, dim3((P + 127) / 128), dim3(128), 0, s, final_boxes, final_idx, pose_best,
                       pose_inv, det_start, match, owner, P, det);
    return hipGetLastError();
}
