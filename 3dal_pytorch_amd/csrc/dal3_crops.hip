// dal3_crops.hip — crop extraction from full sweeps: the per-detection loop of
// _create_pd_detection (det3d/datasets/waymo/waymo_common.py:166-171) — for every tracked detection of a frame, the
// sweep's points inside its rotated box, moved to the global frame with the frame's veh_to_global — for many frames
// in one pass. HBM-bound integer/compare work: every sweep point is read once per pass (12 B), tested against the
// candidate boxes of its own grid cell (see CROP_GRID below), and only the members are written.
//
// Ordered output without a sort: a wavefront owns a chunk of CROP_CHUNK consecutive points of one frame.
//   pass 1 (count):  cc[box][chunk] = members of the chunk                       (ballot + popcount)
//   scan:            per box, exclusive prefix over its frame's chunks; counts[box] = total
//   (caller: exclusive prefix of counts over boxes -> box_start, and the grand total to size the output)
//   pass 2 (fill):   recompute membership; member rank inside the chunk = running popcount + lanes-below mask
// so each detection's points come out in sweep order, exactly as `lidars[indices]` gives them.
#include "dal3_geom.h"
#include "dal3_kernels.h"

#ifndef CROP_CHUNK
#define CROP_CHUNK 1024                  // points per wavefront: 16 rounds of 64
#endif
#define CROP_ROUNDS (CROP_CHUNK / 64)
#ifndef CROP_WAVES
#define CROP_WAVES 8                     // chunks (waves) per workgroup: they share the frame's grid
#endif
// The cull in front of the exact test (round 5; rounds 2-4 tested every 64-point round against every detection's
// ball: 10 VALU per point and box, 1.3e12 point-box tests/s, and three times slower on a shuffled sweep than on a
// range-image-ordered one, where most rounds miss most balls). Now: a workgroup (four chunks of ONE frame) rasterises
// the frame's detections — 64 at a time, one bit each — into a CROP_GRID x CROP_GRID grid of 64-bit masks over the
// vehicle frame's x/y plane in LDS (cell CROP_CELL m; the detection's ball, as handed in `spheres`, projected and
// rounded outwards; coordinates beyond +-CROP_GRID*CROP_CELL/2 clamp to the border cells, which is monotonic, so a
// point inside a ball always lands in a marked cell). A point then looks up ITS cell: the set bits are its candidate
// detections (none for most points, one or two near an object), and only those get the exact face test, per lane,
// with the face equations read from LDS. Whatever the order of the sweep, a point costs one lookup plus its own
// candidates. Non-finite coordinates are candidates of every detection (the reference's NaN behaviour: dal3_geom.h),
// non-finite balls cover the whole grid.
#ifndef CROP_GRID
#define CROP_GRID 60                     // 60 x 2.5 m = +-75 m (the lidar's range); 64 put the fill pass's LDS 2 KiB over a quarter of a CU's
#endif
#define CROP_CELL 2.5f
#define CROP_BATCH 64                    // detections per grid pass (one mask bit each)

__device__ __forceinline__ int lanes_below(uint64_t m) {
    return __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
}
// monotonic non-decreasing in v (float add, multiply by a positive constant, clamp, truncation of a non-negative
// value); NaN -> 0 (callers never rely on it: non-finite points bypass the grid, non-finite balls span it)
__device__ __forceinline__ int crop_cell(float v) {
    float t = (v + 0.5f * CROP_GRID * CROP_CELL) * (1.0f / CROP_CELL);
    t = fminf(fmaxf(t, 0.0f), (float)(CROP_GRID - 1));
    return (int)t;
}
__device__ __forceinline__ bool finite_f32(float v) { return (__float_as_uint(v) & 0x7f800000u) != 0x7f800000u; }

// inside_box_f32 (dal3_geom.h) with the six face equations already converted to float (the same conversion, done once
// per workgroup when they are staged in LDS)
__device__ __forceinline__ bool inside_box_lds(const float* pl, float x, float y, float z) {
    bool in = true;
#pragma unroll
    for (int f = 0; f < 6; ++f) {
        const float sgn = __fadd_rn(__fadd_rn(__fadd_rn(__fmul_rn(x, pl[f * 4 + 0]), __fmul_rn(y, pl[f * 4 + 1])),
                                              __fmul_rn(z, pl[f * 4 + 2])), pl[f * 4 + 3]);
        in = in && !ge_zero(sgn);
    }
    return in;
}

template <bool FILL>
__global__ __launch_bounds__(64 * CROP_WAVES) void crop_pass_kernel(
    const float* __restrict__ points, const int64_t* __restrict__ point_offsets, const double* __restrict__ planes,
    const float* __restrict__ spheres, const int64_t* __restrict__ box_offsets, int max_chunks, int32_t* __restrict__ cc,
    const int64_t* __restrict__ counts, const double* __restrict__ pose_all, const int64_t* __restrict__ box_start,
    double* __restrict__ out_points, int32_t* __restrict__ out_index, int64_t out_capacity) {
    __shared__ unsigned long long s_grid[CROP_GRID * CROP_GRID];         // 28.8 KiB
    __shared__ float s_pl[CROP_BATCH * DAL3_PLANE_DOUBLES];               // 6 KiB
    __shared__ int s_cnt[FILL ? 1 : CROP_WAVES][FILL ? 1 : CROP_BATCH];   // count pass: a wave's members per detection
    __shared__ unsigned long long s_row[FILL ? CROP_WAVES : 1][FILL ? CROP_BATCH : 1];   // fill pass: a detection's next output row, per wave
    __shared__ uint32_t s_seen[FILL ? CROP_WAVES : 1][2];                 // fill pass: detections met in the current round
    const int frame = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int chunk = blockIdx.x * CROP_WAVES + (threadIdx.x >> 6);
    const int64_t p0 = point_offsets[frame], n_pts = point_offsets[frame + 1] - p0;
    if ((int64_t)blockIdx.x * CROP_WAVES * CROP_CHUNK >= n_pts) return;  // the whole workgroup is past the sweep's end
    const bool active = (int64_t)chunk * CROP_CHUNK < n_pts;             // (a wave past it still joins the barriers)
    const bool last_chunk = (int64_t)(chunk + 1) * CROP_CHUNK >= n_pts;
    const int64_t k0 = box_offsets[frame], k1 = box_offsets[frame + 1];
    // The chunk's points are NOT held in registers across the rounds (round 5, first version: 48 registers of coordinates,
    // a fully unrolled 16-round body, 227 VGPRs in the fill pass = two waves per SIMD, and counters that said the kernel
    // WAITS: SQ_WAIT_ANY 0.53 of the wave cycles, the vector ALUs active in 0.10 of the SIMD cycles — profiles/
    // r05_pmc_crops.txt). A rolled loop loads round r + 1 while round r is worked on; at ~48 registers eight waves share a
    // SIMD and hide each other's LDS and memory latency.
    const float* pbase = points + p0 * 3;
    const int64_t i0 = (int64_t)chunk * CROP_CHUNK + lane;
    auto load_round = [&](int r, float& px, float& py, float& pz) -> bool {
        const int64_t i = i0 + r * 64;
        const bool ok = active && i < n_pts;
        const float* p = pbase + (ok ? i : 0) * 3;
        px = p[0];
        py = p[1];
        pz = p[2];
        return ok;
    };
    const double* pose = FILL ? pose_all + (int64_t)frame * 16 : nullptr;
    for (int64_t kb = k0; kb < k1; kb += CROP_BATCH) {
        const int nb = (int)(k1 - kb < CROP_BATCH ? k1 - kb : CROP_BATCH);
        if (kb > k0) __syncthreads();                                    // everyone is done with the previous batch's grid
        for (int i = threadIdx.x; i < CROP_GRID * CROP_GRID; i += 64 * CROP_WAVES) s_grid[i] = 0ull;
        for (int i = threadIdx.x; i < nb * DAL3_PLANE_DOUBLES; i += 64 * CROP_WAVES)
            s_pl[i] = (float)planes[kb * DAL3_PLANE_DOUBLES + i];
        __syncthreads();
        if ((int)threadIdx.x < nb) {                                     // one thread rasterises one detection's ball
            const float* sp = spheres + (kb + threadIdx.x) * 4;
            // radius rounded UP (sqrt's last bit, then a relative 1e-6 on top: the ball itself carries a margin of
            // 1 mm + 1e-3 r, geom.cull_spheres); r2 = +inf (no cull) -> the whole grid
            const float rad = sqrtf(sp[3]) * 1.000001f;
            const int cx0 = crop_cell(sp[0] - rad), cx1 = crop_cell(sp[0] + rad);
            const int cy0 = crop_cell(sp[1] - rad), cy1 = crop_cell(sp[1] + rad);
            const bool all = !finite_f32(rad) || !finite_f32(sp[0]) || !finite_f32(sp[1]);
            const unsigned long long bit = 1ull << threadIdx.x;
            for (int cy = all ? 0 : cy0; cy <= (all ? CROP_GRID - 1 : cy1); ++cy)
                for (int cx = all ? 0 : cx0; cx <= (all ? CROP_GRID - 1 : cx1); ++cx)
                    atomicOr(&s_grid[cy * CROP_GRID + cx], bit);
        }
        __syncthreads();
        if (!active) continue;
        // the running state per detection is wave-private LDS (a wave's LDS operations execute in order): its member
        // count in this chunk (count pass, s_cnt) / the output row of its next member (fill pass, s_row)
        const int w = threadIdx.x >> 6;
        if (FILL) {
            bool any = false;
            if (lane < nb) {
                const int64_t k = kb + lane;
                const int32_t excl = cc[k * max_chunks + chunk];
                const int64_t next = last_chunk ? counts[k] : (int64_t)cc[k * max_chunks + chunk + 1];
                any = next != excl;
                if constexpr (FILL) s_row[w][lane] = (unsigned long long)(box_start[k] + excl);
            }
            if (__ballot(any) == 0) continue;                            // nothing of this batch in this chunk
        }
        const unsigned long long all_bits = nb == 64 ? ~0ull : ((1ull << nb) - 1ull);
        if constexpr (!FILL) s_cnt[w][lane] = 0;
        float nx, ny, nz;
        bool nok = load_round(0, nx, ny, nz);
#ifndef CROP_UNROLL
#define CROP_UNROLL 4
#endif
#pragma unroll CROP_UNROLL
        for (int r = 0; r < CROP_ROUNDS; ++r) {
            const float px = nx, py = ny, pz = nz;
            const bool ok = nok;
            if (r + 1 < CROP_ROUNDS) nok = load_round(r + 1, nx, ny, nz);
            unsigned long long cand = 0ull;
            if (ok)
                cand = (finite_f32(px) && finite_f32(py) && finite_f32(pz)) ? s_grid[crop_cell(py) * CROP_GRID + crop_cell(px)]
                                                                           : all_bits;
            unsigned long long in = 0ull;
            while (cand) {                                               // per lane: its own candidates only
                const int j = __builtin_ctzll(cand);
                cand &= cand - 1ull;
                if (inside_box_lds(s_pl + j * DAL3_PLANE_DOUBLES, px, py, pz)) in |= 1ull << j;
            }
            if (!FILL) {
                // counting needs no order: every member adds one to its detection's counter, per lane (with the ordered
                // loop below a shuffled sweep — half a dozen detections with a member in every round — paid an iteration
                // per detection and round in BOTH passes)
                if constexpr (!FILL) {
                    while (in) {
                        atomicAdd(&s_cnt[w][__builtin_ctzll(in)], 1);
                        in &= in - 1ull;
                    }
                }
                continue;
            }
            if constexpr (FILL) {
            if (__ballot(in != 0ull) == 0) continue;
            auto emit = [&](int64_t o) {                                 // this lane's point -> output row o
                if (o < out_capacity) {                  // (a caller that sized `out` from an estimate: rows past it are dropped)
                    const double dx = px, dy = py, dz = pz;              // concatenate([lidars_o, ones]) is float64
                    out_points[o * 3 + 0] = pose[0] * dx + pose[1] * dy + pose[2] * dz + pose[3];
                    out_points[o * 3 + 1] = pose[4] * dx + pose[5] * dy + pose[6] * dz + pose[7];
                    out_points[o * 3 + 2] = pose[8] * dx + pose[9] * dy + pose[10] * dz + pose[11];
                    if (out_index) out_index[o] = (int32_t)((int64_t)chunk * CROP_CHUNK + r * 64 + lane);
                }
            };
            // Does any detection have TWO members in this round? (A shuffled sweep: half a dozen detections with one member
            // each in every round, and the ordered loop below paid an iteration of two ballots and four readlanes for each;
            // a range-image sweep: runs of one detection's points.) Every member sets its detection's bit in a per-wave
            // word and looks at what was there.
            // (first the cheap case of a run: every member lane holds the SAME set of detections — then the ordered loop
            // below is one iteration per detection of that set, and the LDS round trip of the check is not worth taking)
            const uint64_t has0 = __ballot(in != 0ull);
            const int src0 = __builtin_ctzll(has0);
            const unsigned long long v0 = ((unsigned long long)__builtin_amdgcn_readlane((uint32_t)(in >> 32), src0) << 32) |
                                          (unsigned long long)__builtin_amdgcn_readlane((uint32_t)in, src0);
            bool dup = true;
            if (__ballot(in != 0ull && in != v0) != 0) {
                if (lane < 2) s_seen[w][lane] = 0u;
                dup = false;
                for (unsigned long long t = in; t; t &= t - 1ull) {
                    const int j = __builtin_ctzll(t);
                    dup |= (atomicOr(&s_seen[w][j >> 5], 1u << (j & 31)) >> (j & 31)) & 1u;
                }
            }
            if (__ballot(dup) == 0) {
                // no: the one member of a detection takes the detection's running row and bumps it — no order to keep
                // inside the round, and the rounds follow each other in program order
                for (unsigned long long t = in; t; t &= t - 1ull) emit((int64_t)atomicAdd(&s_row[w][__builtin_ctzll(t)], 1ull));
                continue;
            }
            // yes: ordered compaction, detection by detection, over the detections that have a member in this round
            for (;;) {
                const uint64_t has = __ballot(in != 0ull);
                if (has == 0) break;
                const int src = __builtin_ctzll(has);
                const uint32_t lo = __builtin_amdgcn_readlane((uint32_t)in, src);
                const uint32_t hi = __builtin_amdgcn_readlane((uint32_t)(in >> 32), src);
                const int j = lo ? __builtin_ctz(lo) : 32 + __builtin_ctz(hi);      // wave-uniform
                const bool mine = (in >> j) & 1ull;
                const uint64_t m = __ballot(mine);
                const int64_t base = (int64_t)s_row[w][j];               // (one address: a broadcast read)
                if (mine) emit(base + lanes_below(m));
                if (lane == src) s_row[w][j] = (unsigned long long)(base + __popcll(m));
                in &= ~(1ull << j);
            }
            }   // FILL
        }
        if constexpr (!FILL) {
            if (lane < nb) cc[(kb + lane) * max_chunks + chunk] = s_cnt[w][lane];
        }
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
                           (const double*)nullptr, (const int64_t*)nullptr, (double*)nullptr, (int32_t*)nullptr, (int64_t)0);
    if (K_total > 0)
        hipLaunchKernelGGL(crop_scan_kernel, dim3((unsigned)((K_total + 3) / 4)), dim3(256), 0, s, point_offsets, box_offsets,
                           F, max_chunks > 0 ? max_chunks : 1, cc, counts);
    return hipGetLastError();
}

hipError_t launch_crop_fill(const float* points, const int64_t* point_offsets, const double* planes,
                            const float* spheres, const int64_t* box_offsets, int F, int64_t K_total,
                            int64_t max_points_per_frame, const double* pose, const int64_t* counts,
                            const int64_t* box_start, const int32_t* cc, double* out_points, int32_t* out_index,
                            int64_t out_capacity, hipStream_t s) {
    const int max_chunks = (int)((max_points_per_frame + CROP_CHUNK - 1) / CROP_CHUNK);
    if (F > 0 && max_chunks > 0 && K_total > 0)
        hipLaunchKernelGGL(crop_pass_kernel<true>, dim3((max_chunks + CROP_WAVES - 1) / CROP_WAVES, F), dim3(64 * CROP_WAVES),
                           0, s, points, point_offsets, planes, spheres, box_offsets, max_chunks, const_cast<int32_t*>(cc),
                           counts, pose, box_start, out_points, out_index, out_capacity);
    return hipGetLastError();
}


// box_start / out_offsets from counts, on the device (round 5: the chained run has no host round trip between count and
// fill). The detections are laid out in the caller's ORDER (order[i] = the detection at output position i; NULL = as
// numbered): out_offsets[i] = rows in front of position i (K+1 entries, the last = the total), box_start[order[i]] =
// out_offsets[i], box_start[K] = the total. cap >= 0 (dal3_crop_starts_capped): out_offsets — what CONSUMERS of the filled
// buffer index it with — never point past the buffer's `cap` rows (the fill drops those rows); box_start keeps the true
// prefix sums (the fill needs them, and box_start[K] is how the caller learns of the overflow). One workgroup: K is a segment's detections (~1e4), two trips.
#define STARTS_PER 8                     // positions per thread and trip: a segment's ~1e4 detections are two trips of the workgroup
__global__ __launch_bounds__(1024) void crop_starts_kernel(const int64_t* __restrict__ counts, const int64_t* __restrict__ order,
                                                           int64_t K, int64_t* __restrict__ box_start,
                                                           int64_t* __restrict__ out_offsets, int64_t cap) {
    __shared__ int64_t s_wave[16];
    __shared__ int64_t s_run;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_run = 0;
    __syncthreads();
    for (int64_t i0 = 0; i0 < K; i0 += 1024 * STARTS_PER) {
        const int64_t i = i0 + (int64_t)threadIdx.x * STARTS_PER;        // this thread's STARTS_PER consecutive positions
        int64_t k[STARTS_PER], v[STARTS_PER], mine = 0;
#pragma unroll
        for (int q = 0; q < STARTS_PER; ++q) {
            k[q] = i + q < K ? (order ? order[i + q] : i + q) : 0;
            v[q] = i + q < K ? counts[k[q]] : 0;
            mine += v[q];
        }
        int64_t incl = mine;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int64_t up = __shfl_up(incl, d);
            if (lane >= d) incl += up;
        }
        if (lane == 63) s_wave[wave] = incl;
        __syncthreads();
        int64_t before = s_run;
        for (int w = 0; w < wave; ++w) before += s_wave[w];
        int64_t excl = before + incl - mine;
#pragma unroll
        for (int q = 0; q < STARTS_PER; ++q) {
            if (i + q < K) {
                box_start[k[q]] = excl;
                if (out_offsets) out_offsets[i + q] = cap >= 0 && excl > cap ? cap : excl;
            }
            excl += v[q];
        }
        __syncthreads();
        if (threadIdx.x == 1023) s_run = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        box_start[K] = s_run;
        if (out_offsets) out_offsets[K] = cap >= 0 && s_run > cap ? cap : s_run;
    }
}

hipError_t launch_crop_starts(const int64_t* counts, const int64_t* order, int64_t K, int64_t* box_start, int64_t* out_offsets,
                              int64_t out_capacity, hipStream_t s) {
    hipLaunchKernelGGL(crop_starts_kernel, dim3(1), dim3(1024), 0, s, counts, order, K, box_start, out_offsets, out_capacity);
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
