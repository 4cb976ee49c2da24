/* dal3.h — C ABI of lib3dal_hip.so: the MI355X (gfx950) implementation of the 3DAL
 * Frustum-PointNet auto-labeling heads (eval-mode forward + box decode).
 *
 * Boundary contract:
 *   - plain C, no torch types, no C++ mangling; every pointer is a raw DEVICE pointer unless a
 *     parameter says "host"; the caller (the PyTorch caching allocator, via tensor.data_ptr())
 *     owns every buffer including the workspace and the packed weights;
 *   - every call is asynchronous on the given HIP stream; the library never synchronises the
 *     device, allocates nothing persistent and keeps no mutable global state except a
 *     thread-local error string;
 *   - return 0 on success, a negative DAL3_E* code on failure; dal3_last_error() describes it.
 *
 * Each entry point names the reference interface it replaces (paths relative to the
 * jacky121298/3DAL_PyTorch checkout).
 */
#ifndef DAL3_H
#define DAL3_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* dal3_stream;               /* hipStream_t */

#define DAL3_VERSION 150                 /* 0.1.5: dal3_crop_starts_capped; upper bounds on B, N (DAL3_MAX_*); DAL3_BCN_NO_LDS_SAMPLER; 0.1.4: dal3_crop_starts, dal3_crop_fill takes out_capacity; 0.1.3: dal3_bcn.flags (DAL3_BCN_*); dal3_tr_linear_bn_stats / dal3_tr_linear_bnbwd_sums; .1: dal3_tr_fc_*, dal3_tr_wgrad_final_many, dal3_parse_box_pred* */

enum {
    DAL3_OK = 0,
    DAL3_EINVAL = -1,                    /* bad shape / stride / alignment / null pointer */
    DAL3_EWORKSPACE = -2,                /* workspace too small */
    DAL3_EHIP = -3                       /* a HIP runtime call failed (message has hipGetErrorString) */
};

/* Upper bounds of ONE call's batch, checked by every entry that takes (B, N) or (B, M) before anything is carved or
 * launched: a call beyond them returns DAL3_EINVAL ("split the batch"). The kernels index an item's points and the
 * per-item outputs with 32-bit ints and launch one workgroup (or worklist entry) per 32-point tile of an item, so a
 * larger job would get a truncated grid or a wrapped offset instead of an error. None of them is reachable with buffers
 * that fit the 288 GB of an MI355X except through a wrong argument: 2^31 tiles of fp32 xyz points alone are 824 GB. */
#define DAL3_MAX_ITEMS 16777216            /* B  <= 2^24 items (crops / track-frames) per call */
#define DAL3_MAX_POINTS_PER_ITEM 16777216  /* N, M, n_box <= 2^24 points per item */
#define DAL3_MAX_TILES 2147483647          /* B * ceil(N / 32) <= 2^31 - 1 (the launch grid) */

/* arithmetic dtype of the shared-MLP kernels of a packed head. DAL3_F32: exact-fp32 MFMA (the reference's
 * precision). DAL3_BF16 / DAL3_F16: weights and inter-layer activations rounded to 16 bits, fp32 accumulate,
 * v_mfma_f32_32x32x16_{bf16,f16} (BASELINE.json configs C3 / C5); dconv5 (128 -> 2) runs as one more 16-bit
 * out-tile of the decode stack; first layer, the per-crop term of dconv1, FC heads, mask and all I/O stay fp32.
 * tests/emu16.py is the reference model of this arithmetic: the kernels' error against the fp32 oracle equals the
 * model's to a few per cent (rms), and the tests hold them to 1.5 x (max) / 1.15 x (rms) of it. The same dtype must be given to dal3_pack_weights and to the forward calls. */
enum { DAL3_F32 = 0, DAL3_BF16 = 1, DAL3_F16 = 2,
       /* fp16 MFMAs on (hi, lo) SPLIT operands: x = x_hi + x_lo, w = w_hi + w_lo in fp16, w x ~ w_hi x_hi + w_hi x_lo + w_lo x_hi,
        * fp32 accumulate — fp32 ACCURACY (logits ~1e-6 of their range, like DAL3_F32) from three fp16 MFMAs per fp32 one.
        * An arithmetic dtype of packed weights only (never a storage dtype of dal3_bcn / dal3_maxpool_n_dtype).
        * Range: folded weights and every layer's activations must stay below fp16's largest finite value (65504) in
        * magnitude. A folded WEIGHT beyond it is packed as NaN — and a binding should refuse it outright at packing
        * time (3dal_pytorch_amd/_hip.py check_f16x3_range raises: the shared-MLP kernels are built without NaN semantics,
        * so a NaN weight is not guaranteed to surface). An ACTIVATION beyond it saturates in the split and the crop's
        * outputs are WRONG WITHOUT NOTICE (DAL3_F32 has no such limit; DAL3_F16 turns NaN there). The margin is three orders of magnitude on this path: the first layer, which sees
        * the raw coordinates, runs in fp32, and crops scaled 1,000 x (box-frame coordinates of kilometres) still match
        * DAL3_F32 to 1e-6 (tests/test_gpu_x3.py). Values below fp16's normal range lose nothing that fp32 accumulation
        * would keep. */
       DAL3_F16X3 = 3 };

/* which sub-network a packed-weight blob belongs to */
enum {
    DAL3_HEAD_INS_SEG = 0,               /* PointNetInstanceSeg: 10 layers, c_in 3 (static) or 4 (dynamic) */
    DAL3_HEAD_STATIC_BOX_EST = 1,        /* static PointNetEstimation: 4 conv + 3 fc */
    DAL3_HEAD_POINT_EMB = 2,             /* PointEmbedding: 4 conv + 2 fc */
    DAL3_HEAD_BOX_EMB = 3,               /* BoxEmbedding: 4 conv + 2 fc */
    DAL3_HEAD_DYNAMIC_BOX_EST = 4        /* dynamic PointNetEstimation: 3 fc */
};

/* how the M object points of a crop are drawn from its segmented points
 * (replaces gather_object_pts, tools/static_model.py:23-49) */
enum {
    DAL3_SAMPLER_DEVICE = 0,             /* counter-based RNG keyed on (seed, global item index, point) */
    DAL3_SAMPLER_CHOICE = 1              /* caller supplies `choice` (B,M): positions into the ordered
                                            list of segmented points, e.g. NumPy's legacy stream drawn
                                            in the reference's order for bit-reproducible eval runs */
};

/* One Conv1d(k=1)/Linear layer with its optional eval-mode BatchNorm1d, as the reference's
 * state_dict holds it: weight (c_out, c_in[,1]) row-major, the rest (c_out). bn_* all NULL => no BN. */
typedef struct {
    const float* weight;
    const float* bias;
    const float* bn_weight;
    const float* bn_bias;
    const float* bn_mean;
    const float* bn_var;
    int32_t c_in;
    int32_t c_out;
} dal3_layer;

/* ------------------------------------------------------------------------------------------ */
int dal3_version(void);
const char* dal3_last_error(void);       /* thread-local, static storage */
/* MEAN_SIZE_ARR (tools/static_model.py:17-21) as the library holds it: 9 floats, (3 size clusters) x (l, w, h), HOST
 * memory with static storage. The one copy that the decode and criterion kernels are compiled with; a binding that
 * keeps the table on its side as well (3dal_pytorch_amd/arch.py) compares the two when it loads the library. */
const float* dal3_mean_size(void);

/* Fold BN (eps 1e-5) into each layer and write the MFMA-fragment-ordered / row-major image the
 * kernels consume. Call with packed_dev == NULL to query *bytes_inout. Layers come in forward
 * order (ins_seg: conv1..5, dconv1..5; *_BOX_EST / *_EMB: conv1..4 then fc1..).
 * Replaces: nothing in the reference (it keeps nn.Conv1d/nn.BatchNorm1d modules,
 * static_model.py:249-269); this is the derived cache of SURVEY.md 8(b). */
int dal3_pack_weights(int head_kind, const dal3_layer* layers, int n_layers, int dtype,
                      void* packed_dev, size_t* bytes_inout, dal3_stream stream);

/* logical (B, C, N) tensor with element strides: the callers hand pts.transpose(2,1) of a
 * point-major buffer, i.e. strides (N*C, 1, C) (static_eval.py:265); contiguous (C*N, N, 1)
 * is accepted too. dtype = how the values are STORED: DAL3_F32 (0, what a zero-initialised struct says), DAL3_BF16 or
 * DAL3_F16 — 16-bit points / box windows (BASELINE.json configs C3, C5: "bf16 storage") are read in place and widened
 * exactly in the kernels' loads, no fp32 copy is made; `data` then points at 2-byte elements and the strides count
 * those. Independent of the arithmetic dtype of the packed weights. Outputs are fp32 whatever the storage. */
typedef struct {
    const float* data;
    int64_t stride_b, stride_c, stride_n;
    int32_t dtype;
    int32_t flags;                       /* MUST be 0 or a mask of DAL3_BCN_* (was `reserved` before 0.1.3: a caller that never
                                          * zeroed it now gets DAL3_EINVAL for unknown bits, or a different — bit-identical
                                          * — kernel family for bits 1 / 2 / 4; zero-initialise the struct) */
} dal3_bcn;

/* Per-call dispatch hints (dal3_bcn.flags; dal3_static_args / dal3_dynamic_args take them from args.pts.flags for every
 * kernel of the call). Which kernel family runs is a function of the job's size and of these bits ONLY — the library
 * reads no environment variable and keeps no process-wide switch. Results are bit-identical either way; the bits exist
 * for A/B measurements and for the tests that pin that identity (tests/test_gpu_latency.py, test_gpu_parity.py).
 *   DAL3_BCN_NO_SMALL_JOB_KERNELS  never the small-job ("latency") family (jobs of <= 512 tiles of 32 points)
 *   DAL3_BCN_NO_WORKLIST           point heads: one workgroup per (item, tile) instead of the live-tile worklist
 *   DAL3_BCN_NO_LDS_SAMPLER        mask compaction + sampling: positions through global memory, keys recomputed per
 *                                  pass (what items of more than 7936 points always take) instead of both in LDS */
enum { DAL3_BCN_NO_SMALL_JOB_KERNELS = 1, DAL3_BCN_NO_WORKLIST = 2, DAL3_BCN_NO_LDS_SAMPLER = 4 };

/* ---- PointNetInstanceSeg.forward (static_model.py:271-296, dynamic_model.py:187-212) plus
 * the mask of point_cloud_masking (static_model.py:59). logits (B,N,2) fp32, mask (B,N) u8.
 * workspace: dal3_ins_seg_workspace_bytes(B). global_feat_out optional (B,1024). */
size_t dal3_ins_seg_workspace_bytes(int B);
int dal3_ins_seg_forward(const void* packed, int dtype, int c_in, dal3_bcn pts, int B, int N,
                         float* logits, uint8_t* mask, float* global_feat_out,
                         void* workspace, size_t workspace_bytes, dal3_stream stream);

/* The three kernels of dal3_ins_seg_forward as separate launches (per-kernel timing and tests):
 *   encode      conv1..conv5 + BN + ReLU with the max over N fused (static_model.py:279-284):
 *               pts -> global_feat (B,1024); global_feat must be zero-filled by the caller
 *   global_bias the per-crop part of dconv1 (its 1024 global-feature columns, :286-289):
 *               global_feat -> gbias (B,512) = W1g' g + b1'
 *   decode      dconv1 (64 per-point columns) .. dconv5 + mask (:289-295, :59) */
int dal3_ins_seg_encode(const void* packed, int dtype, int c_in, dal3_bcn pts, int B, int N, float* global_feat,
                        dal3_stream stream);
int dal3_ins_seg_global_bias(const void* packed, int dtype, const float* global_feat, int B, float* gbias,
                             dal3_stream stream);
int dal3_ins_seg_decode(const void* packed, int dtype, int c_in, dal3_bcn pts, int B, int N, const float* gbias,
                        float* logits, uint8_t* mask, dal3_stream stream);

/* ---- gather_object_pts (static_model.py:23-49 / dynamic_model.py:24-50) on the device.
 * counts (B) i32 = number of segmented points; obj_idx (B,M) i32 = chosen point indices;
 * obj_pts (B,M,C) fp32 point-major = pts[:, :C, idx] (all-zero rows where count == 0, as the
 * reference leaves them). choice: see DAL3_SAMPLER_CHOICE (NULL for DAL3_SAMPLER_DEVICE).
 * item_offset = global index of item 0 (multi-GPU shards draw the same subset as one GPU).
 * workspace: dal3_gather_workspace_bytes(B, N). */
size_t dal3_gather_workspace_bytes(int B, int N);
int dal3_segment_counts(const uint8_t* mask, int B, int N, int32_t* counts, dal3_stream stream);
int dal3_mask_compact_sample(const uint8_t* mask, dal3_bcn pts, int B, int N, int C, int M,
                             int sampler, const int32_t* choice, uint64_t seed, int64_t item_offset,
                             int32_t* counts, int32_t* obj_idx, float* obj_pts,
                             void* workspace, size_t workspace_bytes, dal3_stream stream);

/* The device sampler with its draw counter in DEVICE memory: key = (seed, *step, global item index, rank). Training
 * draws fresh object points every step (the reference's np.random stream moves on, static_model.py:36-47); a step
 * captured into a hipGraph must not freeze that, so the counter is a device scalar the caller bumps with a captured op
 * (step == NULL: as dal3_mask_compact_sample with DAL3_SAMPLER_DEVICE). */
int dal3_mask_compact_sample_step(const uint8_t* mask, dal3_bcn pts, int B, int N, int C, int M, uint64_t seed,
                                  const int64_t* step, int64_t item_offset, int32_t* counts, int32_t* obj_idx,
                                  float* obj_pts, void* workspace, size_t workspace_bytes, dal3_stream stream);

/* ---- shared MLP (4 x Conv1d k=1 + BN + ReLU) + channel-wise max over the point axis + the FC
 * stack of the head: static PointNetEstimation.forward (static_model.py:320-339) -> (B,39);
 * PointEmbedding.forward (dynamic_model.py:234-249) -> (B,256); BoxEmbedding.forward
 * (:271-286) -> (B,128). out has row stride out_stride floats (so the two embeddings can be
 * written side by side: the torch.cat of dynamic_model.py:137).
 * workspace: dal3_point_head_workspace_bytes(B). */
size_t dal3_point_head_workspace_bytes(int B);
int dal3_point_head_forward(int head_kind, const void* packed, int dtype, dal3_bcn x, int B, int M,
                            float* out, int64_t out_stride,
                            void* workspace, size_t workspace_bytes, dal3_stream stream);

/* The per-point stack + max of the same heads as its own launch (conv1..4 + BN + ReLU, torch.max(x, 2)[0]:
 * static_model.py:329-334, dynamic_model.py:240-245, :277-282) -> feat (B,512), no FC tail; per-kernel timing and
 * tests. n_distinct (B) i32 device or NULL: only the first n_distinct[b] points of item b differ, the others repeat
 * one of them (what dal3_mask_compact_sample's device sampler writes when fewer than M points are segmented);
 * repeated points cannot change a max over points, so the kernel skips them. The whole-model sequencers pass
 * `counts` here. */
/* workspace (optional): dal3_point_head_pool_workspace_bytes(B, M) bytes of 16-byte aligned device scratch for the
 * list of tiles that hold distinct points; with it large jobs run as persistent waves over that list instead of one
 * workgroup per (item, tile) — same bits either way. NULL / too small: the per-tile launch. */
size_t dal3_point_head_pool_workspace_bytes(int B, int M);
int dal3_point_head_pool(int head_kind, const void* packed, int dtype, dal3_bcn x, int B, int M,
                         const int32_t* n_distinct, float* feat, void* workspace, size_t workspace_bytes,
                         dal3_stream stream);

/* ---- dynamic PointNetEstimation.forward (dynamic_model.py:300-312): (B,384) -> (B,39). */
int dal3_dynamic_box_est_forward(const void* packed, const float* embedding, int B, float* box_pred,
                                 void* workspace, size_t workspace_bytes, dal3_stream stream);

/* ---- parse_output_to_tensors (static_model.py:64-96) + the eval drivers' box decode
 * (static_eval.py:269-288, dynamic_eval.py:226-242, utils.py:69-79).
 * box_pred (B,39) -> heading_residuals (B,12), size_residuals (B,9), center (B,3), boxes7 (B,7)
 * [cx,cy,cz,l,w,h,yaw]. center = box_pred[:, :3] + center_add[:, :3] (center_add may be NULL),
 * written back into box_pred[:, :3] too when center_inplace != 0 (the reference's
 * `center_one += init_box[:, :3]` mutates box_pred through a view, static_model.py:174).
 * boxes7 centre = center (+ boxes_center_add[:, :3] when not NULL: the dynamic driver adds
 * init_box[:, :3] only at decode time). yaw = class2angle(argmax) + yaw_base[b*yaw_stride]. */
int dal3_decode_boxes(float* box_pred, int B,
                      const float* center_add, int64_t center_add_stride, int center_inplace,
                      const float* boxes_center_add, int64_t boxes_center_add_stride,
                      const float* yaw_base, int64_t yaw_stride,
                      float* heading_residuals, float* size_residuals, float* center, float* boxes7,
                      dal3_stream stream);

/* ---- the re-centring between the two box estimators of StaticModelTwoBoxEst
 * (static_model.py:192-205, rotz :98-106): p <- Rz(-yaw_one) (Rz(yaw_init) p + c_init - c_one),
 * plus the stage-two heading labels angle2class(bbox_gt[:, -1] - box_one[:, -1], 12)
 * (utils.py:53-60). obj_pts/obj_pts_two (B,M,3) point-major. bbox_gt may be NULL. */
int dal3_recenter_rotz(const float* obj_pts, int B, int M, const float* init_box7,
                       const float* box_one7, const float* bbox_gt7, float* obj_pts_two,
                       int64_t* heading_class_label, float* heading_residual_label,
                       dal3_stream stream);

/* ---- torch.max(x, 2)[0] as a standalone kernel (static_model.py:284,334): x (B,C,N)
 * contiguous fp32 -> out (B,C). The HBM-roofline kernel of BASELINE.json. */
int dal3_maxpool_n(const float* x, int64_t rows, int64_t n, float* out, dal3_stream stream);
/* The same for rows stored as DAL3_F32, DAL3_BF16 or DAL3_F16 (BASELINE.json configs C3 / C5: "bf16 storage"; 2-byte rows
 * move half the bytes: B*C*N*2 + B*C*2): x (rows, n) contiguous in `dtype`, out (rows) in the same dtype. Exact — the
 * maximum is one of the inputs — and, as torch.max, NaN for a row that holds a NaN (both entries). */
int dal3_maxpool_n_dtype(const void* x, int dtype, int64_t rows, int64_t n, void* out, dal3_stream stream);

/* ---- crop preparation right before the heads, float64 in, fp32 out ------------------
 * STATICTRACK.__getitem__ (static_model.py:529-546, 568-572) for B tracks at once: points = all frames' points
 * of every track stacked, global frame, (P_total,3) f64; offsets (B+1) delimit the tracks; pose (B,16) =
 * inv(veh_to_global) of each track's best-score frame, row-major; box (B,7) f64 = that frame's detection
 * already moved to the vehicle frame (transform_box, :574-588; a per-track scalar job the host does).
 * pts_out (B,N,3) fp32 point-major = Rz(-yaw)(pose p - centre) of N points drawn WITH replacement: choice
 * (B,N) i32 holds the draws (np.random.choice(P_b, N) for bit-reproducing the reference) or is NULL for the
 * device RNG keyed on (seed, item_offset + b, n). init_box_out (B,7) fp32 = box. */
int dal3_static_crop_prep(const double* points, const int64_t* offsets, const int32_t* choice,
                          const double* pose, const double* box, int B, int N, uint64_t seed,
                          int64_t item_offset, float* pts_out, float* init_box_out, dal3_stream stream);

/* DYNAMICTRACK.__getitem__ (dynamic_model.py:429-453, 490-507) for B items: per-frame point arrays of all
 * tracks concatenated (points (P_total,3) f64, frame_offsets (F_total+1)), per-frame global boxes (F_total,7)
 * f64, track_first (n_tracks+1) = first frame of each track in those arrays; item b = frame item_frame[b] of
 * track item_track[b]; pose (B,16) = inv(veh_to_global) of that frame. Outputs: pts_out (B,(2r+1)*n_per,4) with
 * the 0.1*(j-r) time channel, box_out (B,2s+1,8) with 0.1*(j-s), init_box_out (B,8) (the centre box before
 * re-centring). The reference's quirks are kept: missing/empty frames give zero points that are still moved by
 * the pose and the re-centring; missing boxes are zero rows that are still pose-transformed; points are rotated
 * by -yaw of the centre box, boxes only translated (yaw made relative). choice (B,2r+1,n_per) i32 or NULL. */
int dal3_dynamic_item_prep(const double* points, const int64_t* frame_offsets, const double* boxes,
                           const int64_t* track_first, const int32_t* item_track, const int32_t* item_frame,
                           const int32_t* choice, const double* pose, int B, int n_per, int r, int s,
                           uint64_t seed, int64_t item_offset, float* pts_out, float* box_out,
                           float* init_box_out, dal3_stream stream);

/* ---- write-back of refined boxes into the per-frame detections: the det_annos update of
 * postprocessing() in static_eval.py:71-87,148-155 and dynamic_eval.py:53-64,121-129, for P (track, frame) pairs.
 * final_boxes (n,7) f64, final_idx (P): which refined box a pair carries; pose_best (P,16) f64 veh_to_global of the
 * track's best frame (static) or NULL (dynamic: the box already is in the pair's frame); pose_inv (P,16) f64
 * inv(veh_to_global) of the pair's frame; track_box (P,7) f64 the track's own global box in that frame (the search
 * key); det (n_det,7) fp32 all frames' detections concatenated, UPDATED IN PLACE; det_start/det_count (P): the
 * pair's frame in det; active (P) u8: the reference skips frames that lack the matched GT object.
 * match (P) i32 out: matched row within the frame or -1; owner (n_det) i32 scratch. When two pairs hit one row
 * the later pair wins, as in the reference's sequential loop. */
int dal3_writeback_boxes(const double* final_boxes, const int32_t* final_idx, const double* pose_best,
                         const double* pose_inv, const double* track_box, float* det, const int64_t* det_start,
                         const int32_t* det_count, const uint8_t* active, int P, int64_t n_det, int32_t* match,
                         int32_t* owner, dal3_stream stream);

/* ---- the mask labels of the prepared items (training side of SURVEY.md 8(f) N1): static_model.py:548-556,
 * dynamic_model.py:455-487. Same inputs and the same draws (choice, or seed/item_offset) as the *_prep calls
 * above, so label n belongs to output point n. Face equations: (.,6,4) f64 rows [nx,ny,nz,d] of the matched
 * annotation's box as det3d's surface_equ_3d_jitv2 gives them (geometry.py:351-377) — O(#boxes) host work with
 * NumPy, see dal3_points_in_boxes. mask_label u8 in {0,1}.
 * static: gt_planes (B,6,4); the test runs on the vehicle-frame point `pose p` (before re-centring).
 * dynamic: window frame j of item b is tested in frame j's own vehicle frame: q = xform[b][j] (pose p) with
 * xform (B,2r+1,16) = inv(veh_to_global_j) @ inv(pose_b) (dynamic_model.py:481), planes (B,2r+1,6,4),
 * valid (B,2r+1) u8 = frame j has the matched annotation; out-of-track or invalid frames give zeros. */
int dal3_static_crop_labels(const double* points, const int64_t* offsets, const int32_t* choice, const double* pose,
                            int B, int N, uint64_t seed, int64_t item_offset, const double* gt_planes,
                            uint8_t* mask_label, dal3_stream stream);
int dal3_dynamic_item_labels(const double* points, const int64_t* frame_offsets, const int64_t* track_first,
                             const int32_t* item_track, const int32_t* item_frame, const int32_t* choice,
                             const double* pose, int B, int n_per, int r, uint64_t seed, int64_t item_offset,
                             const double* xform, const double* planes, const uint8_t* valid, uint8_t* mask_label,
                             dal3_stream stream);

/* ---- points-in-rotated-box: det3d box_np_ops.points_in_rbbox (det3d/core/bbox/box_np_ops.py:641-647 ->
 * geometry.py:240-275) as a (P,K) u8 table. points: P rows of >= 3 values, `stride` values apart, float32
 * (points_f64 = 0) or float64; planes (K,6,4) f64. A point is outside as soon as ((x nx + y ny) + z nz) + d >= 0
 * for a face, each operation rounded on its own; evaluated in float32 when f32_math != 0 (float32 points AND
 * float32 boxes, the reference's sweep case), in float64 otherwise. NaN coordinates count as inside (as there). */
int dal3_points_in_boxes(const void* points, int points_f64, int64_t P, int64_t stride, const double* planes, int K,
                         int f32_math, uint8_t* inside, dal3_stream stream);

/* ---- crop extraction from full sweeps: the per-detection loop of _create_pd_detection
 * (det3d/datasets/waymo/waymo_common.py:166-171, 193) for F frames at once. points (P_total,3) f32 vehicle-frame
 * sweeps concatenated, point_offsets (F+1); planes (K_total,6,4) f64 face equations of every frame's detections
 * (already in Waymo convention), box_offsets (F+1); max_points_per_frame bounds the launch. spheres (K_total,4)
 * f32 [cx,cy,cz,r^2]: a ball that CONTAINS the detection with a margin well above fp32 rounding (the library
 * culls with it before the exact test; it never decides membership; r^2 = +inf disables the cull).
 * dal3_crop_count -> counts (K_total) i64 = points inside each detection. box_start (K_total+1) = where each
 * detection's rows begin in out_points, [K_total] = the total: formed by the caller (exclusive prefix of counts) or, on
 * the device, by dal3_crop_starts -> box_start and (optional) out_offsets (K_total+1) for the detections laid out in
 * `order` (order[i] = the detection at output position i, e.g. track-major; NULL = as numbered): out_offsets[i] = rows
 * in front of position i, box_start[order[i]] = out_offsets[i]. out_points is (>= out_capacity, 3) f64; then
 * dal3_crop_fill -> out_points = veh_to_global (pose (F,16) f64 row-major) applied to the members, per detection
 * in sweep order (what `pose @ [lidars[indices]; 1]` yields); out_index (optional, i32) = index within the sweep.
 * Rows at or past out_capacity are not written (a caller that sized out_points from an estimate compares
 * box_start[K_total] with it afterwards; one that sized it from the total passes that total).
 * workspace: dal3_crop_workspace_bytes(K_total, max_points_per_frame); it carries state from count to fill. */
size_t dal3_crop_workspace_bytes(int64_t K_total, int64_t max_points_per_frame);
int dal3_crop_count(const float* points, const int64_t* point_offsets, const double* planes, const float* spheres,
                    const int64_t* box_offsets, int F, int64_t K_total, int64_t max_points_per_frame, int64_t* counts,
                    void* workspace, size_t workspace_bytes, dal3_stream stream);
int dal3_crop_fill(const float* points, const int64_t* point_offsets, const double* planes, const float* spheres,
                   const int64_t* box_offsets, int F, int64_t K_total, int64_t max_points_per_frame, const double* pose,
                   const int64_t* counts, const int64_t* box_start, double* out_points, int32_t* out_index,
                   int64_t out_capacity, const void* workspace, size_t workspace_bytes, dal3_stream stream);
int dal3_crop_starts(const int64_t* counts, const int64_t* order, int64_t K_total, int64_t* box_start,
                     int64_t* out_offsets, dal3_stream stream);
/* The same with out_offsets CAPPED at out_capacity (>= 0), for a caller that fills a buffer sized from an estimate and
 * hands (out_points, out_offsets) on to consumers that index out_points by them (dal3_static_crop_prep,
 * dal3_dynamic_item_prep): rows the fill dropped are then rows no offset points at — a detection past the capacity
 * reads as shorter or empty, never past the buffer. box_start is NOT capped: dal3_crop_fill needs the true starts, and
 * box_start[K_total] > out_capacity is how the caller learns that the buffer was too small. */
int dal3_crop_starts_capped(const int64_t* counts, const int64_t* order, int64_t K_total, int64_t* box_start,
                            int64_t* out_offsets, int64_t out_capacity, dal3_stream stream);

/* ---- training-mode building blocks of the shared-MLP stacks (first slice) --------------
 * What loss.backward() drives through Conv1d(k=1) + BatchNorm1d (batch statistics) + ReLU + max over points
 * (tools/static_model.py:271-295,326-339). Activations are POINT-MAJOR row-major fp32 (M x C), M = B*N points
 * (the memory of (B,C,N).transpose(2,1)); every channel count and M are multiples of 32 (the host pads). Each
 * layer's pre-BN output z is materialised; the consumer applies "act": y = z*scale[c] + shift[c], then max(y,0) if
 * relu_in (scale == NULL: identity). The host composes these per layer (3dal_pytorch_amd/train.py).
 *
 * dal3_tr_linear   z[p][co] (+)= sum_ci act(a[p][ci]) * Wop[co][ci] + bias.  transpose_w == 0: Wop = W, row-major
 *                  (c_out, c_in) with row stride ldw (forward); != 0: Wop[co][ci] = W[ci][co], W row-major
 *                  (c_in, c_out) (dgrad through a layer's own weight). bias: NULL | (c_out) when seg == 0 | per
 *                  segment bias[(p / seg) * c_out + co] (the decoder's per-crop global-feature term).
 * dal3_tr_colred   fixed-order column reductions over the points into out[2*C] f64:
 *                  mode 0: sum z, sum z^2 (batch statistics);
 *                  mode 1: dy = da * [act(z) > 0]: sum dy (= dbeta), sum dy * (z - mu)*rstd (= dgamma).
 *                  da: dense (M x C, row stride ldda) or NULL with (dg, arg, seg): da[p][c] = dg[s][c] if p is the
 *                  arg-max point arg[s][c] of its segment s = p / seg, else 0 (gradient of the max over points).
 * dal3_tr_bnbwd_apply  dz = k1[c] * (dy - k2[c] - xhat*k3[c])  (k1 = gamma*rstd, k2 = dbeta/M, k3 = dgamma/M).
 * dal3_tr_wgrad    dW[co][ci] = sum_p dz[p][co] * act(a[p][ci]); partial sums of point slices are added in a fixed
 *                  order (deterministic).
 * dal3_tr_segmax   g[s][c] = max_p act(z[p][c]) over segment s (relu), arg = index of the first maximum.
 * dal3_tr_segsum   out[s][c] = sum of x[p][c] over segment s. */
int dal3_tr_linear(const float* a, int64_t M, int c_in, int64_t lda, const float* scale, const float* shift, int relu_in,
                   const float* W, int64_t ldw, int transpose_w, const float* bias, int64_t seg, int c_out, float* z,
                   int64_t ldz, int accumulate, void* workspace, size_t workspace_bytes, dal3_stream stream);
size_t dal3_tr_linear_workspace_bytes(int c_in, int c_out);   /* 0 when the layer needs none (c_out % 128 != 0) */
/* dal3_tr_linear re-orders its weights into MFMA fragment order in front of every call (a ~4 us launch; a training step
 * makes 32 such calls). A caller that knows a step's layers up front packs them all — forward and transposed, the weights
 * do not change between a step's forward and backward — with ONE launch and hands each call its own image:
 *   dal3_tr_linear_pack_layout  -> the layout code (> 0) dal3_tr_linear will read for a call of this shape, 0 when that
 *                                  call uses no packed image (then pass it a plain workspace as before);
 *   dal3_tr_pack_many           packs n <= 48 layers (items: HOST array; out: device, 16-byte aligned,
 *                               dal3_tr_linear_workspace_bytes(c_in, c_out) bytes each; c_out / c_in as the CALL sees
 *                               them, i.e. swapped for transpose_w != 0);
 *   dal3_tr_linear_prepacked    dal3_tr_linear reading `packed` instead of packing. */
typedef struct {
    const float* W;
    int64_t ldw;
    int32_t transpose_w, c_out, c_in, mtb;               /* mtb: dal3_tr_linear_pack_layout() of the call */
    float* out;
} dal3_tr_pack_item;
int dal3_tr_linear_pack_layout(int64_t M, int c_in, int64_t seg, int c_out, int accumulate, int has_act);
int dal3_tr_pack_many(const dal3_tr_pack_item* items, int n, dal3_stream stream);
int dal3_tr_linear_prepacked(const float* a, int64_t M, int c_in, int64_t lda, const float* scale, const float* shift,
                             int relu_in, const float* W, int64_t ldw, int transpose_w, const float* bias, int64_t seg,
                             int c_out, float* z, int64_t ldz, int accumulate, const void* packed, dal3_stream stream);
/* A linear layer TOGETHER with the column reduction that follows it in a training step, taken in the kernel's epilogue
 * (round 4: the separate reduction passes re-read every activation / gradient tensor once — 1.0 of a step's 7.9 ms):
 *   dal3_tr_linear_bn_stats    = dal3_tr_linear_prepacked (forward, no accumulate) + dal3_tr_bn_stats over the first `rows`
 *                                rows of its output z;
 *   dal3_tr_linear_bnbwd_sums  = dal3_tr_linear_prepacked with transpose_w = 1, no activation, no bias (a dgrad:
 *                                da = dz W) + dal3_tr_bnbwd_sums(z = bz, da = its output): the sums of the BatchNorm/ReLU
 *                                backward of the layer whose post-activation gradient the dgrad has just produced
 *                                (bz, bscale .. brstd, gamma: that layer's pre-BN output and BatchNorm).
 * Same results as the two-call sequences up to summation order, deterministic either way. bn_stats: float64 running
 * sums per lane, a reordering of the separate pass's float64 additions (1e-6 of each vector's largest entry, the bar of
 * tests/test_gpu_train_fused.py). bnbwd_sums (RED 2): a tile's 32-64 terms of dy and dy*xhat are first added in FP32
 * (one fp32 partial per tile and channel), then joined to the float64 running sums — so dgamma / dbeta carry fp32
 * partial-sum rounding, bounded by 64 * 2^-24 = 3.8e-6 relative to sum(|terms|) per tile (the tests hold them to 1e-5 of
 * the vector's largest entry against float64 sums), not merely a reordering of float64 additions. Fused
 * when the shape takes the persistent linear kernel and rows == M (padding rows must stay out of the sums); otherwise
 * the library runs the two steps itself. Return: 1 fused, 0 ran as two steps, < 0 error.
 * workspace: dal3_tr_linear_red_workspace_bytes(rows, c_out). */
size_t dal3_tr_linear_red_workspace_bytes(int64_t rows, int c_out);
int dal3_tr_linear_bn_stats(const float* a, int64_t M, int c_in, int64_t lda, const float* scale, const float* shift,
                            int relu_in, const float* W, int64_t ldw, const float* bias, int64_t seg, int c_out, float* z,
                            int64_t ldz, const void* packed, int64_t rows, const float* gamma, const float* beta,
                            float* running_mean, float* running_var, float momentum, float eps, float* mu, float* rstd,
                            float* bn_scale, float* bn_shift, void* workspace, size_t workspace_bytes, dal3_stream stream);
int dal3_tr_linear_bnbwd_sums(const float* dz, int64_t M, int c_in, int64_t lddz, const float* W, int64_t ldw, int c_out,
                              float* da, int64_t ldda, const void* packed, int64_t rows, const float* bz, int64_t ldbz,
                              const float* bscale, const float* bshift, const float* bmu, const float* brstd,
                              const float* gamma, float* dgamma, float* dbeta, float* k1, float* k2, float* k3,
                              void* workspace, size_t workspace_bytes, dal3_stream stream);
/* The same layer on the "f16x3" arithmetic (DAL3_F16X3: fp16 MFMAs on (hi, lo) split operands, fp32 accumulate — the fp32
 * kernels' accuracy at a third of their MFMA time), for the FORWARD's big layers: operands must lie inside fp16's exponent
 * range (post-BatchNorm activations and weights do; gradients do not, so dgrad / wgrad calls stay on dal3_tr_linear).
 *   dal3_tr_linear_x3_layout -> 0 when the call does not qualify (accumulate, M % 256, c_in % 64 or > 2048, c_out % 256, seg % 256,
 *                               M < 4096), else the layout code to put into dal3_tr_pack_item.mtb for dal3_tr_pack_many
 *                               (image size: dal3_tr_linear_workspace_bytes(c_in, c_out), as for the fp32 image);
 *   dal3_tr_linear_x3           z = act(a) W^T + bias from that image. a, z, bias 16-byte aligned. */
int dal3_tr_linear_x3_layout(int64_t M, int c_in, int64_t seg, int c_out, int accumulate, int has_act);
int dal3_tr_linear_x3(const float* a, int64_t M, int c_in, int64_t lda, const float* scale, const float* shift, int relu_in,
                      const float* bias, int64_t seg, int c_out, float* z, int64_t ldz, const void* packed,
                      const uint32_t* in_amax, dal3_stream stream);
/* ... and for an operand far below fp16's range (a dgrad's dz): in_amax (64 device words; scale must be NULL then) hold, as
 * their maximum, the bit pattern of the operand's largest |value| — what dal3_tr_bnbwd_apply_amax leaves there,
 * atomicMax'ed into 64 words the caller zeroed (64, not one: the atomics of 32,768 waves on one word take longer than
 * the layer) — and the kernel multiplies the operand by the power of two that brings that value to 2^14 and the result
 * by its inverse (both exact). Entries down to 2^-17 of the largest keep the split's 22 bits, down to 2^-28 at least 11;
 * the error stays below 1e-6 of the RESULT's range throughout. */
int dal3_tr_bnbwd_apply_amax(const float* z, int64_t M, int C, int64_t ldz, const float* da, int64_t ldda, const float* dg,
                             const int32_t* arg, int64_t seg, const float* scale, const float* shift, const float* mu,
                             const float* rstd, const float* k1, const float* k2, const float* k3, float* dz, int64_t lddz,
                             uint32_t* amax, dal3_stream stream);
/* dal3_tr_linear_pool on the same arithmetic (same arguments and workspace); _ok: 1 when the shape qualifies
 * (M % 256, c_in % 64, c_out % 256, seg % 256 == 0, M >= 4096). g / arg agree with dal3_tr_linear_pool's to the fp32 kernels'
 * accuracy (not bit for bit: the products are formed differently). */
int dal3_tr_linear_pool_x3_ok(int64_t M, int c_in, int64_t seg, int c_out);
int dal3_tr_linear_pool_x3(const float* a, int64_t M, int c_in, int64_t lda, const float* scale, const float* shift, int relu_in,
                           const float* W, int64_t ldw, const float* bias, const float* out_scale, const float* out_shift,
                           int64_t seg, int c_out, float* g, int32_t* arg, void* workspace, size_t workspace_bytes,
                           dal3_stream stream);
size_t dal3_tr_colred_workspace_bytes(int64_t M, int C);
/* out = act(x) (as dal3_tr_act_dropout without a multiplier) AND sums (2 C float64) = [sum out | sum out^2] over the M rows,
 * one pass; workspace: dal3_tr_colred_workspace_bytes(M, C). */
int dal3_tr_act_colsum(const float* x, int64_t M, int C, int64_t ldx, const float* scale, const float* shift, int relu,
                       float* out, int64_t ldo, void* workspace, size_t workspace_bytes, double* sums, dal3_stream stream);
int dal3_tr_colred(const float* z, int64_t M, int C, int64_t ldz, int mode, const float* da, int64_t ldda,
                   const float* dg, const int32_t* arg, int64_t seg, const float* scale, const float* shift,
                   const float* mu, const float* rstd, void* workspace, size_t workspace_bytes, double* out,
                   dal3_stream stream);
/* The segmentation term of the reference's three criteria (tools/static_model.py:378-380, tools/dynamic_model.py:337-339:
 * F.nll_loss(F.log_softmax(logits.view(-1, 2), dim=1), mask_label.view(-1).long())) in one pass: loss[0] = mean over the
 * M points of -log_softmax(logits[p])[label[p]], and dlogits[p] = softmax(logits[p]) - onehot(label[p]) (the gradient
 * of M * loss; the caller scales it). logits, dlogits: (M, 2) fp32 contiguous; labels: (M,) float32 or int64 (0 / 1).
 * Sums in float64, blocks added in index order (reproducible). workspace: dal3_tr_seg_ce_workspace_bytes(M). */
size_t dal3_tr_seg_ce_workspace_bytes(int64_t M);
/* Per-channel coefficients of that backward (sums over the items in a fixed order: deterministic): with D = dg * [g > 0] and xhat = (zarg - mu) * rstd
 * (dg, g, zarg: (B, C) fp32 — upstream gradient, pooled value, pre-BN value at the pooled point), coef (4, C) float64 =
 * dbeta = sum_b D | dgamma = sum_b D xhat | A = -k1 k2 + k1 k3 rstd mu | Bc = -k1 k3 rstd  (k1 = gamma rstd, k2 = dbeta / M,
 * k3 = dgamma / M), and kd (B, C) fp32 = k1 * D. */
int dal3_tr_pool_coef(const float* dg, const float* g, const float* zarg, const float* mu, const float* rstd,
                      const float* gamma, int B, int C, int64_t M, double* coef, float* kd, dal3_stream stream);
/* The float64 algebra around that layer, K = its input channels (64, 128 or 256), C its output channels, W (C, K) row-major
 * with row stride ldw, b (C):
 *   dal3_tr_pool_moments  forward: sums (2 C) float64 = [sum z | sum z^2] of z = W a + b over M points, from m1 = sum a
 *                         (K, float64) and Sc = sum of the CENTRED a a^T (K x K fp32): mu = W m1 / M + b,
 *                         var_c = max(w_c^T (Sc / M) w_c, 0), sums = [mu M | (var + mu^2) M];
 *   dal3_tr_pool_gv       backward: G (K x K fp32) = W^T diag(Bc) W and v (K fp32) = (A + Bc b)^T W, with coef (4, C) from
 *                         dal3_tr_pool_coef — the operands of the dense part of da = a G + v;
 *   dal3_tr_pool_dw       backward: dW (C, K) fp32 = A m1^T + diag(Bc) (W S + b m1^T) + dWs, S = sum a a^T (K x K fp32)
 *                         given directly (centred = 0) or as Sc + m1 m1^T / M (centred != 0); dWs (C, K): the sparse term. */
/* zarg (B,C) = the pooled layer's pre-BatchNorm value at each pooled point: W[c] . a[b*N + arg[b][c]] + bias[c], from the
 * layer's input activation a (B*N, K) — what dal3_tr_pool_coef takes, when the fused forward (dal3_tr_linear_pool) has not
 * written the layer's output. fp32, a fixed order of additions. */
/* The FIRST layer of a stack (conv1: 3, 4 or 8 input channels -> 64 or 128) as VALU kernels, on the points as they are
 * (x: M rows of c_in floats, row stride ldx — no zero-padded copy):
 *   dal3_tr_conv1_bn_stats  z[p][c] = bias[c] + sum_k W[c][k] x[p][k] for the Mp >= M rows of z (rows >= M get x = 0), and
 *                           dal3_tr_bn_stats over the M real rows in the same pass;
 *   dal3_tr_conv1_wgrad     dW (c_out, c_in) float32, row-major: dW[c][k] = sum_{p < M} dz[p][c] x[p][k].
 * workspace: dal3_tr_conv1_workspace_bytes(rows, c_out). Float64 partial sums per 256 rows, added in a fixed order. */
size_t dal3_tr_conv1_workspace_bytes(int64_t Mp, int c_out);
int dal3_tr_conv1_bn_stats(const float* x, int64_t M, int64_t Mp, int c_in, int64_t ldx, const float* W, int64_t ldw,
                           const float* bias, int c_out, float* z, int64_t ldz, const float* gamma, const float* beta,
                           float* running_mean, float* running_var, float momentum, float eps, float* mu, float* rstd,
                           float* scale, float* shift, void* workspace, size_t workspace_bytes, dal3_stream stream);
int dal3_tr_conv1_wgrad(const float* dz, int64_t lddz, const float* x, int64_t M, int c_in, int64_t ldx, int c_out, void* workspace,
                        size_t workspace_bytes, float* dW, dal3_stream stream);
/* parse_output_to_tensors (tools/static_model.py:64-92) in train mode, where its seven results must be tensors of their own:
 * the (B, 39) box_pred -> center (B,3), heading_scores (B,12), heading_residuals_normalized (B,12), heading_residuals =
 * normalized * (pi / 12), size_scores (B,3), size_residuals_normalized (B,3,3), size_residuals = normalized * MEAN_SIZE, all
 * contiguous, in one launch; _backward puts their gradients (any of them NULL = zeros) back side by side as the (B, 39)
 * gradient of box_pred: scaled gradients multiplied, then added to the normalized ones, as autograd would. */
int dal3_parse_box_pred(const float* box_pred, int64_t ldb, int64_t B, float* center, float* heading_scores,
                        float* heading_residuals_normalized, float* heading_residuals, float* size_scores,
                        float* size_residuals_normalized, float* size_residuals, dal3_stream stream);
int dal3_parse_box_pred_backward(const float* g_center, const float* g_heading_scores, const float* g_heading_residuals_normalized,
                                 const float* g_heading_residuals, const float* g_size_scores,
                                 const float* g_size_residuals_normalized, const float* g_size_residuals, int64_t B,
                                 float* g_box_pred, dal3_stream stream);
/* The per-item FC tails in train mode (static_model.py:336-338, dynamic_model.py:247-248, :284-285, :306-311;
 * `_PointHead.tail` with self.training): Linear -> BatchNorm1d over the B ITEMS -> ReLU with rows = items,
 * 2 <= B <= dal3_tr_fc_max_rows(). One launch per layer forward, two backward; no padding, no packed weight image; every
 * sum in index order (float64 for the batch statistics and the BatchNorm-backward sums).
 *   dal3_tr_fc_forward     z (B, c_out) = act(a) Wop^T + bias, act = (in_scale, in_shift, relu_in) of the layer below (NULL:
 *                          identity), Wop = W (c_out, c_in) or, transpose_w, W^T of a (c_in, c_out) matrix (the dgrad:
 *                          da_prev = dz W); with gamma != NULL also dal3_tr_bn_stats of z over its B rows (mu, rstd,
 *                          scale, shift, the running statistics when given) in the same launch
 *   dal3_tr_fc_backward_w  of the layer whose pre-BN output is z: from da (gradient w.r.t. relu(bn(z)); with scale == NULL
 *                          the layer has no BatchNorm and da IS dz) -> dz (B, c_out; may be NULL), dgamma, dbeta, db
 *                          (zeros in front of a BatchNorm), dW (c_out, c_in) = dz^T act(a_prev) (may be NULL) */
int dal3_tr_fc_max_rows(void);
int dal3_tr_fc_max_act_cin(void);       /* most input channels of a layer whose input carries an activation (hidden layers of a tail) */
int dal3_tr_fc_forward(const float* a, int64_t B, int c_in, int64_t lda, const float* in_scale, const float* in_shift, int relu_in,
                       const float* W, int64_t ldw, int transpose_w, const float* bias, int c_out, float* z, int64_t ldz,
                       const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                       float* mu, float* rstd, float* scale, float* shift, dal3_stream stream);
int dal3_tr_fc_backward_w(const float* da, int64_t ldda, int64_t B, int c_out, const float* z, int64_t ldz, const float* scale,
                          const float* shift, const float* mu, const float* rstd, const float* gamma, float* dgamma, float* dbeta,
                          const float* a_prev, int c_in, int64_t lda, const float* in_scale, const float* in_shift, int relu_in,
                          float* dz, int64_t lddz, float* dW, int64_t lddw, float* db, dal3_stream stream);
/* out (n_seg, C) = z[s*seg + arg[s][c]][c]: the same values gathered from a MATERIALISED layer output (dal3_tr_segmax's arg) */
int dal3_tr_gather_at(const float* z, int64_t ldz, const int32_t* arg, int64_t seg, int n_seg, int C, float* out,
                      dal3_stream stream);
int dal3_tr_pool_zarg(const int32_t* arg, const float* a, int64_t lda, const float* W, int64_t ldw, const float* bias, int B, int C,
                      int K, int N, float* zarg, dal3_stream stream);
int dal3_tr_pool_moments(const float* W, int64_t ldw, const float* b, const double* m1, const float* Sc, int64_t M, int C, int K,
                         double* sums, dal3_stream stream);
size_t dal3_tr_pool_gv_workspace_bytes(int K);
int dal3_tr_pool_gv(const double* coef, const float* W, int64_t ldw, const float* b, int C, int K, float* G, float* v,
                    void* workspace, size_t workspace_bytes, dal3_stream stream);
int dal3_tr_pool_dw(const double* coef, const float* W, int64_t ldw, const float* b, const float* S, const double* m1, int64_t M,
                    int centred, const float* dWs, int C, int K, float* dW, dal3_stream stream);
/* The two sparse terms of the backward of conv -> BN -> ReLU -> max over an item's N points (one pooled point per item
 * and channel: arg (B,C) int32, as dal3_tr_segmax / dal3_tr_linear_pool return it; kd (B,C) = k1 * dy at those points):
 *   da[b*N + arg[b][c]][0..K) += kd[b][c] * W[c][0..K)     in place, deterministic (channels of a point added in channel order)
 *   dWs[c][0..K) = sum over b (in order) of kd[b][c] * a[b*N + arg[b][c]][0..K)
 * K = 64, 128 or 256; 2 N + C + 1 <= 16384 and C <= 4096 (the buckets of an item live in LDS). */
int dal3_tr_pool_sparse(const int32_t* arg, const float* kd, const float* W, int64_t ldw, const float* a, int64_t lda, int B,
                        int C, int K, int N, float* da, int64_t ldda, float* dWs, dal3_stream stream);
/* The five box terms of one box estimate (tools/static_model.py:382-424, tools/dynamic_model.py:341-383), each the mean
 * over the B items: losses[0..4] = centre (Huber, delta 2, of ||center - label||), heading class (cross-entropy, 12
 * bins), heading residual (Huber, delta 1, of the label bin's normalised residual against label / (pi/12)), size class
 * (cross-entropy, 3), size residual (Huber, delta 1, of ||label / mean_size[class] - the label class's normalised
 * residual||) — unweighted; and g_* = the gradient of the matching loss w.r.t. that input (the other entries 0).
 * All inputs contiguous fp32 except the two int64 class labels. A class label outside [0, 12) / [0, 3) (an ignore value
 * such as -1: F.nll_loss of the stock criterion raises on those) makes every entry of `losses` NaN and that item's rows
 * of all five gradients NaN (a step taken on them poisons the parameters: loud whether or not the loss is looked at);
 * nothing is read out of bounds. One launch. */
int dal3_tr_box_loss(const float* center, const float* center_label, const float* heading_scores,
                     const float* heading_residuals_normalized, const int64_t* heading_class_label,
                     const float* heading_residuals_label, const float* size_scores,
                     const float* size_residuals_normalized, const int64_t* size_class_label,
                     const float* size_residuals_label, int B, float* losses, float* g_center, float* g_heading_scores,
                     float* g_heading_residuals_normalized, float* g_size_scores, float* g_size_residuals_normalized,
                     dal3_stream stream);
int dal3_tr_seg_ce(const float* logits, const void* labels, int labels_are_int64, int64_t M, float* loss, float* dlogits,
                   void* workspace, size_t workspace_bytes, dal3_stream stream);
/* The two reductions WITH their per-channel epilogues (what a training step calls: the second stage of the reduction
 * carries the epilogue, a layer's statistics are two launches): dal3_tr_bn_stats = dal3_tr_colred mode 0 followed by
 * dal3_tr_bn_finalize, dal3_tr_bnbwd_sums = mode 1 followed by dal3_tr_bnbwd_coef — same sums, same results.
 * workspace: dal3_tr_colred_workspace_bytes(M, C). */
int dal3_tr_bn_stats(const float* z, int64_t M, int C, int64_t ldz, const float* gamma, const float* beta,
                     float* running_mean, float* running_var, float momentum, float eps, float* mu, float* rstd,
                     float* scale, float* shift, void* workspace, size_t workspace_bytes, dal3_stream stream);
int dal3_tr_bnbwd_sums(const float* z, int64_t M, int C, int64_t ldz, const float* da, int64_t ldda, const float* dg,
                       const int32_t* arg, int64_t seg, const float* scale, const float* shift, const float* mu,
                       const float* rstd, const float* gamma, float* dgamma, float* dbeta, float* k1, float* k2, float* k3,
                       void* workspace, size_t workspace_bytes, dal3_stream stream);
/* per-channel epilogues of the two reductions, one launch each: batch mean / biased variance -> mu, rstd, the folded
 * affine scale = gamma*rstd, shift = beta - mu*scale, and torch's running-statistics update (unbiased variance,
 * `momentum`; running_* may both be NULL); backward sums -> dgamma, dbeta and k1..k3 of dal3_tr_bnbwd_apply. */
int dal3_tr_bn_finalize(const double* sums, int C, int64_t M, const float* gamma, const float* beta, float* running_mean,
                        float* running_var, float momentum, float eps, float* mu, float* rstd, float* scale,
                        float* shift, dal3_stream stream);
int dal3_tr_bnbwd_coef(const double* sums, int C, int64_t M, const float* gamma, const float* rstd, float* dgamma,
                       float* dbeta, float* k1, float* k2, float* k3, dal3_stream stream);
int dal3_tr_bnbwd_apply(const float* z, int64_t M, int C, int64_t ldz, const float* da, int64_t ldda, const float* dg,
                        const int32_t* arg, int64_t seg, const float* scale, const float* shift, const float* mu,
                        const float* rstd, const float* k1, const float* k2, const float* k3, float* dz, int64_t lddz,
                        dal3_stream stream);
/* dal3_tr_bnbwd_apply (dense da) that also returns the column sums of dz over every segment of sum_seg rows — seg_sums
 * (M / sum_seg, C) fp32, float64 inside, a fixed order — in the same pass: the per-crop gradient of dconv1's per-crop
 * term. C % 64 == 0, sum_seg % 128 == 0, M % sum_seg == 0. */
size_t dal3_tr_bnbwd_apply_segsum_workspace_bytes(int64_t M, int C);
int dal3_tr_bnbwd_apply_segsum(const float* z, int64_t M, int C, int64_t ldz, const float* da, int64_t ldda,
                               const float* scale, const float* shift, const float* mu, const float* rstd, const float* k1,
                               const float* k2, const float* k3, float* dz, int64_t lddz, int64_t sum_seg, float* seg_sums,
                               void* workspace, size_t workspace_bytes, dal3_stream stream);
/* dal3_tr_wgrad on the f16x3 arithmetic (both operands split in two fp16 halves, fp32 accumulate). dz_amax: 64 device words
 * whose maximum is the bit pattern of dz's largest |value| (dal3_tr_bnbwd_apply_amax), or NULL for a dz inside fp16's
 * range. _workspace_bytes() == 0: the shape does not qualify (c_out x c_in must be cut by 256 x 256, 128 x 256, 128 x 128
 * or 512 x 64 blocks; M >= 8192) — use dal3_tr_wgrad. */
size_t dal3_tr_wgrad_x3_workspace_bytes(int64_t M, int c_out, int c_in);
int dal3_tr_wgrad_x3(const float* dz, int64_t lddz, const float* a, int64_t lda, const float* scale, const float* shift,
                     int relu_in, const uint32_t* dz_amax, int64_t M, int c_out, int c_in, void* workspace,
                     size_t workspace_bytes, float* dW, dal3_stream stream);
size_t dal3_tr_wgrad_workspace_bytes(int64_t M, int c_out, int c_in);
int dal3_tr_wgrad(const float* dz, int64_t lddz, const float* a, int64_t lda, const float* scale, const float* shift,
                  int relu_in, int64_t M, int c_out, int c_in, void* workspace, size_t workspace_bytes, float* dW,
                  dal3_stream stream);
/* dW == NULL in dal3_tr_wgrad / dal3_tr_wgrad_x3: only the per-slice partial sums are left in `workspace`
 * (dal3_tr_wgrad*_workspace_bytes / (4 c_out c_in) slices of c_out * c_in floats, which the caller keeps), and
 * dal3_tr_wgrad_final_many adds the slices of up to 24 such calls in ONE launch — the same fixed-order sums as the
 * per-call second stage (nothing reads a weight gradient before the backward pass ends). */
typedef struct dal3_tr_wgrad_part {
    const float* part;                                   /* the call's workspace */
    int64_t n_slices, n;                                 /* n = c_out * c_in */
    float* dW;
} dal3_tr_wgrad_part;
int dal3_tr_wgrad_final_many(const dal3_tr_wgrad_part* items, int n, dal3_stream stream);
int dal3_tr_segmax(const float* z, int64_t ldz, int64_t seg, int C, const float* scale, const float* shift, float* g,
                   int32_t* arg, int64_t n_seg, void* workspace /* n_seg*C*8 bytes, 8-byte aligned */,
                   size_t workspace_bytes, dal3_stream stream);
/* dal3_tr_act_dropout: out = act(x) * m, one pass. act as in dal3_tr_linear (scale == NULL: identity; relu applied after the
 * affine when relu != 0). m = mult[p][c] (row stride ldm) when mult != NULL — a caller-supplied multiplier, e.g. the
 * reference's own draw in a parity test — else m = keep / (1 - p_drop) with keep ~ Bernoulli(1 - p_drop) drawn by a
 * counter-based generator keyed on (seed, *step, element index; one draw per four consecutive elements, the keep
 * probability exact to 2^-16): forward and backward pass the same (seed, step) and get the same multiplier without storing it; step (device int64, may be NULL = 0) lets a hipGraph replay draw afresh.
 * Replaces: self.dropout = nn.Dropout(p=0.5) applied to relu(dbn4(dconv4(x))) (static_model.py:268,292-293) and its
 * backward. */
/* The 128 -> 2 logits layer with the Dropout in front of it (static_model.py:292-294: dconv5(dropout(relu(dbn4(z))))) as three
 * VALU kernels; the post-Dropout activation is never materialised, the multiplier m is re-created from its key exactly as
 * dal3_tr_act_dropout draws it (same mult / seed / step / p_drop arguments). C must be 128.
 *   dal3_tr_head2_forward  logits[p][j] = bias[j] + sum_c W[j][c] * m[p][c] * act(z[p][c])       (W: 2 rows of C, row stride ldw)
 *   dal3_tr_head2_dgrad    da[p][c] = m[p][c] * (dlogits[p][0] W[0][c] + dlogits[p][1] W[1][c])  (d / d act(z): Dropout undone)
 *   dal3_tr_head2_wgrad    dWb (258 floats): dW (2, 128) row-major, then db[0], db[1]
 *                          with dW[j][c] = sum_p dlogits[p][j] * m[p][c] * act(z[p][c]), db[j] = sum_p dlogits[p][j];
 *                          float64 partial sums per 256 rows added in a fixed order, rounded to float32 at the end.
 *                          workspace: dal3_tr_head2_wgrad_workspace_bytes(M). */
int dal3_tr_head2_forward(const float* z, int64_t M, int C, int64_t ldz, const float* scale, const float* shift, int relu,
                          const float* mult, int64_t ldm, uint64_t seed, const int64_t* step, float p_drop, const float* W,
                          int64_t ldw, const float* bias, float* logits, dal3_stream stream);
int dal3_tr_head2_dgrad(const float* dlogits, int64_t M, int C, const float* mult, int64_t ldm, uint64_t seed, const int64_t* step,
                        float p_drop, const float* W, int64_t ldw, float* da, int64_t ldda, dal3_stream stream);
/* dal3_tr_head2_dgrad together with dal3_tr_bnbwd_sums of the layer below (z = bz, da = the gradient just formed; the layer
 * whose relu(bn(bz)) feeds the Dropout): one kernel forms da, gates it and takes the two sums; workspace:
 * dal3_tr_colred_workspace_bytes(M, 128). Same results as the two calls up to the order of the float64 additions. */
int dal3_tr_head2_dgrad_bnbwd(const float* dlogits, int64_t M, int C, const float* mult, int64_t ldm, uint64_t seed,
                              const int64_t* step, float p_drop, const float* W, int64_t ldw, float* da, int64_t ldda,
                              const float* bz, int64_t ldbz, const float* bscale, const float* bshift, const float* bmu,
                              const float* brstd, const float* gamma, float* dgamma, float* dbeta, float* k1, float* k2, float* k3,
                              void* workspace, size_t workspace_bytes, dal3_stream stream);
size_t dal3_tr_head2_wgrad_workspace_bytes(int64_t M);
int dal3_tr_head2_wgrad(const float* dlogits, const float* z, int64_t M, int C, int64_t ldz, const float* scale, const float* shift,
                        int relu, const float* mult, int64_t ldm, uint64_t seed, const int64_t* step, float p_drop, void* workspace,
                        size_t workspace_bytes, float* dWb, dal3_stream stream);
int dal3_tr_act_dropout(const float* x, int64_t M, int C, int64_t ldx, const float* scale, const float* shift, int relu,
                        const float* mult, int64_t ldm, uint64_t seed, const int64_t* step, float p_drop, float* out,
                        int64_t ldo, dal3_stream stream);

/* dal3_tr_linear_pool: a layer followed by BN + ReLU + max over the points of each segment, WITHOUT materialising the
 * layer's output: g[s][co] = max_p relu(z[p][co] * out_scale[co] + out_shift[co]), arg = index (within the segment) of
 * the first maximum, with z = act(a) . W^T + bias as in dal3_tr_linear (forward orientation, per-channel bias).
 * For layers whose BN affine is known before the layer runs (ins_seg's conv5, the point heads' conv4: train.py gets
 * their batch statistics from the second moments of the layer's INPUT). Bit-identical to dal3_tr_linear followed by
 * dal3_tr_segmax. c_out a multiple of 128, seg a multiple of 32 dividing M.
 * Replaces: F.relu(self.bn5(self.conv5(out4))) + torch.max(out5, 2) (static_model.py:283-284, :333-334). */
size_t dal3_tr_linear_pool_workspace_bytes(int c_in, int c_out, int64_t n_seg);
int dal3_tr_linear_pool(const float* a, int64_t M, int c_in, int64_t lda, const float* scale, const float* shift, int relu_in,
                        const float* W, int64_t ldw, const float* bias, const float* out_scale, const float* out_shift,
                        int64_t seg, int c_out, float* g, int32_t* arg, void* workspace, size_t workspace_bytes,
                        dal3_stream stream);
int dal3_tr_segsum(const float* x, int64_t ldx, int64_t seg, int C, float* out, int64_t n_seg, dal3_stream stream);

/* ---- one fused shared-MLP layer, for layer-wise tests: y = relu?(W' x + b') with BN folded,
 * x (B,C_in,N) strided -> y (B,N,C_out) point-major. */
int dal3_shared_mlp_layer(const dal3_layer* layer, int relu, dal3_bcn x, int B, int N, float* y,
                          void* workspace, size_t workspace_bytes, dal3_stream stream);
size_t dal3_shared_mlp_layer_workspace_bytes(int c_in, int c_out);

/* ---- whole-model sequencers --------------------------------------------------------------- */
typedef struct {
    int32_t B, N;                        /* crops, points per crop */
    int32_t two_stage;                   /* 0: StaticModelOneBoxEst, 1: StaticModelTwoBoxEst */
    int32_t sampler;                     /* DAL3_SAMPLER_* */
    int32_t dtype;                       /* DAL3_F32 | DAL3_BF16 | DAL3_F16: what the w_* blobs were packed as */
    int32_t reserved;
    uint64_t seed;
    int64_t item_offset;
    dal3_bcn pts;                        /* (B,3,N) logical */
    const float* init_box;               /* (B,7) */
    const float* bbox_gt;                /* (B,7) or NULL (only feeds the stage-two labels) */
    const int32_t* choice;               /* (B,512) for DAL3_SAMPLER_CHOICE, else NULL */
    const void* w_ins_seg;
    const void* w_box_est_one;           /* the only estimator when two_stage == 0 */
    const void* w_box_est_two;           /* NULL when two_stage == 0 */
    /* outputs */
    float* logits;                       /* (B,N,2) */
    uint8_t* mask;                       /* (B,N) */
    float* box_pred_one;                 /* (B,39); [:, :3] comes back with init_box[:, :3] added when two_stage */
    float* heading_residuals_one;        /* (B,12) */
    float* size_residuals_one;           /* (B,9) */
    float* center_one;                   /* (B,3) = box_pred_one[:, :3] + init_box[:, :3] */
    float* box_one;                      /* (B,7) decoded stage-one box (yaw + init yaw) */
    float* box_pred_two;                 /* two_stage only from here */
    float* heading_residuals_two;
    float* size_residuals_two;
    float* center_two;                   /* (B,3) = box_pred_two[:, :3] + center_one */
    int64_t* heading_class_label_two;    /* (B) */
    float* heading_residuals_label_two;  /* (B) */
    float* boxes7;                       /* (B,7) refined boxes (static_eval.py:269-288) */
    int32_t* counts;                     /* (B) segmented points per crop */
    int32_t* obj_idx;                    /* (B,512) */
    void* workspace;
    size_t workspace_bytes;
} dal3_static_args;

/* phases: the CHOICE sampler needs `counts` on the host between them */
enum { DAL3_PHASE_SEG = 1, DAL3_PHASE_BOX = 2, DAL3_PHASE_ALL = 3 };

size_t dal3_static_workspace_bytes(int B, int N, int two_stage);
/* StaticModelOneBoxEst.forward / StaticModelTwoBoxEst.forward (static_model.py:117-146,158-239)
 * + refined-box decode (static_eval.py:269-288) */
int dal3_static_forward(const dal3_static_args* args, int phases, dal3_stream stream);

typedef struct {
    int32_t B, N, n_box;                 /* items, points per item (5*1024), boxes per window (101) */
    int32_t sampler;
    int32_t dtype;                       /* DAL3_F32 | DAL3_BF16 | DAL3_F16 */
    int32_t reserved;
    uint64_t seed;
    int64_t item_offset;
    dal3_bcn pts;                        /* (B,4,N) logical */
    dal3_bcn box;                        /* (B,8,n_box) logical */
    const float* init_box8;              /* (B,8) or NULL: decode adds [:, :3] and yaw [:, 6] (dynamic_eval.py:236-240) */
    const int32_t* choice;               /* (B,2560) for DAL3_SAMPLER_CHOICE */
    const void* w_ins_seg;
    const void* w_point_emb;
    const void* w_box_emb;
    const void* w_box_est;
    float* logits;                       /* (B,N,2) */
    uint8_t* mask;                       /* (B,N) */
    float* embedding;                    /* (B,384) = [point_e | box_e] */
    float* box_pred;                     /* (B,39) */
    float* heading_residuals;            /* (B,12) */
    float* size_residuals;               /* (B,9) */
    float* boxes7;                       /* (B,7) */
    int32_t* counts;                     /* (B) */
    int32_t* obj_idx;                    /* (B,2560) */
    void* workspace;
    size_t workspace_bytes;
} dal3_dynamic_args;

size_t dal3_dynamic_workspace_bytes(int B, int N, int n_box);
/* DynamicModel.forward (dynamic_model.py:121-155) + decode (dynamic_eval.py:226-242) */
int dal3_dynamic_forward(const dal3_dynamic_args* args, int phases, dal3_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* DAL3_H */
