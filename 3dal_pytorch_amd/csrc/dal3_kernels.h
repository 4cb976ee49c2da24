// dal3_kernels.h — host-visible views of the packed weights and the kernel launchers.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/dal3.h"
#include "dal3_device.h"

// point tiles (of 32 points) per wave; see profiles/LEDGER_r01_r03.md §4 "register budget"
#ifndef DAL3_ENC_T
#define DAL3_ENC_T 2
#endif
#ifndef DAL3_DEC_T
#define DAL3_DEC_T 1
#endif
#ifndef DAL3_HEAD_T
#define DAL3_HEAD_T 1
#endif

// waves per workgroup of the shared-MLP kernels (waves never cooperate, so this only sets the
// granularity at which the dispatcher refills a CU)
#ifndef DAL3_WG_WAVES
#define DAL3_WG_WAVES 4
#endif

// weight-fragment prefetch depth (fragments of 1 KiB per wave kept in flight), see WRing
#ifndef DAL3_PF
#define DAL3_PF 8
#endif
// MEAN_SIZE_ARR of the reference (tools/static_model.py:17-21), the ONE copy in the library: the decode kernels, the
// criterion kernel and dal3_mean_size() (which the Python side checks against arch.MEAN_SIZE when it loads the library)
#define DAL3_MEAN_SIZE_VALUES 4.8, 1.8, 1.5, 10.0, 2.6, 3.2, 2.0, 1.0, 1.6
#define DAL3_BLOB_TAIL_FLOATS 8192   // 32 KiB: room for a ring's over-read past the last stream

// ---- packed PointNetInstanceSeg (BN folded). A "frag block" is [4 q][64 lanes] float4 x KT k-tiles
// of one 32-row output tile, see dal3_device.h::mma_block_ring. Each kernel reads ONE stream of
// blocks laid out in the order it consumes them:
//   enc_stream  conv2 (64->64) | conv3 (64->64) | conv4 (64->128) | conv5 (128->1024), out-tile major
//   dec_stream  conv2 | dconv1a chunk 0 | { dconv1a chunk c+1 (8 frags), dconv2 chunk c (32 frags) }
//               for c = 0..15 (chunk 16 = zeros) | dconv3 (256->128) | dconv4 (128->128)
// where dconv1a = dconv1 columns 0..63 (the per-point out2 part, one 32-output chunk = 8 fragments)
// and dconv2 is K-major (the 32 fragments that multiply dconv1 chunk c into the 8 output tiles).
struct InsSegW {
    const float* w1;          // first layer [2][2][64]
    const float* b1;          // 64
    const float* b2;
    const float* b3;
    const float* b4;
    const float* b5;          // 1024
    const float* dw1g;        // row-major (512,1024): dconv1 columns 64..1087 (the global-feature part)
    const float* db1;         // 512
    const float* db2;
    const float* db3;
    const float* db4;
    const float* dw5;         // row-major (2,128)
    const float* db5;         // 2
    const f32x4* enc_stream;
    const f32x4* dec_stream;
    const f32x4* lat_stream;  // dconv1a (16 out-tiles x 2 k-tiles) | dconv2 (8 x 16), OUT-TILE major: the latency kernels
};                            // (dal3_latency.hip) give each wave whole output tiles; dec_stream interleaves them K-major
enum {                        // stream geometry in fragments of 256 floats
    ENC_W2 = 0, ENC_W3 = 16, ENC_W4 = 32, ENC_W5 = 64, ENC_FRAGS = 64 + 512,
    DEC_W2 = 0, DEC_MIX = 16, DEC_MIX_FRAGS = 8 + 16 * 40, DEC_W3 = 16 + 648, DEC_W4 = 16 + 648 + 128,
    DEC_FRAGS = 16 + 648 + 128 + 64,
    LAT_W1A = 0, LAT_W2 = 128, LAT_FRAGS = 128 + 512
};

struct FcW {
    const float* w[3];   // row-major (c_out, c_in), BN folded
    const float* b[3];
    int c_in[3], c_out[3], relu[3];
    int n;
};

struct PointHeadW {
    const float* w1;     // first layer [C1/32][KS][64]
    const float* b1;
    const float* b2;
    const float* b3;
    const float* b4;
    const f32x4* stream; // conv2 | conv3 | conv4 fragment blocks, out-tile major, back to back
    FcW fc;
};

// ---- 16-bit (bf16 / fp16) packed heads: fp32 first layer, biases and FC; MFMA weights as one stream of
// 1-KiB fragments (512 16-bit elements) per kernel, cut into ring segments (dal3_lp.h)
#define LP_ENC_SEG 32     // fragments per ring segment
#define LP_ENC_SEGS 9     // conv2|conv3|conv4 , then conv5 four out-tiles per segment
#define LP_DEC_SEG 40
#define LP_DEC_SEGS 12    // [conv2, dconv1a(0)] , 8 x [1a(2i+1), 2(2i), 1a(2i+2), 2(2i+1)] , dconv3 x2 , [dconv4, dconv5]
#define LP_HEAD_SEG 32
#ifndef DAL3_LP_ENC_T
#define DAL3_LP_ENC_T 4
#endif
#ifndef DAL3_LP_DEC_T
#define DAL3_LP_DEC_T 2
#endif
#ifndef DAL3_LP_HEAD_T
#define DAL3_LP_HEAD_T 2
#endif
struct InsSegLpW {
    const float* w1;          // fp32 first layer [2][2][64]
    const float* b1;
    const float* bias_enc;    // b2 64 | b3 64 | b4 128 | b5 1024
    const float* bias_dec;    // b2 64 | db2 256 | db3 128 | db4 128 | dw5 (2,128) | db5 (32)
    const float* dw1g;        // fp32 row-major (512,1024)
    const float* db1;         // 512
    const uint16_t* enc_stream;
    const uint16_t* dec_stream;
};
struct PointHeadLpW {
    const float* w1;
    const float* b1;
    const float* bias;        // b2 C2 | b3 C3 | b4 512
    const uint16_t* stream;
    FcW fc;
};
// ---- "f16x3" (dal3_pointmlp_x3.hip): fp16 MFMAs on (hi, lo) split operands, fp32 accuracy. Same blob shape as the 16-bit
// heads (fp32 first layer, biases, FC; one fragment stream), the stream holding a (hi, lo) fragment pair per k-step and
// padded to whole ring segments per group of points.
#ifndef DAL3_X3_HEAD_T
#define DAL3_X3_HEAD_T 2
#endif
#ifndef DAL3_X3_ENC_T
#define DAL3_X3_ENC_T 2
#endif
#ifndef DAL3_X3_DEC_T
#define DAL3_X3_DEC_T 2
#endif
typedef PointHeadLpW PointHeadX3W;
struct InsSegX3W {
    const float* w1;          // fp32 first layer [2][2][64]
    const float* b1;
    const float* bias_enc;    // b2 64 | b3 64 | b4 128 | b5 1024
    const float* bias_dec;    // b2 64 | db2 256 | db3 128 | db4 128 | db5 (32: rows 0, 1 real)
    const float* dw1g;        // fp32 row-major (512,1024): the per-crop part of dconv1
    const float* db1;         // 512
    const uint16_t* enc_stream;   // 18 segments of 32 fragments
    const uint16_t* dec_stream;   // 27 segments
};
size_t ins_seg_x3_packed_bytes();
InsSegX3W ins_seg_x3_view(const void* base);
hipError_t launch_ins_seg_encode_x3(const InsSegX3W& w, BCN pts, int c_in, int B, int N, float* g, hipStream_t s);
hipError_t launch_ins_seg_decode_x3(const InsSegX3W& w, BCN pts, int c_in, int B, int N, const float* gbias, float* logits,
                                    uint8_t* mask, hipStream_t s);
size_t point_head_x3_packed_bytes(int head_kind);
PointHeadX3W point_head_x3_view(const void* base, int head_kind);
int point_head_x3_segments(int head_kind);
hipError_t launch_point_head_x3(int head_kind, const PointHeadX3W& w, BCN x, int c_in, int B, int M, float* feat,
                                const int32_t* distinct, hipStream_t s);
// (hi, lo) fragment pairs of a layer, out-tile major: [mt][kt][k-step][hi | lo][64 lanes][8]; element order as
// launch_pack_weight_lp; kt_major: [kt][mt][...] (a K-major layer consumed chunk by chunk)
hipError_t launch_pack_weight_x3(const dal3_layer& L, int kt_major, int col_off, int n_cols, int mt_n, int kt_n, uint16_t* out,
                                 hipStream_t s, int grp_blocks = 0, int64_t grp_a0 = 0, int64_t grp_a1 = 0,
                                 int64_t grp_stride = 0, int64_t grp_last = -1);
size_t ins_seg_lp_packed_bytes();
InsSegLpW ins_seg_lp_view(const void* base);
size_t point_head_lp_packed_bytes(int head_kind);
PointHeadLpW point_head_lp_view(const void* base, int head_kind);
int point_head_lp_segments(int head_kind);
hipError_t launch_ins_seg_encode_lp(int dtype, const InsSegLpW& w, BCN pts, int c_in, int B, int N, float* g, hipStream_t s);
hipError_t launch_ins_seg_decode_lp(int dtype, const InsSegLpW& w, BCN pts, int c_in, int B, int N, const float* gbias,
                                    float* logits, uint8_t* mask, hipStream_t s);
hipError_t launch_point_head_lp(int dtype, int head_kind, const PointHeadLpW& w, BCN x, int c_in, int B, int M,
                                float* feat, const int32_t* distinct, hipStream_t s);
// 16-bit fragment packing: element j of lane l of fragment (mt,kt,s) = W'[32mt + (l&31)][32kt + 16s + 8(j>>2) + 4(l>>5) + (j&3)]
hipError_t launch_pack_weight_lp(const dal3_layer& L, int dtype, int kt_major, int col_off, int n_cols, int mt_n, int kt_n,
                                 uint16_t* out, hipStream_t s, int grp_blocks = 0, int64_t grp_a0 = 0, int64_t grp_a1 = 0,
                                 int64_t grp_stride = 0);

// layout (offsets in floats from the blob start); all sections 256-byte aligned
size_t ins_seg_packed_floats(int c_in);
InsSegW ins_seg_view(const float* base, int c_in);
size_t point_head_packed_floats(int head_kind);
PointHeadW point_head_view(const float* base, int head_kind);
size_t fc_head_packed_floats();
FcW fc_head_view(const float* base);
void point_head_dims(int head_kind, int* c_in, int* ks, int c[4], int* n_fc, int fc_in[3], int fc_out[3]);

// ---- launchers (all asynchronous on s)
hipError_t launch_ins_seg_encode(const InsSegW& w, BCN pts, int c_in, int B, int N, float* g, hipStream_t s);
hipError_t launch_ins_seg_decode(const InsSegW& w, BCN pts, int c_in, int B, int N, const float* gbias,
                                 float* logits, uint8_t* mask, hipStream_t s);
// distinct (B) i32 or NULL: only the first distinct[b] points of item b are distinct (the rest duplicate them)
// feat (B,512) need NOT be initialised: the launchers zero it (NaN rows for items with non-finite inputs) themselves.
// worklist (optional, point_head_worklist_bytes(B, M) bytes of device scratch): lets the throughput family run as
// persistent waves over the compacted list of live tiles (dal3_pointmlp.hip); NULL -> one workgroup per (item, tile)
size_t point_head_worklist_bytes(int B, int M);
hipError_t launch_point_head(int head_kind, const PointHeadW& w, BCN x, int c_in, int B, int M, float* feat,
                             const int32_t* distinct, hipStream_t s, void* worklist = nullptr, size_t worklist_bytes = 0);
hipError_t launch_generic_layer(const f32x4* wf, const float* w1, const float* bias, int kt_n, int ks_n, int mt_n,
                                int relu, BCN x, int c_in, int B, int N, float* y, hipStream_t s);

// y[b, :c_out] = relu?(W x[b] + bias); x row stride xs, y row stride ys (floats)
hipError_t launch_fc(const float* W, const float* bias, const float* x, int64_t xs, float* y, int64_t ys, int B,
                     int c_in, int c_out, int relu, hipStream_t s);

enum { PACK_FRAG_MT_MAJOR = 0, PACK_FRAG_KT_MAJOR = 1, PACK_FIRST = 2, PACK_ROWMAJOR = 3 };
// grp_blocks > 0: the (mt,kt) blocks are written in groups of grp_blocks; group 0 at out + grp_a0, group g >= 1 at
// out + grp_a1 + (g-1)*grp_stride (floats). grp_blocks == 0: dense.
hipError_t launch_pack_weight(const dal3_layer& L, int mode, int col_off, int n_cols, int mt_n, int kt_n, float* out,
                              hipStream_t s, int grp_blocks = 0, int64_t grp_a0 = 0, int64_t grp_a1 = 0,
                              int64_t grp_stride = 0);
hipError_t launch_pack_bias(const dal3_layer& L, float* out, hipStream_t s);

hipError_t launch_static_crop_prep(const double* points, const int64_t* offsets, const int32_t* choice, const double* pose,
                                   const double* box, int B, int N, uint64_t seed, int64_t item_offset, float* pts_out,
                                   float* init_box_out, hipStream_t s);
hipError_t launch_dynamic_item_prep(const double* points, const int64_t* frame_offsets, const double* boxes,
                                    const int64_t* track_first, const int32_t* item_track, const int32_t* item_frame,
                                    const int32_t* choice, const double* pose, int B, int n_per, int r, int s_,
                                    uint64_t seed, int64_t item_offset, float* pts_out, float* box_out,
                                    float* init_box_out, hipStream_t st);
hipError_t launch_static_crop_labels(const double* points, const int64_t* offsets, const int32_t* choice, const double* pose,
                                     int B, int N, uint64_t seed, int64_t item_offset, const double* gt_planes,
                                     uint8_t* mask_label, hipStream_t s);
hipError_t launch_dynamic_item_labels(const double* points, const int64_t* frame_offsets, const int64_t* track_first,
                                      const int32_t* item_track, const int32_t* item_frame, const int32_t* choice,
                                      const double* pose, int B, int n_per, int r, uint64_t seed, int64_t item_offset,
                                      const double* xform, const double* planes, const uint8_t* valid, uint8_t* mask_label,
                                      hipStream_t st);
// dal3_crops.hip
size_t crop_workspace_bytes(int64_t K_total, int64_t max_points_per_frame);
hipError_t launch_crop_count(const float* points, const int64_t* point_offsets, const double* planes,
                             const float* spheres, const int64_t* box_offsets, int F, int64_t K_total, int64_t max_points_per_frame,
                             int64_t* counts, int32_t* cc, hipStream_t s);
hipError_t launch_crop_fill(const float* points, const int64_t* point_offsets, const double* planes,
                            const float* spheres, const int64_t* box_offsets, int F, int64_t K_total,
                            int64_t max_points_per_frame, const double* pose, const int64_t* counts,
                            const int64_t* box_start, const int32_t* cc, double* out_points, int32_t* out_index,
                            int64_t out_capacity, hipStream_t s);
hipError_t launch_crop_starts(const int64_t* counts, const int64_t* order, int64_t K, int64_t* box_start, int64_t* out_offsets,
                              int64_t out_capacity, hipStream_t s);
hipError_t launch_points_in_boxes(const void* points, int points_f64, int64_t P, int64_t stride, const double* planes,
                                  int K, int f32_math, uint8_t* inside, hipStream_t s);
hipError_t launch_writeback(const double* final_boxes, const int32_t* final_idx, const double* pose_best,
                            const double* pose_inv, const double* track_box, float* det, const int64_t* det_start,
                            const int32_t* det_count, const uint8_t* active, int P, int64_t n_det, int32_t* match,
                            int32_t* owner, hipStream_t s);
hipError_t launch_maxpool_n(const void* x, int dtype, int64_t rows, int64_t n, void* out, hipStream_t s);
hipError_t launch_segment_counts(const uint8_t* mask, int B, int N, int32_t* counts, hipStream_t s);
// p[0..n_words) = value (32-bit words) as a kernel launch; used instead of hipMemsetAsync wherever the call may be
// captured into a hipGraph (see dal3_misc.hip)
hipError_t launch_fill_words(void* p, size_t n_words, uint32_t value, hipStream_t s);
// dst (B, C) = 0, or the quiet-NaN pattern in the rows of items whose input x (B, c_in, n_pts) holds a NaN / Inf
// distinct / worklist (optional): also build the point heads' worklist (ctl at worklist, entries at worklist + 256)
hipError_t launch_nonfinite_rows(BCN x, int B, int n_pts, int c_in, float* dst, int C, hipStream_t s,
                                 const int32_t* distinct = nullptr, void* worklist = nullptr);
hipError_t launch_compact_sample(const uint8_t* mask, BCN pts, int B, int N, int C, int M, int sampler,
                                 const int32_t* choice, uint64_t seed, int64_t item_offset, int32_t* counts,
                                 int32_t* pos, int32_t* obj_idx, float* obj_pts, hipStream_t s, const int64_t* step = nullptr);
// arguments of the box decode (dal3_decode_boxes of include/dal3.h), also taken by the fused last-FC + decode launch
struct DecodeArgs {
    float* box_pred;
    const float* center_add;
    int64_t ca_stride;
    int center_inplace;
    const float* boxes_center_add;
    int64_t bca_stride;
    const float* yaw_base;
    int64_t yaw_stride;
    float* heading_residuals;
    float* size_residuals;
    float* center;
    float* boxes7;
};
// box_pred (B,39) = W (39, c_in) x + bias, then the decode of those rows, one launch; box_pred = d.box_pred
hipError_t launch_fc39_decode(const float* W, const float* bias, const float* x, int64_t xs, int B, int c_in,
                              const DecodeArgs& d, hipStream_t s);
hipError_t launch_decode_boxes(float* box_pred, int B, const float* center_add, int64_t center_add_stride,
                               int center_inplace, const float* boxes_center_add, int64_t boxes_center_add_stride,
                               const float* yaw_base, int64_t yaw_stride, float* heading_residuals,
                               float* size_residuals, float* center, float* boxes7, hipStream_t s);
hipError_t launch_recenter(const float* obj_pts, int B, int M, const float* init_box7, const float* box_one7,
                           const float* bbox_gt7, float* obj_pts_two, int64_t* hcl, float* hrl, hipStream_t s);

// dal3_train.hip: training-mode building blocks over point-major (M x C) fp32 activations
hipError_t launch_tr_linear(const float* a, int64_t M, int c_in, int64_t lda, const float* scale, const float* shift,
                            int relu_in, const float* W, int64_t ldw, int transpose_w, const float* bias, int64_t seg,
                            int c_out, float* z, int64_t ldz, int accumulate, float* ws, hipStream_t s, bool prepacked = false);
size_t tr_linear_workspace_bytes(int c_in, int c_out);
// the fragment layout (output tiles per wave) launch_tr_linear will read for this shape, 0: it packs nothing
int tr_linear_pack_mtb(int64_t M, int c_in, int64_t seg, int c_out, int accumulate, int has_act);
hipError_t launch_tr_pack_many(const dal3_tr_pack_item* items, int n, hipStream_t s);
// the training forward's big layers on the f16x3 engine (dal3_train_x3.hip): layout code 0x100 | MTB, or 0 when the call
// does not qualify (then it takes launch_tr_linear)
int tr_linear_x3_layout(int64_t M, int c_in, int64_t seg, int c_out, int accumulate, int has_act);
hipError_t launch_tr_linear_x3(const float* a, int64_t M, int c_in, int64_t lda, const float* scale, const float* shift, int relu_in,
                               const uint16_t* wpk, int layout, const float* bias, int64_t seg, int c_out, float* z, int64_t ldz,
                               hipStream_t s, const uint32_t* in_amax = nullptr);
size_t tr_colred_workspace_bytes(int64_t M, int C);
hipError_t launch_tr_act_colsum(const float* x, int64_t M, int C, int64_t ldx, const float* scale, const float* shift, int relu,
                                float* out, int64_t ldo, double* part, double* sums, hipStream_t s);
hipError_t launch_tr_colred(const float* z, int64_t M, int C, int64_t ldz, int mode, const float* da, int64_t ldda,
                            const float* dg, const int32_t* arg, int64_t seg, const float* scale, const float* shift,
                            const float* mu, const float* rstd, double* part, double* out, hipStream_t s);
hipError_t launch_tr_pool_coef(const float* dg, const float* g, const float* zarg, const float* mu, const float* rstd,
                               const float* gamma, int B, int C, int64_t M, double* coef, float* kd, hipStream_t s);
hipError_t launch_tr_pool_moments(const float* W, int64_t ldw, const float* b, const double* m1, const float* Sc, int64_t M, int C,
                                  int K, double* sums, hipStream_t s);
size_t tr_pool_gv_workspace_bytes(int K);
hipError_t launch_tr_pool_gv(const double* coef, const float* W, int64_t ldw, const float* b, int C, int K, float* G, float* v,
                             double* ws, hipStream_t s);
hipError_t launch_tr_pool_dw(const double* coef, const float* W, int64_t ldw, const float* b, const float* S, const double* m1,
                             int64_t M, int centred, const float* dWs, int C, int K, float* dW, hipStream_t s);
// the 128 -> 2 logits layer with its Dropout as VALU kernels (dal3_train.hip)
hipError_t launch_tr_head2_forward(const float* z, int64_t M, int64_t ldz, const float* scale, const float* shift, int relu,
                                   const float* mult, int64_t ldm, uint64_t seed, const int64_t* step, float p_drop, const float* W,
                                   int64_t ldw, const float* bias, float* logits, hipStream_t s);
hipError_t launch_tr_head2_dgrad(const float* dl, int64_t M, const float* mult, int64_t ldm, uint64_t seed, const int64_t* step,
                                 float p_drop, const float* W, int64_t ldw, float* da, int64_t ldda, hipStream_t s);
hipError_t launch_tr_head2_dgrad_bnbwd(const float* dl, int64_t M, const float* mult, int64_t ldm, uint64_t seed, const int64_t* step,
                                       float p_drop, const float* W, int64_t ldw, float* da, int64_t ldda, const float* bz,
                                       int64_t ldbz, const float* bscale, const float* bshift, const float* bmu, const float* brstd,
                                       const float* gamma, float* dgamma, float* dbeta, float* k1, float* k2, float* k3, double* part,
                                       hipStream_t s);
size_t tr_head2_wgrad_workspace_bytes(int64_t M);
hipError_t launch_tr_head2_wgrad(const float* dl, const float* z, int64_t M, int64_t ldz, const float* scale, const float* shift,
                                 int relu, const float* mult, int64_t ldm, uint64_t seed, const int64_t* step, float p_drop,
                                 double* ws, float* dWb, hipStream_t s);
// the first layer of a stack (c_in <= 8) as VALU kernels (dal3_train.hip)
hipError_t launch_tr_wgrad_final_many(const dal3_tr_wgrad_part* items, int n, hipStream_t s);
hipError_t launch_parse_box_pred(const float* bp, int64_t ldb, int B, float* c, float* hs, float* hrn, float* hr, float* ss, float* srn,
                                 float* sr, hipStream_t s);
hipError_t launch_parse_box_pred_backward(const float* gc, const float* ghs, const float* ghrn, const float* ghr, const float* gss,
                                          const float* gsrn, const float* gsr, int B, float* g, hipStream_t s);
// dal3_train_fc.hip
int tr_fc_max_rows();
int tr_fc_max_act_cin();
hipError_t launch_tr_fc_forward(const float* a, int B, int c_in, int64_t lda, const float* in_scale, const float* in_shift, int relu_in,
                                const float* W, int64_t ldw, int transpose_w, const float* bias, int c_out, float* z, int64_t ldz,
                                const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum,
                                float eps, float* mu, float* rstd, float* scale, float* shift, hipStream_t s);
hipError_t launch_tr_fc_backward_w(const float* da, int64_t ldda, int B, int c_out, const float* z, int64_t ldz, const float* scale,
                                   const float* shift, const float* mu, const float* rstd, const float* gamma, float* dgamma,
                                   float* dbeta, const float* a_prev, int c_in, int64_t lda, const float* in_scale,
                                   const float* in_shift, int relu_in, float* dz, int64_t lddz, float* dW, int64_t lddw, float* db,
                                   hipStream_t s);
size_t tr_conv1_workspace_bytes(int64_t Mp, int c_out);
hipError_t launch_tr_conv1_bn_stats(const float* x, int64_t M, int64_t Mp, int c_in, int64_t ldx, const float* W, int64_t ldw,
                                    const float* bias, int c_out, float* z, int64_t ldz, const float* gamma, const float* beta,
                                    float* running_mean, float* running_var, float momentum, float eps, float* mu, float* rstd,
                                    float* scale, float* shift, double* part, hipStream_t s);
hipError_t launch_tr_conv1_wgrad(const float* dz, int64_t lddz, const float* x, int64_t M, int c_in, int64_t ldx, int c_out, double* part,
                                 float* dW, hipStream_t s);
hipError_t launch_tr_gather_at(const float* z, int64_t ldz, const int32_t* arg, int64_t seg, int n_seg, int C, float* out, hipStream_t s);
hipError_t launch_tr_pool_zarg(const int32_t* arg, const float* a, int64_t lda, const float* W, int64_t ldw, const float* bias, int B,
                               int C, int K, int N, float* zarg, hipStream_t s);
hipError_t launch_tr_pool_sparse(const int32_t* arg, const float* kd, const float* W, int64_t ldw, const float* a, int64_t lda,
                                 int B, int C, int K, int N, float* da, int64_t ldda, float* dWs, hipStream_t s);
hipError_t launch_tr_box_loss(const float* center, const float* center_label, const float* hs, const float* hrn,
                              const int64_t* hcl, const float* hrl, const float* ss, const float* srn, const int64_t* scl,
                              const float* srl, int B, float* losses, float* g_center, float* g_hs, float* g_hrn, float* g_ss,
                              float* g_srn, hipStream_t s);
size_t tr_seg_ce_workspace_bytes(int64_t M);
hipError_t launch_tr_seg_ce(const float* logits, const void* labels, int labels_i64, int64_t M, float* loss, float* dlogits,
                            double* part, hipStream_t s);
// linear + a column reduction of its output in the kernel's epilogue (dal3_train.hip, TrRed)
size_t tr_linear_red_workspace_bytes();
hipError_t launch_tr_linear_bn_stats(const float* a, int64_t M, int c_in, int64_t lda, const float* scale, const float* shift,
                                     int relu_in, const float* W, int64_t ldw, const float* bias, int64_t seg, int c_out,
                                     float* z, int64_t ldz, float* packed, int64_t rows, const float* gamma, const float* beta,
                                     float* running_mean, float* running_var, float momentum, float eps, float* mu, float* rstd,
                                     float* bn_scale, float* bn_shift, double* part, hipStream_t s, int* fused);
hipError_t launch_tr_linear_bnbwd_sums(const float* a, int64_t M, int c_in, int64_t lda, const float* W, int64_t ldw, int c_out,
                                       float* da, int64_t ldda, float* packed, int64_t rows, const float* bz, int64_t ldbz,
                                       const float* bscale, const float* bshift, const float* bmu, const float* brstd,
                                       const float* gamma, float* dgamma, float* dbeta, float* k1, float* k2, float* k3,
                                       double* part, hipStream_t s, int* fused);
hipError_t launch_tr_bn_stats(const float* z, int64_t M, int C, int64_t ldz, const float* gamma, const float* beta,
                              float* running_mean, float* running_var, float momentum, float eps, float* mu, float* rstd,
                              float* scale, float* shift, double* part, hipStream_t s);
hipError_t launch_tr_bnbwd_sums(const float* z, int64_t M, int C, int64_t ldz, const float* da, int64_t ldda, const float* dg,
                                const int32_t* arg, int64_t seg, const float* scale, const float* shift, const float* mu,
                                const float* rstd, const float* gamma, float* dgamma, float* dbeta, float* k1, float* k2,
                                float* k3, double* part, hipStream_t s);
hipError_t launch_tr_bn_finalize(const double* sums, int C, int64_t M, const float* gamma, const float* beta,
                                 float* running_mean, float* running_var, float momentum, float eps, float* mu, float* rstd,
                                 float* scale, float* shift, hipStream_t s);
hipError_t launch_tr_bnbwd_coef(const double* sums, int C, int64_t M, const float* gamma, const float* rstd, float* dgamma,
                                float* dbeta, float* k1, float* k2, float* k3, hipStream_t s);
hipError_t launch_tr_bnbwd_apply(const float* z, int64_t M, int C, int64_t ldz, const float* da, int64_t ldda,
                                 const float* dg, const int32_t* arg, int64_t seg, const float* scale, const float* shift,
                                 const float* mu, const float* rstd, const float* k1, const float* k2, const float* k3,
                                 float* dz, int64_t lddz, hipStream_t s, uint32_t* amax = nullptr);
size_t tr_bnbwd_apply_segsum_workspace_bytes(int64_t M, int C);
hipError_t launch_tr_bnbwd_apply_segsum(const float* z, int64_t M, int C, int64_t ldz, const float* da, int64_t ldda,
                                        const float* scale, const float* shift, const float* mu, const float* rstd,
                                        const float* k1, const float* k2, const float* k3, float* dz, int64_t lddz,
                                        int64_t sum_seg, float* seg_sums, double* ws, hipStream_t s);
size_t tr_wgrad_workspace_bytes(int64_t M, int c_out, int c_in);
hipError_t launch_tr_wgrad(const float* dz, int64_t lddz, const float* a, int64_t lda, const float* scale,
                           const float* shift, int relu_in, int64_t M, int c_out, int c_in, float* part, float* dW,
                           hipStream_t s);
hipError_t launch_tr_linear_pool(const float* a, int64_t M, int c_in, int64_t lda, const float* scale, const float* shift,
                                 int relu_in, const float* W, int64_t ldw, const float* bias, const float* out_scale,
                                 const float* out_shift, int64_t seg, int c_out, float* g, int32_t* arg, float* ws,
                                 unsigned long long* packed, hipStream_t s);
// wgrad on the f16x3 engine (dal3_train_x3.hip): ok -> the layer's shape qualifies; the workspace holds its slices' partial sums
bool tr_wgrad_x3_ok(int64_t M, int c_out, int c_in);
size_t tr_wgrad_x3_workspace_bytes(int64_t M, int c_out, int c_in);
hipError_t launch_tr_wgrad_x3(const float* dz, int64_t lddz, const float* a, int64_t lda, const float* scale, const float* shift,
                              int relu_in, const uint32_t* dz_amax, int64_t M, int c_out, int c_in, float* part, float* dW,
                              hipStream_t s);
hipError_t launch_tr_segmax_unpack(const unsigned long long* packed, int64_t total, float* g, int32_t* arg, hipStream_t s);
bool tr_linear_pool_x3_ok(int64_t M, int c_in, int64_t seg, int c_out);
hipError_t launch_tr_linear_pool_x3(const float* a, int64_t M, int c_in, int64_t lda, const float* scale, const float* shift,
                                    int relu_in, const float* W, int64_t ldw, const float* bias, const float* out_scale,
                                    const float* out_shift, int64_t seg, int c_out, float* g, int32_t* arg, float* ws,
                                    unsigned long long* packed, hipStream_t s);
// A persistent kernel's workgroup w walks the groups w, w + G, w + 2G, ... Workgroup ids are dealt round-robin over the 8
// XCDs (an L2 each), so consecutive groups — the groups of ONE crop, which share the crop's 2-KiB dconv1 term — land on
// different XCDs and each L2 fetches the term for itself. With this virtual id XCD x (ids = x mod 8) owns the
// contiguous virtual blocks [x G/8, (x+1) G/8): a crop's groups of one round share an L2 (round 6; a bijection on [0, G),
// so every group is still walked exactly once; G not a multiple of 8: the plain id).
#if defined(__HIPCC__)
__device__ __forceinline__ int xcd_contiguous_block() {
    const unsigned g = gridDim.x, b = blockIdx.x;
    return (g & 7u) ? (int)b : (int)((b & 7u) * (g >> 3) + (b >> 3));
}
#endif
// small jobs: one 16-wave workgroup per 32-point tile, activations through LDS (dal3_latency.hip); bit-identical results
hipError_t launch_ins_seg_encode_lat(const InsSegW& w, BCN pts, int c_in, int B, int N, float* g, hipStream_t s);
hipError_t launch_ins_seg_decode_lat(const InsSegW& w, BCN pts, int c_in, int B, int N, const float* gbias, float* logits,
                                     uint8_t* mask, hipStream_t s);
hipError_t launch_point_head_lat(int head_kind, const PointHeadW& w, BCN x, int c_in, int B, int M, float* feat,
                                 const int32_t* distinct, hipStream_t s);
bool lat_use(int64_t tiles, int flags, int64_t max_tiles);    // the dispatch rule (dal3_pointmlp.hip)
hipError_t launch_tr_act_dropout(const float* x, int64_t M, int C, int64_t ldx, const float* scale, const float* shift, int relu,
                                 const float* mult, int64_t ldm, uint64_t seed, const int64_t* step, float p_drop, float* out,
                                 int64_t ldo, hipStream_t s);
hipError_t launch_tr_segmax(const float* z, int64_t ldz, int64_t seg, int C, const float* scale, const float* shift,
                            float* g, int32_t* arg, int64_t n_seg, unsigned long long* packed, hipStream_t s);
hipError_t launch_tr_segsum(const float* x, int64_t ldx, int64_t seg, int C, float* out, int64_t n_seg, hipStream_t s);

