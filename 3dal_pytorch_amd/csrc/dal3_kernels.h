// dal3_kernels.h — host-visible views of the packed weights and the kernel launchers.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "../../include/dal3.h"
#include "dal3_device.h"

// point tiles (of 32 points) per wave; see DESIGN.md "register budget"
#ifndef DAL3_ENC_T
#define DAL3_ENC_T 2
#endif
#ifndef DAL3_DEC_T
#define DAL3_DEC_T 1
#endif
#ifndef DAL3_HEAD_T
#define DAL3_HEAD_T 1
#endif

// ---- packed PointNetInstanceSeg (BN folded). "frag" = [MT][KT][4][64] float4 fragment order of
// dal3_device.h::mma_block; dw2 is K-major ([KT][MT][4][64]) because dconv2 consumes dconv1's
// output chunk by chunk.
struct InsSegW {
    const float* w1;     // first layer [2][2][64]
    const float* b1;     // 64
    const f32x4* w2;     // frag 64->64
    const float* b2;
    const f32x4* w3;     // frag 64->64
    const float* b3;
    const f32x4* w4;     // frag 64->128
    const float* b4;
    const f32x4* w5;     // frag 128->1024
    const float* b5;
    const f32x4* dw1a;   // frag 64->512: dconv1 columns 0..63 (the out2 part)
    const float* dw1g;   // row-major (512,1024): dconv1 columns 64..1087 (the global-feature part)
    const float* db1;    // 512
    const f32x4* dw2;    // frag 512->256, K-major
    const float* db2;
    const f32x4* dw3;    // frag 256->128
    const float* db3;
    const f32x4* dw4;    // frag 128->128
    const float* db4;
    const float* dw5;    // row-major (2,128)
    const float* db5;    // 2
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
    const f32x4* w2;
    const float* b2;
    const f32x4* w3;
    const float* b3;
    const f32x4* w4;     // frag C3->512
    const float* b4;
    FcW fc;
};

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
hipError_t launch_point_head(int head_kind, const PointHeadW& w, BCN x, int c_in, int B, int M, float* feat, hipStream_t s);
hipError_t launch_generic_layer(const f32x4* wf, const float* w1, const float* bias, int kt_n, int ks_n, int mt_n,
                                int relu, BCN x, int c_in, int B, int N, float* y, hipStream_t s);

// y[b, :c_out] = relu?(W x[b] + bias); x row stride xs, y row stride ys (floats)
hipError_t launch_fc(const float* W, const float* bias, const float* x, int64_t xs, float* y, int64_t ys, int B,
                     int c_in, int c_out, int relu, hipStream_t s);

enum { PACK_FRAG_MT_MAJOR = 0, PACK_FRAG_KT_MAJOR = 1, PACK_FIRST = 2, PACK_ROWMAJOR = 3 };
hipError_t launch_pack_weight(const dal3_layer& L, int mode, int col_off, int n_cols, int mt_n, int kt_n, float* out,
                              hipStream_t s);
hipError_t launch_pack_bias(const dal3_layer& L, float* out, hipStream_t s);

hipError_t launch_maxpool_n(const float* x, int64_t rows, int64_t n, float* out, hipStream_t s);
hipError_t launch_segment_counts(const uint8_t* mask, int B, int N, int32_t* counts, hipStream_t s);
hipError_t launch_compact_sample(const uint8_t* mask, BCN pts, int B, int N, int C, int M, int sampler,
                                 const int32_t* choice, uint64_t seed, int64_t item_offset, int32_t* counts,
                                 int32_t* pos, int32_t* obj_idx, float* obj_pts, hipStream_t s);
hipError_t launch_decode_boxes(float* box_pred, int B, const float* center_add, int64_t center_add_stride,
                               int center_inplace, const float* boxes_center_add, int64_t boxes_center_add_stride,
                               const float* yaw_base, int64_t yaw_stride, float* heading_residuals,
                               float* size_residuals, float* center, float* boxes7, hipStream_t s);
hipError_t launch_recenter(const float* obj_pts, int B, int M, const float* init_box7, const float* box_one7,
                           const float* bbox_gt7, float* obj_pts_two, int64_t* hcl, float* hrl, hipStream_t s);
