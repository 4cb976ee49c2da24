// dal3_api.hip — the C ABI of include/dal3.h: packed-weight layout, argument checks, sequencing of
// the kernels on the caller's stream. Host code only; no allocation, no synchronisation.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "dal3_kernels.h"
#include "dal3_lp.h"

// ---------------------------------------------------------------------------------- errors
static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                           \
    do {                                                                                        \
        hipError_t e_ = (expr);                                                                 \
        if (e_ != hipSuccess) return fail(DAL3_EHIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)
#define TRY(expr)               \
    do {                        \
        int r_ = (expr);        \
        if (r_ != 0) return r_; \
    } while (0)

extern "C" int dal3_version(void) { return DAL3_VERSION; }
extern "C" const float* dal3_mean_size(void) {
    static const float v[9] = {DAL3_MEAN_SIZE_VALUES};
    return v;
}
extern "C" const char* dal3_last_error(void) { return g_err; }

// ---------------------------------------------------------------------------------- layout
static inline size_t al(size_t floats) { return (floats + 63) & ~(size_t)63; }   // 256-byte sections

struct Cursor {
    const float* base;
    size_t off;
    const float* take(size_t floats) {
        const float* p = base ? base + off : nullptr;
        off += al(floats);
        return p;
    }
    const f32x4* take4(size_t floats) { return reinterpret_cast<const f32x4*>(take(floats)); }
};

static InsSegW ins_seg_walk(Cursor& c) {
    InsSegW w;
    w.w1 = c.take(2 * 2 * 64);
    w.b1 = c.take(64);    w.b2 = c.take(64);    w.b3 = c.take(64);    w.b4 = c.take(128);   w.b5 = c.take(1024);
    w.dw1g = c.take(512 * 1024);
    w.db1 = c.take(512);  w.db2 = c.take(256);  w.db3 = c.take(128);  w.db4 = c.take(128);
    w.dw5 = c.take(2 * 128);
    w.db5 = c.take(32);
    w.enc_stream = c.take4((size_t)ENC_FRAGS * 256);
    w.dec_stream = c.take4((size_t)DEC_FRAGS * 256);
    w.lat_stream = c.take4((size_t)LAT_FRAGS * 256);
    return w;
}

size_t ins_seg_packed_floats(int) {
    Cursor c{nullptr, 0};
    ins_seg_walk(c);
    return c.off + DAL3_BLOB_TAIL_FLOATS;
}
InsSegW ins_seg_view(const float* base, int) {
    Cursor c{base, 0};
    return ins_seg_walk(c);
}

void point_head_dims(int head_kind, int* c_in, int* ks, int c[4], int* n_fc, int fc_in[3], int fc_out[3]) {
    switch (head_kind) {
        case DAL3_HEAD_STATIC_BOX_EST:
            *c_in = 3; *ks = 2; c[0] = 128; c[1] = 128; c[2] = 256; c[3] = 512; *n_fc = 3;
            fc_in[0] = 512; fc_out[0] = 512; fc_in[1] = 512; fc_out[1] = 256; fc_in[2] = 256; fc_out[2] = 39;
            break;
        case DAL3_HEAD_POINT_EMB:
            *c_in = 4; *ks = 2; c[0] = 64; c[1] = 128; c[2] = 256; c[3] = 512; *n_fc = 2;
            fc_in[0] = 512; fc_out[0] = 512; fc_in[1] = 512; fc_out[1] = 256; fc_in[2] = 0; fc_out[2] = 0;
            break;
        default:  // DAL3_HEAD_BOX_EMB
            *c_in = 8; *ks = 4; c[0] = 64; c[1] = 64; c[2] = 128; c[3] = 512; *n_fc = 2;
            fc_in[0] = 512; fc_out[0] = 128; fc_in[1] = 128; fc_out[1] = 128; fc_in[2] = 0; fc_out[2] = 0;
            break;
    }
}

static PointHeadW point_head_walk(Cursor& cur, int head_kind) {
    int c_in, ks, c[4], n_fc, fi[3], fo[3];
    point_head_dims(head_kind, &c_in, &ks, c, &n_fc, fi, fo);
    PointHeadW w;
    w.w1 = cur.take((size_t)(c[0] / 32) * ks * 64);
    w.b1 = cur.take(c[0]);    w.b2 = cur.take(c[1]);    w.b3 = cur.take(c[2]);    w.b4 = cur.take(c[3]);
    w.stream = cur.take4(((size_t)(c[1] / 32) * (c[0] / 32) + (size_t)(c[2] / 32) * (c[1] / 32) +
                          (size_t)(c[3] / 32) * (c[2] / 32)) * 1024);
    w.fc.n = n_fc;
    for (int i = 0; i < 3; ++i) {
        w.fc.c_in[i] = fi[i];
        w.fc.c_out[i] = fo[i];
        w.fc.relu[i] = !(head_kind == DAL3_HEAD_STATIC_BOX_EST && i == 2);
        w.fc.w[i] = i < n_fc ? cur.take((size_t)fi[i] * fo[i]) : nullptr;
        w.fc.b[i] = i < n_fc ? cur.take((fo[i] + 31) / 32 * 32) : nullptr;
    }
    return w;
}
size_t point_head_packed_floats(int head_kind) {
    Cursor c{nullptr, 0};
    point_head_walk(c, head_kind);
    return c.off + DAL3_BLOB_TAIL_FLOATS;
}
PointHeadW point_head_view(const float* base, int head_kind) {
    Cursor c{base, 0};
    return point_head_walk(c, head_kind);
}

static const int kDynFcIn[3] = {384, 128, 128}, kDynFcOut[3] = {128, 128, 39};
static FcW fc_head_walk(Cursor& cur) {
    FcW f;
    f.n = 3;
    for (int i = 0; i < 3; ++i) {
        f.c_in[i] = kDynFcIn[i];
        f.c_out[i] = kDynFcOut[i];
        f.relu[i] = i < 2;
        f.w[i] = cur.take((size_t)kDynFcIn[i] * kDynFcOut[i]);
        f.b[i] = cur.take((kDynFcOut[i] + 31) / 32 * 32);
    }
    return f;
}
size_t fc_head_packed_floats() {
    Cursor c{nullptr, 0};
    fc_head_walk(c);
    return c.off;
}
FcW fc_head_view(const float* base) {
    Cursor c{base, 0};
    return fc_head_walk(c);
}

// ---- 16-bit heads: fp32 sections first (float cursor), then the fragment streams (bytes)
static InsSegLpW ins_seg_lp_walk(Cursor& c) {
    InsSegLpW w;
    w.w1 = c.take(2 * 2 * 64);
    w.b1 = c.take(64);
    w.bias_enc = c.take(1280);
    w.bias_dec = c.take(864);
    w.dw1g = c.take(512 * 1024);
    w.db1 = c.take(512);
    w.enc_stream = reinterpret_cast<const uint16_t*>(c.take((size_t)LP_ENC_SEGS * LP_ENC_SEG * 256));
    w.dec_stream = reinterpret_cast<const uint16_t*>(c.take((size_t)LP_DEC_SEGS * LP_DEC_SEG * 256));
    return w;
}
size_t ins_seg_lp_packed_bytes() {
    Cursor c{nullptr, 0};
    ins_seg_lp_walk(c);
    return c.off * sizeof(float);
}
InsSegLpW ins_seg_lp_view(const void* base) {
    Cursor c{static_cast<const float*>(base), 0};
    return ins_seg_lp_walk(c);
}
static int lp_tps(int kt, int mt) { return lp_tiles_per_seg(kt, mt); }
int point_head_lp_segments(int head_kind) {
    int c_in, ks, c[4], n_fc, fi[3], fo[3];
    point_head_dims(head_kind, &c_in, &ks, c, &n_fc, fi, fo);
    int n = 0;
    for (int l = 1; l < 4; ++l) n += (c[l] / 32) / lp_tps(c[l - 1] / 32, c[l] / 32);
    return n;
}
static PointHeadLpW point_head_lp_walk(Cursor& cur, int head_kind) {
    int c_in, ks, c[4], n_fc, fi[3], fo[3];
    point_head_dims(head_kind, &c_in, &ks, c, &n_fc, fi, fo);
    PointHeadLpW w;
    w.w1 = cur.take((size_t)(c[0] / 32) * ks * 64);
    w.b1 = cur.take(c[0]);
    w.bias = cur.take(c[1] + c[2] + c[3]);
    w.fc.n = n_fc;
    for (int i = 0; i < 3; ++i) {
        w.fc.c_in[i] = fi[i];
        w.fc.c_out[i] = fo[i];
        w.fc.relu[i] = !(head_kind == DAL3_HEAD_STATIC_BOX_EST && i == 2);
        w.fc.w[i] = i < n_fc ? cur.take((size_t)fi[i] * fo[i]) : nullptr;
        w.fc.b[i] = i < n_fc ? cur.take((fo[i] + 31) / 32 * 32) : nullptr;
    }
    w.stream = reinterpret_cast<const uint16_t*>(cur.take((size_t)point_head_lp_segments(head_kind) * LP_HEAD_SEG * 256));
    return w;
}
size_t point_head_lp_packed_bytes(int head_kind) {
    Cursor c{nullptr, 0};
    point_head_lp_walk(c, head_kind);
    return c.off * sizeof(float);
}
PointHeadLpW point_head_lp_view(const void* base, int head_kind) {
    Cursor c{static_cast<const float*>(base), 0};
    return point_head_lp_walk(c, head_kind);
}

// ---- "f16x3" ins_seg: fp32 sections, then the two fragment streams (18 / 27 segments of 32 KiB)
static InsSegX3W ins_seg_x3_walk(Cursor& c) {
    InsSegX3W w;
    w.w1 = c.take(2 * 2 * 64);
    w.b1 = c.take(64);
    w.bias_enc = c.take(1280);
    w.bias_dec = c.take(608);
    w.dw1g = c.take(512 * 1024);
    w.db1 = c.take(512);
    w.enc_stream = reinterpret_cast<const uint16_t*>(c.take((size_t)18 * 32 * 256));
    w.dec_stream = reinterpret_cast<const uint16_t*>(c.take((size_t)27 * 32 * 256));
    return w;
}
size_t ins_seg_x3_packed_bytes() {
    Cursor c{nullptr, 0};
    ins_seg_x3_walk(c);
    return c.off * sizeof(float) + DAL3_BLOB_TAIL_FLOATS * sizeof(float);
}
InsSegX3W ins_seg_x3_view(const void* base) {
    Cursor c{static_cast<const float*>(base), 0};
    return ins_seg_x3_walk(c);
}

// ---- "f16x3" heads: the 16-bit heads' blob with a stream of (hi, lo) fragment pairs, padded to whole ring segments
static PointHeadX3W point_head_x3_walk(Cursor& cur, int head_kind) {
    int c_in, ks, c[4], n_fc, fi[3], fo[3];
    point_head_dims(head_kind, &c_in, &ks, c, &n_fc, fi, fo);
    PointHeadX3W w;
    w.w1 = cur.take((size_t)(c[0] / 32) * ks * 64);
    w.b1 = cur.take(c[0]);
    w.bias = cur.take(c[1] + c[2] + c[3]);
    w.fc.n = n_fc;
    for (int i = 0; i < 3; ++i) {
        w.fc.c_in[i] = fi[i];
        w.fc.c_out[i] = fo[i];
        w.fc.relu[i] = !(head_kind == DAL3_HEAD_STATIC_BOX_EST && i == 2);
        w.fc.w[i] = i < n_fc ? cur.take((size_t)fi[i] * fo[i]) : nullptr;
        w.fc.b[i] = i < n_fc ? cur.take((fo[i] + 31) / 32 * 32) : nullptr;
    }
    w.stream = reinterpret_cast<const uint16_t*>(cur.take((size_t)point_head_x3_segments(head_kind) * 32 * 256));   // 32 fragments of 256 floats per segment
    return w;
}
size_t point_head_x3_packed_bytes(int head_kind) {
    Cursor c{nullptr, 0};
    point_head_x3_walk(c, head_kind);
    return c.off * sizeof(float) + DAL3_BLOB_TAIL_FLOATS * sizeof(float);
}
PointHeadX3W point_head_x3_view(const void* base, int head_kind) {
    Cursor c{static_cast<const float*>(base), 0};
    return point_head_x3_walk(c, head_kind);
}

// ---------------------------------------------------------------------------------- packing
static float* mut(const void* p) { return const_cast<float*>(reinterpret_cast<const float*>(p)); }

static int check_layer(const dal3_layer& L, int c_in, int c_out, const char* what) {
    if (!L.weight || !L.bias) return fail(DAL3_EINVAL, "%s: null weight/bias", what);
    if (L.c_in != c_in || L.c_out != c_out)
        return fail(DAL3_EINVAL, "%s: expected (%d -> %d), got (%d -> %d)", what, c_in, c_out, L.c_in, L.c_out);
    const bool any = L.bn_weight || L.bn_bias || L.bn_mean || L.bn_var;
    const bool all = L.bn_weight && L.bn_bias && L.bn_mean && L.bn_var;
    if (any && !all) return fail(DAL3_EINVAL, "%s: BN pointers must be all set or all NULL", what);
    return 0;
}

// fragment blocks of one layer written at fragment offset `frag` of a stream (+ its folded bias)
static int pack_frag(const dal3_layer& L, int mode, int col_off, int n_cols, const f32x4* stream, int frag,
                     const float* b, hipStream_t s, int grp_blocks = 0, int64_t a0 = 0, int64_t a1 = 0,
                     int64_t stride = 0) {
    HIP_TRY(launch_pack_weight(L, mode, col_off, n_cols, L.c_out / 32, n_cols / 32, mut(stream) + (size_t)frag * 256, s,
                               grp_blocks, a0, a1, stride));
    if (b) HIP_TRY(launch_pack_bias(L, mut(b), s));
    return 0;
}

static int pack_fc(const dal3_layer* L, const FcW& f, hipStream_t s) {
    for (int i = 0; i < f.n; ++i) {
        TRY(check_layer(L[i], f.c_in[i], f.c_out[i], "fc layer"));
        HIP_TRY(launch_pack_weight(L[i], PACK_ROWMAJOR, 0, f.c_in[i], 0, 0, mut(f.w[i]), s));
        HIP_TRY(launch_pack_bias(L[i], mut(f.b[i]), s));
    }
    return 0;
}

extern "C" int dal3_pack_weights(int head_kind, const dal3_layer* L, int n_layers, int dtype, void* packed_dev,
                                 size_t* bytes_inout, dal3_stream stream) {
    if (dtype != DAL3_F32 && dtype != DAL3_BF16 && dtype != DAL3_F16 && dtype != DAL3_F16X3)
        return fail(DAL3_EINVAL, "dal3_pack_weights: unknown dtype %d", dtype);
    if (!bytes_inout) return fail(DAL3_EINVAL, "dal3_pack_weights: bytes_inout is NULL");
    const bool x3 = dtype == DAL3_F16X3;
    const bool lp = dtype != DAL3_F32 && !x3;
    size_t need;
    switch (head_kind) {
        case DAL3_HEAD_INS_SEG:
            need = x3 ? ins_seg_x3_packed_bytes() : lp ? ins_seg_lp_packed_bytes() : ins_seg_packed_floats(0) * sizeof(float);
            break;
        case DAL3_HEAD_STATIC_BOX_EST:
        case DAL3_HEAD_POINT_EMB:
        case DAL3_HEAD_BOX_EMB:
            need = x3 ? point_head_x3_packed_bytes(head_kind)
                      : lp ? point_head_lp_packed_bytes(head_kind) : point_head_packed_floats(head_kind) * sizeof(float);
            break;
        case DAL3_HEAD_DYNAMIC_BOX_EST: need = fc_head_packed_floats() * sizeof(float); break;   // FC only: fp32 in every dtype
        default: return fail(DAL3_EINVAL, "dal3_pack_weights: unknown head_kind %d", head_kind);
    }
    if (!packed_dev) {
        *bytes_inout = need;
        return 0;
    }
    if (*bytes_inout < need) return fail(DAL3_EWORKSPACE, "dal3_pack_weights: need %zu bytes, got %zu", need, *bytes_inout);
    if (!L) return fail(DAL3_EINVAL, "dal3_pack_weights: layers is NULL");
    if (reinterpret_cast<uintptr_t>(packed_dev) & 255) return fail(DAL3_EINVAL, "packed_dev must be 256-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const float* base = static_cast<const float*>(packed_dev);
    HIP_TRY(hipMemsetAsync(packed_dev, 0, need, s));

    if (head_kind == DAL3_HEAD_INS_SEG) {
        if (n_layers != 10) return fail(DAL3_EINVAL, "ins_seg expects 10 layers, got %d", n_layers);
        const int c_in = L[0].c_in;
        if (c_in != 3 && c_in != 4) return fail(DAL3_EINVAL, "ins_seg c_in must be 3 or 4, got %d", c_in);
        static const int co[10] = {64, 64, 64, 128, 1024, 512, 256, 128, 128, 2};
        static const int ci[10] = {0, 64, 64, 64, 128, 1088, 512, 256, 128, 128};
        for (int i = 0; i < 10; ++i) TRY(check_layer(L[i], i == 0 ? c_in : ci[i], co[i], "ins_seg layer"));
        if (x3) {
            const InsSegX3W w = ins_seg_x3_view(packed_dev);
            uint16_t* enc = const_cast<uint16_t*>(w.enc_stream);
            uint16_t* dec = const_cast<uint16_t*>(w.dec_stream);
            float* be = mut(w.bias_enc);
            float* bd = mut(w.bias_dec);
            const int64_t F = 512;                                    // 16-bit elements per fragment
            HIP_TRY(launch_pack_weight(L[0], PACK_FIRST, 0, c_in, 2, 2, mut(w.w1), s));
            HIP_TRY(launch_pack_bias(L[0], mut(w.b1), s));
            HIP_TRY(launch_pack_bias(L[1], be, s));
            HIP_TRY(launch_pack_bias(L[2], be + 64, s));
            HIP_TRY(launch_pack_bias(L[3], be + 128, s));
            HIP_TRY(launch_pack_bias(L[4], be + 256, s));
            HIP_TRY(launch_pack_bias(L[1], bd, s));
            HIP_TRY(launch_pack_bias(L[6], bd + 64, s));
            HIP_TRY(launch_pack_bias(L[7], bd + 320, s));
            HIP_TRY(launch_pack_bias(L[8], bd + 448, s));
            HIP_TRY(launch_pack_bias(L[9], bd + 576, s));             // db5: 2 real rows, padded to 32
            HIP_TRY(launch_pack_weight(L[5], PACK_ROWMAJOR, 64, 1024, 0, 0, mut(w.dw1g), s));
            HIP_TRY(launch_pack_bias(L[5], mut(w.db1), s));
            // encode stream (fragments): conv2 @0 (16) | conv3 @16 (16) | conv4 @32 (32) | conv5 @64 (512)
            HIP_TRY(launch_pack_weight_x3(L[1], 0, 0, 64, 2, 2, enc, s));
            HIP_TRY(launch_pack_weight_x3(L[2], 0, 0, 64, 2, 2, enc + 16 * F, s));
            HIP_TRY(launch_pack_weight_x3(L[3], 0, 0, 64, 4, 2, enc + 32 * F, s));
            HIP_TRY(launch_pack_weight_x3(L[4], 0, 0, 128, 32, 4, enc + 64 * F, s));
            // decode stream (chunks software-pipelined, dal3_pointmlp_x3.hip): conv2 @0 (16) | A_0 @16 (8) |
            // 15 x { A_{c+1} @24+40c (8), D_c @32+40c (32) } | D_15 @624 (32) | dconv3 @656 (128) | dconv4 @784 (64) |
            // dconv5 @848 (16: one out-tile, rows 0, 1 real)
            HIP_TRY(launch_pack_weight_x3(L[1], 0, 0, 64, 2, 2, dec, s));
            HIP_TRY(launch_pack_weight_x3(L[5], 0, 0, 64, 16, 2, dec, s, 2, 16 * F, 24 * F, 40 * F));
            HIP_TRY(launch_pack_weight_x3(L[6], 1, 0, 512, 8, 16, dec, s, 8, 32 * F, 72 * F, 40 * F, 624 * F));
            HIP_TRY(launch_pack_weight_x3(L[7], 0, 0, 256, 4, 8, dec + 656 * F, s));
            HIP_TRY(launch_pack_weight_x3(L[8], 0, 0, 128, 4, 4, dec + 784 * F, s));
            HIP_TRY(launch_pack_weight_x3(L[9], 0, 0, 128, 1, 4, dec + 848 * F, s));
            return 0;
        }
        if (lp) {
            const InsSegLpW w = ins_seg_lp_view(packed_dev);
            uint16_t* enc = const_cast<uint16_t*>(w.enc_stream);
            uint16_t* dec = const_cast<uint16_t*>(w.dec_stream);
            float* be = mut(w.bias_enc);
            float* bd = mut(w.bias_dec);
            const int64_t F = 512;                                    // 16-bit elements per fragment
            HIP_TRY(launch_pack_weight(L[0], PACK_FIRST, 0, c_in, 2, 2, mut(w.w1), s));
            HIP_TRY(launch_pack_bias(L[0], mut(w.b1), s));
            HIP_TRY(launch_pack_bias(L[1], be, s));
            HIP_TRY(launch_pack_bias(L[2], be + 64, s));
            HIP_TRY(launch_pack_bias(L[3], be + 128, s));
            HIP_TRY(launch_pack_bias(L[4], be + 256, s));
            HIP_TRY(launch_pack_bias(L[1], bd, s));
            HIP_TRY(launch_pack_bias(L[6], bd + 64, s));
            HIP_TRY(launch_pack_bias(L[7], bd + 320, s));
            HIP_TRY(launch_pack_bias(L[8], bd + 448, s));
            HIP_TRY(launch_pack_weight(L[9], PACK_ROWMAJOR, 0, 128, 0, 0, bd + 576, s));
            HIP_TRY(launch_pack_bias(L[9], bd + 832, s));
            HIP_TRY(launch_pack_weight(L[5], PACK_ROWMAJOR, 64, 1024, 0, 0, mut(w.dw1g), s));
            HIP_TRY(launch_pack_bias(L[5], mut(w.db1), s));
            // encode stream (fragments): conv2 @0 | conv3 @8 | conv4 @16 | conv5 @32
            HIP_TRY(launch_pack_weight_lp(L[1], dtype, 0, 0, 64, 2, 2, enc, s));
            HIP_TRY(launch_pack_weight_lp(L[2], dtype, 0, 0, 64, 2, 2, enc + 8 * F, s));
            HIP_TRY(launch_pack_weight_lp(L[3], dtype, 0, 0, 64, 4, 2, enc + 16 * F, s));
            HIP_TRY(launch_pack_weight_lp(L[4], dtype, 0, 0, 128, 32, 4, enc + 32 * F, s));
            // decode stream, segments of 40: [conv2 @0, 1a(0) @8] ; 1a(c>=1) @40+20(c-1) ; 2(c) @44+20c ; dconv3 @360,@400 ; dconv4 @440, dconv5 @472
            HIP_TRY(launch_pack_weight_lp(L[1], dtype, 0, 0, 64, 2, 2, dec, s));
            HIP_TRY(launch_pack_weight_lp(L[5], dtype, 0, 0, 64, 16, 2, dec, s, 2, 8 * F, 40 * F, 20 * F));
            HIP_TRY(launch_pack_weight_lp(L[6], dtype, 1, 0, 512, 8, 16, dec, s, 8, 44 * F, 64 * F, 20 * F));
            HIP_TRY(launch_pack_weight_lp(L[7], dtype, 0, 0, 256, 4, 8, dec, s, 16, 360 * F, 400 * F, 40 * F));
            HIP_TRY(launch_pack_weight_lp(L[8], dtype, 0, 0, 128, 4, 4, dec + 440 * F, s));
            HIP_TRY(launch_pack_weight_lp(L[9], dtype, 0, 0, 128, 1, 4, dec + 472 * F, s));   // dconv5: rows 0, 1 of one out-tile
            return 0;
        }
        InsSegW w = ins_seg_view(base, c_in);
        HIP_TRY(launch_pack_weight(L[0], PACK_FIRST, 0, c_in, 2, 2, mut(w.w1), s));
        HIP_TRY(launch_pack_bias(L[0], mut(w.b1), s));
        // encode stream: conv2 | conv3 | conv4 | conv5
        TRY(pack_frag(L[1], PACK_FRAG_MT_MAJOR, 0, 64, w.enc_stream, ENC_W2, w.b2, s));
        TRY(pack_frag(L[2], PACK_FRAG_MT_MAJOR, 0, 64, w.enc_stream, ENC_W3, w.b3, s));
        TRY(pack_frag(L[3], PACK_FRAG_MT_MAJOR, 0, 64, w.enc_stream, ENC_W4, w.b4, s));
        TRY(pack_frag(L[4], PACK_FRAG_MT_MAJOR, 0, 128, w.enc_stream, ENC_W5, w.b5, s));
        // decode stream: conv2 | dconv1a(0) | { dconv1a(c+1), dconv2(c) } | dconv3 | dconv4 (fragments of 256 floats)
        TRY(pack_frag(L[1], PACK_FRAG_MT_MAJOR, 0, 64, w.dec_stream, DEC_W2, nullptr, s));
        TRY(pack_frag(L[5], PACK_FRAG_MT_MAJOR, 0, 64, w.dec_stream, DEC_MIX, w.db1, s, /*chunk = 2 blocks*/ 2,
                      0, 8 * 256, 40 * 256));                                              // columns of out2
        HIP_TRY(launch_pack_weight(L[5], PACK_ROWMAJOR, 64, 1024, 0, 0, mut(w.dw1g), s));   // columns of g
        TRY(pack_frag(L[6], PACK_FRAG_KT_MAJOR, 0, 512, w.dec_stream, DEC_MIX, w.db2, s, /*chunk = 8 blocks*/ 8,
                      16 * 256, 56 * 256, 40 * 256));
        TRY(pack_frag(L[7], PACK_FRAG_MT_MAJOR, 0, 256, w.dec_stream, DEC_W3, w.db3, s));
        TRY(pack_frag(L[8], PACK_FRAG_MT_MAJOR, 0, 128, w.dec_stream, DEC_W4, w.db4, s));
        // the same dconv1a / dconv2 out-tile major, for the latency kernels
        TRY(pack_frag(L[5], PACK_FRAG_MT_MAJOR, 0, 64, w.lat_stream, LAT_W1A, nullptr, s));
        TRY(pack_frag(L[6], PACK_FRAG_MT_MAJOR, 0, 512, w.lat_stream, LAT_W2, nullptr, s));
        HIP_TRY(launch_pack_weight(L[9], PACK_ROWMAJOR, 0, 128, 0, 0, mut(w.dw5), s));
        HIP_TRY(launch_pack_bias(L[9], mut(w.db5), s));
        return 0;
    }
    if (head_kind == DAL3_HEAD_DYNAMIC_BOX_EST) {
        if (n_layers != 3) return fail(DAL3_EINVAL, "dynamic box_est expects 3 layers, got %d", n_layers);
        return pack_fc(L, fc_head_view(base), s);
    }
    int c_in, ks, c[4], n_fc, fi[3], fo[3];
    point_head_dims(head_kind, &c_in, &ks, c, &n_fc, fi, fo);
    if (n_layers != 4 + n_fc) return fail(DAL3_EINVAL, "head %d expects %d layers, got %d", head_kind, 4 + n_fc, n_layers);
    TRY(check_layer(L[0], c_in, c[0], "conv1"));
    for (int i = 1; i < 4; ++i) TRY(check_layer(L[i], c[i - 1], c[i], "conv"));
    if (x3) {
        const PointHeadX3W w = point_head_x3_view(packed_dev, head_kind);
        HIP_TRY(launch_pack_weight(L[0], PACK_FIRST, 0, c_in, c[0] / 32, ks, mut(w.w1), s));
        HIP_TRY(launch_pack_bias(L[0], mut(w.b1), s));
        HIP_TRY(launch_pack_bias(L[1], mut(w.bias), s));
        HIP_TRY(launch_pack_bias(L[2], mut(w.bias) + c[1], s));
        HIP_TRY(launch_pack_bias(L[3], mut(w.bias) + c[1] + c[2], s));
        uint16_t* st = const_cast<uint16_t*>(w.stream);          // conv2 | conv3 | conv4, out-tile major, back to back
        int64_t off = 0;
        for (int l = 1; l < 4; ++l) {
            const int kt = c[l - 1] / 32, mt = c[l] / 32;
            HIP_TRY(launch_pack_weight_x3(L[l], 0, 0, c[l - 1], mt, kt, st + off, s));
            off += (int64_t)mt * kt * 2048;
        }
        return pack_fc(L + 4, w.fc, s);
    }
    if (lp) {
        const PointHeadLpW w = point_head_lp_view(packed_dev, head_kind);
        HIP_TRY(launch_pack_weight(L[0], PACK_FIRST, 0, c_in, c[0] / 32, ks, mut(w.w1), s));
        HIP_TRY(launch_pack_bias(L[0], mut(w.b1), s));
        HIP_TRY(launch_pack_bias(L[1], mut(w.bias), s));
        HIP_TRY(launch_pack_bias(L[2], mut(w.bias) + c[1], s));
        HIP_TRY(launch_pack_bias(L[3], mut(w.bias) + c[1] + c[2], s));
        uint16_t* st = const_cast<uint16_t*>(w.stream);
        const int64_t SEGE = (int64_t)LP_HEAD_SEG * 512;           // elements per ring segment
        int seg = 0;
        for (int l = 1; l < 4; ++l) {                              // each layer: TPS out-tiles per segment
            const int kt = c[l - 1] / 32, mt = c[l] / 32, tps = lp_tps(kt, mt);
            HIP_TRY(launch_pack_weight_lp(L[l], dtype, 0, 0, c[l - 1], mt, kt, st, s, tps * kt, seg * SEGE,
                                          (seg + 1) * SEGE, SEGE));
            seg += mt / tps;
        }
        return pack_fc(L + 4, w.fc, s);
    }
    PointHeadW w = point_head_view(base, head_kind);
    HIP_TRY(launch_pack_weight(L[0], PACK_FIRST, 0, c_in, c[0] / 32, ks, mut(w.w1), s));
    HIP_TRY(launch_pack_bias(L[0], mut(w.b1), s));
    const int f3 = (c[1] / 32) * (c[0] / 32) * 4, f4 = f3 + (c[2] / 32) * (c[1] / 32) * 4;   // fragment offsets
    TRY(pack_frag(L[1], PACK_FRAG_MT_MAJOR, 0, c[0], w.stream, 0, w.b2, s));
    TRY(pack_frag(L[2], PACK_FRAG_MT_MAJOR, 0, c[1], w.stream, f3, w.b3, s));
    TRY(pack_frag(L[3], PACK_FRAG_MT_MAJOR, 0, c[2], w.stream, f4, w.b4, s));
    return pack_fc(L + 4, w.fc, s);
}

// ---------------------------------------------------------------------------------- workspace carving
struct Carver {
    char* base;
    size_t size, off;
    bool ok;
    Carver(void* p, size_t n) : base(static_cast<char*>(p)), size(n), off(0), ok(true) {}
    template <typename T>
    T* take(size_t count) {
        const size_t bytes = (count * sizeof(T) + 255) & ~(size_t)255;
        char* p = base ? base + off : nullptr;
        off += bytes;
        if (base && off > size) ok = false;
        return reinterpret_cast<T*>(p);
    }
};

static BCN to_bcn(const dal3_bcn& t) { return BCN{t.data, t.stride_b, t.stride_c, t.stride_n, t.dtype, t.flags}; }

static int check_bcn(const dal3_bcn& t, const char* what) {
    if (!t.data) return fail(DAL3_EINVAL, "%s: null data pointer", what);
    if (t.stride_n <= 0 || t.stride_c <= 0) return fail(DAL3_EINVAL, "%s: strides must be positive", what);
    if (t.dtype != DAL3_F32 && t.dtype != DAL3_BF16 && t.dtype != DAL3_F16)
        return fail(DAL3_EINVAL, "%s: storage dtype %d is none of DAL3_F32 / DAL3_BF16 / DAL3_F16", what, t.dtype);
    if (t.dtype != DAL3_F32 && (reinterpret_cast<uintptr_t>(t.data) & 1)) return fail(DAL3_EINVAL, "%s: 16-bit data must be 2-byte aligned", what);
    if (t.flags & ~(DAL3_BCN_NO_SMALL_JOB_KERNELS | DAL3_BCN_NO_WORKLIST | DAL3_BCN_NO_LDS_SAMPLER)) return fail(DAL3_EINVAL, "%s: unknown bits in flags (%d)", what, t.flags);
    return 0;
}

// Upper bounds of a batch (dal3.h: DAL3_MAX_ITEMS, DAL3_MAX_POINTS_PER_ITEM, DAL3_MAX_TILES). The kernels index an item's
// points and the per-item outputs with 32-bit ints and launch one workgroup (or worklist entry) per 32-point tile of an
// item: a job beyond these would get a truncated grid or a wrapped offset, so it is refused here, before
// any workspace is carved or anything is launched.
static int check_extent(int64_t B, int64_t N, const char* what) {
    if (B <= 0 || N <= 0) return fail(DAL3_EINVAL, "%s: B and N must be positive (B=%lld N=%lld)", what, (long long)B, (long long)N);
    if (B > DAL3_MAX_ITEMS) return fail(DAL3_EINVAL, "%s: B=%lld items exceed DAL3_MAX_ITEMS (%d)", what, (long long)B, DAL3_MAX_ITEMS);
    if (N > DAL3_MAX_POINTS_PER_ITEM)
        return fail(DAL3_EINVAL, "%s: N=%lld points per item exceed DAL3_MAX_POINTS_PER_ITEM (%d)", what, (long long)N, DAL3_MAX_POINTS_PER_ITEM);
    if (B * ((N + 31) / 32) > (int64_t)DAL3_MAX_TILES)
        return fail(DAL3_EINVAL, "%s: B=%lld x N=%lld is %lld 32-point tiles, more than DAL3_MAX_TILES (%d): split the batch", what,
                    (long long)B, (long long)N, (long long)(B * ((N + 31) / 32)), DAL3_MAX_TILES);
    return 0;
}

// ---------------------------------------------------------------------------------- ins_seg
struct InsSegWs { float* g; float* gb; };
static InsSegWs carve_ins_seg(Carver& c, int B) {
    InsSegWs w;
    w.g = c.take<float>((size_t)B * 1024);
    w.gb = c.take<float>((size_t)B * 512);
    return w;
}
extern "C" size_t dal3_ins_seg_workspace_bytes(int B) {
    Carver c(nullptr, 0);
    carve_ins_seg(c, B);
    return c.off;
}

static int check_dtype(int dtype) {
    if (dtype != DAL3_F32 && dtype != DAL3_BF16 && dtype != DAL3_F16 && dtype != DAL3_F16X3)
        return fail(DAL3_EINVAL, "unknown dtype %d", dtype);
    return 0;
}

static int ins_seg_run(const void* packed, int dtype, int c_in, const dal3_bcn& pts, int B, int N, float* logits,
                       uint8_t* mask, float* global_feat_out, const InsSegWs& ws, hipStream_t s) {
    TRY(check_dtype(dtype));
    if (!packed || !logits || !mask) return fail(DAL3_EINVAL, "ins_seg: null pointer");
    TRY(check_extent(B, N, "ins_seg"));
    if (c_in != 3 && c_in != 4) return fail(DAL3_EINVAL, "ins_seg: c_in must be 3 or 4");
    TRY(check_bcn(pts, "pts"));
    const BCN x = to_bcn(pts);
    HIP_TRY(launch_nonfinite_rows(x, B, N, c_in, ws.g, 1024, s));      // g = 0 (NaN for a crop with a non-finite coordinate)
    if (dtype == DAL3_F32) {
        const InsSegW w = ins_seg_view(static_cast<const float*>(packed), c_in);
        HIP_TRY(launch_ins_seg_encode(w, x, c_in, B, N, ws.g, s));
        // per-crop part of dconv1: gb = W1g' . g + b1'   (static_model.py:286-289 without the repeat+cat)
        HIP_TRY(launch_fc(w.dw1g, w.db1, ws.g, 1024, ws.gb, 512, B, 1024, 512, 0, s));
        HIP_TRY(launch_ins_seg_decode(w, x, c_in, B, N, ws.gb, logits, mask, s));
    } else if (dtype == DAL3_F16X3) {
        const InsSegX3W w = ins_seg_x3_view(packed);
        HIP_TRY(launch_ins_seg_encode_x3(w, x, c_in, B, N, ws.g, s));
        HIP_TRY(launch_fc(w.dw1g, w.db1, ws.g, 1024, ws.gb, 512, B, 1024, 512, 0, s));
        HIP_TRY(launch_ins_seg_decode_x3(w, x, c_in, B, N, ws.gb, logits, mask, s));
    } else {
        const InsSegLpW w = ins_seg_lp_view(packed);
        HIP_TRY(launch_ins_seg_encode_lp(dtype, w, x, c_in, B, N, ws.g, s));
        HIP_TRY(launch_fc(w.dw1g, w.db1, ws.g, 1024, ws.gb, 512, B, 1024, 512, 0, s));
        HIP_TRY(launch_ins_seg_decode_lp(dtype, w, x, c_in, B, N, ws.gb, logits, mask, s));
    }
    if (global_feat_out)
        HIP_TRY(hipMemcpyAsync(global_feat_out, ws.g, (size_t)B * 1024 * sizeof(float), hipMemcpyDeviceToDevice, s));
    return 0;
}

extern "C" int dal3_ins_seg_forward(const void* packed, int dtype, int c_in, dal3_bcn pts, int B, int N, float* logits,
                                    uint8_t* mask, float* global_feat_out, void* workspace, size_t workspace_bytes,
                                    dal3_stream stream) {
    if (!workspace) return fail(DAL3_EINVAL, "ins_seg: workspace is NULL");
    TRY(check_extent(B, N, "ins_seg"));
    Carver c(workspace, workspace_bytes);
    const InsSegWs ws = carve_ins_seg(c, B);
    if (!c.ok) return fail(DAL3_EWORKSPACE, "ins_seg: workspace needs %zu bytes, got %zu", c.off, workspace_bytes);
    return ins_seg_run(packed, dtype, c_in, pts, B, N, logits, mask, global_feat_out, ws, static_cast<hipStream_t>(stream));
}

extern "C" int dal3_ins_seg_encode(const void* packed, int dtype, int c_in, dal3_bcn pts, int B, int N,
                                   float* global_feat, dal3_stream stream) {
    TRY(check_dtype(dtype));
    if (!packed || !global_feat || (c_in != 3 && c_in != 4)) return fail(DAL3_EINVAL, "ins_seg_encode: bad argument");
    TRY(check_extent(B, N, "ins_seg_encode"));
    TRY(check_bcn(pts, "pts"));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == DAL3_F32)
        HIP_TRY(launch_ins_seg_encode(ins_seg_view(static_cast<const float*>(packed), c_in), to_bcn(pts), c_in, B, N,
                                      global_feat, s));
    else if (dtype == DAL3_F16X3)
        HIP_TRY(launch_ins_seg_encode_x3(ins_seg_x3_view(packed), to_bcn(pts), c_in, B, N, global_feat, s));
    else
        HIP_TRY(launch_ins_seg_encode_lp(dtype, ins_seg_lp_view(packed), to_bcn(pts), c_in, B, N, global_feat, s));
    return 0;
}

extern "C" int dal3_ins_seg_global_bias(const void* packed, int dtype, const float* global_feat, int B, float* gbias,
                                        dal3_stream stream) {
    TRY(check_dtype(dtype));
    if (!packed || !global_feat || !gbias) return fail(DAL3_EINVAL, "ins_seg_global_bias: bad argument");
    TRY(check_extent(B, 1, "ins_seg_global_bias"));
    const float* dw1g = dtype == DAL3_F32 ? ins_seg_view(static_cast<const float*>(packed), 3).dw1g
                        : dtype == DAL3_F16X3 ? ins_seg_x3_view(packed).dw1g : ins_seg_lp_view(packed).dw1g;
    const float* db1 = dtype == DAL3_F32 ? ins_seg_view(static_cast<const float*>(packed), 3).db1
                       : dtype == DAL3_F16X3 ? ins_seg_x3_view(packed).db1 : ins_seg_lp_view(packed).db1;
    HIP_TRY(launch_fc(dw1g, db1, global_feat, 1024, gbias, 512, B, 1024, 512, 0, static_cast<hipStream_t>(stream)));
    return 0;
}

extern "C" int dal3_ins_seg_decode(const void* packed, int dtype, int c_in, dal3_bcn pts, int B, int N,
                                   const float* gbias, float* logits, uint8_t* mask, dal3_stream stream) {
    TRY(check_dtype(dtype));
    if (!packed || !gbias || !logits || !mask || (c_in != 3 && c_in != 4)) return fail(DAL3_EINVAL, "ins_seg_decode: bad argument");
    TRY(check_extent(B, N, "ins_seg_decode"));
    TRY(check_bcn(pts, "pts"));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == DAL3_F32)
        HIP_TRY(launch_ins_seg_decode(ins_seg_view(static_cast<const float*>(packed), c_in), to_bcn(pts), c_in, B, N,
                                      gbias, logits, mask, s));
    else if (dtype == DAL3_F16X3)
        HIP_TRY(launch_ins_seg_decode_x3(ins_seg_x3_view(packed), to_bcn(pts), c_in, B, N, gbias, logits, mask, s));
    else
        HIP_TRY(launch_ins_seg_decode_lp(dtype, ins_seg_lp_view(packed), to_bcn(pts), c_in, B, N, gbias, logits, mask, s));
    return 0;
}

// ---------------------------------------------------------------------------------- gather
extern "C" size_t dal3_gather_workspace_bytes(int B, int N) {
    Carver c(nullptr, 0);
    c.take<int32_t>((size_t)B * N);
    return c.off;
}

extern "C" int dal3_segment_counts(const uint8_t* mask, int B, int N, int32_t* counts, dal3_stream stream) {
    if (!mask || !counts) return fail(DAL3_EINVAL, "segment_counts: bad argument");
    TRY(check_extent(B, N, "segment_counts"));
    HIP_TRY(launch_segment_counts(mask, B, N, counts, static_cast<hipStream_t>(stream)));
    return 0;
}

static int gather_run(const uint8_t* mask, const dal3_bcn& pts, int B, int N, int C, int M, int sampler,
                      const int32_t* choice, uint64_t seed, int64_t item_offset, int32_t* counts, int32_t* obj_idx,
                      float* obj_pts, int32_t* pos, hipStream_t s, const int64_t* step = nullptr) {
    if (!mask || !counts || !obj_idx || !obj_pts) return fail(DAL3_EINVAL, "gather: null pointer");
    if (M <= 0 || C <= 0 || C > 8) return fail(DAL3_EINVAL, "gather: bad shape");
    TRY(check_extent(B, N, "gather"));
    TRY(check_extent(B, M, "gather (object points)"));
    if (sampler == DAL3_SAMPLER_CHOICE && !choice) return fail(DAL3_EINVAL, "gather: DAL3_SAMPLER_CHOICE needs choice");
    if (sampler != DAL3_SAMPLER_CHOICE && sampler != DAL3_SAMPLER_DEVICE) return fail(DAL3_EINVAL, "gather: bad sampler");
    TRY(check_bcn(pts, "pts"));
    HIP_TRY(launch_compact_sample(mask, to_bcn(pts), B, N, C, M, sampler, choice, seed, item_offset, counts, pos,
                                  obj_idx, obj_pts, s, step));
    return 0;
}

extern "C" int dal3_mask_compact_sample(const uint8_t* mask, dal3_bcn pts, int B, int N, int C, int M, int sampler,
                                        const int32_t* choice, uint64_t seed, int64_t item_offset, int32_t* counts,
                                        int32_t* obj_idx, float* obj_pts, void* workspace, size_t workspace_bytes,
                                        dal3_stream stream) {
    if (!workspace) return fail(DAL3_EINVAL, "gather: workspace is NULL");
    TRY(check_extent(B, N, "gather"));
    Carver c(workspace, workspace_bytes);
    int32_t* pos = c.take<int32_t>((size_t)B * N);
    if (!c.ok) return fail(DAL3_EWORKSPACE, "gather: workspace needs %zu bytes, got %zu", c.off, workspace_bytes);
    return gather_run(mask, pts, B, N, C, M, sampler, choice, seed, item_offset, counts, obj_idx, obj_pts, pos,
                      static_cast<hipStream_t>(stream));
}

extern "C" int dal3_mask_compact_sample_step(const uint8_t* mask, dal3_bcn pts, int B, int N, int C, int M, uint64_t seed,
                                             const int64_t* step, int64_t item_offset, int32_t* counts, int32_t* obj_idx,
                                             float* obj_pts, void* workspace, size_t workspace_bytes, dal3_stream stream) {
    if (!workspace) return fail(DAL3_EINVAL, "gather: workspace is NULL");
    TRY(check_extent(B, N, "gather"));
    Carver c(workspace, workspace_bytes);
    int32_t* pos = c.take<int32_t>((size_t)B * N);
    if (!c.ok) return fail(DAL3_EWORKSPACE, "gather: workspace needs %zu bytes, got %zu", c.off, workspace_bytes);
    return gather_run(mask, pts, B, N, C, M, DAL3_SAMPLER_DEVICE, nullptr, seed, item_offset, counts, obj_idx, obj_pts, pos,
                      static_cast<hipStream_t>(stream), step);
}

// ---------------------------------------------------------------------------------- point heads
// `work`: the live-tile worklist of the point heads' persistent kernel (dal3_pointmlp.hip), sized for items of up to
// HEAD_WORK_MAX_POINTS points (the dynamic head's 5 x 512 object points); longer items take the per-tile kernel
#define HEAD_WORK_MAX_POINTS 2560
struct HeadWs { float* feat; float* t1; float* t2; void* work; size_t work_bytes; };
static HeadWs carve_head(Carver& c, int B) {
    HeadWs w;
    w.feat = c.take<float>((size_t)B * 512);
    w.t1 = c.take<float>((size_t)B * 512);
    w.t2 = c.take<float>((size_t)B * 512);
    w.work_bytes = point_head_worklist_bytes(B, HEAD_WORK_MAX_POINTS);
    w.work = c.take<char>(w.work_bytes);
    return w;
}
extern "C" size_t dal3_point_head_workspace_bytes(int B) {
    Carver c(nullptr, 0);
    carve_head(c, B);
    return c.off;
}

// dec (optional): the chain ends in a box estimator's 39-wide layer written to dec->box_pred (= out, row stride 39), and
// that layer is launched together with the decode of its rows (fc39_decode_kernel) instead of on its own
static int fc_chain(const FcW& f, const float* x, int64_t xs, float* out, int64_t out_stride, int B, float* t1,
                    float* t2, hipStream_t s, const DecodeArgs* dec = nullptr) {
    const float* cur = x;
    int64_t cs = xs;
    for (int i = 0; i < f.n; ++i) {
        const bool last = i == f.n - 1;
        float* dst = last ? out : (i % 2 == 0 ? t1 : t2);
        const int64_t ds = last ? out_stride : f.c_out[i];
        if (last && dec) {
            if (f.c_out[i] != 39 || out_stride != 39 || dec->box_pred != out || f.relu[i])
                return fail(DAL3_EINVAL, "fc_chain: a fused decode needs the 39-wide box_pred layer last");
            HIP_TRY(launch_fc39_decode(f.w[i], f.b[i], cur, cs, B, f.c_in[i], *dec, s));
            return 0;
        }
        HIP_TRY(launch_fc(f.w[i], f.b[i], cur, cs, dst, ds, B, f.c_in[i], f.c_out[i], f.relu[i], s));
        cur = dst;
        cs = ds;
    }
    return 0;
}

static int point_head_run(int head_kind, const void* packed, int dtype, const dal3_bcn& x, int B, int M, float* out,
                          int64_t out_stride, const HeadWs& ws, hipStream_t s, const int32_t* distinct = nullptr,
                          const DecodeArgs* dec = nullptr) {
    TRY(check_dtype(dtype));
    if (head_kind != DAL3_HEAD_STATIC_BOX_EST && head_kind != DAL3_HEAD_POINT_EMB && head_kind != DAL3_HEAD_BOX_EMB)
        return fail(DAL3_EINVAL, "point_head: head_kind %d is not a point head", head_kind);
    if (!packed || !out) return fail(DAL3_EINVAL, "point_head: null pointer");
    TRY(check_extent(B, M, "point_head"));
    TRY(check_bcn(x, "x"));
    int c_in, ks, c[4], n_fc, fi[3], fo[3];
    point_head_dims(head_kind, &c_in, &ks, c, &n_fc, fi, fo);
    if (dtype == DAL3_F16X3) {
        const PointHeadX3W w = point_head_x3_view(packed, head_kind);
        HIP_TRY(launch_point_head_x3(head_kind, w, to_bcn(x), c_in, B, M, ws.feat, distinct, s));
        return fc_chain(w.fc, ws.feat, 512, out, out_stride, B, ws.t1, ws.t2, s, dec);
    }
    if (dtype != DAL3_F32) {
        const PointHeadLpW w = point_head_lp_view(packed, head_kind);
        HIP_TRY(launch_point_head_lp(dtype, head_kind, w, to_bcn(x), c_in, B, M, ws.feat, distinct, s));
        return fc_chain(w.fc, ws.feat, 512, out, out_stride, B, ws.t1, ws.t2, s, dec);
    }
    const PointHeadW w = point_head_view(static_cast<const float*>(packed), head_kind);
    HIP_TRY(launch_point_head(head_kind, w, to_bcn(x), c_in, B, M, ws.feat, distinct, s, ws.work, ws.work_bytes));
    return fc_chain(w.fc, ws.feat, 512, out, out_stride, B, ws.t1, ws.t2, s, dec);
}

extern "C" int dal3_point_head_forward(int head_kind, const void* packed, int dtype, dal3_bcn x, int B, int M, float* out,
                                       int64_t out_stride, void* workspace, size_t workspace_bytes,
                                       dal3_stream stream) {
    if (!workspace) return fail(DAL3_EINVAL, "point_head: workspace is NULL");
    TRY(check_extent(B, M, "point_head"));
    Carver c(workspace, workspace_bytes);
    const HeadWs ws = carve_head(c, B);
    if (!c.ok) return fail(DAL3_EWORKSPACE, "point_head: workspace needs %zu bytes, got %zu", c.off, workspace_bytes);
    return point_head_run(head_kind, packed, dtype, x, B, M, out, out_stride, ws, static_cast<hipStream_t>(stream));
}

/* the per-point stack + max of a point head alone (no FC tail): feat (B,512). n_distinct: see dal3.h */
extern "C" size_t dal3_point_head_pool_workspace_bytes(int B, int M) {
    return B > 0 && M > 0 ? point_head_worklist_bytes(B, M) : 0;
}

extern "C" int dal3_point_head_pool(int head_kind, const void* packed, int dtype, dal3_bcn x, int B, int M,
                                    const int32_t* n_distinct, float* feat, void* workspace, size_t workspace_bytes,
                                    dal3_stream stream) {
    TRY(check_dtype(dtype));
    if (head_kind != DAL3_HEAD_STATIC_BOX_EST && head_kind != DAL3_HEAD_POINT_EMB && head_kind != DAL3_HEAD_BOX_EMB)
        return fail(DAL3_EINVAL, "point_head_pool: head_kind %d is not a point head", head_kind);
    if (!packed || !feat) return fail(DAL3_EINVAL, "point_head_pool: null pointer");
    if (workspace && (reinterpret_cast<uintptr_t>(workspace) & 15)) return fail(DAL3_EINVAL, "point_head_pool: workspace must be 16-byte aligned");
    TRY(check_extent(B, M, "point_head_pool"));
    TRY(check_bcn(x, "x"));
    hipStream_t s = static_cast<hipStream_t>(stream);
    int c_in, ks, c[4], n_fc, fi[3], fo[3];
    point_head_dims(head_kind, &c_in, &ks, c, &n_fc, fi, fo);
    if (dtype == DAL3_F16X3) {
        HIP_TRY(launch_point_head_x3(head_kind, point_head_x3_view(packed, head_kind), to_bcn(x), c_in, B, M, feat, n_distinct, s));
        return 0;
    }
    if (dtype != DAL3_F32) {
        const PointHeadLpW w = point_head_lp_view(packed, head_kind);
        HIP_TRY(launch_point_head_lp(dtype, head_kind, w, to_bcn(x), c_in, B, M, feat, n_distinct, s));
        return 0;
    }
    const PointHeadW w = point_head_view(static_cast<const float*>(packed), head_kind);
    HIP_TRY(launch_point_head(head_kind, w, to_bcn(x), c_in, B, M, feat, n_distinct, s, workspace, workspace_bytes));
    return 0;
}

extern "C" int dal3_dynamic_box_est_forward(const void* packed, const float* embedding, int B, float* box_pred,
                                            void* workspace, size_t workspace_bytes, dal3_stream stream) {
    if (!packed || !embedding || !box_pred || !workspace) return fail(DAL3_EINVAL, "dynamic_box_est: bad argument");
    TRY(check_extent(B, 1, "dynamic_box_est"));
    Carver c(workspace, workspace_bytes);
    const HeadWs ws = carve_head(c, B);
    if (!c.ok) return fail(DAL3_EWORKSPACE, "dynamic_box_est: workspace needs %zu bytes, got %zu", c.off, workspace_bytes);
    return fc_chain(fc_head_view(static_cast<const float*>(packed)), embedding, 384, box_pred, 39, B, ws.t1, ws.t2,
                    static_cast<hipStream_t>(stream));
}

// ---------------------------------------------------------------------------------- small entries
extern "C" int dal3_decode_boxes(float* box_pred, int B, const float* center_add, int64_t center_add_stride,
                                 int center_inplace, const float* boxes_center_add, int64_t boxes_center_add_stride,
                                 const float* yaw_base, int64_t yaw_stride, float* heading_residuals,
                                 float* size_residuals, float* center, float* boxes7, dal3_stream stream) {
    if (!box_pred || B <= 0) return fail(DAL3_EINVAL, "decode_boxes: bad argument");
    HIP_TRY(launch_decode_boxes(box_pred, B, center_add, center_add_stride, center_inplace, boxes_center_add,
                                boxes_center_add_stride, yaw_base, yaw_stride, heading_residuals, size_residuals,
                                center, boxes7, static_cast<hipStream_t>(stream)));
    return 0;
}

extern "C" int dal3_recenter_rotz(const float* obj_pts, int B, int M, const float* init_box7, const float* box_one7,
                                  const float* bbox_gt7, float* obj_pts_two, int64_t* heading_class_label,
                                  float* heading_residual_label, dal3_stream stream) {
    if (!obj_pts || !init_box7 || !box_one7 || !obj_pts_two || B <= 0 || M <= 0)
        return fail(DAL3_EINVAL, "recenter_rotz: bad argument");
    if (bbox_gt7 && (!heading_class_label || !heading_residual_label))
        return fail(DAL3_EINVAL, "recenter_rotz: bbox_gt given without label outputs");
    HIP_TRY(launch_recenter(obj_pts, B, M, init_box7, box_one7, bbox_gt7, obj_pts_two, heading_class_label,
                            heading_residual_label, static_cast<hipStream_t>(stream)));
    return 0;
}

extern "C" int dal3_static_crop_prep(const double* points, const int64_t* offsets, const int32_t* choice,
                                     const double* pose, const double* box, int B, int N, uint64_t seed,
                                     int64_t item_offset, float* pts_out, float* init_box_out, dal3_stream stream) {
    if (!points || !offsets || !pose || !box || !pts_out || !init_box_out || B <= 0 || N < 7)
        return fail(DAL3_EINVAL, "static_crop_prep: bad argument (N must be >= 7)");
    HIP_TRY(launch_static_crop_prep(points, offsets, choice, pose, box, B, N, seed, item_offset, pts_out, init_box_out,
                                    static_cast<hipStream_t>(stream)));
    return 0;
}

extern "C" int dal3_dynamic_item_prep(const double* points, const int64_t* frame_offsets, const double* boxes,
                                      const int64_t* track_first, const int32_t* item_track, const int32_t* item_frame,
                                      const int32_t* choice, const double* pose, int B, int n_per, int r, int s,
                                      uint64_t seed, int64_t item_offset, float* pts_out, float* box_out,
                                      float* init_box_out, dal3_stream stream) {
    if (!points || !frame_offsets || !boxes || !track_first || !item_track || !item_frame || !pose || !pts_out ||
        !box_out || !init_box_out || B <= 0 || n_per <= 0 || r < 0 || s < 0)
        return fail(DAL3_EINVAL, "dynamic_item_prep: bad argument");
    HIP_TRY(launch_dynamic_item_prep(points, frame_offsets, boxes, track_first, item_track, item_frame, choice, pose, B,
                                     n_per, r, s, seed, item_offset, pts_out, box_out, init_box_out,
                                     static_cast<hipStream_t>(stream)));
    return 0;
}

extern "C" int dal3_static_crop_labels(const double* points, const int64_t* offsets, const int32_t* choice,
                                       const double* pose, int B, int N, uint64_t seed, int64_t item_offset,
                                       const double* gt_planes, uint8_t* mask_label, dal3_stream stream) {
    if (!points || !offsets || !pose || !gt_planes || !mask_label || B <= 0 || N <= 0)
        return fail(DAL3_EINVAL, "static_crop_labels: bad argument");
    HIP_TRY(launch_static_crop_labels(points, offsets, choice, pose, B, N, seed, item_offset, gt_planes, mask_label,
                                      static_cast<hipStream_t>(stream)));
    return 0;
}

extern "C" int dal3_dynamic_item_labels(const double* points, const int64_t* frame_offsets, const int64_t* track_first,
                                        const int32_t* item_track, const int32_t* item_frame, const int32_t* choice,
                                        const double* pose, int B, int n_per, int r, uint64_t seed, int64_t item_offset,
                                        const double* xform, const double* planes, const uint8_t* valid,
                                        uint8_t* mask_label, dal3_stream stream) {
    if (!points || !frame_offsets || !track_first || !item_track || !item_frame || !pose || !xform || !planes || !valid ||
        !mask_label || B <= 0 || n_per <= 0 || r < 0)
        return fail(DAL3_EINVAL, "dynamic_item_labels: bad argument");
    HIP_TRY(launch_dynamic_item_labels(points, frame_offsets, track_first, item_track, item_frame, choice, pose, B, n_per, r,
                                       seed, item_offset, xform, planes, valid, mask_label,
                                       static_cast<hipStream_t>(stream)));
    return 0;
}

extern "C" int dal3_points_in_boxes(const void* points, int points_f64, int64_t P, int64_t stride, const double* planes,
                                    int K, int f32_math, uint8_t* inside, dal3_stream stream) {
    if (P < 0 || K < 0 || stride < 3 || ((P > 0 && K > 0) && (!points || !planes || !inside)))
        return fail(DAL3_EINVAL, "points_in_boxes: bad argument (stride must be >= 3)");
    if (points_f64 && f32_math) return fail(DAL3_EINVAL, "points_in_boxes: float32 arithmetic needs float32 points");
    HIP_TRY(launch_points_in_boxes(points, points_f64, P, stride, planes, K, f32_math, inside,
                                   static_cast<hipStream_t>(stream)));
    return 0;
}

extern "C" size_t dal3_crop_workspace_bytes(int64_t K_total, int64_t max_points_per_frame) {
    if (K_total <= 0 || max_points_per_frame < 0) return 0;
    return crop_workspace_bytes(K_total, max_points_per_frame);
}

static int crop_args_ok(const float* points, const int64_t* point_offsets, const double* planes, const float* spheres,
                        const int64_t* box_offsets, int F, int64_t K_total, int64_t max_pts, const void* ws,
                        size_t ws_bytes) {
    if (!point_offsets || !box_offsets || F <= 0 || F > 65535 || K_total < 0 || max_pts < 0)
        return fail(DAL3_EINVAL, "crop: bad argument (1 <= F <= 65535)");
    if (K_total > 0 && max_pts > 0 && (!points || !planes || !spheres))
        return fail(DAL3_EINVAL, "crop: null points / planes / spheres");
    if (K_total > 0 && (!ws || ws_bytes < crop_workspace_bytes(K_total, max_pts)))
        return fail(DAL3_EWORKSPACE, "crop: workspace smaller than dal3_crop_workspace_bytes()");
    return 0;
}

extern "C" int dal3_crop_count(const float* points, const int64_t* point_offsets, const double* planes,
                               const float* spheres, const int64_t* box_offsets, int F, int64_t K_total,
                               int64_t max_points_per_frame, int64_t* counts, void* workspace, size_t workspace_bytes,
                               dal3_stream stream) {
    if (int e = crop_args_ok(points, point_offsets, planes, spheres, box_offsets, F, K_total, max_points_per_frame,
                             workspace, workspace_bytes))
        return e;
    if (K_total > 0 && !counts) return fail(DAL3_EINVAL, "crop_count: null counts");
    HIP_TRY(launch_crop_count(points, point_offsets, planes, spheres, box_offsets, F, K_total, max_points_per_frame, counts,
                              static_cast<int32_t*>(workspace), static_cast<hipStream_t>(stream)));
    return 0;
}

extern "C" int dal3_crop_fill(const float* points, const int64_t* point_offsets, const double* planes,
                              const float* spheres, const int64_t* box_offsets, int F, int64_t K_total,
                              int64_t max_points_per_frame, const double* pose, const int64_t* counts,
                              const int64_t* box_start, double* out_points, int32_t* out_index, int64_t out_capacity,
                              const void* workspace, size_t workspace_bytes, dal3_stream stream) {
    if (int e = crop_args_ok(points, point_offsets, planes, spheres, box_offsets, F, K_total, max_points_per_frame,
                             workspace, workspace_bytes))
        return e;
    if (K_total > 0 && (!pose || !counts || !box_start || !out_points || out_capacity < 0))
        return fail(DAL3_EINVAL, "crop_fill: null pose / counts / box_start / out, or a negative out_capacity");
    HIP_TRY(launch_crop_fill(points, point_offsets, planes, spheres, box_offsets, F, K_total, max_points_per_frame, pose,
                             counts, box_start, static_cast<const int32_t*>(workspace), out_points, out_index, out_capacity,
                             static_cast<hipStream_t>(stream)));
    return 0;
}

extern "C" int dal3_crop_starts(const int64_t* counts, const int64_t* order, int64_t K_total, int64_t* box_start,
                                int64_t* out_offsets, dal3_stream stream) {
    if (K_total < 0 || !box_start || (K_total > 0 && !counts))
        return fail(DAL3_EINVAL, "crop_starts: null counts / box_start or K_total < 0");
    HIP_TRY(launch_crop_starts(counts, order, K_total, box_start, out_offsets, -1, static_cast<hipStream_t>(stream)));
    return 0;
}

extern "C" int dal3_crop_starts_capped(const int64_t* counts, const int64_t* order, int64_t K_total, int64_t* box_start,
                                       int64_t* out_offsets, int64_t out_capacity, dal3_stream stream) {
    if (K_total < 0 || !box_start || (K_total > 0 && !counts) || out_capacity < 0)
        return fail(DAL3_EINVAL, "crop_starts_capped: null counts / box_start, K_total < 0 or out_capacity < 0");
    HIP_TRY(launch_crop_starts(counts, order, K_total, box_start, out_offsets, out_capacity, static_cast<hipStream_t>(stream)));
    return 0;
}

extern "C" int dal3_writeback_boxes(const double* final_boxes, const int32_t* final_idx, const double* pose_best,
                                    const double* pose_inv, const double* track_box, float* det,
                                    const int64_t* det_start, const int32_t* det_count, const uint8_t* active, int P,
                                    int64_t n_det, int32_t* match, int32_t* owner, dal3_stream stream) {
    if (!final_boxes || !final_idx || !pose_inv || !track_box || !det || !det_start || !det_count || !active || !match ||
        !owner || P <= 0 || n_det <= 0)
        return fail(DAL3_EINVAL, "writeback_boxes: bad argument");
    HIP_TRY(launch_writeback(final_boxes, final_idx, pose_best, pose_inv, track_box, det, det_start, det_count, active, P,
                             n_det, match, owner, static_cast<hipStream_t>(stream)));
    return 0;
}

// ---------------------------------------------------------------------------------------------- training blocks (N4)
static bool mult32(int64_t v) { return v > 0 && v % 32 == 0; }

extern "C" int dal3_tr_linear(const float* a, int64_t M, int c_in, int64_t lda, const float* scale, const float* shift,
                              int relu_in, const float* W, int64_t ldw, int transpose_w, const float* bias, int64_t seg,
                              int c_out, float* z, int64_t ldz, int accumulate, void* workspace, size_t workspace_bytes,
                              dal3_stream stream) {
    if (!a || !W || !z || !mult32(M) || !mult32(c_in) || !mult32(c_out) || lda < c_in || ldz < c_out || lda % 4 || ldz % 4 ||
        (!transpose_w && (ldw < c_in || ldw % 4)) || (transpose_w && ldw < c_out) || (scale && !shift) || seg < 0)
        return fail(DAL3_EINVAL, "tr_linear: bad argument (M, c_in, c_out multiples of 32; strides multiples of 4)");
    const size_t need = tr_linear_workspace_bytes(c_in, c_out);
    if (need && (!workspace || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 15)))
        return fail(DAL3_EWORKSPACE, "tr_linear: workspace smaller than dal3_tr_linear_workspace_bytes() or not 16-byte aligned");
    HIP_TRY(launch_tr_linear(a, M, c_in, lda, scale, shift, relu_in, W, ldw, transpose_w, bias, seg, c_out, z, ldz,
                             accumulate, need ? static_cast<float*>(workspace) : nullptr, static_cast<hipStream_t>(stream)));
    return 0;
}

extern "C" int dal3_tr_linear_pack_layout(int64_t M, int c_in, int64_t seg, int c_out, int accumulate, int has_act) {
    if (!mult32(M) || !mult32(c_in) || !mult32(c_out)) return 0;
    return tr_linear_pack_mtb(M, c_in, seg, c_out, accumulate, has_act);
}

extern "C" int dal3_tr_pack_many(const dal3_tr_pack_item* items, int n, dal3_stream stream) {
    if (!items || n <= 0 || n > 48) return fail(DAL3_EINVAL, "tr_pack_many: 1 .. 48 items");
    for (int i = 0; i < n; ++i) {
        const dal3_tr_pack_item& t = items[i];
        const int mtb = t.mtb & 0xff;
        if ((t.mtb & ~0x1ff) || ((t.mtb & 0x100) && mtb != 8)) return fail(DAL3_EINVAL, "tr_pack_many: bad layout code in item %d", i);
        if (!t.W || !t.out || !mult32(t.c_in) || !mult32(t.c_out) || mtb <= 0 || (t.c_out / 32) % mtb != 0 ||
            (reinterpret_cast<uintptr_t>(t.out) & 15) || (!t.transpose_w && t.ldw < t.c_in) || (t.transpose_w && t.ldw < t.c_out))
            return fail(DAL3_EINVAL, "tr_pack_many: bad item %d", i);
    }
    HIP_TRY(launch_tr_pack_many(items, n, static_cast<hipStream_t>(stream)));
    return 0;
}

extern "C" int dal3_tr_linear_prepacked(const float* a, int64_t M, int c_in, int64_t lda, const float* scale,
                                        const float* shift, int relu_in, const float* W, int64_t ldw, int transpose_w,
                                        const float* bias, int64_t seg, int c_out, float* z, int64_t ldz, int accumulate,
                                        const void* packed, dal3_stream stream) {
    if (!a || !W || !z || !mult32(M) || !mult32(c_in) || !mult32(c_out) || lda < c_in || ldz < c_out || lda % 4 || ldz % 4 ||
        (scale && !shift) || seg < 0 || !packed || (reinterpret_cast<uintptr_t>(packed) & 15))
        return fail(DAL3_EINVAL, "tr_linear_prepacked: bad argument");
    if (tr_linear_pack_mtb(M, c_in, seg, c_out, accumulate, scale != nullptr) == 0)
        return fail(DAL3_EINVAL, "tr_linear_prepacked: this call reads no packed image (dal3_tr_linear_pack_layout() == 0)");
    HIP_TRY(launch_tr_linear(a, M, c_in, lda, scale, shift, relu_in, W, ldw, transpose_w, bias, seg, c_out, z, ldz, accumulate,
                             const_cast<float*>(static_cast<const float*>(packed)), static_cast<hipStream_t>(stream), true));
    return 0;
}

extern "C" size_t dal3_tr_linear_red_workspace_bytes(int64_t rows, int c_out) {
    if (rows <= 0 || c_out <= 0) return 0;
    const size_t a = tr_colred_workspace_bytes(rows, c_out), b = tr_linear_red_workspace_bytes();
    return a > b ? a : b;
}

static int tr_linear_red_args_ok(const float* a, int64_t M, int c_in, int64_t lda, int c_out, const float* z, int64_t ldz,
                                 const void* packed, int64_t rows) {
    return a && z && mult32(M) && mult32(c_in) && mult32(c_out) && lda >= c_in && ldz >= c_out && lda % 4 == 0 && ldz % 4 == 0 &&
           packed && !(reinterpret_cast<uintptr_t>(packed) & 15) && rows >= 2 && rows <= M;
}

extern "C" int dal3_tr_linear_bn_stats(const float* a, int64_t M, int c_in, int64_t lda, const float* scale, const float* shift,
                                       int relu_in, const float* W, int64_t ldw, const float* bias, int64_t seg, int c_out,
                                       float* z, int64_t ldz, const void* packed, int64_t rows, const float* gamma,
                                       const float* beta, float* running_mean, float* running_var, float momentum, float eps,
                                       float* mu, float* rstd, float* bn_scale, float* bn_shift, void* workspace,
                                       size_t workspace_bytes, dal3_stream stream) {
    if (!tr_linear_red_args_ok(a, M, c_in, lda, c_out, z, ldz, packed, rows) || !W || (scale && !shift) || seg < 0 || !gamma ||
        !beta || !mu || !rstd || !bn_scale || !bn_shift || (!running_mean != !running_var))
        return fail(DAL3_EINVAL, "tr_linear_bn_stats: bad argument");
    if (tr_linear_pack_mtb(M, c_in, seg, c_out, 0, scale != nullptr) == 0)
        return fail(DAL3_EINVAL, "tr_linear_bn_stats: this call reads no packed image (dal3_tr_linear_pack_layout() == 0)");
    if (!workspace || workspace_bytes < dal3_tr_linear_red_workspace_bytes(rows, c_out))
        return fail(DAL3_EWORKSPACE, "tr_linear_bn_stats: workspace smaller than dal3_tr_linear_red_workspace_bytes()");
    int fused = 0;
    HIP_TRY(launch_tr_linear_bn_stats(a, M, c_in, lda, scale, shift, relu_in, W, ldw, bias, seg, c_out, z, ldz,
                                      const_cast<float*>(static_cast<const float*>(packed)), rows, gamma, beta, running_mean,
                                      running_var, momentum, eps, mu, rstd, bn_scale, bn_shift, static_cast<double*>(workspace),
                                      static_cast<hipStream_t>(stream), &fused));
    return fused;
}

extern "C" int dal3_tr_linear_bnbwd_sums(const float* dz, int64_t M, int c_in, int64_t lddz, const float* W, int64_t ldw, int c_out,
                                         float* da, int64_t ldda, const void* packed, int64_t rows, const float* bz, int64_t ldbz,
                                         const float* bscale, const float* bshift, const float* bmu, const float* brstd,
                                         const float* gamma, float* dgamma, float* dbeta, float* k1, float* k2, float* k3,
                                         void* workspace, size_t workspace_bytes, dal3_stream stream) {
    if (!tr_linear_red_args_ok(dz, M, c_in, lddz, c_out, da, ldda, packed, rows) || !W || !bz || ldbz < c_out || ldbz % 4 ||
        !bscale || !bshift || !bmu || !brstd || !gamma || !dgamma || !dbeta || !k1 || !k2 || !k3)
        return fail(DAL3_EINVAL, "tr_linear_bnbwd_sums: bad argument");
    if (tr_linear_pack_mtb(M, c_in, 0, c_out, 0, 0) == 0)
        return fail(DAL3_EINVAL, "tr_linear_bnbwd_sums: this call reads no packed image (dal3_tr_linear_pack_layout() == 0)");
    if (!workspace || workspace_bytes < dal3_tr_linear_red_workspace_bytes(rows, c_out))
        return fail(DAL3_EWORKSPACE, "tr_linear_bnbwd_sums: workspace smaller than dal3_tr_linear_red_workspace_bytes()");
    int fused = 0;
    HIP_TRY(launch_tr_linear_bnbwd_sums(dz, M, c_in, lddz, W, ldw, c_out, da, ldda, const_cast<float*>(static_cast<const float*>(packed)),
                                        rows, bz, ldbz, bscale, bshift, bmu, brstd, gamma, dgamma, dbeta, k1, k2, k3,
                                        static_cast<double*>(workspace), static_cast<hipStream_t>(stream), &fused));
    return fused;
}

extern "C" int dal3_tr_linear_x3_layout(int64_t M, int c_in, int64_t seg, int c_out, int accumulate, int has_act) {
    if (!mult32(M) || !mult32(c_in) || !mult32(c_out)) return 0;
    return tr_linear_x3_layout(M, c_in, seg, c_out, accumulate, has_act);
}

extern "C" int dal3_tr_linear_x3(const float* a, int64_t M, int c_in, int64_t lda, const float* scale, const float* shift,
                                 int relu_in, const float* bias, int64_t seg, int c_out, float* z, int64_t ldz,
                                 const void* packed, const uint32_t* in_amax, dal3_stream stream) {
    if (!a || !z || !mult32(M) || !mult32(c_in) || !mult32(c_out) || lda < c_in || ldz < c_out || lda % 4 || ldz % 4 ||
        (scale && !shift) || seg < 0 || !packed || (reinterpret_cast<uintptr_t>(packed) & 15) || (in_amax && scale) ||
        (reinterpret_cast<uintptr_t>(in_amax) & 3) ||
        (reinterpret_cast<uintptr_t>(a) & 15) || (reinterpret_cast<uintptr_t>(z) & 15) || (bias && (reinterpret_cast<uintptr_t>(bias) & 15)))
        return fail(DAL3_EINVAL, "tr_linear_x3: bad argument (16-byte aligned a / z / bias / packed, row strides multiples of 4)");
    const int layout = tr_linear_x3_layout(M, c_in, seg, c_out, 0, scale != nullptr);
    if (layout == 0) return fail(DAL3_EINVAL, "tr_linear_x3: this shape does not take the f16x3 kernel (dal3_tr_linear_x3_layout() == 0)");
    if (lda * 64 * (int64_t)sizeof(float) >= ((int64_t)1 << 31)) return fail(DAL3_EINVAL, "tr_linear_x3: lda too large");
    HIP_TRY(launch_tr_linear_x3(a, M, c_in, lda, scale, shift, relu_in, static_cast<const uint16_t*>(packed), layout, bias, seg, c_out,
                                z, ldz, static_cast<hipStream_t>(stream), in_amax));
    return 0;
}

extern "C" size_t dal3_tr_linear_workspace_bytes(int c_in, int c_out) {
    return (c_in > 0 && c_out > 0) ? tr_linear_workspace_bytes(c_in, c_out) : 0;
}

extern "C" size_t dal3_tr_colred_workspace_bytes(int64_t M, int C) {
    return (M > 0 && C > 0) ? tr_colred_workspace_bytes(M, C) : 0;
}

extern "C" int dal3_tr_act_colsum(const float* x, int64_t M, int C, int64_t ldx, const float* scale, const float* shift, int relu,
                                  float* out, int64_t ldo, void* workspace, size_t workspace_bytes, double* sums, dal3_stream stream) {
    if (!x || !out || !sums || M <= 0 || C <= 0 || (C & 3) || ldx < C || (ldx & 3) || ldo < C || (ldo & 3) || (scale && !shift))
        return fail(DAL3_EINVAL, "tr_act_colsum: bad argument (C and the row strides multiples of 4)");
    if (!workspace || workspace_bytes < tr_colred_workspace_bytes(M, C))
        return fail(DAL3_EWORKSPACE, "tr_act_colsum: workspace smaller than dal3_tr_colred_workspace_bytes()");
    HIP_TRY(launch_tr_act_colsum(x, M, C, ldx, scale, shift, relu, out, ldo, static_cast<double*>(workspace), sums,
                                 static_cast<hipStream_t>(stream)));
    return 0;
}

extern "C" int dal3_tr_colred(const float* z, int64_t M, int C, int64_t ldz, int mode, const float* da, int64_t ldda,
                              const float* dg, const int32_t* arg, int64_t seg, const float* scale, const float* shift,
                              const float* mu, const float* rstd, void* workspace, size_t workspace_bytes, double* out,
                              dal3_stream stream) {
    if (!z || M <= 0 || C <= 0 || C % 4 || ldz < C || ldz % 4 || !out || (mode != 0 && mode != 1) || (da && ldda % 4))
        return fail(DAL3_EINVAL, "tr_colred: bad argument (C and the row strides must be multiples of 4)");
    if (mode == 1 && (!scale || !shift || !mu || !rstd || (!da && (!dg || !arg || seg <= 0))))
        return fail(DAL3_EINVAL, "tr_colred: mode 1 needs scale/shift/mu/rstd and da or (dg, arg, seg)");
    if (!workspace || workspace_bytes < tr_colred_workspace_bytes(M, C))
        return fail(DAL3_EWORKSPACE, "tr_colred: workspace smaller than dal3_tr_colred_workspace_bytes()");
    HIP_TRY(launch_tr_colred(z, M, C, ldz, mode, da, ldda, dg, arg, seg, scale, shift, mu, rstd,
                             static_cast<double*>(workspace), out, static_cast<hipStream_t>(stream)));
    return 0;
}

extern "C" int dal3_tr_pool_coef(const float* dg, const float* g, const float* zarg, const float* mu, const float* rstd,
                                 const float* gamma, int B, int C, int64_t M, double* coef, float* kd, dal3_stream stream) {
    if (!dg || !g || !zarg || !mu || !rstd || !gamma || B <= 0 || C <= 0 || M <= 0 || !coef || !kd)
        return fail(DAL3_EINVAL, "tr_pool_coef: bad argument");
    HIP_TRY(launch_tr_pool_coef(dg, g, zarg, mu, rstd, gamma, B, C, M, coef, kd, static_cast<hipStream_t>(stream)));
    return 0;
}

static bool pool_k_ok(int K) { return K == 64 || K == 128 || K == 256; }

extern "C" int dal3_tr_pool_moments(const float* W, int64_t ldw, const float* b, const double* m1, const float* Sc, int64_t M, int C,
                                    int K, double* sums, dal3_stream stream) {
    if (!W || !b || !m1 || !Sc || !sums || M <= 0 || C <= 0 || !pool_k_ok(K) || ldw < K)
        return fail(DAL3_EINVAL, "tr_pool_moments: bad argument (K = 64, 128 or 256; ldw >= K)");
    HIP_TRY(launch_tr_pool_moments(W, ldw, b, m1, Sc, M, C, K, sums, static_cast<hipStream_t>(stream)));
    return 0;
}

static bool conv1_shape_ok(int c_in, int c_out) { return c_in >= 1 && c_in <= 8 && (c_out == 64 || c_out == 128); }

extern "C" size_t dal3_tr_conv1_workspace_bytes(int64_t Mp, int c_out) { return Mp > 0 && c_out > 0 ? tr_conv1_workspace_bytes(Mp, c_out) : 0; }

extern "C" int dal3_tr_conv1_bn_stats(const float* x, int64_t M, int64_t Mp, int c_in, int64_t ldx, const float* W, int64_t ldw,
                                      const float* bias, int c_out, float* z, int64_t ldz, const float* gamma, const float* beta,
                                      float* running_mean, float* running_var, float momentum, float eps, float* mu, float* rstd,
                                      float* scale, float* shift, void* workspace, size_t workspace_bytes, dal3_stream stream) {
    if (!x || !W || !bias || !z || !gamma || !beta || !mu || !rstd || !scale || !shift || M < 2 || Mp < M || !conv1_shape_ok(c_in, c_out) ||
        ldx < c_in || ldw < c_in || ldz < c_out || (ldz & 3) || (!running_mean != !running_var) ||
        (reinterpret_cast<uintptr_t>(z) & 15) || (reinterpret_cast<uintptr_t>(bias) & 15))
        return fail(DAL3_EINVAL, "tr_conv1_bn_stats: bad argument (c_in <= 8, c_out 64 or 128, 16-byte aligned z / bias, M >= 2)");
    if (!workspace || workspace_bytes < tr_conv1_workspace_bytes(Mp, c_out))
        return fail(DAL3_EWORKSPACE, "tr_conv1_bn_stats: workspace smaller than dal3_tr_conv1_workspace_bytes()");
    HIP_TRY(launch_tr_conv1_bn_stats(x, M, Mp, c_in, ldx, W, ldw, bias, c_out, z, ldz, gamma, beta, running_mean, running_var, momentum,
                                     eps, mu, rstd, scale, shift, static_cast<double*>(workspace), static_cast<hipStream_t>(stream)));
    return 0;
}

extern "C" int dal3_tr_conv1_wgrad(const float* dz, int64_t lddz, const float* x, int64_t M, int c_in, int64_t ldx, int c_out,
                                   void* workspace, size_t workspace_bytes, float* dW, dal3_stream stream) {
    if (!dz || !x || !dW || M <= 0 || !conv1_shape_ok(c_in, c_out) || ldx < c_in || lddz < c_out || (lddz & 3) ||
        (reinterpret_cast<uintptr_t>(dz) & 15))
        return fail(DAL3_EINVAL, "tr_conv1_wgrad: bad argument (c_in <= 8, c_out 64 or 128, 16-byte aligned dz)");
    if (!workspace || workspace_bytes < tr_conv1_workspace_bytes(M, c_out))
        return fail(DAL3_EWORKSPACE, "tr_conv1_wgrad: workspace smaller than dal3_tr_conv1_workspace_bytes()");
    HIP_TRY(launch_tr_conv1_wgrad(dz, lddz, x, M, c_in, ldx, c_out, static_cast<double*>(workspace), dW, static_cast<hipStream_t>(stream)));
    return 0;
}

extern "C" int dal3_parse_box_pred(const float* box_pred, int64_t ldb, int64_t B, float* center, float* heading_scores,
                                   float* heading_residuals_normalized, float* heading_residuals, float* size_scores,
                                   float* size_residuals_normalized, float* size_residuals, dal3_stream stream) {
    if (!box_pred || !center || !heading_scores || !heading_residuals_normalized || !heading_residuals || !size_scores ||
        !size_residuals_normalized || !size_residuals || B <= 0 || B > (1 << 24) || ldb < 39)
        return fail(DAL3_EINVAL, "parse_box_pred: bad argument (seven outputs, ldb >= 39)");
    HIP_TRY(launch_parse_box_pred(box_pred, ldb, (int)B, center, heading_scores, heading_residuals_normalized, heading_residuals,
                                  size_scores, size_residuals_normalized, size_residuals, static_cast<hipStream_t>(stream)));
    return 0;
}

extern "C" int dal3_parse_box_pred_backward(const float* g_center, const float* g_heading_scores,
                                            const float* g_heading_residuals_normalized, const float* g_heading_residuals,
                                            const float* g_size_scores, const float* g_size_residuals_normalized,
                                            const float* g_size_residuals, int64_t B, float* g_box_pred, dal3_stream stream) {
    if (!g_box_pred || B <= 0 || B > (1 << 24)) return fail(DAL3_EINVAL, "parse_box_pred_backward: bad argument");
    HIP_TRY(launch_parse_box_pred_backward(g_center, g_heading_scores, g_heading_residuals_normalized, g_heading_residuals, g_size_scores,
                                           g_size_residuals_normalized, g_size_residuals, (int)B, g_box_pred,
                                           static_cast<hipStream_t>(stream)));
    return 0;
}

extern "C" int dal3_tr_fc_max_rows(void) { return tr_fc_max_rows(); }
extern "C" int dal3_tr_fc_max_act_cin(void) { return tr_fc_max_act_cin(); }

extern "C" int dal3_tr_fc_forward(const float* a, int64_t B, int c_in, int64_t lda, const float* in_scale, const float* in_shift,
                                  int relu_in, const float* W, int64_t ldw, int transpose_w, const float* bias, int c_out, float* z,
                                  int64_t ldz, const float* gamma, const float* beta, float* running_mean, float* running_var,
                                  float momentum, float eps, float* mu, float* rstd, float* scale, float* shift, dal3_stream stream) {
    if (!a || !W || !z || B < 1 || B > tr_fc_max_rows() || c_in <= 0 || c_out <= 0 || lda < c_in || ldz < c_out ||
        ldw < (transpose_w ? c_out : c_in) || (in_scale && (!in_shift || c_in > tr_fc_max_act_cin())) || (!running_mean != !running_var))
        return fail(DAL3_EINVAL, "tr_fc_forward: bad argument (1 <= B <= dal3_tr_fc_max_rows(); strides at least the row lengths; "
                                 "c_in <= 2048 under an input activation)");
    if (gamma && (!beta || !mu || !rstd || !scale || !shift || B < 2))
        return fail(DAL3_EINVAL, "tr_fc_forward: a BatchNorm needs beta, the four outputs and at least two rows");
    HIP_TRY(launch_tr_fc_forward(a, (int)B, c_in, lda, in_scale, in_shift, relu_in, W, ldw, transpose_w, bias, c_out, z, ldz, gamma, beta,
                                 gamma ? running_mean : nullptr, gamma ? running_var : nullptr, momentum, eps, mu, rstd, scale, shift,
                                 static_cast<hipStream_t>(stream)));
    return 0;
}

extern "C" int dal3_tr_fc_backward_w(const float* da, int64_t ldda, int64_t B, int c_out, const float* z, int64_t ldz, const float* scale,
                                     const float* shift, const float* mu, const float* rstd, const float* gamma, float* dgamma,
                                     float* dbeta, const float* a_prev, int c_in, int64_t lda, const float* in_scale,
                                     const float* in_shift, int relu_in, float* dz, int64_t lddz, float* dW, int64_t lddw, float* db,
                                     dal3_stream stream) {
    if (!da || B < 1 || B > tr_fc_max_rows() || c_out <= 0 || ldda < c_out || (dz && lddz < c_out) ||
        (in_scale && (!in_shift || c_in > tr_fc_max_act_cin())) ||
        (dW && (!a_prev || c_in <= 0 || lda < c_in || lddw < c_in)))
        return fail(DAL3_EINVAL, "tr_fc_backward_w: bad argument (1 <= B <= dal3_tr_fc_max_rows(); strides at least the row lengths)");
    if (scale && (!z || ldz < c_out || !shift || !mu || !rstd || !gamma || !dgamma || !dbeta))
        return fail(DAL3_EINVAL, "tr_fc_backward_w: a BatchNorm needs z, shift, mu, rstd, gamma and the two gradient outputs");
    HIP_TRY(launch_tr_fc_backward_w(da, ldda, (int)B, c_out, z, ldz, scale, shift, mu, rstd, gamma, dgamma, dbeta, a_prev, c_in, lda,
                                    in_scale, in_shift, relu_in, dz, lddz, dW, lddw, db, static_cast<hipStream_t>(stream)));
    return 0;
}

extern "C" int dal3_tr_gather_at(const float* z, int64_t ldz, const int32_t* arg, int64_t seg, int n_seg, int C, float* out,
                                 dal3_stream stream) {
    if (!z || !arg || !out || seg <= 0 || n_seg <= 0 || C <= 0 || ldz < C) return fail(DAL3_EINVAL, "tr_gather_at: bad argument");
    HIP_TRY(launch_tr_gather_at(z, ldz, arg, seg, n_seg, C, out, static_cast<hipStream_t>(stream)));
    return 0;
}

extern "C" int dal3_tr_pool_zarg(const int32_t* arg, const float* a, int64_t lda, const float* W, int64_t ldw, const float* bias,
                                 int B, int C, int K, int N, float* zarg, dal3_stream stream) {
    if (!arg || !a || !W || !bias || !zarg || B <= 0 || C <= 0 || N <= 0 || K <= 0 || K % 4 || lda < K || ldw < K || lda % 4 || ldw % 4 ||
        (reinterpret_cast<uintptr_t>(a) & 15) || (reinterpret_cast<uintptr_t>(W) & 15))
        return fail(DAL3_EINVAL, "tr_pool_zarg: bad argument (K and the row strides multiples of 4, 16-byte aligned a / W)");
    HIP_TRY(launch_tr_pool_zarg(arg, a, lda, W, ldw, bias, B, C, K, N, zarg, static_cast<hipStream_t>(stream)));
    return 0;
}

extern "C" size_t dal3_tr_pool_gv_workspace_bytes(int K) { return pool_k_ok(K) ? tr_pool_gv_workspace_bytes(K) : 0; }

extern "C" int dal3_tr_pool_gv(const double* coef, const float* W, int64_t ldw, const float* b, int C, int K, float* G, float* v,
                               void* workspace, size_t workspace_bytes, dal3_stream stream) {
    if (!coef || !W || !b || !G || !v || C <= 0 || !pool_k_ok(K) || ldw < K)
        return fail(DAL3_EINVAL, "tr_pool_gv: bad argument (K = 64, 128 or 256; ldw >= K)");
    if (!workspace || workspace_bytes < tr_pool_gv_workspace_bytes(K))
        return fail(DAL3_EWORKSPACE, "tr_pool_gv: workspace too small (dal3_tr_pool_gv_workspace_bytes)");
    HIP_TRY(launch_tr_pool_gv(coef, W, ldw, b, C, K, G, v, static_cast<double*>(workspace), static_cast<hipStream_t>(stream)));
    return 0;
}

extern "C" int dal3_tr_pool_dw(const double* coef, const float* W, int64_t ldw, const float* b, const float* S, const double* m1,
                               int64_t M, int centred, const float* dWs, int C, int K, float* dW, dal3_stream stream) {
    if (!coef || !W || !b || !S || !m1 || !dWs || !dW || M <= 0 || C <= 0 || !pool_k_ok(K) || ldw < K)
        return fail(DAL3_EINVAL, "tr_pool_dw: bad argument (K = 64, 128 or 256; ldw >= K)");
    HIP_TRY(launch_tr_pool_dw(coef, W, ldw, b, S, m1, M, centred, dWs, C, K, dW, static_cast<hipStream_t>(stream)));
    return 0;
}

extern "C" int dal3_tr_pool_sparse(const int32_t* arg, const float* kd, const float* W, int64_t ldw, const float* a,
                                   int64_t lda, int B, int C, int K, int N, float* da, int64_t ldda, float* dWs,
                                   dal3_stream stream) {
    if (!arg || !kd || !W || !a || !da || !dWs || B <= 0 || C <= 0 || N <= 0 || (K != 128 && K != 256 && K != 64) ||
        ldw < K || lda < K || ldda < K || ldw % 4 || ldda % 4)
        return fail(DAL3_EINVAL, "tr_pool_sparse: bad argument (K = 64, 128 or 256; row strides >= K, multiples of 4)");
    if ((size_t)(2 * (size_t)N + 1 + (size_t)C) * sizeof(int) > 65536 || C > 4096)
        return fail(DAL3_EINVAL, "tr_pool_sparse: 2 N + C + 1 must not exceed 16384 and C must not exceed 4096 (an item's buckets live in LDS)");
    HIP_TRY(launch_tr_pool_sparse(arg, kd, W, ldw, a, lda, B, C, K, N, da, ldda, dWs, static_cast<hipStream_t>(stream)));
    return 0;
}

extern "C" int dal3_tr_box_loss(const float* center, const float* center_label, const float* heading_scores,
                                const float* heading_residuals_normalized, const int64_t* heading_class_label,
                                const float* heading_residuals_label, const float* size_scores,
                                const float* size_residuals_normalized, const int64_t* size_class_label,
                                const float* size_residuals_label, int B, float* losses, float* g_center,
                                float* g_heading_scores, float* g_heading_residuals_normalized, float* g_size_scores,
                                float* g_size_residuals_normalized, dal3_stream stream) {
    if (!center || !center_label || !heading_scores || !heading_residuals_normalized || !heading_class_label ||
        !heading_residuals_label || !size_scores || !size_residuals_normalized || !size_class_label ||
        !size_residuals_label || B <= 0 || !losses || !g_center || !g_heading_scores || !g_heading_residuals_normalized ||
        !g_size_scores || !g_size_residuals_normalized)
        return fail(DAL3_EINVAL, "tr_box_loss: bad argument");
    HIP_TRY(launch_tr_box_loss(center, center_label, heading_scores, heading_residuals_normalized, heading_class_label,
                               heading_residuals_label, size_scores, size_residuals_normalized, size_class_label,
                               size_residuals_label, B, losses, g_center, g_heading_scores, g_heading_residuals_normalized,
                               g_size_scores, g_size_residuals_normalized, static_cast<hipStream_t>(stream)));
    return 0;
}

extern "C" size_t dal3_tr_seg_ce_workspace_bytes(int64_t M) { return M > 0 ? tr_seg_ce_workspace_bytes(M) : 0; }

extern "C" int dal3_tr_seg_ce(const float* logits, const void* labels, int labels_are_int64, int64_t M, float* loss,
                              float* dlogits, void* workspace, size_t workspace_bytes, dal3_stream stream) {
    if (!logits || !labels || M <= 0 || !loss || !dlogits) return fail(DAL3_EINVAL, "tr_seg_ce: bad argument");
    if ((reinterpret_cast<uintptr_t>(logits) | reinterpret_cast<uintptr_t>(dlogits)) & 7)
        return fail(DAL3_EINVAL, "tr_seg_ce: logits and dlogits must be 8-byte aligned");
    if (!workspace || workspace_bytes < tr_seg_ce_workspace_bytes(M) || (reinterpret_cast<uintptr_t>(workspace) & 7))
        return fail(DAL3_EWORKSPACE, "tr_seg_ce: workspace smaller than dal3_tr_seg_ce_workspace_bytes() or misaligned");
    HIP_TRY(launch_tr_seg_ce(logits, labels, labels_are_int64, M, loss, dlogits, static_cast<double*>(workspace),
                             static_cast<hipStream_t>(stream)));
    return 0;
}

extern "C" int dal3_tr_bn_stats(const float* z, int64_t M, int C, int64_t ldz, const float* gamma, const float* beta,
                                float* running_mean, float* running_var, float momentum, float eps, float* mu, float* rstd,
                                float* scale, float* shift, void* workspace, size_t workspace_bytes, dal3_stream stream) {
    if (!z || M < 2 || C <= 0 || C % 4 || ldz < C || ldz % 4 || !gamma || !beta || !mu || !rstd || !scale || !shift ||
        (!running_mean != !running_var))
        return fail(DAL3_EINVAL, "tr_bn_stats: bad argument (C and the row stride must be multiples of 4, M >= 2)");
    if (!workspace || workspace_bytes < tr_colred_workspace_bytes(M, C))
        return fail(DAL3_EWORKSPACE, "tr_bn_stats: workspace smaller than dal3_tr_colred_workspace_bytes()");
    HIP_TRY(launch_tr_bn_stats(z, M, C, ldz, gamma, beta, running_mean, running_var, momentum, eps, mu, rstd, scale, shift,
                               static_cast<double*>(workspace), static_cast<hipStream_t>(stream)));
    return 0;
}

extern "C" int dal3_tr_bnbwd_sums(const float* z, int64_t M, int C, int64_t ldz, const float* da, int64_t ldda,
                                  const float* dg, const int32_t* arg, int64_t seg, const float* scale, const float* shift,
                                  const float* mu, const float* rstd, const float* gamma, float* dgamma, float* dbeta,
                                  float* k1, float* k2, float* k3, void* workspace, size_t workspace_bytes,
                                  dal3_stream stream) {
    if (!z || M <= 0 || C <= 0 || C % 4 || ldz < C || ldz % 4 || (da && ldda % 4) || !scale || !shift || !mu || !rstd ||
        (!da && (!dg || !arg || seg <= 0)) || !gamma || !dgamma || !dbeta || !k1 || !k2 || !k3)
        return fail(DAL3_EINVAL, "tr_bnbwd_sums: bad argument (C and the row strides must be multiples of 4)");
    if (!workspace || workspace_bytes < tr_colred_workspace_bytes(M, C))
        return fail(DAL3_EWORKSPACE, "tr_bnbwd_sums: workspace smaller than dal3_tr_colred_workspace_bytes()");
    HIP_TRY(launch_tr_bnbwd_sums(z, M, C, ldz, da, ldda, dg, arg, seg, scale, shift, mu, rstd, gamma, dgamma, dbeta, k1, k2,
                                 k3, static_cast<double*>(workspace), static_cast<hipStream_t>(stream)));
    return 0;
}

extern "C" int dal3_tr_bn_finalize(const double* sums, int C, int64_t M, const float* gamma, const float* beta,
                                   float* running_mean, float* running_var, float momentum, float eps, float* mu,
                                   float* rstd, float* scale, float* shift, dal3_stream stream) {
    if (!sums || C <= 0 || M < 2 || !gamma || !beta || !mu || !rstd || !scale || !shift || (!running_mean != !running_var))
        return fail(DAL3_EINVAL, "tr_bn_finalize: bad argument");
    HIP_TRY(launch_tr_bn_finalize(sums, C, M, gamma, beta, running_mean, running_var, momentum, eps, mu, rstd, scale, shift,
                                  static_cast<hipStream_t>(stream)));
    return 0;
}

extern "C" int dal3_tr_bnbwd_coef(const double* sums, int C, int64_t M, const float* gamma, const float* rstd, float* dgamma,
                                  float* dbeta, float* k1, float* k2, float* k3, dal3_stream stream) {
    if (!sums || C <= 0 || M <= 0 || !gamma || !rstd || !dgamma || !dbeta || !k1 || !k2 || !k3)
        return fail(DAL3_EINVAL, "tr_bnbwd_coef: bad argument");
    HIP_TRY(launch_tr_bnbwd_coef(sums, C, M, gamma, rstd, dgamma, dbeta, k1, k2, k3, static_cast<hipStream_t>(stream)));
    return 0;
}

extern "C" int dal3_tr_bnbwd_apply(const float* z, int64_t M, int C, int64_t ldz, const float* da, int64_t ldda,
                                   const float* dg, const int32_t* arg, int64_t seg, const float* scale,
                                   const float* shift, const float* mu, const float* rstd, const float* k1,
                                   const float* k2, const float* k3, float* dz, int64_t lddz, dal3_stream stream) {
    if (!z || M <= 0 || C <= 0 || C % 4 || !scale || !shift || !mu || !rstd || !k1 || !k2 || !k3 || !dz || ldz < C ||
        lddz < C || ldz % 4 || lddz % 4 || (da && ldda % 4) || (!da && (!dg || !arg || seg <= 0)))
        return fail(DAL3_EINVAL, "tr_bnbwd_apply: bad argument (C and the row strides must be multiples of 4)");
    HIP_TRY(launch_tr_bnbwd_apply(z, M, C, ldz, da, ldda, dg, arg, seg, scale, shift, mu, rstd, k1, k2, k3, dz, lddz,
                                  static_cast<hipStream_t>(stream)));
    return 0;
}

extern "C" int dal3_tr_bnbwd_apply_amax(const float* z, int64_t M, int C, int64_t ldz, const float* da, int64_t ldda,
                                        const float* dg, const int32_t* arg, int64_t seg, const float* scale,
                                        const float* shift, const float* mu, const float* rstd, const float* k1,
                                        const float* k2, const float* k3, float* dz, int64_t lddz, uint32_t* amax,
                                        dal3_stream stream) {
    if (!z || M <= 0 || C <= 0 || C % 64 || !scale || !shift || !mu || !rstd || !k1 || !k2 || !k3 || !dz || ldz < C ||
        lddz < C || ldz % 4 || lddz % 4 || (da && ldda % 4) || (!da && (!dg || !arg || seg <= 0)) || !amax ||
        (reinterpret_cast<uintptr_t>(amax) & 3))
        return fail(DAL3_EINVAL, "tr_bnbwd_apply_amax: bad argument (C a multiple of 64, the row strides of 4, amax 64 device words)");
    HIP_TRY(launch_tr_bnbwd_apply(z, M, C, ldz, da, ldda, dg, arg, seg, scale, shift, mu, rstd, k1, k2, k3, dz, lddz,
                                  static_cast<hipStream_t>(stream), amax));
    return 0;
}

extern "C" size_t dal3_tr_bnbwd_apply_segsum_workspace_bytes(int64_t M, int C) {
    return (M > 0 && C > 0) ? tr_bnbwd_apply_segsum_workspace_bytes(M, C) : 0;
}

extern "C" int dal3_tr_bnbwd_apply_segsum(const float* z, int64_t M, int C, int64_t ldz, const float* da, int64_t ldda,
                                          const float* scale, const float* shift, const float* mu, const float* rstd,
                                          const float* k1, const float* k2, const float* k3, float* dz, int64_t lddz,
                                          int64_t sum_seg, float* seg_sums, void* workspace, size_t workspace_bytes,
                                          dal3_stream stream) {
    if (!z || !da || M <= 0 || C <= 0 || C % 64 || !scale || !shift || !mu || !rstd || !k1 || !k2 || !k3 || !dz || ldz < C ||
        lddz < C || ldda < C || ldz % 4 || lddz % 4 || ldda % 4 || sum_seg <= 0 || sum_seg % 128 || M % sum_seg || !seg_sums)
        return fail(DAL3_EINVAL, "tr_bnbwd_apply_segsum: bad argument (C % 64 == 0, sum_seg % 128 == 0, M % sum_seg == 0, dense da)");
    if (!workspace || workspace_bytes < tr_bnbwd_apply_segsum_workspace_bytes(M, C))
        return fail(DAL3_EWORKSPACE, "tr_bnbwd_apply_segsum: workspace too small");
    HIP_TRY(launch_tr_bnbwd_apply_segsum(z, M, C, ldz, da, ldda, scale, shift, mu, rstd, k1, k2, k3, dz, lddz, sum_seg, seg_sums,
                                         static_cast<double*>(workspace), static_cast<hipStream_t>(stream)));
    return 0;
}

extern "C" size_t dal3_tr_wgrad_workspace_bytes(int64_t M, int c_out, int c_in) {
    return (M > 0 && c_out > 0 && c_in > 0) ? tr_wgrad_workspace_bytes(M, c_out, c_in) : 0;
}

extern "C" int dal3_tr_wgrad(const float* dz, int64_t lddz, const float* a, int64_t lda, const float* scale,
                             const float* shift, int relu_in, int64_t M, int c_out, int c_in, void* workspace,
                             size_t workspace_bytes, float* dW, dal3_stream stream) {
    if (!dz || !a || !mult32(M) || !mult32(c_out) || !mult32(c_in) || lddz < c_out || lda < c_in || (scale && !shift))
        return fail(DAL3_EINVAL, "tr_wgrad: bad argument (M, c_in, c_out multiples of 32)");
    if (!workspace || workspace_bytes < tr_wgrad_workspace_bytes(M, c_out, c_in))
        return fail(DAL3_EWORKSPACE, "tr_wgrad: workspace smaller than dal3_tr_wgrad_workspace_bytes()");
    HIP_TRY(launch_tr_wgrad(dz, lddz, a, lda, scale, shift, relu_in, M, c_out, c_in, static_cast<float*>(workspace), dW,
                            static_cast<hipStream_t>(stream)));
    return 0;
}

extern "C" int dal3_tr_wgrad_final_many(const dal3_tr_wgrad_part* items, int n, dal3_stream stream) {
    if (!items || n <= 0 || n > 24) return fail(DAL3_EINVAL, "tr_wgrad_final_many: 1 .. 24 items");
    for (int i = 0; i < n; ++i)
        if (!items[i].part || !items[i].dW || items[i].n <= 0 || items[i].n_slices <= 0 || items[i].n_slices > 0x7fffffff)
            return fail(DAL3_EINVAL, "tr_wgrad_final_many: bad item %d", i);
    HIP_TRY(launch_tr_wgrad_final_many(items, n, static_cast<hipStream_t>(stream)));
    return 0;
}

extern "C" size_t dal3_tr_wgrad_x3_workspace_bytes(int64_t M, int c_out, int c_in) {
    return (M > 0 && c_out > 0 && c_in > 0 && tr_wgrad_x3_ok(M, c_out, c_in)) ? tr_wgrad_x3_workspace_bytes(M, c_out, c_in) : 0;
}

extern "C" int dal3_tr_wgrad_x3(const float* dz, int64_t lddz, const float* a, int64_t lda, const float* scale,
                                const float* shift, int relu_in, const uint32_t* dz_amax, int64_t M, int c_out, int c_in,
                                void* workspace, size_t workspace_bytes, float* dW, dal3_stream stream) {
    if (!dz || !a || !mult32(M) || !mult32(c_out) || !mult32(c_in) || lddz < c_out || lda < c_in || (scale && !shift) ||
        (lddz & 3) || (lda & 3) || (reinterpret_cast<uintptr_t>(dz) & 15) || (reinterpret_cast<uintptr_t>(a) & 15) ||
        (scale && ((reinterpret_cast<uintptr_t>(scale) | reinterpret_cast<uintptr_t>(shift)) & 15)) ||
        (reinterpret_cast<uintptr_t>(dz_amax) & 3))
        return fail(DAL3_EINVAL, "tr_wgrad_x3: bad argument (M, c_in, c_out multiples of 32; 16-byte aligned operands, strides multiples of 4)");
    if (!tr_wgrad_x3_ok(M, c_out, c_in))
        return fail(DAL3_EINVAL, "tr_wgrad_x3: this shape does not take the f16x3 kernel (dal3_tr_wgrad_x3_workspace_bytes() == 0)");
    if (!workspace || workspace_bytes < tr_wgrad_x3_workspace_bytes(M, c_out, c_in))
        return fail(DAL3_EWORKSPACE, "tr_wgrad_x3: workspace smaller than dal3_tr_wgrad_x3_workspace_bytes()");
    HIP_TRY(launch_tr_wgrad_x3(dz, lddz, a, lda, scale, shift, relu_in, dz_amax, M, c_out, c_in, static_cast<float*>(workspace), dW,
                               static_cast<hipStream_t>(stream)));
    return 0;
}

extern "C" int dal3_tr_segmax(const float* z, int64_t ldz, int64_t seg, int C, const float* scale, const float* shift,
                              float* g, int32_t* arg, int64_t n_seg, void* workspace, size_t workspace_bytes,
                              dal3_stream stream) {
    if (!z || !scale || !shift || !g || !arg || seg <= 0 || C <= 0 || n_seg <= 0 || ldz < C)
        return fail(DAL3_EINVAL, "tr_segmax: bad argument");
    if (!workspace || workspace_bytes < (size_t)n_seg * C * 8 || (reinterpret_cast<uintptr_t>(workspace) & 7))
        return fail(DAL3_EWORKSPACE, "tr_segmax: workspace needs n_seg * C * 8 bytes, 8-byte aligned");
    HIP_TRY(launch_tr_segmax(z, ldz, seg, C, scale, shift, g, arg, n_seg, static_cast<unsigned long long*>(workspace),
                             static_cast<hipStream_t>(stream)));
    return 0;
}

extern "C" int dal3_tr_act_dropout(const float* x, int64_t M, int C, int64_t ldx, const float* scale, const float* shift,
                                   int relu, const float* mult, int64_t ldm, uint64_t seed, const int64_t* step, float p_drop,
                                   float* out, int64_t ldo, dal3_stream stream) {
    if (!x || !out || M <= 0 || C <= 0 || (C & 3) || ldx < C || ldo < C || (ldx & 3) || (ldo & 3) || (scale && !shift) ||
        (mult && (ldm < C || (ldm & 3))) || !(p_drop >= 0.0f && p_drop <= 1.0f))
        return fail(DAL3_EINVAL, "tr_act_dropout: bad argument (C and the row strides multiples of 4, 0 <= p <= 1)");
    HIP_TRY(launch_tr_act_dropout(x, M, C, ldx, scale, shift, relu, mult, ldm, seed, step, p_drop, out, ldo,
                                  static_cast<hipStream_t>(stream)));
    return 0;
}

static bool head2_common_ok(int64_t M, int C, const float* mult, int64_t ldm, float p_drop) {
    return M > 0 && C == 128 && (!mult || (ldm >= C && !(ldm & 3) && !(reinterpret_cast<uintptr_t>(mult) & 15))) &&
           p_drop >= 0.0f && p_drop <= 1.0f;
}

extern "C" int dal3_tr_head2_forward(const float* z, int64_t M, int C, int64_t ldz, const float* scale, const float* shift, int relu,
                                     const float* mult, int64_t ldm, uint64_t seed, const int64_t* step, float p_drop,
                                     const float* W, int64_t ldw, const float* bias, float* logits, dal3_stream stream) {
    if (!z || !W || !bias || !logits || !head2_common_ok(M, C, mult, ldm, p_drop) || ldz < C || (ldz & 3) || ldw < C || (ldw & 3) ||
        (scale && !shift) || (reinterpret_cast<uintptr_t>(z) & 15) || (reinterpret_cast<uintptr_t>(W) & 15) ||
        (reinterpret_cast<uintptr_t>(logits) & 7))
        return fail(DAL3_EINVAL, "tr_head2_forward: bad argument (C == 128, row strides multiples of 4, 16-byte aligned z / W / mult)");
    HIP_TRY(launch_tr_head2_forward(z, M, ldz, scale, shift, relu, mult, ldm, seed, step, p_drop, W, ldw, bias, logits,
                                    static_cast<hipStream_t>(stream)));
    return 0;
}

extern "C" int dal3_tr_head2_dgrad(const float* dlogits, int64_t M, int C, const float* mult, int64_t ldm, uint64_t seed,
                                   const int64_t* step, float p_drop, const float* W, int64_t ldw, float* da, int64_t ldda,
                                   dal3_stream stream) {
    if (!dlogits || !W || !da || !head2_common_ok(M, C, mult, ldm, p_drop) || ldw < C || (ldw & 3) || ldda < C || (ldda & 3) ||
        (reinterpret_cast<uintptr_t>(dlogits) & 7) || (reinterpret_cast<uintptr_t>(W) & 15) || (reinterpret_cast<uintptr_t>(da) & 15))
        return fail(DAL3_EINVAL, "tr_head2_dgrad: bad argument (C == 128, row strides multiples of 4, aligned pointers)");
    HIP_TRY(launch_tr_head2_dgrad(dlogits, M, mult, ldm, seed, step, p_drop, W, ldw, da, ldda, static_cast<hipStream_t>(stream)));
    return 0;
}

extern "C" int dal3_tr_head2_dgrad_bnbwd(const float* dlogits, int64_t M, int C, const float* mult, int64_t ldm, uint64_t seed,
                                         const int64_t* step, float p_drop, const float* W, int64_t ldw, float* da, int64_t ldda,
                                         const float* bz, int64_t ldbz, const float* bscale, const float* bshift, const float* bmu,
                                         const float* brstd, const float* gamma, float* dgamma, float* dbeta, float* k1, float* k2,
                                         float* k3, void* workspace, size_t workspace_bytes, dal3_stream stream) {
    if (!dlogits || !W || !da || !bz || !bscale || !bshift || !bmu || !brstd || !gamma || !dgamma || !dbeta || !k1 || !k2 || !k3 ||
        !head2_common_ok(M, C, mult, ldm, p_drop) || ldw < C || (ldw & 3) || ldda < C || (ldda & 3) || ldbz < C || (ldbz & 3) ||
        (reinterpret_cast<uintptr_t>(dlogits) & 7) || (reinterpret_cast<uintptr_t>(W) & 15) || (reinterpret_cast<uintptr_t>(da) & 15) ||
        (reinterpret_cast<uintptr_t>(bz) & 15))
        return fail(DAL3_EINVAL, "tr_head2_dgrad_bnbwd: bad argument (C == 128, row strides multiples of 4, aligned pointers)");
    if (!workspace || workspace_bytes < tr_colred_workspace_bytes(M, C))
        return fail(DAL3_EWORKSPACE, "tr_head2_dgrad_bnbwd: workspace smaller than dal3_tr_colred_workspace_bytes(M, 128)");
    HIP_TRY(launch_tr_head2_dgrad_bnbwd(dlogits, M, mult, ldm, seed, step, p_drop, W, ldw, da, ldda, bz, ldbz, bscale, bshift, bmu, brstd,
                                        gamma, dgamma, dbeta, k1, k2, k3, static_cast<double*>(workspace),
                                        static_cast<hipStream_t>(stream)));
    return 0;
}

extern "C" size_t dal3_tr_head2_wgrad_workspace_bytes(int64_t M) { return M > 0 ? tr_head2_wgrad_workspace_bytes(M) : 0; }

extern "C" int dal3_tr_head2_wgrad(const float* dlogits, const float* z, int64_t M, int C, int64_t ldz, const float* scale,
                                   const float* shift, int relu, const float* mult, int64_t ldm, uint64_t seed, const int64_t* step,
                                   float p_drop, void* workspace, size_t workspace_bytes, float* dWb, dal3_stream stream) {
    if (!dlogits || !z || !dWb || !head2_common_ok(M, C, mult, ldm, p_drop) || ldz < C || (ldz & 3) || (scale && !shift) ||
        (reinterpret_cast<uintptr_t>(z) & 15) || (reinterpret_cast<uintptr_t>(dlogits) & 7))
        return fail(DAL3_EINVAL, "tr_head2_wgrad: bad argument (C == 128, row strides multiples of 4, aligned pointers)");
    if (!workspace || workspace_bytes < tr_head2_wgrad_workspace_bytes(M))
        return fail(DAL3_EWORKSPACE, "tr_head2_wgrad: workspace smaller than dal3_tr_head2_wgrad_workspace_bytes()");
    HIP_TRY(launch_tr_head2_wgrad(dlogits, z, M, ldz, scale, shift, relu, mult, ldm, seed, step, p_drop,
                                  static_cast<double*>(workspace), dWb, static_cast<hipStream_t>(stream)));
    return 0;
}

extern "C" size_t dal3_tr_linear_pool_workspace_bytes(int c_in, int c_out, int64_t n_seg) {
    return tr_linear_workspace_bytes(c_in, c_out) + (size_t)n_seg * c_out * 8;
}

extern "C" int dal3_tr_linear_pool(const float* a, int64_t M, int c_in, int64_t lda, const float* scale, const float* shift,
                                   int relu_in, const float* W, int64_t ldw, const float* bias, const float* out_scale,
                                   const float* out_shift, int64_t seg, int c_out, float* g, int32_t* arg, void* workspace,
                                   size_t workspace_bytes, dal3_stream stream) {
    if (!a || !W || !out_scale || !out_shift || !g || !arg || !mult32(M) || !mult32(c_in) || c_out <= 0 || c_out % 128 != 0 ||
        lda < c_in || (lda & 3) || ldw < c_in || (ldw & 3) || (scale && !shift) || (scale && c_in > 1024) || seg <= 0 ||
        seg % 32 != 0 || M % seg != 0)
        return fail(DAL3_EINVAL, "tr_linear_pool: bad argument (M, c_in multiples of 32; c_out of 128; seg a multiple of 32 "
                                 "that divides M; strides multiples of 4)");
    const size_t wbytes = tr_linear_workspace_bytes(c_in, c_out), need = wbytes + (size_t)(M / seg) * c_out * 8;
    if (!workspace || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 15) || (wbytes & 7))
        return fail(DAL3_EWORKSPACE, "tr_linear_pool: workspace smaller than dal3_tr_linear_pool_workspace_bytes() or not 16-byte aligned");
    char* base = static_cast<char*>(workspace);
    HIP_TRY(launch_tr_linear_pool(a, M, c_in, lda, scale, shift, relu_in, W, ldw, bias, out_scale, out_shift, seg, c_out, g, arg,
                                  reinterpret_cast<float*>(base), reinterpret_cast<unsigned long long*>(base + wbytes),
                                  static_cast<hipStream_t>(stream)));
    return 0;
}

extern "C" int dal3_tr_linear_pool_x3_ok(int64_t M, int c_in, int64_t seg, int c_out) {
    return (mult32(M) && mult32(c_in) && c_out > 0 && seg > 0 && M % seg == 0 && tr_linear_pool_x3_ok(M, c_in, seg, c_out)) ? 1 : 0;
}

extern "C" int dal3_tr_linear_pool_x3(const float* a, int64_t M, int c_in, int64_t lda, const float* scale, const float* shift,
                                      int relu_in, const float* W, int64_t ldw, const float* bias, const float* out_scale,
                                      const float* out_shift, int64_t seg, int c_out, float* g, int32_t* arg, void* workspace,
                                      size_t workspace_bytes, dal3_stream stream) {
    if (!a || !W || !out_scale || !out_shift || !g || !arg || !mult32(M) || !mult32(c_in) || c_out <= 0 || lda < c_in || (lda & 3) ||
        ldw < c_in || (ldw & 3) || (scale && !shift) || seg <= 0 || M % seg != 0 || (reinterpret_cast<uintptr_t>(a) & 15) ||
        lda * 64 * (int64_t)sizeof(float) >= ((int64_t)1 << 31))
        return fail(DAL3_EINVAL, "tr_linear_pool_x3: bad argument");
    if (!tr_linear_pool_x3_ok(M, c_in, seg, c_out))
        return fail(DAL3_EINVAL, "tr_linear_pool_x3: this shape does not take the f16x3 kernel (dal3_tr_linear_pool_x3_ok() == 0)");
    const size_t wbytes = tr_linear_workspace_bytes(c_in, c_out), need = wbytes + (size_t)(M / seg) * c_out * 8;
    if (!workspace || workspace_bytes < need || (reinterpret_cast<uintptr_t>(workspace) & 15) || (wbytes & 7))
        return fail(DAL3_EWORKSPACE, "tr_linear_pool_x3: workspace smaller than dal3_tr_linear_pool_workspace_bytes() or not 16-byte aligned");
    char* base = static_cast<char*>(workspace);
    HIP_TRY(launch_tr_linear_pool_x3(a, M, c_in, lda, scale, shift, relu_in, W, ldw, bias, out_scale, out_shift, seg, c_out, g, arg,
                                     reinterpret_cast<float*>(base), reinterpret_cast<unsigned long long*>(base + wbytes),
                                     static_cast<hipStream_t>(stream)));
    return 0;
}

extern "C" int dal3_tr_segsum(const float* x, int64_t ldx, int64_t seg, int C, float* out, int64_t n_seg,
                              dal3_stream stream) {
    if (!x || !out || seg <= 0 || C <= 0 || n_seg <= 0 || ldx < C) return fail(DAL3_EINVAL, "tr_segsum: bad argument");
    HIP_TRY(launch_tr_segsum(x, ldx, seg, C, out, n_seg, static_cast<hipStream_t>(stream)));
    return 0;
}

extern "C" int dal3_maxpool_n(const float* x, int64_t rows, int64_t n, float* out, dal3_stream stream) {
    if (!x || !out || rows <= 0 || n <= 0) return fail(DAL3_EINVAL, "maxpool_n: bad argument");
    HIP_TRY(launch_maxpool_n(x, DAL3_F32, rows, n, out, static_cast<hipStream_t>(stream)));
    return 0;
}

extern "C" int dal3_maxpool_n_dtype(const void* x, int dtype, int64_t rows, int64_t n, void* out, dal3_stream stream) {
    if (dtype != DAL3_F32 && dtype != DAL3_BF16 && dtype != DAL3_F16) return fail(DAL3_EINVAL, "maxpool_n: storage dtype %d", dtype);
    if (!x || !out || rows <= 0 || n <= 0) return fail(DAL3_EINVAL, "maxpool_n: bad argument");
    if (dtype != DAL3_F32 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(out)) & 1))
        return fail(DAL3_EINVAL, "maxpool_n: 16-bit rows must be 2-byte aligned");
    HIP_TRY(launch_maxpool_n(x, dtype, rows, n, out, static_cast<hipStream_t>(stream)));
    return 0;
}

extern "C" size_t dal3_shared_mlp_layer_workspace_bytes(int c_in, int c_out) {
    const size_t mt = (size_t)(c_out + 31) / 32;
    const size_t w = c_in <= 8 ? mt * 4 * 64 : mt * (size_t)(c_in / 32) * 1024;
    return (al(w) + al(mt * 32)) * sizeof(float);
}

extern "C" int dal3_shared_mlp_layer(const dal3_layer* layer, int relu, dal3_bcn x, int B, int N, float* y,
                                     void* workspace, size_t workspace_bytes, dal3_stream stream) {
    if (!layer || !y || !workspace) return fail(DAL3_EINVAL, "shared_mlp_layer: bad argument");
    TRY(check_extent(B, N, "shared_mlp_layer"));
    TRY(check_bcn(x, "x"));
    if (x.dtype != DAL3_F32) return fail(DAL3_EINVAL, "shared_mlp_layer: the layer-wise test entry takes fp32 input only");
    const dal3_layer& L = *layer;
    if (L.c_out % 32 != 0) return fail(DAL3_EINVAL, "shared_mlp_layer: c_out must be a multiple of 32");
    const bool first = L.c_in <= 8;
    if (!first && L.c_in % 32 != 0) return fail(DAL3_EINVAL, "shared_mlp_layer: c_in must be <= 8 or a multiple of 32");
    TRY(check_layer(L, L.c_in, L.c_out, "layer"));
    const size_t need = dal3_shared_mlp_layer_workspace_bytes(L.c_in, L.c_out);
    if (workspace_bytes < need) return fail(DAL3_EWORKSPACE, "shared_mlp_layer: workspace needs %zu bytes", need);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int mt = L.c_out / 32;
    const int ks = first ? (L.c_in + 1) / 2 : 0;
    const int kt = first ? 0 : L.c_in / 32;
    float* wbuf = static_cast<float*>(workspace);
    float* bbuf = wbuf + al(first ? (size_t)mt * 4 * 64 : (size_t)mt * kt * 1024);
    if (first) HIP_TRY(launch_pack_weight(L, PACK_FIRST, 0, L.c_in, mt, ks, wbuf, s));
    else HIP_TRY(launch_pack_weight(L, PACK_FRAG_MT_MAJOR, 0, L.c_in, mt, kt, wbuf, s));
    HIP_TRY(launch_pack_bias(L, bbuf, s));
    HIP_TRY(launch_generic_layer(reinterpret_cast<const f32x4*>(wbuf), wbuf, bbuf, kt, ks, mt, relu, to_bcn(x), L.c_in,
                                 B, N, y, s));
    return 0;
}

// ---------------------------------------------------------------------------------- static model
struct StaticWs { InsSegWs seg; int32_t* pos; float* obj; float* obj2; HeadWs head; };
static StaticWs carve_static(Carver& c, int B, int N, int two_stage) {
    StaticWs w;
    w.seg = carve_ins_seg(c, B);
    w.pos = c.take<int32_t>((size_t)B * N);
    w.obj = c.take<float>((size_t)B * 512 * 3);
    w.obj2 = two_stage ? c.take<float>((size_t)B * 512 * 3) : nullptr;
    w.head = carve_head(c, B);
    return w;
}
extern "C" size_t dal3_static_workspace_bytes(int B, int N, int two_stage) {
    Carver c(nullptr, 0);
    carve_static(c, B, N, two_stage);
    return c.off;
}

extern "C" int dal3_static_forward(const dal3_static_args* a, int phases, dal3_stream stream) {
    if (!a) return fail(DAL3_EINVAL, "static_forward: args is NULL");
    TRY(check_extent(a->B, a->N, "static_forward"));
    if (!a->workspace) return fail(DAL3_EINVAL, "static_forward: workspace is NULL");
    if (!(phases & DAL3_PHASE_ALL)) return fail(DAL3_EINVAL, "static_forward: no phase selected");
    hipStream_t s = static_cast<hipStream_t>(stream);
    Carver c(a->workspace, a->workspace_bytes);
    const StaticWs ws = carve_static(c, a->B, a->N, a->two_stage);
    if (!c.ok) return fail(DAL3_EWORKSPACE, "static_forward: workspace needs %zu bytes, got %zu", c.off, a->workspace_bytes);
    const int B = a->B, N = a->N, M = 512;

    if (phases & DAL3_PHASE_SEG) {
        TRY(ins_seg_run(a->w_ins_seg, a->dtype, 3, a->pts, B, N, a->logits, a->mask, nullptr, ws.seg, s));
        // (when the box phase follows in this call, its compaction kernel writes the same counts)
        if (a->counts && !(phases & DAL3_PHASE_BOX)) HIP_TRY(launch_segment_counts(a->mask, B, N, a->counts, s));
    }
    if (!(phases & DAL3_PHASE_BOX)) return 0;

    if (!a->init_box || !a->box_pred_one || !a->counts || !a->obj_idx || !a->w_box_est_one)
        return fail(DAL3_EINVAL, "static_forward: null pointer in box phase");
    TRY(gather_run(a->mask, a->pts, B, N, 3, M, a->sampler, a->choice, a->seed, a->item_offset, a->counts, a->obj_idx,
                   ws.obj, ws.pos, s));
    const dal3_bcn obj{ws.obj, (int64_t)M * 3, 1, 3, DAL3_F32, a->pts.flags};
    // device sampler: the first min(count, M) object points are distinct, the rest are copies -> skipped by the head
    const int32_t* distinct = a->sampler == DAL3_SAMPLER_DEVICE ? a->counts : nullptr;
    // (the estimator's last FC layer and the decode of its output are one launch: fc39_decode_kernel)
    if (!a->two_stage) {
        // center = center_boxnet + init_box[:, :3] (static_model.py:132); yaw += init yaw (static_eval.py:280)
        const DecodeArgs d1{a->box_pred_one, a->init_box, 7, 0, nullptr, 0, a->init_box + 6, 7,
                            a->heading_residuals_one, a->size_residuals_one, a->center_one, a->boxes7};
        TRY(point_head_run(DAL3_HEAD_STATIC_BOX_EST, a->w_box_est_one, a->dtype, obj, B, M, a->box_pred_one, 39, ws.head, s,
                           distinct, &d1));
        return 0;
    }
    if (!a->w_box_est_two || !a->box_pred_two || !a->box_one || !a->center_one)
        return fail(DAL3_EINVAL, "static_forward: two_stage needs w_box_est_two, box_pred_two, box_one, center_one");
    // center_one += init_box[:, :3] in place (static_model.py:174); box_one (:176-190)
    const DecodeArgs d1{a->box_pred_one, a->init_box, 7, 1, nullptr, 0, a->init_box + 6, 7,
                        a->heading_residuals_one, a->size_residuals_one, a->center_one, a->box_one};
    TRY(point_head_run(DAL3_HEAD_STATIC_BOX_EST, a->w_box_est_one, a->dtype, obj, B, M, a->box_pred_one, 39, ws.head, s,
                       distinct, &d1));
    HIP_TRY(launch_recenter(ws.obj, B, M, a->init_box, a->box_one, a->bbox_gt, ws.obj2,
                            a->bbox_gt ? a->heading_class_label_two : nullptr,
                            a->bbox_gt ? a->heading_residuals_label_two : nullptr, s));
    const dal3_bcn obj2{ws.obj2, (int64_t)M * 3, 1, 3, DAL3_F32, a->pts.flags};
    // center_two += center_one (static_model.py:211); final yaw += box_one yaw (static_eval.py:282)
    const DecodeArgs d2{a->box_pred_two, a->center_one, 3, 1, nullptr, 0, a->box_one + 6, 7,
                        a->heading_residuals_two, a->size_residuals_two, a->center_two, a->boxes7};
    TRY(point_head_run(DAL3_HEAD_STATIC_BOX_EST, a->w_box_est_two, a->dtype, obj2, B, M, a->box_pred_two, 39, ws.head, s,
                       distinct, &d2));
    return 0;
}

// ---------------------------------------------------------------------------------- dynamic model
struct DynamicWs { InsSegWs seg; int32_t* pos; float* obj; HeadWs head; };
static DynamicWs carve_dynamic(Carver& c, int B, int N, int M) {
    DynamicWs w;
    w.seg = carve_ins_seg(c, B);
    w.pos = c.take<int32_t>((size_t)B * N);
    w.obj = c.take<float>((size_t)B * M * 4);
    w.head = carve_head(c, B);
    return w;
}
extern "C" size_t dal3_dynamic_workspace_bytes(int B, int N, int) {
    Carver c(nullptr, 0);
    carve_dynamic(c, B, N, 2560);
    return c.off;
}

extern "C" int dal3_dynamic_forward(const dal3_dynamic_args* a, int phases, dal3_stream stream) {
    if (!a) return fail(DAL3_EINVAL, "dynamic_forward: args is NULL");
    TRY(check_extent(a->B, a->N, "dynamic_forward"));
    TRY(check_extent(a->B, a->n_box, "dynamic_forward (n_box)"));
    if (!a->workspace) return fail(DAL3_EINVAL, "dynamic_forward: workspace is NULL");
    if (!(phases & DAL3_PHASE_ALL)) return fail(DAL3_EINVAL, "dynamic_forward: no phase selected");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int B = a->B, N = a->N, M = 2560;
    Carver c(a->workspace, a->workspace_bytes);
    const DynamicWs ws = carve_dynamic(c, B, N, M);
    if (!c.ok) return fail(DAL3_EWORKSPACE, "dynamic_forward: workspace needs %zu bytes, got %zu", c.off, a->workspace_bytes);

    if (phases & DAL3_PHASE_SEG) {
        TRY(ins_seg_run(a->w_ins_seg, a->dtype, 4, a->pts, B, N, a->logits, a->mask, nullptr, ws.seg, s));
        // (when the box phase follows in this call, its compaction kernel writes the same counts)
        if (a->counts && !(phases & DAL3_PHASE_BOX)) HIP_TRY(launch_segment_counts(a->mask, B, N, a->counts, s));
    }
    if (!(phases & DAL3_PHASE_BOX)) return 0;

    if (!a->embedding || !a->box_pred || !a->counts || !a->obj_idx || !a->w_point_emb || !a->w_box_emb || !a->w_box_est)
        return fail(DAL3_EINVAL, "dynamic_forward: null pointer in box phase");
    TRY(gather_run(a->mask, a->pts, B, N, 4, M, a->sampler, a->choice, a->seed, a->item_offset, a->counts, a->obj_idx,
                   ws.obj, ws.pos, s));
    const dal3_bcn obj{ws.obj, (int64_t)M * 4, 1, 4, DAL3_F32, a->pts.flags};
    // embedding = cat[point_e (256), box_e (128)] (dynamic_model.py:133-137): written side by side
    TRY(point_head_run(DAL3_HEAD_POINT_EMB, a->w_point_emb, a->dtype, obj, B, M, a->embedding, 384, ws.head, s,
                       a->sampler == DAL3_SAMPLER_DEVICE ? a->counts : nullptr));
    TRY(point_head_run(DAL3_HEAD_BOX_EMB, a->w_box_emb, a->dtype, a->box, B, a->n_box, a->embedding + 256, 384, ws.head, s));
    // forward() adds nothing to the centre; the eval driver adds init_box[:, :3] and yaw init_box[:, -2]
    const DecodeArgs dd{a->box_pred, nullptr, 0, 0, a->init_box8, 8, a->init_box8 ? a->init_box8 + 6 : nullptr, 8,
                        a->heading_residuals, a->size_residuals, nullptr, a->boxes7};
    TRY(fc_chain(fc_head_view(static_cast<const float*>(a->w_box_est)), a->embedding, 384, a->box_pred, 39, B,
                 ws.head.t1, ws.head.t2, s, &dd));
    return 0;
}
