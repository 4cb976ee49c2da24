/* examples/c_host_static_refine.c — the C ABI of lib3dal_hip.so driven from a plain C host (no Python, no torch):
 * what INTEGRATION.md section 3 sketches, as a program. It reads a little-endian file written by the caller
 *     int32 B, N, n_layers(=17: ins_seg conv1..5, dconv1..5, then box_est conv1..4, fc1..3)
 *     per layer: int32 c_in, c_out, has_bn; float32 weight[c_out*c_in], bias[c_out], then bn weight/bias/mean/var[c_out]
 *     float32 pts[B*N*3] (point-major), init_box[B*7]
 * runs StaticModelOneBoxEst's eval forward + decode on the GPU through dal3_pack_weights / dal3_static_forward, and
 * writes the (B,7) refined boxes as float32 to the output file. tests/test_gpu_c_host.py builds it with gcc and checks
 * the boxes against the Python module bit for bit.
 *
 *     gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include examples/c_host_static_refine.c \
 *         -L 3dal_pytorch_amd -l:lib3dal_hip.so -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/3dal_pytorch_amd -o c_host
 */
#include <hip/hip_runtime_api.h>
#include <stdio.h>
#include <stdlib.h>

#include "dal3.h"

#define CHECK_HIP(x)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                       \
            return 2;                                                                     \
        }                                                                                 \
    } while (0)
#define CHECK_DAL3(x)                                                                     \
    do {                                                                                  \
        if ((x) != DAL3_OK) {                                                             \
            fprintf(stderr, "%s: %s\n", #x, dal3_last_error());                           \
            return 3;                                                                     \
        }                                                                                 \
    } while (0)

static float* upload(FILE* f, size_t n) {
    float* h = (float*)malloc(n * sizeof(float));
    float* d = NULL;
    if (!h || fread(h, sizeof(float), n, f) != n) return NULL;
    if (hipMalloc((void**)&d, n * sizeof(float)) != hipSuccess) return NULL;
    if (hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return NULL;
    free(h);
    return d;
}

int main(int argc, char** argv) {
    if (argc != 3) {
        fprintf(stderr, "usage: %s input.bin boxes_out.bin\n", argv[0]);
        return 1;
    }
    FILE* f = fopen(argv[1], "rb");
    int32_t hdr[3];
    if (!f || fread(hdr, sizeof(int32_t), 3, f) != 3 || hdr[2] != 17) {
        fprintf(stderr, "bad input file\n");
        return 1;
    }
    const int B = hdr[0], N = hdr[1];
    dal3_layer layers[17];
    for (int l = 0; l < 17; ++l) {
        int32_t meta[3];
        if (fread(meta, sizeof(int32_t), 3, f) != 3) return 1;
        dal3_layer L = {0};
        L.c_in = meta[0];
        L.c_out = meta[1];
        L.weight = upload(f, (size_t)meta[0] * meta[1]);
        L.bias = upload(f, meta[1]);
        if (meta[2]) {
            L.bn_weight = upload(f, meta[1]);
            L.bn_bias = upload(f, meta[1]);
            L.bn_mean = upload(f, meta[1]);
            L.bn_var = upload(f, meta[1]);
        }
        if (!L.weight || !L.bias || (meta[2] && !L.bn_var)) return 1;
        layers[l] = L;
    }
    float* d_pts = upload(f, (size_t)B * N * 3);
    float* d_init = upload(f, (size_t)B * 7);
    fclose(f);
    if (!d_pts || !d_init) return 1;

    hipStream_t stream;
    CHECK_HIP(hipStreamCreate(&stream));
    void *w_seg = NULL, *w_box = NULL;
    size_t bytes = 0;
    CHECK_DAL3(dal3_pack_weights(DAL3_HEAD_INS_SEG, layers, 10, DAL3_F32, NULL, &bytes, stream));
    CHECK_HIP(hipMalloc(&w_seg, bytes));
    CHECK_DAL3(dal3_pack_weights(DAL3_HEAD_INS_SEG, layers, 10, DAL3_F32, w_seg, &bytes, stream));
    CHECK_DAL3(dal3_pack_weights(DAL3_HEAD_STATIC_BOX_EST, layers + 10, 7, DAL3_F32, NULL, &bytes, stream));
    CHECK_HIP(hipMalloc(&w_box, bytes));
    CHECK_DAL3(dal3_pack_weights(DAL3_HEAD_STATIC_BOX_EST, layers + 10, 7, DAL3_F32, w_box, &bytes, stream));

    dal3_static_args a = {0};
    a.B = B;
    a.N = N;
    a.sampler = DAL3_SAMPLER_DEVICE;
    a.dtype = DAL3_F32;
    a.seed = 10922081u;                                  /* the Python modules' default */
    a.pts.data = d_pts;                                  /* point-major (B,N,3) read in place */
    a.pts.stride_b = (int64_t)N * 3;
    a.pts.stride_c = 1;
    a.pts.stride_n = 3;
    a.init_box = d_init;
    a.w_ins_seg = w_seg;
    a.w_box_est_one = w_box;
    CHECK_HIP(hipMalloc((void**)&a.logits, (size_t)B * N * 2 * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&a.mask, (size_t)B * N));
    CHECK_HIP(hipMalloc((void**)&a.box_pred_one, (size_t)B * 39 * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&a.heading_residuals_one, (size_t)B * 12 * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&a.size_residuals_one, (size_t)B * 9 * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&a.center_one, (size_t)B * 3 * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&a.boxes7, (size_t)B * 7 * sizeof(float)));
    CHECK_HIP(hipMalloc((void**)&a.counts, (size_t)B * sizeof(int32_t)));
    CHECK_HIP(hipMalloc((void**)&a.obj_idx, (size_t)B * 512 * sizeof(int32_t)));
    a.workspace_bytes = dal3_static_workspace_bytes(B, N, 0);
    CHECK_HIP(hipMalloc(&a.workspace, a.workspace_bytes));

    CHECK_DAL3(dal3_static_forward(&a, DAL3_PHASE_ALL, stream));
    CHECK_HIP(hipStreamSynchronize(stream));

    float* boxes = (float*)malloc((size_t)B * 7 * sizeof(float));
    CHECK_HIP(hipMemcpy(boxes, a.boxes7, (size_t)B * 7 * sizeof(float), hipMemcpyDeviceToHost));
    FILE* o = fopen(argv[2], "wb");
    if (!o || fwrite(boxes, sizeof(float), (size_t)B * 7, o) != (size_t)B * 7) return 1;
    fclose(o);
    printf("dal3 %d: refined %d crops x %d points; box 0 = [%.4f %.4f %.4f %.4f %.4f %.4f %.4f]\n", dal3_version(), B, N,
           boxes[0], boxes[1], boxes[2], boxes[3], boxes[4], boxes[5], boxes[6]);
    return 0;
}
