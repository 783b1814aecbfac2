/*
 * The drop-in boundary used from plain C: no Python, no torch - only include/dsdenoise.h and the HIP runtime for
 * device memory.  Builds a small WaveNet denoiser from weights in a flat file, hoists the conditioner projections,
 * runs one backbone evaluation (dsd_denoise) and a 5-step DDIM-shaped sampling program (dsd_sample), and writes
 * the outputs to a file.  tests/test_gpu_c_abi.py compiles this with hipcc, runs it on the MI355X and compares
 * both outputs with the Python shim on the same inputs.
 *
 *   gcc -std=c99 -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude examples/c_abi_denoise.c \
 *       -Ldiffsinger_amd -ldsdenoise -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/diffsinger_amd -o demo
 *   ./demo weights.bin inputs.bin outputs.bin
 *
 * weights.bin: int32 n, then n records { int32 name_len, name bytes, int32 ndim, int64 shape[ndim], float data[] }
 * inputs.bin : int32 B, T, H, M; float cond[B*H*T], x[B*M*T], t[B]
 */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "dsdenoise.h"

#define CHECK(call)                                                                     \
    do {                                                                                \
        int rc_ = (call);                                                               \
        if (rc_ != 0) {                                                                 \
            fprintf(stderr, "%s failed (%d): %s\n", #call, rc_, dsd_last_error(h));     \
            return 1;                                                                   \
        }                                                                               \
    } while (0)

static void* to_device(const void* src, size_t bytes) {
    void* d = NULL;
    if (hipMalloc(&d, bytes) != hipSuccess) return NULL;
    if (hipMemcpy(d, src, bytes, hipMemcpyHostToDevice) != hipSuccess) return NULL;
    return d;
}

int main(int argc, char** argv) {
    if (argc != 4) {
        fprintf(stderr, "usage: %s weights.bin inputs.bin outputs.bin\n", argv[0]);
        return 2;
    }
    dsd_handle* h = NULL;
    dsd_config cfg;
    memset(&cfg, 0, sizeof(cfg));
    cfg.struct_size = (int32_t)sizeof(cfg);
    cfg.backbone = DSD_BACKBONE_WAVENET;
    cfg.n_feats = 1;
    cfg.num_layers = 4;
    cfg.num_channels = 64;
    cfg.dilation_cycle_length = 2;
    cfg.device = 0;

    FILE* fi = fopen(argv[2], "rb");
    if (!fi) return 2;
    int32_t dims[4];
    if (fread(dims, sizeof(int32_t), 4, fi) != 4) return 2;
    const int B = dims[0], T = dims[1], H = dims[2], M = dims[3];
    cfg.in_dims = M;
    cfg.hidden_size = H;
    const size_t n_cond = (size_t)B * H * T, n_x = (size_t)B * M * T;
    float* cond = (float*)malloc(n_cond * sizeof(float));
    float* x = (float*)malloc(n_x * sizeof(float));
    float* t = (float*)malloc((size_t)B * sizeof(float));
    if (fread(cond, sizeof(float), n_cond, fi) != n_cond || fread(x, sizeof(float), n_x, fi) != n_x ||
        fread(t, sizeof(float), (size_t)B, fi) != (size_t)B)
        return 2;
    fclose(fi);

    CHECK(dsd_create(&cfg, &h));
    FILE* fw = fopen(argv[1], "rb");
    if (!fw) return 2;
    int32_t n_tensors = 0;
    if (fread(&n_tensors, sizeof(int32_t), 1, fw) != 1) return 2;
    for (int32_t i = 0; i < n_tensors; ++i) {
        int32_t name_len, ndim;
        char name[256];
        int64_t shape[4];
        if (fread(&name_len, sizeof(int32_t), 1, fw) != 1 || name_len <= 0 || name_len > 255) return 2;
        if (fread(name, 1, (size_t)name_len, fw) != (size_t)name_len) return 2;
        name[name_len] = 0;
        if (fread(&ndim, sizeof(int32_t), 1, fw) != 1 || ndim < 1 || ndim > 4) return 2;
        if (fread(shape, sizeof(int64_t), (size_t)ndim, fw) != (size_t)ndim) return 2;
        size_t numel = 1;
        for (int d = 0; d < ndim; ++d) numel *= (size_t)shape[d];
        float* data = (float*)malloc(numel * sizeof(float));
        if (fread(data, sizeof(float), numel, fw) != numel) return 2;
        CHECK(dsd_load_weight(h, name, data, shape, ndim, /*on_device=*/0));
        free(data);
    }
    fclose(fw);
    CHECK(dsd_finalize_weights(h));

    float* d_cond = (float*)to_device(cond, n_cond * sizeof(float));
    float* d_x = (float*)to_device(x, n_x * sizeof(float));
    float* d_t = (float*)to_device(t, (size_t)B * sizeof(float));
    float *d_out = NULL, *d_samp = NULL;
    if (!d_cond || !d_x || !d_t || hipMalloc((void**)&d_out, n_x * sizeof(float)) != hipSuccess ||
        hipMalloc((void**)&d_samp, n_x * sizeof(float)) != hipSuccess)
        return 3;

    /* cond is [B, H, T]: strides (H*T, T, 1) */
    CHECK(dsd_prepare_cond(h, d_cond, B, T, (int64_t)H * T, T, 1, NULL));
    CHECK(dsd_denoise(h, d_x, d_t, B, d_out, NULL));

    /* a 5-evaluation program of the DDIM shape: x <- a_k * x + b_k * model(x, t_k), in place in buffer 0 */
    dsd_eval evals[5];
    memset(evals, 0, sizeof(evals));
    for (int k = 0; k < 5; ++k) {
        evals[k].x_buf = 0;
        evals[k].t = 900.0f - 200.0f * (float)k;
        evals[k].n_out = 1;
        evals[k].out[0].dst = 0;
        evals[k].out[0].n_terms = 2;
        evals[k].out[0].terms[0].src = 0;
        evals[k].out[0].terms[0].coef = 0.9f + 0.01f * (float)k;
        evals[k].out[0].terms[1].src = DSD_SRC_MODEL;
        evals[k].out[0].terms[1].coef = -0.2f + 0.03f * (float)k;
    }
    dsd_program prog;
    prog.n_bufs = 1;
    prog.result_buf = 0;
    prog.n_evals = 5;
    prog.n_noise = 0;
    prog.evals = evals;
    CHECK(dsd_sample(h, &prog, d_x, NULL, d_samp, NULL, NULL, DSD_SAMPLE_GRAPH, NULL));
    if (hipDeviceSynchronize() != hipSuccess) return 3;

    float* out = (float*)malloc(n_x * sizeof(float));
    float* samp = (float*)malloc(n_x * sizeof(float));
    if (hipMemcpy(out, d_out, n_x * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(samp, d_samp, n_x * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess)
        return 3;
    FILE* fo = fopen(argv[3], "wb");
    if (!fo) return 2;
    fwrite(out, sizeof(float), n_x, fo);
    fwrite(samp, sizeof(float), n_x, fo);
    fclose(fo);

    dsd_stats st;
    CHECK(dsd_get_stats(h, &st));
    printf("api v%d  weights %lld B  workspace %lld B  kernels/NFE %d  graphs %d\n", dsd_api_version(),
           (long long)st.weight_bytes, (long long)st.workspace_bytes, (int)st.kernels_per_nfe, (int)st.graphs_cached);
    dsd_destroy(h);
    return 0;
}
