// Measured peaks of THIS box, beside the spec-sheet ones bench.py divides by ("peak figures"):
//   * fp32 MFMA (v_mfma_f32_16x16x4_f32): every SIMD runs independent accumulator chains back to back
//   * HBM: device-to-device stream copy of 2 x 1 GiB (read + write counted)
// Diagnostic tool, never part of the product library:  hipcc --offload-arch=gfx950 -O3 tools/peak_bench.hip -o X && ./X
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                         \
    do {                                                                                 \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) {                                                          \
            fprintf(stderr, "%s failed: %s\n", #x, hipGetErrorString(e_));               \
            exit(1);                                                                     \
        }                                                                                \
    } while (0)

__global__ __launch_bounds__(256) void mfma_f32_loop(float* out, int iters, float a, float b) {
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    f32x4 s = acc[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}

__global__ __launch_bounds__(256) void stream_copy(const f32x4* __restrict__ src, f32x4* __restrict__ dst, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = src[i];
}

// Cost of a device-wide barrier among co-resident workgroups (one atomic counter, agent-scope release/acquire):
// what a persistent multi-layer kernel would pay INSTEAD of a kernel boundary.  Every workgroup reaches every
// barrier and the spin has a wall-clock deadline, so the grid always drains.
__global__ __launch_bounds__(256) void grid_barrier_loop(unsigned* counter, float* data, int rounds, int nwg) {
    const long long t0 = wall_clock64();
    float v = data[blockIdx.x * 256 + threadIdx.x];
    for (int r = 0; r < rounds; ++r) {
        data[blockIdx.x * 256 + threadIdx.x] = v + 1.f;             // something to release
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            atomicAdd(counter, 1u);
            const unsigned target = (unsigned)(r + 1) * (unsigned)nwg;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target &&
                   wall_clock64() - t0 < 50000000LL /* 0.5 s of the 100 MHz clock */)
                __builtin_amdgcn_s_sleep(1);
            __threadfence();
        }
        __syncthreads();
        v = data[((blockIdx.x + 1) % nwg) * 256 + threadIdx.x];     // something to acquire from a neighbour
    }
    data[blockIdx.x * 256 + threadIdx.x] = v;
}

int main() {
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    // ---- MFMA ----
    const int wgs = cus * 2, iters = 20000;            // 2 workgroups x 4 waves per CU = 2 waves per SIMD
    float* out;
    CHECK(hipMalloc(&out, (size_t)wgs * 256 * sizeof(float)));
    hipLaunchKernelGGL(mfma_f32_loop, dim3(wgs), dim3(256), 0, 0, out, 100, 1.0f, 0.5f);
    CHECK(hipDeviceSynchronize());
    double best_tf = 0;
    for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(mfma_f32_loop, dim3(wgs), dim3(256), 0, 0, out, iters, 1.0f, 0.5f);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double flops = (double)wgs * 4 /*waves*/ * iters * 8 /*mfma*/ * (2.0 * 16 * 16 * 4);
        const double tf = flops / (ms * 1e-3) / 1e12;
        if (tf > best_tf) best_tf = tf;
    }
    // ---- HBM stream copy ----
    const size_t bytes = (size_t)1 << 30;
    f32x4 *a, *b;
    CHECK(hipMalloc(&a, bytes));
    CHECK(hipMalloc(&b, bytes));
    CHECK(hipMemset(a, 1, bytes));
    CHECK(hipMemset(b, 0, bytes));
    double best_gbs = 0, best_memcpy = 0;
    for (int rep = 0; rep < 6; ++rep) {
        CHECK(hipEventRecord(e0));
        hipLaunchKernelGGL(stream_copy, dim3(cus * 8), dim3(256), 0, 0, a, b, bytes / sizeof(f32x4));
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double gbs = 2.0 * bytes / (ms * 1e-3) / 1e9;
        if (rep > 0 && gbs > best_gbs) best_gbs = gbs;
        CHECK(hipEventRecord(e0));
        CHECK(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0));
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double g2 = 2.0 * bytes / (ms * 1e-3) / 1e9;
        if (rep > 0 && g2 > best_memcpy) best_memcpy = g2;
    }
    // ---- grid barrier ----
    double barrier_us[2] = {0, 0};
    {
        unsigned* counter;
        float* data;
        CHECK(hipMalloc(&counter, 4));
        CHECK(hipMalloc(&data, (size_t)cus * 2 * 256 * sizeof(float)));
        CHECK(hipMemset(data, 0, (size_t)cus * 2 * 256 * sizeof(float)));
        for (int k = 0; k < 2; ++k) {
            const int nwg = cus * (k + 1), rounds = 200;
            float best = 1e30f;
            for (int rep = 0; rep < 3; ++rep) {
                CHECK(hipMemset(counter, 0, 4));
                CHECK(hipEventRecord(e0));
                hipLaunchKernelGGL(grid_barrier_loop, dim3(nwg), dim3(256), 0, 0, counter, data, rounds, nwg);
                CHECK(hipEventRecord(e1));
                CHECK(hipEventSynchronize(e1));
                float ms;
                CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (ms < best) best = ms;
            }
            barrier_us[k] = best * 1e3 / rounds;
        }
    }
    printf("{\"grid_barrier_us_1wg_per_cu\": %.2f, \"grid_barrier_us_2wg_per_cu\": %.2f, ", barrier_us[0], barrier_us[1]);
    printf("\"device\": \"%s\", \"compute_units\": %d, \"clock_mhz\": %d, \"mfma_f32_16x16x4_tflops\": %.1f, "
           "\"stream_copy_GBps\": %.0f, \"hipMemcpyDtoD_GBps\": %.0f, \"note\": \"read+write bytes counted; best of 5\"}\n",
           prop.name, cus, prop.clockRate / 1000, best_tf, best_gbs, best_memcpy);
    return 0;
}
