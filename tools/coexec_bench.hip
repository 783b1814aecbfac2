// Diagnostic (never shipped): how many fp32 VALU FMAs fit in the shadow of a v_mfma_f32_16x16x4_f32 stream?
// One or two waves per SIMD, every CU busy.  Per MFMA (4 independent accumulators in rotation) N v_fma_f32 with a
// scalar (SGPR) multiplicand - the form a lanes-along-rows VALU GEMM would use - on independent accumulators.
// Build: hipcc -O3 --offload-arch=gfx950 tools/coexec_bench.hip -o /tmp/coexec ; prints cycles per MFMA and the
// combined FLOP rate relative to the MFMA-only stream.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int N, int MODE>      // MODE 0: v_fma_f32 vgpr*sgpr; 1: v_fma_f32 vgpr*vgpr; 2: v_pk_fma_f32; 3: no MFMA (VALU only, N per slot)
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* cyc, float sa_in, int iters) {
    f32x4 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = 1.0f + threadIdx.x * 1e-3f, b = 0.5f + threadIdx.x * 1e-4f;
    float c[16];
    for (int i = 0; i < 16; ++i) c[i] = (float)i;
    float2 cp[8];
    for (int i = 0; i < 8; ++i) cp[i] = make_float2((float)i, (float)-i);
    float2 ap = make_float2(a, b), bp = make_float2(b, a);
    float sa = __builtin_amdgcn_readfirstlane(sa_in);
    float sb[8];
    for (int i = 0; i < 8; ++i) sb[i] = __builtin_amdgcn_readfirstlane(sa_in + i);
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (MODE != 3)
                asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(a), "v"(b));
#pragma unroll
            for (int n = 0; n < N; ++n) {
                const int ci = (m * N + n) & 15;
                if (MODE == 0 || MODE == 3) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(c[ci]) : "v"(b), "s"(sb[n & 7]));
                if (MODE == 1) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(c[ci]) : "v"(b), "v"(a));
                if (MODE == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(cp[ci & 7]) : "v"(ap), "v"(bp));
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 16; ++i) s += c[i];
    for (int i = 0; i < 8; ++i) s += cp[i].x + cp[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int N, int MODE>
void run(const char* name, int threads, float* out, unsigned long long* cyc) {
    const int iters = 2000, nb = 256;
    hipLaunchKernelGGL((k<N, MODE>), dim3(nb), dim3(threads), 0, 0, out, cyc, 1.0001f, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<N, MODE>), dim3(nb), dim3(threads), 0, 0, out, cyc, 1.0001f, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(nb);
    hipMemcpy(h.data(), cyc, nb * 8, hipMemcpyDeviceToHost);
    double mean = 0; for (auto v : h) mean += (double)v; mean /= nb;
    const double slots = (double)iters * 8;
    const int wps = threads / 256;                       // waves per SIMD
    const double mf = MODE == 3 ? 0.0 : 2048.0, vf = (MODE == 2 ? 256.0 : 128.0) * N;
    const double flops = (mf + vf) * slots * (threads / 64) * nb;
    printf("%-34s N=%d waves/SIMD=%d: %7.2f cycles per slot per wave; chip %7.1f TFLOP/s (MFMA part %6.1f, VALU part %6.1f)\n", name, N, wps,
           mean / slots, flops / (ms * 1e-3) / 1e12, mf * slots * (threads / 64) * nb / (ms * 1e-3) / 1e12,
           vf * slots * (threads / 64) * nb / (ms * 1e-3) / 1e12);
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
#define ROW(N, M, name) run<N, M>(name, 256, out, cyc); run<N, M>(name, 512, out, cyc);
    ROW(0, 0, "MFMA only")
    ROW(2, 0, "MFMA + v_fma_f32 (v*s)")
    ROW(4, 0, "MFMA + v_fma_f32 (v*s)")
    ROW(5, 0, "MFMA + v_fma_f32 (v*s)")
    ROW(6, 0, "MFMA + v_fma_f32 (v*s)")
    ROW(7, 0, "MFMA + v_fma_f32 (v*s)")
    ROW(8, 0, "MFMA + v_fma_f32 (v*s)")
    ROW(4, 1, "MFMA + v_fma_f32 (v*v)")
    ROW(6, 1, "MFMA + v_fma_f32 (v*v)")
    ROW(2, 2, "MFMA + v_pk_fma_f32")
    ROW(3, 2, "MFMA + v_pk_fma_f32")
    ROW(8, 3, "v_fma_f32 (v*s) only")
    return 0;
}
