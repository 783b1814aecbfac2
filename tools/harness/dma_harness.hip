// Diagnostic harness (GPU box): semantics of the LDS-direct 16-byte buffer load on gfx950 (buffer_load_dwordx4 ... lds), as
// lynx_x3.hip's lx_x3w_kernel uses it - where each lane's 16 bytes land, and what makes them visible (vmcnt, barrier).
//   hipcc -O2 --offload-arch=gfx950 tools/harness/dma_harness.hip -o tools/harness/dma_harness.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4 rsrc_words(const void* ptr) {
    const unsigned long long a = (unsigned long long)ptr;
    return i32x4{__builtin_amdgcn_readfirstlane((int)(unsigned)a), __builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xffffu)),
                 (int)0x7FFFFFF0u, 0x00020000};
}
__device__ __forceinline__ void dma_b128(i32x4 rs, unsigned lds_byte, int voff, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_byte), "v"(voff), "s"(rs), "s"(soff) : "memory");
}
// in: [rows][64] floats per workgroup tile of 256 rows; mode 0: wait vmcnt(0) + barrier; mode 1: barrier only
template <int MODE>
__global__ __launch_bounds__(256, 1) void k(const float* in, float* out, const float* other, float* sink) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane >> 4, lcol = lane & 15;
    const float* src = in + (size_t)blockIdx.x * 256 * 64;
    const i32x4 rs = rsrc_words(src);
    // MODE 2 / 3: the destination is the SECOND 64 KiB of a 128 KiB allocation (M0 must carry more than 16 address bits);
    // the first 64 KiB holds a sentinel that must survive
    constexpr unsigned HI = MODE >= 2 ? 65536u : 0u;
    if (MODE >= 2) {
        for (int i = tid; i < 256 * 64 / 4; i += 256) reinterpret_cast<f32x4*>(lds)[i] = f32x4{-7.f, -7.f, -7.f, -7.f};
        __syncthreads();
    }
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)lds + HI;
    // some ordinary loads in flight around the LDS-direct ones
    f32x4 o0 = *reinterpret_cast<const f32x4*>(other + (size_t)(blockIdx.x * 256 + tid) * 4);
    for (int j = 0; j < 16; ++j) {
        const int s_lds = __builtin_amdgcn_readfirstlane((int)lds0 + (64 * wave + 4 * j) * 256);
        dma_b128(rs, (unsigned)s_lds, ((64 * wave + lrow) * 64 + lcol * 4) * 4, __builtin_amdgcn_readfirstlane(4 * j * 64 * 4));
    }
    f32x4 o1 = *reinterpret_cast<const f32x4*>(other + (size_t)(blockIdx.x * 256 + tid) * 4 + 1024 * 1024);
    if (MODE != 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float* dst = out + (size_t)blockIdx.x * 256 * 64;
    if (MODE == 3) {        // the sentinel half instead
        for (int i = tid; i < 256 * 64 / 4; i += 256) reinterpret_cast<f32x4*>(dst)[i] = reinterpret_cast<const f32x4*>(lds)[i];
    } else {
        for (int i = tid; i < 256 * 64 / 4; i += 256) reinterpret_cast<f32x4*>(dst)[i] = reinterpret_cast<const f32x4*>(lds + HI / 4)[i];
    }
    sink[blockIdx.x * 256 + tid] = o0[0] + o1[1];
}
int main() {
    const int nwg = 1024;
    const size_t n = (size_t)nwg * 256 * 64;
    std::vector<float> h(n), r(n);
    for (size_t i = 0; i < n; ++i) h[i] = (float)(i % 1000003);
    float *in, *out, *other, *sink;
    hipMalloc(&in, n * 4); hipMalloc(&out, n * 4); hipMalloc(&other, (size_t)8 << 20 << 2); hipMalloc(&sink, nwg * 256 * 4);
    hipMemcpy(in, h.data(), n * 4, hipMemcpyHostToDevice);
    hipMemset(other, 0, (size_t)8 << 20 << 2);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<0>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    for (int mode = 0; mode < 4; ++mode)
        for (int it = 0; it < 2; ++it) {
            hipMemset(out, 0xff, n * 4);
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(nwg), dim3(256), 64 * 1024, 0, in, out, other, sink);
            else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(nwg), dim3(256), 64 * 1024, 0, in, out, other, sink);
            else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(nwg), dim3(256), 128 * 1024, 0, in, out, other, sink);
            else hipLaunchKernelGGL(k<3>, dim3(nwg), dim3(256), 128 * 1024, 0, in, out, other, sink);
            hipError_t e = hipDeviceSynchronize();
            hipMemcpy(r.data(), out, n * 4, hipMemcpyDeviceToHost);
            size_t bad = 0, first = 0;
            for (size_t i = 0; i < n; ++i)
                if (r[i] != (mode == 3 ? -7.f : h[i])) { if (!bad) first = i; ++bad; }
            const char* names[4] = {"vmcnt(0) + barrier", "barrier only", "destination = second 64 KiB", "second 64 KiB: the first must keep its sentinel"};
            printf("mode %d (%s) run %d: %s, %zu of %zu floats differ", mode, names[mode], it, hipGetErrorString(e), bad, n);
            if (bad) printf("; first at wg %zu row %zu col %zu: got %g want %g", first / 16384, first % 16384 / 64, first % 64, r[first], h[first]);
            printf("\n");
        }
    return 0;
}
