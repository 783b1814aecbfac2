// Time-major MFMA convolution for FEW channels (16 or 32 in and out): the last two resolutions of the NSF-HiFiGAN
// generator run 36 k in {3,7,11} convolutions over 256 k / 512 k frames with 32 / 16 channels.  On the GEMM family of
// gemm.hip (output channels = MFMA rows, 64-row tiles) a 16-channel layer fills a quarter of every tile and a
// workgroup's K walk is a handful of steps - those two stages ran at 5-12 % of either roofline.  Here the roles are
// swapped: TIME is the MFMA row dimension and the output channels are the 16 columns of a 16x16x4 block,
//     D[t][o] += A[t][k] * B[k][o],   k = (tap, input channel),
// so no MFMA lane is wasted whatever the channel count, a workgroup owns 256 frames x all channels, and the
// (tiny) weights sit in LDS as ready-made B fragments for the whole kernel.
//   A fragment (lane: row t = lane & 15, k = lane >> 4): one ds_read_b32 per 16-frame block from the staged input
//     tile [CI][256 + halo] (row stride = 16 mod 32 floats: the 4 k-rows x 16 frames hit 64 distinct banks)
//   B fragment (lane: k = lane >> 4, column o = lane & 15): one conflict-free ds_read_b32 from the fragment array
//   D (row = (lane >> 4) * 4 + reg, column = lane & 15) goes through an LDS [CO][256] tile so that the global
//     stores are 16-byte pieces of a channel row, with bias / leaky-ReLU / tanh / residual applied row-major.
#include "dsd_internal.h"

namespace dsd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TC_TT = 256;          // frames per workgroup (4 waves x 4 blocks of 16)

template <int CI, int CO>
__global__ __launch_bounds__(256) void tconv_kernel(const TConvP p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NBN = CO / 16;                   // 16-column output blocks
    constexpr int KC4 = CI / 4;                    // k4 steps per tap
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.y;
    const int t0 = blockIdx.x * TC_TT;
    const int HP = p.HP, SP = p.SP;                // halo (multiple of 4) and LDS row stride
    float* wfrag = lds;                            // [taps][KC4][NBN][64]
    float* tile = lds + p.taps * KC4 * NBN * 64;   // [CI][SP]
    // ---- weights: already in fragment order, straight copy ----
    const int nw = p.taps * KC4 * NBN * 64;
    for (int i = tid * 4; i < nw; i += 1024) *reinterpret_cast<f32x4*>(wfrag + i) = *reinterpret_cast<const f32x4*>(p.W + i);
    // ---- input tile: frames [t0 - HP, t0 + TT + HP), leaky ReLU on load, zero outside [0, T) ----
    const float* xb = p.x + (long)b * p.x_bstride;
    const int w4 = (TC_TT + 2 * HP) / 4;
    for (int i = tid; i < CI * w4; i += 256) {
        const int c = i / w4, q = i - c * w4;
        const int t = t0 - HP + q * 4;
        f32x4 v = f32x4{0.f, 0.f, 0.f, 0.f};
        if (t >= -3 && t < p.T) {                  // some element of the float4 may be valid (rows are 16-byte aligned)
            const f32x4 g = *reinterpret_cast<const f32x4*>(xb + (long)c * p.x_rstride + t);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float y = g[e];
                if (p.slope_in != 1.f) y = y >= 0.f ? y : y * p.slope_in;
                v[e] = (t + e >= 0 && t + e < p.T) ? y : 0.f;
            }
        }
        *reinterpret_cast<f32x4*>(tile + c * SP + q * 4) = v;
    }
    __syncthreads();
    // ---- K walk: (tap, 4-channel group); 4 time blocks x NBN column blocks of accumulators per wave ----
    f32x4 acc[4][NBN];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < NBN; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int trow = lane & 15, kk = lane >> 4;
    const int half = p.taps / 2;
    // frame of (wave, block m, row trow) inside the tile, before the tap offset
    const float* abase = tile + kk * SP + HP + wave * 64 + trow;
    for (int tap = 0; tap < p.taps; ++tap) {
        const float* ap = abase + (tap - half) * p.dil;
        const float* wp = wfrag + (tap * KC4) * NBN * 64 + lane;
#pragma unroll
        for (int c4 = 0; c4 < KC4; ++c4) {
            float bv[NBN], av[4];
#pragma unroll
            for (int n = 0; n < NBN; ++n) bv[n] = wp[(c4 * NBN + n) * 64];
#pragma unroll
            for (int m = 0; m < 4; ++m) av[m] = ap[(c4 * 4) * SP + m * 16];
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < NBN; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bv[n], acc[m][n], 0, 0, 0);
        }
    }
    __syncthreads();                               // everyone is done with the input tile: reuse it for the output
    // ---- D -> LDS [CO][TT + 4] ----
    constexpr int ES = TC_TT + 4;
    float* otile = tile;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < NBN; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) otile[(n * 16 + trow) * ES + wave * 64 + m * 16 + kk * 4 + r] = acc[m][n][r];
    __syncthreads();
    // ---- row-major epilogue: bias, activation, residual, 16-byte stores ----
    float* ob = p.out + (long)b * p.o_bstride;
    const float* rb = p.res ? p.res + (long)b * p.o_bstride : nullptr;
    for (int i = tid; i < p.co_real * (TC_TT / 4); i += 256) {
        const int o = i / (TC_TT / 4), q = i - o * (TC_TT / 4);
        const int t = t0 + q * 4;
        if (t >= p.Ts_out) continue;
        f32x4 v = *reinterpret_cast<const f32x4*>(otile + o * ES + q * 4);
        const float bo = p.bias[o];
        f32x4 r4 = f32x4{0.f, 0.f, 0.f, 0.f};
        if (rb) r4 = *reinterpret_cast<const f32x4*>(rb + (long)o * p.o_rstride + t);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float y = v[e] + bo;
            if (p.act == ACT_LRELU) y = y >= 0.f ? y : y * 0.1f;
            else if (p.act == ACT_TANH) y = tanhf(y);
            if (rb) y = y + r4[e];
            v[e] = y;
        }
        *reinterpret_cast<f32x4*>(ob + (long)o * p.o_rstride + t) = v;
    }
}

int tconv_lds_bytes(int ci, int co, int taps, int SP) {
    const int in_tile = ci * SP, out_tile = co * (TC_TT + 4);
    return (taps * ci * co + (in_tile > out_tile ? in_tile : out_tile)) * 4;
}

template <int CI, int CO>
static hipError_t tconv_go(const TConvP& p, int batch, hipStream_t st) {
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(tconv_kernel<CI, CO>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set = true;
    }
    hipLaunchKernelGGL((tconv_kernel<CI, CO>), dim3((p.T + TC_TT - 1) / TC_TT, batch), dim3(256), p.lds_bytes, st, p);
    return hipGetLastError();
}

hipError_t tconv_init_all() {      // raise the dynamic-LDS limits once, outside any stream capture
    hipError_t e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(tconv_kernel<16, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(tconv_kernel<32, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) != hipSuccess) return e;
    return hipSuccess;
}

hipError_t launch_tconv(const TConvP& p, int ci, int co, int batch, hipStream_t st) {
    if (ci == 16 && co == 16) return tconv_go<16, 16>(p, batch, st);
    if (ci == 32 && co == 32) return tconv_go<32, 32>(p, batch, st);
    return hipErrorInvalidValue;
}

}  // namespace dsd
