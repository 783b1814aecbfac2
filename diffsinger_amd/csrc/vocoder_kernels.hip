// Non-GEMM kernels of the NSF-HiFiGAN generator (mel + f0 -> waveform).  Every convolution
// with more than one input channel runs on the GEMM family of gemm.hip (transposed convolutions as phase-row GEMMs
// with a scatter epilogue); what is here is the harmonic-plus-noise source (SineGen / SourceModuleHnNSF), the
// single-input-channel strided "noise convs" that inject it at every resolution, and the residual-block average.
// Internal layout [batch][channel][Ts], lanes along time.
#include "dsd_internal.h"

namespace dsd {

// ---------------------------------------------------------------------------------------------
// SineGen._f02sine, frame level (models.py:139-142): the phase each frame starts from is the running fp32 sum of
// the per-frame phase advances wrapped into [-0.5, 0.5) - torch.cumsum's sequential order, one lane per utterance.
//   rad_last = f0 / sr * upp;  rad2 = fmod(rad_last + 0.5, 1) - 0.5;  acc[t] = fmod(sum_{t' <= t} rad2[t'], 1)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void voc_phase_kernel(const float* __restrict__ f0, int T, float sr, int upp,
                                                        float* __restrict__ acc) {
    __shared__ float buf[2048];
    const int b = blockIdx.x;
    float run = 0.f;                               // carried by thread 0 across the 2048-frame pieces
    for (int base = 0; base < T; base += 2048) {
        const int n = min(2048, T - base);
        for (int i = threadIdx.x; i < n; i += 256) {       // per-frame advance, in parallel
            const float rad_last = f0[(long)b * T + base + i] / sr * (float)upp;
            buf[i] = fmodf(rad_last + 0.5f, 1.0f) - 0.5f;
        }
        __syncthreads();
        if (threadIdx.x == 0)                              // the running sum itself is sequential: same order as torch.cumsum
            for (int i = 0; i < n; ++i) {
                run += buf[i];
                buf[i] = fmodf(run, 1.0f);
            }
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += 256) acc[(long)b * T + base + i] = buf[i];
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// SineGen.forward + SourceModuleHnNSF.forward, sample level (models.py:139-168, 200-203):
//   rad = f0/sr * (n+1) + acc[t-1];  per harmonic d: sin(2 pi (rad (d+1) + rand_ini[d])) * amp
//   sine * uv + (uv * noise_std + (1 - uv) * amp / 3) * noise  ->  tanh(Linear(dim -> 1))
// rand_ini [dim] (element 0 is forced to 0, models.py:146) and noise [B, T*upp, dim] are inputs: the reference
// draws them with torch.rand / torch.randn_like.
// ---------------------------------------------------------------------------------------------
constexpr int VOC_MAXDIM = 16;
__global__ void voc_source_kernel(const float* __restrict__ f0, const float* __restrict__ acc,
                                  const float* __restrict__ rand_ini, const float* __restrict__ noise,
                                  const float* __restrict__ lin_w, const float* __restrict__ lin_b, int T, int upp,
                                  int dim, float sr, float sine_amp, float noise_std, int Tsu, float* __restrict__ har) {
    const int b = blockIdx.y;
    const long s = (long)blockIdx.x * blockDim.x + threadIdx.x;       // sample index within the utterance
    if (s >= (long)T * upp) return;
    const int t = (int)(s / upp), n = (int)(s - (long)t * upp);
    const float f = f0[(long)b * T + t];
    float rad = f / sr * (float)(n + 1);
    if (t > 0) rad += acc[(long)b * T + t - 1];
    const float uv = f > 0.f ? 1.f : 0.f;
    const float namp = uv * noise_std + (1.f - uv) * sine_amp / 3.f;
    const float* nz = noise + ((long)b * T * upp + s) * dim;
    float m = 0.f;
    for (int d = 0; d < dim; ++d) {
        float r = rad * (float)(d + 1);
        r += (d == 0) ? 0.f : rand_ini[d];
        const float sine = sinf(6.283185307179586f * r) * sine_amp;
        m += (sine * uv + namp * nz[d]) * lin_w[d];
    }
    har[(long)b * Tsu + s] = tanhf(m + lin_b[0]);
}

// ---------------------------------------------------------------------------------------------
// noise_convs[i] (models.py:239-245, 275-276): Conv1d(1, C, 2*sf, stride sf, padding sf/2) (or k = 1 at the last
// stage) on the source, ADDED to the upsampled activations:
//   x[b][o][q] += bias[o] + sum_k w[o][k] * har[b][q*sf - sf/2 + k]        (zero outside [0, Tup))
// Threads run along the output CHANNELS (weights are stored transposed, [k][C]: coalesced), every thread produces
// NQ consecutive frames of its channel, and the source window of the workgroup's frames sits in LDS, read as a
// broadcast.  One workgroup = min(C, 256) channels x (256 / C) groups of NQ frames.
// ---------------------------------------------------------------------------------------------
constexpr int VOC_NQ = 16;
__global__ __launch_bounds__(256) void voc_noise_conv_kernel(float* __restrict__ x, const float* __restrict__ har,
                                                             const float* __restrict__ wt, const float* __restrict__ bias,
                                                             int C, int Tq, int Tsq, int sf, int ksz, long Tup, int Tsu) {
    extern __shared__ float win[];                  // (frames per workgroup - 1) * sf + ksz source samples
    const int b = blockIdx.z;
    const int cpb = C < 256 ? C : 256;              // channels per workgroup
    const int groups = 256 / cpb;                   // frame groups per workgroup
    const int fpb = groups * VOC_NQ;                // frames per workgroup
    const int q0 = blockIdx.x * fpb;
    const int o = blockIdx.y * cpb + threadIdx.x % cpb, grp = threadIdx.x / cpb;
    const int pad = ksz > 1 ? sf / 2 : 0;
    const int nwin = (fpb - 1) * sf + ksz;
    const long s0 = (long)q0 * sf - pad;
    for (int i = threadIdx.x; i < nwin; i += 256) {
        const long sidx = s0 + i;
        win[i] = (sidx >= 0 && sidx < Tup) ? har[(long)b * Tsu + sidx] : 0.f;
    }
    __syncthreads();
    if (grp >= groups || o >= C) return;
    float acc[VOC_NQ];
#pragma unroll
    for (int j = 0; j < VOC_NQ; ++j) acc[j] = 0.f;
    const float* wbase = win + (grp * VOC_NQ) * sf;
    for (int k = 0; k < ksz; ++k) {
        const float w = wt[(long)k * C + o];
#pragma unroll
        for (int j = 0; j < VOC_NQ; ++j) acc[j] += w * wbase[j * sf + k];
    }
    const float bo = bias[o];
    float* xo = x + ((long)b * C + o) * Tsq + q0 + grp * VOC_NQ;
#pragma unroll
    for (int j = 0; j < VOC_NQ; ++j)
        if (q0 + grp * VOC_NQ + j < Tq) xo[j] += acc[j] + bo;
}

// acc = first ? r : acc + r;  if scale != 1: acc /= scale      (xs accumulation and `xs / num_kernels`, models.py:280-286)
__global__ void voc_accum_kernel(float* __restrict__ acc, const float* __restrict__ r, long n, int first, float div) {
    const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    if (i >= n) return;
    f32x4_t a = *reinterpret_cast<const f32x4_t*>(r + i);
    if (!first) {
        const f32x4_t c = *reinterpret_cast<const f32x4_t*>(acc + i);
        a = c + a;
    }
    if (div != 1.f) a = a / div;
    *reinterpret_cast<f32x4_t*>(acc + i) = a;
}

// ---------------------------------------------------------------------------------------------
// Generator.fastsinegen (models.py:251-260, mini_nsf): no harmonics, no noise; the per-sample phase advance is
// interpolated linearly towards the next frame's:
//   s0 = f0 / sr;  ds0 = s0[t+1] - s0[t] (0 at the last frame);  rad(n) = s0 n + 0.5 ds0 n (n - 1) / upp,  n = 1..upp
//   frame t starts from acc[t-1] = fmod(sum_{t' < t} (fmod(rad(upp) + 0.5, 1) - 0.5), 1);   out = sin(2 pi rad)
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float voc_fast_rad(float s0, float ds0, int n, int upp) {
    const float fn = (float)n;
    return s0 * fn + 0.5f * ds0 * fn * (float)(n - 1) / (float)upp;
}

__global__ __launch_bounds__(256) void voc_fast_phase_kernel(const float* __restrict__ f0, int T, float sr, int upp,
                                                             float* __restrict__ acc) {
    __shared__ float buf[2048];
    const int b = blockIdx.x;
    float run = 0.f;
    for (int base = 0; base < T; base += 2048) {
        const int n = min(2048, T - base);
        for (int i = threadIdx.x; i < n; i += 256) {
            const int t = base + i;
            const float s0 = f0[(long)b * T + t] / sr;
            const float ds0 = t + 1 < T ? f0[(long)b * T + t + 1] / sr - s0 : 0.f;
            buf[i] = fmodf(voc_fast_rad(s0, ds0, upp, upp) + 0.5f, 1.0f) - 0.5f;
        }
        __syncthreads();
        if (threadIdx.x == 0)
            for (int i = 0; i < n; ++i) {
                run += buf[i];
                buf[i] = fmodf(run, 1.0f);
            }
        __syncthreads();
        for (int i = threadIdx.x; i < n; i += 256) acc[(long)b * T + base + i] = buf[i];
        __syncthreads();
    }
}

__global__ void voc_fast_source_kernel(const float* __restrict__ f0, const float* __restrict__ acc, int T, int upp, float sr,
                                       int Tsu, float* __restrict__ har) {
    const int b = blockIdx.y;
    const long s = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= (long)T * upp) return;
    const int t = (int)(s / upp), n = (int)(s - (long)t * upp) + 1;
    const float s0 = f0[(long)b * T + t] / sr;
    const float ds0 = t + 1 < T ? f0[(long)b * T + t + 1] / sr - s0 : 0.f;
    float rad = voc_fast_rad(s0, ds0, n, upp);
    if (t > 0) rad += acc[(long)b * T + t - 1];
    har[(long)b * Tsu + s] = sinf(6.283185307179586f * rad);
}

hipError_t launch_voc_fast_source(const float* f0, int B, int T, int upp, float source_sr, float* acc_tmp, int Tsu, float* har,
                                  hipStream_t st) {
    hipLaunchKernelGGL(voc_fast_phase_kernel, dim3(B), dim3(256), 0, st, f0, T, source_sr, upp, acc_tmp);
    const long n = (long)T * upp;
    hipLaunchKernelGGL(voc_fast_source_kernel, dim3((unsigned)((n + 255) / 256), B), dim3(256), 0, st, f0, acc_tmp, T, upp,
                       source_sr, Tsu, har);
    return hipGetLastError();
}

hipError_t launch_voc_source(const float* f0, const float* rand_ini, const float* noise, const float* lin_w,
                             const float* lin_b, int B, int T, int upp, int dim, float sr, float sine_amp, float noise_std,
                             float* acc_tmp, int Tsu, float* har, hipStream_t st) {
    if (dim > VOC_MAXDIM) return hipErrorInvalidValue;
    hipLaunchKernelGGL(voc_phase_kernel, dim3(B), dim3(256), 0, st, f0, T, sr, upp, acc_tmp);
    const long n = (long)T * upp;
    hipLaunchKernelGGL(voc_source_kernel, dim3((unsigned)((n + 255) / 256), B), dim3(256), 0, st, f0, acc_tmp, rand_ini, noise,
                       lin_w, lin_b, T, upp, dim, sr, sine_amp, noise_std, Tsu, har);
    return hipGetLastError();
}

hipError_t launch_voc_noise_conv(float* x, const float* har, const float* w, const float* bias, int B, int C, int Tq,
                                 int Tsq, int sf, int ksz, long Tup, int Tsu, hipStream_t st) {
    const int cpb = C < 256 ? C : 256, fpb = (256 / cpb) * VOC_NQ;
    const int nwin = (fpb - 1) * sf + ksz;
    hipLaunchKernelGGL(voc_noise_conv_kernel, dim3((Tq + fpb - 1) / fpb, (C + cpb - 1) / cpb, B), dim3(256),
                       nwin * sizeof(float), st, x, har, w, bias, C, Tq, Tsq, sf, ksz, Tup, Tsu);
    return hipGetLastError();
}

// x[row][t] += sigma * noise[row][t]  (models.py:272-273; x rows are padded to Ts, the caller's noise is dense)
__global__ void voc_add_noise_kernel(float* __restrict__ x, const float* __restrict__ noise, int T, int Ts, float sigma) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const long row = (long)blockIdx.z * gridDim.y + blockIdx.y;
    x[row * Ts + t] += sigma * noise[row * T + t];
}

hipError_t launch_voc_add_noise(float* x, const float* noise, int B, int C, int T, int Ts, float sigma, hipStream_t st) {
    hipLaunchKernelGGL(voc_add_noise_kernel, dim3((T + 255) / 256, C, B), dim3(256), 0, st, x, noise, T, Ts, sigma);
    return hipGetLastError();
}

hipError_t launch_voc_accum(float* acc, const float* r, long n, int first, float div, hipStream_t st) {
    hipLaunchKernelGGL(voc_accum_kernel, dim3((unsigned)((n / 4 + 255) / 256)), dim3(256), 0, st, acc, r, n, first, div);
    return hipGetLastError();
}

}  // namespace dsd
