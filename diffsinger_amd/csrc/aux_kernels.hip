// Non-GEMM kernels of the denoiser path (gfx950): layout conversion at the boundary, the sinusoidal
// step embedding, LYNXNet's LayerNorm statistics and its depthwise k=31 convolution.
// All of them are HBM/L2-bound streaming kernels: 64-wide wavefronts run along the time axis so
// that every global access is a contiguous 256-B row segment.
#include "dsd_internal.h"

namespace dsd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// pack: caller tensor (element (b, r, t) at src[b*sb + r*sr + t*st]) -> internal [B][R][Ts].
// A 32x32 tile goes through LDS so that both the [B,R,T] (st == 1) and the [B,T,R] (sr == 1)
// caller layouts are read with their contiguous axis on the lanes.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_kernel(const float* __restrict__ src, long sb, long sr, long st,
                                                   float* __restrict__ dst, int R, int T, int Ts) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int r0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;   // 32 x 8
    const float* s = src + (long)b * sb;
    if (st == 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = r0 + ly + i * 8, t = t0 + lx;
            tile[ly + i * 8][lx] = (r < R && t < T) ? s[(long)r * sr + t] : 0.f;
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int t = t0 + ly + i * 8, r = r0 + lx;
            tile[lx][ly + i * 8] = (r < R && t < T) ? s[(long)r * sr + (long)t * st] : 0.f;
        }
    }
    __syncthreads();
    float* d = dst + (long)b * R * Ts;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ly + i * 8, t = t0 + lx;
        if (r < R && t < Ts) d[(long)r * Ts + t] = tile[ly + i * 8][lx];
    }
}

hipError_t launch_pack(const float* src, long sb, long sr, long st, float* dst, int B, int R, int T, int Ts,
                       hipStream_t stream) {
    dim3 grid((round_up(T, 32)) / 32, (R + 31) / 32, B);
    hipLaunchKernelGGL(pack_kernel, grid, dim3(256), 0, stream, src, sb, sr, st, dst, R, T, Ts);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// unpack: internal [B][F*M][Ts] -> caller [B,F,M,T] (transpose == 0) or [B,F,T,M] with the
// per-bin affine of denorm_spec (ddpm.py:350,382-383): out = x * scale[f*M+m] + shift[f*M+m].
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void unpack_kernel(const float* __restrict__ src, int Ts, float* __restrict__ dst,
                                                     int F, int M, int T, int transpose,
                                                     const float* __restrict__ scale, const float* __restrict__ shift) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z / F, f = blockIdx.z % F;
    const int m0 = blockIdx.y * 32, t0 = blockIdx.x * 32;
    const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
    const float* s = src + ((long)b * F + f) * M * Ts;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int m = m0 + ly + i * 8, t = t0 + lx;
        float v = 0.f;
        if (m < M && t < T) {
            v = s[(long)m * Ts + t];
            if (scale) v = v * scale[f * M + m];
            if (shift) v = v + shift[f * M + m];
        }
        tile[ly + i * 8][lx] = v;
    }
    __syncthreads();
    float* d = dst + ((long)b * F + f) * M * T;
    if (!transpose) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int m = m0 + ly + i * 8, t = t0 + lx;
            if (m < M && t < T) d[(long)m * T + t] = tile[ly + i * 8][lx];
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int t = t0 + ly + i * 8, m = m0 + lx;
            if (m < M && t < T) d[(long)t * M + m] = tile[lx][ly + i * 8];
        }
    }
}

hipError_t launch_unpack(const float* src, int Ts, float* dst, int B, int F, int M, int T, int transpose,
                         const float* scale, const float* shift, hipStream_t stream) {
    dim3 grid((T + 31) / 32, (M + 31) / 32, B * F);
    hipLaunchKernelGGL(unpack_kernel, grid, dim3(256), 0, stream, src, Ts, dst, F, M, T, transpose, scale, shift);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// SinusoidalPosEmb (common_layers.py:273-280) for `ncols` step values at once:
//   dst[i][col] = sin(t[col] * f_i),  dst[C/2 + i][col] = cos(t[col] * f_i)
// written as a [C][colstride] matrix (columns = steps), i.e. already the B operand of the MLP GEMM.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sinemb_kernel(const float* __restrict__ t, int ncols, int colstride,
                                                     const float* __restrict__ freqs, int C, float* __restrict__ dst) {
    const int half = C >> 1;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= half * colstride) return;
    const int i = idx / colstride, col = idx - i * colstride;
    float s = 0.f, c = 0.f;
    if (col < ncols) {
        const float arg = t[col] * freqs[i];     // fp32 product first, as the reference does
        s = sinf(arg);
        c = cosf(arg);
    }
    dst[(long)i * colstride + col] = s;
    dst[(long)(half + i) * colstride + col] = c;
}

// dst[c][r] = src[r][c] for r < rows, c < cols (32 x 32 tiles through LDS): the step table D [L*C rows][steps] -> Dt
// [steps][L*C], so that a layer's FiLM vector for one step is L*C / ... contiguous floats instead of one cache line per row
__global__ __launch_bounds__(256) void transpose_kernel(const float* __restrict__ src, int rows, int cols, int src_stride,
                                                        float* __restrict__ dst, int dst_stride) {
    __shared__ float tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = r0 + ty + 8 * i, c = c0 + tx;
        tile[ty + 8 * i][tx] = (r < rows && c < cols) ? src[(long)r * src_stride + c] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, r = r0 + tx;
        if (c < cols && r < rows) dst[(long)c * dst_stride + r] = tile[tx][ty + 8 * i];
    }
}

hipError_t launch_transpose(const float* src, int rows, int cols, int src_stride, float* dst, int dst_stride, hipStream_t st) {
    hipLaunchKernelGGL(transpose_kernel, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(256), 0, st, src, rows, cols, src_stride, dst,
                       dst_stride);
    return hipGetLastError();
}

hipError_t launch_sinemb(const float* t_dev, int ncols, int colstride, const float* freqs, int C, float* dst,
                         hipStream_t stream) {
    const int n = (C / 2) * colstride;
    hipLaunchKernelGGL(sinemb_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, t_dev, ncols, colstride, freqs, C,
                       dst);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// LYNXNet layer prologue (lynxnet.py:76-84) + LayerNorm statistics (lynxnet.py:53, :151):
//   strong_cond:   x <- x + cp ;  xin = x + d          (residual taken AFTER the conditioner add)
//   otherwise:     xin = x + cp + d                    (residual = untouched x)
//   stats[b][0][t] = mean_c xin,  stats[b][1][t] = 1/sqrt(var_c xin + eps)   (biased variance)
// One workgroup = 16 waves owns 64 frames x all channels; lanes run along time (256-B row segments), wave w
// takes channels w, w+16, ...  Up to 64 channels per wave stay in registers, so x / cp are read ONCE and the
// centred second moment (two-pass, like torch's LayerNorm) needs no second trip to memory; loads are issued
// 8 deep.  The per-(b,t) reduction over channels is a register loop + a 16-wave LDS combine.
// ---------------------------------------------------------------------------------------------
constexpr int LP_WAVES = 16;
constexpr int LP_REG = 64;      // channels per lane kept in registers; the rest is re-read
// FT = frames per workgroup: 64 (one lane per frame, wave w takes channels w, w+16, ...) or 16 (a wave covers
// 16 frames x 4 channel sub-groups: 4x the workgroups for small B*T, where 64-frame tiles leave most CUs idle -
// B=1, T=1000 is 16 workgroups otherwise)
template <int FT>
__global__ __launch_bounds__(1024) void lynx_pre_kernel(float* __restrict__ x, float* __restrict__ xin,
                                                        const float* __restrict__ cp, long cp_bstride,
                                                        const float* __restrict__ film, int film_cstride,
                                                        int film_col0, int film_colb, long bstride, int rstride, int C,
                                                        int T, int strong, float* __restrict__ stats, int ts,
                                                        float eps) {
    constexpr int CG = 64 / FT;                  // channel sub-groups inside a wave
    constexpr int NPART = LP_WAVES * CG;         // partial sums per frame
    __shared__ float red[NPART][FT];
    __shared__ float mean_s[FT];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int f = lane % FT, part = wave * CG + lane / FT;
    const int t = blockIdx.x * FT + f;
    float* xb = x + (long)b * bstride + t;
    float* xi = xin ? xin + (long)b * bstride + t : nullptr;
    const float* cpb = cp ? cp + (long)b * cp_bstride + t : nullptr;
    const int nper = (C + NPART - 1) / NPART;               // channels per lane
    float keep[LP_REG];
    float sum = 0.f;
#pragma unroll
    for (int i0 = 0; i0 < LP_REG; i0 += 8) {
        if (i0 < nper) {                                     // uniform: skips the unused part of the register window
            float a[8], c8[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = min(part + (i0 + j) * NPART, C - 1);        // clamped: branch-free loads
                a[j] = xb[(long)c * rstride];
                c8[j] = cpb ? cpb[(long)c * rstride] : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = part + (i0 + j) * NPART;
                const bool ok = (i0 + j) < nper && c < C;
                float v = a[j] + c8[j];
                if (ok && cpb && strong) xb[(long)c * rstride] = v;
                if (film && ok) v = v + film[(long)c * film_cstride + film_col0 + b * film_colb];
                if (ok && xi) xi[(long)c * rstride] = v;
                keep[i0 + j] = ok ? v : 0.f;
                sum += ok ? v : 0.f;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) keep[i0 + j] = 0.f;
        }
    }
    for (int i = LP_REG; i < nper; ++i) {                    // beyond the register window
        const int c = part + i * NPART;
        if (c < C) {
            float v = xb[(long)c * rstride];
            if (cpb) {
                v += cpb[(long)c * rstride];
                if (strong) xb[(long)c * rstride] = v;
            }
            if (film) v = v + film[(long)c * film_cstride + film_col0 + b * film_colb];
            if (xi) xi[(long)c * rstride] = v;
            sum += v;
        }
    }
    red[part][f] = sum;
    __syncthreads();
    if (part == 0) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < NPART; ++w) s += red[w][f];
        mean_s[f] = s / (float)C;
    }
    __syncthreads();
    const float mean = mean_s[f];
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < LP_REG; ++i) {
        const bool ok = i < nper && (part + i * NPART) < C;
        const float dlt = keep[i] - mean;
        sq += ok ? dlt * dlt : 0.f;
    }
    const float* rd = xi ? xi : xb;
    for (int i = LP_REG; i < nper; ++i) {
        const int c = part + i * NPART;
        if (c < C) {
            const float dlt = rd[(long)c * rstride] - mean;
            sq += dlt * dlt;
        }
    }
    __syncthreads();
    red[part][f] = sq;
    __syncthreads();
    if (part == 0 && t < ts) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < NPART; ++w) s += red[w][f];
        float* st = stats + (long)b * 2 * ts;
        st[t] = mean;
        st[ts + t] = 1.f / sqrtf(s / (float)C + eps);
    }
}

hipError_t launch_lynx_pre(float* x, float* xin, const float* cp, long cp_bstride, const float* film,
                           int film_cstride, int film_col0, int film_colb, long bstride, int rstride, int C, int B,
                           int T, int strong, float* stats, int ts, float eps, hipStream_t stream) {
    const int tiles64 = round_up(T, 64) / 64;
    if ((long)tiles64 * B >= 256) {
        dim3 grid(tiles64, B);
        hipLaunchKernelGGL(lynx_pre_kernel<64>, grid, dim3(64 * LP_WAVES), 0, stream, x, xin, cp, cp_bstride, film,
                           film_cstride, film_col0, film_colb, bstride, rstride, C, T, strong, stats, ts, eps);
    } else {
        dim3 grid(round_up(T, 16) / 16, B);
        hipLaunchKernelGGL(lynx_pre_kernel<16>, grid, dim3(64 * LP_WAVES), 0, stream, x, xin, cp, cp_bstride, film,
                           film_cstride, film_col0, film_colb, bstride, rstride, C, T, strong, stats, ts, eps);
    }
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Depthwise Conv1d(k, padding=k//2, groups=C) + activation (lynxnet.py:57-58):
//   dst[c, t] = act( bias[c] + sum_j w[c, j] * src[c, t + j - k/2] ),   zero padded on [0, T)
// One workgroup = 4 channels x 256 frames; each wave stages its channel's (256 + k - 1)-frame row
// in LDS (masked to [0, T)) and every lane produces 4 consecutive frames from registers.
// ---------------------------------------------------------------------------------------------
constexpr int DW_TT = 256;
constexpr int DW_MAXK = 63;
// KS > 0: kernel size known at compile time - the lane's (KS + 3)-float input window is pulled from LDS with
// aligned 16-byte reads into registers once and the taps run on registers (k = 31: 9 LDS reads per lane instead
// of 124); KS == 0: any odd k <= 63, taps read from LDS one by one.
template <int KS>
__global__ __launch_bounds__(256) void dwconv_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                     long bstride, int rstride, int C, int T, const int* __restrict__ lens,
                                                     const float* __restrict__ w, const float* __restrict__ bias,
                                                     int ksz_rt, int act, const float* __restrict__ prelu) {
    __shared__ __attribute__((aligned(16))) float row[4][DW_TT + 64];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.z;
    const int Tb = lens ? lens[b] : T;                   // ragged batch: zero padding starts at this item's own length
    const int c = blockIdx.y * 4 + wave;                 // wave-uniform: weights and bias are scalar loads
    const int t0 = blockIdx.x * DW_TT;
    const int ksz = KS > 0 ? KS : ksz_rt;
    const int pad = ksz >> 1;
    if (c < C) {
        const float* s = src + (long)b * bstride + (long)c * rstride;
        for (int i = lane; i < DW_TT + 64; i += 64) {
            const int t = t0 - pad + i;
            row[wave][i] = (t >= 0 && t < Tb && i < DW_TT + ksz - 1) ? s[t] : 0.f;
        }
    }
    __syncthreads();
    if (c >= C) return;
    float acc[4];
    const float bv = bias[c];
#pragma unroll
    for (int e = 0; e < 4; ++e) acc[e] = bv;
    const float* wc = w + (long)c * ksz;
    const float* r = &row[wave][lane * 4];
    if constexpr (KS > 0) {
        constexpr int NW = (KS + 3 + 3) / 4 * 4;         // window, rounded up to whole float4s (stays inside the row)
        float win[NW];
#pragma unroll
        for (int i = 0; i < NW; i += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(&r[i]);
            win[i] = v[0]; win[i + 1] = v[1]; win[i + 2] = v[2]; win[i + 3] = v[3];
        }
#pragma unroll
        for (int j = 0; j < KS; ++j) {
            const float wj = wc[j];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += wj * win[j + e];
        }
    } else {
        for (int j = 0; j < ksz; ++j) {
            const float wj = wc[j];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += wj * r[j + e];
        }
    }
    float* d = dst + (long)b * bstride + (long)c * rstride + t0 + lane * 4;
    const float slope = (act == 0) ? prelu[c] : 0.f;
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float v = acc[e];
        if (act == 0) v = v >= 0.f ? v : v * slope;                 // PReLU(per channel)
        else if (act == 1) v = v * (1.f / (1.f + expf(-v)));        // SiLU
        else if (act == 2) v = fmaxf(v, 0.f);                       // ReLU
                                                                    // act == 3: none (ConvNeXt dwconv, convnext.py:42)
        o[e] = v;
    }
    if (t0 + lane * 4 < rstride - 3) *reinterpret_cast<f32x4*>(d) = o;
}

// Row-streaming form for the kernel sizes the models use (31: LYNXNet, 7: ConvNeXt): ONE WAVE per (item, channel) row walks
// the row in 256-frame segments.  A segment's 320-float window [t0 - 16, t0 + 304) arrives as 16-byte buffer loads whose
// range check IS the zero padding (the descriptor covers exactly the item's valid frames: offsets before 0 wrap around and
// offsets from Tb on read as 0, per dword), goes into the wave's own LDS row (two rows in turn: no workgroup barrier, the
// four waves of a workgroup never meet), and the NEXT segment's loads are in flight while this one's taps run from
// registers.  The first form above launches 16 k four-row workgroups of one segment each, stages with 4-byte loads and
// meets at a barrier: 36.9 us per LYNXNet layer at B = 8 (131 MB at 3.5 TB/s) against this one's.
template <int KS>
__global__ __launch_bounds__(256) void dwconv_rows_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                          long bstride, int rstride, int C, int T, int nrows,
                                                          const int* __restrict__ lens, const float* __restrict__ w,
                                                          const float* __restrict__ bias, int act,
                                                          const float* __restrict__ prelu) {
    constexpr int SEG = DW_TT, WIN = SEG + 64;                   // staged floats per segment (whole float4s, 16 before t0)
    constexpr int OFF = 16 - KS / 2;                             // first tap of frame t0 + 4 lane sits at staged index 4 lane + OFF
    constexpr int OA = OFF & ~3, O2 = OFF & 3;                   // ... read as aligned float4s from 4 lane + OA, taps from O2 in
    constexpr int NW = (O2 + KS + 3 + 3) / 4 * 4;                // the lane's window, whole float4s
    static_assert(KS / 2 <= 16 && 4 * 63 + OA + NW <= WIN, "window must stay inside the staged row");
    __shared__ __attribute__((aligned(16))) float rows[4][2][WIN];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int row = blockIdx.x * 4 + wave;                       // (item, channel), wave-uniform
    if (row >= nrows) return;                                    // (no workgroup barrier below)
    const int b = row / C, c = row - b * C;
    const int Tb = lens ? lens[b] : T;
    const float* s = src + (long)b * bstride + (long)c * rstride;
    // the descriptor's range = the valid frames: everything outside reads as zero = the convolution's zero padding
    const __amdgpu_buffer_rsrc_t r_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(s), 0, Tb * 4, 0x00020000);
    float* d = dst + (long)b * bstride + (long)c * rstride;
    const int nseg = (T + SEG - 1) / SEG;
    float wr[KS];                                                // the row's taps, bias and slope: scalar loads
#pragma unroll
    for (int j = 0; j < KS; ++j) wr[j] = w[(long)c * KS + j];
    const float bv = bias[c];
    const float slope = (act == 0) ? prelu[c] : 0.f;
    auto fetch = [&](int seg, f32x4& a, f32x4& e) {              // 80 float4 per segment: one per lane + one for lanes 0-15
        const int base = (seg * SEG - 16) * 4;
        a = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_s, base + lane * 16, 0, 0));
        e = lane < 16 ? __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_s, base + (64 + lane) * 16, 0, 0))
                      : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    f32x4 na, ne;
    fetch(0, na, ne);
    for (int seg = 0; seg < nseg; ++seg) {
        float* rw = rows[wave][seg & 1];
        *reinterpret_cast<f32x4*>(&rw[lane * 4]) = na;
        if (lane < 16) *reinterpret_cast<f32x4*>(&rw[(64 + lane) * 4]) = ne;
        if (seg + 1 < nseg) fetch(seg + 1, na, ne);              // in flight under this segment's taps
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float win[NW];
#pragma unroll
        for (int i = 0; i < NW; i += 4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(&rw[lane * 4 + OA + i]);
            win[i] = v[0]; win[i + 1] = v[1]; win[i + 2] = v[2]; win[i + 3] = v[3];
        }
        float acc[4] = {bv, bv, bv, bv};
#pragma unroll
        for (int j = 0; j < KS; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] += wr[j] * win[O2 + j + e];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float v = acc[e];
            if (act == 0) v = v >= 0.f ? v : v * slope;
            else if (act == 1) v = v * (1.f / (1.f + expf(-v)));
            else if (act == 2) v = fmaxf(v, 0.f);
            o[e] = v;
        }
        const int t = seg * SEG + lane * 4;
        if (t < rstride - 3) *reinterpret_cast<f32x4*>(&d[t]) = o;
    }
}

hipError_t launch_dwconv(const float* src, float* dst, long bstride, int rstride, int C, int B, int T, const int* lens,
                         const float* w, const float* bias, int ksz, int act, const float* prelu,
                         hipStream_t stream) {
    if (ksz > DW_MAXK || ksz < 1 || ksz % 2 == 0) return hipErrorInvalidValue;
    const int rows_env = path_opts().dwconv_rows;      // 0: the first form (A/B)
    if (rows_env != 0 && (ksz == 31 || ksz == 7)) {
        const int nrows = B * C;
        if (ksz == 31)
            hipLaunchKernelGGL(dwconv_rows_kernel<31>, dim3((nrows + 3) / 4), dim3(256), 0, stream, src, dst, bstride, rstride, C, T,
                               nrows, lens, w, bias, act, prelu);
        else
            hipLaunchKernelGGL(dwconv_rows_kernel<7>, dim3((nrows + 3) / 4), dim3(256), 0, stream, src, dst, bstride, rstride, C, T,
                               nrows, lens, w, bias, act, prelu);
        return hipGetLastError();
    }
    dim3 grid((T + DW_TT - 1) / DW_TT, (C + 3) / 4, B);
    if (ksz == 31)
        hipLaunchKernelGGL(dwconv_kernel<31>, grid, dim3(256), 0, stream, src, dst, bstride, rstride, C, T, lens, w, bias, ksz,
                           act, prelu);
    else if (ksz == 7)
        hipLaunchKernelGGL(dwconv_kernel<7>, grid, dim3(256), 0, stream, src, dst, bstride, rstride, C, T, lens, w, bias, ksz,
                           act, prelu);
    else
        hipLaunchKernelGGL(dwconv_kernel<0>, grid, dim3(256), 0, stream, src, dst, bstride, rstride, C, T, lens, w, bias, ksz,
                           act, prelu);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// LayerNorm statistics from per-tile partials (EP_LYNX_NEXT writes, per 64-row tile i: n_i rows, mean_i, M2_i = sum of
// squared deviations from mean_i).  Parallel-variance merge:  mean = sum n_i mean_i / C,
// M2 = sum M2_i + sum n_i (mean_i - mean)^2;  stats[b][0][t] = mean, stats[b][1][t] = 1 / sqrt(M2 / C + eps).
// ---------------------------------------------------------------------------------------------
// One workgroup = 16 frames x 16 tile slots: every thread fetches ONE partial pair (a single load round, coalesced
// along the frames), the tiles are combined through LDS.
__global__ __launch_bounds__(256) void ln_merge_kernel(const float* __restrict__ lnpart, int mtiles, int C, int T, int ts,
                                                       float eps, float* __restrict__ stats) {
    __shared__ float sm[16][17], sq[16][17], mean_s[16];
    const int b = blockIdx.y;
    const int f = threadIdx.x & 15, slot = threadIdx.x >> 4;
    const int t = min(blockIdx.x * 16 + f, ts - 1);
    const float* lp = lnpart + (long)b * mtiles * 2 * ts + t;
    float wsum = 0.f;                              // sum over this slot's tiles of n_i * mean_i
    for (int i = slot; i < mtiles; i += 16) wsum += (float)min(64, C - i * 64) * lp[(long)i * 2 * ts];
    sm[slot][f] = wsum;
    __syncthreads();
    if (slot == 0) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += sm[k][f];
        mean_s[f] = s / (float)C;
    }
    __syncthreads();
    const float mean = mean_s[f];
    float m2 = 0.f;
    for (int i = slot; i < mtiles; i += 16) {
        const float d = lp[(long)i * 2 * ts] - mean;
        m2 += lp[(long)i * 2 * ts + ts] + (float)min(64, C - i * 64) * d * d;
    }
    sq[slot][f] = m2;
    __syncthreads();
    if (slot == 0 && blockIdx.x * 16 + f < ts) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += sq[k][f];
        float* st = stats + (long)b * 2 * ts;
        st[t] = mean;
        st[ts + t] = 1.f / sqrtf(s / (float)C + eps);
    }
}

hipError_t launch_ln_merge(const float* lnpart, int mtiles, int C, int B, int T, int ts, float eps, float* stats,
                           hipStream_t stream) {
    hipLaunchKernelGGL(ln_merge_kernel, dim3((ts + 15) / 16, B), dim3(256), 0, stream, lnpart, mtiles, C, T, ts, eps, stats);
    return hipGetLastError();
}

}  // namespace dsd
