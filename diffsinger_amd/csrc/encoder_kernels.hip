// Kernels of the FastSpeech2 acoustic encoder (the producer of the denoiser's `cond`).
// The encoder runs ONCE per utterance on T_txt phoneme tokens (tens to a few hundred), so these kernels are written
// for clarity and exactness, not throughput: everything dense goes through the GEMM family of gemm.hip; what is
// here is the glue that is not a GEMM - embedding, LayerNorm, rotary attention, padding masks, and the
// length-regulator gather that expands token states to mel frames.  Activations use the internal layout
// [batch][channel][Ls] (time innermost), lanes run along tokens / frames.
#include "dsd_internal.h"

namespace dsd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------
// mel2ph_to_dur (tts_modules.py:344-350): dur[b][k] = #{t : mel2ph[b][t] == k + 1}.  Integer atomics: exact.
// ---------------------------------------------------------------------------------------------
__global__ void enc_dur_kernel(const long long* __restrict__ mel2ph, int T, int L, int* __restrict__ dur) {
    const int b = blockIdx.y;
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const long long m = mel2ph[(long)b * T + t];
    if (m >= 1 && m <= L) atomicAdd(&dur[(long)b * L + (m - 1)], 1);
}

// ---------------------------------------------------------------------------------------------
// FastSpeech2Encoder.forward_embedding (tts_modules.py:385-398) for the rotary configuration (no additive
// positional embedding):  x[b][h][l] = (sqrt(H) * txt_embed[tok][h] + (dur_embed(dur) [+ lang_embed[lang]])) * nonpad
// also writes nonpad[b][l] = (tok != 0).
// ---------------------------------------------------------------------------------------------
__global__ void enc_embed_kernel(const long long* __restrict__ tokens, const long long* __restrict__ langs,
                                 const int* __restrict__ dur, const float* __restrict__ txt_embed, int vocab,
                                 const float* __restrict__ lang_embed, int n_lang_rows,
                                 const float* __restrict__ dur_w, const float* __restrict__ dur_b, float embed_scale,
                                 int H, int L, int Ls, float* __restrict__ x, float* __restrict__ nonpad) {
    const int b = blockIdx.z;
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    const int h = blockIdx.y;
    if (l >= L) return;
    long long tok = tokens[(long)b * L + l];
    tok = tok < 0 ? 0 : (tok >= vocab ? vocab - 1 : tok);        // torch would raise on an out-of-range index
    const float np = tok != 0 ? 1.f : 0.f;
    if (h == 0) nonpad[(long)b * Ls + l] = np;
    const float main_e = embed_scale * txt_embed[tok * H + h];
    float extra = (float)dur[(long)b * L + l] * dur_w[h] + dur_b[h];
    if (lang_embed) {
        long long lg = langs[(long)b * L + l];
        lg = lg < 0 ? 0 : (lg >= n_lang_rows ? n_lang_rows - 1 : lg);
        extra = extra + lang_embed[lg * H + h];
    }
    x[((long)b * H + h) * Ls + l] = (main_e + extra) * np;
}

// ---------------------------------------------------------------------------------------------
// LayerNorm over channels, materialised (nn.LayerNorm(H), eps 1e-5, biased variance, two-pass):
//   y[b][c][l] = ((x - mean) * rstd * g[c] + beta[c]) * (mask ? mask[b][l] : 1)
// One workgroup = 64 token columns (lanes) x 16 waves over the channels; a lane keeps its <= 32 channel values in
// registers between the two passes (C <= 512; beyond that they are re-read).
// ---------------------------------------------------------------------------------------------
constexpr int ELN_WAVES = 16;
constexpr int ELN_REG = 32;
__global__ __launch_bounds__(1024) void enc_layernorm_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                             const float* __restrict__ g, const float* __restrict__ beta,
                                                             const float* __restrict__ mask, int C, int L, int Ls,
                                                             float eps) {
    __shared__ float red[ELN_WAVES][64];
    __shared__ float stat[2][64];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int l = min(blockIdx.x * 64 + lane, Ls - 1);          // columns >= L are computed on padding and never stored
    const float* xb = x + (long)b * C * Ls + l;
    const int nper = (C + ELN_WAVES - 1) / ELN_WAVES;
    float keep[ELN_REG];
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < ELN_REG; ++i) {
        const int c = wave + i * ELN_WAVES;
        keep[i] = (i < nper && c < C) ? xb[(long)c * Ls] : 0.f;
        sum += keep[i];
    }
    for (int i = ELN_REG; i < nper; ++i) {
        const int c = wave + i * ELN_WAVES;
        if (c < C) sum += xb[(long)c * Ls];
    }
    red[wave][lane] = sum;
    __syncthreads();
    if (wave == 0) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < ELN_WAVES; ++w) s += red[w][lane];
        stat[0][lane] = s / (float)C;
    }
    __syncthreads();
    const float mean = stat[0][lane];
    float sq = 0.f;
#pragma unroll
    for (int i = 0; i < ELN_REG; ++i) {
        const int c = wave + i * ELN_WAVES;
        const float d = keep[i] - mean;
        sq += (i < nper && c < C) ? d * d : 0.f;
    }
    for (int i = ELN_REG; i < nper; ++i) {
        const int c = wave + i * ELN_WAVES;
        if (c < C) {
            const float d = xb[(long)c * Ls] - mean;
            sq += d * d;
        }
    }
    __syncthreads();
    red[wave][lane] = sq;
    __syncthreads();
    if (wave == 0) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < ELN_WAVES; ++w) s += red[w][lane];
        stat[1][lane] = 1.f / sqrtf(s / (float)C + eps);
    }
    __syncthreads();
    if (blockIdx.x * 64 + lane >= L) return;
    const float rstd = stat[1][lane];
    const float m = mask ? mask[(long)b * Ls + l] : 1.f;
    float* yb = y + (long)b * C * Ls + l;
#pragma unroll
    for (int i = 0; i < ELN_REG; ++i) {
        const int c = wave + i * ELN_WAVES;
        if (i < nper && c < C) yb[(long)c * Ls] = ((keep[i] - mean) * rstd * g[c] + beta[c]) * m;
    }
    for (int i = ELN_REG; i < nper; ++i) {
        const int c = wave + i * ELN_WAVES;
        if (c < C) yb[(long)c * Ls] = ((xb[(long)c * Ls] - mean) * rstd * g[c] + beta[c]) * m;
    }
}

// x[b][c][l] *= mask[b][l]   (EncSALayer: `x * (1 - encoder_padding_mask)` after each residual, common_layers.py:259,266)
__global__ void enc_mask_kernel(float* __restrict__ x, const float* __restrict__ mask, int C, int L, int Ls) {
    const int b = blockIdx.z, c = blockIdx.y;
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= L) return;
    x[((long)b * C + c) * Ls + l] *= mask[(long)b * Ls + l];
}

// SwiGLU of TransformerFFNLayer (common_layers.py:107-117, 134): ffn_1 produced 2 * half rows; out = rows [0, half),
// gate = rows [half, 2 half); x[b][c][l] = out * silu(gate), in place on the first half (ffn_2 reads those rows)
__global__ void enc_swiglu_kernel(float* __restrict__ x, int half, long bstride, int L, int Ls) {
    const int b = blockIdx.z, c = blockIdx.y;
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= L) return;
    float* o = x + (long)b * bstride + (long)c * Ls + l;
    const float g = o[(long)half * Ls];
    *o = *o * (g / (1.f + expf(-g)));
}

// ---------------------------------------------------------------------------------------------
// Rotary embedding on the Q and K thirds of qkv[b][3H][Ls], in place (rotary_embedding_torch.py:35-75,174-188):
// head-local channel pairs (2i, 2i+1) at position l are rotated by angle l * freqs[i]:
//   out[2i] = x[2i] cos - x[2i+1] sin,   out[2i+1] = x[2i+1] cos + x[2i] sin
// ---------------------------------------------------------------------------------------------
__global__ void enc_rope_kernel(float* __restrict__ qkv, const float* __restrict__ freqs, int H, int head_dim, int L,
                                int Ls) {
    const int b = blockIdx.z;
    const int pair = blockIdx.y;                    // over 2 (q, k) * H / 2 pairs
    const int l = blockIdx.x * blockDim.x + threadIdx.x;
    if (l >= L) return;
    const int which = pair / (H / 2);               // 0 = q, 1 = k
    const int c0 = (pair % (H / 2)) * 2;            // even channel within [0, H)
    const int i = (c0 % head_dim) / 2;
    const float ang = (float)l * freqs[i];
    const float cs = cosf(ang), sn = sinf(ang);
    float* p0 = qkv + ((long)b * 3 * H + which * H + c0) * Ls + l;
    float* p1 = p0 + Ls;
    const float x0 = *p0, x1 = *p1;
    *p0 = x0 * cs + (-x1) * sn;
    *p1 = x1 * cs + x0 * sn;
}

// ---------------------------------------------------------------------------------------------
// Scaled dot-product attention with key padding mask (common_layers.py:192-207), one wave per query:
//   s_j = (q . k_j) / sqrt(D);  s_j = -inf where key j is padding;  p = softmax(s);  o = sum_j p_j v_j
// Keys run along the lanes (k[d][j] is contiguous in j); the query's D values sit in LDS and are broadcast.
// A query whose keys are ALL padding gives NaN in the reference (softmax of -inf); it cannot occur: a padded
// batch row still has its own non-padding tokens as keys, and the output of padding queries is masked afterwards.
// ---------------------------------------------------------------------------------------------
constexpr int ATT_MAXD = 256;
constexpr int ATT_MAXCH = 32;          // key chunks of 64 per lane: L <= 2048
__global__ __launch_bounds__(256) void enc_attention_kernel(const float* __restrict__ qkv, const float* __restrict__ nonpad,
                                                            float* __restrict__ out, int H, int heads, int L, int Ls) {
    __shared__ __attribute__((aligned(16))) float qs[4][ATT_MAXD];
    __shared__ __attribute__((aligned(16))) float ps[4][ATT_MAXCH * 64];
    const int b = blockIdx.z, head = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int D = H / heads;
    const int q = blockIdx.x * 4 + wave;
    if (q >= L) return;                             // whole wave leaves together (no workgroup barrier below)
    const float* Q = qkv + ((long)b * 3 * H + head * D) * Ls;
    const float* K = Q + (long)H * Ls;
    const float* V = K + (long)H * Ls;
    for (int d = lane; d < D; d += 64) qs[wave][d] = Q[(long)d * Ls + q];
    __builtin_amdgcn_wave_barrier();
    const float scale = sqrtf((float)D);
    const int nch = (L + 63) / 64;
    // ---- scores: keys along the lanes, the query broadcast from LDS; 8 independent loads in flight ----
    float mx = -INFINITY;
    for (int ch = 0; ch < nch; ++ch) {
        const int j = ch * 64 + lane;
        const int jc = min(j, L - 1);
        float acc = 0.f;
        for (int d0 = 0; d0 < D; d0 += 8) {
            float kv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) kv[i] = K[(long)(d0 + i) * Ls + jc];
#pragma unroll
            for (int i = 0; i < 8; ++i) acc += qs[wave][d0 + i] * kv[i];
        }
        const float s = (j < L && nonpad[(long)b * Ls + jc] != 0.f) ? acc / scale : -INFINITY;
        ps[wave][ch * 64 + lane] = s;
        mx = fmaxf(mx, s);
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) mx = fmaxf(mx, __shfl_xor(mx, m, 64));
    float sum = 0.f;
    for (int ch = 0; ch < nch; ++ch) {
        const float e = expf(ps[wave][ch * 64 + lane] - mx);     // exp(-inf) = 0 for masked / out-of-range keys
        ps[wave][ch * 64 + lane] = e;
        sum += e;
    }
#pragma unroll
    for (int m = 1; m < 64; m <<= 1) sum += __shfl_xor(sum, m, 64);
    for (int ch = 0; ch < nch; ++ch) ps[wave][ch * 64 + lane] /= sum;
    __builtin_amdgcn_wave_barrier();
    // ---- o[d] = sum_j p_j v[d][j]: output channels along the lanes, each lane streams its own row of V in 16-byte
    //      pieces (rows are contiguous in j), the probabilities are broadcast from LDS; no cross-lane reduction.
    //      Keys in [L, nch*64) carry p = 0 and read finite padding of the row. ----
    const int n4 = nch * 16;
    for (int dd = lane; dd < D; dd += 64) {
        const f32x4* vr = reinterpret_cast<const f32x4*>(V + (long)dd * Ls);
        const f32x4* pr = reinterpret_cast<const f32x4*>(&ps[wave][0]);
        float acc = 0.f;
        for (int j4 = 0; j4 < n4; j4 += 4) {
            f32x4 v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = vr[j4 + i];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const f32x4 pv = pr[j4 + i];
                acc += pv[0] * v[i][0];
                acc += pv[1] * v[i][1];
                acc += pv[2] * v[i][2];
                acc += pv[3] * v[i][3];
            }
        }
        out[((long)b * H + head * D + dd) * Ls + q] = acc;
    }
}

// ---------------------------------------------------------------------------------------------
// Length regulator + frame-level embeddings (acoustic_encoder.py:96-116), output [B, T, H] (h innermost):
//   cond = pad(enc)[mel2ph]  (+ spk)  + pitch_embed(log(1 + f0/700))  (+ variance embeds)  (+ key shift)  (+ speed)
// added in the reference's order.  `lin` holds (weight[H], bias[H]) pairs of the Linear(1, H) layers:
//   0 pitch, 1..4 energy / breathiness / voicing / tension, 5 key shift, 6 speed; `feat[k]` the matching [B,T] input
//   (nullptr = layer absent).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void enc_expand_kernel(const float* __restrict__ enc, const long long* __restrict__ mel2ph,
                                                         EncExpandArgs a, int H, int L, int Ls, int T,
                                                         float* __restrict__ cond) {
    const int b = blockIdx.y;
    const int t = blockIdx.x;
    const long long m = mel2ph[(long)b * T + t];
    const bool hit = m >= 1 && m <= L;
    const float f0 = a.feat[0][(long)b * T + t];
    const float f0_mel = logf(1.f + f0 / 700.f);
    for (int h = threadIdx.x; h < H; h += blockDim.x) {
        float v = hit ? enc[((long)b * H + h) * Ls + (m - 1)] : 0.f;
        if (a.spk_mix) {
            v += a.spk_mix[(long)b * a.spk_mix_bstride + (long)t * a.spk_mix_tstride + h];
        } else if (a.spk_table) {
            long long s = a.spk_id[b];
            s = s < 0 ? 0 : (s >= a.num_spk ? a.num_spk - 1 : s);
            v += a.spk_table[s * H + h];
        }
        v += f0_mel * a.lin_w[0][h] + a.lin_b[0][h];
        float ve = 0.f;
        bool any = false;
#pragma unroll
        for (int k = 1; k <= 4; ++k)
            if (a.lin_w[k]) {
                ve += a.feat[k][(long)b * T + t] * a.lin_w[k][h] + a.lin_b[k][h];
                any = true;
            }
        if (any) v += ve;
        if (a.lin_w[5]) v += a.feat[5][(long)b * T + t] * a.lin_w[5][h] + a.lin_b[5][h];
        if (a.lin_w[6]) v += a.feat[6][(long)b * T + t] * a.lin_w[6][h] + a.lin_b[6][h];
        cond[((long)b * T + t) * H + h] = v;
    }
}

// ---------------------------------------------------------------------------------------------
// Variance model glue (dsd_token_encode / dsd_predict_dur / dsd_cond_assemble).
// ---------------------------------------------------------------------------------------------
// nonpad[b][l] = !padding_mask[b][l]   (tts_modules.py:402: `1 - padding_mask.float()`)
__global__ void enc_nonpad_kernel(const unsigned char* __restrict__ pad, int L, int Ls, float* __restrict__ nonpad) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (l < L) nonpad[(long)b * Ls + l] = pad[(long)b * L + l] ? 0.f : 1.f;
}

// DurationPredictor head (tts_modules.py:128-134): Linear(C, 1) over the channels, mask, exp - offset, clamp at 0.
// Lanes run along the tokens; the channel loop reads x[b][c][l] coalesced.
__global__ void enc_dur_head_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                    const float* __restrict__ nonpad, int C, int L, int Ls, float offset,
                                    float* __restrict__ dur) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
    if (l >= L) return;
    const float* xb = x + (long)b * C * Ls + l;
    float acc = 0.f;
    for (int c = 0; c < C; ++c) acc += xb[(long)c * Ls] * w[c];
    const float v = (acc + bias[0]) * nonpad[(long)b * Ls + l];
    dur[(long)b * L + l] = fmaxf(expf(v) - offset, 0.f);
}

// dsd_cond_assemble: out[b][t][:] = sum_g scale_g * rowscale_g[b,t] * table_g[b][idx_g[b,t] + offset_g][:] + sum_k s_k[b,t] * v_k[:]
// One workgroup per (b, t) row, lanes along H: every table row and v_k is read contiguously.
__global__ __launch_bounds__(256) void assemble_kernel(const AssembleArgs a, float* __restrict__ out) {
    const int t = blockIdx.x, b = blockIdx.y;
    const long bt = (long)b * a.T + t;
    for (int h = threadIdx.x; h < a.H; h += blockDim.x) {
        float acc = 0.f;
        for (int g = 0; g < a.n_gather; ++g) {
            const long row = (long)a.g_idx[g][bt] + a.g_off[g];
            if (row < 0 || row >= a.g_rows[g]) continue;
            float v = a.g_table[g][(long)b * a.g_bstride[g] + row * a.H + h] * a.g_scale[g];
            if (a.g_rowscale[g]) v *= a.g_rowscale[g][bt];
            acc += v;
        }
        for (int k = 0; k < a.n_terms; ++k) acc += (a.t_s[k] ? a.t_s[k][bt] : 1.f) * a.t_v[k][h];
        out[bt * a.H + h] = acc;
    }
}

// RelPositionalEncoding.forward (espnet_positional_embedding.py:102-113) on x[b][c][l]:
//   x = x * sqrt(H) + pe[l][c],  pe[l][2i] = sin((max_len - 1 - l) * div[i]),  pe[l][2i + 1] = cos(same)
// (the table is built once for max_len = 5000 with reversed positions and sliced from the front).
__global__ void enc_relpos_kernel(float* __restrict__ x, const float* __restrict__ div, int C, int L, int Ls, float xscale,
                                  float last_pos) {
    const int l = blockIdx.x * blockDim.x + threadIdx.x, c = blockIdx.y, b = blockIdx.z;
    if (l >= L) return;
    const float ang = (last_pos - (float)l) * div[c >> 1];
    float* p = x + ((long)b * C + c) * Ls + l;
    *p = *p * xscale + ((c & 1) ? cosf(ang) : sinf(ang));
}

// SinusoidalPositionalEmbedding (common_layers.py:44-99) through make_positions (utils/__init__.py:118-128):
//   pos[b][l] = nonpad[b][l] * #{l' <= l : nonpad[b][l']};   x[b][c][l] += pos ? (c < H/2 ? sin(pos f_c) : cos(pos f_{c-H/2})) : 0
// One workgroup per batch row scans the (<= 2048) tokens, then every lane adds its channels.
__global__ __launch_bounds__(256) void enc_sinpos_kernel(float* __restrict__ x, const float* __restrict__ nonpad,
                                                         const float* __restrict__ freqs, int C, int L, int Ls) {
    __shared__ int pos[2048];
    const int b = blockIdx.x;
    if (threadIdx.x == 0) {
        int run = 0;
        for (int l = 0; l < L; ++l) {
            const bool keep = nonpad[(long)b * Ls + l] != 0.f;
            run += keep;
            pos[l] = keep ? run : 0;
        }
    }
    __syncthreads();
    const int half = C >> 1;
    for (long i = threadIdx.x; i < (long)C * L; i += blockDim.x) {
        const int c = (int)(i / L), l = (int)(i - (long)c * L);
        const int p = pos[l];
        if (p == 0 || c >= 2 * half) continue;
        const float ang = (float)p * freqs[c < half ? c : c - half];
        x[((long)b * C + c) * Ls + l] += c < half ? sinf(ang) : cosf(ang);
    }
}

// ------------------------------------------- launchers -------------------------------------------
hipError_t launch_enc_sinpos(float* x, const float* nonpad, const float* freqs, int C, int B, int L, int Ls, hipStream_t st) {
    if (L > 2048) return hipErrorInvalidValue;
    hipLaunchKernelGGL(enc_sinpos_kernel, dim3(B), dim3(256), 0, st, x, nonpad, freqs, C, L, Ls);
    return hipGetLastError();
}

hipError_t launch_enc_relpos(float* x, const float* div, int C, int B, int L, int Ls, hipStream_t st) {
    hipLaunchKernelGGL(enc_relpos_kernel, dim3((L + 63) / 64, C, B), dim3(64), 0, st, x, div, C, L, Ls, sqrtf((float)C), 4999.f);
    return hipGetLastError();
}

hipError_t launch_enc_nonpad(const unsigned char* pad, int B, int L, int Ls, float* nonpad, hipStream_t st) {
    hipLaunchKernelGGL(enc_nonpad_kernel, dim3((L + 63) / 64, B), dim3(64), 0, st, pad, L, Ls, nonpad);
    return hipGetLastError();
}

hipError_t launch_enc_dur_head(const float* x, const float* w, const float* bias, const float* nonpad, int C, int B, int L,
                               int Ls, float offset, float* dur, hipStream_t st) {
    hipLaunchKernelGGL(enc_dur_head_kernel, dim3((L + 63) / 64, B), dim3(64), 0, st, x, w, bias, nonpad, C, L, Ls, offset, dur);
    return hipGetLastError();
}

hipError_t launch_assemble(const AssembleArgs& a, float* out, hipStream_t st) {
    if (a.T > 2147483647 || a.B > 65535) return hipErrorInvalidValue;
    hipLaunchKernelGGL(assemble_kernel, dim3(a.T, a.B), dim3(a.H >= 256 ? 256 : 64), 0, st, a, out);
    return hipGetLastError();
}

hipError_t launch_enc_dur(const long long* mel2ph, int B, int T, int L, int* dur, hipStream_t st) {
    hipError_t e = hipMemsetAsync(dur, 0, sizeof(int) * (size_t)B * L, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(enc_dur_kernel, dim3((T + 255) / 256, B), dim3(256), 0, st, mel2ph, T, L, dur);
    return hipGetLastError();
}

hipError_t launch_enc_embed(const long long* tokens, const long long* langs, const int* dur, const float* txt_embed,
                            int vocab, const float* lang_embed, int n_lang_rows, const float* dur_w, const float* dur_b,
                            float embed_scale, int H, int B, int L, int Ls, float* x, float* nonpad, hipStream_t st) {
    hipLaunchKernelGGL(enc_embed_kernel, dim3((L + 63) / 64, H, B), dim3(64), 0, st, tokens, langs, dur, txt_embed, vocab,
                       lang_embed, n_lang_rows, dur_w, dur_b, embed_scale, H, L, Ls, x, nonpad);
    return hipGetLastError();
}

hipError_t launch_enc_layernorm(const float* x, float* y, const float* g, const float* beta, const float* mask, int C,
                                int B, int L, int Ls, float eps, hipStream_t st) {
    hipLaunchKernelGGL(enc_layernorm_kernel, dim3((L + 63) / 64, B), dim3(64 * ELN_WAVES), 0, st, x, y, g, beta, mask, C,
                       L, Ls, eps);
    return hipGetLastError();
}

hipError_t launch_enc_mask(float* x, const float* mask, int C, int B, int L, int Ls, hipStream_t st) {
    hipLaunchKernelGGL(enc_mask_kernel, dim3((L + 63) / 64, C, B), dim3(64), 0, st, x, mask, C, L, Ls);
    return hipGetLastError();
}

hipError_t launch_enc_swiglu(float* x, int half, long bstride, int B, int L, int Ls, hipStream_t st) {
    hipLaunchKernelGGL(enc_swiglu_kernel, dim3((L + 63) / 64, half, B), dim3(64), 0, st, x, half, bstride, L, Ls);
    return hipGetLastError();
}

hipError_t launch_enc_rope(float* qkv, const float* freqs, int H, int head_dim, int B, int L, int Ls, hipStream_t st) {
    hipLaunchKernelGGL(enc_rope_kernel, dim3((L + 63) / 64, H, B), dim3(64), 0, st, qkv, freqs, H, head_dim, L, Ls);
    return hipGetLastError();
}

hipError_t launch_enc_attention(const float* qkv, const float* nonpad, float* out, int H, int heads, int B, int L, int Ls,
                                hipStream_t st) {
    if (H / heads > ATT_MAXD || (L + 63) / 64 > ATT_MAXCH) return hipErrorInvalidValue;
    hipLaunchKernelGGL(enc_attention_kernel, dim3((L + 3) / 4, heads, B), dim3(256), 0, st, qkv, nonpad, out, H, heads, L, Ls);
    return hipGetLastError();
}

hipError_t launch_enc_expand(const float* enc, const long long* mel2ph, const EncExpandArgs& a, int H, int B, int L, int Ls,
                             int T, float* cond, hipStream_t st) {
    hipLaunchKernelGGL(enc_expand_kernel, dim3(T, B), dim3(256), 0, st, enc, mel2ph, a, H, L, Ls, T, cond);
    return hipGetLastError();
}

}  // namespace dsd
