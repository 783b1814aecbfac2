// LYNXNet's two pointwise GEMMs in SPLIT-bf16 arithmetic (opt-in precision mode, dsd_set_precision; see wn_layer_x3.hip for
// the arithmetic and its measured tolerance): every operand x = hi + lo (two bf16 values), a product = hi.hi + hi.lo + lo.hi on
// v_mfma_f32_16x16x32_bf16, fp32 accumulation.  LayerNorm statistics and normalisation, bias, SwiGLU, residual, the layer
// transition and its LayerNorm partials stay fp32 - they are lynx_layer.hip's, unchanged.
//
//   MODE 0 (pw1): u = out * silu(gate),  [out; gate] = W1 LayerNorm(xin) + b1           (lynxnet.py:53-56; LN affine folded into W1 / b1)
//   MODE 1 (pw2): v = W2 u' + b2 + x, then the NEXT layer's transition (gemm.hip EP_LYNX_NEXT; lynxnet.py:76-87)
//
// One workgroup = 512 output rows x 32 frames (4 waves x 8 row blocks), a work item per (frame tile, row tile) as in
// lx_pw1_kernel / lx_pw2_kernel.  Three bf16 MFMAs cost 3 / 16 of the fp32 walk, so the kernel is bound by its weight stream
// (hi + lo = 4 bytes per weight: 2 MiB per workgroup at K = 1024, at the ~70 GB/s a CU takes from L2) - hence the same build
// as wn_layer_x3.hip: weights pre-split and packed in the order a wave consumes them ([row tile][wave][k32 step][row
// block][hi | lo][lane][8 bf16]) through a register ring whose refills are pinned behind their MFMAs; the activation tile
// staged TRANSPOSED as two bf16 images [frame][channel] (a B fragment = 8 consecutive channels of one frame = one ds_read_b128
// per image), K in resident phases of up to 1024 channels (132 KiB of LDS for the two images).
#include <hip/hip_ext.h>

#include "dsd_internal.h"

namespace dsd {

#ifdef DSD_STAMPS
// [pw1 / pw2][workgroup][0..7]: s_memtime at the phase boundaries of wave 0 (tools/stamp_lynx_x3.py)
__device__ unsigned long long g_x3_stamps[2][4096][8];
#define X3_STAMP(i)                                                                     \
    do {                                                                                \
        if (threadIdx.x == 0 && blockIdx.x < 4096) {                                    \
            __builtin_amdgcn_sched_barrier(0);                                          \
            g_x3_stamps[MODE][blockIdx.x][i] = __builtin_amdgcn_s_memtime();            \
            __builtin_amdgcn_sched_barrier(0);                                          \
        }                                                                               \
    } while (0)
extern "C" int dsd_dbg_read_x3_stamps(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_x3_stamps), sizeof(g_x3_stamps));
}
#else
#define X3_STAMP(i)
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

__device__ __forceinline__ int fdiv_floor(int x, float inv) { return (int)(((float)x + 0.5f) * inv); }
constexpr unsigned kRange = 0x7FFFFFF0u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* ptr) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, kRange, 0x00020000);
}
__device__ __forceinline__ f32x4 ld4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ bf16x8 ldw(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ float ld1(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void st4(f32x4 v, dsd_i32x4 r, int voff, int soff) {
    dsd_store_b128<0>(__builtin_bit_cast(dsd_u32x4, v), r, voff, soff);
}
__device__ __forceinline__ float sigmoid_f(float v) { return __builtin_amdgcn_rcpf(1.f + expf(-v)); }

constexpr int BN = 32, MBW = 8, ES = BN + 4;

__device__ __forceinline__ int xcd_work() {
    const int nwg = gridDim.x;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int q8 = nwg >> 3, r8 = nwg & 7;
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
}

// one row block x all NCB column blocks of a k32 step: lo.hi, hi.lo, hi.hi, the accumulators alternating
template <int NCB>
__device__ __forceinline__ void x3_products(f32x4 (&a)[NCB], bf16x8 wh, bf16x8 wl, const bf16x8 (&bh)[NCB], const bf16x8 (&bl)[NCB]) {
#pragma unroll
    for (int n = 0; n < NCB; ++n) a[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, bh[n], a[n], 0, 0, 0);
#pragma unroll
    for (int n = 0; n < NCB; ++n) a[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, bl[n], a[n], 0, 0, 0);
#pragma unroll
    for (int n = 0; n < NCB; ++n) a[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, bh[n], a[n], 0, 0, 0);
}

}  // namespace

// MODE 0: pw1, KT = C (512 / 1024); MODE 1: pw2, KT = inner (1024 / 2048); NCB: 16-frame column blocks of the tile - 2 (32
// frames, K phases of <= 1024 channels) or 4 (64 frames in phases of 512: the SAME weight stream serves twice the frames, which
// is what a stream-bound kernel needs; for grids that still fill the chip with half the frame tiles)
template <int MODE, int KT, int RAG, int NCB>
__global__ __launch_bounds__(256, 1) void lx_x3_kernel(const LxLayerP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    constexpr int BNW = 16 * NCB;                    // frames of the tile
    constexpr int FQ = BNW / 4;                      // frame quads
    constexpr int KP = NCB == 4 ? 512 : (KT < 1024 ? KT : 1024);        // channels of a resident phase
    constexpr int NPH = KT / KP;
    constexpr int RS = KP + 8;                       // image row stride (bf16 elements): 16 lanes x 16 B on 64 distinct banks
    constexpr int NSP = KP / 32;                     // k32 steps per phase
    constexpr int NST = KT / 32;                     // ... of the whole weight stream
    constexpr int NB = NST * MBW;                    // row-block loads of a wave's stream
    constexpr int RB = 14;                           // weight ring: row-block slots (hi + lo = 8 VGPRs each)
    constexpr int NUT = (KP / 8) * FQ / 256;         // staging units (8 channels x 4 frames) per thread: 4 (2 at KP 512, 32 frames)
    __bf16* xhi = reinterpret_cast<__bf16*>(lds_raw);            // [BNW][RS]
    __bf16* xlo = xhi + BNW * RS;
    float* ep = reinterpret_cast<float*>(lds_raw);               // epilogue tiles over the dead images
    constexpr int IMG_B = 2 * BNW * RS * 2, EPT_B = 4 * 128 * ES * 4;
    float* tbl = reinterpret_cast<float*>(lds_raw + (IMG_B > EPT_B ? IMG_B : EPT_B));      // pw2: bias [512], step-projection scalar [512]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane >> 4, lcol = lane & 15, rq = lrow * 4;
    const int work = xcd_work();
    const int nft = RAG ? p.ncg : p.nft;
    const int mtile = fdiv_floor(work, p.inv_nft);
    const int ft = work - mtile * nft;
    const int rest = RAG ? p.cgmap[ft] : ft;
    const int b = fdiv_floor(rest, p.inv_tiles_per_b);
    const int t0 = (rest - b * p.tiles_per_b) * BNW;
    const int Ts = p.Ts;
    const int bu = __builtin_amdgcn_readfirstlane(b), t0u = __builtin_amdgcn_readfirstlane(t0);
    const int mu = __builtin_amdgcn_readfirstlane(mtile);
    X3_STAMP(0);

    // ---------------- weight stream of this wave; ring fill ----------------
    const __amdgpu_buffer_rsrc_t r_w = rsrc(reinterpret_cast<const unsigned char*>(MODE == 0 ? p.A1 : p.A2) + ((long)mu * 4 + wave) * NB * 2048);
    bf16x8 Whi[RB], Wlo[RB];
    auto w_issue = [&](int i) {
        if (i < NB) {
            Whi[i % RB] = ldw(r_w, lane * 16, i * 2048);
            Wlo[i % RB] = ldw(r_w, lane * 16, i * 2048 + 1024);
        }
    };
#pragma unroll
    for (int i = 0; i < RB; ++i) w_issue(i);
    __builtin_amdgcn_sched_barrier(0);

    // ---------------- staging of a phase: 8 channels x 4 frames per unit, transposed, split hi / lo ----------------
    const int fq = tid % FQ;                                     // the thread's frame quad, for every unit
    f32x4 mean = f32x4{0.f, 0.f, 0.f, 0.f}, rstd = f32x4{1.f, 1.f, 1.f, 1.f};
    if (MODE == 0) {
        // LayerNorm statistics of the tile's frames from the producer's per-64-row partials (ln_merge_kernel's arithmetic and order)
        constexpr int NT = KT / 64;
        const __amdgpu_buffer_rsrc_t r_p = rsrc(p.lnpart_in + (long)bu * NT * 2 * p.lnpart_ts + t0u);
        f32x4 pm[NT], pq[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            pm[i] = ld4(r_p, fq * 16, i * 2 * p.lnpart_ts * 4);
            pq[i] = ld4(r_p, fq * 16, (i * 2 + 1) * p.lnpart_ts * 4);
        }
        f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NT; ++i) s += 64.f * pm[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) mean[e] = s[e] / (float)KT;
        f32x4 m2 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const f32x4 d = pm[i] - mean;
            m2 += pq[i] + 64.f * d * d;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) rstd[e] = 1.f / sqrtf(m2[e] / (float)KT + 1e-5f);
    }
    const __amdgpu_buffer_rsrc_t r_in = MODE == 0 ? rsrc(p.xin + (long)bu * p.x_bstride + t0u) : rsrc(p.v + (long)bu * p.u_bstride + t0u);
    auto stage = [&](int ph) {
#pragma unroll
        for (int i0 = 0; i0 < NUT; i0 += 2) {
            f32x4 sv[2][8];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int co = tid / FQ + (256 / FQ) * (i0 + i);     // channel octet of the phase
#pragma unroll
                for (int c = 0; c < 8; ++c) sv[i][c] = ld4(r_in, ((8 * co + c) * Ts + 4 * fq) * 4, ph * KP * Ts * 4);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int co = tid / FQ + (256 / FQ) * (i0 + i);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    bf16x8 h8, l8;
#pragma unroll
                    for (int c = 0; c < 8; ++c) {
                        const float v = MODE == 0 ? (sv[i][c][e] - mean[e]) * rstd[e] : sv[i][c][e];
                        const __bf16 hv = (__bf16)v;
                        h8[c] = hv;
                        l8[c] = (__bf16)(v - (float)hv);
                    }
                    *reinterpret_cast<bf16x8*>(&xhi[(4 * fq + e) * RS + 8 * co]) = h8;
                    *reinterpret_cast<bf16x8*>(&xlo[(4 * fq + e) * RS + 8 * co]) = l8;
                }
            }
        }
    };
    if (MODE == 1) {                                             // bias and step-projection scalar of the workgroup's 512 rows
        const __amdgpu_buffer_rsrc_t r_b = rsrc(p.bias2 + 512 * mu);
        const float b0 = ld1(r_b, tid * 4, 0), b1 = ld1(r_b, tid * 4, 1024);
        float f0 = 0.f, f1 = 0.f;
        if (p.film) {
            const __amdgpu_buffer_rsrc_t r_f = rsrc(p.film + p.film_col0 + bu * p.film_colb + (long)512 * mu * p.film_cstride);
            f0 = ld1(r_f, tid * p.film_cstride * 4, 0);
            f1 = ld1(r_f, (tid + 256) * p.film_cstride * 4, 0);
        }
        tbl[tid] = b0;
        tbl[256 + tid] = b1;
        tbl[512 + tid] = f0;
        tbl[768 + tid] = f1;
    }
    X3_STAMP(1);
    stage(0);
    __syncthreads();
    X3_STAMP(2);

    // ---------------- K walk: NPH resident phases of NSP k32 steps ----------------
    f32x4 acc[MBW][NCB];
#pragma unroll
    for (int k = 0; k < MBW; ++k)
#pragma unroll
        for (int n = 0; n < NCB; ++n) acc[k][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int bbase = lcol * RS + 8 * lrow;
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph) {
        if (ph > 0) {
            X3_STAMP(5);                                         // (the last phase switch survives)
            __syncthreads();                                     // every wave is done with the previous phase's images
            stage(ph);
            __syncthreads();
            X3_STAMP(6);
        }
#pragma unroll
        for (int s = 0; s < NSP; ++s) {
            bf16x8 bh[NCB], bl[NCB];
#pragma unroll
            for (int n = 0; n < NCB; ++n) {
                const int off = bbase + 16 * n * RS + 32 * s;
                bh[n] = *reinterpret_cast<const bf16x8*>(&xhi[off]);
                bl[n] = *reinterpret_cast<const bf16x8*>(&xlo[off]);
            }
#pragma unroll
            for (int k = 0; k < MBW; ++k) {
                const int i = (ph * NSP + s) * MBW + k;
                x3_products<NCB>(acc[k], Whi[i % RB], Wlo[i % RB], bh, bl);
                w_issue(i + RB);
                __builtin_amdgcn_sched_barrier(0);               // (pinned: see wn_layer_x3.hip)
            }
        }
    }
    X3_STAMP(3);
    __syncthreads();                                             // the images are dead: the epilogue tiles go over them

    // ---------------- epilogues: one 32-frame half of the tile at a time (hf), through the wave's LDS tile ----------------
    const int ev0 = ((lane >> 3) * Ts + (lane & 7) * 4) * 4;
#pragma unroll
    for (int hf = 0; hf < NCB / 2; ++hf) {
        const int th = t0u + 32 * hf;                            // first frame of this half
        if (MODE == 0) {
            // bias + SwiGLU (common_layers.py:116-117: out * silu(gate)), transposed through LDS, float4 stores
            const int ch0 = 256 * mu + 64 * wave;                // first u channel of this wave
            const __amdgpu_buffer_rsrc_t r_b = rsrc(p.bias1);
            f32x4 bo[MBW];
#pragma unroll
            for (int k = 0; k < MBW; ++k) bo[k] = ld4(r_b, rq * 4, ((k & 1) * p.inner + ch0 + (k >> 1) * 16) * 4);
            float* ew = ep + wave * (64 * ES);
#pragma unroll
            for (int i = 0; i < MBW / 2; ++i)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float u0 = acc[2 * i][2 * hf + n][r] + bo[2 * i][r];
                        const float u1 = acc[2 * i + 1][2 * hf + n][r] + bo[2 * i + 1][r];
                        ew[(i * 16 + rq + r) * ES + n * 16 + lcol] = u0 * (u1 * sigmoid_f(u1));
                    }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const dsd_i32x4 w_o = dsd_rsrc_words(p.u + (long)bu * p.u_bstride + (long)ch0 * Ts + th);
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int idx = lane + 64 * m;
                st4(*reinterpret_cast<const f32x4*>(&ew[(idx >> 3) * ES + (idx & 7) * 4]), w_o, ev0, m * 8 * Ts * 4);
            }
        } else {
            // transition (gemm.hip EP_LYNX_NEXT; lynxnet.py:76-84 of the next layer), row-major
            const int row0 = 512 * mu + 128 * wave;
            const __amdgpu_buffer_rsrc_t r_a = rsrc(p.x + (long)bu * p.x_bstride + (long)row0 * Ts + th);
            const __amdgpu_buffer_rsrc_t r_c = rsrc((p.cpn ? p.cpn : p.x) + (long)bu * (p.cpn ? p.cpn_bstride : p.x_bstride) + (long)row0 * Ts + th);
            f32x4 aux[16], cpv[16];
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                aux[m] = ld4(r_a, ev0, m * 8 * Ts * 4);
                cpv[m] = ld4(r_c, ev0, m * 8 * Ts * 4);
            }
            const float* tb = tbl;
            const float* tf = tbl + 512;
            float* ew = ep + wave * (128 * ES);
#pragma unroll
            for (int k = 0; k < MBW; ++k)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int r = 0; r < 4; ++r) ew[(k * 16 + rq + r) * ES + n * 16 + lcol] = acc[k][2 * hf + n][r];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const dsd_i32x4 w_xo = dsd_rsrc_words(p.x + (long)bu * p.x_bstride + (long)row0 * Ts + th);
            const dsd_i32x4 w_xi = dsd_rsrc_words((p.xin_out ? p.xin_out : p.x) + (long)bu * p.x_bstride + (long)row0 * Ts + th);
            f32x4 xi[16];
#pragma unroll
            for (int m = 0; m < 16; ++m) {
                const int idx = lane + 64 * m;
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(&ew[(idx >> 3) * ES + (idx & 7) * 4]);
                const float brow = tb[128 * wave + (idx >> 3)], frow = tf[128 * wave + (idx >> 3)];
                f32x4 xo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float v = (a4[e] + brow) + aux[m][e];      // + bias, + residual (lynxnet.py:86)
                    float o = v, in = v;
                    if (p.cpn) {
                        in = v + cpv[m][e];
                        if (p.strong) o = in;
                    }
                    if (p.film) in = in + frow;
                    xo[e] = o;
                    xi[m][e] = in;
                }
                st4(xo, w_xo, ev0, m * 8 * Ts * 4);
                if (p.xin_out) st4(xi[m], w_xi, ev0, m * 8 * Ts * 4);
            }
            // LayerNorm partials of xin per 64-row tile (tiles 2w, 2w + 1 of this workgroup's 8): two passes over the registers
            if (p.lnpart) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    f32x4 sm = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int m = 0; m < 8; ++m) sm += xi[8 * h + m];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        sm[e] += __shfl_xor(sm[e], 8, 64);
                        sm[e] += __shfl_xor(sm[e], 16, 64);
                        sm[e] += __shfl_xor(sm[e], 32, 64);
                    }
                    const f32x4 mu4 = sm * (1.f / 64.f);
                    f32x4 q = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                    for (int m = 0; m < 8; ++m) {
                        const f32x4 d = xi[8 * h + m] - mu4;
                        q += d * d;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        q[e] += __shfl_xor(q[e], 8, 64);
                        q[e] += __shfl_xor(q[e], 16, 64);
                        q[e] += __shfl_xor(q[e], 32, 64);
                    }
                    if (lane < 8) {
                        const int tile = 8 * mu + 2 * wave + h;
                        float* lp = p.lnpart + ((long)bu * p.ln_tiles + tile) * 2 * p.lnpart_ts + th + lane * 4;
                        *reinterpret_cast<f32x4*>(lp) = mu4;
                        *reinterpret_cast<f32x4*>(lp + p.lnpart_ts) = q;
                    }
                }
            }
        }
        if (hf + 1 < NCB / 2) {                                  // the wave's tile is reused by the next half
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    X3_STAMP(4);
}

int lx_x3_lds_bytes(int kt, int ncb) {
    const int kp = ncb == 4 ? 512 : (kt < 1024 ? kt : 1024);
    const int img = 2 * 16 * ncb * (kp + 8) * 2, ept = 4 * 128 * 36 * 4;
    return (img > ept ? img : ept) + 1024 * 4;
}

bool lx_x3_supported(int C, int inner) { return (C == 1024 && inner == 2048) || (C == 512 && inner == 1024); }

template <int MODE, int KT, int RAG, int NCB>
static hipError_t lx_x3_launch(const LxLayerP& p, int nwg, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lx_x3_kernel<MODE, KT, RAG, NCB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr = true;
    }
    if (nwg == 0) return hipSuccess;
    return launch_timed(lx_x3_kernel<MODE, KT, RAG, NCB>, dim3(nwg), dim3(256), lx_x3_lds_bytes(KT, NCB), st, p, "lx_x3_kernel<%d, %d, %d, %d>", MODE, KT, RAG, NCB);
}

// which = 0: pw1 (p.A1 = the bf16x3 stream), 1: pw2 (p.A2); ncb = 2: 32-frame tiles, 4: 64-frame tiles (p.tiles_per_b / nft / cgmap
// count tiles of that width); p otherwise as for launch_lx_layer
hipError_t launch_lx_x3(const LxLayerP& p, int which, int C, int ncb, hipStream_t st) {
    if (!lx_x3_supported(C, p.inner) || (ncb != 2 && ncb != 4)) return hipErrorInvalidValue;
    const int nft = p.cgmap ? p.ncg : p.nft;
    const int nwg = nft * (which == 0 ? (2 * p.inner) / 512 : C / 512);
#define LX3_CASE(MODE_, KT_)                                                                                                   \
    {                                                                                                                          \
        if (ncb == 2) return p.cgmap ? lx_x3_launch<MODE_, KT_, 1, 2>(p, nwg, st) : lx_x3_launch<MODE_, KT_, 0, 2>(p, nwg, st);   \
        return p.cgmap ? lx_x3_launch<MODE_, KT_, 1, 4>(p, nwg, st) : lx_x3_launch<MODE_, KT_, 0, 4>(p, nwg, st);                 \
    }
    if (which == 0) {
        if (C == 1024) LX3_CASE(0, 1024)
        LX3_CASE(0, 512)
    }
    if (p.inner == 2048) LX3_CASE(1, 2048)
    LX3_CASE(1, 1024)
#undef LX3_CASE
}

hipError_t lx_x3_init_all() {
    LxLayerP p{};
    hipError_t e;
    for (int C : {512, 1024})
        for (int rag = 0; rag < 2; ++rag)
            for (int ncb : {2, 4}) {
                p.inner = 2 * C;
                p.cgmap = rag ? reinterpret_cast<const int*>(&p) : nullptr;
                p.ncg = 0;
                p.nft = 0;
                if ((e = launch_lx_x3(p, 0, C, ncb, nullptr)) != hipSuccess) return e;
                if ((e = launch_lx_x3(p, 1, C, ncb, nullptr)) != hipSuccess) return e;
            }
    return hipSuccess;
}

}  // namespace dsd
