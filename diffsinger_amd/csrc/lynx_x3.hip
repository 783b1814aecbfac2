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

// LDS-direct 16-byte load: lane l's 16 bytes land at LDS byte address lds_byte + 16 l (M0 carries the base, also above 64 KiB)
__device__ __forceinline__ void dma_b128(dsd_i32x4 rsrc_words, unsigned lds_byte, int voff, int soff) {
    asm volatile("s_mov_b32 m0, %0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds" ::"s"(lds_byte), "v"(voff), "s"(rsrc_words), "s"(soff)
                 : "memory");      // (m0 is not allocatable: nothing else in these kernels uses it)
}

}  // namespace

// MODE 0: pw1, KT = C (512 / 1024); MODE 1: pw2, KT = inner (1024 / 2048); NCB: 16-frame column blocks of the tile - 2 (32
// frames, K phases of <= 1024 channels) or 4 (64 frames: the SAME weight stream serves twice the frames, which is what a
// stream-bound kernel needs; for grids that still fill the chip with half the frame tiles).
// The 64-frame tile walks phases of 256 channels and stages them ASYNCHRONOUSLY: with the phase fetched, split and written
// between two barriers the MFMA pipe stood idle for 26 % of a pw1 workgroup's life (2 x 10-16 k cycles: 128 wave-loads of a CU
// that issues one per ~30-50 cycles, their latency, the split, the LDS writes - tools/stamp_lynx_x3.py).  Now phase ph + 1 travels
// as fp32 [channel][64 frames] into a 64 KiB LDS buffer by LDS-direct loads (buffer_load_dwordx4 ... lds: no registers, no
// VALU, one per row block behind the MFMAs of phase ph's first two steps; tools/harness/dma_harness.hip pins where the bytes
// land and that vmcnt covers them), and the phase switch is LDS -> LDS only: barrier, split + transpose raw -> images, barrier.
template <int MODE, int KT, int RAG, int NCB>
__global__ __launch_bounds__(256, 1) void lx_x3_kernel(const LxLayerP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    constexpr int BNW = 16 * NCB;                    // frames of the tile
    constexpr int FQ = BNW / 4;                      // frame quads
    constexpr bool ASYNC = NCB == 4;                 // 64-frame tiles: phases staged through the raw buffer
    constexpr int KP = ASYNC ? 256 : (KT < 1024 ? KT : 1024);           // channels of a resident phase
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
    constexpr int RAW_B = ASYNC ? KP * BNW * 4 : 0;              // [KP][64] fp32: the next phase as it is in memory
    float* rawb = reinterpret_cast<float*>(lds_raw + IMG_B);
    float* tbl = reinterpret_cast<float*>(lds_raw + (IMG_B + RAW_B > EPT_B ? IMG_B + RAW_B : EPT_B));      // pw2: bias [512], step-projection scalar [512]

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
    // LDS-direct loads of a phase: wave w moves channels [64 w, 64 w + 64), 4 channel rows (1 KiB of LDS) per load
    const dsd_i32x4 w_in = dsd_rsrc_words(MODE == 0 ? p.xin + (long)bu * p.x_bstride + t0u : p.v + (long)bu * p.u_bstride + t0u);
    const int dvoff = ((64 * wave + lrow) * Ts + lcol * 4) * 4;
    const unsigned lds_rawb = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)rawb;
    auto dma = [&](int ph, int j) {                              // load j (of 16) of phase ph
        dma_b128(w_in, lds_rawb + (unsigned)((64 * wave + 4 * j) * BNW * 4), dvoff, (ph * KP + 4 * j) * Ts * 4);
    };
    auto stage = [&](int ph) {
#pragma unroll
        for (int i0 = 0; i0 < NUT; i0 += 2) {
            f32x4 sv[2][8];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int co = tid / FQ + (256 / FQ) * (i0 + i);     // channel octet of the phase
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    if (ASYNC) sv[i][c] = *reinterpret_cast<const f32x4*>(&rawb[(8 * co + c) * BNW + 4 * fq]);
                    else sv[i][c] = ld4(r_in, ((8 * co + c) * Ts + 4 * fq) * 4, ph * KP * Ts * 4);
                }
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
    if (ASYNC) {
#pragma unroll
        for (int j = 0; j < 16; ++j) dma(0, j);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // phase 0 has landed (this wave's part) ...
        __syncthreads();                                         // ... and every wave's
    }
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
    struct BFrag { bf16x8 h[NCB], l[NCB]; } bq[2];
    auto read_b = [&](BFrag& f, int s, int n0, int n1) {
#pragma unroll
        for (int n = n0; n < n1; ++n) {
            const int off = bbase + 16 * n * RS + 32 * s;
            f.h[n] = *reinterpret_cast<const bf16x8*>(&xhi[off]);
            f.l[n] = *reinterpret_cast<const bf16x8*>(&xlo[off]);
        }
    };
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph) {
        if (ph > 0) {
            X3_STAMP(5);                                         // (the last phase switch survives)
            // every wave is done with the previous phase's images - and, 64-frame tiles, its part of this phase has landed in the
            // raw buffer: its 16 LDS-direct loads are older than the >= 2 RB ring loads the wave has issued since, and vector
            // loads return in order
            if (ASYNC) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * RB) : "memory");
            __syncthreads();
            stage(ph);
            __syncthreads();
            X3_STAMP(6);
        }
        // B fragments one step ahead (at the head of their own step the first MFMAs waited out the LDS latency of the reads)
        read_b(bq[0], 0, 0, NCB);
#pragma unroll
        for (int s = 0; s < NSP; ++s) {
#pragma unroll
            for (int k = 0; k < MBW; ++k) {
                const int i = (ph * NSP + s) * MBW + k;
                x3_products<NCB>(acc[k], Whi[i % RB], Wlo[i % RB], bq[s & 1].h, bq[s & 1].l);
                w_issue(i + RB);
                if (s + 1 < NSP && k < NCB) read_b(bq[(s + 1) & 1], s + 1, k, k + 1);
                if (ASYNC && ph + 1 < NPH && s < 2) dma(ph + 1, s * MBW + k);      // the next phase travels behind steps 0 and 1
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

// ---------------------------------------------------------------------------------------------------------------------------
// The TALL tile: 256 rows x 128 frames per workgroup (a wave: 4 row blocks x 8 column blocks - the same 128 accumulator
// registers and MFMAs per wave as the 512 x 64 tile).  Why: with the staging hidden, the 512 x 64 tile's K walk still took
// 62 k cycles for 49 k of MFMA (tools/stamp_lynx_x3.py): a CU moves ~36 bytes per cycle through its vector-memory path whatever
// the destination (registers or LDS-direct), and per tile that path carries the 2 MiB weight stream PLUS the 256 KiB of
// activations.  Bytes per MAC are (4 / frames + 4 / rows): 512 x 64 = 0.070, 256 x 128 = 0.047 - 1.5 x less, under the MFMA time.
// The weight stream is the one packed for the 512-row tile: workgroup (row tile rt, half h) takes row blocks 4 h .. 4 h + 3 of
// each wave's eight - wave w then owns rows 128 w + 64 h .. + 63 of the row tile: one LayerNorm tile (pw2), 32 u channels (pw1).
// K in phases of 128 channels, staged asynchronously as in lx_x3_kernel<.., 4>: the next phase by LDS-direct loads into a raw
// fp32 buffer [128 channels][128 frames] behind the first two steps' MFMAs, split + transposed LDS -> LDS at the phase switch.
// ---------------------------------------------------------------------------------------------------------------------------
template <int MODE, int KT, int RAG>
__global__ __launch_bounds__(256, 1) void lx_x3t_kernel(const LxLayerP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    constexpr int NCB = 8, BNW = 128, MB = 4;        // column blocks, frames, row blocks per wave
    constexpr int FQ = BNW / 4;                      // frame quads: 32
    constexpr int KP = 128, NPH = KT / KP, RS = KP + 8, NSP = KP / 32, NST = KT / 32;
    constexpr int NB = NST * MB;                     // row-block loads of a wave's stream
    constexpr int RB = 8;                            // weight ring: two steps ahead
    __bf16* xhi = reinterpret_cast<__bf16*>(lds_raw);            // [BNW][RS]
    __bf16* xlo = xhi + BNW * RS;
    float* ep = reinterpret_cast<float*>(lds_raw);               // epilogue tiles over the dead images: [4 waves][64][ES]
    constexpr int IMG_B = 2 * BNW * RS * 2, RAW_B = KP * BNW * 4;
    float* rawb = reinterpret_cast<float*>(lds_raw + IMG_B);     // [KP][128] fp32: the next phase as it is in memory
    float* tbl = reinterpret_cast<float*>(lds_raw + IMG_B + RAW_B);      // pw2: bias [512], step-projection scalar [512] of the row tile

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane >> 4, lcol = lane & 15, rq = lrow * 4;
    const int work = xcd_work();
    const int nft = RAG ? p.ncg : p.nft;
    const int mtile2 = fdiv_floor(work, p.inv_nft);              // 2 * row tile + half
    const int ft = work - mtile2 * nft;
    const int rest = RAG ? p.cgmap[ft] : ft;
    const int b = fdiv_floor(rest, p.inv_tiles_per_b);
    const int t0 = (rest - b * p.tiles_per_b) * BNW;
    const int Ts = p.Ts;
    const int bu = __builtin_amdgcn_readfirstlane(b), t0u = __builtin_amdgcn_readfirstlane(t0);
    const int mu = __builtin_amdgcn_readfirstlane(mtile2 >> 1), hh = __builtin_amdgcn_readfirstlane(mtile2 & 1);
    X3_STAMP(0);

    // ---------------- weight stream: blocks 4 hh .. 4 hh + 3 of every k32 step of the 512-row tile's wave stream ----------------
    const __amdgpu_buffer_rsrc_t r_w = rsrc(reinterpret_cast<const unsigned char*>(MODE == 0 ? p.A1 : p.A2) + ((long)mu * 4 + wave) * (NST * 8) * 2048 + hh * 4 * 2048);
    bf16x8 Whi[RB], Wlo[RB];
    auto w_issue = [&](int i) {                                  // i = 4 * step + block
        if (i < NB) {
            Whi[i % RB] = ldw(r_w, lane * 16, ((i >> 2) * 8 + (i & 3)) * 2048);
            Wlo[i % RB] = ldw(r_w, lane * 16, ((i >> 2) * 8 + (i & 3)) * 2048 + 1024);
        }
    };
#pragma unroll
    for (int i = 0; i < RB; ++i) w_issue(i);
    __builtin_amdgcn_sched_barrier(0);

    // ---------------- LDS-direct loads of a phase: wave w moves channels [32 w, 32 w + 32), 2 channel rows (1 KiB) per load ----------------
    const dsd_i32x4 w_in = dsd_rsrc_words(MODE == 0 ? p.xin + (long)bu * p.x_bstride + t0u : p.v + (long)bu * p.u_bstride + t0u);
    const int dvoff = ((32 * wave + (lane >> 5)) * Ts + (lane & 31) * 4) * 4;
    const unsigned lds_rawb = (unsigned)(unsigned long long)(__attribute__((address_space(3))) float*)rawb;
    auto dma = [&](int ph, int j) {                              // load j (of 16) of phase ph
        dma_b128(w_in, lds_rawb + (unsigned)((32 * wave + 2 * j) * BNW * 4), dvoff, (ph * KP + 2 * j) * Ts * 4);
    };
#pragma unroll
    for (int j = 0; j < 16; ++j) dma(0, j);

    // ---------------- LayerNorm statistics of the thread's staging frames (pw1) ----------------
    const int fq = tid % FQ;                                     // the thread's frame quad, for every staging unit
    f32x4 mean = f32x4{0.f, 0.f, 0.f, 0.f}, rstd = f32x4{1.f, 1.f, 1.f, 1.f};
    if (MODE == 0) {
        constexpr int NT = KT / 64;
        const __amdgpu_buffer_rsrc_t r_p = rsrc(p.lnpart_in + (long)bu * NT * 2 * p.lnpart_ts + t0u);
        f32x4 sm = f32x4{0.f, 0.f, 0.f, 0.f};
        f32x4 pm[NT];
#pragma unroll
        for (int i = 0; i < NT; ++i) pm[i] = ld4(r_p, fq * 16, i * 2 * p.lnpart_ts * 4);
#pragma unroll
        for (int i = 0; i < NT; ++i) sm += 64.f * pm[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) mean[e] = sm[e] / (float)KT;
        f32x4 m2 = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const f32x4 d = pm[i] - mean;
            m2 += ld4(r_p, fq * 16, (i * 2 + 1) * p.lnpart_ts * 4) + 64.f * d * d;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) rstd[e] = 1.f / sqrtf(m2[e] / (float)KT + 1e-5f);
    }
    // raw buffer -> the two transposed bf16 images: 8 channels x 4 frames per unit, two units per thread
    auto stage = [&]() {
        f32x4 sv[2][8];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int co = tid / FQ + (256 / FQ) * i;            // channel octet of the phase
#pragma unroll
            for (int c = 0; c < 8; ++c) sv[i][c] = *reinterpret_cast<const f32x4*>(&rawb[(8 * co + c) * BNW + 4 * fq]);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int co = tid / FQ + (256 / FQ) * i;
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
    };
    if (MODE == 1) {                                             // bias and step-projection scalar of the row tile's 512 rows
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
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // phase 0 has landed (this wave's part) ...
    __syncthreads();                                             // ... and every wave's
    stage();
    __syncthreads();
    X3_STAMP(2);

    // ---------------- K walk: NPH phases of 4 k32 steps x 4 row blocks x 8 column blocks ----------------
    f32x4 acc[MB][NCB];
#pragma unroll
    for (int k = 0; k < MB; ++k)
#pragma unroll
        for (int n = 0; n < NCB; ++n) acc[k][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int bbase = lcol * RS + 8 * lrow;
    struct BFrag { bf16x8 h[NCB], l[NCB]; } bq[2];
    auto read_b = [&](BFrag& f, int s, int n0, int n1) {
#pragma unroll
        for (int n = n0; n < n1; ++n) {
            const int off = bbase + 16 * n * RS + 32 * s;
            f.h[n] = *reinterpret_cast<const bf16x8*>(&xhi[off]);
            f.l[n] = *reinterpret_cast<const bf16x8*>(&xlo[off]);
        }
    };
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph) {
        if (ph > 0) {
            X3_STAMP(5);
            // this wave's part of the phase has landed (its LDS-direct loads are older than the 2 RB ring loads issued since; vector
            // loads return in order); every wave is done with the previous phase's images
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * RB) : "memory");
            __syncthreads();
            stage();
            __syncthreads();
            X3_STAMP(6);
        }
        // B fragments one step ahead: with the reads at the head of their own step the first MFMAs of every step waited out the
        // LDS latency of 16 ds_read_b128 (~400 of a step's 1536 cycles: the walk ran at 20 cycles per MFMA instead of 16)
        read_b(bq[0], 0, 0, NCB);
#pragma unroll
        for (int s = 0; s < NSP; ++s) {
#pragma unroll
            for (int k = 0; k < MB; ++k) {
                const int i = (ph * NSP + s) * MB + k;
#ifndef DSD_X3_DIAG
#define DSD_X3_DIAG 0
#endif
                x3_products<NCB>(acc[k], Whi[i % RB], Wlo[i % RB], bq[(DSD_X3_DIAG & 2) ? 0 : (s & 1)].h, bq[(DSD_X3_DIAG & 2) ? 0 : (s & 1)].l);
                if (!(DSD_X3_DIAG & 1)) w_issue(i + RB);
                if (!(DSD_X3_DIAG & 2) && s + 1 < NSP) read_b(bq[(s + 1) & 1], s + 1, k * (NCB / MB), (k + 1) * (NCB / MB));
                if (!(DSD_X3_DIAG & 4) && ph + 1 < NPH && s < 2) {                     // the next phase travels behind steps 0 and 1: two loads per row block
                    dma(ph + 1, 2 * (s * MB + k));
                    dma(ph + 1, 2 * (s * MB + k) + 1);
                }
                __builtin_amdgcn_sched_barrier(0);               // (pinned: see wn_layer_x3.hip)
            }
        }
    }
    X3_STAMP(3);
    __syncthreads();                                             // the images are dead: the epilogue tiles go over them

    // ---------------- epilogues: one 32-frame quarter of the tile at a time (hf), through the wave's LDS tile ----------------
    const int ev0 = ((lane >> 3) * Ts + (lane & 7) * 4) * 4;
    float* ew = ep + wave * (64 * ES);
#pragma unroll
    for (int hf = 0; hf < NCB / 2; ++hf) {
        const int th = t0u + 32 * hf;                            // first frame of this quarter
        if (MODE == 0) {
            // bias + SwiGLU (common_layers.py:116-117: out * silu(gate)), transposed through LDS, float4 stores
            const int ch0 = 256 * mu + 64 * wave + 32 * hh;      // first u channel of this wave
            const __amdgpu_buffer_rsrc_t r_b = rsrc(p.bias1);
            f32x4 bo[MB];
#pragma unroll
            for (int k = 0; k < MB; ++k) bo[k] = ld4(r_b, rq * 4, ((k & 1) * p.inner + ch0 + (k >> 1) * 16) * 4);
#pragma unroll
            for (int i = 0; i < MB / 2; ++i)
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
            for (int m = 0; m < 4; ++m) {
                const int idx = lane + 64 * m;
                st4(*reinterpret_cast<const f32x4*>(&ew[(idx >> 3) * ES + (idx & 7) * 4]), w_o, ev0, m * 8 * Ts * 4);
            }
        } else {
            // transition (gemm.hip EP_LYNX_NEXT; lynxnet.py:76-84 of the next layer), row-major: the wave's 64 rows = one LayerNorm tile
            const int wr0 = 128 * wave + 64 * hh;                // first row of the wave within the 512-row tile
            const int row0 = 512 * mu + wr0;
            const __amdgpu_buffer_rsrc_t r_a = rsrc(p.x + (long)bu * p.x_bstride + (long)row0 * Ts + th);
            const __amdgpu_buffer_rsrc_t r_c = rsrc((p.cpn ? p.cpn : p.x) + (long)bu * (p.cpn ? p.cpn_bstride : p.x_bstride) + (long)row0 * Ts + th);
            f32x4 aux[8], cpv[8];
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                aux[m] = ld4(r_a, ev0, m * 8 * Ts * 4);
                cpv[m] = ld4(r_c, ev0, m * 8 * Ts * 4);
            }
            const float* tb = tbl;
            const float* tf = tbl + 512;
#pragma unroll
            for (int k = 0; k < MB; ++k)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int r = 0; r < 4; ++r) ew[(k * 16 + rq + r) * ES + n * 16 + lcol] = acc[k][2 * hf + n][r];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const dsd_i32x4 w_xo = dsd_rsrc_words(p.x + (long)bu * p.x_bstride + (long)row0 * Ts + th);
            const dsd_i32x4 w_xi = dsd_rsrc_words((p.xin_out ? p.xin_out : p.x) + (long)bu * p.x_bstride + (long)row0 * Ts + th);
            f32x4 xi[8];
#pragma unroll
            for (int m = 0; m < 8; ++m) {
                const int idx = lane + 64 * m;
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(&ew[(idx >> 3) * ES + (idx & 7) * 4]);
                const float brow = tb[wr0 + (idx >> 3)], frow = tf[wr0 + (idx >> 3)];
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
            // LayerNorm partials of xin over the wave's 64-row tile
            if (p.lnpart) {
                f32x4 sm = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int m = 0; m < 8; ++m) sm += xi[m];
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
                    const f32x4 d = xi[m] - mu4;
                    q += d * d;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    q[e] += __shfl_xor(q[e], 8, 64);
                    q[e] += __shfl_xor(q[e], 16, 64);
                    q[e] += __shfl_xor(q[e], 32, 64);
                }
                if (lane < 8) {
                    const int tile = 8 * mu + 2 * wave + hh;
                    float* lp = p.lnpart + ((long)bu * p.ln_tiles + tile) * 2 * p.lnpart_ts + th + lane * 4;
                    *reinterpret_cast<f32x4*>(lp) = mu4;
                    *reinterpret_cast<f32x4*>(lp + p.lnpart_ts) = q;
                }
            }
        }
        if (hf + 1 < NCB / 2) {                                  // the wave's tile is reused by the next quarter
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
    }
    X3_STAMP(4);
}

constexpr int kX3tLds = 2 * 128 * 136 * 2 + 128 * 128 * 4 + 1024 * 4;

int lx_x3_lds_bytes(int kt, int ncb) {
    const int kp = ncb == 4 ? 256 : (kt < 1024 ? kt : 1024);
    const int img = 2 * 16 * ncb * (kp + 8) * 2, rawb = ncb == 4 ? kp * 64 * 4 : 0, ept = 4 * 128 * 36 * 4;
    return (img + rawb > ept ? img + rawb : ept) + 1024 * 4;
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

template <int MODE, int KT, int RAG>
static hipError_t lx_x3t_launch(const LxLayerP& p, int nwg, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lx_x3t_kernel<MODE, KT, RAG>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr = true;
    }
    if (nwg == 0) return hipSuccess;
    return launch_timed(lx_x3t_kernel<MODE, KT, RAG>, dim3(nwg), dim3(256), kX3tLds, st, p, "lx_x3t_kernel<%d, %d, %d>", MODE, KT, RAG);
}

// which = 0: pw1 (p.A1 = the bf16x3 stream), 1: pw2 (p.A2); ncb = 2: 32-frame tiles, 4: 64-frame tiles, 8: 128-frame tiles of 256
// rows (lx_x3t_kernel) - p.tiles_per_b / nft / cgmap count tiles of that width; p otherwise as for launch_lx_layer
hipError_t launch_lx_x3(const LxLayerP& p, int which, int C, int ncb, hipStream_t st) {
    if (!lx_x3_supported(C, p.inner) || (ncb != 2 && ncb != 4 && ncb != 8)) return hipErrorInvalidValue;
    const int nft = p.cgmap ? p.ncg : p.nft;
    const int nwg = nft * (which == 0 ? (2 * p.inner) / 512 : C / 512) * (ncb == 8 ? 2 : 1);
#define LX3_CASE(MODE_, KT_)                                                                                                   \
    {                                                                                                                          \
        if (ncb == 2) return p.cgmap ? lx_x3_launch<MODE_, KT_, 1, 2>(p, nwg, st) : lx_x3_launch<MODE_, KT_, 0, 2>(p, nwg, st);   \
        if (ncb == 8) return p.cgmap ? lx_x3t_launch<MODE_, KT_, 1>(p, nwg, st) : lx_x3t_launch<MODE_, KT_, 0>(p, nwg, st);       \
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
            for (int ncb : {2, 4, 8}) {
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
