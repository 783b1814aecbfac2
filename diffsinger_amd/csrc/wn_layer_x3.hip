// Fused WaveNet residual layer in SPLIT-bf16 arithmetic ("bf16x3"; opt-in: DSD_PRECISION=1 / dsd_set_precision).
//
// Every fp32 operand of the layer's two GEMMs is split x = hi + lo with hi = bf16(x), lo = bf16(x - hi) (round to nearest
// even) and a product is evaluated as hi.hi + hi.lo + lo.hi on v_mfma_f32_16x16x32_bf16 with fp32 accumulation; the lo.lo
// term (~2^-16 relative) is dropped.  Everything else is the fp32 kernel of wn_layer.hip: FiLM add, zero padding, the hoisted
// conditioner projection, sigmoid * tanh, bias, residual and skip arithmetic and their buffers stay fp32.  Measured on the
// numpy oracle (tools/bf16x3_tolerance.py): one evaluation of the 20 x 256 network differs from fp32 by 9.9e-6 (max / max),
// the 50-NFE DPM-Solver++ sample by 1.9e-6 - inside the fp32 path's own parity tolerances (2e-5 / 1.5e-5), which the -m gpu
// tests assert for this mode too (tests/test_gpu_bf16x3.py).
//
// Why: a bf16 MFMA delivers 16 x the FLOPs of the fp32 one per cycle, so three of them cost 3 / 16 of the fp32 walk and the
// layer stops being MFMA-bound.  What bounds it instead is the weight stream: hi + lo are 4 bytes per weight like fp32, 2 MB per
// 32-frame tile from L2 at the ~70 GB/s a CU takes from its XCD's L2 (MI355X_MICROARCH.md, gather table) = ~30 us against the
// fp32 kernel's 63 us of MFMA time.  The kernel is therefore built around that stream:
//   * weights pre-split and packed in the order a wave consumes them, [wave][k32 step][row block][hi, lo][lane][8 bf16] - one
//     linear stream of 1 KiB blocks per wave - through a ring of RB row-block slots in registers (compiler-counted waits);
//   * the x tile goes to LDS as two bf16 images [frame][channel] (channel-contiguous, 528-byte rows: the 16 lanes of a fragment
//     read hit 64 distinct banks), so a B fragment - 8 consecutive channels of one frame - is ONE ds_read_b128 per image at ANY
//     dilation (a tap shifts the row); the transposition happens in the staging pass, 8 channels x 4 frames per thread;
//   * the gate's z is written the same way (4 consecutive channels of a frame per lane: one 8-byte store per image).
// C = 256 only (acoustic / pitch networks), 32-frame tiles, halo 8 (dilation <= 8) or 16.
#include <hip/hip_ext.h>

#include "dsd_internal.h"

namespace dsd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ float sigmoid_fast(float v) { return __builtin_amdgcn_rcpf(1.f + __expf(-v)); }
__device__ __forceinline__ float tanh_fast(float v) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * v)); }
__device__ __forceinline__ int fdiv_floor(int x, float inv) { return (int)(((float)x + 0.5f) * inv); }

constexpr int C = 256, BN = 32, RS = C + 8;          // LDS image row stride in bf16 elements: 528 bytes
constexpr int MBW = 8;                               // 16-row blocks per wave (2C rows / 4 waves / 16), both GEMMs
constexpr int NS1 = 3 * C / 32, NS2 = C / 32;        // k32 steps: conv [tap][32-channel chunk], out-proj
constexpr int ES = BN + 4;                           // epilogue tile row stride (floats)

// one row block x both column blocks of a k32 step: lo.hi, hi.lo, hi.hi (smallest terms first), the two accumulators alternating
__device__ __forceinline__ void x3_products(f32x4 (&a)[2], bf16x8 wh, bf16x8 wl, const bf16x8 (&bh)[2], const bf16x8 (&bl)[2]) {
    a[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, bh[0], a[0], 0, 0, 0);
    a[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl, bh[1], a[1], 0, 0, 0);
    a[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, bl[0], a[0], 0, 0, 0);
    a[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, bl[1], a[1], 0, 0, 0);
    a[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, bh[0], a[0], 0, 0, 0);
    a[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh, bh[1], a[1], 0, 0, 0);
}

}  // namespace

// HL: staged halo frames on each side (8: dilation <= 8; 16: dilation 16); RAG: ragged batch
template <int HL, int RAG>
__global__ __launch_bounds__(256, 1) void wn_layer_x3_kernel(const WnLayerP p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    constexpr int NF = BN + 2 * HL;                  // staged frames
    constexpr int IMG = NF * RS;                     // one bf16 image, elements
    constexpr int NUNIT = (C / 8) * (NF / 4);        // staging units: 8 channels x 4 frames
    constexpr int NUT = (NUNIT + 255) / 256;         // ... per thread (2)
    constexpr int RB = 14;                           // weight ring: row-block slots in registers (hi + lo: 8 VGPRs each)
    __bf16* xhi = reinterpret_cast<__bf16*>(lds_raw);            // [NF][RS]; later z hi [BN][RS]
    __bf16* xlo = xhi + IMG;                                     // [NF][RS]; later z lo
    float* es = reinterpret_cast<float*>(lds_raw + 2 * IMG * 2); // [4 waves][16 * MBW][ES] epilogue / conditioner tiles
    float* fl = es + 4 * 16 * MBW * ES;                          // FiLM vector [C]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane >> 4, lcol = lane & 15, rq = lrow * 4;
    const int nwg = gridDim.x;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int q8 = nwg >> 3, r8 = nwg & 7;
    const int work = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    const int rest = RAG ? p.cgmap[work] : work + p.tile0;
    const int b = fdiv_floor(rest, p.inv_tiles_per_b);
    const int t0 = (rest - b * p.tiles_per_b) * BN;
    const int Tb = (RAG && p.lens) ? p.lens[b] : p.T;
    const int Ts = p.Ts;
    const int bu = __builtin_amdgcn_readfirstlane(b), t0u = __builtin_amdgcn_readfirstlane(t0);

    constexpr unsigned kRange = 0x7FFFFFF0u;
    auto rsrc = [](const void* ptr) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, kRange, 0x00020000); };
    auto ld4 = [](__amdgpu_buffer_rsrc_t r, int voff, int soff) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    };
    auto ldw = [](__amdgpu_buffer_rsrc_t r, int voff, int soff) {
        return __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    };
    auto ld1 = [](__amdgpu_buffer_rsrc_t r, int voff, int soff) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
    };

    // ---------------- prologue: FiLM vector, x tile (transposed to [frame][channel], split hi / lo), first weight blocks ----------------
    const __amdgpu_buffer_rsrc_t r_x = rsrc(p.xin + (long)bu * p.x_bstride + (t0u - HL));       // inside the arena's guard at t0 = 0
    const __amdgpu_buffer_rsrc_t r_f = rsrc(p.film + p.film_col0 + bu * p.film_colb);
    const float fmine = ld1(r_f, tid * p.film_cstride * 4, 0);
    f32x4 sv[NUT][8];
#pragma unroll
    for (int i = 0; i < NUT; ++i) {
        const int u = tid + 256 * i;
        const int co = u / (NF / 4), fq = u - co * (NF / 4);         // channel octet, frame quad
        if (NUNIT % 256 == 0 || u < NUNIT) {
#pragma unroll
            for (int c = 0; c < 8; ++c) sv[i][c] = ld4(r_x, ((8 * co + c) * Ts + 4 * fq) * 4, 0);
        }
    }
    // this wave's weight streams: one 1 KiB block per (k32 step, row block, hi | lo), linear in that order
    const __amdgpu_buffer_rsrc_t r_w1 = rsrc(reinterpret_cast<const unsigned char*>(p.Aconv) + (long)wave * NS1 * MBW * 2048);
    const __amdgpu_buffer_rsrc_t r_w2 = rsrc(reinterpret_cast<const unsigned char*>(p.Aout) + (long)wave * NS2 * MBW * 2048);
    bf16x8 Whi[RB], Wlo[RB];
    constexpr int NB1 = NS1 * MBW, NB2 = NS2 * MBW;      // row-block loads of the two walks: 192, 64
    auto w_issue = [&](int i) {                          // block i of the concatenated stream (conv, then out-proj) -> slot i % RB
        if (i < NB1) {
            Whi[i % RB] = ldw(r_w1, lane * 16, i * 2048);
            Wlo[i % RB] = ldw(r_w1, lane * 16, i * 2048 + 1024);
        } else if (i < NB1 + NB2) {
            Whi[i % RB] = ldw(r_w2, lane * 16, (i - NB1) * 2048);
            Wlo[i % RB] = ldw(r_w2, lane * 16, (i - NB1) * 2048 + 1024);
        }
    };
#pragma unroll
    for (int i = 0; i < RB; ++i) w_issue(i);
    __builtin_amdgcn_sched_barrier(0);
    fl[tid] = fmine;
    __syncthreads();
    // FiLM add, zero padding (wavenet.py:36-38: the pad applies to x + d), split, transpose
#pragma unroll
    for (int i = 0; i < NUT; ++i) {
        const int u = tid + 256 * i;
        const int co = u / (NF / 4), fq = u - co * (NF / 4);
        if (NUNIT % 256 == 0 || u < NUNIT) {
            const f32x4 fa0 = *reinterpret_cast<const f32x4*>(&fl[8 * co]), fa1 = *reinterpret_cast<const f32x4*>(&fl[8 * co + 4]);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int fr = 4 * fq + e;
                const int t = t0 - HL + fr;
                const bool ok = t >= 0 && t < Tb;
                bf16x8 h8, l8;
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float v = ok ? sv[i][c][e] + (c < 4 ? fa0[c] : fa1[c - 4]) : 0.f;
                    const __bf16 hv = (__bf16)v;
                    h8[c] = hv;
                    l8[c] = (__bf16)(v - (float)hv);
                }
                *reinterpret_cast<bf16x8*>(&xhi[fr * RS + 8 * co]) = h8;
                *reinterpret_cast<bf16x8*>(&xlo[fr * RS + 8 * co]) = l8;
            }
        }
    }
    __syncthreads();

    // ---------------- GEMM 1: dilated conv, 24 k32 steps = [tap][32-channel chunk] ----------------
    f32x4 acc[MBW][2];
#pragma unroll
    for (int k = 0; k < MBW; ++k) {
        acc[k][0] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc[k][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    // B fragment of a k32 step: lane (g = lrow, column lcol) holds channels 32 cs + 8 g .. + 7 of frame HL + 16 n + lcol + (tap - 1) dil
    const int bbase = (HL + lcol - p.dil) * RS + 8 * lrow;
    const int dstep = p.dil * RS;
    // operands fetched during GEMM 1 (behind the weight stream): the hoisted conditioner projection of this wave's rows as
    // row-major float4 (gate rows [0, 8 MBW), filter rows C below), and the out-proj bias in the accumulator layout
    constexpr int NE = 16 * MBW * (BN / 4) / 64;                 // 16
    const int orow0 = 16 * MBW * wave;
    const int ev0 = ((lane >> 3) * Ts + (lane & 7) * 4) * 4;
    const __amdgpu_buffer_rsrc_t r_c = rsrc(p.cp + (long)bu * p.cp_bstride + (long)(64 * wave) * Ts + t0u);
    const __amdgpu_buffer_rsrc_t r_b = rsrc(p.bias_out + orow0);
    f32x4 cpv[NE];
    f32x4 bo[MBW];
#pragma unroll
    for (int s = 0; s < NS1; ++s) {
        const int tap = s >> 3, cs = s & 7;
        bf16x8 bh[2], bl[2];
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int off = bbase + tap * dstep + 16 * n * RS + 32 * cs;
            bh[n] = *reinterpret_cast<const bf16x8*>(&xhi[off]);
            bl[n] = *reinterpret_cast<const bf16x8*>(&xlo[off]);
        }
        if (s < NE) cpv[s] = ld4(r_c, ev0, ((s % (NE / 2)) * 8 + (s >= NE / 2 ? C : 0)) * Ts * 4);
        else if (s < NE + MBW) bo[s - NE] = ld4(r_b, rq * 4, (s - NE) * 64);
#pragma unroll
        for (int k = 0; k < MBW; ++k) {
            const int i = s * MBW + k;
            const bf16x8 wh = Whi[i % RB], wl = Wlo[i % RB];
            x3_products(acc[k], wh, wl, bh, bl);
            w_issue(i + RB);                                     // the slot is free again: next block of the stream
            // (pinned: left to itself the scheduler sinks every weight load to just before its use - the ring collapses to two
            // registers and the walk runs at one L2 round trip per row block, 102 us per tile instead of ~35)
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    static_assert(NE + MBW <= NS1, "one extra operand load per step");

    // ---------------- gate (wavenet.py:41-42); z -> LDS over the dead x images, split hi / lo ----------------
    float* ew = es + wave * (16 * MBW * ES);                     // wave-private [16 * MBW][ES]
#pragma unroll
    for (int m = 0; m < NE; ++m) {
        const int idx = lane + 64 * m;
        *reinterpret_cast<f32x4*>(&ew[(idx >> 3) * ES + (idx & 7) * 4]) = cpv[m];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float zr[4][2][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float cg = ew[(i * 16 + rq + r) * ES + n * 16 + lcol];
                const float cf = ew[(8 * MBW + i * 16 + rq + r) * ES + n * 16 + lcol];
                zr[i][n][r] = sigmoid_fast(acc[2 * i][n][r] + cg) * tanh_fast(acc[2 * i + 1][n][r] + cf);
            }
    __syncthreads();                                             // every wave is done reading the x images
    __bf16* zhi = xhi;                                           // [BN][RS]
    __bf16* zlo = xhi + BN * RS;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            bf16x4 h4, l4;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const __bf16 hv = (__bf16)zr[i][n][r];
                h4[r] = hv;
                l4[r] = (__bf16)(zr[i][n][r] - (float)hv);
            }
            // lane (lrow, lcol): channels (4 wave + i) * 16 + rq .. + 3 of frame 16 n + lcol
            const int off = (16 * n + lcol) * RS + (4 * wave + i) * 16 + rq;
            *reinterpret_cast<bf16x4*>(&zhi[off]) = h4;
            *reinterpret_cast<bf16x4*>(&zlo[off]) = l4;
        }
#pragma unroll
    for (int k = 0; k < MBW; ++k)                                // GEMM 2 starts from its bias
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            acc[k][0][r] = bo[k][r];
            acc[k][1][r] = bo[k][r];
        }
    __syncthreads();

    // ---------------- GEMM 2: output projection, 8 k32 steps ----------------
    f32x4 pre[NE];
    const bool is_res = orow0 < C;                               // wave-uniform
    const long eoff = (long)bu * p.x_bstride + (long)(is_res ? orow0 : orow0 - C) * Ts + t0u;
    const unsigned long long xa = (unsigned long long)p.xin, sa = (unsigned long long)p.skip;
    const __amdgpu_buffer_rsrc_t r_e = rsrc((const float*)(is_res ? xa : sa) + eoff);
    const int zbase = lcol * RS + 8 * lrow;
#pragma unroll
    for (int s = 0; s < NS2; ++s) {
        bf16x8 bh[2], bl[2];
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const int off = zbase + 16 * n * RS + 32 * s;
            bh[n] = *reinterpret_cast<const bf16x8*>(&zhi[off]);
            bl[n] = *reinterpret_cast<const bf16x8*>(&zlo[off]);
        }
        pre[2 * s] = ld4(r_e, ev0, (2 * s) * 8 * Ts * 4);
        pre[2 * s + 1] = ld4(r_e, ev0, (2 * s + 1) * 8 * Ts * 4);
#pragma unroll
        for (int k = 0; k < MBW; ++k) {
            const int i = NB1 + s * MBW + k;
            const bf16x8 wh = Whi[i % RB], wl = Wlo[i % RB];
            x3_products(acc[k], wh, wl, bh, bl);
            w_issue(i + RB);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    static_assert(2 * NS2 == NE, "two epilogue operand loads per out-proj step");

    // ---------------- epilogue: residual / skip (wavenet.py:45-48), fp32, row-major through the wave's LDS tile ----------------
#pragma unroll
    for (int k = 0; k < MBW; ++k)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) ew[(k * 16 + rq + r) * ES + n * 16 + lcol] = acc[k][n][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {
        const unsigned long long xo = (unsigned long long)p.xout;
        const dsd_i32x4 w_o = dsd_rsrc_words((const float*)(is_res ? xo : sa) + eoff);
        const float scale = is_res ? 0.70710678118654752440f : 1.f;
        const bool add_pre = is_res || !p.first_layer;
#pragma unroll
        for (int m = 0; m < NE; ++m) {
            const int idx = lane + 64 * m;
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(&ew[(idx >> 3) * ES + (idx & 7) * 4]);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = ((add_pre ? pre[m][e] : 0.f) + a4[e]) * scale;
            dsd_store_b128<DSD_ST_AUX>(__builtin_bit_cast(dsd_u32x4, o), w_o, ev0, m * 8 * Ts * 4);
        }
    }
}

int wn_layer_x3_lds_bytes(int hl) { return 2 * (32 + 2 * hl) * (256 + 8) * 2 + 4 * 16 * 8 * 36 * 4 + 256 * 4; }

bool wn_layer_x3_supported(int C_, int dil) { return C_ == 256 && dil >= 1 && dil <= 16; }

template <int HL, int RAG>
static hipError_t x3_launch(const WnLayerP& p, int ntiles, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wn_layer_x3_kernel<HL, RAG>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    if (ntiles == 0) return hipSuccess;
    return launch_timed(wn_layer_x3_kernel<HL, RAG>, dim3(ntiles), dim3(256), wn_layer_x3_lds_bytes(HL), st, p,
                        "wn_layer_x3_kernel<%d, %d>", HL, RAG);
}

// p.Aconv / p.Aout point at the layer's bf16x3 weight streams (api.hip: pack_x3)
hipError_t launch_wn_layer_x3(const WnLayerP& p, int C_, int batch, hipStream_t st) {
    if (C_ != 256) return hipErrorInvalidValue;
    const int ntiles = p.cgmap ? p.ncg : (p.ntiles > 0 ? p.ntiles : batch * p.tiles_per_b);
    if (p.dil <= 8) return p.cgmap ? x3_launch<8, 1>(p, ntiles, st) : x3_launch<8, 0>(p, ntiles, st);
    return p.cgmap ? x3_launch<16, 1>(p, ntiles, st) : x3_launch<16, 0>(p, ntiles, st);
}

hipError_t wn_layer_x3_init_all() {
    WnLayerP p{};
    hipError_t e;
    for (int dil : {1, 16})
        for (int rag = 0; rag < 2; ++rag) {
            p.dil = dil;
            p.cgmap = rag ? reinterpret_cast<const int*>(&p) : nullptr;
            p.ncg = 0;
            p.tiles_per_b = 0;
            if ((e = launch_wn_layer_x3(p, 256, 0, nullptr)) != hipSuccess) return e;
        }
    return hipSuccess;
}

}  // namespace dsd
