// fp32 MFMA "conv-GEMM" family for gfx950 (MI355X).
//
// Every dense contraction on the denoiser path has the same shape:
//     out[m, t] = epilogue( sum_{tap, c} W[m, c, tap] * stage(x)[c, t + (tap - 1) * dil] )
// with m = output channel, c = input channel, t = frame (time innermost in HBM).  It covers
//   * the k=3 dilated conv of ResidualBlock (wavenet.py:22-28,36-38) with the FiLM add folded into
//     staging and the sigmoid*tanh gate (wavenet.py:41-42) folded into the epilogue,
//   * every 1x1 conv / Linear on the path (wavenet.py:29-31,56-62,71-72; lynxnet.py:55,59,71-72,104-124),
//   * the residual/skip update (wavenet.py:44-48), SwiGLU (common_layers.py:116-117) and the
//     solver's linear combination as epilogues.
//
// Mapping to CDNA4:
//   * v_mfma_f32_16x16x4_f32 (exact fp32, 64 FLOP/clk/SIMD).  A = weights, B = activations.
//   * A is pre-packed on the host in fragment order AND in the order the K walk consumes it
//     ([64-channel chunk][tap][k16]), so a wave streams ONE linear sequence of coalesced 1 KiB
//     global_load_dwordx4 blocks straight into VGPRs (weights are L2/MALL resident; no LDS round trip)
//     through an 8-deep register ring: inline-asm loads (SGPR base + lane offset + immediate) waited
//     for with hand-counted s_waitcnt vmcnt(N).
//   * B (fast path): 64-channel chunks [64][BN + 2*halo] double-buffered in LDS; chunk c+1 travels
//     global -> VGPR (asm loads) under the MFMAs of chunk c, gets the FiLM / LayerNorm / [0,T) mask
//     transform and is written to the other buffer; one barrier per chunk.  Row stride S = 16 (mod 32)
//     floats: the 4 k-rows x 16 columns of a fragment read hit 64 distinct banks; S is a template
//     constant, so every fragment read is "base VGPR + immediate".
//   * workgroup = 4 waves (one per SIMD) as 2 (rows) x 2 (frames); tile = 64 rows x 32*NB frames.
//     Gate / SwiGLU pairs (row r and row r + C) are packed into the same wave.
//   * epilogues: accumulators -> LDS tile -> row-major float4 arithmetic and 16-B coalesced stores
//     (gate, residual/skip), or straight from registers (bias+activation, SwiGLU, solver update).
#include <hip/hip_ext.h>

#include <type_traits>

#include "dsd_internal.h"

#ifndef DSD_GEMM_PIN_ARGS
#define DSD_GEMM_PIN_ARGS 1
#endif

namespace dsd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float act_apply(float v, int act) {
    switch (act) {
        case ACT_RELU: return fmaxf(v, 0.f);
        case ACT_MISH: return v * tanhf(log1pf(expf(v)));           // nn.Mish (wavenet.py:60)
        case ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));  // exact GELU (lynxnet.py:106)
        case ACT_LRELU: return v >= 0.f ? v : v * 0.1f;                 // LRELU_SLOPE (nsf_hifigan/models.py:15)
        case ACT_TANH: return tanhf(v);
        case ACT_SILU: return v / (1.f + expf(-v));                     // nn.SiLU ('swish', common_layers.py:130)
        default: return v;
    }
}
__device__ __forceinline__ float sigmoid_f(float v) { return 1.f / (1.f + expf(-v)); }
// Gate nonlinearities on the hardware exp / rcp units (v_exp_f32, v_rcp_f32: ~1 ulp each).  Absolute error of
// sigmoid(g) * tanh(f) stays below 3e-7 - far inside the 2e-5 per-evaluation parity tolerance - at a fifth of
// the instruction count of libm's expf / tanhf (the gate is ~1 k cycles of a 23 k-cycle workgroup at B = 1).
__device__ __forceinline__ float sigmoid_fast(float v) { return __builtin_amdgcn_rcpf(1.f + __expf(-v)); }
__device__ __forceinline__ float tanh_fast(float v) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * v)); }

#ifdef DSD_STAMPS
// Diagnostic build only (tools/stamp_profile.py): wave 0 of every workgroup records s_memtime at phase
// boundaries into a buffer of its own; no output value depends on a stamp.
__device__ unsigned long long g_stamps[8][4096][8];
#define DSD_STAMP(i)                                                                              \
    do {                                                                                          \
        if (threadIdx.x == 0 && blockIdx.x < 4096) {                                              \
            __builtin_amdgcn_sched_barrier(0);                                                    \
            g_stamps[EPI][blockIdx.x][i] = __builtin_amdgcn_s_memtime();                          \
            __builtin_amdgcn_sched_barrier(0);                                                    \
        }                                                                                         \
    } while (0)
// finer stamps inside the first chunk pair of the pipelined K loop
__device__ unsigned long long g_stamps2[8][4096][16];
#define DSD_STAMP2(i)                                                                             \
    do {                                                                                          \
        if (threadIdx.x == 0 && blockIdx.x < 4096 && c == 0) {                                    \
            __builtin_amdgcn_sched_barrier(0);                                                    \
            g_stamps2[EPI][blockIdx.x][i] = __builtin_amdgcn_s_memtime();                         \
            __builtin_amdgcn_sched_barrier(0);                                                    \
        }                                                                                         \
    } while (0)
extern "C" int dsd_dbg_read_stamps(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps), sizeof(g_stamps));
}
extern "C" int dsd_dbg_read_stamps2(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps2), sizeof(g_stamps2));
}
#else
#define DSD_STAMP(i)
#define DSD_STAMP2(i)
#endif

// floor(x / d) for 0 <= x < 2^22 with inv = 1.0f / d: one cvt + mul + cvt instead of the ~40-instruction
// integer-division expansion (the kernel prologue is on the latency-critical path at B = 1)
__device__ __forceinline__ int fdiv_floor(int x, float inv) { return (int)(((float)x + 0.5f) * inv); }

constexpr int PF = 8;   // A-fragment prefetch distance in k16 iterations (2 x 1 KiB loads each)

// The weight-fragment ring is loaded with inline asm and waited for with hand-counted s_waitcnt vmcnt(N):
// hipcc (ROCm 7.2) parks a conservative vmcnt(1) on the loop header for a loop-carried register ring, which
// exposes the whole L2 latency once per group.  Form (ii) of cdna_hip_programming.md 5.7: "=v" loads, then a
// wait statement that names the destinations "+v" so no consumer can be scheduled above it.  Inside the K loop
// the ring loads are the only VMEM operations, so "all but the 2*(PF-1) youngest" is exactly "slot u landed".
__device__ __forceinline__ void ring_load(f32x4& dst, const float* ptr) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory");
}
template <int N>
__device__ __forceinline__ void ring_wait(f32x4& a, f32x4& b) {
    asm volatile("s_waitcnt vmcnt(%2)" : "+v"(a), "+v"(b) : "n"(N) : "memory");
}

// Fast-path form: SGPR base + 32-bit lane offset + immediate, so a refill costs one instruction and no
// per-iteration address arithmetic.  `base` is advanced with SALU adds only (an SGPR written by a VALU
// readfirstlane needs 5 wait states before a VMEM reads it; it is produced once, long before its first use).
template <int IMM>
__device__ __forceinline__ void ring_load_s(f32x4& dst, unsigned voff, unsigned long long base) {
    asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=v"(dst) : "v"(voff), "s"(base), "n"(IMM) : "memory");
}

// SW > 0: "fast" instantiation - LDS row stride S == SW is a compile-time constant (every B-fragment read is
// base + immediate), the K walk is linear in groups of 8 steps (K % 128 == 0 and no tap/chunk mixing), so a
// k16 step is 8*NB MFMAs + 4*NB ds_read + 2 loads + 1 counted wait and nothing else.  SW == 0: generic
// instantiation (runtime S, any K multiple of 16, whole chunk staged then walked) for every other shape.
// WN = waves along time: 2 (waves as 2 rows x 2 frames, each 2 row blocks x NB frame blocks, tile 64 x 32*NB) or
// 1 (waves as 4 rows x 1, each ONE 16-row block x NB frame blocks, tile 64 rows x 16*NB frames).  WN = 1, NB = 1 is
// the "narrow" 16-frame tile for grids that would otherwise leave most CUs without a workgroup (T <= ~750 at B = 1:
// 17.4 -> 12.8 ms per 50-NFE utterance at T = 256).  WN = 1, NB = 2 - the same 64 x 32 tile with no weight block
// streamed by two waves - was measured too: halving the weight traffic does not pay for doubling each wave's LDS
// fragment reads (17.5 -> 18.8 ms at B = 1), so 32- and 64-frame tiles keep the 2 x 2 layout.
// RAG = 1: ragged batch.  The grid covers only the column groups (item, frame tile) that hold valid frames - p.cgmap
// lists them, p.ncg of them - and the zero padding of a k-tap convolution's input starts at the item's own length
// p.lens[b].  Separate instantiations so that the dense kernels carry no trace of it (an always-present scalar load of
// the length cost 1.3 % of the B = 1 loop, 4 % when placed in the staging loads' shadow).
template <int STAGE, int TAPS, int EPI, int NB, int SW, int RES = 0, int WN = 2, int RAG = 0>
__global__ __launch_bounds__(256, (SW > 0 && NB * WN == 4) ? 3 : 1) void gemm_kernel(const GemmP p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
#if DSD_GEMM_PIN_ARGS
    // the header of the ~1 KB argument block in SGPRs behind ONE batch of scalar loads (as wn_rowsplit.hip's rs_pin_args: left
    // alone the fields arrive in several dependent, cold round trips before the first vector load - on 5 us kernels)
    asm volatile("" ::"s"(p.A), "s"(p.bias), "s"(p.M), "s"(p.C), "s"(p.B), "s"(p.b_bstride), "s"(p.b_rstride), "s"(p.K), "s"(p.T),
                 "s"(p.tiles_per_b), "s"(p.mtiles), "s"(p.inv_mtiles), "s"(p.inv_tiles_per_b), "s"(p.inv_w4), "s"(p.gm_shift),
                 "s"(p.lpr_shift), "s"(p.dil), "s"(p.HL), "s"(p.in_scale), "s"(p.film), "s"(p.film_cstride), "s"(p.film_col0),
                 "s"(p.film_colb), "s"(p.act), "s"(p.out), "s"(p.o_bstride), "s"(p.o_rstride), "s"((int)gridDim.x));
    if (EPI == EP_GATE || EPI == EP_BIAS_RES || EPI == EP_LYNX_NEXT) asm volatile("" ::"s"(p.aux), "s"(p.aux_bstride), "s"(p.aux_rstride));
    if (EPI == EP_RESSKIP) asm volatile("" ::"s"(p.x), "s"(p.skip), "s"(p.first_layer));
    if (EPI == EP_LINCOMB) asm volatile("" ::"s"(p.nout));
#endif
    static_assert(WN == 2 || (WN == 1 && SW > 0 && EPI != EP_SWIGLU && STAGE != ST_LN), "4x1 wave layout: fast path only");
    static_assert(SW == 0 || (STAGE != ST_LRELU && EPI != EP_SCATTER), "leaky-ReLU staging / scatter epilogue: generic path");
    static_assert(EPI != EP_LYNX_NEXT || WN == 2, "LYNXNet transition epilogue: 2 x 2 wave layout");
    constexpr int WM = 4 / WN;                  // waves along rows
    constexpr int MB = 4 / WM;                  // 16-row blocks per wave
    constexpr int BN = 16 * NB * WN;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably wave-uniform (SGPR)
    const int wm = wave / WN, wn = wave % WN;
    // 1-D grid with an XCD-aware remap (speed only, bijective for any grid size): the dispatcher deals
    // workgroups round-robin over the 8 XCDs, so blocks b and b+8 share an L2.  Work items are numbered with the
    // row tile fastest, and XCD k takes a CONTIGUOUS range of them: the workgroups that stage the SAME activation
    // tile (same frames, all row tiles) share an XCD, so each L2 pulls 1/8 of the activations through the fabric
    // instead of all of them (measured: the staging phase was bound by 8 XCDs each re-reading the whole x).
    // Weights are then read by every XCD, but they stream during the K loop, off the latency-critical path.
    const int nwg = gridDim.x;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int q8 = nwg >> 3, r8 = nwg & 7;
    const int work = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    int rest0, mtile;
    if (p.gm_shift > 0) {           // grouped order: work = (row group * frame tiles + frame tile) << gm_shift | row tile in group
        const int mg = fdiv_floor(work, p.inv_per_group);
        const int r = work - mg * p.per_group;
        rest0 = r >> p.gm_shift;
        mtile = (mg << p.gm_shift) + (r & ((1 << p.gm_shift) - 1));
    } else {
        rest0 = fdiv_floor(work, p.inv_mtiles);
        mtile = work - rest0 * p.mtiles;
    }
    const int rest = RAG ? p.cgmap[__builtin_amdgcn_readfirstlane(rest0)] : rest0;     // scalar loads: off the vmcnt ledger
    const int b = fdiv_floor(rest, p.inv_tiles_per_b);
    const int t0 = (rest - b * p.tiles_per_b) * BN;
    const int Tb = (RAG && p.lens) ? p.lens[__builtin_amdgcn_readfirstlane(b)] : p.T;
    const int K16 = p.K >> 4;
    const int S = SW > 0 ? SW : p.S;
    const int HL = p.HL;
    DSD_STAMP(0);

    f32x4 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // TAPS == 0: kernel size taken from p.taps at run time (dense k-tap convs of the aux decoder; generic path only)
    const int taps = TAPS > 0 ? TAPS : p.taps;
    // packed A: [mblk][taps*K16][64 lanes][4]; this wave owns packed m-blocks 4*mtile + 2*wm + {0,1}
    const long a_blk = (long)(taps * K16) * 256;

    const float* bsrc = p.B + (long)b * p.b_bstride;
    const int W4 = (BN + 2 * HL) >> 2;      // float4 per staged row
    const int lrow = lane >> 4, lcol = lane & 15;

    // EP_GATE: the hoisted conditioner projection of this wave's outputs is fetched now and consumed in the
    // epilogue, so its latency hides under the whole K loop.
    // Epilogue operands (EP_GATE: hoisted conditioner projection; EP_RESSKIP: residual stream / running skip sum)
    // are fetched at kernel start and consumed after the K loop, so their latency hides under it.  They are
    // fetched in the ROW-MAJOR thread mapping of the LDS-staged epilogue below: lane -> 4 consecutive frames,
    // i.e. full 256-B (BN = 64) / 128-B row segments per 16 / 8 lanes instead of 64-B fragment-shaped pieces.
    constexpr int EQ = BN / 4;                      // float4 per output row
    constexpr int EROWS = 256 / EQ;                 // rows covered per pass of the 256 threads
    const int e_c4 = tid % EQ, e_r0 = tid / EQ;
    constexpr int NGATE = (32 + EROWS - 1) / EROWS;         // gate epilogue: channel passes per thread (32 pairs / tile)
    constexpr int NRES = (64 + EROWS - 1) / EROWS;          // res/skip epilogue: row passes per thread
    constexpr int NPRE = (EPI == EP_GATE) ? 2 * NGATE : (EPI == EP_RESSKIP ? NRES : 1);
    f32x4 pre[NPRE];
    auto epi_prefetch = [&]() {
        const long colo = (long)b * (EPI == EP_GATE ? p.aux_bstride : p.o_bstride) + t0 + e_c4 * 4;
        if (EPI == EP_RESSKIP) {
            // rows of this workgroup: [mtile*64, +64) of the 2C outputs; residual half -> x, skip half -> skip sum
            // (pointer chosen with integer arithmetic: a select between the struct FIELDS makes hipcc load the
            // pointer itself through a dependent vector load)
            const unsigned long long xa = (unsigned long long)p.x, sa = (unsigned long long)p.skip;
#pragma unroll
            for (int k = 0; k < NRES; ++k) {
                const int row = min(mtile * 64 + e_r0 + k * EROWS, p.M - 1);
                const bool is_res = row < p.C;
                const float* base = (const float*)(is_res ? xa : sa);
                ring_load(pre[k], base + colo + (long)(is_res ? row : row - p.C) * p.o_rstride);
            }
        }
        if (EPI == EP_GATE) {
#pragma unroll
            for (int k = 0; k < NGATE; ++k) {
                const int ch = min(mtile * 32 + min(e_r0 + k * EROWS, 31), p.C - 1);
                ring_load(pre[2 * k], p.aux + colo + (long)ch * p.aux_rstride);
                ring_load(pre[2 * k + 1], p.aux + colo + (long)(ch + p.C) * p.aux_rstride);
            }
        }
    };

    f32x4 ra0[PF], ra1[PF];
    float bq[2][4][NB];
    auto mfma_step = [&](const f32x4& A0, const f32x4& A1, const float (&bv)[4][NB]) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                acc[0][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(A0[j], bv[j][n], acc[0][n], 0, 0, 0);
                if constexpr (MB == 2)
                    acc[MB - 1][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(A1[j], bv[j][n], acc[MB - 1][n], 0, 0, 0);
            }
    };
    // Weights of this wave: two 16-row blocks, each a LINEAR stream of 1 KiB fragment blocks in exactly the
    // order the K walk consumes them ([64-channel chunk][tap][k16 in chunk], dsd_finalize_weights packs it so).
    const unsigned long long a0s = (unsigned long long)(p.A + (long)(mtile * 4 + wm * MB) * a_blk);
    const unsigned long long a1s = a0s + (unsigned long long)a_blk * 4;

    if constexpr (SW > 0) {
        // =====================================================================================
        // FAST PATH: software-pipelined over 64-channel chunks.
        //   LDS: two chunk buffers [64][SW] (+ the FiLM vector / LayerNorm statistics of the tile).
        //   chunk c+1 is fetched global -> VGPR (asm loads, hand-counted) while the MFMAs of chunk c run from
        //   LDS, then transformed and written to the other buffer; ONE barrier per chunk.  The weight ring keeps
        //   streaming linearly across chunk boundaries.  Chunks are walked in pairs so that every ring slot
        //   index and every LDS offset is a compile-time constant (K a multiple of the chunk size).
        // =====================================================================================
        static_assert(PF == 8, "ring slots are addressed statically modulo 8");
        // chunk = 64 channels (128-channel chunks for the 1x1 GEMMs were measured: the shorter K loop does not
        // pay for the longer chunk-0 fill at B = 1)
        constexpr int CR = 64;
        constexpr int ITERS = TAPS * (CR / 16);         // k16 steps per chunk: 12, 8 or 4
        constexpr int NU = CR * SW / 1024;              // float4 staged per lane and chunk (upper bound)
        constexpr int BUF = CR * SW;                    // floats per chunk buffer
        // RES: all (<= 4) chunks resident - staged in one batch in the prologue, no barrier inside the K loop.  One
        // wave per SIMD cannot hide the wait -> transform -> ds_write -> barrier -> ds_read chain of a chunk
        // boundary behind another wave's MFMAs, so for small grids (B = 1) the boundaries are removed instead.
        constexpr int NBUF = RES ? 4 : 2;
        float* lds_stat = lds + NBUF * BUF;             // ST_LN: [2][BN] (mean | rstd), HL == 0
        const int NC = p.K / CR;

        // ---- per-lane staging geometry: lane's u-th float4 of a [64 x W4] chunk tile ----
        // Only what the loads need is computed before they are issued (the prologue is latency-critical at B = 1:
        // the first activation loads of a kernel take ~1 us to land); masks and LDS offsets follow in their shadow.
        unsigned s_voff[NU];                            // byte offset from the chunk's (row 0, frame t0-HL)
        unsigned f_voff[NU];                            // ST_FILM: byte offset of the row's FiLM scalar
        int s_row[NU], s_c4[NU];
        bool s_valid[NU];
        int s_loff[NU];                                 // float offset in a chunk buffer        (filled below,
        unsigned s_mask[NU];                            // bit e: frame of element e in [0, T)     after the loads)
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int idx = tid + 256 * u;
            const bool valid = idx < CR * W4;
            const int row = valid ? fdiv_floor(idx, p.inv_w4) : CR - 1;
            const int c4 = valid ? idx - row * W4 : 0;
            s_valid[u] = valid;
            s_row[u] = row;
            s_c4[u] = c4;
            s_voff[u] = (unsigned)((row * p.b_rstride + c4 * 4) * 4);
            f_voff[u] = (unsigned)(row * p.film_cstride * 4);
        }
        unsigned long long sbase = (unsigned long long)(bsrc + (t0 - HL));      // chunk 0, row 0
        const unsigned long long sstep = (unsigned long long)p.b_rstride * CR * 4;
        constexpr int NSV = RES ? 4 : 1;
        f32x4 sv[NSV][NU];
        // ST_FILM: the step-embedding scalar of each staged row travels with the row itself (one dword per staged
        // float4, same SGPR-base + lane-offset form, covered by the same counted wait): the transform below is
        // then pure VALU on landed registers - no LDS round trip between the wait and the ds_write.
        float fadd[NSV][NU];
        unsigned long long fbase = 0;
        unsigned long long fstep = 0;
        if (STAGE == ST_FILM) {
            fbase = (unsigned long long)(p.film + p.film_col0 + b * p.film_colb);
            fstep = (unsigned long long)p.film_cstride * CR * 4;
        }
        auto stage_issue = [&](auto slotc) {
            constexpr int q = decltype(slotc)::value;
#pragma unroll
            for (int u = 0; u < NU; ++u) ring_load_s<0>(sv[q][u], s_voff[u], sbase);
            sbase += sstep;
            if (STAGE == ST_FILM) {
#pragma unroll
                for (int u = 0; u < NU; ++u)
                    asm volatile("global_load_dword %0, %1, %2" : "=v"(fadd[q][u]) : "v"(f_voff[u]), "s"(fbase) : "memory");
                fbase += fstep;
            }
        };
        auto stage_write = [&](auto slotc, float* buf) {
            constexpr int q = decltype(slotc)::value;
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                float add = 0.f;
                if (STAGE == ST_FILM) add = fadd[q][u];
                f32x4 mean = f32x4{0.f, 0.f, 0.f, 0.f}, rstd = f32x4{1.f, 1.f, 1.f, 1.f};
                if (STAGE == ST_LN) {
                    const int col = (s_loff[u] - s_row[u] * SW);
                    mean = *reinterpret_cast<const f32x4*>(&lds_stat[col]);
                    rstd = *reinterpret_cast<const f32x4*>(&lds_stat[BN + col]);
                }
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float y = sv[q][u][e];
                    if (STAGE == ST_FILM) y = y + add;
                    else if (STAGE == ST_LN) y = (y - mean[e]) * rstd[e];
                    else if (STAGE == ST_SCALE) y = y / p.in_scale;        // skip sum DIVIDED by sqrt(L) (wavenet.py:96)
                    o[e] = ((s_mask[u] >> e) & 1u) ? y : 0.f;              // zero padding AFTER the FiLM add (wavenet.py:36-38)
                }
                if (s_valid[u]) *reinterpret_cast<f32x4*>(&buf[s_loff[u]]) = o;
            }
        };

        // ---- prologue: chunk 0, the weight ring, tile constants and epilogue operands all in flight together ----
        float cst[1];                                   // LN statistics of this tile
        if (STAGE == ST_LN) {
            const int col = min(t0 + (tid & (BN - 1)), p.ln_ts - 1);
            const float* sp = p.ln_stats + (long)b * 2 * p.ln_ts + ((tid >> (NB == 1 ? 5 : 6)) & 1) * p.ln_ts + col;
            asm volatile("global_load_dword %0, %1, off" : "=v"(cst[0]) : "v"(sp) : "memory");
        }
        using I0 = std::integral_constant<int, 0>;
        using I1 = std::integral_constant<int, RES ? 1 : 0>;
        using I2 = std::integral_constant<int, RES ? 2 : 0>;
        using I3 = std::integral_constant<int, RES ? 3 : 0>;
        stage_issue(I0{});
        if constexpr (RES) {
            if (NC > 1) stage_issue(I1{});
            if (NC > 2) stage_issue(I2{});
            if (NC > 3) stage_issue(I3{});
        }
#ifndef DSD_EXP_NOEPI
        epi_prefetch();
#endif
        // weight ring prologue: blocks 0..7 of both row blocks.  Issued LAST: the vector-memory pipe of a CU moves
        // 64 B/clk, so these 16 KiB per wave take ~1 k cycles to issue - in the shadow of the activation latency -
        // and the counted wait below does not include them.
        unsigned long long an0 = a0s, an1 = a1s;        // base of the block the next refill of "step 0" loads
        const unsigned voffA = lane * 16, voffB = lane * 16 + 4096, voffC = lane * 16 + 8192;
#define DSD_RING_PRO(U)                                                              \
    ring_load_s<((U) & 3) * 1024>(ra0[U], (U) < 4 ? voffA : voffB, an0);             \
    if constexpr (MB == 2) ring_load_s<((U) & 3) * 1024>(ra1[U], (U) < 4 ? voffA : voffB, an1);
        DSD_RING_PRO(0) DSD_RING_PRO(1) DSD_RING_PRO(2) DSD_RING_PRO(3)
        DSD_RING_PRO(4) DSD_RING_PRO(5) DSD_RING_PRO(6) DSD_RING_PRO(7)
#undef DSD_RING_PRO
        an0 += 8192;
        an1 += 8192;
        DSD_STAMP(1);
        // LDS offsets and padding masks, computed while the loads fly
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            s_loff[u] = s_row[u] * SW + s_c4[u] * 4;
            unsigned m = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int t = t0 - HL + s_c4[u] * 4 + e;
                m |= (t >= 0 && t < Tb) ? (1u << e) : 0u;
            }
            s_mask[u] = m;
        }
        __builtin_amdgcn_sched_barrier(0);              // keep the geometry above the wait, in the loads' shadow
        // Staged activations, FiLM scalars and epilogue operands have landed once all but the 16 ring loads are
        // retired (in-order return).  The LN statistics (compiler load, older than all of them) are covered too.
        {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * MB) : "memory");
#pragma unroll
            for (int q = 0; q < NSV; ++q)
#pragma unroll
                for (int u = 0; u < NU; ++u) {
                    asm volatile("" : "+v"(sv[q][u])::"memory");
                    if (STAGE == ST_FILM) asm volatile("" : "+v"(fadd[q][u])::"memory");
                }
#pragma unroll
            for (int k = 0; k < NPRE; ++k) asm volatile("" : "+v"(pre[k])::"memory");
            if (STAGE == ST_LN) asm volatile("" : "+v"(cst[0])::"memory");
        }
        if (STAGE == ST_LN) {
            if (tid < 2 * BN) lds_stat[tid] = cst[0];
            __syncthreads();                            // the transform below reads the tile statistics
        }
        stage_write(I0{}, lds);
        if constexpr (RES) {
            if (NC > 1) stage_write(I1{}, lds + BUF);
            if (NC > 2) stage_write(I2{}, lds + 2 * BUF);
            if (NC > 3) stage_write(I3{}, lds + 3 * BUF);
        }
        DSD_STAMP(2);
        // the ring prologue is retired before the barrier: hipcc does not know those registers are in flight and
        // may move them once the K walk's register pressure starts (5.7: form (ii) pins order, not allocation)
        __builtin_amdgcn_s_waitcnt(0x0F70);
        DSD_STAMP(3);
        __syncthreads();
        DSD_STAMP(4);

        // ---- B fragment addressing: one VGPR base per tap, everything else is an immediate ----
        const float* bt0 = lds + lrow * SW + wn * (16 * NB) + lcol + HL - (TAPS == 3 ? p.dil : 0);   // wn == 0 when WN == 1
        const float* bt1 = bt0 + p.dil;
        const float* bt2 = bt1 + p.dil;
        auto read_b = [&](float (&bv)[4][NB], auto ic, auto bufc) {
            constexpr int i = decltype(ic)::value;       // step within the chunk: tap = i / KPC, k16 = i % KPC
            constexpr int KPC = CR / 16;
            constexpr int boff = decltype(bufc)::value * BUF;
            const float* base = (TAPS == 1 || i / KPC == 0) ? bt0 : (i / KPC == 1 ? bt1 : bt2);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int n = 0; n < NB; ++n) bv[j][n] = base[boff + ((i % KPC) * 16 + j * 4) * SW + n * 16];
        };
        // one k16 step: wait for ring slot, read the next step's B, MFMAs, refill the slot with the block 8 steps on
#define DSD_STEP(I, SLOT, BUFI)                                                                        \
    ring_wait<MB * (PF - 1)>(ra0[SLOT], ra1[SLOT]);                                                     \
    if constexpr ((I) + 1 < ITERS)                                                                      \
        read_b(bq[((I) + 1) & 1], std::integral_constant<int, ((I) + 1) % ITERS>{},                     \
               std::integral_constant<int, BUFI>{});                                                    \
    mfma_step(ra0[SLOT], ra1[SLOT], bq[(I) & 1]);                                                       \
    __builtin_amdgcn_sched_barrier(0);                                                                  \
    ring_load_s<((I) & 3) * 1024>(ra0[SLOT], (I) < 4 ? voffA : ((I) < 8 ? voffB : voffC), an0);         \
    if constexpr (MB == 2)                                                                              \
        ring_load_s<((I) & 3) * 1024>(ra1[SLOT], (I) < 4 ? voffA : ((I) < 8 ? voffB : voffC), an1);
#define DSD_CHUNK(S0, BUFI)                                                                            \
    read_b(bq[0], std::integral_constant<int, 0>{}, std::integral_constant<int, BUFI>{});               \
    DSD_STEP(0, ((S0) + 0) & 7, BUFI) DSD_STEP(1, ((S0) + 1) & 7, BUFI)                                 \
    DSD_STEP(2, ((S0) + 2) & 7, BUFI) DSD_STEP(3, ((S0) + 3) & 7, BUFI)                                 \
    if constexpr (ITERS >= 8) {                                                                         \
        DSD_STEP(4, ((S0) + 4) & 7, BUFI) DSD_STEP(5, ((S0) + 5) & 7, BUFI)                             \
        DSD_STEP(6, ((S0) + 6) & 7, BUFI) DSD_STEP(7, ((S0) + 7) & 7, BUFI)                             \
    }                                                                                                   \
    if constexpr (ITERS >= 12) {                                                                        \
        DSD_STEP(8, ((S0) + 8) & 7, BUFI) DSD_STEP(9, ((S0) + 9) & 7, BUFI)                             \
        DSD_STEP(10, ((S0) + 10) & 7, BUFI) DSD_STEP(11, ((S0) + 11) & 7, BUFI)                         \
    }                                                                                                   \
    an0 += ITERS * 1024;                                                                                \
    an1 += ITERS * 1024;
        // after a chunk's ITERS steps exactly 2*ITERS refills are younger than the staged chunk's loads
        auto stage_wait = [&]() {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(MB * ITERS) : "memory");
#pragma unroll
            for (int u = 0; u < NU; ++u) asm volatile("" : "+v"(sv[0][u])::"memory");   // consumers stay below the wait
            if (STAGE == ST_FILM) {
#pragma unroll
                for (int u = 0; u < NU; ++u) asm volatile("" : "+v"(fadd[0][u])::"memory");
            }
        };
        if constexpr (RES) {
            // every chunk is in LDS already: the K walk is one uninterrupted run of k16 steps
            DSD_CHUNK(0, 0)
            if (NC > 1) { DSD_CHUNK(ITERS & 7, 1) }
            if (NC > 2) { DSD_CHUNK(0, 2) }
            if (NC > 3) { DSD_CHUNK(ITERS & 7, 3) }
        } else
        for (int c = 0; c < NC; c += 2) {
            // even chunk from buffer 0; chunk c+1 (if any) lands in buffer 1 meanwhile
            const bool has1 = c + 1 < NC;
            DSD_STAMP2(0);
            if (has1) stage_issue(I0{});
            DSD_CHUNK(0, 0)
            DSD_STAMP2(1);
            if (has1) {
                stage_wait();
                DSD_STAMP2(2);
                stage_write(I0{}, lds + BUF);
            }
            DSD_STAMP2(3);
            __syncthreads();
            DSD_STAMP2(4);
            if (has1) {
                // odd chunk from buffer 1; chunk c+2 (if any) lands in buffer 0
                const bool has2 = c + 2 < NC;
                if (has2) stage_issue(I0{});
                DSD_CHUNK(ITERS & 7, 1)
                DSD_STAMP2(5);
                if (has2) {
                    stage_wait();
                    DSD_STAMP2(6);
                    stage_write(I0{}, lds);
                }
                DSD_STAMP2(7);
                __syncthreads();
                DSD_STAMP2(8);
            }
        }
#undef DSD_CHUNK
#undef DSD_STEP
#pragma unroll
        for (int u = 0; u < PF; ++u) ring_wait<0>(ra0[u], ra1[u]);     // drain the over-read refills
    } else {
        // =====================================================================================
        // GENERIC PATH (any K multiple of 16, any dilation): runtime LDS stride, whole chunk staged then walked.
        // A k=3 conv needs all its input channels resident (KC == K, checked on the host).
        // =====================================================================================
        const float* a0p = (const float*)a0s + lane * 4;
        const float* a1p = (const float*)a1s + lane * 4;
        long a_it = 0;                                          // linear block index of the next ring refill
        for (int kc = 0; kc < p.K; kc += p.KC) {
            const int kcn = min(p.KC, p.K - kc);
            const int n16 = kcn >> 4;
            const int nit = taps * n16;
            auto ring_issue = [&](f32x4& d0, f32x4& d1) {
                ring_load(d0, a0p + a_it * 256);
                ring_load(d1, a1p + a_it * 256);
                ++a_it;
            };
            if (kc > 0) __syncthreads();
            DSD_STAMP(1);
            // ---- stage rows [kc, kc+kcn), frames [t0-HL, t0+BN+HL): 2^lpr_shift lanes per row, SU rows in flight ----
            {
                constexpr int SU = 8;
                const int rows_per_it = 256 >> p.lpr_shift;
                const int c4 = tid & ((1 << p.lpr_shift) - 1);
                const int r_in = tid >> p.lpr_shift;
                const bool col_ok = c4 < W4;
                const int tcol = t0 - HL + c4 * 4;
                f32x4 mean = f32x4{0.f, 0.f, 0.f, 0.f}, rstd = f32x4{1.f, 1.f, 1.f, 1.f};
                if (STAGE == ST_LN) {
                    const float* st = p.ln_stats + (long)b * 2 * p.ln_ts;
                    if (col_ok && tcol >= 0 && tcol + 3 < p.ln_ts) {
                        mean = *reinterpret_cast<const f32x4*>(st + tcol);
                        rstd = *reinterpret_cast<const f32x4*>(st + p.ln_ts + tcol);
                    }
                }
                // Branch-free on purpose: every lane always loads from a clamped (in-bounds) address and the mask
                // is applied by select afterwards.  Predicated loads become exec-masked branches, and hipcc then
                // parks an s_waitcnt vmcnt(0) between consecutive loads - the batch serialises on memory latency.
                const int c4c = col_ok ? c4 : W4 - 1;
                const int tcolc = t0 - HL + c4c * 4;
                const int ch_last = p.Kreal - 1;
                f32x4 v[SU];
                float add[SU];
                auto issue = [&](int r0) {
#pragma unroll
                    for (int u = 0; u < SU; ++u) {
                        const int ch = min(kc + r0 + u * rows_per_it + r_in, ch_last);
                        v[u] = *reinterpret_cast<const f32x4*>(bsrc + (long)ch * p.b_rstride + tcolc);
                        add[u] = 0.f;
                        if (STAGE == ST_FILM) add[u] = p.film[(long)ch * p.film_cstride + p.film_col0 + b * p.film_colb];
                    }
                };
                auto finish = [&](int r0) {
#pragma unroll
                    for (int u = 0; u < SU; ++u) {
                        const int r = r0 + u * rows_per_it + r_in;
                        const bool row_ok = (kc + r) < p.Kreal;
                        f32x4 o;
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int t = tcol + e;
                            float y = v[u][e];
                            if (STAGE == ST_FILM) y = y + add[u];
                            else if (STAGE == ST_LN) y = (y - mean[e]) * rstd[e];
                            else if (STAGE == ST_SCALE) y = y / p.in_scale;   // skip sum DIVIDED by sqrt(L) (wavenet.py:96)
                            else if (STAGE == ST_LRELU) y = y >= 0.f ? y : y * p.in_scale;   // F.leaky_relu on the input
                            const bool ok = (t >= 0) && (t < Tb) && row_ok;
                            o[e] = ok ? y : 0.f;       // zero padding applies AFTER the FiLM add (wavenet.py:36-38)
                        }
                        if (col_ok && r < kcn) *reinterpret_cast<f32x4*>(&lds[r * S + c4 * 4]) = o;
                    }
                };
                const int batch = rows_per_it * SU;
                issue(0);
                __builtin_amdgcn_s_waitcnt(0x0F70);
#pragma unroll
                for (int u = 0; u < PF; ++u) ring_issue(ra0[u], ra1[u]);
                if (kc == 0) epi_prefetch();
                finish(0);
                for (int r0 = batch; r0 < kcn; r0 += batch) {
                    issue(r0);
                    __builtin_amdgcn_s_waitcnt(0x0F70);
                    finish(r0);
                }
            }
            // asm loads (ring prologue, epilogue operands) are retired before the barrier: hipcc does not know
            // their destinations are in flight and may move those registers (5.7: form (ii) pins order only)
            DSD_STAMP(2);
            __builtin_amdgcn_s_waitcnt(0x0F70);
            DSD_STAMP(3);
            __syncthreads();
            DSD_STAMP(4);
            // ---- walk: [64-channel chunk][tap][k16 in chunk]; with TAPS == 1 that is plain k16 order ----
            int k4 = 0, tap = 0, c64 = 0;
            int nk = min(4, n16);
            const float* bl0 = &lds[lrow * S + wn * (16 * NB) + lcol] + (HL - (taps >> 1) * p.dil);
            auto read_b = [&](float (&bv)[4][NB]) {
                const float* blp = bl0 + (c64 * 4 + k4) * (16 * S) + (TAPS == 1 ? 0 : tap * p.dil);
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int n = 0; n < NB; ++n) bv[j][n] = blp[j * 4 * S + n * 16];
                const bool wk = (++k4 == nk);
                k4 = wk ? 0 : k4;
                tap += wk ? 1 : 0;
                const bool wt = (tap == taps);
                tap = wt ? 0 : tap;
                c64 += wt ? 1 : 0;
                c64 = (c64 * 4 >= n16) ? 0 : c64;        // past the end: wrap to a valid, unused position
                nk = min(4, n16 - c64 * 4);
            };
            read_b(bq[0]);
            const int groups = nit / PF, rem = nit - groups * PF;
            for (int g = 0; g < groups; ++g) {
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    ring_wait<2 * (PF - 1)>(ra0[u], ra1[u]);
                    read_b(bq[(u + 1) & 1]);
                    mfma_step(ra0[u], ra1[u], bq[u & 1]);
                    __builtin_amdgcn_sched_barrier(0);      // refill only after the slot's last use has issued
                    ring_issue(ra0[u], ra1[u]);
                }
            }
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                ring_wait<0>(ra0[u], ra1[u]);       // drain: also keeps the compiler's own counting exact below
                if (u < rem) {
                    read_b(bq[(u + 1) & 1]);
                    mfma_step(ra0[u], ra1[u], bq[u & 1]);
                }
            }
            a_it -= PF - rem;    // the ring ran ahead of the blocks this chunk consumed
        }
    }

    if (EPI == EP_GATE || EPI == EP_RESSKIP) {      // prefetched operands: retired by a vmcnt(0) drain above
#pragma unroll
        for (int k = 0; k < NPRE; ++k) asm volatile("" : "+v"(pre[k])::"memory");
    }
    DSD_STAMP(5);
    // ---------------------------------------- epilogue ----------------------------------------
    // C/D layout of 16x16x4: column = lane & 15, row = (lane >> 4) * 4 + reg.
    const int rq = (lane >> 4) * 4;
    if constexpr (EPI == EP_GATE || EPI == EP_RESSKIP) {
        // LDS-staged: accumulators go to a [64 rows][BN + 4] tile (the chunk buffers are dead by now), then every
        // thread handles whole float4s of a row: the gate / residual arithmetic runs on row-major data and the
        // global stores are 16 B per lane, full lines per row - 4x fewer store instructions than the
        // fragment-shaped (4 rows x 64 B) stores of the accumulator layout.
        constexpr int ES = BN + 4;
        __syncthreads();                               // all waves are done reading the staged activations
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int n = 0; n < NB; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    // packed 16-row block j of the tile; EP_GATE: even j = gate, odd j = filter of channels (j/2)*16..
                    // -> tile rows [0,32) = gate, [32,64) = filter;  else the packed row itself
                    const int j = wm * MB + mb;
                    const int trow = (EPI == EP_GATE) ? (j & 1) * 32 + (j >> 1) * 16 + rq + r : j * 16 + rq + r;
                    lds[trow * ES + wn * (16 * NB) + n * 16 + lcol] = acc[mb][n][r];
                }
        __syncthreads();
        const long colo = (long)b * p.o_bstride + t0 + e_c4 * 4;
        if constexpr (EPI == EP_GATE) {
#pragma unroll
            for (int k = 0; k < NGATE; ++k) {
                const int cw = e_r0 + k * EROWS;           // channel within the workgroup's 32
                if (cw >= 32) continue;                    // narrow tiles: half the threads carry a channel
                const int ch = mtile * 32 + cw;
                const f32x4 g = *reinterpret_cast<const f32x4*>(&lds[cw * ES + e_c4 * 4]);
                const f32x4 f = *reinterpret_cast<const f32x4*>(&lds[(32 + cw) * ES + e_c4 * 4]);
                f32x4 z;
#pragma unroll
                for (int e = 0; e < 4; ++e)
                    z[e] = sigmoid_fast(g[e] + pre[2 * k][e]) * tanh_fast(f[e] + pre[2 * k + 1][e]);   // wavenet.py:41-42
                if (ch < p.C) *reinterpret_cast<f32x4*>(p.out + colo + (long)ch * p.o_rstride) = z;
            }
        } else {
#pragma unroll
            for (int k = 0; k < NRES; ++k) {
                const int pr = e_r0 + k * EROWS;
                const int row = mtile * 64 + pr;
                const f32x4 a4 = *reinterpret_cast<const f32x4*>(&lds[pr * ES + e_c4 * 4]);
                if (row < p.M) {
                    const float bv = p.bias ? p.bias[row] : 0.f;
                    f32x4 o;
                    if (row < p.C) {                                   // residual half (wavenet.py:47-48)
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = (pre[k][e] + (a4[e] + bv)) * 0.70710678118654752440f;   // as torch's division by a Python scalar on the GPU: times the fp32 reciprocal
                        *reinterpret_cast<f32x4*>(p.x + colo + (long)row * p.o_rstride) = o;
                    } else {                                           // skip half: running sum replaces stack+sum (wavenet.py:96)
#pragma unroll
                        for (int e = 0; e < 4; ++e) o[e] = p.first_layer ? (a4[e] + bv) : (pre[k][e] + (a4[e] + bv));
                        *reinterpret_cast<f32x4*>(p.skip + colo + (long)(row - p.C) * p.o_rstride) = o;
                    }
                }
            }
        }
    } else if constexpr (EPI == EP_LYNX_NEXT) {
        // LYNXNet layer transition (lynxnet.py:76-84 of the NEXT layer, fused into this GEMM's epilogue): residual
        // stream x, next layer's pre-LayerNorm input xin, and the LayerNorm partials of xin over this tile's 64 rows
        // (two-pass inside the tile: mean first, then squared deviations; tiles are merged by ln_merge_kernel with
        // the parallel-variance formula, so no E[x^2] - mean^2 cancellation anywhere).
        float xi[2][NB][4];
        const int nrows = min(64, p.M - mtile * 64);                 // valid rows of this tile
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            const int t = t0 + wn * (16 * NB) + n * 16 + lcol;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = mtile * 64 + (wm * 2 + mb) * 16 + rq + r;
                    const int rowc = min(row, p.M - 1);
                    float v = act_apply(acc[mb][n][r] + (p.bias ? p.bias[rowc] : 0.f), p.act);
                    if (p.aux) v = v + p.aux[(long)b * p.aux_bstride + (long)rowc * p.aux_rstride + t];
                    float xo = v, xin = v;
                    if (p.cpn) {
                        const float c = p.cpn[(long)b * p.cpn_bstride + (long)rowc * p.cpn_rstride + t];
                        xin = v + c;
                        if (p.strong) xo = xin;
                    }
                    if (p.film) xin = xin + p.film[(long)rowc * p.film_cstride + p.film_col0 + b * p.film_colb];
                    if (row < p.M) {
                        p.out[(long)b * p.o_bstride + (long)row * p.o_rstride + t] = xo;
                        if (p.out2) p.out2[(long)b * p.o_bstride + (long)row * p.o_rstride + t] = xin;
                    }
                    xi[mb][n][r] = row < p.M ? xin : 0.f;
                }
        }
        // column sums over the wave's 32 rows (8 per lane x the 4 lane groups), then over the two waves sharing wn
        float* red = lds;                                  // [4 waves][16 * NB]  (the chunk buffers are dead)
        __syncthreads();
        float mean[NB];
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                float s = 0.f;
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = mtile * 64 + (wm * 2 + mb) * 16 + rq + r;
                        const float dlt = xi[mb][n][r] - (pass ? mean[n] : 0.f);
                        s += (row < p.M) ? (pass ? dlt * dlt : dlt) : 0.f;
                    }
                s += __shfl_xor(s, 16, 64);
                s += __shfl_xor(s, 32, 64);
                if (lane < 16) red[wave * (16 * NB) + n * 16 + lane] = s;
            }
            __syncthreads();
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                // the partner wave with the same wn holds the other 32 rows
                const float tot = red[(wn) * (16 * NB) + n * 16 + lcol] + red[(2 + wn) * (16 * NB) + n * 16 + lcol];
                if (pass == 0) {
                    mean[n] = tot / (float)nrows;
                } else if (wm == 0 && lane < 16) {
                    const int t = t0 + wn * (16 * NB) + n * 16 + lane;
                    if (t < p.lnpart_ts) {
                        float* lp = p.lnpart + ((long)b * p.mtiles + mtile) * 2 * p.lnpart_ts;
                        lp[t] = mean[n];
                        lp[p.lnpart_ts + t] = tot;
                    }
                }
            }
            __syncthreads();
        }
    } else {
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        const int t = t0 + wn * (16 * NB) + n * 16 + lcol;
        if (EPI == EP_SWIGLU) {
            // packed pair block: acc[0] = first-half rows (out), acc[1] = second-half rows (gate)
            const int chb = (mtile * 2 + wm) * 16 + rq;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ch = chb + r;
                if (ch < p.C) {
                    const float u0 = acc[0][n][r] + p.bias[ch];
                    const float u1 = acc[MB - 1][n][r] + p.bias[ch + p.C];
                    p.out[(long)b * p.o_bstride + (long)ch * p.o_rstride + t] = u0 * (u1 * sigmoid_f(u1));   // out * silu(gate)
                }
            }
        } else {
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) {
                const int rowb = mtile * 64 + (wm * MB + mb) * 16 + rq;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = rowb + r;
                    if (row >= p.M) continue;
                    if (EPI == EP_SCATTER) {
                        // transposed-conv phase rows: row = r_phase * C + o  ->  out[o][up * t + r_phase]
                        const int ph = row / p.C, o = row - ph * p.C;
                        if (t < p.T)
                            p.out[(long)b * p.o_bstride + (long)o * p.o_rstride + (long)t * p.up + ph] =
                                acc[mb][n][r] + (p.bias ? p.bias[o] : 0.f);
                        continue;
                    }
                    const float v = acc[mb][n][r] + (p.bias ? p.bias[row] : 0.f);
                    if (EPI == EP_BIAS_ACT) {
                        p.out[(long)b * p.o_bstride + (long)row * p.o_rstride + t] = act_apply(v, p.act);
                    } else if (EPI == EP_BIAS_RES) {
                        const float res = p.aux[(long)b * p.aux_bstride + (long)row * p.aux_rstride + t];
                        p.out[(long)b * p.o_bstride + (long)row * p.o_rstride + t] = v + res;
                    }
                }
            }
        }
    }
    }
    if (EPI == EP_LINCOMB) {
        // Solver update fused into the last GEMM: dst_o = sum_k coef_k * src_k.  Terms are the OUTER loop and the
        // wave's 8*NB output elements the inner, unrolled one: the loads of one term are issued back to back
        // (branch-free: clamped frame index + select), so a term costs one memory latency, not one per element.
        constexpr int E = MB * NB * 4;
        float ev[E];
        int erow[E], et[E];
#pragma unroll
        for (int mb = 0; mb < MB; ++mb)
#pragma unroll
            for (int n = 0; n < NB; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int e = (mb * NB + n) * 4 + r;
                    const int row = mtile * 64 + (wm * MB + mb) * 16 + rq + r;
                    erow[e] = row;
                    et[e] = t0 + wn * (16 * NB) + n * 16 + lcol;
                    ev[e] = acc[mb][n][r] + ((p.bias && row < p.M) ? p.bias[min(row, p.M - 1)] : 0.f);
                }
        // every combination reads the PRE-evaluation buffers: all outputs are formed before any is stored
        // (a destination may also be a source of another output, e.g. UniPC's model-value slots)
        float outv[kMaxOut][E];
#pragma unroll
        for (int o = 0; o < kMaxOut; ++o) {
#pragma unroll
            for (int e = 0; e < E; ++e) outv[o][e] = 0.f;
            if (o < p.nout) {
                const LinOut& lo = p.lo[o];
                for (int k = 0; k < lo.nterms; ++k) {
                    const LinTerm tm = lo.t[k];
                    if (tm.ptr == nullptr) {
#pragma unroll
                        for (int e = 0; e < E; ++e) outv[o][e] += tm.coef * ev[e];
                    } else {
                        float sv[E];
                        const int tlim = tm.ext ? p.T - 1 : 0x7fffffff;
#pragma unroll
                        for (int e = 0; e < E; ++e)
                            sv[e] = tm.ptr[(long)b * tm.bstride + (long)min(erow[e], p.M - 1) * tm.rstride + min(et[e], tlim)];
#pragma unroll
                        for (int e = 0; e < E; ++e) outv[o][e] += tm.coef * ((et[e] <= tlim) ? sv[e] : 0.f);
                    }
                }
            }
        }
#pragma unroll
        for (int o = 0; o < kMaxOut; ++o) {
            if (o < p.nout) {
#pragma unroll
                for (int e = 0; e < E; ++e)
                    if (erow[e] < p.M)
                        p.lo[o].dst[(long)b * p.o_bstride + (long)erow[e] * p.o_rstride + et[e]] = outv[o][e];
            }
        }
    }
    DSD_STAMP(6);
}

// generic path: one resident chunk [KC][S]; fast path: two 64-row chunk buffers + the tile's FiLM vector / LN stats
int gemm_lds_bytes(int KC, int S) { return KC * S * 4; }
int gemm_fast_chunk_rows(int taps, int nb) { (void)taps; (void)nb; return 64; }
int gemm_lds_bytes_fast(int S, int stage, int taps, int K, int nb, int resident) {
    (void)K;
    return ((resident ? 4 : 2) * gemm_fast_chunk_rows(taps, nb) * S + (stage == ST_LN ? 2 * 32 * (nb ? nb : 1) : 0)) * 4;
}

template <int STAGE, int TAPS, int EPI, int NB, int SW, int RES = 0, int WN = 2, int RAG = 0>
static hipError_t set_attr() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel<STAGE, TAPS, EPI, NB, SW, RES, WN, RAG>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

template <int STAGE, int TAPS, int EPI, int NB, int SW, int RES = 0, int WN = 2, int RAG = 0>
static hipError_t launch_one(const GemmP& p, int batch, hipStream_t st) {
    const int lds = p.lds_bytes;
    dim3 grid((RAG ? p.ncg : batch * p.tiles_per_b) * p.mtiles, 1, 1);
    if (grid.x == 0) return hipSuccess;          // a ragged batch of empty items
    return launch_timed(gemm_kernel<STAGE, TAPS, EPI, NB, SW, RES, WN, RAG>, grid, dim3(256), lds, st, p,
                        "gemm_kernel<%d, %d, %d, %d, %d, %d, %d, %d>", STAGE, TAPS, EPI, NB, SW, RES, WN, RAG);
}

// fast instantiations exist for S = 48 / 80 (32-frame tiles) and 80 / 112 (64-frame tiles); 1x1 GEMMs have
// no halo, so only the smaller stride of each tile width occurs for them
bool gemm_has_fast(int taps, int nb, int S) {
    if (nb == 0) return (taps == 1 && S == 16) || (taps == 3 && S == 48);     // narrow tiles (16 frames)
    if (nb == 1) return S == 48 || (taps == 3 && S == 80);
    return S == 80 || (taps == 3 && S == 112);
}

template <int STAGE, int TAPS, int EPI, int RAG = 0>
static hipError_t dispatch(const GemmP& p, int nb, int fast, int batch, hipStream_t st) {
    if constexpr (STAGE != ST_LN && EPI != EP_SWIGLU) {
        if (nb == 0) {      // narrow tiles: fast path only (the host never asks for them otherwise)
            if constexpr (TAPS == 1) {
                if (fast == 2 && p.S == 16) return launch_one<STAGE, TAPS, EPI, 1, 16, 1, 1, RAG>(p, batch, st);
                if (fast && p.S == 16) return launch_one<STAGE, TAPS, EPI, 1, 16, 0, 1, RAG>(p, batch, st);
            } else {
                if (fast && p.S == 48) return launch_one<STAGE, TAPS, EPI, 1, 48, 0, 1, RAG>(p, batch, st);
            }
            return hipErrorInvalidValue;
        }
    }
    if (nb == 1) {
        if constexpr (TAPS == 1)        // resident variant: 1x1 GEMMs only (measured: no gain for the k=3 conv)
            if (fast == 2 && p.S == 48) return launch_one<STAGE, TAPS, EPI, 1, 48, 1, 2, RAG>(p, batch, st);
        if (fast && p.S == 48) return launch_one<STAGE, TAPS, EPI, 1, 48, 0, 2, RAG>(p, batch, st);
        if constexpr (TAPS == 3)
            if (fast && p.S == 80) return launch_one<STAGE, TAPS, EPI, 1, 80, 0, 2, RAG>(p, batch, st);
        return launch_one<STAGE, TAPS, EPI, 1, 0, 0, 2, RAG>(p, batch, st);
    }
    if (fast && p.S == 80) return launch_one<STAGE, TAPS, EPI, 2, 80, 0, 2, RAG>(p, batch, st);
    if constexpr (TAPS == 3)
        if (fast && p.S == 112) return launch_one<STAGE, TAPS, EPI, 2, 112, 0, 2, RAG>(p, batch, st);
    return launch_one<STAGE, TAPS, EPI, 2, 0, 0, 2, RAG>(p, batch, st);
}

template <int STAGE, int TAPS, int EPI, int RAG = 0>
static hipError_t attr_all() {
    hipError_t e;
    if constexpr (STAGE != ST_LN && EPI != EP_SWIGLU) {
        if constexpr (TAPS == 1) {
            if ((e = set_attr<STAGE, TAPS, EPI, 1, 16, 1, 1, RAG>()) != hipSuccess) return e;
            if ((e = set_attr<STAGE, TAPS, EPI, 1, 16, 0, 1, RAG>()) != hipSuccess) return e;
        } else {
            if ((e = set_attr<STAGE, TAPS, EPI, 1, 48, 0, 1, RAG>()) != hipSuccess) return e;
        }
    }
    if ((e = set_attr<STAGE, TAPS, EPI, 1, 0, 0, 2, RAG>()) != hipSuccess) return e;
    if ((e = set_attr<STAGE, TAPS, EPI, 2, 0, 0, 2, RAG>()) != hipSuccess) return e;
    if ((e = set_attr<STAGE, TAPS, EPI, 1, 48, 0, 2, RAG>()) != hipSuccess) return e;
    if constexpr (TAPS == 1)
        if ((e = set_attr<STAGE, TAPS, EPI, 1, 48, 1, 2, RAG>()) != hipSuccess) return e;
    if ((e = set_attr<STAGE, TAPS, EPI, 2, 80, 0, 2, RAG>()) != hipSuccess) return e;
    if constexpr (TAPS == 3) {
        if ((e = set_attr<STAGE, TAPS, EPI, 1, 80, 0, 2, RAG>()) != hipSuccess) return e;
        if ((e = set_attr<STAGE, TAPS, EPI, 2, 112, 0, 2, RAG>()) != hipSuccess) return e;
    }
    return hipSuccess;
}

// Raise the dynamic-LDS limit of every instantiation once, outside any stream capture.
hipError_t gemm_init_all() {
    hipError_t e;
    if ((e = attr_all<ST_PLAIN, 1, EP_BIAS_ACT>()) != hipSuccess) return e;
    if ((e = attr_all<ST_SCALE, 1, EP_BIAS_ACT>()) != hipSuccess) return e;
    if ((e = attr_all<ST_FILM, 3, EP_GATE>()) != hipSuccess) return e;
    if ((e = attr_all<ST_PLAIN, 1, EP_RESSKIP>()) != hipSuccess) return e;
    if ((e = attr_all<ST_PLAIN, 1, EP_LINCOMB>()) != hipSuccess) return e;
    if ((e = attr_all<ST_LN, 1, EP_SWIGLU>()) != hipSuccess) return e;
    if ((e = attr_all<ST_PLAIN, 1, EP_BIAS_RES>()) != hipSuccess) return e;
    if ((e = attr_all<ST_LN, 1, EP_LINCOMB>()) != hipSuccess) return e;
    if ((e = attr_all<ST_LN, 1, EP_BIAS_ACT>()) != hipSuccess) return e;
    if ((e = set_attr<ST_PLAIN, 1, EP_LYNX_NEXT, 1, 0>()) != hipSuccess) return e;
    if ((e = set_attr<ST_PLAIN, 1, EP_LYNX_NEXT, 2, 0>()) != hipSuccess) return e;
    if ((e = set_attr<ST_PLAIN, 1, EP_LYNX_NEXT, 1, 48>()) != hipSuccess) return e;
    if ((e = set_attr<ST_PLAIN, 1, EP_LYNX_NEXT, 2, 80>()) != hipSuccess) return e;
    if ((e = set_attr<ST_PLAIN, 0, EP_BIAS_ACT, 1, 0>()) != hipSuccess) return e;
    if ((e = set_attr<ST_PLAIN, 0, EP_BIAS_ACT, 2, 0>()) != hipSuccess) return e;
    // ragged batches (RAG = 1): every GEMM of the denoisers and of the aux decoder
    if ((e = attr_all<ST_PLAIN, 1, EP_BIAS_ACT, 1>()) != hipSuccess) return e;
    if ((e = attr_all<ST_SCALE, 1, EP_BIAS_ACT, 1>()) != hipSuccess) return e;
    if ((e = attr_all<ST_FILM, 3, EP_GATE, 1>()) != hipSuccess) return e;
    if ((e = attr_all<ST_PLAIN, 1, EP_RESSKIP, 1>()) != hipSuccess) return e;
    if ((e = attr_all<ST_PLAIN, 1, EP_LINCOMB, 1>()) != hipSuccess) return e;
    if ((e = attr_all<ST_LN, 1, EP_SWIGLU, 1>()) != hipSuccess) return e;
    if ((e = attr_all<ST_PLAIN, 1, EP_BIAS_RES, 1>()) != hipSuccess) return e;
    if ((e = attr_all<ST_LN, 1, EP_LINCOMB, 1>()) != hipSuccess) return e;
    if ((e = attr_all<ST_LN, 1, EP_BIAS_ACT, 1>()) != hipSuccess) return e;
    if ((e = set_attr<ST_PLAIN, 1, EP_LYNX_NEXT, 1, 0, 0, 2, 1>()) != hipSuccess) return e;
    if ((e = set_attr<ST_PLAIN, 1, EP_LYNX_NEXT, 2, 0, 0, 2, 1>()) != hipSuccess) return e;
    if ((e = set_attr<ST_PLAIN, 1, EP_LYNX_NEXT, 1, 48, 0, 2, 1>()) != hipSuccess) return e;
    if ((e = set_attr<ST_PLAIN, 1, EP_LYNX_NEXT, 2, 80, 0, 2, 1>()) != hipSuccess) return e;
    if ((e = set_attr<ST_PLAIN, 0, EP_BIAS_ACT, 1, 0, 0, 2, 1>()) != hipSuccess) return e;
    if ((e = set_attr<ST_PLAIN, 0, EP_BIAS_ACT, 2, 0, 0, 2, 1>()) != hipSuccess) return e;
    // NSF-HiFiGAN: leaky-ReLU staged k-tap convs, residual epilogue, transposed-conv scatter (generic path only)
    if ((e = set_attr<ST_LRELU, 0, EP_BIAS_ACT, 1, 0>()) != hipSuccess) return e;
    if ((e = set_attr<ST_LRELU, 0, EP_BIAS_ACT, 2, 0>()) != hipSuccess) return e;
    if ((e = set_attr<ST_PLAIN, 0, EP_BIAS_RES, 1, 0>()) != hipSuccess) return e;
    if ((e = set_attr<ST_PLAIN, 0, EP_BIAS_RES, 2, 0>()) != hipSuccess) return e;
    if ((e = set_attr<ST_LRELU, 0, EP_BIAS_RES, 1, 0>()) != hipSuccess) return e;
    if ((e = set_attr<ST_LRELU, 0, EP_BIAS_RES, 2, 0>()) != hipSuccess) return e;
    if ((e = set_attr<ST_LRELU, 0, EP_SCATTER, 1, 0>()) != hipSuccess) return e;
    if ((e = set_attr<ST_LRELU, 0, EP_SCATTER, 2, 0>()) != hipSuccess) return e;
    return hipSuccess;
}

#define DSD_CASE(ST, TP, EP) \
    if (stage == ST && taps == TP && epi == EP) return dispatch<ST, TP, EP>(p, nb, fast, batch, st);

hipError_t launch_gemm(const GemmP& p, int stage, int taps, int epi, int nb, int fast, int batch, hipStream_t st) {
    if (p.cgmap) {      // ragged batch: the same choices with the RAG = 1 instantiations
#define DSD_CASE_R(ST, TP, EP) \
        if (stage == ST && taps == TP && epi == EP) return dispatch<ST, TP, EP, 1>(p, nb, fast, batch, st);
        DSD_CASE_R(ST_PLAIN, 1, EP_BIAS_ACT)
        DSD_CASE_R(ST_SCALE, 1, EP_BIAS_ACT)
        DSD_CASE_R(ST_FILM, 3, EP_GATE)
        DSD_CASE_R(ST_PLAIN, 1, EP_RESSKIP)
        DSD_CASE_R(ST_PLAIN, 1, EP_LINCOMB)
        DSD_CASE_R(ST_LN, 1, EP_SWIGLU)
        DSD_CASE_R(ST_PLAIN, 1, EP_BIAS_RES)
        DSD_CASE_R(ST_LN, 1, EP_LINCOMB)
        DSD_CASE_R(ST_LN, 1, EP_BIAS_ACT)
#undef DSD_CASE_R
        if (stage == ST_PLAIN && taps == 1 && epi == EP_LYNX_NEXT && nb >= 1) {
            if (nb == 1) return (fast && p.S == 48) ? launch_one<ST_PLAIN, 1, EP_LYNX_NEXT, 1, 48, 0, 2, 1>(p, batch, st)
                                                    : launch_one<ST_PLAIN, 1, EP_LYNX_NEXT, 1, 0, 0, 2, 1>(p, batch, st);
            return (fast && p.S == 80) ? launch_one<ST_PLAIN, 1, EP_LYNX_NEXT, 2, 80, 0, 2, 1>(p, batch, st)
                                       : launch_one<ST_PLAIN, 1, EP_LYNX_NEXT, 2, 0, 0, 2, 1>(p, batch, st);
        }
        if (stage == ST_PLAIN && epi == EP_BIAS_ACT && taps == p.taps && !fast && nb >= 1)       // aux decoder's k-tap convs
            return nb == 1 ? launch_one<ST_PLAIN, 0, EP_BIAS_ACT, 1, 0, 0, 2, 1>(p, batch, st)
                           : launch_one<ST_PLAIN, 0, EP_BIAS_ACT, 2, 0, 0, 2, 1>(p, batch, st);
        return hipErrorInvalidValue;
    }
    DSD_CASE(ST_PLAIN, 1, EP_BIAS_ACT)
    DSD_CASE(ST_SCALE, 1, EP_BIAS_ACT)
    DSD_CASE(ST_FILM, 3, EP_GATE)
    DSD_CASE(ST_PLAIN, 1, EP_RESSKIP)
    DSD_CASE(ST_PLAIN, 1, EP_LINCOMB)
    DSD_CASE(ST_LN, 1, EP_SWIGLU)
    DSD_CASE(ST_PLAIN, 1, EP_BIAS_RES)
    DSD_CASE(ST_LN, 1, EP_LINCOMB)
    DSD_CASE(ST_LN, 1, EP_BIAS_ACT)
    if (stage == ST_PLAIN && taps == 1 && epi == EP_LYNX_NEXT && nb >= 1) {       // 2 x 2 layout only, no resident variant
        if (nb == 1) return (fast && p.S == 48) ? launch_one<ST_PLAIN, 1, EP_LYNX_NEXT, 1, 48>(p, batch, st)
                                                : launch_one<ST_PLAIN, 1, EP_LYNX_NEXT, 1, 0>(p, batch, st);
        return (fast && p.S == 80) ? launch_one<ST_PLAIN, 1, EP_LYNX_NEXT, 2, 80>(p, batch, st)
                                   : launch_one<ST_PLAIN, 1, EP_LYNX_NEXT, 2, 0>(p, batch, st);
    }
#define DSD_DENSE(ST, EP)                                                                   \
    if (stage == ST && epi == EP && taps == p.taps && !fast && nb >= 1)                      \
        return nb == 1 ? launch_one<ST, 0, EP, 1, 0>(p, batch, st) : launch_one<ST, 0, EP, 2, 0>(p, batch, st);
    // dense k-tap convs with the kernel size taken from the argument block (generic path)
    DSD_DENSE(ST_PLAIN, EP_BIAS_ACT)
    DSD_DENSE(ST_LRELU, EP_BIAS_ACT)
    DSD_DENSE(ST_PLAIN, EP_BIAS_RES)
    DSD_DENSE(ST_LRELU, EP_BIAS_RES)
    DSD_DENSE(ST_LRELU, EP_SCATTER)
#undef DSD_DENSE
    return hipErrorInvalidValue;
}

}  // namespace dsd
