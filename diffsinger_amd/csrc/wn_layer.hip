// Fused WaveNet residual layer for gfx950 (MI355X): ONE launch per layer for grids that can give every CU a full-row tile.
//
//   ResidualBlock.forward (modules/backbones/wavenet.py:33-48) on a tile of 32 frames x ALL channels:
//     y = dilated_conv(x + d) + cond_proj            (wavenet.py:34-38; cond_proj + both biases hoisted per utterance)
//     z = sigmoid(y[:C]) * tanh(y[C:])                (wavenet.py:41-42)
//     o = output_projection(z)                        (wavenet.py:44)
//     x' = (x + o[:C]) / sqrt(2);  skip += o[C:]      (wavenet.py:45-48; the running sum replaces stack + sum, :96)
//
// Why one kernel: as two launches (gemm.hip: conv + gate, then out-proj + residual / skip) the 8 row-tile workgroups of
// a frame tile each stage the same x tile, z makes a round trip through memory, and every layer pays two launch
// boundaries with their cold prologues and drains - at B = 8, T = 1000 that is ~15 % of the layer.  Here a workgroup owns
// 32 frames and ALL 2C output rows of both GEMMs:
//   * the C x (32 + 2 * halo) tile of x + d (zero outside [0, T), wavenet.py:36-38) is staged ONCE into LDS;
//   * 4 waves (one per SIMD) split the 2C rows: wave w owns the gate AND filter rows of channels [C/4 * w, C/4 * (w+1))
//     (the packed weight blocks of gemm.hip: even block = gate, odd block = filter of 16 channels), 2C/64 16-row blocks
//     x two 16-frame column blocks = 4C/64 accumulators per lane, initialised with the hoisted conditioner projection;
//   * the K walk is one uninterrupted stream: weights as 1 KiB fragment blocks straight from L2 into VGPRs (double
//     buffered one k16 step = 64 MFMAs = 2 k cycles ahead), activations as B fragments from the resident LDS tile;
//   * the gate runs on the accumulators, z goes to LDS over the dead x tile (C x 32), the out-proj walks it the same
//     way (wave w: rows [C/2 * w, C/2 * (w+1)): waves 0,1 the residual half, waves 2,3 the skip half);
//   * the epilogue transposes each wave's accumulators through a wave-private LDS tile, so residual / skip arithmetic and
//     the global loads / stores are row-major float4 (full 128-B lines).
// x is double-buffered across layers (a neighbouring tile's halo must read the layer's INPUT): xin -> xout.
// Per tile: 33.5 MFLOP (C = 256) on 4 SIMDs = 131 k MFMA cycles; weights 2 MB per workgroup from L2 (16 B/clk/CU),
// algorithmic HBM/fabric bytes 24 C per frame + the layer's weights once per XCD.
#include <hip/hip_ext.h>

#include "dsd_internal.h"

namespace dsd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ float sigmoid_fast(float v) { return __builtin_amdgcn_rcpf(1.f + __expf(-v)); }
__device__ __forceinline__ float tanh_fast(float v) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * v)); }
__device__ __forceinline__ int fdiv_floor(int x, float inv) { return (int)(((float)x + 0.5f) * inv); }

// 1: the FiLM values of a thread's staging rows read from LDS in one batch, the late chunks written after step 9 and the barrier
// taken after step 10 (as in wn_rowsplit.hip, finding (7) of DESIGN 4.2).  0: first version (A/B build).
#ifndef DSD_WN_LATE
#define DSD_WN_LATE 1
#endif
// 1: only step 0's weight blocks are part of the prologue's burst of loads (a CU issues one vector-memory wave-instruction per
// ~50 cycles and the burst is the prologue's critical path); step 0 issues steps 1 and 2.  0: steps 0 and 1 in the prologue.
#ifndef DSD_WN_RAMP
#define DSD_WN_RAMP 1
#endif

#ifdef DSD_STAMPS
// [workgroup][0..7]: s_memtime at the phase boundaries; [8], [9]: s_memrealtime (100 MHz) at the first and last stamp
__device__ unsigned long long g_wn_stamps[4096][10];
#define WN_STAMP(i)                                                                     \
    do {                                                                                \
        if (threadIdx.x == 0 && blockIdx.x < 4096) {                                    \
            __builtin_amdgcn_sched_barrier(0);                                          \
            g_wn_stamps[blockIdx.x][i] = __builtin_amdgcn_s_memtime();                  \
            if ((i) == 0) g_wn_stamps[blockIdx.x][8] = __builtin_amdgcn_s_memrealtime(); \
            if ((i) == 6) g_wn_stamps[blockIdx.x][9] = __builtin_amdgcn_s_memrealtime(); \
            __builtin_amdgcn_sched_barrier(0);                                          \
        }                                                                               \
    } while (0)
#else
#define WN_STAMP(i)
#endif

}  // namespace

#ifdef DSD_STAMPS
extern "C" int dsd_dbg_read_wn_stamps(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_wn_stamps), sizeof(g_wn_stamps));
}
#endif

// NCH = C / 64 (3: multi-variance nets, 4: acoustic / pitch nets, 2: C = 128 nets); SW = LDS row stride of the x tile in floats,
// 48 (halo 8: dilation <= 8) or 80 (halo 16: dilation 16), both 16 (mod 32) so the 4 k-rows x 16 columns of a B
// fragment read hit 64 distinct banks; RAG = 1: ragged batch (valid-tile list + per-item lengths, as in gemm.hip).
//
// NCB = 16-frame column blocks of the tile: 2 (32 frames, wn_layer_kernel) or 1 (16 frames, wn_layer16_kernel - round 3: for grids of
// 65 ... 128 32-frame tiles, where the 32-frame kernel would leave half the CUs idle and the two-launch forms pay two boundaries:
// B = 3, 4 at T = 1000, one utterance of 2100 ... 4096 frames.  Half the MFMAs per workgroup against the same 2 MB of weights:
// 8 bytes per clock per wave, the most a CU's vector-memory path gives - see DESIGN 4.4b).
template <int NCH, int SW, int RAG, int NCB>
__device__ __forceinline__ void wn_layer_body(const WnLayerP& p, float* lds) {
    constexpr int C = 64 * NCH;
    constexpr int MBW = 2 * NCH;                    // 16-row blocks per wave, both GEMMs (2C rows / 4 waves / 16)
    constexpr int BN = 16 * NCB;
    constexpr int F4 = BN / 4;                      // float4 per row of a row-major tile
    constexpr int RPM = 64 / F4;                    // rows a wave's 64 lanes cover per row-major float4 pass
    constexpr int HL = SW == 48 ? 8 : 16;           // staged halo columns on each side
    constexpr int W4 = (BN + 2 * HL) / 4;           // float4 per staged row
    constexpr int NU = C * W4 / 256;                // staged float4 per thread (exact for C = 192, 256)
    constexpr int SZ = 48;                          // z tile row stride
    constexpr int ES = BN + 4;                      // epilogue tile row stride
    constexpr int NS1 = NCH * 12;                   // k16 steps of the conv:  [64-channel chunk][tap][k16 in chunk]
    constexpr int NS2 = NCH * 4;                    // k16 steps of the out-proj
    static_assert(C * W4 % 256 == 0, "staging assumes a whole number of float4 per thread");
    float* xs = lds;                                 // [C][SW]: x + d tile; later z [C][SZ]
    float* es = lds + C * SW;                        // [4 waves][16 * MBW][ES]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane >> 4, lcol = lane & 15, rq = lrow * 4;
    // XCD-aware bijective remap (speed only): XCD k takes a contiguous range of tiles, so the tiles that share halo
    // columns share an L2
    const int nwg = gridDim.x;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int q8 = nwg >> 3, r8 = nwg & 7;
    const int work = (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
    const int rest = RAG ? p.cgmap[work] : work + p.tile0;
    const int b = fdiv_floor(rest, p.inv_tiles_per_b);
    const int t0 = (rest - b * p.tiles_per_b) * BN;
    const int Tb = (RAG && p.lens) ? p.lens[b] : p.T;
    const int Ts = p.Ts;
    WN_STAMP(0);

    // Every global access goes through a buffer descriptor built from wave-uniform values (kernel arguments, the tile's
    // item / first frame, the wave number): 32-bit lane offsets + SGPR offsets + immediates, no 64-bit address arithmetic.
    // The vector-memory pipe of a CU takes ~25 cycles per wave-instruction whatever its width, so the prologue issues
    // only what GEMM 1 cannot start without (x tile, FiLM vector, two weight steps: 30 instructions per wave); every
    // other operand (conditioner projection, biases, residual / skip) is fetched between the MFMAs of the K walks.
    const int bu = __builtin_amdgcn_readfirstlane(b), t0u = __builtin_amdgcn_readfirstlane(t0);
    constexpr unsigned kRange = 0x7FFFFFF0u;
    auto rsrc = [](const void* ptr) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, kRange, 0x00020000); };
    auto ld4 = [](__amdgpu_buffer_rsrc_t r, int voff, int soff) {
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
    };
    auto ld1 = [](__amdgpu_buffer_rsrc_t r, int voff, int soff) {
        return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
    };

    // ---------------- prologue: x tile, FiLM vector, first two weight steps ----------------
    const __amdgpu_buffer_rsrc_t r_x = rsrc(p.xin + (long)bu * p.x_bstride + (t0u - HL));       // inside the arena's guard at t0 = 0
    const __amdgpu_buffer_rsrc_t r_f = rsrc(p.film + p.film_col0 + bu * p.film_colb);
    const float fmine = ld1(r_f, min(tid, C - 1) * p.film_cstride * 4, 0);      // d[channel tid] of this item / step
    // Only the first 64-channel chunk of the tile (all the first 12 steps of GEMM 1 read) is fetched before the walk starts;
    // the other chunks travel one float4 per step behind its MFMAs and go to LDS after step 11.
    constexpr int NU0 = W4 / 4;                     // float4 per thread of a 64-channel chunk (idx = tid + 256 u)
    static_assert(64 * W4 == NU0 * 256 && NU - NU0 <= 12, "chunk 0 = the first NU0 staging slots; the rest fit 12 steps");
    // the late chunks' loads are all issued by step 8: one per step, or (the 80-float tile at C = 256: 12 late float4 per
    // thread) two on steps 1 .. 3 - which the step's schedule below gives a second slot
    constexpr int NLATE = NU - NU0;
    constexpr bool EARLY_LATE = DSD_WN_LATE && NLATE <= 12;
    constexpr int NDBL = NLATE > 9 ? NLATE - 9 : 0;              // steps 1 .. NDBL carry two late loads
    static_assert(NDBL <= 3, "late loads: at most two on steps 1 .. 3, one on the others up to step 8");
    f32x4 sv[NU];
    auto x_voff = [&](int u) {
        const int idx = tid + 256 * u;
        const int row = idx / W4, c4 = idx - row * W4;
        return (row * Ts + c4 * 4) * 4;
    };
#pragma unroll
    for (int u = 0; u < NU0; ++u) sv[u] = ld4(r_x, x_voff(u), 0);
    // weight streams of this wave: MBW row blocks, each a linear sequence of 1 KiB fragment blocks in K-walk order.
    // Three fragment sets in rotation: step s runs from W[s % 3] while step s + 2's MBW loads are spread between its MFMAs
    // (a set loaded during the previous step only would have a quarter of a step of cover for its last block).
    const __amdgpu_buffer_rsrc_t r_w1 = rsrc(p.Aconv + (long)(MBW * wave) * NS1 * 256);
    const __amdgpu_buffer_rsrc_t r_w2 = rsrc(p.Aout + (long)(MBW * wave) * NS2 * 256);
    int wk1[MBW], wk2[MBW];                                     // lane offset + row-block base: the step goes into soffset / imm
#pragma unroll
    for (int k = 0; k < MBW; ++k) {
        wk1[k] = lane * 16 + k * NS1 * 1024;
        wk2[k] = lane * 16 + k * NS2 * 1024;
    }
    f32x4 W[3][MBW];
    auto load_w1 = [&](f32x4 (&dst)[MBW], int s) {
#pragma unroll
        for (int k = 0; k < MBW; ++k) dst[k] = ld4(r_w1, wk1[k] + (s & 3) * 1024, (s >> 2) * 4096);
    };
    auto load_w2 = [&](f32x4 (&dst)[MBW], int s) {
#pragma unroll
        for (int k = 0; k < MBW; ++k) dst[k] = ld4(r_w2, wk2[k] + (s & 3) * 1024, (s >> 2) * 4096);
    };
    load_w1(W[0], 0);
#if !DSD_WN_RAMP
    load_w1(W[1], 1);
#endif
    WN_STAMP(1);
    // FiLM vector -> LDS (the staging region is free until the gate), so each thread can pick the scalars of its rows
    es[tid] = fmine;                                             // (threads beyond C: a copy of the last channel's, unused)
    __syncthreads();
    // FiLM add, then the zero padding (wavenet.py:36-38: the pad is applied to x + d), then LDS
#if DSD_WN_LATE
    float fav[NU];                                               // one batch of LDS reads, not one round trip per float4 staged
#pragma unroll
    for (int u = 0; u < NU; ++u) fav[u] = es[(tid + 256 * u) / W4];
#endif
    auto stage_write = [&](int u) {
        const int idx = tid + 256 * u;
        const int row = idx / W4, c4 = idx - row * W4;
#if DSD_WN_LATE
        const float fa = fav[u];
#else
        const float fa = es[row];
#endif
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int t = t0 - HL + c4 * 4 + e;
            o[e] = (t >= 0 && t < Tb) ? sv[u][e] + fa : 0.f;
        }
        *reinterpret_cast<f32x4*>(&xs[row * SW + c4 * 4]) = o;
    };
#pragma unroll
    for (int u = 0; u < NU0; ++u) stage_write(u);
    __syncthreads();
    WN_STAMP(2);

    // ---------------- GEMM 1: dilated conv, K = 3 taps x C channels ----------------
    f32x4 acc[MBW][NCB];
#pragma unroll
    for (int k = 0; k < MBW; ++k)
#pragma unroll
        for (int n = 0; n < NCB; ++n) acc[k][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    // B fragment of a k4 step: lane (lrow, lcol) holds stage(x)[channel 4j + lrow][column lcol (+16)] at the tap's shift;
    // one base register per tap, every other part of the address is an immediate
    const float* bt0 = xs + lrow * SW + HL + lcol - p.dil;
    const float* bt1 = bt0 + p.dil;
    const float* bt2 = bt1 + p.dil;
    float bq[2][4][NCB];
    auto read_b1 = [&](float (&bv)[4][NCB], int s) {               // s = k16 step: [64-channel chunk][tap][k16 in chunk]
        const int c = s / 12, i = s % 12, tap = i >> 2;
        const float* base = (tap == 0 ? bt0 : (tap == 1 ? bt1 : bt2)) + (c * 64 + (i & 3) * 16) * SW;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int n = 0; n < NCB; ++n) bv[j][n] = base[j * 4 * SW + 16 * n];
    };
    auto mfma_step = [&](const f32x4 (&w)[MBW], const float (&bv)[4][NCB]) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < MBW; ++k)
#pragma unroll
                for (int n = 0; n < NCB; ++n) acc[k][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[k][j], bv[j][n], acc[k][n], 0, 0, 0);
    };
    // One k16 step = 8 * MBW MFMAs on the current weights / B fragments, with the loads of step s + 2 (and one extra
    // operand load on some steps) and the LDS reads of step s + 1 spread between them: an MFMA holds the issue port for 8
    // of its 32 cycles, so one load behind every 8 MFMAs costs nothing, MBW loads in a row cost their issue time.
#define WN_SPREAD()                                                                  \
    __builtin_amdgcn_sched_group_barrier(0x008, 2 * NCB, 0);                               \
    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);   /* the step's extra operand load, if it has one */ \
    __builtin_amdgcn_sched_group_barrier(0x008, 2 * NCB, 0);                               \
    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                               \
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                               \
    _Pragma("unroll") for (int g_ = 1; g_ < MBW; ++g_) {                             \
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * NCB, 0);                           \
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                           \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                           \
    }                                                                                \
    __builtin_amdgcn_sched_barrier(0);
    // step 0 of a ramped ring issues TWO steps' weights (DSD_WN_RAMP): two loads behind every 8 MFMAs
#define WN_SPREAD2()                                                                 \
    __builtin_amdgcn_sched_group_barrier(0x008, 2 * NCB, 0);                               \
    __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);                               \
    __builtin_amdgcn_sched_group_barrier(0x008, 2 * NCB, 0);                               \
    __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);                               \
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                               \
    _Pragma("unroll") for (int g_ = 1; g_ < MBW; ++g_) {                             \
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * NCB, 0);                           \
        __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);                           \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                           \
    }                                                                                \
    __builtin_amdgcn_sched_barrier(0);

    // a step with TWO extra operand loads (late x chunks of the 80-float tile): one behind each of the first two MFMA groups
#define WN_SPREAD3()                                                                 \
    __builtin_amdgcn_sched_group_barrier(0x008, 2 * NCB, 0);                               \
    __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);                               \
    __builtin_amdgcn_sched_group_barrier(0x008, 2 * NCB, 0);                               \
    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                               \
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                               \
    _Pragma("unroll") for (int g_ = 1; g_ < MBW; ++g_) {                             \
        __builtin_amdgcn_sched_group_barrier(0x008, 4 * NCB, 0);                           \
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                           \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                           \
    }                                                                                \
    __builtin_amdgcn_sched_barrier(0);

    // operands fetched during GEMM 1: the hoisted conditioner projection (+ conv bias + its own bias) of this wave's rows
    // as row-major float4 (local row rl = idx >> 3: [0, 8 MBW) gate rows, [8 MBW, 16 MBW) filter rows), and the
    // output-projection bias in the accumulator layout
    constexpr int NE = 16 * MBW * F4 / 64;                      // float4 per lane over the wave's 16 * MBW rows: 16 / 12 (32 frames)
    const int orow0 = 16 * MBW * wave;                           // first output-projection row of this wave (of 2C)
    const int ev0 = ((lane / F4) * Ts + (lane % F4) * 4) * 4;   // lane's float4 of row lane / F4; + RPM rows per m
    const __amdgpu_buffer_rsrc_t r_c = rsrc(p.cp + (long)bu * p.cp_bstride + (long)(16 * NCH * wave) * Ts + t0u);
    const __amdgpu_buffer_rsrc_t r_b = rsrc(p.bias_out + orow0);
    f32x4 cpv[NE];
    f32x4 bo[MBW];
    read_b1(bq[0], 0);
    __builtin_amdgcn_sched_barrier(0);
    auto conv_step = [&](int s) {
        // step s + 2: a conv step, or one of the out-proj's first two blocks behind the last conv steps
        if (DSD_WN_RAMP && s == 0) load_w1(W[1], 1);           // (only step 0's weights are in the prologue burst)
        if (s + 2 < NS1) load_w1(W[(s + 2) % 3], s + 2);
        else load_w2(W[(s + 2) % 3], s + 2 - NS1);
        // the step's one extra operand load: the rest of the x tile first, then the conditioner projection (filter rows sit
        // C rows below the gate rows), then the out-proj bias
        if (EARLY_LATE && NDBL > 0 && s <= 8) {
            // late slot index of step s: 0 at step 0, then two per step on steps 1 .. NDBL, one per step after
            const int first = s == 0 ? 0 : (s <= NDBL ? 2 * s - 1 : s + NDBL);
            const int cnt = s == 0 ? 1 : (s <= NDBL ? 2 : 1);
            if (first < NLATE) sv[NU0 + first] = ld4(r_x, x_voff(NU0 + first), 0);
            if (cnt == 2 && first + 1 < NLATE) sv[NU0 + first + 1] = ld4(r_x, x_voff(NU0 + first + 1), 0);
        } else if (!(EARLY_LATE && NDBL > 0) && s < NLATE)
            sv[NU0 + s] = ld4(r_x, x_voff(NU0 + s), 0);
        else if (s >= 12 && s < 12 + NE)          // (one chain of else-ifs: as separate ifs hipcc no longer folds the register arrays' indices)
            cpv[s - 12] = ld4(r_c, ev0, (((s - 12) % (NE / 2)) * RPM + (s - 12 >= NE / 2 ? C : 0)) * Ts * 4);
        else if (s >= 12 + NE && s < 12 + NE + MBW)
            bo[s - 12 - NE] = ld4(r_b, rq * 4, (s - 12 - NE) * 64);
        if (EARLY_LATE || s != 11) read_b1(bq[(s + 1) & 1], s + 1 < NS1 ? s + 1 : 0);      // (after the last step: unused)
        mfma_step(W[s % 3], bq[s & 1]);
        if (DSD_WN_RAMP && s == 0) {
            WN_SPREAD2()
        } else if (EARLY_LATE && NDBL > 0 && s >= 1 && s <= NDBL) {
            WN_SPREAD3()
        } else {
            WN_SPREAD()
        }
    };
    if constexpr (EARLY_LATE) {
        // chunks 1.. of the x tile are read from step 12 on and their loads ride behind steps 0 .. 8: written after step 9, the
        // workgroup meets after step 10, step 11 fetches step 12's operands in its normal slot
#pragma unroll
        for (int s = 0; s < 10; ++s) conv_step(s);
#pragma unroll
        for (int u = NU0; u < NU; ++u) stage_write(u);
        __builtin_amdgcn_sched_barrier(0);
        conv_step(10);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 11; s < NS1; ++s) conv_step(s);
    } else {
        // (the 80-float tile's late loads ride behind steps 0 .. 11: written right before they are read)
#pragma unroll
        for (int s = 0; s < 12; ++s) conv_step(s);
#pragma unroll
        for (int u = NU0; u < NU; ++u) stage_write(u);
        __syncthreads();
        read_b1(bq[0], 12);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 12; s < NS1; ++s) conv_step(s);
    }
    static_assert(12 + NE + MBW <= NS1, "one extra operand load per step");
    WN_STAMP(3);

    // ---------------- gate (wavenet.py:41-42); z -> LDS over the dead x tile ----------------
    // conditioner projection: row-major registers -> wave-private LDS tile -> accumulator layout
    float* ew = es + wave * (16 * MBW * ES);                     // [16 * MBW][ES]
#pragma unroll
    for (int m = 0; m < NE; ++m) {
        const int idx = lane + 64 * m;
        *reinterpret_cast<f32x4*>(&ew[(idx / F4) * ES + (idx % F4) * 4]) = cpv[m];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");       // same wave wrote what it reads: LDS is in order per wave
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float zr[NCH][NCB][4];
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int n = 0; n < NCB; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float cg = ew[(i * 16 + rq + r) * ES + n * 16 + lcol];
                const float cf = ew[(8 * MBW + i * 16 + rq + r) * ES + n * 16 + lcol];
                zr[i][n][r] = sigmoid_fast(acc[2 * i][n][r] + cg) * tanh_fast(acc[2 * i + 1][n][r] + cf);
            }
    __syncthreads();                                             // every wave is done reading the x tile
    float* zs = xs;
#pragma unroll
    for (int i = 0; i < NCH; ++i)
#pragma unroll
        for (int n = 0; n < NCB; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                zs[((NCH * wave + i) * 16 + rq + r) * SZ + n * 16 + lcol] = zr[i][n][r];
#pragma unroll
    for (int k = 0; k < MBW; ++k)                                // GEMM 2 starts from its bias
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int n = 0; n < NCB; ++n) acc[k][n][r] = bo[k][r];
    __syncthreads();
    WN_STAMP(4);

    // ---------------- GEMM 2: output projection, K = C ----------------
    // epilogue operands (row-major float4 of the wave's 16 * MBW output rows): residual stream x for waves 0, 1, running
    // skip sum for waves 2, 3 - fetched between the MFMAs
    f32x4 pre[NE];
    const bool is_res = orow0 < C;                               // wave-uniform
    const long eoff = (long)bu * p.x_bstride + (long)(is_res ? orow0 : orow0 - C) * Ts + t0u;
    // (pointer chosen with integer arithmetic: a select between the struct FIELDS makes hipcc load the pointer itself
    // through a dependent vector load)
    const unsigned long long xa = (unsigned long long)p.xin, sa = (unsigned long long)p.skip;
    const __amdgpu_buffer_rsrc_t r_e = rsrc((const float*)(is_res ? xa : sa) + eoff);
    const float* zt = zs + lrow * SZ + lcol;
    auto read_b2 = [&](float (&bv)[4][NCB], int s) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int n = 0; n < NCB; ++n) bv[j][n] = zt[(s * 16 + j * 4) * SZ + 16 * n];
    };
    static_assert(NS1 % 3 == 0 && NE <= NS2, "buffer rotation continues across the two GEMMs; one operand load per step");
    read_b2(bq[0], 0);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < NS2; ++s) {
        if (s + 2 < NS2) load_w2(W[(s + 2) % 3], s + 2);
        if (s < NE) pre[s] = ld4(r_e, ev0, s * RPM * Ts * 4);
        read_b2(bq[(s + 1) & 1], s + 1 < NS2 ? s + 1 : 0);
        mfma_step(W[s % 3], bq[s & 1]);
        WN_SPREAD()
    }
#undef WN_SPREAD
#undef WN_SPREAD2
#undef WN_SPREAD3
    WN_STAMP(5);

    // ---------------- epilogue: residual / skip (wavenet.py:45-48) ----------------
#pragma unroll
    for (int k = 0; k < MBW; ++k)
#pragma unroll
        for (int n = 0; n < NCB; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) ew[(k * 16 + rq + r) * ES + n * 16 + lcol] = acc[k][n][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {
        const unsigned long long xo = (unsigned long long)p.xout;
        const dsd_i32x4 w_o = dsd_rsrc_words((const float*)(is_res ? xo : sa) + eoff);
        // (x + o) / sqrt(2) as a multiplication by the fp32 reciprocal, the way torch's CUDA division by a Python scalar
        // evaluates it (the IEEE division sequence is ~10 VALU operations per element: 4 k cycles of this epilogue);
        // the first layer's skip sum is the layer's own output (the buffer holds the previous evaluation's sum)
        const float scale = is_res ? 0.70710678118654752440f : 1.f;
        const bool add_pre = is_res || !p.first_layer;
#pragma unroll
        for (int m = 0; m < NE; ++m) {
            const int idx = lane + 64 * m;
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(&ew[(idx / F4) * ES + (idx % F4) * 4]);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = ((add_pre ? pre[m][e] : 0.f) + a4[e]) * scale;
            dsd_store_b128<DSD_ST_AUX>(__builtin_bit_cast(dsd_u32x4, o), w_o, ev0, m * RPM * Ts * 4);
        }
    }
    WN_STAMP(6);
}

template <int NCH, int SW, int RAG>
__global__ __launch_bounds__(256, 1) void wn_layer_kernel(const WnLayerP p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    wn_layer_body<NCH, SW, RAG, 2>(p, lds);
}
template <int NCH, int SW, int RAG>
__global__ __launch_bounds__(256, 1) void wn_layer16_kernel(const WnLayerP p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    wn_layer_body<NCH, SW, RAG, 1>(p, lds);
}

int wn_layer_lds_bytes(int nch, int sw, int ncb) { return (64 * nch * sw + 4 * 16 * 2 * nch * (16 * ncb + 4)) * 4; }

bool wn_layer_supported(int C, int dil) { return (C == 256 || C == 192 || C == 128) && dil >= 1 && dil <= 16; }

template <int NCH, int SW, int RAG, int NCB>
static hipError_t wn_launch(const WnLayerP& p, int ntiles, hipStream_t st) {
    const int ldsb = wn_layer_lds_bytes(NCH, SW, NCB);
    static bool attr_done = false;       // per instantiation; set outside any capture by wn_layer_init_all
    void (*kern)(const WnLayerP);
    if constexpr (NCB == 2) kern = wn_layer_kernel<NCH, SW, RAG>;
    else kern = wn_layer16_kernel<NCH, SW, RAG>;
    if (!attr_done) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    if (ntiles == 0) return hipSuccess;
    return launch_timed(kern, dim3(ntiles), dim3(256), ldsb, st, p, NCB == 2 ? "wn_layer_kernel<%d, %d, %d>" : "wn_layer16_kernel<%d, %d, %d>", NCH, SW, RAG);
}

bool wn_layer16_supported(int C, int dil) { return (C == 256 || C == 192) && dil >= 1 && dil <= 16; }

// bn = frames per tile: 32, or 16 (wn_layer16_kernel; p.tiles_per_b / tile0 / ntiles / cgmap count tiles of that width)
hipError_t launch_wn_layer(const WnLayerP& p, int C, int batch, hipStream_t st, int bn) {
    const int sw = p.dil <= 8 ? 48 : 80;
    const int ntiles = p.cgmap ? p.ncg : (p.ntiles > 0 ? p.ntiles : batch * p.tiles_per_b);
    if (bn != 32 && !(bn == 16 && wn_layer16_supported(C, p.dil))) return hipErrorInvalidValue;
#define WN_CASE(NCH_, SW_)                                                                                  \
    if (C == 64 * NCH_ && sw == SW_ && bn == 32)                                                            \
        return p.cgmap ? wn_launch<NCH_, SW_, 1, 2>(p, ntiles, st) : wn_launch<NCH_, SW_, 0, 2>(p, ntiles, st);
#define WN_CASE16(NCH_, SW_)                                                                                \
    if (C == 64 * NCH_ && sw == SW_ && bn == 16)                                                            \
        return p.cgmap ? wn_launch<NCH_, SW_, 1, 1>(p, ntiles, st) : wn_launch<NCH_, SW_, 0, 1>(p, ntiles, st);
    WN_CASE(4, 48)
    WN_CASE(4, 80)
    WN_CASE(3, 48)
    WN_CASE(3, 80)
    WN_CASE(2, 48)
    WN_CASE(2, 80)
    WN_CASE16(4, 48)
    WN_CASE16(4, 80)
    WN_CASE16(3, 48)
    WN_CASE16(3, 80)
#undef WN_CASE
#undef WN_CASE16
    return hipErrorInvalidValue;
}

// raise the dynamic-LDS limit of every instantiation once, outside any stream capture
hipError_t wn_layer_init_all() {
    WnLayerP p{};
    hipError_t e;
#define WN_INIT(NCH_, SW_, NCB_)                                                                  \
    if ((e = wn_launch<NCH_, SW_, 0, NCB_>(p, 0, nullptr)) != hipSuccess) return e;               \
    if ((e = wn_launch<NCH_, SW_, 1, NCB_>(p, 0, nullptr)) != hipSuccess) return e;
    WN_INIT(4, 48, 2)
    WN_INIT(4, 80, 2)
    WN_INIT(3, 48, 2)
    WN_INIT(3, 80, 2)
    WN_INIT(2, 48, 2)
    WN_INIT(2, 80, 2)
    WN_INIT(4, 48, 1)
    WN_INIT(4, 80, 1)
    WN_INIT(3, 48, 1)
    WN_INIT(3, 80, 1)
#undef WN_INIT
    return hipSuccess;
}

}  // namespace dsd
