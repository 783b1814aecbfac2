// WaveNet residual layer as TWO launches for grids too small for full-row tiles (one utterance: B = 1, T ~ 1000 is 32
// frame tiles of 32 frames for 256 CUs): the 2C output rows of each GEMM are split over 2C / 64 workgroups per frame tile.
//
//   wn_conv_rs_kernel : z = sigmoid(y[:C]) * tanh(y[C:]),  y = dilated_conv(x + d) + cond_proj    (wavenet.py:34-42)
//   wn_out_rs_kernel  : o = output_projection(z);  x = (x + o[:C]) / sqrt(2);  skip += o[C:]       (wavenet.py:44-48, :96)
//
// Same structure as the fused kernel of wn_layer.hip (buffer-descriptor loads, compiler-counted waits, operand loads
// spread between the MFMAs), shaped for the small grid:
//   * tile = 64 packed rows x 32 frames, 4 waves as 4 (rows) x 1: a wave owns ONE 16-row block and both 16-frame column
//     blocks.  The 2 x 2 layout of gemm.hip has the two waves of a row pair stream the same weight blocks (twice the
//     vector-memory instructions - and at B = 1 a CU's loop is bound by how many of them its memory pipe takes, ~1 per 50
//     cycles - plus 8 over-read blocks per ring); here every weight block is loaded once: 48 per wave for the conv.
//   * the weight stream runs DEPTH - 1 steps ahead through a register rotation (a k16 step is only 8 MFMAs); DEPTH = 3: see
//     its definition - the first version's 6 put five blocks per wave into the latency-critical prologue burst.
//   * gate / filter rows of a channel sit in different waves: the gate runs after an LDS transpose of the accumulators,
//     row-major, with the conditioner projection fetched as float4 during the K walk.
#include <hip/hip_ext.h>

#include <cstdlib>
#include <type_traits>

#include "dsd_internal.h"

namespace dsd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

__device__ __forceinline__ float sigmoid_fast(float v) { return __builtin_amdgcn_rcpf(1.f + __expf(-v)); }
__device__ __forceinline__ float tanh_fast(float v) { return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + __expf(2.f * v)); }
__device__ __forceinline__ int fdiv_floor(int x, float inv) { return (int)(((float)x + 0.5f) * inv); }
// byte offset of a row as a 24-bit multiply (rows < 512; the host keeps Ts below 2^22): a 32-bit `row * Ts + c` compiles to
// v_mad_u64_u32, whose 64-bit addend has an undefined high half - the register allocator parked it on a register with a load
// in flight (the FiLM value) and the hardware dependency put an s_waitcnt vmcnt(0) in front of the x-tile loads.
template <int B4>
__device__ __forceinline__ int div_b4(int x) { return B4 == 8 ? x >> 3 : fdiv_floor(x, 1.0f / B4); }     // x / B4, x < 2^16
__device__ __forceinline__ int row_ts(int row, int Ts) { return (int)__umul24((unsigned)row, (unsigned)(Ts * 4)); }   // BYTES

constexpr unsigned kRange = 0x7FFFFFF0u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void* ptr) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(ptr), 0, kRange, 0x00020000);
}
__device__ __forceinline__ f32x4 ld4(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
__device__ __forceinline__ float ld1(__amdgpu_buffer_rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void st4(f32x4 v, dsd_i32x4 r, int voff, int soff) {
    dsd_store_b128<DSD_ST_AUX>(__builtin_bit_cast(dsd_u32x4, v), r, voff, soff);
}
// The conv's z is read back by the very next launch: kept in L2 (plain).  Same-box A/B of the 50-NFE loop at B = 1, three runs
// each (tools/ab_store.sh): x / skip write-through + z plain 16.52 ms, both write-through 16.55, x / skip plain + z
// write-through 16.70, both plain 16.67.
#ifndef DSD_ST_AUX_Z
#define DSD_ST_AUX_Z 0
#endif
__device__ __forceinline__ void st4z(f32x4 v, dsd_i32x4 r, int voff, int soff) {
    dsd_store_b128<DSD_ST_AUX_Z>(__builtin_bit_cast(dsd_u32x4, v), r, voff, soff);
}

#ifdef DSD_STAMPS
// [kernel 0 = conv, 1 = out][workgroup][0..6]: s_memtime at the phase boundaries; [8], [9]: s_memrealtime at the first / last
__device__ unsigned long long g_rs_stamps[2][4096][40];      // [10 + s]: after local step s of the conv walk
#define RS_STAMP(K, i)                                                                          \
    do {                                                                                        \
        if (threadIdx.x == 0 && blockIdx.x < 4096) {                                            \
            __builtin_amdgcn_sched_barrier(0);                                                  \
            g_rs_stamps[K][blockIdx.x][i] = __builtin_amdgcn_s_memtime();                       \
            if ((i) == 0) g_rs_stamps[K][blockIdx.x][8] = __builtin_amdgcn_s_memrealtime();     \
            if ((i) == 5) g_rs_stamps[K][blockIdx.x][9] = __builtin_amdgcn_s_memrealtime();     \
            __builtin_amdgcn_sched_barrier(0);                                                  \
        }                                                                                       \
    } while (0)
#else
#define RS_STAMP(K, i)
#endif

// Every field of the argument block in SGPRs behind ONE batch of scalar loads at the top of the kernel: left to itself the
// compiler fetches them in two or three dependent batches (each a cold scalar-cache round trip) before the first vector load
// can issue - on kernels whose whole life is 8-20 k cycles.  -DDSD_RS_PIN_ARGS=0: A/B build.
#ifndef DSD_RS_FILM4
#define DSD_RS_FILM4 1
#endif
#ifndef DSD_RS_PIN_ARGS
#define DSD_RS_PIN_ARGS 1
#endif
__device__ __forceinline__ void rs_pin_args(const WnLayerP& p) {
#if DSD_RS_PIN_ARGS
    asm volatile("" ::"s"(p.Aconv), "s"(p.Aout), "s"(p.bias_out), "s"(p.xin), "s"(p.xout), "s"(p.skip), "s"(p.z), "s"(p.x_bstride),
                 "s"(p.Ts), "s"(p.cp), "s"(p.cp_bstride), "s"(p.film), "s"(p.film_cstride), "s"(p.film_col0), "s"(p.film_colb),
                 "s"(p.dil), "s"(p.T), "s"(p.tiles_per_b), "s"(p.first_layer), "s"(p.inv_tiles_per_b), "s"(p.tile0),
                 "s"((int)gridDim.x));                           // (the grid size is an implicit argument: same segment)
#endif
}

// Weight fragments in rotation: step s runs from W[s % DEPTH], step s + DEPTH - 1 is in flight.  The prologue of these short
// kernels is bound by how many wave-level loads a CU can issue before the walk starts, and DEPTH - 1 blocks per wave are part of
// that burst: same-box scan of the 50-NFE loop at the headline (tools/ab_flags.sh) - depth 8: 16.08 ms, 6: 15.73, 5: 15.73,
// 4: 15.37, 3: 15.28, 2: 15.55; with separate depths for the two kernels (3, 3) 15.29, (3, 2) 15.36, (4, 3) 15.31, (3, 4) 15.33;
// depth 3 also wins at B = 2 (24.72 -> 24.02), T = 2048 (24.69 -> 23.96), T = 900 and on the pitch network (tools/ab_depth.sh).
#ifndef DSD_RS_DEPTH
#define DSD_RS_DEPTH 3
#endif
// (Tried: the two MFMA operand registers swapped - they have the same lane pattern, so D^T = X^T W^T needs no other change and a
// lane's four accumulator values become four consecutive FRAMES of one row: the tails' LDS transposes are then one
// ds_write_b128 per accumulator instead of four ds_write_b32, bit-identical results.  No effect on the loop time, here
// (14.76 = 14.75 ms) or in the fused kernel's gate / epilogue (67.3 = 67.3 ms at B = 8): those phases are not LDS-issue bound.)
#ifndef DSD_RS_DEPTH_OUT
#define DSD_RS_DEPTH_OUT DSD_RS_DEPTH
#endif
// Conv: the late chunks go to LDS after local step DSD_RS_LATE_W and the workgroup meets after step DSD_RS_LATE_B; step 12 is
// the first to read them and its operands are fetched during step 11 like any other step's.  (11, 11) is the first version:
// write, barrier and step 12's LDS reads back to back, all of it exposed.  DSD_RS_FA_BATCH: the FiLM values of a thread's
// staging rows are read from LDS in one batch (left inline, every ds_write_b128 waited for its own ds_read_b32 round trip).
// Same-box A/B at the headline (tools/ab_flags.sh, ms per 50-NFE loop): first version 15.08; batch alone 14.99; batch +
// (W, B) = (5, 10) 14.96, (8, 10) 14.97, (6, 7) 14.97, (4, 5) 14.97, (8, 8) 15.00.
#ifndef DSD_RS_LATE_W
#define DSD_RS_LATE_W 5
#endif
#ifndef DSD_RS_LATE_B
#define DSD_RS_LATE_B 10
#endif
#ifndef DSD_RS_FA_BATCH
#define DSD_RS_FA_BATCH 1
#endif
// (Measured and removed.  Out-proj: only the two 64-channel chunks that the first four steps of either K half read staged before
// the walk, the other two written after step 1 behind a barrier after step 2 - the walk is 8 steps and the extra barrier costs
// what the earlier start gains, 14.99 against 14.97 ms.  Both kernels: the loads spread over the walk's first steps - late x
// chunks, conditioner projection, residual / skip operand - issued in the prologue right behind the first weight blocks:
// 15.44 against 14.98 ms per loop, ten more wave-level loads per wave in the burst before the walk.)
// The K-half conv's ring can RAMP: DEPTH - 1 blocks per wave in the prologue burst, then two blocks per step until
// DSD_RS_DEPTH_MAX - 1 are in flight (= DEPTH: no ramp; measured equal for 4 .. 8).
// Out-proj: 1 = one weight block less in the prologue's burst (DEPTH_OUT - 2), step 0 issues two (as the K-quarter conv does)
#ifndef DSD_RS_OUT_RAMP
#define DSD_RS_OUT_RAMP 1
#endif
#ifndef DSD_RS_DEPTH_MAX
#define DSD_RS_DEPTH_MAX DSD_RS_DEPTH
#endif
// TIMING-ONLY diagnostic builds of the conv walk (wrong results): bit 0 no weight loads inside the walk, bit 1 no B-fragment
// LDS reads inside the walk, bit 2 no late chunks (loads, LDS writes, barrier), bit 3 no conditioner-projection loads
#ifndef DSD_RS_DIAG
#define DSD_RS_DIAG 0
#endif
constexpr int DEPTH_OUT = DSD_RS_DEPTH_OUT;   // ... of the out-proj kernel (8 steps per wave)
constexpr int DEPTH = DSD_RS_DEPTH;           // ... of the conv kernel (24 steps per wave)
constexpr int DMAX = DSD_RS_DEPTH_MAX;
static_assert(DMAX >= DEPTH, "the ring ramps up, not down");
// blocks issued before local step s of an n-step walk: DEPTH - 1 in the prologue, then at most two per step and never past
// block s + DMAX - 1 (whose slot the step before freed)
constexpr int rs_issued_before(int s, int n) {
    int c = DEPTH - 1;
    for (int i = 0; i < s; ++i) {
        int lim = i + DMAX;
        if (lim > n) lim = n;
        c = c + 2 < lim ? c + 2 : lim;
    }
    return c < n ? c : n;
}

// the walk's steps with their index a constant expression (ring slots, the issue plan above)
template <int I, int N, typename F>
__device__ __forceinline__ void rs_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        rs_static_for<I + 1, N>(f);
    }
}

// XCD-aware bijective remap (speed only): an XCD takes a contiguous range of work items, row tile fastest, so the row
// tiles of a frame tile - which stage the same activations - share an L2
__device__ __forceinline__ int xcd_work() {
    const int nwg = gridDim.x;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int q8 = nwg >> 3, r8 = nwg & 7;
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
}

}  // namespace

#ifdef DSD_STAMPS
extern "C" int dsd_dbg_read_rs_stamps(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_rs_stamps), sizeof(g_rs_stamps));
}
#endif

// One k16 step = 8 MFMAs on two accumulators, written in the order acc0, acc1, acc0, ...: a dependent
// v_mfma_f32_16x16x4_f32 can issue 40 cycles after its predecessor, an independent one after 32, so the two chains must
// ALTERNATE (left to itself the scheduler groups each chain: 8 x 40 instead of 8 x 32 cycles per step - measured).  Every
// MFMA is therefore pinned by a scheduling barrier, with the step's other instructions placed by hand behind it: the
// weight load for step s + 5 and an operand load behind the first two, one LDS read pair of step s + 1 behind each of
// the first four.
#define RS_PIN() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ f32x4 rs_mfma(float wfrag, float xfrag, f32x4 acc) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(wfrag, xfrag, acc, 0, 0, 0);
}

// ---------------------------------------------------------------------------------------------------------------
// Both kernels: C = 256 (NCH = 4 chunks of 64 channels), 512 threads = 8 waves = two K HALVES of four row waves: wave
// (kh, w) walks half kh of the k16 steps for row block w, the halves' accumulators are added through LDS at the end.
// Why two waves per SIMD: a k16 step is 8 MFMAs and 4 LDS read pairs, and a wave alone on its SIMD pays ~20 cycles per
// LDS read instruction on top of the MFMAs (measured: 357 cycles per step, 279 with the reads removed, 256 = the MFMAs);
// a partner wave's MFMAs fill those bubbles.  The weight loads cost nothing (measured), so every wave still streams its
// own blocks.
// ---------------------------------------------------------------------------------------------------------------
constexpr int NCH = 4, C = 256, MT = 8, BN = 32, ES = 36;

// SW = LDS row stride of the x tile (48: halo 8, dilation <= 8; 80: halo 16, dilation 16); RAG: ragged batch
template <int SW, int RAG>
__global__ __launch_bounds__(512, 2) void wn_conv_rs_kernel(const WnLayerP p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    RS_STAMP(0, 6);                                              // (diagnostic builds: before any argument is touched ...
    rs_pin_args(p);
    RS_STAMP(0, 7);                                              //  ... and with all of them in SGPRs)
    constexpr int HL = SW == 48 ? 8 : 16;
    constexpr int W4 = (BN + 2 * HL) / 4;
    constexpr int NE = 128 * W4 / 512;              // float4 per thread of two 64-channel chunks: 3 (SW 48), 4 (SW 80)
    constexpr int NS = NCH * 12;                    // k16 steps: [64-channel chunk][tap][k16 in chunk]
    constexpr int NH = NS / 2;                      // per K half: chunks {0, 1} / {2, 3}
    float* xs = lds;                                 // [C][SW]
    float* et = lds + C * SW;                        // [64][ES]: FiLM vector first, accumulator transpose last
    float* red = et + 64 * ES;                       // [4 row waves][2][64 lanes][4]: the second half's accumulators

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = wave >> 2, w = wave & 3;
    const int lrow = lane >> 4, lcol = lane & 15, rq = lrow * 4;
    const int work = xcd_work();
    const int rest0 = work / MT, mtile = work - rest0 * MT;
    const int rest = RAG ? p.cgmap[rest0] : rest0 + p.tile0;
    const int b = fdiv_floor(rest, p.inv_tiles_per_b);
    const int t0 = (rest - b * p.tiles_per_b) * BN;
    const int Tb = (RAG && p.lens) ? p.lens[b] : p.T;
    const int Ts = p.Ts;
    const int bu = __builtin_amdgcn_readfirstlane(b), t0u = __builtin_amdgcn_readfirstlane(t0);
    RS_STAMP(0, 0);

    // ---------------- prologue: the chunks the first 12 steps of either half read (0 and 2), FiLM vector, weights ----------------
    const __amdgpu_buffer_rsrc_t r_x = rsrc(p.xin + (long)bu * p.x_bstride + (t0u - HL));
    const __amdgpu_buffer_rsrc_t r_f = rsrc(p.film + p.film_col0 + bu * p.film_colb);
#if DSD_RS_FILM4
    float fmine = 0.f;                                           // 256 values: the first four waves fetch them (wave-uniform branch)
    if (wave < 4) fmine = ld1(r_f, tid * p.film_cstride * 4, 0);
#else
    const float fmine = ld1(r_f, (tid & 255) * p.film_cstride * 4, 0);
#endif
    // staging slot u of a thread: float4 (row, c4) of a 128-row set; `late` = 0: chunks 0 and 2 (rows [0,64) + [128,192)),
    // 1: chunks 1 and 3 - fetched one float4 per step behind the first steps' MFMAs, written to LDS after step 11
    auto x_row = [&](int u, int late) {
        const int e = tid + 512 * u;
        const int re = e / W4;
        return re + (re & 64) + 64 * late;
    };
    auto x_c4 = [&](int u) {
        const int e = tid + 512 * u;
        return e - (e / W4) * W4;
    };
    f32x4 sv[NE], svl[NE];
#pragma unroll
    for (int u = 0; u < NE; ++u) sv[u] = ld4(r_x, row_ts(x_row(u, 0), Ts) + x_c4(u) * 16, 0);
    RS_PIN();                                                    // (issue order = return order: what the walk needs first, first)
    // this wave's row block: packed block 4 * mtile + w (even: gate rows, odd: filter rows of 16 channels), steps [NH kh, +NH)
    const __amdgpu_buffer_rsrc_t r_w = rsrc(p.Aconv + ((long)(4 * mtile + w) * NS + NH * kh) * 256);
    const int wl = lane * 16;
    f32x4 W[DMAX];
#pragma unroll
    for (int s = 0; s < DEPTH - 1; ++s) W[s] = ld4(r_w, wl + (s & 3) * 1024, (s >> 2) * 4096);
    RS_PIN();
    // the hoisted conditioner projection (+ biases) of this tile's 32 channels, row-major float4 for the gate below:
    // thread (of the first 256) -> channel tid >> 3, frames 4 * (tid & 7)
    const int gch = 32 * mtile + ((tid & 255) >> 3);
    const __amdgpu_buffer_rsrc_t r_c = rsrc(p.cp + (long)bu * p.cp_bstride + t0u);
    f32x4 cpg = f32x4{0.f, 0.f, 0.f, 0.f}, cpf = cpg;
    RS_STAMP(0, 1);
#if DSD_RS_FILM4
    if (wave < 4) et[tid] = fmine;
#else
    et[tid & 255] = fmine;
#endif
    __syncthreads();
#if DSD_RS_FA_BATCH
    float fa0[NE], fa1[NE];
#pragma unroll
    for (int u = 0; u < NE; ++u) {
        fa0[u] = et[x_row(u, 0)];
        fa1[u] = et[x_row(u, 1)];
    }
#endif
    auto stage_write = [&](const f32x4& v, int u, int late) {    // FiLM add, then the zero padding (wavenet.py:36-38), then LDS
        const int row = x_row(u, late), c4 = x_c4(u);
#if DSD_RS_FA_BATCH
        const float fa = late ? fa1[u] : fa0[u];
#else
        const float fa = et[row];
#endif
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int t = t0 - HL + c4 * 4 + e;
            o[e] = (t >= 0 && t < Tb) ? v[e] + fa : 0.f;
        }
        *reinterpret_cast<f32x4*>(&xs[row * SW + c4 * 4]) = o;
    };
#pragma unroll
    for (int u = 0; u < NE; ++u) stage_write(sv[u], u, 0);
    __syncthreads();
    RS_STAMP(0, 2);

    // ---------------- K walk: local step sl of half kh = global step NH kh + sl ----------------
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    const float* bt0 = xs + (kh * 128 + lrow) * SW + HL + lcol - p.dil;
    const float* bt1 = bt0 + p.dil;
    const float* bt2 = bt1 + p.dil;
    float bq[2][4][2];
    auto read_b1 = [&](float (&bv)[4][2], int sl, int j) {       // both column blocks of k4 step j of local step sl
        const int c = sl / 12, i = sl % 12, tap = i >> 2;
        const float* base = (tap == 0 ? bt0 : (tap == 1 ? bt1 : bt2)) + (c * 64 + (i & 3) * 16 + j * 4) * SW;
        bv[j][0] = base[0];
        bv[j][1] = base[16];
    };
#pragma unroll
    for (int j = 0; j < 4; ++j) read_b1(bq[0], 0, j);
    RS_PIN();
    rs_static_for<0, NH>([&](auto sc) __attribute__((always_inline)) {
        constexpr int s = decltype(sc)::value;
        const f32x4 wv = W[(DSD_RS_DIAG & 1) ? s % (DEPTH - 1) : s % DMAX];
        float (&bc)[4][2] = bq[(DSD_RS_DIAG & 2) ? 0 : s & 1];
        float (&bn)[4][2] = bq[(s + 1) & 1];
        constexpr int nb0 = rs_issued_before(s, NH), nb1 = rs_issued_before(s + 1, NH);     // this step issues blocks [nb0, nb1)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[0] = rs_mfma(wv[j], bc[j][0], acc[0]);
            if (!(DSD_RS_DIAG & 1) && j < 2 && nb0 + j < nb1) W[(nb0 + j) % DMAX] = ld4(r_w, wl + ((nb0 + j) & 3) * 1024, ((nb0 + j) >> 2) * 4096);
            if (!(DSD_RS_DIAG & 2) && j == 0 && !(DSD_RS_LATE_B == 11 && s == 11) && s + 1 < NH) {      // the next step's 4 LDS read pairs in one burst
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) read_b1(bn, s + 1, jj);
            }
            RS_PIN();
            acc[1] = rs_mfma(wv[j], bc[j][1], acc[1]);
            if (!(DSD_RS_DIAG & 4) && j == 0 && s < NE) svl[s] = ld4(r_x, row_ts(x_row(s, 1), Ts) + x_c4(s) * 16, 0);
            if (!(DSD_RS_DIAG & 8) && j == 0 && s == 12) cpg = ld4(r_c, row_ts(gch, Ts) + (tid & 7) * 16, 0);
            if (!(DSD_RS_DIAG & 8) && j == 0 && s == 13) cpf = ld4(r_c, row_ts(gch + C, Ts) + (tid & 7) * 16, 0);
            RS_PIN();
        }
        if (!(DSD_RS_DIAG & 4) && s == DSD_RS_LATE_W) {
#pragma unroll
            for (int u = 0; u < NE; ++u) stage_write(svl[u], u, 1);
            RS_PIN();
        }
        if (!(DSD_RS_DIAG & 4) && s == DSD_RS_LATE_B) {
            __syncthreads();
            if (DSD_RS_LATE_B == 11) {                           // (first version: step 12's operands behind the barrier)
#pragma unroll
                for (int j = 0; j < 4; ++j) read_b1(bn, 12, j);
            }
            RS_PIN();
        }
        RS_STAMP(0, 10 + s);
    });
    static_assert(DSD_RS_LATE_W >= 3 && DSD_RS_LATE_W <= DSD_RS_LATE_B && DSD_RS_LATE_B <= 11, "late chunks: read from step 12 on");
    RS_STAMP(0, 3);

    // ---------------- the two K halves' sums; accumulators -> LDS tile (rows [0, 32): gate, [32, 64): filter) ----------------
    // each half writes its own accumulators into its own tile; ONE barrier; the gate's threads add the two tiles (first version:
    // half 1 writes, barrier, half 0 adds and transposes, barrier - 15.29 against 15.02 ms per loop)
    {
        float* tk = kh == 0 ? et : red;
        const int trow = (w & 1) * 32 + (w >> 1) * 16;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
#pragma unroll
            for (int r = 0; r < 4; ++r) tk[(trow + rq + r) * ES + n * 16 + lcol] = acc[n][r];
        }
    }
    __syncthreads();
    if (tid < 256) {
        const int cw = tid >> 3, c4 = tid & 7;
        const f32x4 g = *reinterpret_cast<const f32x4*>(&et[cw * ES + c4 * 4]) + *reinterpret_cast<const f32x4*>(&red[cw * ES + c4 * 4]);
        const f32x4 f = *reinterpret_cast<const f32x4*>(&et[(32 + cw) * ES + c4 * 4]) +
                        *reinterpret_cast<const f32x4*>(&red[(32 + cw) * ES + c4 * 4]);
        f32x4 z;
#pragma unroll
        for (int e = 0; e < 4; ++e) z[e] = sigmoid_fast(g[e] + cpg[e]) * tanh_fast(f[e] + cpf[e]);      // wavenet.py:41-42
        const dsd_i32x4 w_z = dsd_rsrc_words(p.z + (long)bu * p.x_bstride + t0u);
        st4z(z, w_z, row_ts(gch, Ts) + c4 * 16, 0);
    }
    RS_STAMP(0, 4);
    RS_STAMP(0, 5);
}

// ---------------------------------------------------------------------------------------------------------------
// Conv, second wave layout (DSD_RS_CONV_Q = 1): 8 waves = four K QUARTERS (one 64-channel chunk each) x two row waves of 32 rows -
// the gate AND the filter rows of 16 channels.  What the walk of the first layout loses over its MFMAs is the issue of its
// B-fragment LDS reads (timing-only builds at the headline: no reads in the walk -0.54 us per layer, no weight loads -0.09,
// no late chunks -0.20, no conditioner loads -0.13); here a fragment read feeds FOUR MFMAs instead of two, a wave walks 12 steps
// of 16 MFMAs instead of 24 of 8, and every wave streams two row blocks' weights (the loads are free).  Price: four partial
// sums per output instead of two meet in LDS at the end.  Step order inside a quarter: [32-channel half][tap][k16 of the half],
// so the first six steps read only the first 32 channels of each chunk - those 128 rows are staged before the walk, the
// others are fetched behind the first two steps' MFMAs, written after step 3, barrier after step 4.  The ring holds three
// steps but only step 0's two blocks are in the prologue burst; step 0 issues steps 1 and 2.
// ---------------------------------------------------------------------------------------------------------------
// NCB: 16-frame column blocks of the tile (2: 32 frames; 3: 48 frames - one round of workgroups for T in (1024, 1536] at B = 1);
// SW: LDS row stride of the x tile (= 16 or 48 mod 64: the four k rows of a B fragment in different banks), HL: halo
template <int NCB, int SW, int HL, int RAG>
__global__ __launch_bounds__(512, 2) void wn_conv_rq_kernel(const WnLayerP p) {
    constexpr int BN = 16 * NCB, ES = BN + 4, B4 = BN / 4;      // (shadow the 32-frame constants of the K-half kernels)
    static_assert(SW >= BN + 2 * HL && (SW % 64 == 16 || SW % 64 == 48), "x tile row stride");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    RS_STAMP(0, 6);
    rs_pin_args(p);
    RS_STAMP(0, 7);
    constexpr int W4 = (BN + 2 * HL) / 4;
    constexpr int NE = 128 * W4 / 512;              // float4 per thread of 128 rows: 3 - 5
    static_assert(128 * W4 % 512 == 0, "whole float4 slots per thread");
    constexpr int NS = NCH * 12;                    // weight blocks per packed row block: [chunk][tap][k16 in chunk]
    constexpr int NQ = 12;                          // steps per wave (one chunk)
#ifndef DSD_RQ_LW
#define DSD_RQ_LW 3
#endif
#ifndef DSD_RQ_LB
#define DSD_RQ_LB 4
#endif
#ifndef DSD_RQ_CP
#define DSD_RQ_CP 7
#endif
    constexpr int LW = DSD_RQ_LW, LB = DSD_RQ_LB;   // late rows: written after step LW, barrier after step LB, read from step 6 on
    constexpr int CPS = DSD_RQ_CP;                  // the conditioner projection's two loads: steps CPS and CPS + 1
    static_assert(LW >= 2 && LW <= LB && LB <= 4 && CPS >= 2 && CPS <= 10, "step 5 fetches step 6's operands");
    static_assert(NE <= 6 && DSD_RQ_LW >= 2, "the late rows are fetched two per step during steps 0 .. 2");
    float* xs = lds;                                 // [C][SW]
    float* et = lds + C * SW;                        // [4 quarters][64][ES]: FiLM vector first, the quarters' accumulators last

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kq = wave >> 1, wr = wave & 1;
    const int lrow = lane >> 4, lcol = lane & 15, rq = lrow * 4;
    const int work = xcd_work();
    const int rest0 = work / MT, mtile = work - rest0 * MT;
    const int rest = RAG ? p.cgmap[rest0] : rest0 + p.tile0;
    const int b = fdiv_floor(rest, p.inv_tiles_per_b);
    const int t0 = (rest - b * p.tiles_per_b) * BN;
    const int Tb = (RAG && p.lens) ? p.lens[b] : p.T;
    const int Ts = p.Ts;
    const int bu = __builtin_amdgcn_readfirstlane(b), t0u = __builtin_amdgcn_readfirstlane(t0);
    RS_STAMP(0, 0);

    // ---------------- prologue: the first 32 channels of every chunk, FiLM vector, step 0's weights ----------------
    const __amdgpu_buffer_rsrc_t r_x = rsrc(p.xin + (long)bu * p.x_bstride + (t0u - HL));
    const __amdgpu_buffer_rsrc_t r_f = rsrc(p.film + p.film_col0 + bu * p.film_colb);
    float fmine = 0.f;                                           // 256 values: the first four waves fetch them (wave-uniform branch)
    if (wave < 4) fmine = ld1(r_f, tid * p.film_cstride * 4, 0);
    // staging slot u of a thread: float4 (row, c4) of a 128-row set; late = 0: channels [0, 32) of each chunk, 1: [32, 64)
    auto x_row = [&](int u, int late) {
        const int e = tid + 512 * u;
        const int re = e / W4;
        return (re >> 5) * 64 + (re & 31) + 32 * late;
    };
    auto x_c4 = [&](int u) {
        const int e = tid + 512 * u;
        return e - (e / W4) * W4;
    };
    f32x4 sv[NE], svl[NE];
#pragma unroll
    for (int u = 0; u < NE; ++u) sv[u] = ld4(r_x, row_ts(x_row(u, 0), Ts) + x_c4(u) * 16, 0);
    RS_PIN();
    // this wave's two row blocks: packed blocks 4 mtile + 2 wr (gate rows) and + 1 (filter rows) of the same 16 channels;
    // local step t of quarter kq = block 12 kq + tap * 4 + k16 of either
    const __amdgpu_buffer_rsrc_t r_w = rsrc(p.Aconv + ((long)(4 * mtile + 2 * wr) * NS + 12 * kq) * 256);
    const int wl = lane * 16;
    auto blk = [](int t) { return ((t % 6) / 2) * 4 + (t / 6) * 2 + (t % 2); };
    f32x4 W[3][2];
    auto w_load = [&](int t, int rb) {
        const int g = blk(t);
        W[t % 3][rb] = ld4(r_w, wl + (g & 3) * 1024, (g >> 2) * 4096 + rb * NS * 1024);
    };
    w_load(0, 0);
    w_load(0, 1);
    RS_PIN();
    // the hoisted conditioner projection (+ biases) of this tile's 32 channels, row-major float4 for the gate below:
    // thread (of the first 256) -> channel tid >> 3, frames 4 * (tid & 7)
    // (threads beyond the tile's 32 x B4 float4: a copy of the last channel's, unused)
    const int gcw = min(div_b4<B4>(tid), 31), gc4 = tid - div_b4<B4>(tid) * B4;
    const int gch = 32 * mtile + gcw;
    const __amdgpu_buffer_rsrc_t r_c = rsrc(p.cp + (long)bu * p.cp_bstride + t0u);
    f32x4 cpg = f32x4{0.f, 0.f, 0.f, 0.f}, cpf = cpg;
    RS_STAMP(0, 1);
    if (wave < 4) et[tid] = fmine;
    __syncthreads();
    float fa0[NE], fa1[NE];
#pragma unroll
    for (int u = 0; u < NE; ++u) {
        fa0[u] = et[x_row(u, 0)];
        fa1[u] = et[x_row(u, 1)];
    }
    auto stage_write = [&](const f32x4& v, int u, int late) {    // FiLM add, then the zero padding (wavenet.py:36-38), then LDS
        const int row = x_row(u, late), c4 = x_c4(u);
        const float fa = late ? fa1[u] : fa0[u];
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int t = t0 - HL + c4 * 4 + e;
            o[e] = (t >= 0 && t < Tb) ? v[e] + fa : 0.f;
        }
        *reinterpret_cast<f32x4*>(&xs[row * SW + c4 * 4]) = o;
    };
    // (the FiLM values are in registers: the barrier below also orders these reads before the tail's writes to `et`)
#pragma unroll
    for (int u = 0; u < NE; ++u) stage_write(sv[u], u, 0);
    __syncthreads();
    RS_STAMP(0, 2);

    // ---------------- K walk ----------------
    f32x4 acc[2][NCB];
#pragma unroll
    for (int rb = 0; rb < 2; ++rb)
#pragma unroll
        for (int n = 0; n < NCB; ++n) acc[rb][n] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* bt0 = xs + (kq * 64 + lrow) * SW + HL + lcol - p.dil;
    const float* bt1 = bt0 + p.dil;
    const float* bt2 = bt1 + p.dil;
    float bq[2][4][NCB];
    auto read_b1 = [&](float (&bv)[4][NCB], int t, int j) {      // every column block of k4 step j of local step t
        const int tap = (t % 6) / 2, k16 = (t / 6) * 2 + (t % 2);
        const float* base = (tap == 0 ? bt0 : (tap == 1 ? bt1 : bt2)) + (k16 * 16 + j * 4) * SW;
#pragma unroll
        for (int n = 0; n < NCB; ++n) bv[j][n] = base[16 * n];
    };
#pragma unroll
    for (int j = 0; j < 4; ++j) read_b1(bq[0], 0, j);
    RS_PIN();
    rs_static_for<0, NQ>([&](auto tc) __attribute__((always_inline)) {
        constexpr int t = decltype(tc)::value;
        const f32x4 wv0 = W[t % 3][0], wv1 = W[t % 3][1];
        float (&bc)[4][NCB] = bq[t & 1];
        float (&bn)[4][NCB] = bq[(t + 1) & 1];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[0][0] = rs_mfma(wv0[j], bc[j][0], acc[0][0]);
            if (t == 0) w_load(1 + j / 2, j & 1);                // step 0 issues steps 1 and 2 ...
            else if (j < 2 && t + 2 < NQ) w_load(t + 2, j);      // ... step t >= 1 issues step t + 2
            if (j == 0 && t + 1 < NQ) {                          // the next step's LDS reads in one burst
#pragma unroll
                for (int jj = 0; jj < 4; ++jj) read_b1(bn, t + 1, jj);
            }
            RS_PIN();
            acc[0][1] = rs_mfma(wv0[j], bc[j][1], acc[0][1]);
            if (j < 2 && t < 3 && 2 * t + j < NE) svl[2 * t + j] = ld4(r_x, row_ts(x_row(2 * t + j, 1), Ts) + x_c4(2 * t + j) * 16, 0);
            if (j == 0 && t == CPS) cpg = ld4(r_c, row_ts(gch, Ts) + gc4 * 16, 0);
            if (j == 0 && t == CPS + 1) cpf = ld4(r_c, row_ts(gch + C, Ts) + gc4 * 16, 0);
            RS_PIN();
#pragma unroll
            for (int n = 2; n < NCB; ++n) {
                acc[0][n] = rs_mfma(wv0[j], bc[j][n], acc[0][n]);
                RS_PIN();
            }
#pragma unroll
            for (int n = 0; n < NCB; ++n) {
                acc[1][n] = rs_mfma(wv1[j], bc[j][n], acc[1][n]);
                RS_PIN();
            }
        }
        if (t == LW) {
#pragma unroll
            for (int u = 0; u < NE; ++u) stage_write(svl[u], u, 1);
            RS_PIN();
        }
        if (t == LB) {
            __syncthreads();
            RS_PIN();
        }
    });
    RS_STAMP(0, 3);

    // ---------------- the four quarters' sums: every wave transposes its accumulators into its quarter's tile (rows [0, 32):
    // gate, [32, 64): filter), ONE barrier, the gate's threads add the four tiles ----------------
    {
        float* tk = et + kq * (64 * ES);
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int n = 0; n < NCB; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) tk[(rb * 32 + wr * 16 + rq + r) * ES + n * 16 + lcol] = acc[rb][n][r];
    }
    __syncthreads();
    if (tid < 32 * B4) {
        const int cw = gcw, c4 = gc4;
        f32x4 g = *reinterpret_cast<const f32x4*>(&et[cw * ES + c4 * 4]);
        f32x4 f = *reinterpret_cast<const f32x4*>(&et[(32 + cw) * ES + c4 * 4]);
#pragma unroll
        for (int q = 1; q < 4; ++q) {
            g += *reinterpret_cast<const f32x4*>(&et[q * (64 * ES) + cw * ES + c4 * 4]);
            f += *reinterpret_cast<const f32x4*>(&et[q * (64 * ES) + (32 + cw) * ES + c4 * 4]);
        }
        f32x4 z;
#pragma unroll
        for (int e = 0; e < 4; ++e) z[e] = sigmoid_fast(g[e] + cpg[e]) * tanh_fast(f[e] + cpf[e]);      // wavenet.py:41-42
        const dsd_i32x4 w_z = dsd_rsrc_words(p.z + (long)bu * p.x_bstride + t0u);
        st4z(z, w_z, row_ts(gch, Ts) + c4 * 16, 0);
    }
    RS_STAMP(0, 4);
    RS_STAMP(0, 5);
}

template <int NCB, int RAG>
__global__ __launch_bounds__(512, 2) void wn_out_rs_kernel(const WnLayerP p) {
    constexpr int BN = 16 * NCB, ES = BN + 4, B4 = BN / 4;      // (shadow the 32-frame constants)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    rs_pin_args(p);
    constexpr int SZ = 48;
    constexpr int NZ = C * B4 / 512;                // staged float4 per thread: 4 / 6
    static_assert(SZ >= BN && NCB <= 3, "z tile row stride");
    constexpr int NS = NCH * 4;
    constexpr int NH = NS / 2;
    float* zs = lds;                                 // [C][SZ]
    float* et = lds + C * SZ;                        // [64][ES]
    float* red = et + 64 * ES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kh = wave >> 2, w = wave & 3;
    const int lrow = lane >> 4, lcol = lane & 15, rq = lrow * 4;
    const int work = xcd_work();
    const int rest0 = work / MT, mtile = work - rest0 * MT;
    const int rest = RAG ? p.cgmap[rest0] : rest0 + p.tile0;
    const int b = fdiv_floor(rest, p.inv_tiles_per_b);
    const int t0 = (rest - b * p.tiles_per_b) * BN;
    const int Ts = p.Ts;
    const int bu = __builtin_amdgcn_readfirstlane(b), t0u = __builtin_amdgcn_readfirstlane(t0);
    RS_STAMP(1, 0);

    // ---------------- prologue: z tile (all C channels), the first weight blocks, bias ----------------
    const __amdgpu_buffer_rsrc_t r_z = rsrc(p.z + (long)bu * p.x_bstride + t0u);
    // staging slot u of a thread = float4 idx % B4 of row idx / B4, idx = tid + 512 u
    f32x4 sv[NZ];
    auto z_load = [&](int u) {
        const int idx = tid + 512 * u, row = div_b4<B4>(idx);
        sv[u] = ld4(r_z, row_ts(row, Ts) + (idx - row * B4) * 16, 0);
    };
    auto z_write = [&](int u) {
        const int idx = tid + 512 * u, row = div_b4<B4>(idx);
        *reinterpret_cast<f32x4*>(&zs[row * SZ + (idx - row * B4) * 4]) = sv[u];
    };
#pragma unroll
    for (int u = 0; u < NZ; ++u) z_load(u);
    RS_PIN();                                                    // (issue order = return order: what the walk needs first, first)
    const int orow = 64 * mtile + 16 * w;                        // this wave's 16 output rows (of 2C)
    const __amdgpu_buffer_rsrc_t r_w = rsrc(p.Aout + ((long)(4 * mtile + w) * NS + NH * kh) * 256);
    const int wl = lane * 16;
    f32x4 W[DEPTH_OUT];
#ifndef DSD_RS_BIAS_KH0
#define DSD_RS_BIAS_KH0 1
#endif
    f32x4 bo = f32x4{0.f, 0.f, 0.f, 0.f};                        // the bias rides in the first K half: only its waves fetch it
    if (!DSD_RS_BIAS_KH0 || kh == 0) bo = ld4(rsrc(p.bias_out + orow), rq * 4, 0);
#pragma unroll
    for (int s = 0; s < DEPTH_OUT - 1 - DSD_RS_OUT_RAMP; ++s) W[s] = ld4(r_w, wl + (s & 3) * 1024, (s >> 2) * 4096);
    RS_PIN();
    // residual stream (row tiles of the first C rows) or running skip sum (the other half), row-major float4:
    // thread (of the first 256) -> rows (tid >> 3) and 32 + (tid >> 3) of the tile, frames 4 * (tid & 7)
    const bool is_res = mtile < NCH;                             // workgroup-uniform
    const long eoff = (long)bu * p.x_bstride + (long)(is_res ? 64 * mtile : 64 * mtile - C) * Ts + t0u;
    const unsigned long long xa = (unsigned long long)p.xin, sa = (unsigned long long)p.skip, xo = (unsigned long long)p.xout;
    const __amdgpu_buffer_rsrc_t r_e = rsrc((const float*)(is_res ? xa : sa) + eoff);
    // (threads beyond the tile's 32 x B4 float4 per half: a copy of the last row's, unused)
    // 32-frame tiles: 64 rows x 8 float4 = one item per thread, all eight waves; 48-frame tiles: rows r and r + 32 per thread
#ifndef DSD_RS_OUT_ITEMS1
#define DSD_RS_OUT_ITEMS1 1
#endif
    constexpr int ITEMS = (DSD_RS_OUT_ITEMS1 && NCB == 2) ? 1 : 2, RP = 64 / ITEMS;
    const int erow = min(div_b4<B4>(tid), RP - 1), ec4 = tid - div_b4<B4>(tid) * B4;
    const int ev0 = row_ts(erow, Ts) + ec4 * 16;
    f32x4 pre[2];
    RS_STAMP(1, 1);
#pragma unroll
    for (int u = 0; u < NZ; ++u) z_write(u);
    __syncthreads();
    RS_STAMP(1, 2);

    // ---------------- K walk ----------------
    f32x4 acc[NCB];
#pragma unroll
    for (int n = 0; n < NCB; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[n][r] = DSD_RS_BIAS_KH0 ? bo[r] : (kh == 0 ? bo[r] : 0.f);
    const float* zt = zs + (kh * 128 + lrow) * SZ + lcol;
    float bq[2][4][NCB];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int n = 0; n < NCB; ++n) bq[0][j][n] = zt[(j * 4) * SZ + 16 * n];
    RS_PIN();
#pragma unroll
    for (int s = 0; s < NH; ++s) {
        const f32x4 wv = W[s % DEPTH_OUT];
        float (&bc)[4][NCB] = bq[s & 1];
        float (&bn)[4][NCB] = bq[(s + 1) & 1];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            acc[0] = rs_mfma(wv[j], bc[j][0], acc[0]);
            if (DSD_RS_OUT_RAMP && s == 0 && j == 0)             // (the prologue's burst holds one block less: step 0 issues two)
                W[DEPTH_OUT - 2] = ld4(r_w, wl + ((DEPTH_OUT - 2) & 3) * 1024, ((DEPTH_OUT - 2) >> 2) * 4096);
            if (j == (DSD_RS_OUT_RAMP && s == 0 ? 1 : 0) && s + DEPTH_OUT - 1 < NH)
                W[(s + DEPTH_OUT - 1) % DEPTH_OUT] = ld4(r_w, wl + ((s + DEPTH_OUT - 1) & 3) * 1024, ((s + DEPTH_OUT - 1) >> 2) * 4096);
            if (j == 0 && s + 1 < NH) {          // the next step's 4 LDS read pairs in one burst: spread one per MFMA pair
#pragma unroll                                   // they cost the walk 12 % more (measured 5.0 k vs 4.45 k cycles)
                for (int jj = 0; jj < 4; ++jj)
#pragma unroll
                    for (int n = 0; n < NCB; ++n) bn[jj][n] = zt[((s + 1) * 16 + jj * 4) * SZ + 16 * n];
            }
            RS_PIN();
            acc[1] = rs_mfma(wv[j], bc[j][1], acc[1]);
            if (j == 0 && s == 1) pre[0] = ld4(r_e, ev0, 0);
            if (ITEMS == 2 && j == 0 && s == 2) pre[1] = ld4(r_e, ev0, 32 * Ts * 4);
            RS_PIN();
#pragma unroll
            for (int n = 2; n < NCB; ++n) {
                acc[n] = rs_mfma(wv[j], bc[j][n], acc[n]);
                RS_PIN();
            }
        }
    }
    RS_STAMP(1, 3);

    // ---------------- the two K halves' sums; residual / skip (wavenet.py:45-48), row-major ----------------
    {
        float* tk = kh == 0 ? et : red;
#pragma unroll
        for (int n = 0; n < NCB; ++n) {
#pragma unroll
            for (int r = 0; r < 4; ++r) tk[(16 * w + rq + r) * ES + n * 16 + lcol] = acc[n][r];
        }
    }
    __syncthreads();
    if (tid < RP * B4) {
        const dsd_i32x4 w_o = dsd_rsrc_words((const float*)(is_res ? xo : sa) + eoff);
        const float scale = is_res ? 0.70710678118654752440f : 1.f;     // (x + o) / sqrt(2): times the fp32 reciprocal
        const bool add_pre = is_res || !p.first_layer;                  // the first layer's skip sum is its own output
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(&et[(erow + 32 * k) * ES + ec4 * 4]) +
                             *reinterpret_cast<const f32x4*>(&red[(erow + 32 * k) * ES + ec4 * 4]);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = ((add_pre ? pre[k][e] : 0.f) + a4[e]) * scale;
            st4(o, w_o, ev0, k * 32 * Ts * 4);
        }
    }
    RS_STAMP(1, 4);
    RS_STAMP(1, 5);
}
// (The out-proj in the same K-quarter layout - 4 steps of 16 MFMAs per wave, four partial tiles - was built and measured: 14.76
// against 14.75 ms per loop; its walk is too short for the halved LDS reads to pay for the wider reduction.  Not kept.)
#undef RS_PIN

template <typename K>
static hipError_t rs_attr(K kern) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

// x tile + the K parts' transpose tiles
int wn_rs_conv_lds_bytes(int sw, int bn, bool quarters) { return (256 * sw + (quarters ? 4 : 2) * 64 * (bn + 4)) * 4; }
int wn_rs_out_lds_bytes(int bn) { return (256 * 48 + 2 * 64 * (bn + 4)) * 4; }

// 48-frame tiles (NCB = 3) exist in the K-quarter layout only, on dense batches: they are chosen where they make the launch ONE
// round of workgroups (api.hip)
bool wn_rowsplit_supported(int C, int dil, long Ts) { return C == 256 && dil >= 1 && dil <= 16 && Ts < (1L << 22); }

template <int SW, int RAG>
static hipError_t rs_launch_conv(const WnLayerP& p, int nwg, int bn, hipStream_t st) {
    constexpr int HL = SW == 48 ? 8 : 16;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = rs_attr(wn_conv_rq_kernel<2, SW, HL, RAG>);
        if (e == hipSuccess) e = rs_attr(wn_conv_rs_kernel<SW, RAG>);
        if (e == hipSuccess && !RAG) e = rs_attr(wn_conv_rq_kernel<3, 80, HL, 0>);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    if (nwg == 0) return hipSuccess;
    if (bn == 48) {
        if (RAG) return hipErrorInvalidValue;
        return launch_timed(wn_conv_rq_kernel<3, 80, HL, 0>, dim3(nwg), dim3(512), wn_rs_conv_lds_bytes(80, 48, true), st, p,
                            "wn_conv_rq_kernel<3, 80, %d, 0>", HL);
    }
    // The K-quarter layout needs 86 KiB of LDS (SW 48), one workgroup per CU; the K-half layout 67 KiB, two.  DSD_RS_CONV_Q=0/1
    // forces the choice (A/B, tests).
    const int q_ev = path_opts().rs_conv_q;                      // (tests/test_gpu_rowsplit.py switches it between handles)
    const bool quarters = q_ev >= 0 ? q_ev != 0 : nwg <= 256;
    const int ldsb = wn_rs_conv_lds_bytes(SW, 32, quarters);
    if (quarters)
        return launch_timed(wn_conv_rq_kernel<2, SW, HL, RAG>, dim3(nwg), dim3(512), ldsb, st, p, "wn_conv_rq_kernel<2, %d, %d, %d>",
                            SW, HL, RAG);
    return launch_timed(wn_conv_rs_kernel<SW, RAG>, dim3(nwg), dim3(512), ldsb, st, p, "wn_conv_rs_kernel<%d, %d>", SW, RAG);
}

template <int RAG>
static hipError_t rs_launch_out(const WnLayerP& p, int nwg, int bn, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = rs_attr(wn_out_rs_kernel<2, RAG>);
        if (e == hipSuccess && !RAG) e = rs_attr(wn_out_rs_kernel<3, 0>);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    if (nwg == 0) return hipSuccess;
    if (bn == 48) {
        if (RAG) return hipErrorInvalidValue;
        return launch_timed(wn_out_rs_kernel<3, 0>, dim3(nwg), dim3(512), wn_rs_out_lds_bytes(48), st, p, "wn_out_rs_kernel<3, 0>");
    }
    return launch_timed(wn_out_rs_kernel<2, RAG>, dim3(nwg), dim3(512), wn_rs_out_lds_bytes(32), st, p, "wn_out_rs_kernel<2, %d>", RAG);
}

// which = 0: conv + FiLM + gate (p.xin -> p.z);  which = 1: out-proj + residual / skip (p.z, p.xin -> p.xout, p.skip);
// bn = frames per tile (32, or 48 on dense batches): p.tiles_per_b counts tiles of that width
hipError_t launch_wn_rowsplit(const WnLayerP& p, int which, int C_, int batch, int bn, hipStream_t st) {
    if (C_ != 256 || (bn != 32 && bn != 48)) return hipErrorInvalidValue;
    // every store of a tile stays inside its row: the last tile of an item reaches column tiles_per_b * bn (48-frame tiles: up to
    // 47 past T), and rows are Ts = padded_ts(T) floats apart - holds for every T with the present padded_ts, checked here so that
    // a change of the padding rule cannot make x / skip / z spill into the next row
    if ((long)p.tiles_per_b * bn > p.Ts) return hipErrorInvalidValue;
    const int nt = p.cgmap ? p.ncg : (p.ntiles > 0 ? p.ntiles : batch * p.tiles_per_b);
    const int nwg = nt * 8;
    if (which == 1) return p.cgmap ? rs_launch_out<1>(p, nwg, bn, st) : rs_launch_out<0>(p, nwg, bn, st);
    if (p.dil <= 8) return p.cgmap ? rs_launch_conv<48, 1>(p, nwg, bn, st) : rs_launch_conv<48, 0>(p, nwg, bn, st);
    return p.cgmap ? rs_launch_conv<80, 1>(p, nwg, bn, st) : rs_launch_conv<80, 0>(p, nwg, bn, st);
}

hipError_t wn_rowsplit_init_all() {
    WnLayerP p{};
    hipError_t e;
    for (int dil : {1, 16})
        for (int rag = 0; rag < 2; ++rag) {
            p.dil = dil;
            p.cgmap = rag ? reinterpret_cast<const int*>(&p) : nullptr;      // (no launch: the grid is empty)
            p.ncg = 0;
            p.tiles_per_b = 0;
            if ((e = launch_wn_rowsplit(p, 0, 256, 0, 32, nullptr)) != hipSuccess) return e;
            if ((e = launch_wn_rowsplit(p, 1, 256, 0, 32, nullptr)) != hipSuccess) return e;
        }
    return hipSuccess;
}

}  // namespace dsd
