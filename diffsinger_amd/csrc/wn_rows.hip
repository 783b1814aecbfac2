// WaveNet residual layer as two launches with WIDE row tiles: 128 or 256 of a GEMM's 2C = 512 rows per workgroup (4 or 2
// workgroups per 32-frame tile), for grids between the one-utterance case of wn_rowsplit.hip (64 rows, 8 workgroups per
// tile: fills 256 CUs from 32 tiles) and the full-row tiles of wn_layer.hip (one workgroup per tile, from ~one tile per CU):
// two utterances of ~1000 frames (64 tiles), four (125 tiles), or the remainder segment of a mixed layer plan (api.hip).
//
//   wn_conv_rw_kernel<MP> : z = sigmoid(y[:C]) * tanh(y[C:]),  y = dilated_conv(x + d) + cond_proj    (wavenet.py:34-42)
//   wn_out_rw_kernel<MP>  : o = output_projection(z);  x' = (x + o[:C]) / sqrt(2);  skip += o[C:]      (wavenet.py:44-48, :96)
//
// Why not just more 64-row workgroups: in wn_rowsplit.hip's walk a B fragment read from LDS feeds 2 (K halves) or 4 (K
// quarters) MFMAs, and the issue of those LDS reads - not the MFMAs, not the weight stream - is what bounds it (DESIGN 4.2
// (8): 0.55 of the fp32-MFMA peak), and every one of the 8 workgroups of a tile stages the same x tile behind its own cold
// prologue.  Here a wave owns MP (gate, filter) block pairs = 2 MP row blocks x both column blocks: a fragment read feeds
// 4 MP MFMAs (8 / 16), a tile is staged 4 / 2 times, and the walk's steps are long enough (32 / 64 MFMAs) for every load to
// hide.  Same wave layouts as the 64-row kernels: conv = four K quarters x two row waves, out-proj = two K halves x four
// row waves, partial sums met in LDS - over the dead x tile here, which is what lets 256 rows fit.
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
// byte offset of a row as a 24-bit multiply (see wn_rowsplit.hip: a 32-bit mad's undefined high half)
__device__ __forceinline__ int row_ts(int row, int Ts) { return (int)__umul24((unsigned)row, (unsigned)(Ts * 4)); }

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
__device__ __forceinline__ void st4(f32x4 v, dsd_i32x4 r, int voff, int soff) {       // x / skip: write-through
    dsd_store_b128<DSD_ST_AUX>(__builtin_bit_cast(dsd_u32x4, v), r, voff, soff);
}
__device__ __forceinline__ void st4z(f32x4 v, dsd_i32x4 r, int voff, int soff) {      // z: read back by the next launch
    dsd_store_b128<0>(__builtin_bit_cast(dsd_u32x4, v), r, voff, soff);
}
__device__ __forceinline__ void rw_pin_args(const WnLayerP& p) {     // every argument in SGPRs behind ONE batch of scalar loads
    asm volatile("" ::"s"(p.Aconv), "s"(p.Aout), "s"(p.bias_out), "s"(p.xin), "s"(p.xout), "s"(p.skip), "s"(p.z), "s"(p.x_bstride),
                 "s"(p.Ts), "s"(p.cp), "s"(p.cp_bstride), "s"(p.film), "s"(p.film_cstride), "s"(p.film_col0), "s"(p.film_colb),
                 "s"(p.dil), "s"(p.T), "s"(p.tiles_per_b), "s"(p.first_layer), "s"(p.inv_tiles_per_b), "s"(p.tile0),
                 "s"((int)gridDim.x));
}
template <int I, int N, typename F>
__device__ __forceinline__ void rw_static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        rw_static_for<I + 1, N>(f);
    }
}
// XCD-aware bijective remap (speed only): an XCD takes a contiguous range of work items, row tile fastest, so the row tiles
// of a frame tile - which stage the same activations - share an L2
__device__ __forceinline__ int xcd_work() {
    const int nwg = gridDim.x;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int q8 = nwg >> 3, r8 = nwg & 7;
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
}
#define RW_PIN() __builtin_amdgcn_sched_barrier(0)
__device__ __forceinline__ f32x4 rw_mfma(float wfrag, float xfrag, f32x4 acc) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(wfrag, xfrag, acc, 0, 0, 0);
}

#ifndef DSD_RW_ES
#define DSD_RW_ES 36
#endif
constexpr int NCH = 4, C = 256, BN = 32, ES = DSD_RW_ES, B4 = 8;
constexpr int cmax(int a, int b) { return a > b ? a : b; }

}  // namespace

// ---------------------------------------------------------------------------------------------------------------
// Conv + FiLM + gate.  512 threads = 8 waves = four K quarters (kq: one 64-channel chunk each) x two row waves (wr); a
// workgroup owns the gate AND filter rows of 32 MP channels (packed blocks [4 MP mtile, +4 MP): even = gate, odd = filter of
// 16 channels), wave wr the 2 MP blocks of its 16 MP channels.  Step order inside a quarter: [32-channel half][tap][k16 of
// the half], so the first six steps read only the first 32 channels of each chunk: those 128 rows are staged before the
// walk, the other 128 are fetched behind the first steps' MFMAs, written after step LW, barrier after step LB.
// The weight ring holds DW step sets of 2 MP blocks (3; 2 at MP = 4, whose 64-MFMA steps cover a set's latency alone and
// whose 16 accumulators leave no room for a third); only step 0's set is in the prologue's burst of loads.
// ---------------------------------------------------------------------------------------------------------------
template <int MP, int SW, int HL, int RAG>
__global__ __launch_bounds__(512, 1) void wn_conv_rw_kernel(const WnLayerP p) {
    static_assert(MP == 2 || MP == 4, "128 or 256 rows per workgroup");
    static_assert(SW >= BN + 2 * HL && (SW % 64 == 16 || SW % 64 == 48), "x tile row stride");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    rw_pin_args(p);
    constexpr int MT = 8 / MP;                      // row tiles (workgroups) per frame tile
    constexpr int NB2 = 2 * MP;                     // row blocks per wave
    constexpr int W4 = (BN + 2 * HL) / 4;
    constexpr int NE = 128 * W4 / 512;              // float4 per thread of 128 rows: 3 / 4
    static_assert(128 * W4 % 512 == 0 && NE <= 6, "whole float4 slots per thread; late rows: two per step during steps 0 .. 2");
    constexpr int NS = NCH * 12;                    // weight blocks per packed row block: [chunk][tap][k16 in chunk]
    constexpr int NQ = 12;                          // steps per wave (one chunk)
#ifndef DSD_RW_DW4
#define DSD_RW_DW4 2
#endif
    constexpr int DW = MP == 4 ? DSD_RW_DW4 : 3;    // weight ring depth (step sets)
    constexpr int LW = 3, LB = 4;                   // late rows: written after step LW, barrier after step LB, read from step 6 on
    constexpr int CPS = 7;                          // the conditioner projection's loads: steps CPS (gate rows), CPS + 1 (filter rows)
    constexpr int NG = MP / 2;                      // gate items (channel, float4) per thread: 32 MP channels x 8 / 512
    constexpr int NM = 8 * NB2;                     // MFMAs per step
    constexpr int XS_FLOATS = cmax(C * SW, 4 * 64 * MP * ES);
    float* xs = lds;                                 // [C][SW]; after the walk: the quarters' accumulator tiles [4][64 MP][ES]
    float* fl = lds + XS_FLOATS;                     // FiLM vector [C]

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
    RW_PIN();
    // this wave's row blocks: packed blocks [4 MP mtile + NB2 wr, + NB2); local step t of quarter kq = block 12 kq + blk(t)
    const __amdgpu_buffer_rsrc_t r_w = rsrc(p.Aconv + ((long)(4 * MP * mtile + NB2 * wr) * NS + 12 * kq) * 256);
    const int wl = lane * 16;
    auto blk = [](int t) { return ((t % 6) / 2) * 4 + (t / 6) * 2 + (t % 2); };
    f32x4 W[DW][NB2];
    auto w_load = [&](int t, int rb) {
        const int g = blk(t);
        W[t % DW][rb] = ld4(r_w, wl + (g & 3) * 1024, (g >> 2) * 4096 + rb * NS * 1024);
    };
#pragma unroll
    for (int rb = 0; rb < NB2; ++rb) w_load(0, rb);
    RW_PIN();
    // the hoisted conditioner projection (+ biases) of this tile's 32 MP channels, row-major float4 for the gate below:
    // item i = tid + 512 k -> channel i >> 3 of the tile, frames 4 * (i & 7)
    const int gc4 = tid & 7;
    const int gcw0 = tid >> 3;                                   // + 64 k
    const int gch0 = 32 * MP * mtile + gcw0;
    const __amdgpu_buffer_rsrc_t r_c = rsrc(p.cp + (long)bu * p.cp_bstride + t0u);
    f32x4 cpg[NG], cpf[NG];
    if (wave < 4) fl[tid] = fmine;
    __syncthreads();
    float fa0[NE], fa1[NE];
#pragma unroll
    for (int u = 0; u < NE; ++u) {
        fa0[u] = fl[x_row(u, 0)];
        fa1[u] = fl[x_row(u, 1)];
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
#pragma unroll
    for (int u = 0; u < NE; ++u) stage_write(sv[u], u, 0);
    __syncthreads();

    // ---------------- K walk ----------------
    f32x4 acc[NB2][2];
#pragma unroll
    for (int rb = 0; rb < NB2; ++rb) {
        acc[rb][0] = f32x4{0.f, 0.f, 0.f, 0.f};
        acc[rb][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    const float* bt0 = xs + (kq * 64 + lrow) * SW + HL + lcol - p.dil;
    const float* bt1 = bt0 + p.dil;
    const float* bt2 = bt1 + p.dil;
    float bq[2][4][2];
    auto read_b1 = [&](float (&bv)[4][2], int t, int j) {        // both column blocks of k4 step j of local step t
        const int tap = (t % 6) / 2, k16 = (t / 6) * 2 + (t % 2);
        const float* base = (tap == 0 ? bt0 : (tap == 1 ? bt1 : bt2)) + (k16 * 16 + j * 4) * SW;
        bv[j][0] = base[0];
        bv[j][1] = base[16];
    };
#pragma unroll
    for (int j = 0; j < 4; ++j) read_b1(bq[0], 0, j);
    RW_PIN();
    rw_static_for<0, NQ>([&](auto tc) __attribute__((always_inline)) {
        constexpr int t = decltype(tc)::value;
        float (&bc)[4][2] = bq[t & 1];
        float (&bn)[4][2] = bq[(t + 1) & 1];
        // MFMA m = (j * NB2 + rb) * 2 + n of the step, each pinned, with the step's other instructions placed behind chosen ones:
        //   weights: step 0 issues the sets of steps 1 .. DW - 1 (one load behind every 4th MFMA at DW = 3, every 8th at DW = 2);
        //            step t >= 1 the set of step t + DW - 1, one load behind every 4th MFMA of its first half
        //   LDS    : the next step's eight fragment reads in one burst behind MFMA 1
        //   late x : two float4 per step during steps 0 .. 2 (behind MFMAs 2 and 6)
        //   cond   : the gate rows' float4 in step CPS, the filter rows' in step CPS + 1 (behind MFMAs 3, 7)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int rb = 0; rb < NB2; ++rb)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const int m = (j * NB2 + rb) * 2 + n;
                    acc[rb][n] = rw_mfma(W[t % DW][rb][j], bc[j][n], acc[rb][n]);
                    if (t == 0) {
                        if (DW == 3 && m % 4 == 0 && m / 4 < 2 * NB2) w_load(1 + (m / 4) / NB2, (m / 4) % NB2);
                        if (DW == 2 && m % 8 == 0 && m / 8 < NB2) w_load(1, m / 8);
                    } else if (t + DW - 1 < NQ && m % 4 == 0 && m / 4 < NB2) {
                        w_load(t + DW - 1, m / 4);
                    }
                    if (m == 1 && t + 1 < NQ) {
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) read_b1(bn, t + 1, jj);
                    }
                    if (t < 3 && (m == 2 || m == 6)) {
                        const int u = 2 * t + (m == 6 ? 1 : 0);
                        if (u < NE) svl[u] = ld4(r_x, row_ts(x_row(u, 1), Ts) + x_c4(u) * 16, 0);
                    }
                    if (t == CPS && (m == 3 || (NG == 2 && m == 7))) {
                        const int k = m == 3 ? 0 : 1;
                        cpg[k] = ld4(r_c, row_ts(gch0 + 64 * k, Ts) + gc4 * 16, 0);
                    }
                    if (t == CPS + 1 && (m == 3 || (NG == 2 && m == 7))) {
                        const int k = m == 3 ? 0 : 1;
                        cpf[k] = ld4(r_c, row_ts(gch0 + 64 * k + C, Ts) + gc4 * 16, 0);
                    }
                    RW_PIN();
                }
        static_assert(NM >= 8, "slots");
        if (t == LW) {
#pragma unroll
            for (int u = 0; u < NE; ++u) stage_write(svl[u], u, 1);
            RW_PIN();
        }
        if (t == LB) {
            __syncthreads();
            RW_PIN();
        }
    });

    // ---------------- the four quarters' sums.  The x tile is dead once every wave has left the walk: each wave transposes its
    // accumulators into its quarter's tile OVER it (rows [0, 32 MP): gate, [32 MP, 64 MP): filter of the tile's channels), one
    // more barrier, the gate's threads add the four tiles ----------------
    __syncthreads();
    {
        float* tk = xs + kq * (64 * MP * ES);
#pragma unroll
        for (int i = 0; i < MP; ++i)
#pragma unroll
            for (int gf = 0; gf < 2; ++gf)
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        tk[(gf * 32 * MP + (wr * MP + i) * 16 + rq + r) * ES + n * 16 + lcol] = acc[2 * i + gf][n][r];
    }
    __syncthreads();
    {
        const dsd_i32x4 w_z = dsd_rsrc_words(p.z + (long)bu * p.x_bstride + t0u);
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            const int cw = gcw0 + 64 * k;
            f32x4 g = *reinterpret_cast<const f32x4*>(&xs[cw * ES + gc4 * 4]);
            f32x4 f = *reinterpret_cast<const f32x4*>(&xs[(32 * MP + cw) * ES + gc4 * 4]);
#pragma unroll
            for (int q = 1; q < 4; ++q) {
                g += *reinterpret_cast<const f32x4*>(&xs[q * (64 * MP * ES) + cw * ES + gc4 * 4]);
                f += *reinterpret_cast<const f32x4*>(&xs[q * (64 * MP * ES) + (32 * MP + cw) * ES + gc4 * 4]);
            }
            f32x4 z;
#pragma unroll
            for (int e = 0; e < 4; ++e) z[e] = sigmoid_fast(g[e] + cpg[k][e]) * tanh_fast(f[e] + cpf[k][e]);      // wavenet.py:41-42
            st4z(z, w_z, row_ts(gch0 + 64 * k, Ts) + gc4 * 16, 0);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Out-proj + residual / skip.  8 waves = two K halves (kh) x four row waves (w); a workgroup owns output rows
// [64 MP mtile, +64 MP) of the 2C (the first C: residual half, the rest: skip half), wave w the MP blocks [MP w, MP w + MP) of them.
// The whole z tile [C][32] is staged before the walk (8 steps per half, 8 MP MFMAs each).
// ---------------------------------------------------------------------------------------------------------------
template <int MP, int RAG>
__global__ __launch_bounds__(512, 1) void wn_out_rw_kernel(const WnLayerP p) {
    static_assert(MP == 2 || MP == 4, "128 or 256 rows per workgroup");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    rw_pin_args(p);
    constexpr int MT = 8 / MP;
    constexpr int SZ = 48;
    constexpr int NZ = C * B4 / 512;                // staged float4 per thread: 4
    constexpr int NS = NCH * 4, NH = NS / 2;        // k16 steps, per K half
    constexpr int DO = 3;                           // weight ring depth
    constexpr int RWS = 64 * MP;                    // rows of the workgroup
    float* zs = lds;                                 // [C][SZ]
    float* et = lds + C * SZ;                        // [RWS][ES]: K half 0's accumulators
    float* red = et + RWS * ES;                      // K half 1's

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

    // ---------------- prologue: z tile (all C channels), the first weight sets, bias ----------------
    const __amdgpu_buffer_rsrc_t r_z = rsrc(p.z + (long)bu * p.x_bstride + t0u);
    f32x4 sv[NZ];
#pragma unroll
    for (int u = 0; u < NZ; ++u) {
        const int idx = tid + 512 * u, row = idx >> 3;
        sv[u] = ld4(r_z, row_ts(row, Ts) + (idx & 7) * 16, 0);
    }
    RW_PIN();
    const int orow = RWS * mtile + 16 * MP * w;                  // this wave's first output row (of 2C)
    const __amdgpu_buffer_rsrc_t r_w = rsrc(p.Aout + ((long)(4 * MP * mtile + MP * w) * NS + NH * kh) * 256);
    const int wl = lane * 16;
    f32x4 W[DO][MP];
    auto w_load = [&](int s, int k) { W[s % DO][k] = ld4(r_w, wl + (s & 3) * 1024, (s >> 2) * 4096 + k * NS * 1024); };
    f32x4 bo[MP];
#pragma unroll
    for (int k = 0; k < MP; ++k) {
        bo[k] = f32x4{0.f, 0.f, 0.f, 0.f};                       // the bias rides in the first K half: only its waves fetch it
        // (every wave fetches it, the second K half multiplies it away: a load under `if (kh == 0)` made hipcc wait for each of
        // the MP loads with vmcnt(0) - four exposed round trips in a 9 us kernel)
        bo[k] = ld4(rsrc(p.bias_out + orow + 16 * k), rq * 4, 0);
    }
#pragma unroll
    for (int k = 0; k < MP; ++k) w_load(0, k);
    RW_PIN();
    // residual stream (row tiles of the first C rows) or running skip sum (the other half), row-major float4:
    // item i = tid + 512 k -> row i >> 3 of the workgroup's RWS rows, frames 4 * (i & 7)
    const bool is_res = RWS * mtile < C;                         // workgroup-uniform
    const long eoff = (long)bu * p.x_bstride + (long)(is_res ? RWS * mtile : RWS * mtile - C) * Ts + t0u;
    const unsigned long long xa = (unsigned long long)p.xin, sa = (unsigned long long)p.skip, xo = (unsigned long long)p.xout;
    const __amdgpu_buffer_rsrc_t r_e = rsrc((const float*)(is_res ? xa : sa) + eoff);
    const int erow = tid >> 3, ec4 = tid & 7;
    const int ev0 = row_ts(erow, Ts) + ec4 * 16;
    f32x4 pre[MP];
#pragma unroll
    for (int u = 0; u < NZ; ++u) {
        const int idx = tid + 512 * u, row = idx >> 3;
        *reinterpret_cast<f32x4*>(&zs[row * SZ + (idx & 7) * 4]) = sv[u];
    }
    __syncthreads();

    // ---------------- K walk ----------------
    f32x4 acc[MP][2];
    const float bsel = kh == 0 ? 1.f : 0.f;                      // the bias rides in the first K half
#pragma unroll
    for (int k = 0; k < MP; ++k)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            acc[k][0][r] = bo[k][r] * bsel;
            acc[k][1][r] = bo[k][r] * bsel;
        }
    const float* zt = zs + (kh * 128 + lrow) * SZ + lcol;
    float bq[2][4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        bq[0][j][0] = zt[(j * 4) * SZ];
        bq[0][j][1] = zt[(j * 4) * SZ + 16];
    }
    RW_PIN();
    rw_static_for<0, NH>([&](auto sc) __attribute__((always_inline)) {
        constexpr int s = decltype(sc)::value;
        float (&bc)[4][2] = bq[s & 1];
        float (&bn)[4][2] = bq[(s + 1) & 1];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < MP; ++k)
#pragma unroll
                for (int n = 0; n < 2; ++n) {
                    const int m = (j * MP + k) * 2 + n;          // MFMA of the step, 8 MP of them
                    acc[k][n] = rw_mfma(W[s % DO][k][j], bc[j][n], acc[k][n]);
                    if (s == 0) {                                // step 0 issues the sets of steps 1 and 2
                        if (m % 2 == 0 && m / 2 < 2 * MP) w_load(1 + (m / 2) / MP, (m / 2) % MP);
                    } else if (s + DO - 1 < NH && m % 2 == 0 && m / 2 < MP) {
                        w_load(s + DO - 1, m / 2);
                    }
                    if (m == 1 && s + 1 < NH) {
#pragma unroll
                        for (int jj = 0; jj < 4; ++jj) {
                            bn[jj][0] = zt[((s + 1) * 16 + jj * 4) * SZ];
                            bn[jj][1] = zt[((s + 1) * 16 + jj * 4) * SZ + 16];
                        }
                    }
                    if (s >= 1 && s <= MP && m == 3) pre[s - 1] = ld4(r_e, ev0, (s - 1) * 64 * Ts * 4);
                    RW_PIN();
                }
    });

    // ---------------- the two K halves' sums; residual / skip (wavenet.py:45-48), row-major ----------------
    {
        float* tk = kh == 0 ? et : red;
#pragma unroll
        for (int k = 0; k < MP; ++k)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) tk[(16 * (MP * w + k) + rq + r) * ES + n * 16 + lcol] = acc[k][n][r];
    }
    __syncthreads();
    {
        const dsd_i32x4 w_o = dsd_rsrc_words((const float*)(is_res ? xo : sa) + eoff);
        const float scale = is_res ? 0.70710678118654752440f : 1.f;     // (x + o) / sqrt(2): times the fp32 reciprocal
        const bool add_pre = is_res || !p.first_layer;                  // the first layer's skip sum is its own output
#pragma unroll
        for (int k = 0; k < MP; ++k) {
            const f32x4 a4 = *reinterpret_cast<const f32x4*>(&et[(erow + 64 * k) * ES + ec4 * 4]) +
                             *reinterpret_cast<const f32x4*>(&red[(erow + 64 * k) * ES + ec4 * 4]);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = ((add_pre ? pre[k][e] : 0.f) + a4[e]) * scale;
            st4(o, w_o, ev0, k * 64 * Ts * 4);
        }
    }
}
#undef RW_PIN

int wn_rw_conv_lds_bytes(int mp, int sw) { return (cmax(C * sw, 4 * 64 * mp * ES) + 256) * 4; }
int wn_rw_out_lds_bytes(int mp) { return (C * 48 + 2 * 64 * mp * ES) * 4; }

bool wn_rows_supported(int C_, int dil, long Ts) { return C_ == 256 && dil >= 1 && dil <= 16 && Ts < (1L << 22); }

template <typename K>
static hipError_t rw_attr(K kern) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

template <int MP, int SW, int RAG>
static hipError_t rw_launch_conv(const WnLayerP& p, int nwg, hipStream_t st) {
    constexpr int HL = SW == 48 ? 8 : 16;
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = rw_attr(wn_conv_rw_kernel<MP, SW, HL, RAG>);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    if (nwg == 0) return hipSuccess;
    return launch_timed(wn_conv_rw_kernel<MP, SW, HL, RAG>, dim3(nwg), dim3(512), wn_rw_conv_lds_bytes(MP, SW), st, p,
                        "wn_conv_rw_kernel<%d, %d, %d, %d>", MP, SW, HL, RAG);
}

template <int MP, int RAG>
static hipError_t rw_launch_out(const WnLayerP& p, int nwg, hipStream_t st) {
    static bool attr_done = false;
    if (!attr_done) {
        hipError_t e = rw_attr(wn_out_rw_kernel<MP, RAG>);
        if (e != hipSuccess) return e;
        attr_done = true;
    }
    if (nwg == 0) return hipSuccess;
    return launch_timed(wn_out_rw_kernel<MP, RAG>, dim3(nwg), dim3(512), wn_rw_out_lds_bytes(MP), st, p, "wn_out_rw_kernel<%d, %d>", MP, RAG);
}

// which = 0: conv + FiLM + gate (p.xin -> p.z);  1: out-proj + residual / skip (p.z, p.xin -> p.xout, p.skip);
// rows = 128 or 256 rows per workgroup; 32-frame tiles
hipError_t launch_wn_rows(const WnLayerP& p, int which, int C_, int batch, int rows, hipStream_t st) {
    if (C_ != 256 || (rows != 128 && rows != 256)) return hipErrorInvalidValue;
    if ((long)p.tiles_per_b * 32 > p.Ts) return hipErrorInvalidValue;          // a tile's stores stay inside its row
    const int nt = p.cgmap ? p.ncg : (p.ntiles > 0 ? p.ntiles : batch * p.tiles_per_b);
    const int nwg = nt * (512 / rows);
#define RW_CASE(MP_)                                                                                                       \
    if (rows == 64 * MP_) {                                                                                                \
        if (which == 1) return p.cgmap ? rw_launch_out<MP_, 1>(p, nwg, st) : rw_launch_out<MP_, 0>(p, nwg, st);           \
        if (p.dil <= 8) return p.cgmap ? rw_launch_conv<MP_, 48, 1>(p, nwg, st) : rw_launch_conv<MP_, 48, 0>(p, nwg, st); \
        return p.cgmap ? rw_launch_conv<MP_, 80, 1>(p, nwg, st) : rw_launch_conv<MP_, 80, 0>(p, nwg, st);                 \
    }
    RW_CASE(2)
    RW_CASE(4)
#undef RW_CASE
    return hipErrorInvalidValue;
}

// raise the dynamic-LDS limit of every instantiation once, outside any stream capture
hipError_t wn_rows_init_all() {
    WnLayerP p{};
    hipError_t e;
    for (int rows : {128, 256})
        for (int dil : {1, 16})
            for (int rag = 0; rag < 2; ++rag) {
                p.dil = dil;
                p.cgmap = rag ? reinterpret_cast<const int*>(&p) : nullptr;      // (no launch: the grid is empty)
                p.ncg = 0;
                p.tiles_per_b = 0;
                if ((e = launch_wn_rows(p, 0, 256, 0, rows, nullptr)) != hipSuccess) return e;
                if ((e = launch_wn_rows(p, 1, 256, 0, rows, nullptr)) != hipSuccess) return e;
            }
    return hipSuccess;
}

}  // namespace dsd
