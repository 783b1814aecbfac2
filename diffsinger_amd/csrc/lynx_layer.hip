// LYNXNet's two pointwise GEMMs (lynxnet.py:53-59 inside LYNXConvModule, :76-84 around it) for batched grids, in the
// style of wn_layer.hip: a workgroup owns 32 frames and 512 output rows, the WHOLE K extent (up to 1024 channels) of its
// activation tile is resident in LDS, and the K walk is one uninterrupted, fully unrolled stream - weights as 1 KiB
// fragment blocks straight from L2 into a three-set register rotation, compiler-counted waits, one load behind every 8
// MFMAs.  (gemm.hip walks K in 64-channel chunks with a barrier per chunk and 16 MFMAs per k16 step: 71 % of the fp32
// MFMA peak on these shapes; the resident walk of wn_layer.hip runs at 33 cycles per MFMA.)
//
//   lx_pw1_kernel : u = out * silu(gate),  [out; gate] = W1 LayerNorm(xin) + b1      (LN affine folded into W1 / b1;
//                   per-frame (mean, rstd) from ln_merge_kernel applied while the tile is staged)
//   lx_pw2_kernel : v = W2 u' + b2 + x  (residual), then the NEXT layer's transition exactly as gemm.hip's EP_LYNX_NEXT:
//                   strong: x = v + cpn, xin = x + d;  else: x = v, xin = v + cpn + d;  no next layer: x = xin = v;
//                   plus the LayerNorm partials (mean, sum of squared deviations) of xin per 64-row tile and frame.
//                   K = inner (2048) is walked as two resident phases of 1024 channels.
//
// LDS tile: [K][32] floats with NO padding (a 1024-channel tile is 128 KiB): the 4 k-rows x 16 columns of a B-fragment
// read hit 64 distinct banks because odd rows are stored with their two 16-column halves swapped (physical column =
// column XOR 16 * (row & 1)); a lane's row parity is its own (lane >> 4) & 1, so the swizzle is a per-lane constant.
#include <hip/hip_ext.h>

#include "dsd_internal.h"

namespace dsd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifdef DSD_STAMPS
// lx_pw1p_kernel, wave 0 of each workgroup: [0] start, [1] activation tile staged (after the barrier), [2] sum over the row
// tiles of the walk's cycles, [3] sum of the epilogues' cycles, [4] end
__device__ unsigned long long g_lx_stamps[4096][8];
extern "C" int dsd_dbg_read_lx_stamps(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_lx_stamps), sizeof(g_lx_stamps));
}
#define LX_T() ([]() { __builtin_amdgcn_sched_barrier(0); unsigned long long t_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0); return t_; }())
#endif

namespace {

__device__ __forceinline__ int fdiv_floor(int x, float inv) { return (int)(((float)x + 0.5f) * inv); }
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
    dsd_store_b128<0>(__builtin_bit_cast(dsd_u32x4, v), r, voff, soff);      // plain: the next kernel re-reads these lines from L2 (write-through measured +0.5 %)
}
#ifndef DSD_LX_RCP
#define DSD_LX_RCP 1
#endif
// SiLU's sigmoid: expf as the library computes it; the reciprocal as v_rcp_f32 (<= 1 ulp) instead of an IEEE division
// sequence (~10 VALU instructions per element, 64 elements per lane and row tile in the SwiGLU epilogue)
__device__ __forceinline__ float sigmoid_f(float v) {
    return DSD_LX_RCP ? __builtin_amdgcn_rcpf(1.f + expf(-v)) : 1.f / (1.f + expf(-v));
}

constexpr int BN = 32;          // frames per tile
constexpr int MBW = 8;          // 16-row blocks per wave: 128 rows, 512 per workgroup
constexpr int ES = BN + 4;      // row stride of the epilogue transpose tiles

// XCD-aware remap (speed only): an XCD takes a contiguous range of work items; with the row tile SLOWEST an XCD walks all
// frame tiles of (about) one row tile, whose 2 MiB of weights then stay in its L2 while the activation tiles stream
__device__ __forceinline__ int xcd_work() {
    const int nwg = gridDim.x;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int q8 = nwg >> 3, r8 = nwg & 7;
    return (xcd < r8 ? xcd * (q8 + 1) : r8 * (q8 + 1) + (xcd - r8) * q8) + slot;
}

// 64 MFMAs of a k16 step with the step's MBW weight loads (for step s + 2), one optional operand load and the 8 LDS reads
// of step s + 1 between them
#define LX_SPREAD()                                                                  \
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                               \
    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                               \
    __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                               \
    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                               \
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                               \
    _Pragma("unroll") for (int g_ = 1; g_ < MBW; ++g_) {                             \
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);                           \
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                           \
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                           \
    }                                                                                \
    __builtin_amdgcn_sched_barrier(0);

// One resident K phase of KT channels: NS = KT / 16 steps from the weight stream r_w starting at block step0.  The three
// weight sets rotate: local step s runs from W[(s + ROT) % 3]; W[ROT % 3], W[(ROT + 1) % 3] must hold steps step0,
// step0 + 1 on entry, and with MORE the last two steps fetch the next phase's first two blocks (whose ROT is
// (ROT + NS) % 3), so that two phases chain without a bubble.  extra(s) is called once per step for operand loads that
// ride along.
template <int KT, int ROT, bool MORE, typename Extra>
__device__ __forceinline__ void k_phase(f32x4 (&acc)[MBW][2], f32x4 (&W)[3][MBW], const __amdgpu_buffer_rsrc_t r_w, const int (&wk)[MBW],
                                        int step0, const float* zt0, const float* zt1, Extra extra) {
    constexpr int NS = KT / 16;
    float bq[2][4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        bq[0][j][0] = zt0[(j * 4) * BN];
        bq[0][j][1] = zt1[(j * 4) * BN];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (s + 2 < NS || MORE) {
            const int g = step0 + s + 2;
#pragma unroll
            for (int k = 0; k < MBW; ++k) W[(s + 2 + ROT) % 3][k] = ld4(r_w, wk[k] + (g & 3) * 1024, (g >> 2) * 4096);
        }
        extra(s);
        if (s + 1 < NS) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bq[(s + 1) & 1][j][0] = zt0[((s + 1) * 16 + j * 4) * BN];
                bq[(s + 1) & 1][j][1] = zt1[((s + 1) * 16 + j * 4) * BN];
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < MBW; ++k) {
                acc[k][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(W[(s + ROT) % 3][k][j], bq[s & 1][j][0], acc[k][0], 0, 0, 0);
                acc[k][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(W[(s + ROT) % 3][k][j], bq[s & 1][j][1], acc[k][1], 0, 0, 0);
            }
        LX_SPREAD()
    }
}

}  // namespace

// LayerNorm statistics of a tile's 32 frames from the producer's per-64-row partials (mean_i, M2_i): ln_merge_kernel's arithmetic
// in ln_merge_kernel's order (tiles ascending; parallel-variance formula), every thread for its own four frames c4 * 4 .. + 3,
// all 2 * KT / 64 partial vectors in flight at once (a run-time loop here serialised 48 load round trips: + 1.4 % on config 3)
template <int KT>
__device__ __forceinline__ void lx_merge_stats(const LxLayerP& p, int bu, int t0u, int c4, f32x4& mean, f32x4& rstd) {
    constexpr int NT = KT / 64;                                  // 64-row tiles of xin (= p.ln_tiles)
    const __amdgpu_buffer_rsrc_t r_p = rsrc(p.lnpart_in + (long)bu * NT * 2 * p.lnpart_ts + t0u);
    f32x4 pm[NT], pq[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        pm[i] = ld4(r_p, c4 * 16, i * 2 * p.lnpart_ts * 4);
        pq[i] = ld4(r_p, c4 * 16, (i * 2 + 1) * p.lnpart_ts * 4);
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

// ---------------------------------------------------------------------------------------------------------------
// pw1: LayerNorm -> 1x1 C -> 2 * inner -> SwiGLU.  KT = C (512 or 1024); packed rows: pairs (out row r, gate row r + inner)
// interleaved per 16-row block (PackedGemm with pairC = inner): even block = out, odd = gate of 16 channels.
// Workgroup = 256 pairs (channels [256 mtile, +256) of u) x 32 frames; wave w: channels [256 mtile + 64 w, +64).
// ---------------------------------------------------------------------------------------------------------------
template <int KT, int RAG>
__global__ __launch_bounds__(256, 1) void lx_pw1_kernel(const LxLayerP p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NS = KT / 16;
    constexpr int NU = KT * (BN / 4) / 256;         // staged float4 per thread: 32 (K = 1024), 16 (K = 512)
    float* xs = lds;                                 // [KT][32], odd rows half-swapped; later the epilogue tiles

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
    const int t0 = (rest - b * p.tiles_per_b) * BN;
    const int Ts = p.Ts;
    const int bu = __builtin_amdgcn_readfirstlane(b), t0u = __builtin_amdgcn_readfirstlane(t0);
    const int mu = __builtin_amdgcn_readfirstlane(mtile);

    // ---------------- prologue: LayerNorm statistics of the tile's frames, the activation tile, two weight steps ----------------
    const int c4 = tid & 7;                                      // the thread's frames 4 c4 .. 4 c4 + 3, for every staged row
    // (the 2 * inner / 512 row-tile workgroups of a frame tile each repeat this small merge: cheaper than a launch of its own)
    f32x4 mean, rstd;
#if DSD_LX_PW1_MERGE
    lx_merge_stats<KT>(p, bu, t0u, c4, mean, rstd);
#else
    {
        const __amdgpu_buffer_rsrc_t r_s = rsrc(p.stats + (long)bu * 2 * Ts + t0u);
        mean = ld4(r_s, c4 * 16, 0);
        rstd = ld4(r_s, c4 * 16, Ts * 4);
    }
#endif
    const __amdgpu_buffer_rsrc_t r_x = rsrc(p.xin + (long)bu * p.x_bstride + t0u);
    const __amdgpu_buffer_rsrc_t r_w = rsrc(p.A1 + (long)(MBW * 4 * mu + MBW * wave) * NS * 256);
    int wk[MBW];
#pragma unroll
    for (int k = 0; k < MBW; ++k) wk[k] = lane * 16 + k * NS * 1024;
    f32x4 W[3][MBW];
    const int xv0 = ((tid >> 3) * Ts + c4 * 4) * 4;              // row tid >> 3 (+ 32 per slot)
    constexpr int NB4 = 8;                                       // slots per staging batch (bounds the registers in flight)
#pragma unroll
    for (int u0 = 0; u0 < NU; u0 += NB4) {
        f32x4 sv[NB4];
#pragma unroll
        for (int u = 0; u < NB4; ++u) sv[u] = ld4(r_x, xv0, (u0 + u) * 32 * Ts * 4);
        if (u0 == 0) {
#pragma unroll
            for (int k = 0; k < MBW; ++k) {
                W[0][k] = ld4(r_w, wk[k], 0);
                W[1][k] = ld4(r_w, wk[k] + 1024, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < NB4; ++u) {
            const int row = (tid >> 3) + 32 * (u0 + u);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (sv[u][e] - mean[e]) * rstd[e];        // LayerNorm (lynxnet.py:53), affine in W1 / b1
            *reinterpret_cast<f32x4*>(&xs[row * BN + ((c4 * 4) ^ ((row & 1) << 4))]) = o;
        }
    }
    __syncthreads();

    // ---------------- K walk ----------------
    f32x4 acc[MBW][2];
#pragma unroll
    for (int k = 0; k < MBW; ++k) acc[k][0] = acc[k][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int sw = (lrow & 1) << 4;
    const float* zt0 = xs + lrow * BN + (lcol ^ sw);
    const float* zt1 = xs + lrow * BN + ((16 + lcol) ^ sw);
    // biases of the wave's rows in the accumulator layout (rows rq .. rq + 3 of each block), fetched during the walk
    const int ch0 = 256 * mu + 64 * wave;                        // first u channel of this wave
    const __amdgpu_buffer_rsrc_t r_b = rsrc(p.bias1);
    f32x4 bo[MBW];
    k_phase<KT, 0, false>(acc, W, r_w, wk, 0, zt0, zt1, [&](int s) {
        if (s < MBW) bo[s] = ld4(r_b, rq * 4, ((s & 1) * p.inner + ch0 + (s >> 1) * 16) * 4);
    });

    // ---------------- SwiGLU (common_layers.py:116-117: out * silu(gate)), transposed through LDS, float4 stores ----------------
    __syncthreads();                                             // every wave is done with the activation tile
    float* ew = xs + wave * (64 * ES);                           // wave-private [64 channels][ES]
#pragma unroll
    for (int i = 0; i < MBW / 2; ++i)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float u0 = acc[2 * i][n][r] + bo[2 * i][r];
                const float u1 = acc[2 * i + 1][n][r] + bo[2 * i + 1][r];
                ew[(i * 16 + rq + r) * ES + n * 16 + lcol] = u0 * (u1 * sigmoid_f(u1));
            }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    {
        const dsd_i32x4 w_o = dsd_rsrc_words(p.u + (long)bu * p.u_bstride + (long)ch0 * Ts + t0u);
        const int ev0 = ((lane >> 3) * Ts + (lane & 7) * 4) * 4;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int idx = lane + 64 * m;
            st4(*reinterpret_cast<const f32x4*>(&ew[(idx >> 3) * ES + (idx & 7) * 4]), w_o, ev0, m * 8 * Ts * 4);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// pw1, one workgroup per FRAME tile: the activation tile (and its LayerNorm) is staged ONCE and the workgroup loops over
// the 2 * inner / 512 row tiles - lx_pw1_kernel stages the same 128 KiB tile in each of them (7 of 8 stagings, 262 of the
// 467 MB a launch moves at C = 1024, B = 8) and synchronises its waves twice per tile.  The SwiGLU transpose tiles sit
// BEHIND the activation tile (two halves of 32 channels per wave: 18 KiB), so after the prologue barrier the four waves
// never meet again; the next row tile's first two weight steps are fetched under the epilogue.  For grids of about one
// frame tile per CU (B = 8 at T = 1000); launch_lx_layer picks it by rounds.
// ---------------------------------------------------------------------------------------------------------------
template <int KT, int RAG>
__global__ __launch_bounds__(256, 1) void lx_pw1p_kernel(const LxLayerP p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NS = KT / 16;
    constexpr int NU = KT * (BN / 4) / 256;
    float* xs = lds;                                 // [KT][32], odd rows half-swapped: lives for the whole kernel

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane >> 4, lcol = lane & 15, rq = lrow * 4;
    // work item = (row-tile group g, frame tile ft), g slowest (an XCD walks one group's weights): the workgroup loops over the
    // nmt / rt_groups row tiles of its group (rt_groups = 1: all of them)
    const int work = xcd_work();
    const int nft = RAG ? p.ncg : p.nft;
    const int grp = p.rt_groups > 1 ? fdiv_floor(work, p.inv_nft) : 0;
    const int ft = work - grp * nft;
    const int rest = RAG ? p.cgmap[ft] : ft;
    const int b = fdiv_floor(rest, p.inv_tiles_per_b);
    const int t0 = (rest - b * p.tiles_per_b) * BN;
    const int Ts = p.Ts;
    const int bu = __builtin_amdgcn_readfirstlane(b), t0u = __builtin_amdgcn_readfirstlane(t0);
    const int nmt_all = (2 * p.inner) / 512;                      // row tiles of 256 u channels (512 packed rows)
    const int nmt_g = p.rt_groups > 1 ? nmt_all / p.rt_groups : nmt_all;
    const int mt0 = __builtin_amdgcn_readfirstlane(grp * nmt_g), nmt = mt0 + nmt_g;
#ifdef DSD_STAMPS
    unsigned long long st0 = LX_T(), st1 = 0, swalk = 0, sepi = 0;
#endif

    const int c4 = tid & 7;
    // LayerNorm statistics of the tile's frames: this workgroup is the only reader of its 32 frames, so it merges the
    // producer's per-64-row partials (mean_i, M2_i) itself - ln_merge_kernel's arithmetic in ln_merge_kernel's order (tiles
    // in ascending order; parallel-variance formula), every thread for its own four frames - and no merge launch runs
    // between the layers
    // LayerNorm statistics of the tile's frames: this workgroup is the only reader of its 32 frames and merges the producer's
    // partials itself - no merge launch runs between the layers
    f32x4 mean, rstd;
    lx_merge_stats<KT>(p, bu, t0u, c4, mean, rstd);
    const __amdgpu_buffer_rsrc_t r_x = rsrc(p.xin + (long)bu * p.x_bstride + t0u);
    constexpr long kMtBlocks = (long)MBW * 4 * NS * 256;         // floats of packed weights per row tile
    const __amdgpu_buffer_rsrc_t r_w0 = rsrc(p.A1 + mt0 * kMtBlocks + (long)(MBW * wave) * NS * 256);
    int wk[MBW];
#pragma unroll
    for (int k = 0; k < MBW; ++k) wk[k] = lane * 16 + k * NS * 1024;
    f32x4 W[3][MBW];
    const int xv0 = ((tid >> 3) * Ts + c4 * 4) * 4;
    constexpr int NB4 = 8;
#pragma unroll
    for (int u0 = 0; u0 < NU; u0 += NB4) {
        f32x4 sv[NB4];
#pragma unroll
        for (int u = 0; u < NB4; ++u) sv[u] = ld4(r_x, xv0, (u0 + u) * 32 * Ts * 4);
        if (u0 == 0) {
#pragma unroll
            for (int k = 0; k < MBW; ++k) {
                W[0][k] = ld4(r_w0, wk[k], 0);
                W[1][k] = ld4(r_w0, wk[k] + 1024, 0);
            }
        }
#pragma unroll
        for (int u = 0; u < NB4; ++u) {
            const int row = (tid >> 3) + 32 * (u0 + u);
            f32x4 o;
#pragma unroll
            for (int e = 0; e < 4; ++e) o[e] = (sv[u][e] - mean[e]) * rstd[e];        // LayerNorm (lynxnet.py:53), affine in W1 / b1
            *reinterpret_cast<f32x4*>(&xs[row * BN + ((c4 * 4) ^ ((row & 1) << 4))]) = o;
        }
    }
    __syncthreads();                                             // the only workgroup barrier
#ifdef DSD_STAMPS
    st1 = LX_T();
#endif

    const int sw = (lrow & 1) << 4;
    const float* zt0 = xs + lrow * BN + (lcol ^ sw);
    const float* zt1 = xs + lrow * BN + ((16 + lcol) ^ sw);
    const __amdgpu_buffer_rsrc_t r_b = rsrc(p.bias1);
    float* ew = lds + KT * BN + wave * (32 * ES);                // wave-private [32 channels][ES], behind the activation tile
    const int ev0 = ((lane >> 3) * Ts + (lane & 7) * 4) * 4;
#pragma unroll 1
    for (int mt = mt0; mt < nmt; ++mt) {
        const __amdgpu_buffer_rsrc_t r_w = rsrc(p.A1 + mt * kMtBlocks + (long)(MBW * wave) * NS * 256);
        f32x4 acc[MBW][2];
#pragma unroll
        for (int k = 0; k < MBW; ++k) acc[k][0] = acc[k][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        const int ch0 = 256 * mt + 64 * wave;                    // first u channel of this wave in this row tile
        f32x4 bo[MBW];
#ifdef DSD_STAMPS
        const unsigned long long ta = LX_T();
#endif
        k_phase<KT, 0, false>(acc, W, r_w, wk, 0, zt0, zt1, [&](int s) {
            if (s < MBW) bo[s] = ld4(r_b, rq * 4, ((s & 1) * p.inner + ch0 + (s >> 1) * 16) * 4);
        });
#ifdef DSD_STAMPS
        const unsigned long long tb = LX_T();
        swalk += tb - ta;
#endif
        if (mt + 1 < nmt) {                                      // the next row tile's steps 0 and 1 land under the epilogue
            const __amdgpu_buffer_rsrc_t r_n = rsrc(p.A1 + (mt + 1) * kMtBlocks + (long)(MBW * wave) * NS * 256);
#pragma unroll
            for (int k = 0; k < MBW; ++k) {
                W[0][k] = ld4(r_n, wk[k], 0);
                W[1][k] = ld4(r_n, wk[k] + 1024, 0);
            }
        }
        // SwiGLU (common_layers.py:116-117: out * silu(gate)) in two halves of 32 channels through the wave's own tile
        const dsd_i32x4 w_o = dsd_rsrc_words(p.u + (long)bu * p.u_bstride + (long)ch0 * Ts + t0u);
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
            for (int ii = 0; ii < 2; ++ii) {
                const int i = 2 * hf + ii;
#pragma unroll
                for (int n = 0; n < 2; ++n)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float u0 = acc[2 * i][n][r] + bo[2 * i][r];
                        const float u1 = acc[2 * i + 1][n][r] + bo[2 * i + 1][r];
                        ew[(ii * 16 + rq + r) * ES + n * 16 + lcol] = u0 * (u1 * sigmoid_f(u1));
                    }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const int idx = lane + 64 * m;
                st4(*reinterpret_cast<const f32x4*>(&ew[(idx >> 3) * ES + (idx & 7) * 4]), w_o, ev0, (hf * 32 + m * 8) * Ts * 4);
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();                     // the tile is read before the next half overwrites it
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
#ifdef DSD_STAMPS
        sepi += LX_T() - tb;
#endif
    }
#ifdef DSD_STAMPS
    if (tid == 0 && blockIdx.x < 4096) {
        g_lx_stamps[blockIdx.x][0] = st0;
        g_lx_stamps[blockIdx.x][1] = st1;
        g_lx_stamps[blockIdx.x][2] = swalk;
        g_lx_stamps[blockIdx.x][3] = sepi;
        g_lx_stamps[blockIdx.x][4] = LX_T();
    }
#endif
}

// ---------------------------------------------------------------------------------------------------------------
// pw2: 1x1 inner -> C on the depthwise conv's output, + bias + residual, then the next layer's transition and the
// LayerNorm partials of its input.  K = inner walked as NP = inner / 1024 resident phases of KT = 1024 channels (or one
// phase of 512 / 1024).  Workgroup = 512 rows x 32 frames; wave w: rows [512 mtile + 128 w, +128) = 64-row tiles 2w, 2w+1.
// ---------------------------------------------------------------------------------------------------------------
template <int KT, int RAG>
__global__ __launch_bounds__(256, 1) void lx_pw2_kernel(const LxLayerP p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NS = KT / 16;
    constexpr int NU = KT * (BN / 4) / 256;
    float* xs = lds;

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
    const int t0 = (rest - b * p.tiles_per_b) * BN;
    const int Ts = p.Ts;
    const int bu = __builtin_amdgcn_readfirstlane(b), t0u = __builtin_amdgcn_readfirstlane(t0);
    const int mu = __builtin_amdgcn_readfirstlane(mtile);
    const int np = p.inner / KT;                                 // resident phases
    const int NSA = p.inner / 16;                                // k16 steps of the whole weight stream

    const int c4 = tid & 7;
    const __amdgpu_buffer_rsrc_t r_v = rsrc(p.v + (long)bu * p.u_bstride + t0u);
    const __amdgpu_buffer_rsrc_t r_w = rsrc(p.A2 + (long)(MBW * 4 * mu + MBW * wave) * NSA * 256);
    int wk[MBW];
#pragma unroll
    for (int k = 0; k < MBW; ++k) wk[k] = lane * 16 + k * NSA * 1024;
    f32x4 W[3][MBW];
    const int xv0 = ((tid >> 3) * Ts + c4 * 4) * 4;
    constexpr int NB4 = 8;
    auto stage = [&](int phase, bool first) {
#pragma unroll
        for (int u0 = 0; u0 < NU; u0 += NB4) {
            f32x4 sv[NB4];
#pragma unroll
            for (int u = 0; u < NB4; ++u) sv[u] = ld4(r_v, xv0, (phase * KT + (u0 + u) * 32) * Ts * 4);
            if (first && u0 == 0) {
#pragma unroll
                for (int k = 0; k < MBW; ++k) {
                    W[0][k] = ld4(r_w, wk[k], 0);
                    W[1][k] = ld4(r_w, wk[k] + 1024, 0);
                }
            }
#pragma unroll
            for (int u = 0; u < NB4; ++u) {
                const int row = (tid >> 3) + 32 * (u0 + u);
                *reinterpret_cast<f32x4*>(&xs[row * BN + ((c4 * 4) ^ ((row & 1) << 4))]) = sv[u];
            }
        }
    };
    // bias and the next layer's step-projection scalar of the workgroup's 512 rows -> LDS tables behind the tiles
    constexpr int TBL = (KT * BN > 4 * 128 * ES) ? KT * BN : 4 * 128 * ES;
    {
        const __amdgpu_buffer_rsrc_t r_b = rsrc(p.bias2 + 512 * mu);
        const float b0 = ld1(r_b, tid * 4, 0), b1 = ld1(r_b, tid * 4, 1024);
        float f0 = 0.f, f1 = 0.f;
        if (p.film) {
            const __amdgpu_buffer_rsrc_t r_f = rsrc(p.film + p.film_col0 + bu * p.film_colb + (long)512 * mu * p.film_cstride);
            f0 = ld1(r_f, tid * p.film_cstride * 4, 0);
            f1 = ld1(r_f, (tid + 256) * p.film_cstride * 4, 0);
        }
        lds[TBL + tid] = b0;
        lds[TBL + 256 + tid] = b1;
        lds[TBL + 512 + tid] = f0;
        lds[TBL + 768 + tid] = f1;
    }
    stage(0, true);
    __syncthreads();

    f32x4 acc[MBW][2];
#pragma unroll
    for (int k = 0; k < MBW; ++k) acc[k][0] = acc[k][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int sw = (lrow & 1) << 4;
    const float* zt0 = xs + lrow * BN + (lcol ^ sw);
    const float* zt1 = xs + lrow * BN + ((16 + lcol) ^ sw);
    // epilogue operands, row-major float4 over the wave's 128 rows (lane -> row (lane >> 3) + 8 m, frames 4 (lane & 7)):
    // residual x, the next layer's conditioner projection; and per row its bias and step-projection scalar
    const int row0 = 512 * mu + 128 * wave;
    const int ev0 = ((lane >> 3) * Ts + (lane & 7) * 4) * 4;
    const __amdgpu_buffer_rsrc_t r_a = rsrc(p.x + (long)bu * p.x_bstride + (long)row0 * Ts + t0u);
    const __amdgpu_buffer_rsrc_t r_c = rsrc((p.cpn ? p.cpn : p.x) + (long)bu * (p.cpn ? p.cpn_bstride : p.x_bstride) + (long)row0 * Ts + t0u);
    f32x4 aux[16], cpv[16];
    auto tail_loads = [&](int s) {
        if (s < 16) aux[s] = ld4(r_a, ev0, s * 8 * Ts * 4);
        else if (s < 32) cpv[s - 16] = ld4(r_c, ev0, (s - 16) * 8 * Ts * 4);
    };
    auto none = [](int) {};
    if (np == 1) {
        k_phase<KT, 0, false>(acc, W, r_w, wk, 0, zt0, zt1, tail_loads);
    } else {                                                     // two resident phases (inner = 2 KT)
        k_phase<KT, 0, true>(acc, W, r_w, wk, 0, zt0, zt1, none);
        __syncthreads();                                         // every wave is done with the first phase's tile
        stage(1, false);
        __syncthreads();
        k_phase<KT, NS % 3, false>(acc, W, r_w, wk, NS, zt0, zt1, tail_loads);
    }
    __syncthreads();                                             // the tile is dead: the epilogue tiles go over it
    const float* tb = lds + TBL;                                 // bias and step-projection scalar of the workgroup's 512 rows
    const float* tf = tb + 512;

    // ---------------- transition (gemm.hip EP_LYNX_NEXT; lynxnet.py:76-84 of the next layer), row-major ----------------
    float* ew = xs + wave * (128 * ES);                          // wave-private [128 rows][ES]: 18 KiB x 4 waves
#pragma unroll
    for (int k = 0; k < MBW; ++k)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) ew[(k * 16 + rq + r) * ES + n * 16 + lcol] = acc[k][n][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const dsd_i32x4 w_xo = dsd_rsrc_words(p.x + (long)bu * p.x_bstride + (long)row0 * Ts + t0u);
    const dsd_i32x4 w_xi = dsd_rsrc_words((p.xin_out ? p.xin_out : p.x) + (long)bu * p.x_bstride + (long)row0 * Ts + t0u);
    f32x4 xi[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        const int idx = lane + 64 * m;
        const f32x4 a4 = *reinterpret_cast<const f32x4*>(&ew[(idx >> 3) * ES + (idx & 7) * 4]);
        const float brow = tb[128 * wave + (idx >> 3)], frow = tf[128 * wave + (idx >> 3)];
        f32x4 xo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float v = (a4[e] + brow) + aux[m][e];              // + bias, + residual (lynxnet.py:86)
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
    // LayerNorm partials of xin per 64-row tile (tiles 2w, 2w + 1 of this workgroup's 8): two passes over the registers.
    // A frame's 64 rows sit in 8 slots m of the 8 lanes with equal (lane & 7): sum over m, then over lanes 8, 16, 32 apart.
    if (p.lnpart) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int m = 0; m < 8; ++m) s += xi[8 * h + m];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s[e] += __shfl_xor(s[e], 8, 64);
                s[e] += __shfl_xor(s[e], 16, 64);
                s[e] += __shfl_xor(s[e], 32, 64);
            }
            const f32x4 mu4 = s * (1.f / 64.f);
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
                float* lp = p.lnpart + ((long)bu * p.ln_tiles + tile) * 2 * p.lnpart_ts + t0u + lane * 4;
                *reinterpret_cast<f32x4*>(lp) = mu4;
                *reinterpret_cast<f32x4*>(lp + p.lnpart_ts) = q;
            }
        }
    }
}
// pw2 with the staging HIDDEN: K = inner is walked as NP half-phases of 512 channels through TWO 64 KiB LDS buffers; while
// half-phase i runs from one buffer, half-phase i + 1 is fetched (one 16-byte load per thread and step over the first 16
// steps) and written to the other (one LDS store per step over the last 16) inside the walk, so only the first 64 KiB are
// staged with the MFMA pipe idle - lx_pw2_kernel stages 2 x 128 KiB that way, between two barriers each.
template <int NP, int RAG>
__global__ __launch_bounds__(256, 1) void lx_pw2d_kernel(const LxLayerP p) {
    constexpr int KT = 512;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int NS = KT / 16;
    constexpr int NU = KT * (BN / 4) / 256;
    float* xs = lds;                                             // (the epilogue tiles)

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
    const int t0 = (rest - b * p.tiles_per_b) * BN;
    const int Ts = p.Ts;
    const int bu = __builtin_amdgcn_readfirstlane(b), t0u = __builtin_amdgcn_readfirstlane(t0);
    const int mu = __builtin_amdgcn_readfirstlane(mtile);
    constexpr int NSA = NP * KT / 16;                            // k16 steps of the whole weight stream (inner = NP * 512)
    float* xb[2] = {lds, lds + KT * BN};                         // the two half-phase buffers

    const int c4 = tid & 7;
    const __amdgpu_buffer_rsrc_t r_v = rsrc(p.v + (long)bu * p.u_bstride + t0u);
    const __amdgpu_buffer_rsrc_t r_w = rsrc(p.A2 + (long)(MBW * 4 * mu + MBW * wave) * NSA * 256);
    int wk[MBW];
#pragma unroll
    for (int k = 0; k < MBW; ++k) wk[k] = lane * 16 + k * NSA * 1024;
    f32x4 W[3][MBW];
    const int xv0 = ((tid >> 3) * Ts + c4 * 4) * 4;
    constexpr int NB4 = 8;
    static_assert(NU == 16 && NS == 32, "one staging load per step over the first half of a walk, one store over the second");
    auto stage0 = [&]() {                                        // half-phase 0: the only staging outside a walk
#pragma unroll
        for (int u0 = 0; u0 < NU; u0 += NB4) {
            f32x4 sv[NB4];
#pragma unroll
            for (int u = 0; u < NB4; ++u) sv[u] = ld4(r_v, xv0, (u0 + u) * 32 * Ts * 4);
            if (u0 == 0) {
#pragma unroll
                for (int k = 0; k < MBW; ++k) {
                    W[0][k] = ld4(r_w, wk[k], 0);
                    W[1][k] = ld4(r_w, wk[k] + 1024, 0);
                }
            }
#pragma unroll
            for (int u = 0; u < NB4; ++u) {
                const int row = (tid >> 3) + 32 * (u0 + u);
                *reinterpret_cast<f32x4*>(&xb[0][row * BN + ((c4 * 4) ^ ((row & 1) << 4))]) = sv[u];
            }
        }
    };
    f32x4 nx[NU];                                                // the next half-phase on its way through registers
    // bias and the next layer's step-projection scalar of the workgroup's 512 rows -> LDS tables behind the tiles
    constexpr int TBL = 2 * KT * BN;                             // behind both buffers (>= the epilogue tiles' 4 * 128 * ES)
    static_assert(TBL >= 4 * 128 * ES, "the epilogue tiles go over the two buffers");
    {
        const __amdgpu_buffer_rsrc_t r_b = rsrc(p.bias2 + 512 * mu);
        const float b0 = ld1(r_b, tid * 4, 0), b1 = ld1(r_b, tid * 4, 1024);
        float f0 = 0.f, f1 = 0.f;
        if (p.film) {
            const __amdgpu_buffer_rsrc_t r_f = rsrc(p.film + p.film_col0 + bu * p.film_colb + (long)512 * mu * p.film_cstride);
            f0 = ld1(r_f, tid * p.film_cstride * 4, 0);
            f1 = ld1(r_f, (tid + 256) * p.film_cstride * 4, 0);
        }
        lds[TBL + tid] = b0;
        lds[TBL + 256 + tid] = b1;
        lds[TBL + 512 + tid] = f0;
        lds[TBL + 768 + tid] = f1;
    }
    stage0();
    __syncthreads();

    f32x4 acc[MBW][2];
#pragma unroll
    for (int k = 0; k < MBW; ++k) acc[k][0] = acc[k][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int sw = (lrow & 1) << 4;
    const float* za0 = xb[0] + lrow * BN + (lcol ^ sw);
    const float* za1 = xb[0] + lrow * BN + ((16 + lcol) ^ sw);
    const float* zb0 = za0 + KT * BN;
    const float* zb1 = za1 + KT * BN;
    // epilogue operands, row-major float4 over the wave's 128 rows (lane -> row (lane >> 3) + 8 m, frames 4 (lane & 7)):
    // residual x, the next layer's conditioner projection; and per row its bias and step-projection scalar
    const int row0 = 512 * mu + 128 * wave;
    const int ev0 = ((lane >> 3) * Ts + (lane & 7) * 4) * 4;
    const __amdgpu_buffer_rsrc_t r_a = rsrc(p.x + (long)bu * p.x_bstride + (long)row0 * Ts + t0u);
    const __amdgpu_buffer_rsrc_t r_c = rsrc((p.cpn ? p.cpn : p.x) + (long)bu * (p.cpn ? p.cpn_bstride : p.x_bstride) + (long)row0 * Ts + t0u);
    f32x4 aux[16], cpv[16];
    auto tail_loads = [&](int s) {
        if (s < 16) aux[s] = ld4(r_a, ev0, s * 8 * Ts * 4);
        else if (s < 32) cpv[s - 16] = ld4(r_c, ev0, (s - 16) * 8 * Ts * 4);
    };
    // half-phase ph + 1 travels inside half-phase ph's walk: loads over steps 0..15, LDS stores over steps 16..31
    auto carry = [&](int ph, int s) {
        if (s < NU) {
            nx[s] = ld4(r_v, xv0, ((ph + 1) * KT + s * 32) * Ts * 4);
        } else {
            const int u = s - NU;
            const int row = (tid >> 3) + 32 * u;
            *reinterpret_cast<f32x4*>(&xb[(ph + 1) & 1][row * BN + ((c4 * 4) ^ ((row & 1) << 4))]) = nx[u];
        }
    };
    // (the weight rotation continues across the half-phases: ROT = 32 ph mod 3)
    k_phase<KT, 0, true>(acc, W, r_w, wk, 0, za0, za1, [&](int s) { carry(0, s); });
    __syncthreads();                                             // buffer 1 complete; every wave is done with buffer 0
    if (NP == 2) {
        k_phase<KT, 2, false>(acc, W, r_w, wk, NS, zb0, zb1, tail_loads);
    } else {
        k_phase<KT, 2, true>(acc, W, r_w, wk, NS, zb0, zb1, [&](int s) { carry(1, s); });
        __syncthreads();
        k_phase<KT, 1, true>(acc, W, r_w, wk, 2 * NS, za0, za1, [&](int s) { carry(2, s); });
        __syncthreads();
        k_phase<KT, 0, false>(acc, W, r_w, wk, 3 * NS, zb0, zb1, tail_loads);
    }
    __syncthreads();                                             // the buffers are dead: the epilogue tiles go over them
    const float* tb = lds + TBL;                                 // bias and step-projection scalar of the workgroup's 512 rows
    const float* tf = tb + 512;

    // ---------------- transition (gemm.hip EP_LYNX_NEXT; lynxnet.py:76-84 of the next layer), row-major ----------------
    float* ew = xs + wave * (128 * ES);                          // wave-private [128 rows][ES]: 18 KiB x 4 waves
#pragma unroll
    for (int k = 0; k < MBW; ++k)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
            for (int r = 0; r < 4; ++r) ew[(k * 16 + rq + r) * ES + n * 16 + lcol] = acc[k][n][r];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const dsd_i32x4 w_xo = dsd_rsrc_words(p.x + (long)bu * p.x_bstride + (long)row0 * Ts + t0u);
    const dsd_i32x4 w_xi = dsd_rsrc_words((p.xin_out ? p.xin_out : p.x) + (long)bu * p.x_bstride + (long)row0 * Ts + t0u);
    f32x4 xi[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) {
        const int idx = lane + 64 * m;
        const f32x4 a4 = *reinterpret_cast<const f32x4*>(&ew[(idx >> 3) * ES + (idx & 7) * 4]);
        const float brow = tb[128 * wave + (idx >> 3)], frow = tf[128 * wave + (idx >> 3)];
        f32x4 xo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float v = (a4[e] + brow) + aux[m][e];              // + bias, + residual (lynxnet.py:86)
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
    // LayerNorm partials of xin per 64-row tile (tiles 2w, 2w + 1 of this workgroup's 8): two passes over the registers.
    // A frame's 64 rows sit in 8 slots m of the 8 lanes with equal (lane & 7): sum over m, then over lanes 8, 16, 32 apart.
    if (p.lnpart) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int m = 0; m < 8; ++m) s += xi[8 * h + m];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                s[e] += __shfl_xor(s[e], 8, 64);
                s[e] += __shfl_xor(s[e], 16, 64);
                s[e] += __shfl_xor(s[e], 32, 64);
            }
            const f32x4 mu4 = s * (1.f / 64.f);
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
                float* lp = p.lnpart + ((long)bu * p.ln_tiles + tile) * 2 * p.lnpart_ts + t0u + lane * 4;
                *reinterpret_cast<f32x4*>(lp) = mu4;
                *reinterpret_cast<f32x4*>(lp + p.lnpart_ts) = q;
            }
        }
    }
}
#undef LX_SPREAD

// ---------------------------------------------------------------------------------------------------------------
// pw2 for ONE-UTTERANCE grids (round 3): 128 output rows per workgroup - C / 128 workgroups per frame tile, 256 for one
// utterance of ~1000 frames at C = 1024 - where lx_pw2d_kernel's 512-row workgroups leave three quarters of the CUs idle and
// gemm.hip's EP_LYNX_NEXT GEMM (64-row tiles, 16 MFMAs per k16 step, a barrier per 64-channel chunk) ran at 0.53 of peak.
// 512 threads = 8 waves = four K QUARTERS (kq: inner / 4 channels each) x two row waves (wr: 64 rows = 4 row blocks): a B
// fragment read from LDS feeds 8 MFMAs, a k16 step is 32 MFMAs, two waves share a SIMD - the layout that made the WaveNet's
// two-launch conv MFMA-bound (wn_rows.hip).  Each quarter's channels travel through LDS in sub-phases of 128 (8 steps):
// while sub-phase i runs from one 64 KiB buffer (4 quarters x 128 channels x 32 frames), sub-phase i + 1 is fetched (two
// 16-byte loads per thread and step over steps 0-3) and written to the other (steps 4-7); one barrier per sub-phase.  The
// quarters' partial sums meet in LDS over the dead buffers; the transition (bias, residual, the next layer's conditioner /
// step projections; LayerNorm partials per 64-row tile) is lx_pw2d_kernel's, on 2 items per thread.
// ---------------------------------------------------------------------------------------------------------------
#ifndef DSD_LXQ_SPREAD
#define DSD_LXQ_SPREAD 1
#endif
#if DSD_LXQ_SPREAD == 1
#define LXQ_SPREAD()                                                                 \
    _Pragma("unroll") for (int g_ = 0; g_ < 8; ++g_) {                               \
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                           \
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                           \
        __builtin_amdgcn_sched_group_barrier(0x080, 2, 0);                           \
    }                                                                                \
    __builtin_amdgcn_sched_barrier(0);
#elif DSD_LXQ_SPREAD == 2
#define LXQ_SPREAD()                                                                 \
    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);                               \
    __builtin_amdgcn_sched_group_barrier(0x100, 8, 0);                               \
    _Pragma("unroll") for (int g_ = 0; g_ < 7; ++g_) {                               \
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);                           \
        __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                           \
        __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);                           \
    }                                                                                \
    __builtin_amdgcn_sched_barrier(0);
#else
#define LXQ_SPREAD() __builtin_amdgcn_sched_barrier(0);
#endif

template <int KQ, int RAG>
__global__ __launch_bounds__(512, 1) void lx_pw2q_kernel(const LxLayerP p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int SUB = 128;                         // channels of a quarter per sub-phase
    constexpr int NSUB = KQ / SUB;                   // sub-phases: 4 (inner 2048), 2 (inner 1024)
    constexpr int SS = SUB / 16;                     // k16 steps per sub-phase: 8
    constexpr int MB = 4;                            // row blocks per wave
    constexpr int NSA = 4 * KQ / 16;                 // k16 steps of a row block's whole weight stream
    constexpr int BUF = 4 * SUB * BN;                // one buffer: [quarter][SUB][32] floats = 64 KiB
    float* tbl = lds + 2 * BUF;                      // bias [128], next layer's step-projection scalar [128]
    float* red = tbl + 256;                          // LayerNorm partial sums: [2 tiles][8 waves][32 frames]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kq = wave >> 1, wr = wave & 1;
    const int lrow = lane >> 4, lcol = lane & 15, rq = lrow * 4;
    const int work = xcd_work();
    const int nft = RAG ? p.ncg : p.nft;
    const int mtile = fdiv_floor(work, p.inv_nft);               // row tile of 128 rows (slowest: an XCD keeps one row tile's weights)
    const int ft = work - mtile * nft;
    const int rest = RAG ? p.cgmap[ft] : ft;
    const int b = fdiv_floor(rest, p.inv_tiles_per_b);
    const int t0 = (rest - b * p.tiles_per_b) * BN;
    const int Ts = p.Ts;
    const int bu = __builtin_amdgcn_readfirstlane(b), t0u = __builtin_amdgcn_readfirstlane(t0);
    const int mu = __builtin_amdgcn_readfirstlane(mtile);

    // ---------------- prologue: tables, sub-phase 0 of every quarter, the first two weight steps ----------------
    const int c4 = tid & 7, srow = tid >> 3;                     // staging: row srow + 64 u of the 512 staged rows, frames 4 c4 .. + 3
    const __amdgpu_buffer_rsrc_t r_v = rsrc(p.v + (long)bu * p.u_bstride + t0u);
    const int xv0 = (srow * Ts + c4 * 4) * 4;
    // staged row srow + 64 u = quarter u >> 1, channel (u & 1) * 64 + srow of the sub-phase
    auto v_soff = [&](int sp, int u) { return (((u >> 1) * KQ + sp * SUB + (u & 1) * 64) * Ts) * 4; };
    auto lds_off = [&](int u) {
        const int r = (u & 1) * 64 + srow;
        return ((u >> 1) * SUB + r) * BN + ((c4 * 4) ^ ((r & 1) << 4));
    };
    f32x4 nx[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) nx[u] = ld4(r_v, xv0, v_soff(0, u));
    const __amdgpu_buffer_rsrc_t r_w = rsrc(p.A2 + ((long)(8 * mu + MB * wr) * NSA + kq * (KQ / 16)) * 256);
    int wk[MB];
#pragma unroll
    for (int k = 0; k < MB; ++k) wk[k] = lane * 16 + k * NSA * 1024;
    f32x4 W[3][MB];
#pragma unroll
    for (int k = 0; k < MB; ++k) {
        W[0][k] = ld4(r_w, wk[k], 0);
        W[1][k] = ld4(r_w, wk[k] + 1024, 0);
    }
    if (tid < 128) {
        tbl[tid] = ld1(rsrc(p.bias2 + 128 * mu), tid * 4, 0);
        float f = 0.f;
        if (p.film) f = ld1(rsrc(p.film + p.film_col0 + bu * p.film_colb + (long)128 * mu * p.film_cstride), tid * p.film_cstride * 4, 0);
        tbl[128 + tid] = f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) *reinterpret_cast<f32x4*>(&lds[lds_off(u)]) = nx[u];
    __syncthreads();

    // ---------------- K walk: NSUB sub-phases of 8 steps ----------------
    f32x4 acc[MB][2];
#pragma unroll
    for (int k = 0; k < MB; ++k) acc[k][0] = acc[k][1] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int sw = (lrow & 1) << 4;
    const int zoff0 = (kq * SUB + lrow) * BN + (lcol ^ sw), zoff1 = (kq * SUB + lrow) * BN + ((16 + lcol) ^ sw);
    // epilogue operands: item i = tid + 512 k -> row i >> 3 of the workgroup's 128, frames 4 (i & 7): residual x, next layer's
    // hoisted conditioner projection
    const int row0 = 128 * mu;
    const int ev0 = ((tid >> 3) * Ts + (tid & 7) * 4) * 4;
    const __amdgpu_buffer_rsrc_t r_a = rsrc(p.x + (long)bu * p.x_bstride + (long)row0 * Ts + t0u);
    const __amdgpu_buffer_rsrc_t r_c = rsrc((p.cpn ? p.cpn : p.x) + (long)bu * (p.cpn ? p.cpn_bstride : p.x_bstride) + (long)row0 * Ts + t0u);
    f32x4 aux[2], cpv[2];
    float bq[2][4][2];
#pragma unroll
    for (int sp = 0; sp < NSUB; ++sp) {
        const float* cur = lds + (sp & 1) * BUF;
        float* nxt = lds + ((sp + 1) & 1) * BUF;
        if (sp == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bq[0][j][0] = cur[zoff0 + (j * 4) * BN];
                bq[0][j][1] = cur[zoff1 + (j * 4) * BN];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int s = 0; s < SS; ++s) {
            const int g = sp * SS + s;                           // k16 step of the quarter
            if (g + 2 < NSUB * SS) {
#pragma unroll
                for (int k = 0; k < MB; ++k) W[(g + 2) % 3][k] = ld4(r_w, wk[k] + ((g + 2) & 3) * 1024, ((g + 2) >> 2) * 4096);
            }
            if (sp + 1 < NSUB) {                                 // the next sub-phase: loads over steps 0-3, LDS stores over steps 4-7
                if (s < 4) {
                    nx[2 * s] = ld4(r_v, xv0, v_soff(sp + 1, 2 * s));
                    nx[2 * s + 1] = ld4(r_v, xv0, v_soff(sp + 1, 2 * s + 1));
                } else {
                    *reinterpret_cast<f32x4*>(&nxt[lds_off(2 * (s - 4))]) = nx[2 * (s - 4)];
                    *reinterpret_cast<f32x4*>(&nxt[lds_off(2 * (s - 4) + 1)]) = nx[2 * (s - 4) + 1];
                }
            } else if (s < 2) {                                  // the last sub-phase carries the epilogue operands
                aux[s] = ld4(r_a, ev0, s * 64 * Ts * 4);
            } else if (s < 4) {
                cpv[s - 2] = ld4(r_c, ev0, (s - 2) * 64 * Ts * 4);
            }
            if (s + 1 < SS) {                                    // B fragments of the next step (the next sub-phase's first: behind the barrier)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    bq[(g + 1) & 1][j][0] = cur[zoff0 + ((s + 1) * 16 + j * 4) * BN];
                    bq[(g + 1) & 1][j][1] = cur[zoff1 + ((s + 1) * 16 + j * 4) * BN];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int k = 0; k < MB; ++k) {
                    acc[k][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(W[g % 3][k][j], bq[g & 1][j][0], acc[k][0], 0, 0, 0);
                    acc[k][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(W[g % 3][k][j], bq[g & 1][j][1], acc[k][1], 0, 0, 0);
                }
            LXQ_SPREAD()
        }
        __syncthreads();                                         // the other buffer is complete; every wave is done with this one
        if (sp + 1 < NSUB) {
            const float* nb = lds + ((sp + 1) & 1) * BUF;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bq[((sp + 1) * SS) & 1][j][0] = nb[zoff0 + (j * 4) * BN];
                bq[((sp + 1) * SS) & 1][j][1] = nb[zoff1 + (j * 4) * BN];
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---------------- the quarters' partial sums -> LDS tiles [kq][128 rows][ES] over the dead buffers ----------------
    {
        float* tk = lds + kq * (128 * ES);
#pragma unroll
        for (int k = 0; k < MB; ++k)
#pragma unroll
            for (int n = 0; n < 2; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) tk[((MB * wr + k) * 16 + rq + r) * ES + n * 16 + lcol] = acc[k][n][r];
    }
    __syncthreads();
    // ---------------- transition (gemm.hip EP_LYNX_NEXT; lynxnet.py:76-84 of the next layer), row-major ----------------
    const dsd_i32x4 w_xo = dsd_rsrc_words(p.x + (long)bu * p.x_bstride + (long)row0 * Ts + t0u);
    const dsd_i32x4 w_xi = dsd_rsrc_words((p.xin_out ? p.xin_out : p.x) + (long)bu * p.x_bstride + (long)row0 * Ts + t0u);
    f32x4 xi[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int row = (tid >> 3) + 64 * k;
        f32x4 a4 = *reinterpret_cast<const f32x4*>(&lds[row * ES + (tid & 7) * 4]);
#pragma unroll
        for (int q = 1; q < 4; ++q) a4 += *reinterpret_cast<const f32x4*>(&lds[q * (128 * ES) + row * ES + (tid & 7) * 4]);
        const float brow = tbl[row], frow = tbl[128 + row];
        f32x4 xo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float v = (a4[e] + brow) + aux[k][e];              // + bias, + residual (lynxnet.py:86)
            float o = v, in = v;
            if (p.cpn) {
                in = v + cpv[k][e];
                if (p.strong) o = in;
            }
            if (p.film) in = in + frow;
            xo[e] = o;
            xi[k][e] = in;
        }
        st4(xo, w_xo, ev0, k * 64 * Ts * 4);
        if (p.xin_out) st4(xi[k], w_xi, ev0, k * 64 * Ts * 4);
    }
    // LayerNorm partials of xin per 64-row tile (item k of every thread = tile k of this workgroup): two passes.  A frame quad's 64
    // rows sit in the 8 lanes with equal (lane & 7) of each of the 8 waves: lanes 8, 16, 32 apart, then the waves through LDS.
    if (p.lnpart) {
        auto wave_sum = [&](f32x4 v) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] += __shfl_xor(v[e], 8, 64);
                v[e] += __shfl_xor(v[e], 16, 64);
                v[e] += __shfl_xor(v[e], 32, 64);
            }
            return v;
        };
        auto all_sum = [&](int k) {                              // over the 8 waves, in wave order
            f32x4 t = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < 8; ++w) t += *reinterpret_cast<const f32x4*>(&red[(k * 8 + w) * 32 + (tid & 7) * 4]);
            return t;
        };
        f32x4 mu4[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const f32x4 sv = wave_sum(xi[k]);
            if (lane < 8) *reinterpret_cast<f32x4*>(&red[(k * 8 + wave) * 32 + lane * 4]) = sv;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 2; ++k) mu4[k] = all_sum(k) * (1.f / 64.f);
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const f32x4 d = xi[k] - mu4[k];
            const f32x4 qv = wave_sum(d * d);
            if (lane < 8) *reinterpret_cast<f32x4*>(&red[(k * 8 + wave) * 32 + lane * 4]) = qv;
        }
        __syncthreads();
        if (tid < 16) {                                          // tile k = tid >> 3, frame quad tid & 7
            const int k = tid >> 3;
            const f32x4 q4 = all_sum(k);
            float* lp = p.lnpart + ((long)bu * p.ln_tiles + 2 * mu + k) * 2 * p.lnpart_ts + t0u + (tid & 7) * 4;
            *reinterpret_cast<f32x4*>(lp) = mu4[k];
            *reinterpret_cast<f32x4*>(lp + p.lnpart_ts) = q4;
        }
    }
}
#undef LXQ_SPREAD

int lx_pw2q_lds_bytes() { return (2 * 4 * 128 * 32 + 256 + 2 * 8 * 32) * 4; }
bool lx_pw2q_supported(int C, int inner) { return (C == 1024 && inner == 2048) || (C == 512 && inner == 1024); }

template <int KQ, int RAG>
static hipError_t lx_launch_pw2q(const LxLayerP& p, int nwg, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(lx_pw2q_kernel<KQ, RAG>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr = true;
    }
    if (nwg == 0) return hipSuccess;
    return launch_timed(lx_pw2q_kernel<KQ, RAG>, dim3(nwg), dim3(512), lx_pw2q_lds_bytes(), st, p, "lx_pw2q_kernel<%d, %d>", KQ, RAG);
}

// pw2 with 128 rows per workgroup (one-utterance grids); p as for launch_lx_layer
hipError_t launch_lx_pw2q(const LxLayerP& p, int C, hipStream_t st) {
    if (!lx_pw2q_supported(C, p.inner)) return hipErrorInvalidValue;
    const int nft = p.cgmap ? p.ncg : p.nft;
    const int nwg = nft * (C / 128);
    if (p.inner == 2048) return p.cgmap ? lx_launch_pw2q<512, 1>(p, nwg, st) : lx_launch_pw2q<512, 0>(p, nwg, st);
    return p.cgmap ? lx_launch_pw2q<256, 1>(p, nwg, st) : lx_launch_pw2q<256, 0>(p, nwg, st);
}

int lx_lds_bytes(int kt) { return (kt * 32 * 4 > 4 * 128 * 36 * 4 ? kt * 32 * 4 : 4 * 128 * 36 * 4) + 1024 * 4; }
int lx_pw1p_lds_bytes(int kt) { return kt * 32 * 4 + 4 * 32 * ES * 4; }        // activation tile + the four waves' half tiles

bool lx_layer_supported(int C, int inner) {
    // pw1 needs C = the resident K (512 / 1024) and 2 * inner rows in workgroups of 512; pw2 C rows in workgroups of 512 and
    // inner a whole number of resident phases
    return (C == 512 || C == 1024) && (2 * inner) % 512 == 0 && C % 512 == 0 && inner % C == 0 && inner / C <= 2;
}

template <typename K>
static hipError_t lx_attr(K kern) {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

template <int KT, int RAG>
static hipError_t lx_launch_pw1p(const LxLayerP& p, int nwg, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        hipError_t e = lx_attr(lx_pw1p_kernel<KT, RAG>);
        if (e != hipSuccess) return e;
        attr = true;
    }
    if (nwg == 0) return hipSuccess;
    const int ldsb = lx_pw1p_lds_bytes(KT);
    return launch_timed(lx_pw1p_kernel<KT, RAG>, dim3(nwg), dim3(256), ldsb, st, p, "lx_pw1p_kernel<%d, %d>", KT, RAG);
}

// pw1 with the activation tile staged once per workgroup and a loop over row tiles: how many row-tile GROUPS (= workgroups per
// frame tile) - 1: all row tiles in one workgroup (lx_pw1p_kernel as in round 2), 2 / 4: half / a quarter of them, 0: one
// workgroup per (frame tile, row tile) = lx_pw1_kernel.  By rounds of the chip: a workgroup takes ~11 us of prologue (statistics
// merge, 128 KiB staging, launch ramp) + ~60 us per row tile (484 us at B = 8 for 8 row tiles, 254 for 4, 132 for 2); the
// groups that fill whole rounds win - B = 2 / 4 / 6 at T = 1000: 4 / 2 / 4 groups.  DSD_LYNX_PW1P=0/1 forces none / one group.
static int lx_pw1p_groups(int nft, int mtiles) {
    const int force = path_opts().lynx_pw1p;
    if (force >= 0) return force != 0 ? 1 : 0;
    static int cus = 0;
    if (!cus) {
        int dev = 0;
        hipDeviceProp_t prop;
        cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
    }
    int best = 0;
    double best_t = 1e30;
    for (int g = 1; g <= mtiles; g *= 2) {
        if (mtiles % g) continue;
        const long rounds = ((long)nft * g + cus - 1) / cus;
        // (one row tile per workgroup = lx_pw1_kernel: 72 us alone, 67 per round when rounds follow each other)
        const double t = g == mtiles ? 5.0 + 67.0 * (double)rounds : (double)rounds * (11.0 + 60.5 * (mtiles / g));
        if (t < best_t * 0.995) { best_t = t; best = g; }       // (ties: fewer, longer workgroups)
    }
    return best == mtiles ? 0 : best;                            // one row tile per workgroup: lx_pw1_kernel
}
static bool lx_use_pw1p(int nft, int mtiles) { return lx_pw1p_groups(nft, mtiles) > 0; }

template <int KT, int RAG>
static hipError_t lx_launch(const LxLayerP& p, int which, int nwg, hipStream_t st) {
    static bool attr1 = false, attr2 = false;
    if (which == 0 && !attr1) {
        hipError_t e = lx_attr(lx_pw1_kernel<KT, RAG>);
        if (e != hipSuccess) return e;
        attr1 = true;
    }
    if (which == 1 && !attr2) {
        hipError_t e = lx_attr(lx_pw2_kernel<KT, RAG>);
        if (e != hipSuccess) return e;
        attr2 = true;
    }
    if (nwg == 0) return hipSuccess;
    const int ldsb = lx_lds_bytes(KT);
    if (which == 0) return launch_timed(lx_pw1_kernel<KT, RAG>, dim3(nwg), dim3(256), ldsb, st, p, "lx_pw1_kernel<%d, %d>", KT, RAG);
    return launch_timed(lx_pw2_kernel<KT, RAG>, dim3(nwg), dim3(256), ldsb, st, p, "lx_pw2_kernel<%d, %d>", KT, RAG);
}

template <int NP, int RAG>
static hipError_t lx_launch_pw2d(const LxLayerP& p, int nwg, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        hipError_t e = lx_attr(lx_pw2d_kernel<NP, RAG>);
        if (e != hipSuccess) return e;
        attr = true;
    }
    if (nwg == 0) return hipSuccess;
    return launch_timed(lx_pw2d_kernel<NP, RAG>, dim3(nwg), dim3(256), (2 * 512 * 32 + 1024) * 4, st, p, "lx_pw2d_kernel<%d, %d>", NP, RAG);
}

// which = 0: pw1 (LayerNorm -> C -> 2 inner -> SwiGLU);  1: pw2 (inner -> C + residual + next-layer transition)
hipError_t launch_lx_layer(const LxLayerP& p, int which, int C, hipStream_t st) {
    const int nft = p.cgmap ? p.ncg : p.nft;
    const int mtiles = which == 0 ? (2 * p.inner) / 512 : C / 512;
    const int nwg = nft * mtiles;
    if (which == 0 && (nft == 0 || lx_use_pw1p(nft, mtiles))) {      // (nft == 0: attribute set-up of both forms at create)
        LxLayerP q = p;
        q.rt_groups = nft == 0 ? 1 : lx_pw1p_groups(nft, mtiles);
        const int nwg1 = nft * q.rt_groups;
        hipError_t e = C == 1024 ? (q.cgmap ? lx_launch_pw1p<1024, 1>(q, nwg1, st) : lx_launch_pw1p<1024, 0>(q, nwg1, st))
                     : C == 512 ? (q.cgmap ? lx_launch_pw1p<512, 1>(q, nwg1, st) : lx_launch_pw1p<512, 0>(q, nwg1, st))
                                : hipErrorInvalidValue;
        if (nft != 0 || e != hipSuccess) return e;
    }
    // pw2 with double-buffered 512-channel half-phases (inner = 1024 or 2048); DSD_LYNX_PW2D=0: the two-phase form
    const int pw2d = path_opts().lynx_pw2d;
    if (which == 1 && pw2d != 0 && (p.inner == 2048 || p.inner == 1024) && (C == 1024 || C == 512)) {
        hipError_t e = p.inner == 2048 ? (p.cgmap ? lx_launch_pw2d<4, 1>(p, nwg, st) : lx_launch_pw2d<4, 0>(p, nwg, st))
                                       : (p.cgmap ? lx_launch_pw2d<2, 1>(p, nwg, st) : lx_launch_pw2d<2, 0>(p, nwg, st));
        if (nwg != 0 || e != hipSuccess) return e;
    }
    if (C == 1024) return p.cgmap ? lx_launch<1024, 1>(p, which, nwg, st) : lx_launch<1024, 0>(p, which, nwg, st);
    if (C == 512) return p.cgmap ? lx_launch<512, 1>(p, which, nwg, st) : lx_launch<512, 0>(p, which, nwg, st);
    return hipErrorInvalidValue;
}

// true: launch_lx_layer(p, 0, ...) will take lx_pw1p_kernel, which merges the LayerNorm partials itself (p.lnpart_in): the
// caller skips the ln_merge launch for this layer
bool lx_pw1_merges_stats(const LxLayerP& p, int C) {
    const int nft = p.cgmap ? p.ncg : p.nft;
    if (DSD_LX_PW1_MERGE) return C == 512 || C == 1024;          // both forms of pw1 merge the partials in their prologue
    return nft > 0 && (C == 512 || C == 1024) && lx_use_pw1p(nft, (2 * p.inner) / 512);
}

hipError_t lx_layer_init_all() {
    LxLayerP p{};
    hipError_t e;
    for (int C : {512, 1024})
        for (int rag = 0; rag < 2; ++rag) {
            p.inner = 2 * C;
            p.cgmap = rag ? reinterpret_cast<const int*>(&p) : nullptr;
            p.ncg = 0;
            p.nft = 0;
            if ((e = launch_lx_layer(p, 0, C, nullptr)) != hipSuccess) return e;
            if ((e = launch_lx_layer(p, 1, C, nullptr)) != hipSuccess) return e;
            if ((e = launch_lx_pw2q(p, C, nullptr)) != hipSuccess) return e;
        }
    return hipSuccess;
}

}  // namespace dsd
