// The WaveNet's three small GEMMs AROUND the residual layers, per evaluation (wavenet.py:96-107, :86-88 of the next one):
//
//   h   = relu(W1 (skip / sqrt(L)) + b1)              skip_projection        (C x C)
//   eps = W2 h + b2                                   output_projection      (F*M x C)
//   x'  = the solver's linear combinations of eps and the state buffers (as gemm.hip's EP_LINCOMB: <= 3 outputs x 8 terms)
//   x0  = relu(W3 x' + b3)                            the NEXT evaluation's input_projection (C x F*M), when its input is
//                                                     one of the x' just formed
//
// as ONE launch with one workgroup per FRAME tile and every row in it, the three products chained through LDS.  As three
// launches of gemm.hip (64-row tiles: four workgroups re-stage each frame tile, 4-byte epilogue stores, three boundaries)
// they took 23 + 21 + 14.5 us per evaluation at B = 8 - 4.3 % of the loop for 1.3 % of its FLOPs - and 4.9 + 4.9 + 6.9 us at
// B = 1.  Frame tile = 32 frames on batched grids, 16 on one-utterance grids (63 workgroups at T = 1000: the chain keeps a
// tile's rows together, so this kernel trades chip width for two launches less).  Four waves; a wave owns whole 16-row
// blocks of each product; weights stream as 1 KiB fragment blocks through a three-set register rotation.
#include <hip/hip_ext.h>

#include "dsd_internal.h"

namespace dsd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifdef DSD_STAMPS
// [launch with / without a fused input projection][workgroup][0..9]: s_memtime at the phase boundaries (wave 0)
__device__ unsigned long long g_edge_stamps[2][4096][16];
#define EDGE_STAMP(i)                                                                          \
    do {                                                                                       \
        if (threadIdx.x == 0 && blockIdx.x < 4096) {                                           \
            __builtin_amdgcn_sched_barrier(0);                                                 \
            g_edge_stamps[p.next_src >= 0 ? 1 : 0][blockIdx.x][i] = __builtin_amdgcn_s_memtime(); \
            __builtin_amdgcn_sched_barrier(0);                                                 \
        }                                                                                      \
    } while (0)
extern "C" int dsd_dbg_read_edge_stamps(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_edge_stamps), sizeof(g_edge_stamps));
}
#else
#define EDGE_STAMP(i)
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
__device__ __forceinline__ void st4(f32x4 v, dsd_i32x4 r, int voff, int soff) {      // write-through: dsd_internal.h
    dsd_store_b128<DSD_ST_AUX>(__builtin_bit_cast(dsd_u32x4, v), r, voff, soff);
}

// acc[k][n] += W[block k][:, K] * tile[K, column block n]: NS k16 steps, the wave's MB row blocks (1 KiB fragment blocks at
// wk[k] + 1024 * step), the B fragments from an LDS tile of row stride 16 * NCB floats (NCB = 2: odd rows are stored with
// their 16-column halves swapped, as in lynx_layer.hip - the lane's swizzle is folded into zt[n]).  The weight stream runs
// D - 1 steps ahead through a D-deep register rotation: a step is only 4 MB NCB MFMAs (128 cycles each), and with one wave
// per SIMD nothing else covers the L2 latency - edge_depth() sizes D for >= ~2.5 k cycles of lookahead (a two-step
// rotation left the first version of this kernel at a third of its MFMA rate).  W[0 .. D-2] hold steps 0 .. D-2 on entry
// (edge_prefetch: issued under the previous product's epilogue).
constexpr int edge_depth(int mb, int ncb, int ns) {
    const int d = 1 + (2560 + mb * ncb * 128 - 1) / (mb * ncb * 128);
    return d < 3 ? 3 : (d > ns ? ns : d);
}
template <int MB, int D>
__device__ __forceinline__ void edge_prefetch(f32x4 (&W)[D][MB], const __amdgpu_buffer_rsrc_t r_w, const int (&wk)[MB]) {
#pragma unroll
    for (int s = 0; s < D - 1; ++s)
#pragma unroll
        for (int k = 0; k < MB; ++k) W[s][k] = ld4(r_w, wk[k] + (s & 3) * 1024, (s >> 2) * 4096);
}
template <int MB, int NS, int NCB, int D>
__device__ __forceinline__ void edge_walk(f32x4 (&acc)[MB][NCB], f32x4 (&W)[D][MB], const __amdgpu_buffer_rsrc_t r_w,
                                          const int (&wk)[MB], const float* (&zt)[NCB]) {
    constexpr int RS = 16 * NCB;
    float bq[2][4][NCB];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int n = 0; n < NCB; ++n) bq[0][j][n] = zt[n][(j * 4) * RS];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        if (s + D - 1 < NS) {
#pragma unroll
            for (int k = 0; k < MB; ++k)
                W[(s + D - 1) % D][k] = ld4(r_w, wk[k] + ((s + D - 1) & 3) * 1024, ((s + D - 1) >> 2) * 4096);
        }
        if (s + 1 < NS) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int n = 0; n < NCB; ++n) bq[(s + 1) & 1][j][n] = zt[n][((s + 1) * 16 + j * 4) * RS];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int k = 0; k < MB; ++k)
#pragma unroll
                for (int n = 0; n < NCB; ++n)
                    acc[k][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(W[s % D][k][j], bq[s & 1][j][n], acc[k][n], 0, 0, 0);
        // the step's schedule, pinned (left alone the compiler sinks every weight load to just before its use and waits for
        // it there: the first build of this kernel ran at a third of its MFMA rate with vmcnt(0) in every step): behind each
        // 4 NCB MFMAs one weight load of step s + D - 1 and this group's share of the next step's LDS reads
#pragma unroll
        for (int k = 0; k < MB; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 4 * NCB, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, (4 * NCB + MB - 1) / MB, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
}

}  // namespace

// NCH = C / 64 (2, 3, 4); FMB = 16-row blocks of the F*M output rows (8: 128 mel bins, 5: 80 mel bins, 4: 64 pitch bins,
// 3: 2 x 24 variance bins); NCB = 16-frame column blocks per tile (2: batched grids, 1: one-utterance grids); RAG: ragged batch (tile list)
template <int NCH, int FMB, int NCB, int RAG>
__global__ __launch_bounds__(256, 1) void wn_edge_kernel(const WnEdgeP p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int C = 64 * NCH;
    constexpr int BNW = 16 * NCB;                    // frames per tile = row stride of the operand tiles
    constexpr int PS = BNW + 4;                      // row stride of the row-major tiles (solver sums, last epilogue)
    constexpr int MB1 = NCH;                         // 16-row blocks per wave of the two C-row products (4 NCH blocks / 4 waves)
    constexpr int MB2 = (FMB + 3) / 4;               // ... of the F*M-row product
    constexpr int K3 = FMB * 16;                     // padded F*M: K extent of the input projection
    constexpr int W4 = BNW / 4;                      // float4 per row
    constexpr int NU = C * W4 / 256;                 // staged float4 per thread of the skip tile
    constexpr int NP = (K3 * W4 + 255) / 256;        // ... of an F*M-row tile
    float* sT = lds;                                 // [C][BNW]: skip / sqrt(L); later [K3][BNW]: the next evaluation's input
    float* hT = lds + C * BNW;                       // [C][BNW]: relu(W1 s + b1); later the last epilogue's [C][PS] (runs on
    float* pT = hT + C * BNW;                        //   into pT) ; pT [kMaxOut][K3][PS]: the solver sums without the model term

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lrow = lane >> 4, lcol = lane & 15, rq = lrow * 4;
    const int ft = blockIdx.x;
    const int rest = RAG ? p.cgmap[ft] : ft;
    const int b = fdiv_floor(rest, p.inv_tiles_per_b);
    const int t0 = (rest - b * p.tiles_per_b) * BNW;
    const int Ts = p.Ts;
    const int bu = __builtin_amdgcn_readfirstlane(b), t0u = __builtin_amdgcn_readfirstlane(t0);
    auto swz = [](int row) { return NCB == 2 ? ((row & 1) << 4) : 0; };
    const int wl = lane * 16;
    EDGE_STAMP(0);

    // ---------------- prologue loads: the skip tile, the first product's first weight steps, the solver's state terms ----------------
    const __amdgpu_buffer_rsrc_t r_s = rsrc(p.skip + (long)bu * p.x_bstride + t0u);
    const int c4s = tid % W4, r0s = tid / W4;
    f32x4 sv[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) sv[u] = ld4(r_s, ((r0s + u * (256 / W4)) * Ts + c4s * 4) * 4, 0);
    constexpr int NS1 = C / 16;
    const __amdgpu_buffer_rsrc_t r_w1 = rsrc(p.A1);
    int wk1[MB1];
#pragma unroll
    for (int k = 0; k < MB1; ++k) wk1[k] = wl + (wave * MB1 + k) * NS1 * 1024;
    constexpr int NS3 = K3 / 16;
    constexpr int D1 = edge_depth(MB1, NCB, NS1), D2 = edge_depth(MB2, NCB, NS1), D3 = edge_depth(MB1, NCB, NS3);
    f32x4 W1[D1][MB1];
    edge_prefetch<MB1, D1>(W1, r_w1, wk1);
    // The solver's linear combinations (gemm.hip EP_LINCOMB) WITHOUT their model term: sums over the state buffers / noise
    // tensors as they are BEFORE this evaluation (a destination may be another output's source), formed here from whole-line
    // 16-byte loads that fly under the first product, kept row-major in LDS; eps joins after the second product.
    // (every descriptor the kernel will need is copied out of the argument block in ONE batch of scalar loads here: fetched
    // where they are used - inside uniform branches - each cost its own round trip; the stamps showed 9 k cycles of prologue)
    EdgeTerm tq[kEdgeMaxTerms];
#pragma unroll
    for (int q = 0; q < kEdgeMaxTerms; ++q) tq[q] = p.q[q];
    float cmv[kMaxOut];
    float* dstv[kMaxOut];
#pragma unroll
    for (int o = 0; o < kMaxOut; ++o) {
        cmv[o] = p.cm[o];
        dstv[o] = p.dst[o];
    }
    const int nq = p.nq, nout = p.nout, next_src = p.next_src;
    // (pinned: an empty asm that takes the values in SGPRs here - otherwise the compiler sinks each scalar load back to its
    // use inside the branch and waits for it there, eight round trips in a row)
#pragma unroll
    for (int q = 0; q < kEdgeMaxTerms; ++q)
        asm volatile("" ::"s"(tq[q].ptr), "s"(tq[q].bstride), "s"(tq[q].rstride), "s"(tq[q].coef), "s"(tq[q].out));
#pragma unroll
    for (int o = 0; o < kMaxOut; ++o) asm volatile("" ::"s"(cmv[o]), "s"(dstv[o]));
    asm volatile("" ::"s"(nq), "s"(nout), "s"(next_src));
    // the output projection's bias in the accumulator layout (F*M is a whole number of 16-row blocks)
    f32x4 b2v[MB2];
    {
        const __amdgpu_buffer_rsrc_t r_b2 = rsrc(p.b2);
#pragma unroll
        for (int k = 0; k < MB2; ++k) b2v[k] = ld4(r_b2, (min((wave * MB2 + k) * 16, p.FM - 16) + rq) * 4, 0);
    }
    {
        f32x4 tv[kEdgeMaxTerms][NP];
#pragma unroll
        for (int q = 0; q < kEdgeMaxTerms; ++q) {
            if (q < nq) {                                        // workgroup-uniform
                const EdgeTerm tm = tq[q];
#pragma unroll
                for (int u = 0; u < NP; ++u) {
                    const int idx = tid + 256 * u;
                    const int row = min(idx / W4, p.FM - 1), cc = (idx % W4) * 4;
                    // (state buffers in the internal layout only: whole 16-byte pieces of padded rows; a program with
                    // caller-noise terms - ancestral DDPM - keeps the three GEMMs, the host decides)
                    tv[q][u] = *reinterpret_cast<const f32x4*>(tm.ptr + (long)b * tm.bstride + (long)row * tm.rstride + t0 + cc);
                }
            }
        }
        f32x4 ps[kMaxOut][NP];
#pragma unroll
        for (int o = 0; o < kMaxOut; ++o)
#pragma unroll
            for (int u = 0; u < NP; ++u) ps[o][u] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int q = 0; q < kEdgeMaxTerms; ++q) {
            if (q < nq) {
                const float cf = tq[q].coef;
                const int o = tq[q].out;
#pragma unroll
                for (int oo = 0; oo < kMaxOut; ++oo)
                    if (oo == o) {
#pragma unroll
                        for (int u = 0; u < NP; ++u) ps[oo][u] += cf * tv[q][u];
                    }
            }
        }
#pragma unroll
        for (int o = 0; o < kMaxOut; ++o) {
            if (o < nout) {
#pragma unroll
                for (int u = 0; u < NP; ++u) {
                    const int idx = tid + 256 * u;
                    if (idx < K3 * W4) *reinterpret_cast<f32x4*>(&pT[(o * K3 + idx / W4) * PS + (idx % W4) * 4]) = ps[o][u];
                }
            }
        }
    }
    EDGE_STAMP(1);
    // skip / sqrt(L) (wavenet.py:96; the division of gemm.hip's ST_SCALE stage) -> LDS
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        const int row = r0s + u * (256 / W4);
        f32x4 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) o[e] = sv[u][e] / p.in_scale;
        *reinterpret_cast<f32x4*>(&sT[row * BNW + ((c4s * 4) ^ swz(row))]) = o;
    }
    __syncthreads();
    EDGE_STAMP(2);
    const int sw = swz(lrow);
    const float* zs[NCB];
    const float* zh[NCB];
#pragma unroll
    for (int n = 0; n < NCB; ++n) {
        zs[n] = sT + lrow * BNW + ((n * 16 + lcol) ^ sw);
        zh[n] = hT + lrow * BNW + ((n * 16 + lcol) ^ sw);
    }

    // the second product's weight stream (its first steps are fetched under the first product's epilogue)
    const bool act2 = wave * MB2 < FMB;                          // wave-uniform: this wave owns F*M row blocks
    const __amdgpu_buffer_rsrc_t r_w2 = rsrc(p.A2);
    int wk2[MB2];
#pragma unroll
    for (int k = 0; k < MB2; ++k) wk2[k] = wl + min(wave * MB2 + k, FMB - 1) * NS1 * 1024;      // (FMB = 5: the last wave's second block does not exist)
    f32x4 W2[D2][MB2];

    // ---------------- h = relu(W1 s + b1) -> LDS ----------------
    {
        f32x4 acc[MB1][NCB];
#pragma unroll
        for (int k = 0; k < MB1; ++k)
#pragma unroll
            for (int n = 0; n < NCB; ++n) acc[k][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        const __amdgpu_buffer_rsrc_t r_b = rsrc(p.b1);
        f32x4 bo[MB1];
#pragma unroll
        for (int k = 0; k < MB1; ++k) bo[k] = ld4(r_b, ((wave * MB1 + k) * 16 + rq) * 4, 0);
        edge_walk<MB1, NS1, NCB, D1>(acc, W1, r_w1, wk1, zs);
        EDGE_STAMP(3);
        if (act2) edge_prefetch<MB2, D2>(W2, r_w2, wk2);
#pragma unroll
        for (int k = 0; k < MB1; ++k)
#pragma unroll
            for (int n = 0; n < NCB; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = (wave * MB1 + k) * 16 + rq + r;
                    hT[row * BNW + ((n * 16 + lcol) ^ swz(row))] = fmaxf(acc[k][n][r] + bo[k][r], 0.f);
                }
    }
    __syncthreads();                                             // h complete; the skip tile is dead
    EDGE_STAMP(4);

    // the third product's weight stream (first steps fetched under the solver update)
    const __amdgpu_buffer_rsrc_t r_w3 = rsrc(p.A3);
    int wk3[MB1];
#pragma unroll
    for (int k = 0; k < MB1; ++k) wk3[k] = wl + (wave * MB1 + k) * NS3 * 1024;
    f32x4 W3[D3][MB1];

    // ---------------- eps = W2 h + b2; outputs = LDS sums + c_model eps; x' -> LDS; 16-byte row stores ----------------
    if (act2) {
        f32x4 acc[MB2][NCB];
#pragma unroll
        for (int k = 0; k < MB2; ++k)
#pragma unroll
            for (int n = 0; n < NCB; ++n) acc[k][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        edge_walk<MB2, NS1, NCB, D2>(acc, W2, r_w2, wk2, zh);
        EDGE_STAMP(5);
        if (next_src >= 0) edge_prefetch<MB1, D3>(W3, r_w3, wk3);
#pragma unroll
        for (int o = 0; o < kMaxOut; ++o) {
            if (o < nout) {
                const float cm = cmv[o];
                float* po = pT + o * K3 * PS;
#pragma unroll
                for (int k = 0; k < MB2; ++k) {
                    if (FMB % MB2 != 0 && wave * MB2 + k >= FMB) continue;      // (wave-uniform; a row block past F*M has no LDS rows)
#pragma unroll
                    for (int n = 0; n < NCB; ++n)
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = (wave * MB2 + k) * 16 + rq + r;
                            const float ev = acc[k][n][r] + b2v[k][r];
                            float* q = &po[row * PS + n * 16 + lcol];
                            const float v = *q + cm * ev;
                            *q = v;
                            if (o == next_src) sT[row * BNW + ((n * 16 + lcol) ^ swz(row))] = v;
                        }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        constexpr int NE2 = MB2 * 16 * W4 / 64;                  // float4 per lane over the wave's F*M rows
#pragma unroll
        for (int o = 0; o < kMaxOut; ++o) {
            if (o < nout) {
                const float* po = pT + (o * K3 + wave * MB2 * 16) * PS;
                const dsd_i32x4 w_d = dsd_rsrc_words(dstv[o] + (long)bu * p.o_bstride + (long)(wave * MB2 * 16) * p.o_rstride + t0u);
#pragma unroll
                for (int m = 0; m < NE2; ++m) {
                    const int idx = lane + 64 * m;
                    const int row = idx / W4, cc = (idx % W4) * 4;
                    if (wave * MB2 * 16 + row < p.FM)
                        st4(*reinterpret_cast<const f32x4*>(&po[row * PS + cc]), w_d, (row * p.o_rstride + cc) * 4, 0);
                }
            }
        }
    } else if (next_src >= 0) {
        edge_prefetch<MB1, D3>(W3, r_w3, wk3);
    }
    EDGE_STAMP(6);
    if (next_src < 0) return;                                  // (workgroup-uniform) no input projection to fuse
    __syncthreads();
    EDGE_STAMP(7);                                             // x' complete; every wave is past the h tile and the LDS sums

    // ---------------- x0 = relu(W3 x' + b3): the next evaluation's layer-0 input ----------------
    {
        f32x4 acc[MB1][NCB];
#pragma unroll
        for (int k = 0; k < MB1; ++k)
#pragma unroll
            for (int n = 0; n < NCB; ++n) acc[k][n] = f32x4{0.f, 0.f, 0.f, 0.f};
        const __amdgpu_buffer_rsrc_t r_b = rsrc(p.b3);
        f32x4 bo[MB1];
#pragma unroll
        for (int k = 0; k < MB1; ++k) bo[k] = ld4(r_b, ((wave * MB1 + k) * 16 + rq) * 4, 0);
        edge_walk<MB1, NS3, NCB, D3>(acc, W3, r_w3, wk3, zs);
        EDGE_STAMP(8);
        // accumulators -> the wave's own rows of a row-major tile over the dead h tile -> 16-byte stores of whole row pieces
        float* ew = hT + wave * (MB1 * 16) * PS;
#pragma unroll
        for (int k = 0; k < MB1; ++k)
#pragma unroll
            for (int n = 0; n < NCB; ++n)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    ew[(k * 16 + rq + r) * PS + n * 16 + lcol] = fmaxf(acc[k][n][r] + bo[k][r], 0.f);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const dsd_i32x4 w_xo = dsd_rsrc_words(p.xh + (long)bu * p.x_bstride + (long)(wave * MB1 * 16) * Ts + t0u);
        constexpr int NE = MB1 * 16 * W4 / 64;                   // float4 per lane over the wave's rows
#pragma unroll
        for (int m = 0; m < NE; ++m) {
            const int idx = lane + 64 * m;
            const int row = idx / W4, cc = (idx % W4) * 4;
            st4(*reinterpret_cast<const f32x4*>(&ew[row * PS + cc]), w_xo, (row * Ts + cc) * 4, 0);
        }
    }
    EDGE_STAMP(9);
}

int wn_edge_lds_bytes(int C, int fmb, int ncb) {
    const int bnw = 16 * ncb, ps = bnw + 4;
    const int sums = kMaxOut * fmb * 16 * ps, ep = C * ps;       // behind the two operand tiles: the solver sums; the last epilogue's
    return (2 * C * bnw + sums > C * bnw + ep ? 2 * C * bnw + sums : C * bnw + ep) * 4;      // rows run from the h tile into them
}

bool wn_edge_supported(int C, int FM) { return (C == 256 || C == 192 || C == 128) && (FM == 128 || FM == 80 || FM == 64 || FM == 48); }

template <int NCH, int FMB, int NCB, int RAG>
static hipError_t edge_launch(const WnEdgeP& p, int nwg, hipStream_t st) {
    static bool attr = false;
    if (!attr) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(wn_edge_kernel<NCH, FMB, NCB, RAG>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr = true;
    }
    if (nwg == 0) return hipSuccess;
    return launch_timed(wn_edge_kernel<NCH, FMB, NCB, RAG>, dim3(nwg), dim3(256), wn_edge_lds_bytes(64 * NCH, FMB, NCB), st, p,
                        "wn_edge_kernel<%d, %d, %d, %d>", NCH, FMB, NCB, RAG);
}

// (the kernel also instantiates with 16-frame tiles, NCB = 1; measured at B = 1, T = 1000 - 63 workgroups - it takes 18.6 us
// against 16.7 us for the three launches: the 512 KB of weights per workgroup then exceed what a CU's memory pipe moves in
// the tile's MFMA time.  One-utterance grids keep the three GEMMs.)
template <int NCH, int FMB>
static hipError_t edge_dispatch(const WnEdgeP& p, int ncb, int nwg, hipStream_t st) {
    if (ncb != 2) return hipErrorInvalidValue;
    return p.cgmap ? edge_launch<NCH, FMB, 2, 1>(p, nwg, st) : edge_launch<NCH, FMB, 2, 0>(p, nwg, st);
}

// ncb = 16-frame column blocks per tile (1 or 2); nwg = frame tiles (p.ncg for a ragged batch)
hipError_t launch_wn_edge(const WnEdgeP& p, int C, int ncb, int nwg, hipStream_t st) {
    if (C == 256 && p.FM == 128) return edge_dispatch<4, 8>(p, ncb, nwg, st);
    if (C == 256 && p.FM == 64) return edge_dispatch<4, 4>(p, ncb, nwg, st);
    if (C == 192 && p.FM == 48) return edge_dispatch<3, 3>(p, ncb, nwg, st);
    if (C == 256 && p.FM == 48) return edge_dispatch<4, 3>(p, ncb, nwg, st);
    if (C == 192 && p.FM == 128) return edge_dispatch<3, 8>(p, ncb, nwg, st);
    if (C == 192 && p.FM == 64) return edge_dispatch<3, 4>(p, ncb, nwg, st);
    if (C == 256 && p.FM == 80) return edge_dispatch<4, 5>(p, ncb, nwg, st);
    if (C == 192 && p.FM == 80) return edge_dispatch<3, 5>(p, ncb, nwg, st);
    if (C == 128 && p.FM == 128) return edge_dispatch<2, 8>(p, ncb, nwg, st);
    if (C == 128 && p.FM == 80) return edge_dispatch<2, 5>(p, ncb, nwg, st);
    if (C == 128 && p.FM == 64) return edge_dispatch<2, 4>(p, ncb, nwg, st);
    if (C == 128 && p.FM == 48) return edge_dispatch<2, 3>(p, ncb, nwg, st);
    return hipErrorInvalidValue;
}

hipError_t wn_edge_init_all() {
    WnEdgeP p{};
    hipError_t e;
    for (int C : {256, 192, 128})
        for (int fm : {128, 80, 64, 48})
            for (int ncb = 2; ncb <= 2; ++ncb)
                for (int rag = 0; rag < 2; ++rag) {
                    p.FM = fm;
                    p.cgmap = rag ? reinterpret_cast<const int*>(&p) : nullptr;
                    if ((e = launch_wn_edge(p, C, ncb, 0, nullptr)) != hipSuccess) return e;
                }
    return hipSuccess;
}

}  // namespace dsd
