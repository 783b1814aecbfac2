// Internal declarations shared by the HIP translation units of libdsdenoise (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

namespace dsd {

// ---------------------------------------------------------------------------------------------
// Activation layout in HBM ("internal layout"): [batch][channel][Ts] fp32, time innermost,
// Ts = round_up(T, 64) + 32 floats (a multiple of 32 floats = 128 B that is never a multiple of
// 4 KiB, so the rows of a channel x time tile rotate over the memory channels).  Frames
// t in [T, Ts) are padding: kernels may write garbage there and every consumer masks on t < T
// when it stages a tile.  Every buffer lives in one arena with a 256-float guard in front and
// behind, so the halo over-reads of the first/last tile land in allocated memory (and are masked).
// ---------------------------------------------------------------------------------------------
static inline int round_up(int v, int m) { return (v + m - 1) / m * m; }
static inline int padded_ts(int T) { return round_up(T, 64) + 32; }

enum Stage { ST_PLAIN = 0, ST_FILM = 1, ST_LN = 2, ST_SCALE = 3, ST_LRELU = 4 };
enum Epi { EP_BIAS_ACT = 0, EP_GATE = 1, EP_RESSKIP = 2, EP_LINCOMB = 3, EP_SWIGLU = 4, EP_BIAS_RES = 5, EP_SCATTER = 6, EP_LYNX_NEXT = 7 };
enum Act { ACT_NONE = 0, ACT_RELU = 1, ACT_MISH = 2, ACT_GELU = 3, ACT_LRELU = 4, ACT_TANH = 5, ACT_SILU = 6 };

// lx_pw1_kernel (LYNXNet pw1, one workgroup per (frame tile, row tile)) merges the LayerNorm partials in its prologue like
// lx_pw1p_kernel does (1) or reads ln_merge_kernel's statistics (0: A/B builds)
#ifndef DSD_LX_PW1_MERGE
#define DSD_LX_PW1_MERGE 1
#endif

// Cache policy of the WaveNet layer kernels' 16-byte result stores (raw buffer intrinsics: bit 4 = sc1 = write-through).
// A kernel boundary costs the bytes its predecessor left dirty in the L2s / ~6 TB/s (MI355X_MICROARCH.md, "boundary":
// x + skip of a fused layer at B = 8 are 16 MB); write-through stores spread that over the kernel's own epilogues.
// A/B on fresh boxes (tools/ab_bench.sh; -DDSD_ST_AUX=0 builds the plain form): 50-NFE loop 16.68 -> 16.55 ms at B = 1,
// 26.21 -> 25.79 at B = 2, 70.07 -> 69.00 at B = 8, the variance pair 37.54 -> 36.92; LYNXNet's kernels keep plain stores.
#ifndef DSD_ST_AUX
#define DSD_ST_AUX 16
#endif

// Path switches (diagnostics, A/B runs, tests).  Every entry point of the C ABI that launches kernels re-reads them from the
// environment ONCE (refresh_path_opts), so one process can drive either side of a switch through consecutive calls - this is
// how tests/test_gpu_fused.py puts every instantiation of the layer kernels under oracle parity - and the values are part of
// the hipGraph cache key (a captured graph is the launch sequence of ONE set of choices).  -1 = unset: the library's own rule.
struct PathOpts {
    int fused_layer;        // DSD_FUSED_LAYER     0: never wn_layer.hip, 1: on every supported grid
    int fused16;            // DSD_FUSED16         0: never wn_layer16_kernel (16-frame tiles of the fused layer), 1: every layer on it
    int wn_plan;            // DSD_WN_PLAN         0: one launch shape per layer (round 2), 1/unset: mixed plans (wn_plan_for)
    int rowsplit;           // DSD_ROWSPLIT        0: never wn_rowsplit.hip
    int rs_bn48;            // DSD_RS_BN48         0: no 48-frame tiles of the row-split pair
    int rs_conv_q;          // DSD_RS_CONV_Q       0 / 1: K-half / K-quarter layout of the row-split conv
    int rs_rows;            // DSD_RS_ROWS         64 / 128 / 256: rows per workgroup of the row-split pair
    int rs_rows_out;        // DSD_RS_ROWS_OUT     128 / 256: ... of its out-proj launch alone (diagnostic: the two launches are independent)
    int edge;               // DSD_EDGE            0: never wn_edge.hip, 1: on every grid
    int lynx_resident;      // DSD_LYNX_RESIDENT   0: never lynx_layer.hip, 1: on every supported grid
    int lynx_pw1p;          // DSD_LYNX_PW1P
    int lynx_pw2d;          // DSD_LYNX_PW2D
    int lynx_pw2q;          // DSD_LYNX_PW2Q       0: never the 128-row pw2 of one-utterance grids (gemm.hip instead), 1: on every grid
    int narrow;             // DSD_NARROW          gemm.hip: 16-frame tiles off / on
    int gm_shift;           // DSD_GM_SHIFT        gemm.hip: L2 blocking of the work order
    int film_t;             // DSD_FILM_T          0: FiLM vectors from D [L*C][Ns] instead of the transposed table
    int dwconv_rows;        // DSD_DWCONV_ROWS     0: the first depthwise-convolution kernel
    int precision;          // DSD_PRECISION       1: split-bf16 (bf16x3) layer kernels where they exist (opt-in, own tolerance)
    int x3_wide;            // DSD_X3_WIDE         0: never 64-frame tiles in the split-bf16 LYNXNet kernels, 1: wherever they exist
    long nb2_min;           // DSD_NB2_MIN_WG      gemm.hip: workgroups from which 64-frame tiles are used (default 512)
};
const PathOpts& path_opts();
void refresh_path_opts();

// Timing hook of bench.py (dsd_kernel_timing): api.hip arms the slot with a start / stop event pair before a launch it wants
// timed; the launcher that finds it armed goes through hipExtLaunchKernelGGL, which ties the two events to the dispatch
// packet itself (their elapsed time is the kernel's own begin -> end time, what a rocprofv3 kernel trace reports, not a
// bracket around the launch), records WHICH instantiation ran - kernel name + template arguments as rocprofv3 prints them,
// e.g. "wn_layer_kernel<4, 48, 0>" - and disarms it (one launch per arming).
struct TimingSlot {
    hipEvent_t e0 = nullptr, e1 = nullptr;
    bool taken = false;
    char name[96] = {0};
};
TimingSlot& timing_slot();      // thread-local (api.hip)

template <typename K, typename P, typename... NameArgs>
inline hipError_t launch_timed(K kern, dim3 grid, dim3 block, int lds, hipStream_t st, const P& p, const char* name_fmt,
                               NameArgs... name_args) {
    TimingSlot& ts = timing_slot();
    if (ts.e0 && ts.e1 && !ts.taken) {
        hipExtLaunchKernelGGL(kern, grid, block, lds, st, ts.e0, ts.e1, 0, p);
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Wformat-security"
        snprintf(ts.name, sizeof(ts.name), name_fmt, name_args...);      // (every caller passes a literal)
#pragma clang diagnostic pop
        ts.taken = true;
    } else {
        hipLaunchKernelGGL(kern, grid, block, lds, st, p);
    }
    return hipGetLastError();
}

// 16-byte raw buffer store followed by two wait states, as ONE inline-asm statement.  A store of more than 8 bytes reads its
// data registers over several cycles after issue, and a VALU write to one of them in the next issue slot can land first.
// hipcc (ROCm 7.2) inserts the required wait state for the immediate-soffset form but not when soffset is a register - LLVM's
// hazard model calls that form safe - and on gfx950 it is not: wn_out_rw_kernel<4, *> stored, nondeterministically and in
// ~0.4 % of the elements, the NEXT item's operand as the first element of a vector (found with tools/harness/
// rows_harness.hip; tools/check_store_hazard.py scans the ISA of every kernel file for the pattern, tests/
// test_kernel_resources.py runs it).  A separate `s_nop` (builtin or asm) behind the builtin store does not stay there -
// neither scheduling barriers nor a memory clobber kept the post-RA scheduler from moving VALU instructions in between -
// so the store itself is asm.  `rsrc` = the four descriptor words (dsd_rsrc_words), wave-uniform.
typedef unsigned dsd_u32x4 __attribute__((ext_vector_type(4)));
typedef int dsd_i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ dsd_i32x4 dsd_rsrc_words(const void* ptr) {
    // (readfirstlane: the words must be in SGPRs for the asm's "s" operand - when the compiler cannot prove the pointer
    // wave-uniform, or has spilled it to a VGPR, it would otherwise print a VGPR range into the descriptor slot)
    const unsigned long long a = (unsigned long long)ptr;
    return dsd_i32x4{__builtin_amdgcn_readfirstlane((int)(unsigned)a), __builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xffffu)),
                     (int)0x7FFFFFF0u, 0x00020000};
}
template <int AUX>
__device__ __forceinline__ void dsd_store_b128(dsd_u32x4 data, dsd_i32x4 rsrc, int voff, int soff) {
    static_assert(AUX == 0 || AUX == 16, "plain or sc1 (write-through)");
    if (AUX == 16)
        asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen sc1\n\ts_nop 1" ::"v"(data), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
    else
        asm volatile("buffer_store_dwordx4 %0, %1, %2, %3 offen\n\ts_nop 1" ::"v"(data), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}

constexpr int kMaxTerms = 8;
constexpr int kMaxOut = 3;

struct LinTerm {
    const float* ptr;   // nullptr: the model output of this launch
    long bstride;       // floats between batch items
    int rstride;        // floats between channel rows
    int ext;            // 1: caller tensor with row stride T (mask reads on t < T)
    float coef;
    int pad_;
};
struct LinOut {
    float* dst;         // internal layout
    int nterms;
    int pad_;
    LinTerm t[kMaxTerms];
};

// One GEMM-shaped launch:  out[m, t] = epilogue( sum_{tap, c} A[m, tap, c] * stage(B)[c, t + (tap-1)*dil] )
struct GemmP {
    // A: weights pre-packed in MFMA 16x16x4 fragment order, [mblk][64-ch chunk][tap][k16 in chunk][lane][4]
    const float* A;
    const float* bias;      // original row indexing, may be nullptr
    int M;                  // real output rows (original indexing)
    int C;                  // EP_GATE / EP_SWIGLU: rows per half; EP_RESSKIP: residual rows
    // B: activations, internal layout
    const float* B;
    long b_bstride;
    int b_rstride;
    int K;                  // input channels padded to a multiple of 16
    int Kreal;
    int KC;                 // channels staged in LDS per chunk (multiple of 16)
    int T;                  // valid frames
    int tiles_per_b;
    int mtiles;             // 64-row tiles (EP_GATE / EP_SWIGLU: 32 pairs each)
    int lds_bytes;          // dynamic LDS of this launch
    float inv_mtiles, inv_tiles_per_b, inv_w4;   // reciprocals for the prologue's index arithmetic
    // L2 blocking of the work order (many row tiles: LYNXNet's 1x1 GEMMs): row tiles are taken in groups of 2^gm_shift, a
    // group walks ALL frame tiles before the next group starts - so the ~100 workgroups an XCD runs at a time share one
    // group's weights (which stay in its 4 MiB L2) and each activation tile is fetched once per group.  0: row tile fastest
    // over all row tiles (every frame tile re-streams the whole weight matrix through L2).
    int gm_shift;
    int per_group;          // frame tiles of the launch << gm_shift
    float inv_per_group;
    int lpr_shift;          // staging: 2^lpr_shift lanes per staged row (>= float4 per row)
    int dil;                // dilation (taps > 1)
    int taps;               // kernel size along time (1, 3, or any odd k on the generic path)
    int HL;                 // halo columns staged on each side (multiple of 4, >= (taps / 2) * dil)
    int S;                  // LDS row stride in floats, S % 32 == 16
    float in_scale;         // ST_SCALE: staged value DIVIDED by this;  ST_LRELU: negative slope of the leaky ReLU
    // ST_FILM: y = x + film[c * film_cstride + film_col0 + b * film_colb]
    const float* film;
    int film_cstride, film_col0, film_colb;
    // ST_LN: y = (x - mean[b, t]) * rstd[b, t]; stats laid out [b][2][ln_ts]
    const float* ln_stats;
    int ln_ts;
    // epilogue
    int act;
    float* out;             // EP_BIAS_ACT / EP_GATE / EP_SWIGLU / EP_BIAS_RES destination
    long o_bstride;
    int o_rstride;
    const float* aux;       // EP_GATE: hoisted conditioner projection (+ biases); EP_BIAS_RES: residual
    long aux_bstride;
    int aux_rstride;
    float* x;               // EP_RESSKIP: residual stream, updated in place
    float* skip;            // EP_RESSKIP: running skip sum
    int first_layer;        // EP_RESSKIP: 1 = skip is written, not accumulated
    int up;                 // EP_SCATTER: upsampling factor u; row r*C + o, column t -> out[o][u*t + r] (C = p.C)
    // EP_LYNX_NEXT (LYNXNet layer transition): v = act(acc + bias) (+ aux);  with the NEXT layer's conditioner
    // projection cpn and step projection (film fields):  strong: x = v + cpn, xin = x + d;  else: x = v, xin = v + cpn + d;
    // cpn == nullptr (after the last layer): x = xin = v.  out = x, out2 = xin (may be nullptr), and per 64-row tile the
    // LayerNorm partials of xin over its rows: lnpart[b][mtile][0][t] = mean, [1][t] = sum of squared deviations.
    // ragged batches: item b is valid on [0, lens[b]) and zero-padded beyond, as if it were run alone at T = lens[b]
    // (nullptr: every item is valid on [0, T))
    const int* lens;
    // ragged batches, continued: the launch covers only the column groups (item b, frame tile ft) that hold valid frames;
    // cgmap[i] = b * tiles_per_b + ft of the i-th of them, ncg their number (nullptr: all batch * tiles_per_b of them)
    const int* cgmap;
    int ncg;
    float* out2;
    const float* cpn;
    long cpn_bstride;
    int cpn_rstride;
    int strong;
    float* lnpart;
    int lnpart_ts;
    int nout;               // EP_LINCOMB
    LinOut lo[kMaxOut];
};

// gemm.hip
hipError_t launch_gemm(const GemmP& p, int stage, int taps, int epi, int nb, int fast, int batch, hipStream_t st);
bool gemm_has_fast(int taps, int nb, int S);
int gemm_lds_bytes(int KC, int S);
int gemm_lds_bytes_fast(int S, int stage, int taps, int K, int nb, int resident);
int gemm_fast_chunk_rows(int taps, int nb);
hipError_t gemm_init_all();

// wn_layer.hip: one WaveNet residual layer (conv + FiLM + gate + out-proj + residual / skip) per launch, for grids of
// at least one 32-frame tile per CU
struct WnLayerP {
    const float* Aconv;     // packed dilated-conv weights (PackedGemm, pairC = C): [2C/16 blocks][C/64 * 12 k16][64][4]
    const float* Aout;      // packed output-projection weights: [2C/16 blocks][C/16 k16][64][4]
    const float* bias_out;  // output-projection bias [2C]
    const float* xin;       // residual stream, internal layout [B][C][Ts]: read (tile + halo)
    float* xout;            // residual stream after the layer (a different buffer: neighbours read xin's halo)
    float* skip;            // running skip sum, updated in place
    float* z;               // row-split pair only (wn_rowsplit.hip): the gated conv output between the two launches
    long x_bstride;         // floats between batch items of x / skip / z
    int Ts;
    const float* cp;        // this layer's hoisted conditioner projection rows [2C][Ts] (+ conv bias + its own bias)
    long cp_bstride;
    const float* film;      // step table: d[c] = film[c * film_cstride + film_col0 + b * film_colb]
    int film_cstride, film_col0, film_colb;
    int dil, T, tiles_per_b, first_layer;
    float inv_tiles_per_b;
    const int* lens;        // ragged batches: per-item valid length (nullptr: T)
    const int* cgmap;       // ragged batches: the (item, 32-frame tile) column groups that hold valid frames
    int ncg;
    int tile0;              // dense batches: the launch covers tiles [tile0, tile0 + ntiles) of the batch's (item, frame tile)
                            // order - a layer may run as several launches over disjoint tile ranges (api.hip, wn_plan_for)
    int ntiles;             // ... their number (0: all of batch * tiles_per_b; ragged: ncg)
};
hipError_t launch_wn_layer(const WnLayerP& p, int C, int batch, hipStream_t st, int bn = 32);      // bn: 32, or 16 (wn_layer16_kernel)
bool wn_layer16_supported(int C, int dil);
bool wn_layer_supported(int C, int dil);
hipError_t wn_layer_init_all();
// wn_rowsplit.hip: the same layer as two launches with the 2C rows split over 2C / 64 workgroups per 32-frame tile, for
// grids too small for full-row tiles.  which = 0: conv + FiLM + gate (xin -> z); 1: out-proj + residual / skip (in place
// when xout == xin)
hipError_t launch_wn_rowsplit(const WnLayerP& p, int which, int C, int batch, int bn, hipStream_t st);
hipError_t wn_rowsplit_init_all();
bool wn_rowsplit_supported(int C, int dil, long Ts);

// wn_layer_x3.hip: the fused layer in split-bf16 arithmetic (opt-in precision mode); p.Aconv / p.Aout = the layer's bf16x3
// weight streams [wave][k32 step][row block][hi | lo][lane][8 bf16]
hipError_t launch_wn_layer_x3(const WnLayerP& p, int C, int batch, hipStream_t st);
hipError_t wn_layer_x3_init_all();
bool wn_layer_x3_supported(int C, int dil);

// wn_rows.hip: the same two launches with 128 or 256 rows per workgroup (4 / 2 workgroups per 32-frame tile), for grids between
// the one-utterance case and one tile per CU
hipError_t launch_wn_rows(const WnLayerP& p, int which, int C, int batch, int rows, hipStream_t st);
hipError_t wn_rows_init_all();
bool wn_rows_supported(int C, int dil, long Ts);

// wn_edge.hip: the WaveNet's small GEMMs around the residual layers (skip projection -> output projection + solver update ->
// the next evaluation's input projection) as one launch with one workgroup per frame tile
constexpr int kEdgeMaxTerms = 8;      // state-buffer / noise terms of all outputs of one evaluation together
struct EdgeTerm {
    const float* ptr;       // state buffer (internal layout)
    long bstride;
    int rstride, ext;
    float coef;
    int out;                // the output this term belongs to
};
struct WnEdgeP {
    const float* A1;        // packed skip_projection [C x C], bias b1
    const float* b1;
    const float* A2;        // packed output_projection [F*M x C], bias b2
    const float* b2;
    const float* A3;        // packed input_projection [C x F*M], bias b3 (used when next_src >= 0)
    const float* b3;
    const float* skip;      // running skip sum [B][C][Ts]
    float* xh;              // the next evaluation's layer-0 input [B][C][Ts]
    long x_bstride;
    int Ts, T, FM;
    float in_scale;         // sqrt(L): the staged skip sum is divided by it (wavenet.py:96)
    int tiles_per_b;
    float inv_tiles_per_b;
    int nout, next_src;     // solver outputs; which of them is the next evaluation's input (-1: none - no input projection)
    float* dst[kMaxOut];    // output o = cm[o] * eps + sum of its terms (in the program's order)
    float cm[kMaxOut];
    int nq;
    EdgeTerm q[kEdgeMaxTerms];
    long o_bstride;         // state buffers: floats between batch items / rows
    int o_rstride;
    const int* cgmap;       // ragged batches: the (item, frame tile) list of this tile width
    int ncg;
};
hipError_t launch_wn_edge(const WnEdgeP& p, int C, int ncb, int nwg, hipStream_t st);
hipError_t wn_edge_init_all();
bool wn_edge_supported(int C, int FM);

// lynx_layer.hip: LYNXNet's two pointwise GEMMs with the whole K extent of a 32-frame tile resident in LDS (batched grids)
struct LxLayerP {
    const float* A1;        // packed pw1 weights (PackedGemm, pairC = inner; LayerNorm affine folded in)
    const float* bias1;     // [2 inner]
    const float* A2;        // packed pw2 weights
    const float* bias2;     // [C]
    const float* xin;       // pw1 input: the layer's pre-LayerNorm activations [B][C][Ts]
    const float* stats;     // [B][2][Ts]: mean, rstd per frame (ln_merge_kernel) - lx_pw1_kernel
    const float* lnpart_in; // the producer's LayerNorm partials of xin (same layout as lnpart) - lx_pw1p_kernel merges them itself
    float* u;               // pw1 output [B][inner][Ts]
    const float* v;         // pw2 input (depthwise conv output) [B][inner][Ts]
    float* x;               // residual stream [B][C][Ts]: read and replaced by pw2
    float* xin_out;         // the NEXT layer's pre-LayerNorm input (nullptr after the last layer)
    const float* cpn;       // next layer's hoisted conditioner projection rows [C][Ts] (nullptr after the last layer)
    long cpn_bstride;
    const float* film;      // next layer's step projection: d[c] = film[c * film_cstride + film_col0 + b * film_colb]
    int film_cstride, film_col0, film_colb;
    float* lnpart;          // [B][C / 64][2][lnpart_ts]: per 64-row tile mean and sum of squared deviations of xin_out
    int lnpart_ts, ln_tiles;
    long x_bstride, u_bstride;
    int inner, Ts, T, tiles_per_b, nft, strong;
    float inv_tiles_per_b, inv_nft;     // inv_nft = 1 / (ragged ? ncg : nft)
    const int* cgmap;       // ragged batches: the (item, 32-frame tile) column groups that hold valid frames
    int ncg;
    int rt_groups;          // lx_pw1p_kernel: workgroups per frame tile, each looping over (2 inner / 512) / rt_groups row tiles (0 / 1: one)
};
hipError_t launch_lx_layer(const LxLayerP& p, int which, int C, hipStream_t st);      // which: 0 = pw1, 1 = pw2
bool lx_layer_supported(int C, int inner);
hipError_t launch_lx_pw2q(const LxLayerP& p, int C, hipStream_t st);      // pw2 with 128 rows per workgroup: one-utterance grids
bool lx_pw2q_supported(int C, int inner);
hipError_t lx_layer_init_all();
bool lx_pw1_merges_stats(const LxLayerP& p, int C);

// lynx_x3.hip: the two pointwise GEMMs in split-bf16 arithmetic (opt-in precision mode); p.A1 / p.A2 = the layer's bf16x3 weight
// streams [row tile][wave][k32 step][row block][hi | lo][lane][8 bf16]
hipError_t launch_lx_x3(const LxLayerP& p, int which, int C, int ncb, hipStream_t st);     // ncb: 2 = 32-frame tiles, 4 = 64-frame tiles
hipError_t lx_x3_init_all();
bool lx_x3_supported(int C, int inner);

// aux_kernels.hip
hipError_t launch_pack(const float* src, long sb, long sr, long st, float* dst, int B, int R, int T, int Ts,
                       hipStream_t stream);
hipError_t launch_unpack(const float* src, int Ts, float* dst, int B, int F, int M, int T, int transpose,
                         const float* scale, const float* shift, hipStream_t stream);
hipError_t launch_transpose(const float* src, int rows, int cols, int src_stride, float* dst, int dst_stride, hipStream_t st);
hipError_t launch_sinemb(const float* t_dev, int ncols, int colstride, const float* freqs, int C, float* dst,
                         hipStream_t stream);
hipError_t launch_lynx_pre(float* x, float* xin, const float* cp, long cp_bstride, const float* film,
                           int film_cstride, int film_col0, int film_colb, long bstride, int rstride, int C, int B,
                           int T, int strong, float* stats, int ts, float eps, hipStream_t stream);
// tconv.hip: time-major MFMA convolution for 16 / 32 channels
struct TConvP {
    const float* W;         // B fragments [tap][CI/4][CO/16][64]: lane l = W[o = nb*16 + (l&15)][c = c4*4 + (l>>4)][tap]
    const float* bias;      // [co_real]
    const float* x;         // input, internal layout
    long x_bstride;
    int x_rstride;
    float* out;
    const float* res;       // residual added after the activation, same layout as out (may alias it), or nullptr
    long o_bstride;
    int o_rstride;
    int T, Ts_out;
    int taps, dil;
    int HP, SP;             // staged halo (multiple of 4) and LDS row stride (16 mod 32)
    float slope_in;         // leaky ReLU on the input (1 = none)
    int act;                // ACT_NONE / ACT_LRELU / ACT_TANH on the output (before the residual)
    int co_real;            // output channels actually stored
    int lds_bytes;
};
int tconv_lds_bytes(int ci, int co, int taps, int SP);
hipError_t tconv_init_all();
hipError_t launch_tconv(const TConvP& p, int ci, int co, int batch, hipStream_t st);
typedef float f32x4_t __attribute__((ext_vector_type(4)));
// vocoder_kernels.hip (NSF-HiFiGAN source, noise convs, residual-block average)
hipError_t launch_voc_source(const float* f0, const float* rand_ini, const float* noise, const float* lin_w,
                             const float* lin_b, int B, int T, int upp, int dim, float sr, float sine_amp, float noise_std,
                             float* acc_tmp, int Tsu, float* har, hipStream_t st);
hipError_t launch_voc_add_noise(float* x, const float* noise, int B, int C, int T, int Ts, float sigma, hipStream_t st);
hipError_t launch_voc_fast_source(const float* f0, int B, int T, int upp, float source_sr, float* acc_tmp, int Tsu, float* har,
                                  hipStream_t st);
hipError_t launch_voc_noise_conv(float* x, const float* har, const float* w, const float* bias, int B, int C, int Tq,
                                 int Tsq, int sf, int ksz, long Tup, int Tsu, hipStream_t st);
hipError_t launch_voc_accum(float* acc, const float* r, long n, int first, float div, hipStream_t st);
// encoder_kernels.hip (FastSpeech2 acoustic encoder glue)
struct EncExpandArgs {
    const float* lin_w[7];
    const float* lin_b[7];
    const float* feat[7];
    const float* spk_table;
    const long long* spk_id;
    const float* spk_mix;
    long spk_mix_bstride, spk_mix_tstride;
    int num_spk;
};
struct AssembleArgs {             // dsd_cond_assemble, device view
    int B, T, H, n_gather, n_terms;
    const float* g_table[4];
    long g_bstride[4], g_rows[4], g_off[4];
    const long long* g_idx[4];
    float g_scale[4];
    const float* g_rowscale[4];
    const float* t_s[16];
    const float* t_v[16];
};
hipError_t launch_enc_sinpos(float* x, const float* nonpad, const float* freqs, int C, int B, int L, int Ls, hipStream_t st);
hipError_t launch_enc_relpos(float* x, const float* div, int C, int B, int L, int Ls, hipStream_t st);
hipError_t launch_enc_nonpad(const unsigned char* pad, int B, int L, int Ls, float* nonpad, hipStream_t st);
hipError_t launch_enc_dur_head(const float* x, const float* w, const float* bias, const float* nonpad, int C, int B, int L,
                               int Ls, float offset, float* dur, hipStream_t st);
hipError_t launch_assemble(const AssembleArgs& a, float* out, hipStream_t st);
hipError_t launch_enc_dur(const long long* mel2ph, int B, int T, int L, int* dur, hipStream_t st);
hipError_t launch_enc_embed(const long long* tokens, const long long* langs, const int* dur, const float* txt_embed,
                            int vocab, const float* lang_embed, int n_lang_rows, const float* dur_w, const float* dur_b,
                            float embed_scale, int H, int B, int L, int Ls, float* x, float* nonpad, hipStream_t st);
hipError_t launch_enc_layernorm(const float* x, float* y, const float* g, const float* beta, const float* mask, int C,
                                int B, int L, int Ls, float eps, hipStream_t st);
hipError_t launch_enc_mask(float* x, const float* mask, int C, int B, int L, int Ls, hipStream_t st);
// SwiGLU between ffn_1 and ffn_2: x[b][c][l] *= silu(x[b][half + c][l]) for c < half (common_layers.py:107-117)
hipError_t launch_enc_swiglu(float* x, int half, long bstride, int B, int L, int Ls, hipStream_t st);
hipError_t launch_enc_rope(float* qkv, const float* freqs, int H, int head_dim, int B, int L, int Ls, hipStream_t st);
hipError_t launch_enc_attention(const float* qkv, const float* nonpad, float* out, int H, int heads, int B, int L, int Ls,
                                hipStream_t st);
hipError_t launch_enc_expand(const float* enc, const long long* mel2ph, const EncExpandArgs& a, int H, int B, int L, int Ls,
                             int T, float* cond, hipStream_t st);
hipError_t launch_ln_merge(const float* lnpart, int mtiles, int C, int B, int T, int ts, float eps, float* stats,
                           hipStream_t stream);
hipError_t launch_dwconv(const float* src, float* dst, long bstride, int rstride, int C, int B, int T, const int* lens,
                         const float* w, const float* bias, int ksz, int act, const float* prelu, hipStream_t stream);

}  // namespace dsd
