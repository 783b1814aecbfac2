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
//   * A is pre-packed on the host in fragment order, so a wave fetches the fragments of 4 k-steps
//     of one 16-row block with ONE coalesced 1 KiB global_load_dwordx4 straight into VGPRs
//     (weights are L2/MALL resident; no LDS round trip), software-pipelined PF iterations ahead.
//   * B: a [KC channels x (BN + 2*halo) frames] tile is staged in LDS once per K-chunk with 16-B
//     loads along the time axis; fragments are read with ds_read_b32 (row stride S = 16 mod 32
//     floats -> the 4 k-rows x 16 columns of a fragment hit 64 distinct banks).
//   * workgroup = 4 waves (one per SIMD) as 2 (rows) x 2 (frames); tile = 64 rows x 32*NB frames.
//     Gate/SwiGLU pairs (row r and row r + C) are packed into the same wave so the nonlinearity
//     is a pure register epilogue.
#include <type_traits>

#include "dsd_internal.h"

namespace dsd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float act_apply(float v, int act) {
    switch (act) {
        case ACT_RELU: return fmaxf(v, 0.f);
        case ACT_MISH: return v * tanhf(log1pf(expf(v)));           // nn.Mish (wavenet.py:60)
        case ACT_GELU: return 0.5f * v * (1.f + erff(v * 0.70710678118654752440f));  // exact GELU (lynxnet.py:106)
        default: return v;
    }
}
__device__ __forceinline__ float sigmoid_f(float v) { return 1.f / (1.f + expf(-v)); }

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
extern "C" int dsd_dbg_read_stamps(unsigned long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_stamps), sizeof(g_stamps));
}
#else
#define DSD_STAMP(i)
#endif

constexpr int PF = 8;   // A-fragment prefetch distance in k16 iterations (2 x 1 KiB loads each)

// The weight-fragment ring is loaded with inline asm and waited for with hand-counted s_waitcnt vmcnt(N):
// hipcc (ROCm 7.2) parks a conservative vmcnt(1) on the loop header for a loop-carried register ring, which
// exposes the whole L2 latency once per group.  Form (ii) of cdna_hip_programming.md 5.7: "=v" loads, then a
// wait statement that names the destinations "+v" so no consumer can be scheduled above it.  Inside the K loop
// the ring loads are the only VMEM operations, so "all but the 2*(PF-1) youngest" is exactly "slot u landed".
__device__ __forceinline__ void ring_load(f32x4& dst, const float* ptr) {
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory");
}
__device__ __forceinline__ void pre_load(float& dst, const float* ptr) {
    asm volatile("global_load_dword %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory");
}
__device__ __forceinline__ void pre_wait4(float (&a)[4]) {      // after a vmcnt(0) drain: orders the consumers
    asm volatile("" : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) :: "memory");
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
// instantiation (runtime S, any K multiple of 16, rotation-free wraparound walk) for every other shape.
template <int STAGE, int TAPS, int EPI, int NB, int SW>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmP p) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int BN = 32 * NB;
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);     // provably wave-uniform (SGPR)
    const int wm = wave >> 1, wn = wave & 1;
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
    const int mtile = work % p.mtiles;
    const int rest = work / p.mtiles;
    const int b = rest / p.tiles_per_b;
    const int t0 = (rest - b * p.tiles_per_b) * BN;
    const int K16 = p.K >> 4;
    const int S = SW > 0 ? SW : p.S;
    const int HL = p.HL;
    DSD_STAMP(0);

    f32x4 acc[2][NB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // packed A: [mblk][TAPS*K16][64 lanes][4]; this wave owns packed m-blocks 4*mtile + 2*wm + {0,1}
    const long a_blk = (long)(TAPS * K16) * 256;
    const float* a0p = p.A + (long)(mtile * 4 + wm * 2) * a_blk + lane * 4;
    const float* a1p = a0p + a_blk;

    const float* bsrc = p.B + (long)b * p.b_bstride;
    const int W4 = (BN + 2 * HL) >> 2;      // float4 per staged row
    const int lrow = lane >> 4, lcol = lane & 15;

    // EP_GATE: the hoisted conditioner projection of this wave's outputs is fetched now and consumed in the
    // epilogue, so its latency hides under the whole K loop.
    float cpv[2][NB][4];
    auto epi_prefetch = [&]() {
        if (EPI == EP_RESSKIP) {
            // residual stream / running skip sum of this wave's outputs, fetched before the K loop.
            // A 16-row block lies entirely in the residual half or in the skip half (C % 16 == 0): the base
            // pointer is chosen with integer arithmetic on values already in SGPRs (a select between the two
            // struct FIELDS makes hipcc load the pointer itself through a dependent vector load).
            const unsigned long long xa = (unsigned long long)p.x, sa = (unsigned long long)p.skip;
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const int row0 = mtile * 64 + (wm * 2 + mb) * 16;
                const bool is_res = row0 < p.C;
                const unsigned long long ba = is_res ? xa : sa;
                const int rowb = (is_res ? row0 : row0 - p.C) + (lane >> 4) * 4;
                const float* base = (const float*)ba;
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    const long col = (long)b * p.o_bstride + t0 + wn * (16 * NB) + n * 16 + lcol;
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        pre_load(cpv[mb][n][r], base + col + (long)(rowb + r) * p.o_rstride);
                }
            }
        }
        if (EPI == EP_GATE) {
            const int chb = (mtile * 2 + wm) * 16 + (lane >> 4) * 4;
    #pragma unroll
            for (int n = 0; n < NB; ++n) {
                const float* cp = p.aux + (long)b * p.aux_bstride + t0 + wn * (16 * NB) + n * 16 + lcol;
    #pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int ch = min(chb + r, p.C - 1);
                    pre_load(cpv[0][n][r], cp + (long)ch * p.aux_rstride);
                    pre_load(cpv[1][n][r], cp + (long)(ch + p.C) * p.aux_rstride);
                }
            }
        }
    };

    for (int kc = 0; kc < p.K; kc += p.KC) {
        const int kcn = min(p.KC, p.K - kc);
        const int n16 = kcn >> 4;
        const int nit = TAPS * n16;
        // A-fragment ring: PF iterations (one k16 step of both row blocks each) are kept in flight.  The loop
        // body is branch-free (conditional refills make hipcc serialise every load behind an s_waitcnt vmcnt(0)):
        // the (k16, tap) position simply wraps around, so refills past the end re-read valid, unused blocks.
        // The walk over the nit = TAPS * n16 steps starts at a per-workgroup ROTATION: the workgroups that
        // stream the same weight rows run in lockstep, and without it they all hit the same L2 channel with the
        // same 1 KiB block at the same moment.  Summation order therefore depends on the tile index only
        // (deterministic; fp32 rounding differs between tiles by the usual reassociation error).
        const int kc16 = kc >> 4;
        const int rot = SW > 0 ? 0 : (int)(((long)rest * nit) / p.rot_den) % nit;
        int pf_tap = rot / n16;
        int pf_c = rot - pf_tap * n16;
        f32x4 ra0[PF], ra1[PF];
        // fast path: byte address of this wave's block (tap 0, k16 = kc16) as a scalar; blocks follow linearly
        const unsigned long long a0s =
            (unsigned long long)(p.A + ((long)(mtile * 4 + wm * 2) * (TAPS * K16) + kc16) * 256);
        const unsigned long long a1s = a0s + (unsigned long long)a_blk * 4;
        unsigned long long an0 = a0s, an1 = a1s;            // base of the group the next refills belong to
        const unsigned voff0 = lane * 16, voff1 = lane * 16 + 4096;
        auto ring_issue = [&](f32x4& d0, f32x4& d1) {        // generic path
#ifdef DSD_EXP_NOSTREAM
            const long off = 0;      // diagnostic: every fragment load hits the same (L1-resident) block
#else
            const long off = (long)(pf_tap * K16 + kc16 + pf_c) * 256;
#endif
            ring_load(d0, a0p + off);
            ring_load(d1, a1p + off);
            const bool w = (++pf_c == n16);
            pf_c = w ? 0 : pf_c;
            pf_tap += w ? 1 : 0;
            pf_tap = (pf_tap == TAPS) ? 0 : pf_tap;
        };
        auto ring_group_fast = [&](auto uc) {                // fast path: slot u of the group at an0/an1
            constexpr int u = decltype(uc)::value;
            ring_load_s<(u & 3) * 1024>(ra0[u], u < 4 ? voff0 : voff1, an0);
            ring_load_s<(u & 3) * 1024>(ra1[u], u < 4 ? voff0 : voff1, an1);
        };
        auto ring_prologue = [&]() {
            if constexpr (SW > 0) {
                ring_group_fast(std::integral_constant<int, 0>{});
                ring_group_fast(std::integral_constant<int, 1>{});
                ring_group_fast(std::integral_constant<int, 2>{});
                ring_group_fast(std::integral_constant<int, 3>{});
                ring_group_fast(std::integral_constant<int, 4>{});
                ring_group_fast(std::integral_constant<int, 5>{});
                ring_group_fast(std::integral_constant<int, 6>{});
                ring_group_fast(std::integral_constant<int, 7>{});
                an0 += 8192;
                an1 += 8192;
            } else {
#pragma unroll
                for (int u = 0; u < PF; ++u) ring_issue(ra0[u], ra1[u]);
            }
        };
        if (kc > 0) __syncthreads();
        DSD_STAMP(1);
        // ---------------- stage B chunk: rows [kc, kc+kcn), frames [t0-HL, t0+BN+HL) ----------------
        // 2^lpr_shift lanes walk one row (16-B loads along time), SU row-loads are issued back to back
        // before the first one is consumed, so the L2/MALL latency is paid once per batch, not per row.
        {
            constexpr int SU = 8;       // row-loads in flight per lane and batch (16 would need ~280 VGPRs)
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
            // Branch-free on purpose: every lane always loads from a clamped (in-bounds) address and the mask is
            // applied by select afterwards.  Predicated loads become exec-masked branches, and hipcc then parks an
            // s_waitcnt vmcnt(0) between consecutive loads - the whole batch serialises on memory latency.
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
                        else if (p.in_scale != 1.f) y = y / p.in_scale;   // skip sum DIVIDED by sqrt(L) (wavenet.py:96)
                        const bool ok = (t >= 0) && (t < p.T) && row_ok;
                        o[e] = ok ? y : 0.f;       // zero padding applies AFTER the FiLM add (wavenet.py:36-38)
                    }
                    if (col_ok && r < kcn) *reinterpret_cast<f32x4*>(&lds[r * S + c4 * 4]) = o;
                }
            };
            // 1. every activation load of the (first) batch is issued back to back and RETIRED with a
            //    compiler-visible vmcnt(0): hipcc's own counted waits would otherwise also wait for the younger
            //    asm loads issued next (it cannot see them), i.e. for the whole 64 KB weight ring;
            // 2. the weight ring and the epilogue operands go out;  3. transform + LDS writes run under their flight.
            const int batch = rows_per_it * SU;
            issue(0);
            __builtin_amdgcn_s_waitcnt(0x0F70);
            ring_prologue();
            if (kc == 0) epi_prefetch();
            finish(0);
            for (int r0 = batch; r0 < kcn; r0 += batch) {
                issue(r0);
                __builtin_amdgcn_s_waitcnt(0x0F70);
                finish(r0);
            }
        }
        // The asm loads (weight ring prologue, epilogue operands) are retired here as well.  They must be:
        // hipcc does not know their destinations are still in flight and is free to MOVE those registers
        // (observed: wrong cond-proj / residual values, then a fault) - form (ii) of cdna_hip_programming.md
        // 5.7 pins order, not register allocation.  The wait is cheap: the loads had the whole transform +
        // LDS-write phase to land.
        DSD_STAMP(2);
        __builtin_amdgcn_s_waitcnt(0x0F70);
        DSD_STAMP(3);
        __syncthreads();
        DSD_STAMP(4);

        // ---------------- MFMA over (tap, k16) ----------------
        // B fragments are read one iteration ahead (two register sets), so the ds_read latency of step i+1
        // hides under the 8*NB MFMAs of step i instead of draining the matrix pipe at every step.
        int rtap = rot / n16;
        int rc16 = rot - rtap * n16;
        const float* bl0 = &lds[lrow * S + wn * (16 * NB) + lcol] + (HL - (TAPS == 3 ? p.dil : 0));
        float bq[2][4][NB];
        auto read_b = [&](float (&bv)[4][NB]) {
            const float* blp = bl0 + rc16 * (16 * S) + (TAPS == 3 ? rtap * p.dil : 0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int n = 0; n < NB; ++n) bv[j][n] = blp[j * 4 * S + n * 16];
            const bool w = (++rc16 == n16);
            rc16 = w ? 0 : rc16;
            rtap += w ? 1 : 0;
            rtap = (rtap == TAPS) ? 0 : rtap;
        };
        auto mfma_step = [&](const f32x4& A0, const f32x4& A1, const float (&bv)[4][NB]) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    acc[0][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(A0[j], bv[j][n], acc[0][n], 0, 0, 0);
                    acc[1][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(A1[j], bv[j][n], acc[1][n], 0, 0, 0);
                }
        };
        if constexpr (SW > 0) {
            static_assert(PF == 8, "fast path walks groups of 8 k16 steps");
            const int groups = nit >> 3;
            const int gpt = n16 >> 3;                         // groups per tap
            int gt = 0;
            auto read_bf = [&](float (&bv)[4][NB], const float* base, auto uc) {
                constexpr int u = decltype(uc)::value;
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int n = 0; n < NB; ++n) bv[j][n] = base[(u * 16 + j * 4) * SW + n * 16];
            };
            const float* bt = bl0;
            read_bf(bq[0], bt, std::integral_constant<int, 0>{});
            for (int g = 0; g < groups; ++g) {
                const bool wrap = (++gt == gpt);
                gt = wrap ? 0 : gt;
                const float* btn = bt + (wrap ? (TAPS == 3 ? p.dil : 0) - (n16 - 8) * 16 * SW : 8 * 16 * SW);
#define DSD_FAST_STEP(U)                                                                           \
    ring_wait<2 * (PF - 1)>(ra0[U], ra1[U]);                                                        \
    if constexpr (U < 7) read_bf(bq[(U + 1) & 1], bt, std::integral_constant<int, (U + 1) & 7>{}); \
    else read_bf(bq[0], btn, std::integral_constant<int, 0>{});                                    \
    mfma_step(ra0[U], ra1[U], bq[U & 1]);                                                           \
    __builtin_amdgcn_sched_barrier(0);                                                              \
    ring_group_fast(std::integral_constant<int, U>{});
                DSD_FAST_STEP(0) DSD_FAST_STEP(1) DSD_FAST_STEP(2) DSD_FAST_STEP(3)
                DSD_FAST_STEP(4) DSD_FAST_STEP(5) DSD_FAST_STEP(6) DSD_FAST_STEP(7)
#undef DSD_FAST_STEP
                bt = btn;
                an0 += 8192;
                an1 += 8192;
            }
#pragma unroll
            for (int u = 0; u < PF; ++u) ring_wait<0>(ra0[u], ra1[u]);     // drain
        } else {
        read_b(bq[0]);
        const int groups = nit / PF, rem = nit - groups * PF;
        for (int g = 0; g < groups; ++g) {
#pragma unroll
            for (int u = 0; u < PF; ++u) {
                ring_wait<2 * (PF - 1)>(ra0[u], ra1[u]);
                read_b(bq[(u + 1) & 1]);                 // past the last step this wraps to a valid, unused block
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
        }
    }

    if (EPI == EP_GATE || EPI == EP_RESSKIP) {      // prefetched operands: retired by the ring drain (vmcnt(0)) above
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int n = 0; n < NB; ++n) pre_wait4(cpv[i][n]);
    }
    DSD_STAMP(5);
    // ---------------------------------------- epilogue ----------------------------------------
    // C/D layout of 16x16x4: column = lane & 15, row = (lane >> 4) * 4 + reg.
    const int rq = (lane >> 4) * 4;
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        const int t = t0 + wn * (16 * NB) + n * 16 + lcol;
        if (EPI == EP_GATE || EPI == EP_SWIGLU) {
            // packed pair block: acc[0] = first-half rows (gate / out), acc[1] = second-half rows (filter / gate)
            const int chb = (mtile * 2 + wm) * 16 + rq;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int ch = chb + r;
                if (ch < p.C) {
                    float u0 = acc[0][n][r], u1 = acc[1][n][r];
                    float y;
                    if (EPI == EP_GATE) {
                        u0 += cpv[0][n][r];
                        u1 += cpv[1][n][r];
                        y = sigmoid_f(u0) * tanhf(u1);                 // wavenet.py:41-42
                    } else {
                        u0 += p.bias[ch];
                        u1 += p.bias[ch + p.C];
                        y = u0 * (u1 * sigmoid_f(u1));                 // out * silu(gate), common_layers.py:116-117
                    }
                    p.out[(long)b * p.o_bstride + (long)ch * p.o_rstride + t] = y;
                }
            }
        } else {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const int rowb = mtile * 64 + (wm * 2 + mb) * 16 + rq;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = rowb + r;
                    if (row >= p.M) continue;
                    float v = acc[mb][n][r] + (p.bias ? p.bias[row] : 0.f);
                    if (EPI == EP_BIAS_ACT) {
                        p.out[(long)b * p.o_bstride + (long)row * p.o_rstride + t] = act_apply(v, p.act);
                    } else if (EPI == EP_BIAS_RES) {
                        const float res = p.aux[(long)b * p.aux_bstride + (long)row * p.aux_rstride + t];
                        p.out[(long)b * p.o_bstride + (long)row * p.o_rstride + t] = v + res;
                    } else if (EPI == EP_RESSKIP) {
                        if (row < p.C) {                               // residual half (wavenet.py:47-48)
                            float* xp = p.x + (long)b * p.o_bstride + (long)row * p.o_rstride + t;
                            *xp = (cpv[mb][n][r] + v) / 1.41421356237309504880f;
                        } else {                                       // skip half: running sum replaces stack+sum (wavenet.py:96)
                            float* sp = p.skip + (long)b * p.o_bstride + (long)(row - p.C) * p.o_rstride + t;
                            *sp = p.first_layer ? v : (cpv[mb][n][r] + v);
                        }
                    } else if (EPI == EP_LINCOMB) {
                        float outv[kMaxOut];
#pragma unroll
                        for (int o = 0; o < kMaxOut; ++o) {
                            outv[o] = 0.f;
                            if (o < p.nout) {
                                const LinOut& lo = p.lo[o];
                                for (int k = 0; k < lo.nterms; ++k) {
                                    const LinTerm& tm = lo.t[k];
                                    float s;
                                    if (tm.ptr == nullptr) s = v;
                                    else if (tm.ext && t >= p.T) s = 0.f;
                                    else s = tm.ptr[(long)b * tm.bstride + (long)row * tm.rstride + t];
                                    outv[o] += tm.coef * s;
                                }
                            }
                        }
#pragma unroll
                        for (int o = 0; o < kMaxOut; ++o)
                            if (o < p.nout) p.lo[o].dst[(long)b * p.o_bstride + (long)row * p.o_rstride + t] = outv[o];
                    }
                }
            }
        }
    }
    DSD_STAMP(6);
}

int gemm_lds_bytes(int KC, int S) { return KC * S * 4; }

template <int STAGE, int TAPS, int EPI, int NB, int SW>
static hipError_t set_attr() {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel<STAGE, TAPS, EPI, NB, SW>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

template <int STAGE, int TAPS, int EPI, int NB, int SW>
static hipError_t launch_one(const GemmP& p, int batch, hipStream_t st) {
    const int lds = gemm_lds_bytes(p.KC, p.S);
    dim3 grid(batch * p.tiles_per_b * p.mtiles, 1, 1);
    hipLaunchKernelGGL((gemm_kernel<STAGE, TAPS, EPI, NB, SW>), grid, dim3(256), lds, st, p);
    return hipGetLastError();
}

// fast instantiations exist for S = 48 / 80 (32-frame tiles) and 80 / 112 (64-frame tiles); 1x1 GEMMs have
// no halo, so only the smaller stride of each tile width occurs for them
bool gemm_has_fast(int taps, int nb, int S) {
    if (nb == 1) return S == 48 || (taps == 3 && S == 80);
    return S == 80 || (taps == 3 && S == 112);
}

template <int STAGE, int TAPS, int EPI>
static hipError_t dispatch(const GemmP& p, int nb, int fast, int batch, hipStream_t st) {
    if (nb == 1) {
        if (fast && p.S == 48) return launch_one<STAGE, TAPS, EPI, 1, 48>(p, batch, st);
        if constexpr (TAPS == 3)
            if (fast && p.S == 80) return launch_one<STAGE, TAPS, EPI, 1, 80>(p, batch, st);
        return launch_one<STAGE, TAPS, EPI, 1, 0>(p, batch, st);
    }
    if (fast && p.S == 80) return launch_one<STAGE, TAPS, EPI, 2, 80>(p, batch, st);
    if constexpr (TAPS == 3)
        if (fast && p.S == 112) return launch_one<STAGE, TAPS, EPI, 2, 112>(p, batch, st);
    return launch_one<STAGE, TAPS, EPI, 2, 0>(p, batch, st);
}

template <int STAGE, int TAPS, int EPI>
static hipError_t attr_all() {
    hipError_t e;
    if ((e = set_attr<STAGE, TAPS, EPI, 1, 0>()) != hipSuccess) return e;
    if ((e = set_attr<STAGE, TAPS, EPI, 2, 0>()) != hipSuccess) return e;
    if ((e = set_attr<STAGE, TAPS, EPI, 1, 48>()) != hipSuccess) return e;
    if ((e = set_attr<STAGE, TAPS, EPI, 2, 80>()) != hipSuccess) return e;
    if constexpr (TAPS == 3) {
        if ((e = set_attr<STAGE, TAPS, EPI, 1, 80>()) != hipSuccess) return e;
        if ((e = set_attr<STAGE, TAPS, EPI, 2, 112>()) != hipSuccess) return e;
    }
    return hipSuccess;
}

// Raise the dynamic-LDS limit of every instantiation once, outside any stream capture.
hipError_t gemm_init_all() {
    hipError_t e;
    if ((e = attr_all<ST_PLAIN, 1, EP_BIAS_ACT>()) != hipSuccess) return e;
    if ((e = attr_all<ST_FILM, 3, EP_GATE>()) != hipSuccess) return e;
    if ((e = attr_all<ST_PLAIN, 1, EP_RESSKIP>()) != hipSuccess) return e;
    if ((e = attr_all<ST_PLAIN, 1, EP_LINCOMB>()) != hipSuccess) return e;
    if ((e = attr_all<ST_LN, 1, EP_SWIGLU>()) != hipSuccess) return e;
    if ((e = attr_all<ST_PLAIN, 1, EP_BIAS_RES>()) != hipSuccess) return e;
    if ((e = attr_all<ST_LN, 1, EP_LINCOMB>()) != hipSuccess) return e;
    return hipSuccess;
}

#define DSD_CASE(ST, TP, EP) \
    if (stage == ST && taps == TP && epi == EP) return dispatch<ST, TP, EP>(p, nb, fast, batch, st);

hipError_t launch_gemm(const GemmP& p, int stage, int taps, int epi, int nb, int fast, int batch, hipStream_t st) {
    DSD_CASE(ST_PLAIN, 1, EP_BIAS_ACT)
    DSD_CASE(ST_FILM, 3, EP_GATE)
    DSD_CASE(ST_PLAIN, 1, EP_RESSKIP)
    DSD_CASE(ST_PLAIN, 1, EP_LINCOMB)
    DSD_CASE(ST_LN, 1, EP_SWIGLU)
    DSD_CASE(ST_PLAIN, 1, EP_BIAS_RES)
    DSD_CASE(ST_LN, 1, EP_LINCOMB)
    return hipErrorInvalidValue;
}

}  // namespace dsd
