// Diagnostic harness (GPU box): lynx_x3.hip's pw1 on 32-frame against 64-frame tiles on the same random buffers; prints where
// they differ (u channel, frame: by frame tile, row tile, frame in tile, channel mod 16) and whether each is repeatable.  (Written
// to find why an experimental build of the 64-frame kernel gave different results from run to run: only the ODD frames of a
// tile did - the build formed its hi / lo split with v_pk_add_f32 ... op_sel on register pairs assembled by v_mov)
//   hipcc -O2 --offload-arch=gfx950 -I diffsinger_amd/csrc -I include tools/harness/x3_harness.hip diffsinger_amd/csrc/lynx_x3.o -o tools/harness/x3_harness.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include "dsd_internal.h"
namespace dsd {
static PathOpts g_o = {};
const PathOpts& path_opts() { return g_o; }
void refresh_path_opts() {}
TimingSlot& timing_slot() { static thread_local TimingSlot s; return s; }
}
using namespace dsd;
static unsigned short bf16_bits(float v) { unsigned u; memcpy(&u, &v, 4); u += 0x7fffu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); }
int main(int argc, char** argv) {
    const int C = 1024, inner = 2048, B = 2, T = argc > 1 ? atoi(argv[1]) : 256;
    const int Ts = ((T + 63) / 64) * 64 + 32;
    const size_t xs = (size_t)C * Ts, us = (size_t)inner * Ts;
    const int NT = C / 64, lts = Ts;
    srand(1);
    auto rnd = [] { return (rand() % 2001 - 1000) / 1000.f; };
    std::vector<float> hx(B * xs), hp((size_t)B * NT * 2 * lts), hb(2 * inner);
    for (auto& v : hx) v = rnd();
    // consistent partials: per 64-row tile mean and sum of squared deviations
    for (int b = 0; b < B; ++b)
        for (int i = 0; i < NT; ++i)
            for (int t = 0; t < Ts; ++t) {
                double m = 0, q = 0;
                for (int r = 0; r < 64; ++r) m += hx[b * xs + (size_t)(64 * i + r) * Ts + t];
                m /= 64;
                for (int r = 0; r < 64; ++r) { double d = hx[b * xs + (size_t)(64 * i + r) * Ts + t] - m; q += d * d; }
                hp[((size_t)b * NT + i) * 2 * lts + t] = (float)m;
                hp[((size_t)b * NT + i) * 2 * lts + lts + t] = (float)q;
            }
    for (auto& v : hb) v = rnd();
    // weight stream: [row tile 8][wave 4][k32 step 32][row block 8][hi | lo][lane 64][8 bf16] - random small values
    const size_t nw = (size_t)8 * 4 * 32 * 8 * 2 * 64 * 8;
    std::vector<unsigned short> hw(nw);
    for (size_t i = 0; i < nw; ++i) hw[i] = bf16_bits(rnd() * (((i / 512) & 1) ? 0.0002f : 0.05f));
    float *x, *part, *bias, *u; unsigned short* w;
    hipMalloc(&x, (B * xs + 1024) * 4); hipMalloc(&part, hp.size() * 4 + 4096); hipMalloc(&bias, hb.size() * 4); hipMalloc(&u, (B * us + 1024) * 4); hipMalloc(&w, nw * 2);
    hipMemcpy(x, hx.data(), B * xs * 4, hipMemcpyHostToDevice);
    hipMemcpy(part, hp.data(), hp.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(bias, hb.data(), hb.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(w, hw.data(), nw * 2, hipMemcpyHostToDevice);
    if (lx_x3_init_all() != hipSuccess) { printf("init failed\n"); return 1; }
    LxLayerP p{};
    p.A1 = reinterpret_cast<const float*>(w); p.bias1 = bias; p.xin = x; p.lnpart_in = part; p.lnpart_ts = lts; p.u = u;
    p.x_bstride = (long)xs; p.u_bstride = (long)us; p.inner = inner; p.Ts = Ts; p.T = T;
    std::vector<std::vector<float>> res;
    for (int ncb : {2, 4, 4, 4, 2}) {
        hipMemset(u, 0, (B * us + 1024) * 4);
        const int bn = 16 * ncb;
        p.tiles_per_b = (T + bn - 1) / bn; p.inv_tiles_per_b = 1.f / p.tiles_per_b; p.nft = B * p.tiles_per_b; p.inv_nft = 1.f / p.nft;
        hipError_t e = launch_lx_x3(p, 0, C, ncb, nullptr);
        hipError_t e2 = hipDeviceSynchronize();
        if (e != hipSuccess || e2 != hipSuccess) { printf("launch ncb %d: %s / %s\n", ncb, hipGetErrorString(e), hipGetErrorString(e2)); return 1; }
        std::vector<float> r(B * us);
        hipMemcpy(r.data(), u, B * us * 4, hipMemcpyDeviceToHost);
        res.push_back(r);
    }
    auto cmp = [&](int a, int b, const char* what) {
        size_t bad = 0; double worst = 0; size_t wi = 0;
        std::vector<int> per_tile(64, 0), per_rt(8, 0), per_col(64, 0), per_row16(16, 0);
        for (int bi = 0; bi < B; ++bi)
            for (int ch = 0; ch < inner; ++ch)
                for (int t = 0; t < T; ++t) {
                    const size_t i = bi * us + (size_t)ch * Ts + t;
                    const double d = fabs((double)res[a][i] - res[b][i]);
                    if (d > 1e-5 * (1 + fabs(res[a][i]))) {
                        ++bad; per_tile[bi * ((T + 63) / 64) + t / 64]++; per_rt[ch / 256]++; per_col[t % 64]++; per_row16[ch % 16]++;
                        if (d > worst) { worst = d; wi = i; }
                    }
                }
        printf("%s: %zu of %zu differ (> 1e-5 rel), worst %.3e at item %zu ch %zu frame %zu (%g vs %g)\n", what, bad, (size_t)B * inner * T, worst,
               wi / us, wi % us / Ts, wi % Ts, res[a][wi], res[b][wi]);
        if (bad) {
            printf("   by frame tile:"); for (int i = 0; i < B * ((T + 63) / 64); ++i) printf(" %d", per_tile[i]);
            printf("\n   by row tile:"); for (int i = 0; i < 8; ++i) printf(" %d", per_rt[i]);
            printf("\n   by frame in tile:"); for (int i = 0; i < 64; ++i) printf(" %d", per_col[i]);
            printf("\n   by channel mod 16:"); for (int i = 0; i < 16; ++i) printf(" %d", per_row16[i]);
            printf("\n");
        }
    };
    cmp(0, 4, "32-frame tiles, run 1 vs 2");
    cmp(1, 2, "64-frame tiles, run 1 vs 2");
    cmp(1, 3, "64-frame tiles, run 1 vs 3");
    cmp(0, 1, "32- vs 64-frame tiles");
    return 0;
}
