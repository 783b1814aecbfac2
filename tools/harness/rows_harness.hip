// Diagnostic harness (GPU box): the out-proj kernels of wn_rows.hip at 128 and 256 rows per workgroup on the same random
// buffers; prints where they differ (row of 2C, frame) and whether each is repeatable.
//   hipcc -O2 --offload-arch=gfx950 -I diffsinger_amd/csrc tools/harness/rows_harness.hip diffsinger_amd/csrc/wn_rows.o -o /tmp/rows_harness
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
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
int main() {
    const int C = 256, T = 256, Ts = padded_ts(T), B = 1;
    const size_t xs = (size_t)C * Ts;
    std::vector<float> hA((size_t)32 * 16 * 256), hb(512), hz(xs), hx(xs), hs(xs);
    srand(1);
    auto rnd = [] { return (rand() % 2001 - 1000) / 1000.f; };
    for (auto& v : hA) v = rnd() * 0.1f;
    for (auto& v : hb) v = rnd();
    for (auto& v : hz) v = rnd();
    for (auto& v : hx) v = rnd();
    for (auto& v : hs) v = rnd();
    float *A, *bias, *z, *x, *xo, *skip;
    hipMalloc(&A, hA.size() * 4); hipMalloc(&bias, 512 * 4); hipMalloc(&z, (xs + 512) * 4); hipMalloc(&x, (xs + 512) * 4);
    hipMalloc(&xo, (xs + 512) * 4); hipMalloc(&skip, (xs + 512) * 4);
    hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(bias, hb.data(), 512 * 4, hipMemcpyHostToDevice);
    hipMemcpy(z + 256, hz.data(), xs * 4, hipMemcpyHostToDevice);
    hipMemcpy(x + 256, hx.data(), xs * 4, hipMemcpyHostToDevice);
    if (wn_rows_init_all() != hipSuccess) { printf("init failed\n"); return 1; }
    std::vector<std::vector<float>> res;
    for (int rows : {128, 256, 128, 256, 256}) {
        hipMemcpy(skip + 256, hs.data(), xs * 4, hipMemcpyHostToDevice);
        hipMemset(xo, 0, (xs + 512) * 4);
        WnLayerP p{};
        p.Aout = A; p.bias_out = bias; p.xin = x + 256; p.xout = xo + 256; p.skip = skip + 256; p.z = z + 256;
        p.x_bstride = xs; p.Ts = Ts; p.dil = 1; p.T = T; p.tiles_per_b = T / 32; p.inv_tiles_per_b = 1.f / p.tiles_per_b; p.first_layer = 0;
        hipError_t e = launch_wn_rows(p, 1, C, B, rows, nullptr);
        hipError_t e2 = hipDeviceSynchronize();
        printf("rows %d: launch %s sync %s\n", rows, hipGetErrorString(e), hipGetErrorString(e2));
        std::vector<float> o(2 * xs);
        hipMemcpy(o.data(), xo + 256, xs * 4, hipMemcpyDeviceToHost);
        hipMemcpy(o.data() + xs, skip + 256, xs * 4, hipMemcpyDeviceToHost);
        res.push_back(o);
    }
    auto cmp = [&](int a, int b, const char* what) {
        double mx = 0; int n = 0, fr = -1, fc = -1;
        std::vector<int> rowbad(512, 0);
        for (int r = 0; r < 512; ++r)
            for (int t = 0; t < T; ++t) {
                double d = fabs(res[a][(size_t)r * Ts + t] - res[b][(size_t)r * Ts + t]);
                if (d > 1e-5) { ++n; ++rowbad[r]; if (fr < 0) { fr = r; fc = t; } }
                if (d > mx) mx = d;
            }
        printf("%s: max diff %.3e, %d bad of %d (first row %d frame %d)\n", what, mx, n, 512 * T, fr, fc);
        if (n) {
            int shown = 0;
            for (int r = 0; r < 512 && shown < 48; ++r)
                for (int t = 0; t < T && shown < 48; ++t) {
                    double d = fabs(res[a][(size_t)r * Ts + t] - res[b][(size_t)r * Ts + t]);
                    if (d > 1e-5) { printf("  (%d,%d: %.3f vs %.3f)", r, t, res[a][(size_t)r * Ts + t], res[b][(size_t)r * Ts + t]); ++shown; if (shown % 4 == 0) printf("\n"); }
                }
            printf("\n");
        }
        if (n) { printf("  bad rows per 16-row block:"); for (int bk = 0; bk < 32; ++bk) { int s = 0; for (int r = 0; r < 16; ++r) s += rowbad[bk * 16 + r]; printf(" %d", s); } printf("\n"); }
    };
    {   // CPU reference of the bad elements of run 1 (256 rows): which component is off?
        auto Wd = [&](int row, int c) { int b = row / 16, m = row % 16, sp = c / 16, j = (c % 16) / 4, kk = c % 4; return hA[((size_t)(b * 16 + sp) * 64 + kk * 16 + m) * 4 + j]; };
        int shown = 0;
        for (int r = 0; r < 512 && shown < 24; ++r)
            for (int t = 0; t < T && shown < 24; ++t) {
                double d = fabs(res[1][(size_t)r * Ts + t] - res[0][(size_t)r * Ts + t]);
                if (d <= 1e-5) continue;
                double s0 = hb[r], s1 = 0;
                for (int c = 0; c < 128; ++c) s0 += (double)Wd(r, c) * hz[(size_t)c * Ts + t];
                for (int c = 128; c < 256; ++c) s1 += (double)Wd(r, c) * hz[(size_t)c * Ts + t];
                double pre = r < 256 ? hx[(size_t)r * Ts + t] : hs[(size_t)(r - 256) * Ts + t];
                double sc = r < 256 ? 0.7071067811865476 : 1.0;
                printf("  (%d,%d) got %.4f  ref128 %.4f | full %.4f  s0+pre %.4f  s1+pre %.4f  s0+s1 %.4f  pre %.4f s0 %.4f s1 %.4f\n", r, t,
                       res[1][(size_t)r * Ts + t], res[0][(size_t)r * Ts + t], (s0 + s1 + pre) * sc, (s0 + pre) * sc, (s1 + pre) * sc, (s0 + s1) * sc, pre, s0, s1);
                ++shown;
            }
    }
    cmp(0, 2, "128 vs 128 again");
    cmp(1, 3, "256 vs 256 again");
    cmp(3, 4, "256 vs 256 third");
    cmp(0, 1, "128 vs 256");
    return 0;
}
