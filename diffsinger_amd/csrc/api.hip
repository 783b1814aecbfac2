// libdsdenoise C-ABI (include/dsdenoise.h): handle, weight re-layout, workspace, backbone launch
// sequences (WaveNet / LYNXNet), sampling-program executor and hipGraph cache.  gfx950 only.
//
// Map of this file (one translation unit: every entry point shares the handle, the packed-weight blob and make_gemm):
//   handle + host tensors ............ struct dsd_handle, expected_params*(): the state-dict layouts the library accepts
//   weight re-layout ................. pack_gemm (MFMA fragment order), build_packed{,_aux,_enc,_tok,_voc}
//   workspaces ....................... ensure_workspace / ensure_state / ensure_emb (kept across shape changes)
//   GEMM launch decisions ............ make_gemm (tile width, fast / generic path, ragged lengths), run_gemm
//   denoiser ......................... run_step_tables, run_backbone (WaveNet 43 kernels, LYNXNet), dsd_create ...
//                                      dsd_prepare_cond, dsd_denoise, dsd_sample (+ hipGraph cache), dsd_set_lengths
//   acoustic encoder ................. dsd_encoder_create, enc_workspace, run_fs2_layers, dsd_encode
//   variance-model encoders .......... dsd_token_encoder_create, dsd_token_encode, dsd_predict_dur, dsd_cond_assemble
//   vocoder .......................... run_tconv, dsd_vocoder_create, dsd_vocode
//   aux decoder ...................... dsd_aux_decode
//   diagnostics ...................... dsd_get_stats, dsd_kernel_timing*
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>

#include <algorithm>
#include <functional>
#include <map>
#include <set>
#include <string>
#include <vector>

#include "../../include/dsdenoise.h"
#include "dsd_internal.h"

using namespace dsd;

// ------------------------------------------------------------------------------------------
// path switches: read from the environment once per C-ABI entry point (dsd_internal.h, PathOpts)
// ------------------------------------------------------------------------------------------
namespace dsd {
static PathOpts g_path_opts = {-1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, -1, 512};
const PathOpts& path_opts() { return g_path_opts; }
void refresh_path_opts() {
    auto geti = [](const char* name, int dflt) {
        const char* v = getenv(name);
        return v && *v ? atoi(v) : dflt;
    };
    PathOpts& o = g_path_opts;
    o.fused_layer = geti("DSD_FUSED_LAYER", -1);
    o.wn_plan = geti("DSD_WN_PLAN", -1);
    o.rowsplit = geti("DSD_ROWSPLIT", -1);
    o.rs_bn48 = geti("DSD_RS_BN48", -1);
    o.rs_conv_q = geti("DSD_RS_CONV_Q", -1);
    o.rs_rows = geti("DSD_RS_ROWS", -1);
    o.rs_rows_out = geti("DSD_RS_ROWS_OUT", -1);
    o.edge = geti("DSD_EDGE", -1);
    o.lynx_resident = geti("DSD_LYNX_RESIDENT", -1);
    o.lynx_pw1p = geti("DSD_LYNX_PW1P", -1);
    o.lynx_pw2d = geti("DSD_LYNX_PW2D", -1);
    o.lynx_pw2q = geti("DSD_LYNX_PW2Q", -1);
    o.narrow = geti("DSD_NARROW", -1);
    o.gm_shift = geti("DSD_GM_SHIFT", -1);
    o.film_t = geti("DSD_FILM_T", -1);
    o.dwconv_rows = geti("DSD_DWCONV_ROWS", -1);
    o.precision = geti("DSD_PRECISION", -1);
    o.fused16 = geti("DSD_FUSED16", -1);
    o.x3_wide = geti("DSD_X3_WIDE", -1);
    const char* nb = getenv("DSD_NB2_MIN_WG");
    o.nb2_min = nb && *nb ? atol(nb) : 512;
}
}  // namespace dsd

namespace dsd {
TimingSlot& timing_slot() {
    static thread_local TimingSlot slot;
    return slot;
}
}  // namespace dsd

namespace {

std::string g_create_error = "";

struct HostTensor {
    std::vector<int64_t> shape;
    std::vector<float> data;
};

// one packed GEMM operand set on the device
struct PackedGemm {
    size_t a_off = 0;     // float offset into the weight blob
    size_t bias_off = 0;  // float offset, or SIZE_MAX
    int M = 0;            // real rows
    int K = 0;            // padded input channels
    int Kreal = 0;
    int taps = 1;
    int pairC = 0;        // > 0: paired packing with this many pairs
};

// weights of a few-channel convolution in tconv.hip's B-fragment order
struct PackedTConv {
    size_t w_off = SIZE_MAX, b_off = 0;
    int ci = 0, co = 0, co_real = 0, taps = 0;
    bool valid() const { return w_off != SIZE_MAX; }
};

struct GraphEntry {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
};

}  // namespace

struct dsd_handle {
    dsd_config cfg;
    std::string err;
    std::map<std::string, HostTensor> raw;
    // WaveNet with a channel count that is not a multiple of 32: cfg.num_channels is the count the kernels run with
    // (rounded up), c_user the caller's; `padded` holds the zero-extended tensors build_packed reads (pad_wavenet_weights)
    std::map<std::string, HostTensor> padded;
    int c_user = 0;
    // wn_edge.hip: the state buffer whose input projection the previous evaluation's edge kernel already wrote into xh
    const float* edge_xh_src = nullptr;
    bool finalized = false;

    // packed weights
    std::vector<float> blob_host;
    float* blob = nullptr;
    size_t blob_floats = 0;
    PackedGemm g_inproj, g_emb0, g_emb1, g_dproj, g_cp, g_tail1, g_out;
    std::vector<PackedGemm> g_conv, g_outp;          // WaveNet per layer
    // split-bf16 precision mode (wn_layer_x3.hip): 0 = fp32 (default), 1 = bf16x3 where a kernel exists; the layers' weight
    // streams (float offsets into the blob; empty: not built)
    int precision = 0;
    std::vector<size_t> x3_conv, x3_out;
    int cus = 256;                  // compute units of the device (hipDeviceProp_t::multiProcessorCount): one fused round = `cus` tiles
    std::vector<PackedGemm> g_pw1, g_pw2;            // LYNXNet per layer
    std::vector<size_t> dw_w, dw_b, dw_prelu;        // LYNXNet / ConvNeXt depthwise params (float offsets)
    PackedGemm g_ain, g_aout;                        // ConvNeXt aux decoder: dense k-tap in/out convs
    // NSF-HiFiGAN generator
    dsd_vocoder_config vcfg;
    PackedGemm v_pre, v_post;
    std::vector<PackedGemm> v_ups;                   // transposed convs as phase-row GEMMs
    std::vector<std::vector<PackedGemm>> v_res;      // [stage * n_kernels + j][2 * n_dil (ResBlock1) or n_dil]
    std::vector<std::vector<PackedTConv>> v_rest;    // same indexing: the 16- / 32-channel stages (tconv.hip)
    PackedTConv v_postt;
    std::vector<size_t> v_nw, v_nb;                  // noise conv weights / biases
    std::vector<int> v_uptaps;
    size_t v_linw = 0, v_linb = 0;
    int vB = 0, vT = 0;
    float* v_arena = nullptr;
    std::vector<float*> v_buf;                       // per stage: x, t1, r, acc
    float *v_mel = nullptr, *v_pre_out = nullptr, *v_har = nullptr, *v_phase = nullptr, *v_wav = nullptr;
    // FastSpeech2 acoustic encoder
    dsd_encoder_config ecfg;
    std::vector<PackedGemm> g_qkv, g_oproj, g_ffn1, g_ffn2;
    std::vector<size_t> e_ln1g, e_ln1b, e_ln2g, e_ln2b;
    size_t e_lng = 0, e_lnb = 0, e_txt = 0, e_lang = SIZE_MAX, e_durw = 0, e_durb = 0, e_freqs = 0, e_spk = SIZE_MAX;
    size_t e_linw[7], e_linb[7];                     // pitch, energy, breathiness, voicing, tension, key shift, speed
    int eL = 0, eLs = 0, eB = 0, e_pos = 0;
    int e_ffn_act = DSD_FFN_GELU;                    // TransformerFFNLayer's activation (DSD_FFN_*)
    float *e_x = nullptr, *e_y = nullptr, *e_qkv = nullptr, *e_mid = nullptr, *e_nonpad = nullptr;
    int* e_dur = nullptr;
    float* e_arena = nullptr;
    // token encoder (variance model): FastSpeech2Encoder + out_proj / DurationPredictor
    dsd_token_encoder_config tcfg;
    PackedGemm g_tout;
    std::vector<PackedGemm> g_dconv;
    std::vector<size_t> d_lng, d_lnb;
    size_t d_linw = 0, d_linb = 0;
    float *d_a = nullptr, *d_b = nullptr, *d_in = nullptr;
    int dC = 0;
    size_t freqs_off = 0;
    int emb_act = ACT_MISH;

    // workspace for (B, T)
    int B = 0, T = 0, Ts = 0;
    float* arena = nullptr;
    size_t arena_floats = 0, arena_cap = 0, state_cap = 0, e_cap = 0, v_cap = 0;
    float *cond_i = nullptr, *cp = nullptr, *xh = nullptr, *z = nullptr, *skip = nullptr, *hbuf = nullptr;
    float *xin = nullptr, *ubuf = nullptr, *vbuf = nullptr, *stats = nullptr, *lnpart = nullptr;
    float *io_in = nullptr, *io_out = nullptr;
    bool cond_ready = false;
    // ragged batches (dsd_set_lengths): per-item valid lengths on the device, nullptr = dense
    int* lens_dev = nullptr;
    int lens_cap = 0;
    std::vector<int> lens_host;
    // ... and, per tile width (16 / 32 / 64 frames), the list of column groups (item, frame tile) with valid frames
    int* cg_dev[3] = {nullptr, nullptr, nullptr};
    int cg_cap[3] = {0, 0, 0}, cg_n[3] = {0, 0, 0};
    std::vector<int> cg_host[3];
    int cg_T = -1;                  // the T the lists were built for (-1: stale)
    bool use_cg = false;            // set around the launch sequences that may skip padded tiles
    // sampler state buffers
    float* state = nullptr;
    int state_nbufs = 0;
    size_t state_buf_floats = 0;
    // step-embedding tables (columns = steps or batch items)
    float* emb_arena = nullptr;
    int emb_cols = 0, Ns = 0;
    float *t_dev = nullptr, *E = nullptr, *Hd = nullptr, *E2 = nullptr, *D = nullptr;
    float* Dt = nullptr;            // D transposed: [step column][L * C rows] - what the layer kernels read their FiLM vectors from
    std::vector<float> t_host;

    std::map<std::string, GraphEntry> graphs;
    std::set<std::string> graph_seen;      // programs run once eagerly: a graph is captured when one comes back

    // timing of the layer kernels (dsd_kernel_timing): per kernel CLASS - a launch site of run_backbone and the variant of
    // it that ran (tile width, halo, segment of a mixed plan) - every timing_stride-th launch carries an event pair
    struct TimedClass {
        int key = 0;
        std::string name;           // the instantiation, as rocprofv3 prints it (filled in by the launcher that took the slot)
        double flops = 0, bytes = 0;    // algorithmic work of one launch (valid frames)
        long launches = 0;          // all launches of the class since timing was switched on
        std::vector<size_t> evs;    // indices into ev_pool
    };
    bool timing = false;
    std::vector<TimedClass> tclasses;
    long timing_evals = 0;         // backbone evaluations since timing was switched on
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_pool;
    size_t ev_used = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> cal_pool;   // back-to-back pairs: the cost of the bracket itself
    size_t cal_used = 0;
    int timing_stride = 7;         // coprime with the layer count: every layer is sampled over a pass
};

namespace {

int fail(dsd_handle* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (h) h->err = buf;
    else g_create_error = buf;
    return code;
}

#define HIP_OK(h, expr)                                                                          \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) return fail(h, DSD_EHIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

void destroy_graphs(dsd_handle* h);

// the launch sequences that may skip padded tiles (denoiser evaluations, aux decoder) run inside one of these
struct RaggedScope {
    dsd_handle* h;
    explicit RaggedScope(dsd_handle* hh) : h(hh) { h->use_cg = true; }
    ~RaggedScope() { h->use_cg = false; }
};

// ragged batch bookkeeping: the lengths given by dsd_set_lengths must describe this call's batch; the lists of valid
// column groups are (re)built here, on the caller's stream and outside any graph capture
inline int check_lens(dsd_handle* h, const char* who, int B, int T, hipStream_t st) {
    if (h->lens_host.empty()) return DSD_OK;
    if ((int)h->lens_host.size() != B)
        return fail(h, DSD_ESTATE, "%s: dsd_set_lengths gave %zu lengths but this call runs a batch of %d", who, h->lens_host.size(), B);
    for (int v : h->lens_host)
        if (v > T) return fail(h, DSD_EINVAL, "%s: a length (%d) exceeds T = %d", who, v, T);
    if (h->cg_T == T) return DSD_OK;
    for (int k = 0; k < 3; ++k) {
        const int BN = k == 0 ? 16 : 32 * k, tiles = (T + BN - 1) / BN;
        std::vector<int>& v = h->cg_host[k];
        v.clear();
        for (int b = 0; b < B; ++b)
            for (int ft = 0; ft * BN < h->lens_host[b]; ++ft) v.push_back(b * tiles + ft);
        h->cg_n[k] = (int)v.size();
        if ((int)v.size() > h->cg_cap[k]) {
            if (h->cg_dev[k]) (void)hipFree(h->cg_dev[k]);
            h->cg_dev[k] = nullptr;
            h->cg_cap[k] = 0;
            const size_t cap = (size_t)B * tiles;
            if (hipMalloc(&h->cg_dev[k], sizeof(int) * cap) != hipSuccess)
                return fail(h, DSD_ENOMEM, "%s: hipMalloc of %zu tile indices failed", who, cap);
            h->cg_cap[k] = (int)cap;
            destroy_graphs(h);      // cached graphs captured the old pointer
        }
        if (!v.empty() && hipMemcpyAsync(h->cg_dev[k], v.data(), sizeof(int) * v.size(), hipMemcpyHostToDevice, st) != hipSuccess)
            return fail(h, DSD_EHIP, "%s: upload of the valid-tile list failed", who);
    }
    h->cg_T = T;
    return DSD_OK;
}

inline int C_of(const dsd_handle* h) { return h->cfg.num_channels; }
inline int FM_of(const dsd_handle* h) { return h->cfg.in_dims * h->cfg.n_feats; }
inline int L_of(const dsd_handle* h) { return h->cfg.num_layers; }
inline int inner_of(const dsd_handle* h) { return h->cfg.num_channels * h->cfg.expansion_factor; }
inline bool is_wavenet(const dsd_handle* h) { return h->cfg.backbone == DSD_BACKBONE_WAVENET; }
inline bool is_aux(const dsd_handle* h) { return h->cfg.backbone == DSD_AUX_CONVNEXT; }
inline bool is_enc(const dsd_handle* h) { return h->cfg.backbone == DSD_ENC_FS2_ACOUSTIC; }
inline bool is_tok(const dsd_handle* h) { return h->cfg.backbone == DSD_ENC_FS2_TOKENS; }
inline bool is_voc(const dsd_handle* h) { return h->cfg.backbone == DSD_VOC_NSF_HIFIGAN; }

inline int voc_stage_channels(const dsd_vocoder_config& v, int i) { return v.upsample_initial_channel >> (i + 1); }
inline long voc_upp(const dsd_vocoder_config& v, int from) {     // product of upsample_rates[from:]
    long p = 1;
    for (int i = from; i < v.n_ups; ++i) p *= v.upsample_rates[i];
    return p;
}

// Generator state_dict after remove_weight_norm()  (nsf_hifigan/models.py:207-260)
std::vector<std::pair<std::string, std::vector<int64_t>>> expected_params_voc(const dsd_vocoder_config& v) {
    std::vector<std::pair<std::string, std::vector<int64_t>>> out;
    auto add = [&](const std::string& n, std::vector<int64_t> s) { out.emplace_back(n, std::move(s)); };
    if (!v.mini_nsf) {
        add("m_source.l_linear.weight", {1, v.harmonic_num + 1});
        add("m_source.l_linear.bias", {1});
    }
    add("conv_pre.weight", {v.upsample_initial_channel, v.num_mels, 7});
    add("conv_pre.bias", {v.upsample_initial_channel});
    for (int i = 0; i < v.n_ups; ++i) {
        const int64_t ch = voc_stage_channels(v, i);
        add("ups." + std::to_string(i) + ".weight", {2 * ch, ch, v.upsample_kernel_sizes[i]});
        add("ups." + std::to_string(i) + ".bias", {ch});
        if (v.mini_nsf) {
            if (i == 1) {
                add("source_conv.weight", {ch, 1, 1});
                add("source_conv.bias", {ch});
            }
        } else {
            const int64_t nk = i + 1 < v.n_ups ? 2 * voc_upp(v, i + 1) : 1;
            add("noise_convs." + std::to_string(i) + ".weight", {ch, 1, nk});
            add("noise_convs." + std::to_string(i) + ".bias", {ch});
        }
        for (int j = 0; j < v.n_kernels; ++j) {
            const std::string p = "resblocks." + std::to_string(i * v.n_kernels + j) + ".";
            for (int d = 0; d < v.n_dilations[j]; ++d) {
                const int64_t k = v.resblock_kernel_sizes[j];
                if (v.resblock == 1) {
                    add(p + "convs1." + std::to_string(d) + ".weight", {ch, ch, k});
                    add(p + "convs1." + std::to_string(d) + ".bias", {ch});
                    add(p + "convs2." + std::to_string(d) + ".weight", {ch, ch, k});
                    add(p + "convs2." + std::to_string(d) + ".bias", {ch});
                } else {
                    add(p + "convs." + std::to_string(d) + ".weight", {ch, ch, k});
                    add(p + "convs." + std::to_string(d) + ".bias", {ch});
                }
            }
        }
    }
    add("conv_post.weight", {1, voc_stage_channels(v, v.n_ups - 1), 7});
    add("conv_post.bias", {1});
    return out;
}

const char* const kLinNames[7] = {"pitch_embed", "variance_embeds.energy", "variance_embeds.breathiness",
                                  "variance_embeds.voicing", "variance_embeds.tension", "key_shift_embed", "speed_embed"};
inline bool lin_present(const dsd_encoder_config& e, int k) {
    if (k == 0) return true;
    const uint32_t bit[7] = {0, DSD_EMBED_ENERGY, DSD_EMBED_BREATHINESS, DSD_EMBED_VOICING, DSD_EMBED_TENSION,
                             DSD_EMBED_KEY_SHIFT, DSD_EMBED_SPEED};
    return (e.embed_flags & bit[k]) != 0;
}

// FastSpeech2Encoder state_dict (tts_modules.py:353-383; common_layers.py:120-234)
void expected_fs2_layers(std::vector<std::pair<std::string, std::vector<int64_t>>>& v, int64_t H, int layers, int heads,
                         int64_t ks, int pos_mode, int ffn_act) {
    const int64_t F1 = (ffn_act == DSD_FFN_SWIGLU ? 8 : 4) * H;       // SwiGLU: filter_size * 2 (common_layers.py:134)
    auto add = [&](const std::string& n, std::vector<int64_t> s) { v.emplace_back(n, std::move(s)); };
    if (pos_mode == DSD_POS_REL) add("encoder.embed_positions.div_term", {H / 2});
    if (pos_mode == DSD_POS_SIN) add("encoder.embed_positions.freqs", {H / 2});
    for (int l = 0; l < layers; ++l) {
        const std::string p = "encoder.layers." + std::to_string(l) + ".op.";
        add(p + "layer_norm1.weight", {H});
        add(p + "layer_norm1.bias", {H});
        if (pos_mode == DSD_POS_ROPE) {
            add(p + "self_attn.in_proj.weight", {3 * H, H});
            add(p + "self_attn.rotary_embed.freqs", {H / heads / 2});
        } else {      // torch.nn.MultiheadAttention(bias=False)  (common_layers.py:222-226)
            add(p + "self_attn.in_proj_weight", {3 * H, H});
        }
        add(p + "self_attn.out_proj.weight", {H, H});
        add(p + "layer_norm2.weight", {H});
        add(p + "layer_norm2.bias", {H});
        add(p + "ffn.ffn_1.weight", {F1, H, ks});
        add(p + "ffn.ffn_1.bias", {F1});
        add(p + "ffn.ffn_2.weight", {H, 4 * H});
        add(p + "ffn.ffn_2.bias", {H});
    }
    add("encoder.layer_norm.weight", {H});
    add("encoder.layer_norm.bias", {H});
}

// token encoder of the variance model: FastSpeech2Encoder + MelodyEncoder.out_proj (variance_encoder.py:126) and / or
// DurationPredictor (tts_modules.py:77-100: Sequential(Identity, Conv1d, ReLU, LayerNorm, Dropout) x n, Linear(C, 1))
std::vector<std::pair<std::string, std::vector<int64_t>>> expected_params_tok(const dsd_token_encoder_config& t) {
    std::vector<std::pair<std::string, std::vector<int64_t>>> v;
    const int64_t H = t.hidden_size;
    auto add = [&](const std::string& n, std::vector<int64_t> s) { v.emplace_back(n, std::move(s)); };
    expected_fs2_layers(v, H, t.enc_layers, t.num_heads, t.ffn_kernel_size, t.pos_mode, t.ffn_act);
    if (t.out_dims > 0) {
        add("out_proj.weight", {t.out_dims, H});
        add("out_proj.bias", {t.out_dims});
    }
    for (int l = 0; l < t.dur_layers; ++l) {
        const std::string p = "dur_predictor.conv." + std::to_string(l) + ".";
        add(p + "1.weight", {t.dur_chans, l == 0 ? H : (int64_t)t.dur_chans, t.dur_kernel_size});
        add(p + "1.bias", {t.dur_chans});
        add(p + "3.weight", {t.dur_chans});
        add(p + "3.bias", {t.dur_chans});
    }
    if (t.dur_layers > 0) {
        add("dur_predictor.linear.weight", {1, t.dur_chans});
        add("dur_predictor.linear.bias", {1});
    }
    return v;
}

// FastSpeech2Acoustic state_dict (acoustic_encoder.py:15-63; tts_modules.py:353-383; common_layers.py:120-234)
std::vector<std::pair<std::string, std::vector<int64_t>>> expected_params_enc(const dsd_encoder_config& e) {
    std::vector<std::pair<std::string, std::vector<int64_t>>> v;
    const int64_t H = e.hidden_size, ks = e.ffn_kernel_size;
    auto add = [&](const std::string& n, std::vector<int64_t> s) { v.emplace_back(n, std::move(s)); };
    add("txt_embed.weight", {e.vocab_size, H});
    if (e.num_lang > 0) add("lang_embed.weight", {e.num_lang + 1, H});
    add("dur_embed.weight", {H, 1});
    add("dur_embed.bias", {H});
    expected_fs2_layers(v, H, e.enc_layers, e.num_heads, ks, e.pos_mode, e.ffn_act);
    for (int k = 0; k < 7; ++k)
        if (lin_present(e, k)) {
            add(std::string(kLinNames[k]) + ".weight", {H, 1});
            add(std::string(kLinNames[k]) + ".bias", {H});
        }
    if (e.num_spk > 0) add("spk_embed.weight", {e.num_spk, H});
    return v;
}
inline int cp_rows(const dsd_handle* h) { return is_wavenet(h) ? 2 * C_of(h) : C_of(h); }

// ------------------------------------------------------------------------------------------
// expected parameters (reference state_dict names and shapes)
// ------------------------------------------------------------------------------------------
std::vector<std::pair<std::string, std::vector<int64_t>>> expected_params(const dsd_config& c) {
    std::vector<std::pair<std::string, std::vector<int64_t>>> v;
    const int64_t C = c.num_channels, M = (int64_t)c.in_dims * c.n_feats, H = c.hidden_size;
    auto add = [&](const std::string& n, std::vector<int64_t> s) { v.emplace_back(n, std::move(s)); };
    if (c.backbone == DSD_AUX_CONVNEXT) {      // modules/aux_decoder/convnext.py:58-76
        const int64_t ks = c.kernel_size;
        add("inconv.weight", {C, H, ks});
        add("inconv.bias", {C});
        for (int l = 0; l < c.num_layers; ++l) {
            const std::string p = "conv." + std::to_string(l) + ".";
            add(p + "gamma", {C});
            add(p + "dwconv.weight", {C, 1, 7});
            add(p + "dwconv.bias", {C});
            add(p + "norm.weight", {C});
            add(p + "norm.bias", {C});
            add(p + "pwconv1.weight", {4 * C, C});
            add(p + "pwconv1.bias", {4 * C});
            add(p + "pwconv2.weight", {C, 4 * C});
            add(p + "pwconv2.bias", {C});
        }
        add("outconv.weight", {M, C, ks});
        add("outconv.bias", {M});
        return v;
    }
    add("input_projection.weight", {C, M, 1});
    add("input_projection.bias", {C});
    if (c.backbone == DSD_BACKBONE_WAVENET) {
        add("mlp.0.weight", {4 * C, C});
        add("mlp.0.bias", {4 * C});
        add("mlp.2.weight", {C, 4 * C});
        add("mlp.2.bias", {C});
        for (int l = 0; l < c.num_layers; ++l) {
            const std::string p = "residual_layers." + std::to_string(l) + ".";
            add(p + "dilated_conv.weight", {2 * C, C, 3});
            add(p + "dilated_conv.bias", {2 * C});
            add(p + "diffusion_projection.weight", {C, C});
            add(p + "diffusion_projection.bias", {C});
            add(p + "conditioner_projection.weight", {2 * C, H, 1});
            add(p + "conditioner_projection.bias", {2 * C});
            add(p + "output_projection.weight", {2 * C, C, 1});
            add(p + "output_projection.bias", {2 * C});
        }
        add("skip_projection.weight", {C, C, 1});
        add("skip_projection.bias", {C});
    } else {
        const int64_t inner = C * c.expansion_factor;
        add("diffusion_embedding.1.weight", {4 * C, C});
        add("diffusion_embedding.1.bias", {4 * C});
        add("diffusion_embedding.3.weight", {C, 4 * C});
        add("diffusion_embedding.3.bias", {C});
        for (int l = 0; l < c.num_layers; ++l) {
            const std::string p = "residual_layers." + std::to_string(l) + ".";
            add(p + "diffusion_projection.weight", {C, C, 1});
            add(p + "diffusion_projection.bias", {C});
            add(p + "conditioner_projection.weight", {C, H, 1});
            add(p + "conditioner_projection.bias", {C});
            add(p + "convmodule.net.0.weight", {C});
            add(p + "convmodule.net.0.bias", {C});
            add(p + "convmodule.net.2.weight", {2 * inner, C, 1});
            add(p + "convmodule.net.2.bias", {2 * inner});
            add(p + "convmodule.net.4.weight", {inner, 1, (int64_t)c.kernel_size});
            add(p + "convmodule.net.4.bias", {inner});
            if (c.activation == DSD_ACT_PRELU) add(p + "convmodule.net.5.weight", {inner});
            add(p + "convmodule.net.6.weight", {C, inner, 1});
            add(p + "convmodule.net.6.bias", {C});
        }
        add("norm.weight", {C});
        add("norm.bias", {C});
    }
    add("output_projection.weight", {M, C, 1});
    add("output_projection.bias", {M});
    return v;
}

// ------------------------------------------------------------------------------------------
// weight packing into MFMA 16x16x4 fragment order:  [mblk][tap*K16 + k16][lane][j]
//   = W[rowmap(mblk*16 + (lane & 15))][k = k16*16 + j*4 + (lane >> 4)][tap]
// ------------------------------------------------------------------------------------------
using WGet = std::function<double(int row, int k, int tap)>;

size_t blob_reserve(dsd_handle* h, size_t n) {
    size_t off = (h->blob_host.size() + 63) / 64 * 64;
    h->blob_host.resize(off + n, 0.f);
    return off;
}

PackedGemm pack_gemm(dsd_handle* h, int M, int Kreal, int taps, int pairC, const WGet& w,
                     const std::function<double(int)>* bias) {
    PackedGemm g;
    g.M = M;
    g.Kreal = Kreal;
    g.K = round_up(Kreal, 16);
    g.taps = taps;
    g.pairC = pairC;
    const int K16 = g.K / 16;
    const int ntile = pairC > 0 ? (pairC + 31) / 32 : (M + 63) / 64;
    const int mblks = ntile * 4;
    g.a_off = blob_reserve(h, (size_t)mblks * taps * K16 * 256);
    float* dst = h->blob_host.data() + g.a_off;
    // block order = the order the K walk consumes them: [64-channel chunk][tap][k16 within the chunk]
    std::vector<std::pair<int, int>> order;      // (tap, k16)
    for (int c0 = 0; c0 < K16; c0 += 4)
        for (int tap = 0; tap < taps; ++tap)
            for (int k = c0; k < std::min(c0 + 4, K16); ++k) order.emplace_back(tap, k);
    for (int mblk = 0; mblk < mblks; ++mblk)
        for (int kk = 0; kk < taps * K16; ++kk) {
            const int tap = order[kk].first, k16 = order[kk].second;
            float* blk = dst + ((size_t)mblk * taps * K16 + kk) * 256;
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 4; ++j) {
                    const int pr = mblk * 16 + (lane & 15);
                    const int k = k16 * 16 + j * 4 + (lane >> 4);
                    int orig;
                    if (pairC > 0) {
                        const int tile = pr / 64, within = pr % 64;
                        const int wm = within / 32, mb = (within % 32) / 16, r = within % 16;
                        const int ch = (tile * 2 + wm) * 16 + r;
                        orig = ch < pairC ? mb * pairC + ch : -1;
                    } else {
                        orig = pr < M ? pr : -1;
                    }
                    blk[lane * 4 + j] = (orig >= 0 && k < Kreal) ? (float)w(orig, k, tap) : 0.f;
                }
        }
    g.bias_off = SIZE_MAX;
    if (bias) {
        g.bias_off = blob_reserve(h, (size_t)M);
        for (int i = 0; i < M; ++i) h->blob_host[g.bias_off + i] = (float)(*bias)(i);
    }
    return g;
}

const HostTensor& W(dsd_handle* h, const std::string& n) {
    auto it = h->padded.find(n);
    return it != h->padded.end() ? it->second : h->raw.at(n);
}
bool has_W(const dsd_handle* h, const std::string& n) { return h->padded.count(n) || h->raw.count(n); }

// A WaveNet of C channels, C not a multiple of 32, as the network of Cp = round_up(C, 32) channels whose extra channels are
// exactly zero everywhere: extra weight rows, columns and biases are zero, so the input projection gives relu(0) = 0 there,
// the FiLM shift is 0, the gate sigmoid(0) * tanh(0) = 0, the residual (0 + 0) / sqrt(2) = 0 and the skip 0 (wavenet.py:33-48);
// the real channels see only zero contributions from them.  Rows of the [2C] tensors are two halves (gate | filter,
// residual | skip): each half is extended on its own.  SinusoidalPosEmb keeps the frequencies of the REAL C
// (common_layers.py:275-276) in the first C/2 of Cp/2 slots; mlp.0's columns follow its [sin | cos] halves.
void pad_wavenet_weights(dsd_handle* h) {
    h->padded.clear();
    const int C = h->c_user, Cp = h->cfg.num_channels, L = h->cfg.num_layers;
    if (!is_wavenet(h) || C == 0 || C == Cp) return;
    enum Map { SAME, PLAIN, HALVES, QUAD, SINCOS };
    auto size_of = [&](Map m, int n) { return m == SAME ? n : m == PLAIN ? Cp : m == HALVES ? 2 * Cp : m == QUAD ? 4 * Cp : Cp; };
    auto map_of = [&](Map m, int i) {
        switch (m) {
            case HALVES: return i < C ? i : Cp + (i - C);
            case SINCOS: return i < C / 2 ? i : Cp / 2 + (i - C / 2);
            default: return i;
        }
    };
    auto pad = [&](const std::string& name, Map rm, Map cm) {
        const HostTensor& t = h->raw.at(name);
        const int rows = (int)t.shape[0], cols = t.shape.size() > 1 ? (int)t.shape[1] : 1;
        const int taps = t.shape.size() > 2 ? (int)t.shape[2] : 1;
        HostTensor o;
        o.shape = t.shape;
        o.shape[0] = size_of(rm, rows);
        if (t.shape.size() > 1) o.shape[1] = size_of(cm, cols);
        const int cp = t.shape.size() > 1 ? (int)o.shape[1] : 1;
        o.data.assign((size_t)o.shape[0] * cp * taps, 0.f);
        for (int r = 0; r < rows; ++r)
            for (int c = 0; c < cols; ++c)
                for (int k = 0; k < taps; ++k)
                    o.data[((size_t)map_of(rm, r) * cp + map_of(cm, c)) * taps + k] = t.data[((size_t)r * cols + c) * taps + k];
        h->padded[name] = std::move(o);
    };
    pad("input_projection.weight", PLAIN, SAME);
    pad("input_projection.bias", PLAIN, SAME);
    pad("mlp.0.weight", QUAD, SINCOS);
    pad("mlp.0.bias", QUAD, SAME);
    pad("mlp.2.weight", PLAIN, QUAD);
    pad("mlp.2.bias", PLAIN, SAME);
    for (int l = 0; l < L; ++l) {
        const std::string p = "residual_layers." + std::to_string(l) + ".";
        pad(p + "dilated_conv.weight", HALVES, PLAIN);
        pad(p + "dilated_conv.bias", HALVES, SAME);
        pad(p + "diffusion_projection.weight", PLAIN, PLAIN);
        pad(p + "diffusion_projection.bias", PLAIN, SAME);
        pad(p + "conditioner_projection.weight", HALVES, SAME);
        pad(p + "conditioner_projection.bias", HALVES, SAME);
        pad(p + "output_projection.weight", HALVES, PLAIN);
        pad(p + "output_projection.bias", HALVES, SAME);
    }
    pad("skip_projection.weight", PLAIN, PLAIN);
    pad("skip_projection.bias", PLAIN, SAME);
    pad("output_projection.weight", SAME, PLAIN);
    HostTensor f;                               // [Cp / 2]: the real C's table, then zeros (sin 0 = 0 and cos 0 = 1 meet zero columns)
    f.shape = {Cp / 2};
    f.data.assign((size_t)Cp / 2, 0.f);
    const int half = C / 2;
    if (h->raw.count("diffusion_embedding.freqs")) {
        const auto& src = h->raw.at("diffusion_embedding.freqs").data;
        for (int i = 0; i < half; ++i) f.data[i] = src[i];
    } else {
        const float step = -(float)(log(10000.0) / (half - 1));
        for (int i = 0; i < half; ++i) f.data[i] = expf((float)i * step);
    }
    h->padded["diffusion_embedding.freqs"] = std::move(f);
}

// ConvNeXt aux decoder (convnext.py:17-85).  LayerNorm affine folded into pwconv1, the layer scale gamma folded
// into pwconv2:  x + gamma * (W2 g + b2) = x + (diag(gamma) W2) g + gamma * b2.
int build_packed_aux(dsd_handle* h) {
    const dsd_config& c = h->cfg;
    const int C = c.num_channels, M = FM_of(h), H = c.hidden_size, L = c.num_layers, ks = c.kernel_size;
    h->blob_host.clear();
    auto dense = [&](const std::string& name, int rows, int cin) {
        const HostTensor* t = &W(h, name + ".weight");
        const HostTensor* b = &W(h, name + ".bias");
        WGet w = [t, cin, ks](int r, int k, int tap) { return (double)t->data[((size_t)r * cin + k) * ks + tap]; };
        std::function<double(int)> bf = [b](int i) { return (double)b->data[i]; };
        return pack_gemm(h, rows, cin, ks, 0, w, &bf);
    };
    h->g_ain = dense("inconv", C, H);
    h->g_aout = dense("outconv", M, C);
    h->g_pw1.resize(L);
    h->g_pw2.resize(L);
    h->dw_w.resize(L);
    h->dw_b.resize(L);
    for (int l = 0; l < L; ++l) {
        const std::string p = "conv." + std::to_string(l) + ".";
        const HostTensor* w1 = &W(h, p + "pwconv1.weight");
        const HostTensor* b1 = &W(h, p + "pwconv1.bias");
        const HostTensor* g = &W(h, p + "norm.weight");
        const HostTensor* be = &W(h, p + "norm.bias");
        WGet wf = [w1, g, C](int r, int k, int) { return (double)w1->data[(size_t)r * C + k] * (double)g->data[k]; };
        std::function<double(int)> bf = [w1, b1, be, C](int r) {
            double s = b1->data[r];
            for (int k = 0; k < C; ++k) s += (double)w1->data[(size_t)r * C + k] * (double)be->data[k];
            return s;
        };
        h->g_pw1[l] = pack_gemm(h, 4 * C, C, 1, 0, wf, &bf);
        const HostTensor* w2 = &W(h, p + "pwconv2.weight");
        const HostTensor* b2 = &W(h, p + "pwconv2.bias");
        const HostTensor* ga = &W(h, p + "gamma");
        WGet w2f = [w2, ga, C](int r, int k, int) { return (double)ga->data[r] * (double)w2->data[(size_t)r * 4 * C + k]; };
        std::function<double(int)> b2f = [b2, ga](int r) { return (double)ga->data[r] * (double)b2->data[r]; };
        h->g_pw2[l] = pack_gemm(h, C, 4 * C, 1, 0, w2f, &b2f);
        const auto& dw = W(h, p + "dwconv.weight").data;
        h->dw_w[l] = blob_reserve(h, (size_t)C * 7);
        memcpy(h->blob_host.data() + h->dw_w[l], dw.data(), sizeof(float) * C * 7);
        const auto& db = W(h, p + "dwconv.bias").data;
        h->dw_b[l] = blob_reserve(h, (size_t)C);
        memcpy(h->blob_host.data() + h->dw_b[l], db.data(), sizeof(float) * C);
    }
    return DSD_OK;
}

// FastSpeech2 acoustic encoder: 1x1 / k-tap GEMM operands in fragment order, everything else copied as is.
// The FFN's `x * kernel_size ** -0.5` (common_layers.py:146) is folded into ffn_1's weights and bias.
size_t blob_copy_vec(dsd_handle* h, const std::string& name) {
    const auto& d = W(h, name).data;
    const size_t off = blob_reserve(h, d.size());
    memcpy(h->blob_host.data() + off, d.data(), sizeof(float) * d.size());
    return off;
}

void pack_fs2_layers(dsd_handle* h, int H, int L, int ks, int pos_mode);

int build_packed_enc(dsd_handle* h) {
    const dsd_encoder_config& e = h->ecfg;
    const int H = e.hidden_size, L = e.enc_layers, ks = e.ffn_kernel_size;
    h->blob_host.clear();
    auto copy_vec = [&](const std::string& name) { return blob_copy_vec(h, name); };
    h->e_txt = copy_vec("txt_embed.weight");
    h->e_lang = e.num_lang > 0 ? copy_vec("lang_embed.weight") : SIZE_MAX;
    h->e_durw = copy_vec("dur_embed.weight");
    h->e_durb = copy_vec("dur_embed.bias");
    h->e_spk = e.num_spk > 0 ? copy_vec("spk_embed.weight") : SIZE_MAX;
    for (int k = 0; k < 7; ++k) {
        h->e_linw[k] = h->e_linb[k] = SIZE_MAX;
        if (lin_present(e, k)) {
            h->e_linw[k] = copy_vec(std::string(kLinNames[k]) + ".weight");
            h->e_linb[k] = copy_vec(std::string(kLinNames[k]) + ".bias");
        }
    }
    pack_fs2_layers(h, H, L, ks, e.pos_mode);
    return DSD_OK;
}

// token encoder: the shared layers, then out_proj as a 1x1 GEMM and the duration predictor's convolutions as k-tap GEMMs
int build_packed_tok(dsd_handle* h) {
    const dsd_token_encoder_config& t = h->tcfg;
    const int H = t.hidden_size;
    h->blob_host.clear();
    pack_fs2_layers(h, H, t.enc_layers, t.ffn_kernel_size, t.pos_mode);
    if (t.out_dims > 0) {
        const HostTensor* w = &W(h, "out_proj.weight");
        const HostTensor* b = &W(h, "out_proj.bias");
        std::function<double(int)> bf = [b](int i) { return (double)b->data[i]; };
        h->g_tout = pack_gemm(h, t.out_dims, H, 1, 0, WGet([w, H](int r, int k, int) { return (double)w->data[(size_t)r * H + k]; }), &bf);
    }
    const int Cd = t.dur_chans, kd = t.dur_kernel_size;
    h->g_dconv.resize(t.dur_layers); h->d_lng.resize(t.dur_layers); h->d_lnb.resize(t.dur_layers);
    for (int l = 0; l < t.dur_layers; ++l) {
        const std::string p = "dur_predictor.conv." + std::to_string(l) + ".";
        const int cin = l == 0 ? H : Cd;
        const HostTensor* w = &W(h, p + "1.weight");
        const HostTensor* b = &W(h, p + "1.bias");
        std::function<double(int)> bf = [b](int i) { return (double)b->data[i]; };
        h->g_dconv[l] = pack_gemm(h, Cd, cin, kd, 0,
                                  WGet([w, cin, kd](int r, int k, int tap) { return (double)w->data[((size_t)r * cin + k) * kd + tap]; }), &bf);
        h->d_lng[l] = blob_copy_vec(h, p + "3.weight");
        h->d_lnb[l] = blob_copy_vec(h, p + "3.bias");
    }
    if (t.dur_layers > 0) {
        h->d_linw = blob_copy_vec(h, "dur_predictor.linear.weight");
        h->d_linb = blob_copy_vec(h, "dur_predictor.linear.bias");
    }
    return DSD_OK;
}

void pack_fs2_layers(dsd_handle* h, int H, int L, int ks, int pos_mode) {
    auto copy_vec = [&](const std::string& name) { return blob_copy_vec(h, name); };
    h->e_pos = pos_mode;
    h->e_freqs = SIZE_MAX;
    if (pos_mode == DSD_POS_ROPE) h->e_freqs = copy_vec("encoder.layers.0.op.self_attn.rotary_embed.freqs");     // one shared RotaryEmbedding
    if (pos_mode == DSD_POS_REL) h->e_freqs = copy_vec("encoder.embed_positions.div_term");
    if (pos_mode == DSD_POS_SIN) h->e_freqs = copy_vec("encoder.embed_positions.freqs");
    h->e_lng = copy_vec("encoder.layer_norm.weight");
    h->e_lnb = copy_vec("encoder.layer_norm.bias");
    h->g_qkv.resize(L); h->g_oproj.resize(L); h->g_ffn1.resize(L); h->g_ffn2.resize(L);
    h->e_ln1g.resize(L); h->e_ln1b.resize(L); h->e_ln2g.resize(L); h->e_ln2b.resize(L);
    const double fscale = 1.0 / sqrt((double)ks);
    for (int l = 0; l < L; ++l) {
        const std::string p = "encoder.layers." + std::to_string(l) + ".op.";
        h->e_ln1g[l] = copy_vec(p + "layer_norm1.weight");
        h->e_ln1b[l] = copy_vec(p + "layer_norm1.bias");
        h->e_ln2g[l] = copy_vec(p + "layer_norm2.weight");
        h->e_ln2b[l] = copy_vec(p + "layer_norm2.bias");
        auto lin = [&](const std::string& name, int cols) {
            const HostTensor* t = &W(h, name);
            return WGet([t, cols](int r, int k, int) { return (double)t->data[(size_t)r * cols + k]; });
        };
        h->g_qkv[l] = pack_gemm(h, 3 * H, H, 1, 0,
                                lin(p + (pos_mode == DSD_POS_ROPE ? "self_attn.in_proj.weight" : "self_attn.in_proj_weight"), H), nullptr);
        h->g_oproj[l] = pack_gemm(h, H, H, 1, 0, lin(p + "self_attn.out_proj.weight", H), nullptr);
        const HostTensor* w1 = &W(h, p + "ffn.ffn_1.weight");
        const HostTensor* b1 = &W(h, p + "ffn.ffn_1.bias");
        // float multiply like the reference (x * k**-0.5 in fp32), then packed
        const float fs = (float)fscale;
        WGet w1f = [w1, H, ks, fs](int r, int k, int tap) { return (double)(w1->data[((size_t)r * H + k) * ks + tap] * fs); };
        std::function<double(int)> b1f = [b1, fs](int i) { return (double)(b1->data[i] * fs); };
        h->g_ffn1[l] = pack_gemm(h, (h->e_ffn_act == DSD_FFN_SWIGLU ? 8 : 4) * H, H, ks, 0, w1f, &b1f);
        const HostTensor* b2 = &W(h, p + "ffn.ffn_2.bias");
        std::function<double(int)> b2f = [b2](int i) { return (double)b2->data[i]; };
        h->g_ffn2[l] = pack_gemm(h, H, 4 * H, 1, 0, lin(p + "ffn.ffn_2.weight", 4 * H), &b2f);
    }
}

// NSF-HiFiGAN generator.  A ConvTranspose1d(C_in, C_out, K, stride u, padding (K-u)/2) becomes an ordinary odd-tap
// convolution at the INPUT resolution with u * C_out output rows - row r*C_out + o is output phase r of channel o:
//   y[o][u q + r] = sum_i sum_n W[i][o][k0 + u n] x[i][q + e - n],   k0 = (r + pad) mod u,  e = (r + pad) div u
// so tap a (input offset d = a - D) of phase row (r, o) holds W[i][o][k0 + u (e - d)] when that index is a valid
// kernel position and zero otherwise; the GEMM's scatter epilogue writes column q of row (r, o) to out[o][u q + r].
int build_packed_voc(dsd_handle* h) {
    const dsd_vocoder_config& v = h->vcfg;
    h->blob_host.clear();
    // few-channel convolutions: B fragments [tap][ci/4][co/16][lane] (tconv.hip)
    auto tconv_pack = [&](const std::string& name, int co_real, int ci, int ks) {
        const HostTensor* t = &W(h, name + ".weight");     // [co_real, ci, ks]
        const HostTensor* b = &W(h, name + ".bias");
        PackedTConv pt;
        pt.ci = ci; pt.co = round_up(co_real, 16); pt.co_real = co_real; pt.taps = ks;
        const int kc4 = ci / 4, nbn = pt.co / 16;
        pt.w_off = blob_reserve(h, (size_t)ks * kc4 * nbn * 64);
        float* dst = h->blob_host.data() + pt.w_off;
        for (int tap = 0; tap < ks; ++tap)
            for (int c4 = 0; c4 < kc4; ++c4)
                for (int nb = 0; nb < nbn; ++nb)
                    for (int l = 0; l < 64; ++l) {
                        const int o = nb * 16 + (l & 15), c = c4 * 4 + (l >> 4);
                        dst[((size_t)(tap * kc4 + c4) * nbn + nb) * 64 + l] =
                            o < co_real ? t->data[((size_t)o * ci + c) * ks + tap] : 0.f;
                    }
        pt.b_off = blob_reserve(h, (size_t)co_real);
        memcpy(h->blob_host.data() + pt.b_off, b->data.data(), sizeof(float) * co_real);
        return pt;
    };
    auto few = [](int ch) { return ch == 16 || ch == 32; };
    auto copy_vec = [&](const std::string& name) {
        const auto& d = W(h, name).data;
        const size_t off = blob_reserve(h, d.size());
        memcpy(h->blob_host.data() + off, d.data(), sizeof(float) * d.size());
        return off;
    };
    auto dense = [&](const std::string& name, int rows, int cin, int ks) {
        const HostTensor* t = &W(h, name + ".weight");
        const HostTensor* b = &W(h, name + ".bias");
        WGet w = [t, cin, ks](int r, int k, int tap) { return (double)t->data[((size_t)r * cin + k) * ks + tap]; };
        std::function<double(int)> bf = [b](int i) { return (double)b->data[i]; };
        return pack_gemm(h, rows, cin, ks, 0, w, &bf);
    };
    if (!v.mini_nsf) {
        h->v_linw = copy_vec("m_source.l_linear.weight");
        h->v_linb = copy_vec("m_source.l_linear.bias");
    }
    h->v_pre = dense("conv_pre", v.upsample_initial_channel, v.num_mels, 7);
    h->v_ups.resize(v.n_ups);
    h->v_uptaps.resize(v.n_ups);
    h->v_nw.resize(v.n_ups);
    h->v_nb.resize(v.n_ups);
    h->v_res.assign((size_t)v.n_ups * v.n_kernels, {});
    h->v_rest.assign((size_t)v.n_ups * v.n_kernels, {});
    for (int i = 0; i < v.n_ups; ++i) {
        const int ch = voc_stage_channels(v, i), u = v.upsample_rates[i], K = v.upsample_kernel_sizes[i], pad = (K - u) / 2;
        int D = 0;
        for (int r = 0; r < u; ++r) {
            const int k0 = (r + pad) % u, e = (r + pad) / u, nmax = (K - 1 - k0) / u;
            D = std::max(D, std::max(e, nmax - e));
        }
        const int taps = 2 * D + 1;
        h->v_uptaps[i] = taps;
        const HostTensor* t = &W(h, "ups." + std::to_string(i) + ".weight");      // [2ch, ch, K]
        const HostTensor* b = &W(h, "ups." + std::to_string(i) + ".bias");
        WGet w = [t, ch, u, K, pad, D](int row, int ci, int a) {
            const int r = row / ch, o = row % ch;
            const int k0 = (r + pad) % u, e = (r + pad) / u;
            const int n = e - (a - D);
            const int k = k0 + u * n;
            if (n < 0 || k >= K) return 0.0;
            return (double)t->data[((size_t)ci * ch + o) * K + k];
        };
        h->v_ups[i] = pack_gemm(h, u * ch, 2 * ch, taps, 0, w, nullptr);
        h->v_ups[i].bias_off = blob_reserve(h, (size_t)ch);
        memcpy(h->blob_host.data() + h->v_ups[i].bias_off, b->data.data(), sizeof(float) * ch);
        h->v_nw[i] = h->v_nb[i] = SIZE_MAX;
        if (!v.mini_nsf || i == 1) {   // noise conv weights transposed to [k][C]: the kernel's threads run along the channels
            const std::string nname = v.mini_nsf ? std::string("source_conv") : "noise_convs." + std::to_string(i);
            const auto& wn = W(h, nname + ".weight");     // [ch, 1, k]
            const int nk = (int)wn.shape[2];
            h->v_nw[i] = blob_reserve(h, (size_t)nk * ch);
            for (int k = 0; k < nk; ++k)
                for (int o = 0; o < ch; ++o) h->blob_host[h->v_nw[i] + (size_t)k * ch + o] = wn.data[(size_t)o * nk + k];
            h->v_nb[i] = copy_vec(nname + ".bias");
        }
        for (int j = 0; j < v.n_kernels; ++j) {
            const std::string p = "resblocks." + std::to_string(i * v.n_kernels + j) + ".";
            auto& list = h->v_res[(size_t)i * v.n_kernels + j];
            auto& tlist = h->v_rest[(size_t)i * v.n_kernels + j];
            for (int d = 0; d < v.n_dilations[j]; ++d) {
                if (few(ch)) {
                    const int ks = v.resblock_kernel_sizes[j];
                    if (v.resblock == 1) {
                        tlist.push_back(tconv_pack(p + "convs1." + std::to_string(d), ch, ch, ks));
                        tlist.push_back(tconv_pack(p + "convs2." + std::to_string(d), ch, ch, ks));
                    } else {
                        tlist.push_back(tconv_pack(p + "convs." + std::to_string(d), ch, ch, ks));
                    }
                    continue;
                }
                if (v.resblock == 1) {
                    list.push_back(dense(p + "convs1." + std::to_string(d), ch, ch, v.resblock_kernel_sizes[j]));
                    list.push_back(dense(p + "convs2." + std::to_string(d), ch, ch, v.resblock_kernel_sizes[j]));
                } else {
                    list.push_back(dense(p + "convs." + std::to_string(d), ch, ch, v.resblock_kernel_sizes[j]));
                }
            }
        }
    }
    h->v_post = dense("conv_post", 1, voc_stage_channels(v, v.n_ups - 1), 7);
    if (voc_stage_channels(v, v.n_ups - 1) == 16) h->v_postt = tconv_pack("conv_post", 1, 16, 7);
    return DSD_OK;
}

std::vector<std::pair<std::string, std::vector<int64_t>>> expected_for(const dsd_handle* h) {
    if (h->c_user && h->c_user != h->cfg.num_channels) {      // the state_dict has the caller's channel count
        dsd_config c = h->cfg;
        c.num_channels = h->c_user;
        return expected_params(c);
    }
    return is_enc(h) ? expected_params_enc(h->ecfg) : is_tok(h) ? expected_params_tok(h->tcfg)
         : is_voc(h) ? expected_params_voc(h->vcfg) : expected_params(h->cfg);
}

// bf16 (round to nearest even) of an fp32 value, as the 16 upper bits
inline uint16_t bf16_bits(float v) {
    uint32_t u;
    memcpy(&u, &v, 4);
    if ((u & 0x7F800000u) == 0x7F800000u) return (uint16_t)(u >> 16);      // inf / nan: truncate
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
inline float bf16_value(uint16_t b) {
    const uint32_t u = (uint32_t)b << 16;
    float v;
    memcpy(&v, &u, 4);
    return v;
}

// Weight stream of wn_layer_x3.hip for one GEMM of one layer: [wave 4][k32 step][row block 8][hi | lo][lane 64][8 bf16], the
// order a wave consumes it.  v_mfma_f32_16x16x32_bf16's A operand: lane l holds A[row l & 15][k = 8 (l >> 4) + j], j = 0..7.
// rows_of(block, m) = the original row of packed block `block` (0..31), row m; col_of(step, kk) = (input channel, tap) of k index
// kk (0..31) of k32 step `step`.
size_t pack_x3(dsd_handle* h, int nrt, int nsteps, const std::function<int(int, int, int)>& row_of,
               const std::function<std::pair<int, int>(int, int)>& col_of, const WGet& w) {
    // nrt row tiles of 4 waves x 8 row blocks; row_of(row tile, 8 * wave + k, m)
    const size_t nfloats = (size_t)nrt * 4 * nsteps * 8 * 2 * 64 * 8 / 2;
    const size_t off = blob_reserve(h, nfloats);
    uint16_t* dst = reinterpret_cast<uint16_t*>(h->blob_host.data() + off);
    for (int rt = 0; rt < nrt; ++rt)
        for (int wave = 0; wave < 4; ++wave)
            for (int s = 0; s < nsteps; ++s)
                for (int k = 0; k < 8; ++k) {
                    uint16_t* blk = dst + ((((size_t)rt * 4 + wave) * nsteps + s) * 8 + k) * 2 * 512;
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const int row = row_of(rt, 8 * wave + k, lane & 15);
                            const std::pair<int, int> ct = col_of(s, 8 * (lane >> 4) + j);
                            const float v = (float)w(row, ct.first, ct.second);
                            const uint16_t hi = bf16_bits(v);
                            blk[lane * 8 + j] = hi;
                            blk[512 + lane * 8 + j] = bf16_bits(v - bf16_value(hi));
                        }
                }
    return off;
}

int build_packed(dsd_handle* h) {
    if (is_aux(h)) return build_packed_aux(h);
    if (is_enc(h)) return build_packed_enc(h);
    if (is_tok(h)) return build_packed_tok(h);
    if (is_voc(h)) return build_packed_voc(h);
    const dsd_config& c = h->cfg;
    const int C = c.num_channels, M = FM_of(h), H = c.hidden_size, L = c.num_layers;
    h->blob_host.clear();
    pad_wavenet_weights(h);
    // frequency table of SinusoidalPosEmb (common_layers.py:275-276)
    h->freqs_off = blob_reserve(h, (size_t)C / 2);
    if (has_W(h, "diffusion_embedding.freqs")) {
        const auto& f = W(h, "diffusion_embedding.freqs").data;
        for (int i = 0; i < C / 2; ++i) h->blob_host[h->freqs_off + i] = f[i];
    } else {
        const int half = C / 2;
        const float step = -(float)(log(10000.0) / (half - 1));
        for (int i = 0; i < half; ++i) h->blob_host[h->freqs_off + i] = expf((float)i * step);
    }
    auto conv1 = [&](const std::string& name) {   // Conv1d k=1 / Linear weight [M, K(,1)]
        const HostTensor* t = &W(h, name);
        const int64_t K = t->shape[1];
        return WGet([t, K](int r, int k, int) { return (double)t->data[(size_t)r * K + k]; });
    };
    auto bias_of = [&](const std::string& name) {
        const HostTensor* t = &W(h, name);
        return std::function<double(int)>([t](int i) { return (double)t->data[i]; });
    };
    {
        auto b = bias_of("input_projection.bias");
        h->g_inproj = pack_gemm(h, C, M, 1, 0, conv1("input_projection.weight"), &b);
    }
    const bool wn = is_wavenet(h);
    const std::string e0 = wn ? "mlp.0" : "diffusion_embedding.1";
    const std::string e1 = wn ? "mlp.2" : "diffusion_embedding.3";
    h->emb_act = wn ? ACT_MISH : ACT_GELU;
    {
        auto b0 = bias_of(e0 + ".bias");
        h->g_emb0 = pack_gemm(h, 4 * C, C, 1, 0, conv1(e0 + ".weight"), &b0);
        auto b1 = bias_of(e1 + ".bias");
        h->g_emb1 = pack_gemm(h, C, 4 * C, 1, 0, conv1(e1 + ".weight"), &b1);
    }
    // all layers' diffusion_projection as ONE [L*C x C] GEMM over the step table
    {
        std::vector<const HostTensor*> ws(L), bs(L);
        for (int l = 0; l < L; ++l) {
            ws[l] = &W(h, "residual_layers." + std::to_string(l) + ".diffusion_projection.weight");
            bs[l] = &W(h, "residual_layers." + std::to_string(l) + ".diffusion_projection.bias");
        }
        WGet w = [ws, C](int r, int k, int) { return (double)ws[r / C]->data[(size_t)(r % C) * C + k]; };
        std::function<double(int)> b = [bs, C](int i) { return (double)bs[i / C]->data[i % C]; };
        h->g_dproj = pack_gemm(h, L * C, C, 1, 0, w, &b);
    }
    // all layers' conditioner_projection as ONE [L*R x H] GEMM over cond (R = 2C WaveNet, C LYNXNet);
    // WaveNet: the dilated conv bias is folded in here (both are added before the gate, wavenet.py:38)
    {
        const int R = cp_rows(h);
        std::vector<const HostTensor*> ws(L), bs(L), b2(L, nullptr);
        for (int l = 0; l < L; ++l) {
            const std::string p = "residual_layers." + std::to_string(l) + ".";
            ws[l] = &W(h, p + "conditioner_projection.weight");
            bs[l] = &W(h, p + "conditioner_projection.bias");
            if (wn) b2[l] = &W(h, p + "dilated_conv.bias");
        }
        WGet w = [ws, R, H](int r, int k, int) { return (double)ws[r / R]->data[(size_t)(r % R) * H + k]; };
        std::function<double(int)> b = [bs, b2, R](int i) {
            double v = bs[i / R]->data[i % R];
            if (b2[i / R]) v += b2[i / R]->data[i % R];
            return v;
        };
        h->g_cp = pack_gemm(h, L * R, H, 1, 0, w, &b);
    }
    if (wn) {
        h->g_conv.resize(L);
        h->g_outp.resize(L);
        for (int l = 0; l < L; ++l) {
            const std::string p = "residual_layers." + std::to_string(l) + ".";
            const HostTensor* t = &W(h, p + "dilated_conv.weight");    // [2C, C, 3]
            WGet w = [t, C](int r, int k, int tap) { return (double)t->data[((size_t)r * C + k) * 3 + tap]; };
            h->g_conv[l] = pack_gemm(h, 2 * C, C, 3, C, w, nullptr);
            auto b = bias_of(p + "output_projection.bias");
            h->g_outp[l] = pack_gemm(h, 2 * C, C, 1, 0, conv1(p + "output_projection.weight"), &b);
        }
        // split-bf16 mode: the same two matrices of every layer once more, split hi / lo, as wn_layer_x3.hip's streams
        h->x3_conv.clear();
        h->x3_out.clear();
        const int max_dil = 1 << (std::min(c.dilation_cycle_length, L) - 1);
        if (h->precision == 1 && c.dilation_cycle_length >= 1 && wn_layer_x3_supported(C, max_dil)) {
            h->x3_conv.resize(L);
            h->x3_out.resize(L);
            for (int l = 0; l < L; ++l) {
                const std::string p = "residual_layers." + std::to_string(l) + ".";
                const HostTensor* t = &W(h, p + "dilated_conv.weight");    // [2C, C, 3]
                WGet w = [t, C](int r, int k, int tap) { return (double)t->data[((size_t)r * C + k) * 3 + tap]; };
                // conv rows: block 2 q + gf = the gate (gf = 0) / filter (1) rows of channels [16 q, 16 q + 16); k32 step = [tap][chunk]
                h->x3_conv[l] = pack_x3(h, 1, 3 * C / 32, [C](int, int blk, int m) { return (blk & 1) * C + (blk >> 1) * 16 + m; },
                                        [](int s, int kk) { return std::make_pair((s & 7) * 32 + kk, s >> 3); }, w);
                const HostTensor* to = &W(h, p + "output_projection.weight");    // [2C, C, 1]
                WGet wo = [to, C](int r, int k, int) { return (double)to->data[(size_t)r * C + k]; };
                h->x3_out[l] = pack_x3(h, 1, C / 32, [](int, int blk, int m) { return blk * 16 + m; },
                                       [](int s, int kk) { return std::make_pair(s * 32 + kk, 0); }, wo);
            }
        }
        auto b1 = bias_of("skip_projection.bias");
        h->g_tail1 = pack_gemm(h, C, C, 1, 0, conv1("skip_projection.weight"), &b1);
        auto b2 = bias_of("output_projection.bias");
        h->g_out = pack_gemm(h, M, C, 1, 0, conv1("output_projection.weight"), &b2);
    } else {
        const int inner = inner_of(h), ks = c.kernel_size;
        h->g_pw1.resize(L);
        h->g_pw2.resize(L);
        h->dw_w.resize(L);
        h->dw_b.resize(L);
        h->dw_prelu.assign(L, SIZE_MAX);
        // LayerNorm affine folded into the following 1x1 conv:  W (g*n + beta) + b = (W diag g) n + (W beta + b)
        WGet last_folded;                                        // the LayerNorm-folded matrix of the last fold_ln call (split-bf16 packing)
        auto fold_ln = [&](const std::string& wname, const std::string& bname, const std::string& gname,
                           const std::string& betaname, int rows, int pairC) {
            const HostTensor* w = &W(h, wname);
            const HostTensor* bb = &W(h, bname);
            const HostTensor* g = &W(h, gname);
            const HostTensor* be = &W(h, betaname);
            WGet wf = [w, g, C](int r, int k, int) { return (double)w->data[(size_t)r * C + k] * (double)g->data[k]; };
            last_folded = wf;
            std::function<double(int)> bf = [w, bb, be, C](int r) {
                double s = bb->data[r];
                for (int k = 0; k < C; ++k) s += (double)w->data[(size_t)r * C + k] * (double)be->data[k];
                return s;
            };
            return pack_gemm(h, rows, C, 1, pairC, wf, &bf);
        };
        const bool x3 = h->precision == 1 && lx_x3_supported(C, inner);
        h->x3_conv.clear();
        h->x3_out.clear();
        if (x3) {
            h->x3_conv.resize(L);                                // (LYNXNet: x3_conv = pw1 streams, x3_out = pw2 streams)
            h->x3_out.resize(L);
        }
        for (int l = 0; l < L; ++l) {
            const std::string p = "residual_layers." + std::to_string(l) + ".convmodule.net.";
            h->g_pw1[l] = fold_ln(p + "2.weight", p + "2.bias", p + "0.weight", p + "0.bias", 2 * inner, inner);
            auto b = bias_of(p + "6.bias");
            h->g_pw2[l] = pack_gemm(h, C, inner, 1, 0, conv1(p + "6.weight"), &b);
            if (x3) {
                // pw1: row tile rt = u channels [256 rt, +256); wave w, block k: pair k >> 1 = 16 channels, k & 1 = out (0) / gate (1) rows
                h->x3_conv[l] = pack_x3(h, 2 * inner / 512, C / 32,
                                        [inner](int rt, int blk, int m) { return (blk & 1) * inner + 256 * rt + (blk >> 3) * 64 + ((blk & 7) >> 1) * 16 + m; },
                                        [](int s, int kk) { return std::make_pair(s * 32 + kk, 0); }, last_folded);
                h->x3_out[l] = pack_x3(h, C / 512, inner / 32, [](int rt, int blk, int m) { return 512 * rt + blk * 16 + m; },
                                       [](int s, int kk) { return std::make_pair(s * 32 + kk, 0); }, conv1(p + "6.weight"));
            }
            const auto& dw = W(h, p + "4.weight").data;
            h->dw_w[l] = blob_reserve(h, (size_t)inner * ks);
            memcpy(h->blob_host.data() + h->dw_w[l], dw.data(), sizeof(float) * inner * ks);
            const auto& db = W(h, p + "4.bias").data;
            h->dw_b[l] = blob_reserve(h, (size_t)inner);
            memcpy(h->blob_host.data() + h->dw_b[l], db.data(), sizeof(float) * inner);
            if (c.activation == DSD_ACT_PRELU) {
                const auto& pr = W(h, p + "5.weight").data;
                h->dw_prelu[l] = blob_reserve(h, (size_t)inner);
                memcpy(h->blob_host.data() + h->dw_prelu[l], pr.data(), sizeof(float) * inner);
            }
        }
        h->g_out = fold_ln("output_projection.weight", "output_projection.bias", "norm.weight", "norm.bias", M, 0);
    }
    return DSD_OK;
}

// ------------------------------------------------------------------------------------------
// workspace
// ------------------------------------------------------------------------------------------
constexpr size_t kGuard = 256;

int ensure_workspace(dsd_handle* h, int B, int T, hipStream_t st) {
    if (h->arena && h->B == B && h->T == T) return DSD_OK;
    // shape change: drop everything that depends on it
    for (auto& kv : h->graphs) {
        if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
        if (kv.second.graph) (void)hipGraphDestroy(kv.second.graph);
    }
    h->graphs.clear();
    h->graph_seen.clear();
    h->state_nbufs = 0;          // the state buffers are re-carved for the new shape (ensure_state)
    h->cond_ready = false;
    const int Ts = padded_ts(T);
    const size_t C = C_of(h), FM = FM_of(h), H = h->cfg.hidden_size, L = L_of(h);
    size_t off = kGuard;
    auto take = [&](size_t n) {
        size_t o = off;
        off += (n + 63) / 64 * 64 + 64;
        return o;
    };
    const size_t per = (size_t)B * Ts;
    const size_t o_cond = take(per * H), o_cp = is_aux(h) ? 0 : take(per * L * cp_rows(h)), o_xh = take(per * C);
    const size_t o_in = is_aux(h) ? 0 : take(per * FM), o_out = take(per * FM);
    size_t o_z = 0, o_skip = 0, o_h = 0, o_xin = 0, o_u = 0, o_v = 0, o_st = 0, o_lp = 0;
    if (is_aux(h)) {
        o_xin = take(per * C);
        o_u = take(per * 4 * C);
        o_st = take(per * 2);
    } else if (is_wavenet(h)) {
        o_z = take(per * C);
        o_skip = take(per * C);
        o_h = take(per * C);
    } else {
        o_xin = take(per * C);
        o_u = take(per * inner_of(h));
        o_v = take(per * inner_of(h));
        o_st = take(per * 2);
        o_lp = take(per * 2 * ((C + 63) / 64));
    }
    off += kGuard;
    // A project's segments all differ in length: keep the allocation while the new shape fits (hipFree + hipMalloc per
    // segment is a device-wide synchronisation each), re-carve it, and clear it on the caller's stream.
    float* a = h->arena;
    if (!a || off > h->arena_cap) {
        if (h->arena) (void)hipFree(h->arena);
        h->arena = a = nullptr;
        h->arena_cap = 0;
        if (hipMalloc(&a, off * sizeof(float)) != hipSuccess)
            return fail(h, DSD_ENOMEM, "hipMalloc of %zu bytes for the (B=%d, T=%d) workspace failed", off * 4, B, T);
        h->arena_cap = off;
    }
    // padding frames and guards are masked by every consumer, but start from finite values
    if (hipMemsetAsync(a, 0, off * sizeof(float), st) != hipSuccess) return fail(h, DSD_EHIP, "hipMemset(workspace) failed");
    h->arena = a;
    h->arena_floats = off;
    h->B = B; h->T = T; h->Ts = Ts;
    h->cond_i = a + o_cond; h->cp = a + o_cp; h->xh = a + o_xh; h->io_in = a + o_in; h->io_out = a + o_out;
    h->z = a + o_z; h->skip = a + o_skip; h->hbuf = a + o_h;
    h->xin = a + o_xin; h->ubuf = a + o_u; h->vbuf = a + o_v; h->stats = a + o_st; h->lnpart = a + o_lp;
    return DSD_OK;
}

int ensure_state(dsd_handle* h, int nbufs, hipStream_t st) {
    if (h->state && h->state_nbufs >= nbufs) return DSD_OK;
    const size_t per = ((size_t)h->B * FM_of(h) * h->Ts + 63) / 64 * 64 + 64;
    const size_t total = kGuard * 2 + per * nbufs;
    float* s = h->state;
    if (!s || total > h->state_cap) {
        if (h->state) (void)hipFree(h->state);
        h->state = s = nullptr;
        h->state_cap = 0;
        if (hipMalloc(&s, total * sizeof(float)) != hipSuccess)
            return fail(h, DSD_ENOMEM, "hipMalloc of %zu bytes for %d sampler state buffers failed", total * 4, nbufs);
        h->state_cap = total;
    }
    if (hipMemsetAsync(s, 0, total * sizeof(float), st) != hipSuccess) return fail(h, DSD_EHIP, "hipMemset(state) failed");
    h->state = s;
    h->state_nbufs = nbufs;
    h->state_buf_floats = per;
    // cached graphs captured the old pointers
    for (auto& kv : h->graphs) {
        if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
        if (kv.second.graph) (void)hipGraphDestroy(kv.second.graph);
    }
    h->graphs.clear();
    h->graph_seen.clear();
    return DSD_OK;
}
inline float* state_buf(dsd_handle* h, int i) { return h->state + kGuard + h->state_buf_floats * i; }

int ensure_emb(dsd_handle* h, int ncols) {
    if (h->emb_arena && h->emb_cols >= ncols) return DSD_OK;
    if (h->emb_arena) (void)hipFree(h->emb_arena);
    h->emb_arena = nullptr;
    for (auto& kv : h->graphs) {
        if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
        if (kv.second.graph) (void)hipGraphDestroy(kv.second.graph);
    }
    h->graphs.clear();
    h->graph_seen.clear();
    const int cap = round_up(ncols, 64);
    const int Ns = padded_ts(cap);
    const size_t C = C_of(h), L = L_of(h);
    size_t off = kGuard;
    auto take = [&](size_t n) {
        size_t o = off;
        off += (n + 63) / 64 * 64 + 64;
        return o;
    };
    const size_t o_t = take(Ns), o_E = take(C * Ns), o_H = take(4 * C * Ns), o_E2 = take(C * Ns), o_D = take(L * C * Ns);
    const size_t o_Dt = take(L * C * Ns);
    off += kGuard;
    float* a = nullptr;
    if (hipMalloc(&a, off * sizeof(float)) != hipSuccess) return fail(h, DSD_ENOMEM, "hipMalloc(step tables) failed");
    if (hipMemset(a, 0, off * sizeof(float)) != hipSuccess) return fail(h, DSD_EHIP, "hipMemset(step tables) failed");
    h->emb_arena = a;
    h->emb_cols = cap;
    h->Ns = Ns;
    h->t_dev = a + o_t; h->E = a + o_E; h->Hd = a + o_H; h->E2 = a + o_E2; h->D = a + o_D;
    h->Dt = a + o_Dt;
    return DSD_OK;
}

// ------------------------------------------------------------------------------------------
// GEMM launch helper
// ------------------------------------------------------------------------------------------
struct GemmCall {
    GemmP p;
    int stage, taps, epi, nb, batch, fast;
};

GemmCall make_gemm(dsd_handle* h, const PackedGemm& g, const float* Bsrc, long b_bstride, int b_rstride, int batch,
                   int T, int stage, int epi, int dil, bool generic_only = false, bool rs_pair = false) {
    GemmCall c;
    memset(&c.p, 0, sizeof(c.p));
    GemmP& p = c.p;
    p.A = h->blob + g.a_off;
    p.bias = g.bias_off == SIZE_MAX ? nullptr : h->blob + g.bias_off;
    p.M = g.M;
    p.C = g.pairC;
    p.B = Bsrc;
    p.b_bstride = b_bstride;
    p.b_rstride = b_rstride;
    p.K = g.K;
    p.Kreal = g.Kreal;
    p.KC = g.K < 256 ? g.K : 256;       // re-decided below once the LDS row stride is known
    p.T = T;
    p.dil = dil;
    p.taps = g.taps;
    p.HL = g.taps > 1 ? round_up((g.taps / 2) * dil, 4) : 0;
    p.in_scale = 1.f;
    // ragged batches: only a convolution along time can carry an item's padded frames into its valid ones, so only the
    // k-tap GEMMs over the utterances' (B, T) frames mask their input (not the 1x1s, not the step-embedding MLPs)
    const bool ragged = h->use_cg && !h->lens_host.empty() && batch == h->B && T == h->T;
    p.lens = (ragged && g.taps > 1) ? h->lens_dev : nullptr;
    c.stage = stage;
    c.taps = g.taps;
    c.epi = epi;
    c.batch = batch;
    // tile width: 64 frames when that still fills the chip twice over, else 32
    const int mtiles = g.pairC > 0 ? (g.pairC + 31) / 32 : (g.M + 63) / 64;
    const long wg64 = (long)batch * ((T + 63) / 64) * mtiles;
    const long nb2_min = path_opts().nb2_min;          // (DSD_NB2_MIN_WG: diagnostic override)
    c.nb = wg64 >= nb2_min ? 2 : 1;
    // a conv on the generic path keeps all input channels resident: 64-frame tiles only while that fits in LDS
    if ((g.taps > 3 || (generic_only && g.taps > 1)) &&
        (size_t)g.K * (64 + 2 * round_up((g.taps / 2) * dil, 4) + 16) * 4 > 150 * 1024) c.nb = 1;
    // narrow tiles (64 rows x 16 frames, c.nb == 0): when 32-frame tiles would put at most ~1.5 workgroups on a CU,
    // twice as many half-size workgroups share each SIMD between two waves and halve the latency of a lone one
    if (c.nb == 1) {
        const int force = path_opts().narrow;          // (DSD_NARROW: diagnostic override)
        const long wg32 = (long)batch * ((T + 31) / 32) * mtiles;
        int s16 = 16 + 2 * p.HL;
        while (s16 % 32 != 16) s16 += 4;
        const bool ok = (g.taps == 1 || g.taps == 3) && g.K % 64 == 0 && g.Kreal == g.K && stage != ST_LN &&
                        epi != EP_SWIGLU && epi != EP_LYNX_NEXT && gemm_has_fast(g.taps, 0, s16);
        // ... and above that while they still lower the frames the busiest CU has to cover (a launch this small is one
        // "round" of concurrent workgroups per CU): T = 1100 is 280 32-frame workgroups - 24 CUs get two, 64 frames - or
        // 552 16-frame ones, three per CU at most, 48 frames (measured at B = 1: 22.2 instead of 26.7 ms for T in (1024, 1536])
        const long wg16 = (long)batch * ((T + 15) / 16) * mtiles;
        const long load16 = (wg16 + 255) / 256 * 16, load32 = (wg32 + 255) / 256 * 32;
        // rs_pair: the caller runs 32-frame tiles as the row-split pair (wn_rowsplit.hip), ~15 ms per 50-NFE loop for any grid of
        // <= 256 workgroups; narrow tiles beat that only while they too are one round (T <= 512 at B = 1: 13.0 ms; T = 768 took
        // 16.9 ms as 384 narrow workgroups)
        const bool small = rs_pair ? wg16 <= 256 : wg32 <= 192;
        if (ok && !generic_only && (force == 1 || (force != 0 && (small || (wg32 <= 768 && load16 < load32))))) c.nb = 0;
    }
    const int BN = c.nb == 0 ? 16 : 32 * c.nb;
    p.tiles_per_b = (T + BN - 1) / BN;
    if (ragged) {       // the launch covers only the tiles that hold valid frames (lists built by prepare_ragged)
        p.cgmap = h->cg_dev[c.nb];
        p.ncg = h->cg_n[c.nb];
    }
    int S = BN + 2 * p.HL;
    while (S % 32 != 16) S += 4;
    p.S = S;
    p.mtiles = mtiles;
    const int w4 = (BN + 2 * p.HL) / 4;
    p.inv_mtiles = 1.0f / (float)mtiles;
    p.inv_tiles_per_b = 1.0f / (float)p.tiles_per_b;
    p.inv_w4 = 1.0f / (float)w4;
    p.lpr_shift = 3;
    while ((1 << p.lpr_shift) < w4) ++p.lpr_shift;
    {   // L2 blocking of the work order for GEMMs with many row tiles (GemmP::gm_shift): groups of 8 row tiles
        const int gm_env = path_opts().gm_shift;       // (DSD_GM_SHIFT: diagnostic override)
        const int shift = gm_env >= 0 ? gm_env : 3;
        const long nft = ragged ? (long)p.ncg : (long)batch * p.tiles_per_b;
        if (shift > 0 && mtiles >= (2 << shift) && mtiles % (1 << shift) == 0 && nft * mtiles >= 2048 && (nft << shift) < (1L << 22)) {
            p.gm_shift = shift;
            p.per_group = (int)(nft << shift);
            p.inv_per_group = 1.0f / (float)p.per_group;
        }
    }
    // a k=3 conv keeps all its input channels resident on the generic path (its walk does not mix taps and chunks)
    if (g.taps > 1) p.KC = g.K;
    c.fast = !generic_only && (g.taps == 1 || g.taps == 3) && (g.K % gemm_fast_chunk_rows(g.taps, c.nb) == 0) &&
             (g.Kreal == g.K) && gemm_has_fast(g.taps, c.nb, S);
    // small grids (one workgroup per CU, one wave per SIMD): all <= 4 chunks resident, no barrier in the K walk
    if (c.fast && c.nb <= 1 && g.K <= 256 && g.taps == 1 && epi != EP_LYNX_NEXT) c.fast = 2;
    p.lds_bytes = c.fast ? gemm_lds_bytes_fast(S, stage, g.taps, g.K, c.nb, c.fast == 2) : gemm_lds_bytes(p.KC, S);
    if (epi == EP_GATE || epi == EP_RESSKIP)        // the LDS-staged epilogue tile [64][BN + 4]
        p.lds_bytes = std::max(p.lds_bytes, 64 * (BN + 4) * 4);
    return c;
}

int run_gemm(dsd_handle* h, const GemmCall& c, hipStream_t st) {

    if (c.p.lds_bytes > 160 * 1024)
        return fail(h, DSD_EINVAL, "GEMM tile needs %d bytes of LDS (> 160 KiB): %d input channels x k=%d at dilation %d "
                    "is outside the supported shapes", c.p.lds_bytes, c.p.K, c.taps, c.p.dil);
    if ((long)c.batch * c.p.tiles_per_b * c.p.mtiles >= (1L << 22))
        return fail(h, DSD_EINVAL, "GEMM grid of %ld workgroups exceeds 2^22 (batch %d x %d frame tiles x %d row tiles): "
                    "split the batch", (long)c.batch * c.p.tiles_per_b * c.p.mtiles, c.batch, c.p.tiles_per_b, c.p.mtiles);
    hipError_t e = launch_gemm(c.p, c.stage, c.taps, c.epi, c.nb, c.fast, c.batch, st);
    if (e != hipSuccess) return fail(h, DSD_EHIP, "GEMM launch failed: %s", hipGetErrorString(e));
    return DSD_OK;
}

// WaveNet: how one residual layer runs on this (B, T) / these lengths.  Three launch shapes exist:
//   * wn_layer.hip, ONE launch per layer, a workgroup per 32-frame tile with all 2C rows: a launch is ceil(tiles / 256) rounds
//     of one tile per CU and a round takes the same time whether 1 or 256 of its tiles exist (at C = 256 ~67 us, 63 us of it
//     the tile's own 152 k cycles) - the most efficient form at whole rounds (0.81 of the fp32-MFMA peak), the worst just
//     above one (B = 9 x 1000 frames = 288 tiles: two rounds for 1.125 rounds of work);
//   * wn_rowsplit.hip, TWO launches per layer with the rows split over 8 workgroups per tile: fills the chip from 32 tiles,
//     balanced at any tile count, but two boundaries / cold prologues per layer and every x tile staged eight times
//     (one utterance: ~14 us; slope ~0.30 us per tile);
//   * the two GEMMs of gemm.hip (any shape; 64-frame tiles from 512 workgroups up: 13 us + 0.275 us per tile).
// A PLAN is a list of segments, each a launch shape over a contiguous range of the batch's (item, frame tile) order; every
// segment reads the layer's input buffer and writes the other one (x ping-pongs between xh and z, so tiles are independent
// within a layer and a neighbour's halo is always the layer's INPUT), the row-split segments put their gated tile into hbuf.
// Mixed plans: the whole rounds on the fused kernel, the remainder on the row-split pair (B = 9: 67 + 14 us instead of
// 2 x 67 fused or ~92 split).  Costs below are in units of one fused round; they are RATIOS measured on one box
// (profiles/r03_plan_sweep.txt), not absolute times.  DSD_FUSED_LAYER=0/1 forces no / only fused segments, DSD_WN_PLAN=0
// keeps one launch shape per layer (round 2's rule).  An empty plan = the per-layer choice between the row-split pair
// and the GEMM pair in run_backbone.
enum { WN_FUSED = 0, WN_ROWSPLIT = 1, WN_FUSED_X3 = 2 };
struct WnSeg {
    int kind, bn;       // launch shape, frames per tile
    int t0, nt;         // tiles [t0, t0 + nt) of the (item, frame tile) order at this width (ragged: of the valid-tile list)
    int rows;           // WN_ROWSPLIT: rows per workgroup - 64 (wn_rowsplit.hip), 128 or 256 (wn_rows.hip)
};

inline long wn_tiles32(const dsd_handle* h) {
    if (h->lens_host.empty()) return (long)h->B * ((h->T + 31) / 32);
    long tiles = 0;
    for (int v : h->lens_host) tiles += (v + 31) / 32;
    return tiles;
}
inline long wn_tiles16(const dsd_handle* h) {
    if (h->lens_host.empty()) return (long)h->B * ((h->T + 15) / 16);
    long tiles = 0;
    for (int v : h->lens_host) tiles += (v + 15) / 16;
    return tiles;
}

// Cost of the two-launch path over `tiles` 32-frame tiles by rows per workgroup, in fused rounds (one round = 256 tiles on
// wn_layer.hip, ~67 us at C = 256); RATIOS measured on one box at T = 1000 (profiles/r03_rows_sweep.txt):
//   64 rows  (wn_rowsplit.hip up to ~110 tiles, the 64-frame-tile GEMM pair of gemm.hip above): 23.6 us at 64 tiles, 32.9 at 96,
//             46.8 at 125, 63.4 at 157, 64.7 at 188
//   128 rows (wn_rows.hip, 4 workgroups per tile, two resident per CU): 22.9 us per started round of 64 tiles + ~18.3 per further
//   256 rows (2 workgroups per tile, one per CU): 39.6 us per started round of 128 tiles (5.7 + 33.9 r)
//             as the remainder segment of a mixed plan (always wn_rowsplit.hip): 15.2 us at 32 tiles, 23.9 at 64, 33.7 at 96,
//             42.4 at 128, 51.8 at 160 (profiles/r03_plan_sweep.txt)
inline double wn_split_cost(long tiles, int rows, bool segment) {
    if (rows == 128) return 0.069 + 0.273 * (double)((tiles + 63) / 64);
    if (rows == 256) return 0.085 + 0.506 * (double)((tiles + 127) / 128);
    if (segment) return 0.085 + (double)tiles / 232.0;
    // (above ~110 tiles: the GEMM pair's 64-frame tiles fill the chip two utterances of ~1000 frames at a time)
    return tiles <= 110 ? 0.067 + (double)tiles / 223.0 : 0.04 + 0.30 * (double)((tiles + 63) / 64);
}
// ... and the rows per workgroup that minimise it.  DSD_RS_ROWS forces.
inline int wn_rows_for(long tiles, bool segment) {
    const int force = path_opts().rs_rows;
    if (force == 64 || force == 128 || force == 256) return force;
    int best = 64;
    for (int rows : {128, 256})
        if (wn_split_cost(tiles, rows, segment) < wn_split_cost(tiles, best, segment)) best = rows;
    return best;
}

bool wn_plan_for(const dsd_handle* h, std::vector<WnSeg>& segs) {
    segs.clear();
    if (!is_wavenet(h)) return false;
    const int C = C_of(h);
    const int max_dil = 1 << (std::min(h->cfg.dilation_cycle_length, L_of(h)) - 1);
    if (h->cfg.dilation_cycle_length < 1 || !wn_layer_supported(C, max_dil)) return false;
    const PathOpts& o = path_opts();
    const long tiles = wn_tiles32(h);
    if (tiles == 0 || tiles >= (1L << 22)) return false;
    const bool rs_ok = o.rowsplit != 0 && wn_rowsplit_supported(C, max_dil, h->Ts) && wn_rows_supported(C, max_dil, h->Ts);
    // split-bf16 mode: the fused segments run wn_layer_x3.hip, whose round takes ~0.55 of an fp32 round (bound by the weight
    // stream, not the MFMAs); the two-launch kernels have no bf16x3 form, so the remainder of a mixed plan stays fp32
    const bool x3 = h->precision == 1 && !h->x3_conv.empty();
    const int fused_kind = x3 ? WN_FUSED_X3 : WN_FUSED;
    const double round_cost = x3 ? 0.55 : 1.0;
    if (o.fused_layer == 1) {                      // forced: every tile through the fused kernel, whatever the grid
        segs.push_back({fused_kind, 32, 0, (int)tiles, 0});
        return true;
    }
    // the fused kernel on 16-frame tiles (wn_layer16_kernel): a round of one tile per CU takes kRound16 of a 32-frame round (half the
    // MFMAs against the same 2 MB of weights per workgroup: profiles/r03_plan_sweep.txt); fp32 only, whole layers only
    const long tiles16 = wn_tiles16(h);
    const bool f16_ok = !x3 && wn_layer16_supported(C, max_dil) && tiles16 < (1L << 22);
    if (o.fused16 == 1 && f16_ok) {                // forced
        segs.push_back({WN_FUSED, 16, 0, (int)tiles16, 0});
        return true;
    }
    if (o.wn_plan == 2 && rs_ok && o.fused_layer != 0) {
        // test hook: a mixed plan at any size - the first half of the tiles fused, the rest on the two-launch path
        const int nfh = (int)(tiles / 2);
        if (nfh > 0) segs.push_back({fused_kind, 32, 0, nfh, 0});
        segs.push_back({WN_ROWSPLIT, 32, nfh, (int)tiles - nfh, wn_rows_for(tiles - nfh, true)});
        return true;
    }
    if ((o.wn_plan == 3 || o.wn_plan == 4) && f16_ok && rs_ok && o.fused_layer != 0) {
        // test hooks: plans with a 16-frame fused segment at any size - 3: the first half of the 32-frame tiles on 16-frame fused
        // tiles, the rest on the two-launch path; 4: the first half on the 32-frame fused kernel, the rest on 16-frame tiles
        const long nfh = tiles / 2;
        auto m16 = [h](long n) -> long {
            if (h->lens_host.empty()) {
                const long tpb32 = (h->T + 31) / 32, tpb16 = (h->T + 15) / 16;
                return (n / tpb32) * tpb16 + 2 * (n % tpb32);
            }
            long n32 = 0, n16 = 0;
            for (int v : h->lens_host) {
                const long t32 = (v + 31) / 32, t16 = (v + 15) / 16;
                if (n < n32 + t32) return n16 + 2 * (n - n32);
                n32 += t32; n16 += t16;
            }
            return n16;
        };
        if (o.wn_plan == 3) {
            if (nfh > 0) segs.push_back({WN_FUSED, 16, 0, (int)m16(nfh), 0});
            segs.push_back({WN_ROWSPLIT, 32, (int)nfh, (int)(tiles - nfh), wn_rows_for(tiles - nfh, true)});
        } else {
            if (nfh > 0) segs.push_back({WN_FUSED, 32, 0, (int)nfh, 0});
            segs.push_back({WN_FUSED, 16, (int)m16(nfh), (int)(tiles16 - m16(nfh)), 0});
        }
        return true;
    }
    const bool rows_forced = rs_ok && (o.rs_rows == 128 || o.rs_rows == 256);
    if (rows_forced && o.fused_layer == 0) {       // forced: the whole layer on wide row tiles
        segs.push_back({WN_ROWSPLIT, 32, 0, (int)tiles, o.rs_rows});
        return true;
    }
    if (o.fused_layer == 0) return false;
    const bool plans = rs_ok && o.wn_plan != 0;     // DSD_WN_PLAN=0: one launch shape per layer, no wide row tiles (round 2's rule)
    // one fused round = one tile per CU; the two-launch costs were measured on 256 CUs, so their tile counts are taken in
    // 256ths of the chip (`eq`): a part with fewer CUs sees proportionally "more" tiles
    const long cus = h->cus;
    const long rounds = (tiles + cus - 1) / cus, nf = tiles / cus * cus, rem = tiles - nf;
    auto eq = [cus](long t) { return (t * 256 + cus - 1) / cus; };
    const double fused_all = 2 * tiles >= cus ? round_cost * (double)rounds : 1e30;
    const int rows_all = plans ? wn_rows_for(eq(tiles), false) : 64;
    const double split_all = wn_split_cost(eq(tiles), rows_all, false);
    double mixed = 1e30;
    if (plans && nf > 0 && rem > 0) mixed = round_cost * (double)(nf / cus) + wn_split_cost(eq(rem), wn_rows_for(eq(rem), true), true);
    // ... and the 16-frame fused kernel as a whole layer, as the FIRST segment of a layer of up to ~1.5 rounds (one full round of
    // 16-frame tiles, the rest on the two-launch path: B = 5 at T = 1000), or as the remainder behind whole 32-frame rounds
    constexpr double kRound16 = 0.53;               // a round of 16-frame tiles / a round of 32-frame tiles (35.3 / 67.1 ms per loop)
    const bool use16 = f16_ok && o.fused16 != 0;
    // index in the 16-frame tile order of the first 16-frame tile of 32-frame tile n (of the dense order / the valid-tile list)
    auto map16 = [h](long n) -> long {
        if (h->lens_host.empty()) {
            const long tpb32 = (h->T + 31) / 32, tpb16 = (h->T + 15) / 16;
            return (n / tpb32) * tpb16 + 2 * (n % tpb32);
        }
        long n32 = 0, n16 = 0;
        for (int v : h->lens_host) {
            const long t32 = (v + 31) / 32, t16 = (v + 15) / 16;
            if (n < n32 + t32) return n16 + 2 * (n - n32);
            n32 += t32; n16 += t16;
        }
        return n16;
    };
    const double fused16_all = (use16 && 2 * tiles16 >= cus) ? kRound16 * (double)((tiles16 + cus - 1) / cus) : 1e30;
    double head16 = 1e30, tail16 = 1e30;
    long head_n32 = 0;
    if (use16 && plans && tiles16 > cus) {          // one round of 16-frame tiles in front, the rest two launches
        long lo = 0, hi = tiles;                    // the largest n with map16(n) <= cus
        while (lo < hi) {
            const long mid = (lo + hi + 1) / 2;
            if (map16(mid) <= cus) lo = mid; else hi = mid - 1;
        }
        head_n32 = lo;
        if (head_n32 > 0 && head_n32 < tiles)
            head16 = kRound16 + wn_split_cost(eq(tiles - head_n32), wn_rows_for(eq(tiles - head_n32), true), true);
    }
    if (use16 && nf > 0 && rem > 0) {               // whole 32-frame rounds, the remainder as one round of 16-frame tiles
        const long rem16 = tiles16 - map16(nf);
        if (rem16 <= cus && 2 * rem16 >= cus) tail16 = round_cost * (double)(nf / cus) + kRound16;
    }
    const double best_other = std::min(std::min(mixed, fused_all), split_all);
    if (fused16_all < best_other && fused16_all <= head16 && fused16_all <= tail16) {
        segs.push_back({WN_FUSED, 16, 0, (int)tiles16, 0});
        return true;
    }
    if (head16 < best_other && head16 <= tail16) {
        segs.push_back({WN_FUSED, 16, 0, (int)map16(head_n32), 0});
        segs.push_back({WN_ROWSPLIT, 32, (int)head_n32, (int)(tiles - head_n32), wn_rows_for(eq(tiles - head_n32), true)});
        return true;
    }
    if (tail16 < best_other) {
        segs.push_back({fused_kind, 32, 0, (int)nf, 0});
        segs.push_back({WN_FUSED, 16, (int)map16(nf), (int)(tiles16 - map16(nf)), 0});
        return true;
    }
    if (mixed < fused_all && mixed < split_all) {
        segs.push_back({fused_kind, 32, 0, (int)nf, 0});
        segs.push_back({WN_ROWSPLIT, 32, (int)nf, (int)rem, wn_rows_for(eq(rem), true)});
        return true;
    }
    if (fused_all < split_all) {
        segs.push_back({fused_kind, 32, 0, (int)tiles, 0});
        return true;
    }
    if (rows_all > 64) {
        segs.push_back({WN_ROWSPLIT, 32, 0, (int)tiles, rows_all});
        return true;
    }
    return false;                                   // 64 rows per workgroup: the per-layer choice in run_backbone
}

// DSD_EDGE: 0 = never the edge kernel (wn_edge.hip), 1 = on every grid, unset = by grid size.  Read per call: tests/
// test_gpu_edge.py switches it between two handles of one process.
inline int edge_choice() { return path_opts().edge; }

// Layer `layer`'s FiLM vector d[c] for step column col0 (+ colb per batch item): kernels read film[c * cstride + c0 + b * cb].
// From the transposed table Dt [step][L * C] that is C contiguous floats (DSD_FILM_T=0: from D [L * C][Ns], one line per row - A/B)
inline void film_of(const dsd_handle* h, int layer, int col0, int colb, const float*& film, int& cstride, int& c0, int& cb) {
    const int transposed = path_opts().film_t != 0;
    const int C = C_of(h), LC = L_of(h) * C;
    if (transposed) {
        film = h->Dt + (long)layer * C; cstride = 1; c0 = col0 * LC; cb = colb * LC;
    } else {
        film = h->D + (long)layer * C * h->Ns; cstride = h->Ns; c0 = col0; cb = colb;
    }
}

// step tables: E = sinemb(t) -> Hd = act(W0 E + b0) -> E2 = W1 Hd + b1 -> D[l*C + c][col] = Wd_l E2 + bd_l
int run_step_tables(dsd_handle* h, int ncols, hipStream_t st) {
    const int C = C_of(h), Ns = h->Ns;
    hipError_t e = launch_sinemb(h->t_dev, ncols, Ns, h->blob + h->freqs_off, C, h->E, st);
    if (e != hipSuccess) return fail(h, DSD_EHIP, "sinemb launch failed: %s", hipGetErrorString(e));
    GemmCall g0 = make_gemm(h, h->g_emb0, h->E, 0, Ns, 1, ncols, ST_PLAIN, EP_BIAS_ACT, 0);
    g0.p.act = h->emb_act; g0.p.out = h->Hd; g0.p.o_bstride = 0; g0.p.o_rstride = Ns;
    int rc = run_gemm(h, g0, st);
    if (rc) return rc;
    GemmCall g1 = make_gemm(h, h->g_emb1, h->Hd, 0, Ns, 1, ncols, ST_PLAIN, EP_BIAS_ACT, 0);
    g1.p.act = ACT_NONE; g1.p.out = h->E2; g1.p.o_rstride = Ns;
    rc = run_gemm(h, g1, st);
    if (rc) return rc;
    GemmCall g2 = make_gemm(h, h->g_dproj, h->E2, 0, Ns, 1, ncols, ST_PLAIN, EP_BIAS_ACT, 0);
    g2.p.act = ACT_NONE; g2.p.out = h->D; g2.p.o_rstride = Ns;
    if ((rc = run_gemm(h, g2, st))) return rc;
    // ... and transposed: a layer's FiLM vector for one step is then C contiguous floats.  Read from D it is a gather of one
    // cache line per channel (256 lines per workgroup of the one-utterance conv kernel: halving the waves that issue it
    // alone was worth 2.2 % of the 50-NFE loop)
    const int LC = (int)L_of(h) * C;
    hipError_t te = launch_transpose(h->D, LC, ncols, Ns, h->Dt, LC, st);
    if (te != hipSuccess) return fail(h, DSD_EHIP, "step-table transpose launch failed: %s", hipGetErrorString(te));
    return DSD_OK;
}

// One backbone evaluation on the internal-layout input `xin_state` ([B][F*M][Ts]); the last GEMM's
// epilogue writes the `nout` linear combinations `lo` (LinTerm.ptr == nullptr = model output).
int run_backbone(dsd_handle* h, const float* xin_state, int film_col0, int film_colb, const LinOut* lo, int nout,
                 hipStream_t st, const float* next_xin = nullptr) {
    const int B = h->B, T = h->T, Ts = h->Ts, C = C_of(h), FM = FM_of(h), L = L_of(h), Ns = h->Ns;
    const long xs = (long)C * Ts;
    int rc;
    RaggedScope ragged_scope(h);
    // timing pass: per kernel class (launch site + variant) every timing_stride-th launch carries events (a dispatch with
    // profiling events costs the command processor more than a plain one; sampling keeps the pass close to the untimed pace)
    dsd_handle::TimedClass* tcls = nullptr;
    auto timed_begin = [&](int key, double flops, double bytes) {
        tcls = nullptr;
        if (!h->timing) return;
        for (auto& c : h->tclasses)
            if (c.key == key) tcls = &c;
        if (!tcls) {
            h->tclasses.emplace_back();
            tcls = &h->tclasses.back();
            tcls->key = key;
        }
        tcls->flops = flops;
        tcls->bytes = bytes;
        if (tcls->launches++ % h->timing_stride != 0) { tcls = nullptr; return; }
        if (h->ev_used == h->ev_pool.size()) {
            hipEvent_t a, b;
            (void)hipEventCreate(&a);
            (void)hipEventCreate(&b);
            h->ev_pool.emplace_back(a, b);
        }
        TimingSlot& ts = timing_slot();
        ts.e0 = h->ev_pool[h->ev_used].first;
        ts.e1 = h->ev_pool[h->ev_used].second;
        ts.taken = false;
    };
    auto timed_end = [&]() {
        if (!tcls) return;
        TimingSlot& ts = timing_slot();
        if (ts.taken) {
            if (tcls->name.empty()) tcls->name = ts.name;
            tcls->evs.push_back(h->ev_used++);
        }
        ts.e0 = ts.e1 = nullptr;
        ts.taken = false;
        tcls = nullptr;
    };
    if (h->timing) ++h->timing_evals;
    if (h->timing) {      // one empty bracket per evaluation calibrates what a hipEvent pair itself costs
        if (h->cal_used == h->cal_pool.size()) {
            hipEvent_t a, b;
            (void)hipEventCreate(&a);
            (void)hipEventCreate(&b);
            h->cal_pool.emplace_back(a, b);
        }
        (void)hipEventRecord(h->cal_pool[h->cal_used].first, st);
        (void)hipEventRecord(h->cal_pool[h->cal_used].second, st);
        ++h->cal_used;
    }
    const bool lynx = !is_wavenet(h);
    const int ln_tiles = (C + 63) / 64;
    // LYNXNet layer transition, fused into the producing GEMM's epilogue (lynxnet.py:76-84): `next` = the layer whose
    // conditioner / step projections are added (L = none: only the final LayerNorm follows)
    auto lynx_next = [&](GemmCall& g, int next) {
        g.p.out = h->xh; g.p.o_bstride = xs; g.p.o_rstride = Ts;
        g.p.out2 = next < L ? h->xin : nullptr;
        g.p.lnpart = h->lnpart; g.p.lnpart_ts = Ts;
        g.p.strong = h->cfg.strong_cond;
        if (next < L) {
            g.p.cpn = h->cp + (long)next * C * Ts; g.p.cpn_bstride = (long)L * C * Ts; g.p.cpn_rstride = Ts;
            film_of(h, next, film_col0, film_colb, g.p.film, g.p.film_cstride, g.p.film_col0, g.p.film_colb);
        }
    };
    // WaveNet: the previous evaluation's edge kernel (wn_edge.hip) may already have projected this very input into xh
    const bool inproj_done = !lynx && h->edge_xh_src != nullptr && h->edge_xh_src == xin_state;
    h->edge_xh_src = nullptr;
    if (!inproj_done) {   // input projection (+ReLU for WaveNet, wavenet.py:86-88; GELU unless strong_cond for LYNXNet, lynxnet.py:141-143)
        GemmCall g = make_gemm(h, h->g_inproj, xin_state, (long)FM * Ts, Ts, B, T, ST_PLAIN,
                               lynx ? EP_LYNX_NEXT : EP_BIAS_ACT, 0);
        g.p.act = is_wavenet(h) ? ACT_RELU : (h->cfg.strong_cond ? ACT_NONE : ACT_GELU);
        g.p.out = h->xh; g.p.o_bstride = xs; g.p.o_rstride = Ts;
        if (lynx) lynx_next(g, 0);
        if ((rc = run_gemm(h, g, st))) return rc;
    }
    if (is_wavenet(h)) {
        const long cps = (long)L * 2 * C * Ts;
        const bool ragged = h->use_cg && !h->lens_host.empty();
        // algorithmic work of the layer kernels per VALID frame: conv 3*C*2C MACs, out-proj C*2C;
        // bytes: the conv launch reads x and the hoisted conditioner projection and writes z (16 C), the out-proj launch
        // reads z, updates x and the skip sum (20 C); fused: x r/w, conditioner projection, skip r/w (24 C)
        const double fl_conv = 2.0 * 3 * C * 2 * C, fl_out = 2.0 * C * 2 * C;
        auto seg_frames = [&](int bn, int t0, int nt) -> double {       // valid frames in tiles [t0, t0 + nt) at width bn
            const int tpb = (T + bn - 1) / bn;
            double fr = 0;
            if (!ragged) {
                for (int i = t0; i < t0 + nt; ++i) fr += std::min(bn, T - (i % tpb) * bn);
            } else {
                const int k = bn == 16 ? 0 : bn / 32;
                for (int i = t0; i < t0 + nt; ++i) {
                    const int e = h->cg_host[k][i], b = e / tpb, ft = e - b * tpb;
                    fr += std::min(bn, h->lens_host[b] - ft * bn);
                }
            }
            return fr;
        };
        auto layer_params = [&](WnLayerP& p, int l, int bn) {
            memset(&p, 0, sizeof(p));
            p.Aconv = h->blob + h->g_conv[l].a_off;
            p.Aout = h->blob + h->g_outp[l].a_off;
            p.bias_out = h->blob + h->g_outp[l].bias_off;
            p.skip = h->skip;
            p.x_bstride = xs; p.Ts = Ts;
            p.cp = h->cp + (long)l * 2 * C * Ts; p.cp_bstride = cps;
            film_of(h, l, film_col0, film_colb, p.film, p.film_cstride, p.film_col0, p.film_colb);
            p.dil = 1 << (l % h->cfg.dilation_cycle_length);
            p.T = T; p.tiles_per_b = (T + bn - 1) / bn; p.inv_tiles_per_b = 1.0f / (float)p.tiles_per_b;
            p.first_layer = (l == 0);
        };
        std::vector<WnSeg> plan;
        if (wn_plan_for(h, plan)) {
            // the residual stream ping-pongs between xh and z (a tile's halo columns must come from the layer's INPUT, which
            // a neighbouring tile - of this or another segment - may already have replaced); row-split segments gate into hbuf
            const float* xi = h->xh;
            float* xo = h->z;
            std::vector<double> seg_fr(plan.size());
            for (size_t k = 0; k < plan.size(); ++k) seg_fr[k] = seg_frames(plan[k].bn, plan[k].t0, plan[k].nt);
            for (int l = 0; l < L; ++l) {
                for (size_t k = 0; k < plan.size(); ++k) {
                    const WnSeg& sg = plan[k];
                    WnLayerP p;
                    layer_params(p, l, sg.bn);
                    p.xin = xi; p.xout = xo; p.z = h->hbuf;
                    p.tile0 = sg.t0; p.ntiles = sg.nt;
                    if (ragged) { p.lens = h->lens_dev; p.cgmap = h->cg_dev[sg.bn == 16 ? 0 : 1] + sg.t0; p.ncg = sg.nt; }
                    const int vkey = (int)k * 4 + (p.dil > 8 ? 2 : 0) + (sg.bn == 16 ? 1 : 0);
                    hipError_t le;
                    if (sg.kind == WN_FUSED_X3) {
                        p.Aconv = h->blob + h->x3_conv[l];
                        p.Aout = h->blob + h->x3_out[l];
                        timed_begin(150 + vkey, (fl_conv + fl_out) * seg_fr[k], 24.0 * C * seg_fr[k]);
                        le = launch_wn_layer_x3(p, C, B, st);
                        timed_end();
                        if (le != hipSuccess) return fail(h, DSD_EHIP, "bf16x3 fused WaveNet layer launch failed: %s", hipGetErrorString(le));
                    } else if (sg.kind == WN_FUSED) {
                        timed_begin(100 + vkey, (fl_conv + fl_out) * seg_fr[k], 24.0 * C * seg_fr[k]);
                        le = launch_wn_layer(p, C, B, st, sg.bn);
                        timed_end();
                        if (le != hipSuccess) return fail(h, DSD_EHIP, "fused WaveNet layer launch failed: %s", hipGetErrorString(le));
                    } else {
                        const bool wide = sg.rows > 64;
                        timed_begin(200 + vkey, fl_conv * seg_fr[k], 16.0 * C * seg_fr[k]);
                        le = wide ? launch_wn_rows(p, 0, C, B, sg.rows, st) : launch_wn_rowsplit(p, 0, C, B, sg.bn, st);
                        timed_end();
                        if (le == hipSuccess) {
                            timed_begin(300 + vkey, fl_out * seg_fr[k], 20.0 * C * seg_fr[k]);
                            le = wide ? launch_wn_rows(p, 1, C, B, path_opts().rs_rows_out > 64 ? path_opts().rs_rows_out : sg.rows, st)
                                      : launch_wn_rowsplit(p, 1, C, B, sg.bn, st);
                            timed_end();
                        }
                        if (le != hipSuccess) return fail(h, DSD_EHIP, "row-split WaveNet layer launch failed: %s", hipGetErrorString(le));
                    }
                }
                xi = xo;
                xo = (xo == h->z) ? h->xh : h->z;
            }
        } else
        for (int l = 0; l < L; ++l) {
            const int dil = 1 << (l % h->cfg.dilation_cycle_length);
            // 32-frame tiles on a grid of about one workgroup per CU (one utterance of ~1000 frames): the row-split pair
            // of wn_rowsplit.hip - every weight block loaded once, compiler-counted waits - instead of the two GEMMs
            const bool rs_ok = path_opts().rowsplit != 0 && wn_rowsplit_supported(C, dil, Ts);
            GemmCall g = make_gemm(h, h->g_conv[l], h->xh, xs, Ts, B, T, ST_FILM, EP_GATE, dil, false, rs_ok);
            // 48-frame tiles where they make a dense launch ONE round of workgroups and 32-frame tiles do not (T in (1024, 1536]
            // at B = 1: 35-48 tiles of 32 frames = 280-384 workgroups for 256 CUs): 21.9 -> see DESIGN 4.2.  DSD_RS_BN48=0: off
            const bool bn48 = rs_ok && path_opts().rs_bn48 != 0 && !ragged && (long)B * ((T + 31) / 32) * 8 > 256 &&
                              (long)B * ((T + 47) / 48) * 8 <= 256;
            const double fr_all = ragged ? seg_frames(32, 0, h->cg_n[1]) : (double)B * T;
            if (rs_ok && ((g.nb == 1 && g.fast) || bn48)) {
                const int bn = bn48 ? 48 : 32;
                WnLayerP p;
                layer_params(p, l, bn);
                p.xin = h->xh; p.xout = h->xh; p.z = h->z;
                if (ragged) { p.lens = h->lens_dev; p.cgmap = h->cg_dev[1]; p.ncg = h->cg_n[1]; }
                timed_begin(200 + (bn48 ? 1 : 0) + (dil > 8 ? 2 : 0), fl_conv * fr_all, 16.0 * C * fr_all);
                hipError_t le = launch_wn_rowsplit(p, 0, C, B, bn, st);
                timed_end();
                if (le == hipSuccess) {
                    timed_begin(300 + (bn48 ? 1 : 0) + (dil > 8 ? 2 : 0), fl_out * fr_all, 20.0 * C * fr_all);
                    le = launch_wn_rowsplit(p, 1, C, B, bn, st);
                    timed_end();
                }
                if (le != hipSuccess) return fail(h, DSD_EHIP, "row-split WaveNet layer launch failed: %s", hipGetErrorString(le));
                continue;
            }
            film_of(h, l, film_col0, film_colb, g.p.film, g.p.film_cstride, g.p.film_col0, g.p.film_colb);
            g.p.aux = h->cp + (long)l * 2 * C * Ts; g.p.aux_bstride = cps; g.p.aux_rstride = Ts;
            g.p.out = h->z; g.p.o_bstride = xs; g.p.o_rstride = Ts;
            timed_begin(400 + (dil > 8 ? 2 : 0), fl_conv * fr_all, 16.0 * C * fr_all);
            rc = run_gemm(h, g, st);
            timed_end();
            if (rc) return rc;
            GemmCall o = make_gemm(h, h->g_outp[l], h->z, xs, Ts, B, T, ST_PLAIN, EP_RESSKIP, 0);
            o.p.C = C; o.p.x = h->xh; o.p.skip = h->skip; o.p.first_layer = (l == 0);
            o.p.o_bstride = xs; o.p.o_rstride = Ts;
            timed_begin(500, fl_out * fr_all, 20.0 * C * fr_all);
            rc = run_gemm(h, o, st);
            timed_end();
            if (rc) return rc;
        }
        // skip projection -> output projection + solver update (-> the next evaluation's input projection) in one launch with
        // one workgroup per frame tile (wn_edge.hip); DSD_EDGE=0: the three GEMMs of gemm.hip
        const int edge_env = edge_choice();
        const bool e_ragged = h->use_cg && !h->lens_host.empty();
        const long e_t32 = e_ragged ? (long)h->cg_n[1] : (long)B * ((T + 31) / 32);       // 32-frame tiles = workgroups
        if (edge_env != 0 && wn_edge_supported(C, FM) && nout >= 1 && nout <= kMaxOut && (edge_env == 1 || e_t32 >= 128)) {
            const bool ragged = e_ragged;
            const int ncb = 2;
            const int bnw = 16 * ncb;
            WnEdgeP p;
            memset(&p, 0, sizeof(p));
            p.A1 = h->blob + h->g_tail1.a_off; p.b1 = h->blob + h->g_tail1.bias_off;
            p.A2 = h->blob + h->g_out.a_off; p.b2 = h->blob + h->g_out.bias_off;
            p.A3 = h->blob + h->g_inproj.a_off; p.b3 = h->blob + h->g_inproj.bias_off;
            p.skip = h->skip; p.xh = h->xh; p.x_bstride = xs; p.Ts = Ts; p.T = T; p.FM = FM;
            p.in_scale = sqrtf((float)L);
            p.tiles_per_b = (T + bnw - 1) / bnw; p.inv_tiles_per_b = 1.0f / (float)p.tiles_per_b;
            p.nout = nout;
            bool fits = true;
            for (int i = 0; i < nout; ++i) {
                p.dst[i] = lo[i].dst;
                for (int k = 0; k < lo[i].nterms; ++k) {
                    const LinTerm& tm = lo[i].t[k];
                    if (tm.ptr == nullptr) { p.cm[i] += tm.coef; continue; }
                    if (p.nq == kEdgeMaxTerms || tm.ext) { fits = false; break; }      // (caller-noise terms: the GEMM path)
                    EdgeTerm& q = p.q[p.nq++];
                    q.ptr = tm.ptr; q.bstride = tm.bstride; q.rstride = tm.rstride; q.ext = tm.ext; q.coef = tm.coef; q.out = i;
                }
            }
            p.o_bstride = (long)FM * Ts; p.o_rstride = Ts;
            p.next_src = -1;
            if (next_xin)
                for (int i = 0; i < nout; ++i)
                    if (lo[i].dst == next_xin) p.next_src = i;
            int nwg = B * p.tiles_per_b;
            if (ragged) { p.cgmap = h->cg_dev[ncb == 2 ? 1 : 0]; p.ncg = h->cg_n[ncb == 2 ? 1 : 0]; nwg = p.ncg; }
            if (fits) {      // (more state terms than the kernel holds at once: the three GEMMs below)
                double fr_all = (double)B * T;
                if (ragged) { fr_all = 0; for (int v : h->lens_host) fr_all += v; }
                timed_begin(700, 2.0 * (C * C + 2.0 * C * FM) * fr_all, 4.0 * (2 * C + 3 * FM) * fr_all);
                hipError_t ee = launch_wn_edge(p, C, ncb, nwg, st);
                timed_end();
                if (ee != hipSuccess) return fail(h, DSD_EHIP, "WaveNet edge-kernel launch failed: %s", hipGetErrorString(ee));
                if (p.next_src >= 0) h->edge_xh_src = next_xin;
                return DSD_OK;
            }
        }
        GemmCall t1 = make_gemm(h, h->g_tail1, h->skip, xs, Ts, B, T, ST_SCALE, EP_BIAS_ACT, 0);
        t1.p.in_scale = sqrtf((float)L);      // staged value is DIVIDED by in_scale (wavenet.py:96)
        t1.p.act = ACT_RELU; t1.p.out = h->hbuf; t1.p.o_bstride = xs; t1.p.o_rstride = Ts;
        if ((rc = run_gemm(h, t1, st))) return rc;
        GemmCall t2 = make_gemm(h, h->g_out, h->hbuf, xs, Ts, B, T, ST_PLAIN, EP_LINCOMB, 0);
        t2.p.nout = nout;
        for (int i = 0; i < nout; ++i) t2.p.lo[i] = lo[i];
        t2.p.o_bstride = (long)FM * Ts; t2.p.o_rstride = Ts;
        return run_gemm(h, t2, st);
    }
    // ---- LYNXNet (lynxnet.py:76-87, 145-154) ----
    const int inner = inner_of(h);
    const long us = (long)inner * Ts;
    hipError_t e;
    // LayerNorm statistics of the next GEMM's input: merged from the producer's per-tile partials by a small kernel
    // (merging inside the consuming GEMM's prologue was measured slower: every one of its ~1 k workgroups repeats it)
    auto ln_input = [&](GemmCall& g) -> int {
        hipError_t me = launch_ln_merge(h->lnpart, ln_tiles, C, B, T, Ts, 1e-5f, h->stats, st);
        if (me != hipSuccess) return fail(h, DSD_EHIP, "LayerNorm merge launch failed: %s", hipGetErrorString(me));
        g.p.ln_stats = h->stats; g.p.ln_ts = Ts;
        return DSD_OK;
    };
    // Batched grids: the two pointwise GEMMs with the whole K extent of a 32-frame tile resident in LDS (lynx_layer.hip);
    // pw2 launches only C / 512 workgroups per frame tile, so a single utterance stays on the GEMM family
    const bool lx_ragged = h->use_cg && !h->lens_host.empty();
    const long lx_tiles = lx_ragged ? (long)h->cg_n[1] : (long)B * ((T + 31) / 32);
    const int lx_env = path_opts().lynx_resident;
    const bool lx_ok = lx_env != 0 && lx_layer_supported(C, inner);
    // pw1 launches 2 inner / 512 workgroups per frame tile (8 at C = 1024: one utterance of ~1000 frames already fills the
    // chip), pw2 only C / 512 (measured at C = 1024: slower at B = 2, +4 % at 3, +12 % at 4, +10 % at 8)
    const bool lx_res1 = lx_ok && (lx_env == 1 || lx_tiles * (2 * inner / 512) >= 192);
    const bool lx_res2 = lx_ok && (lx_env == 1 || lx_tiles * (C / 512) >= 192);
    const bool lx_res = lx_res1;
    // algorithmic work per valid frame of the two pointwise GEMMs: pw1 C -> 2 inner (reads x_in, writes the
    // SwiGLU product), pw2 inner -> C (reads the depthwise conv's output, the residual stream and the next layer's hoisted
    // conditioner projection, writes x and x_in)
    double lx_fr = (double)B * T;
    if (lx_ragged) { lx_fr = 0; for (int v : h->lens_host) lx_fr += v; }
    const double lx_fl1 = 2.0 * C * 2 * inner, lx_by1 = 4.0 * (C + inner), lx_fl2 = 2.0 * inner * C, lx_by2 = 4.0 * (inner + 4 * C);
    for (int l = 0; l < L; ++l) {
        if (lx_res) {
            LxLayerP p;
            memset(&p, 0, sizeof(p));
            p.A1 = h->blob + h->g_pw1[l].a_off; p.bias1 = h->blob + h->g_pw1[l].bias_off;
            p.A2 = h->blob + h->g_pw2[l].a_off; p.bias2 = h->blob + h->g_pw2[l].bias_off;
            p.xin = h->xin; p.stats = h->stats; p.u = h->ubuf; p.v = h->vbuf; p.x = h->xh;
            p.x_bstride = xs; p.u_bstride = us; p.inner = inner; p.Ts = Ts; p.T = T;
            p.tiles_per_b = (T + 31) / 32; p.inv_tiles_per_b = 1.0f / (float)p.tiles_per_b;
            p.nft = B * p.tiles_per_b;
            if (lx_ragged) { p.cgmap = h->cg_dev[1]; p.ncg = h->cg_n[1]; }
            p.inv_nft = 1.0f / (float)std::max(1, lx_ragged ? p.ncg : p.nft);
            p.strong = h->cfg.strong_cond;
            p.lnpart = h->lnpart; p.lnpart_ts = Ts; p.ln_tiles = ln_tiles;
            p.lnpart_in = h->lnpart;        // (read by pw1 before pw2 of this layer replaces it with the next layer's partials)
            if (!(h->precision == 1 && !h->x3_conv.empty()) && !lx_pw1_merges_stats(p, C)) {      // (both fp32 forms and the bf16x3 one merge their own frames' partials)
                hipError_t me = launch_ln_merge(h->lnpart, ln_tiles, C, B, T, Ts, 1e-5f, h->stats, st);
                if (me != hipSuccess) return fail(h, DSD_EHIP, "LayerNorm merge launch failed: %s", hipGetErrorString(me));
            }
            const int next = l + 1;
            p.xin_out = next < L ? h->xin : nullptr;
            if (next < L) {
                p.cpn = h->cp + (long)next * C * Ts; p.cpn_bstride = (long)L * C * Ts;
                film_of(h, next, film_col0, film_colb, p.film, p.film_cstride, p.film_col0, p.film_colb);
            }
            // split-bf16 mode (lynx_x3.hip): both pointwise GEMMs as weight-stream-bound bf16x3 kernels; pw2 only where its C / 512
            // workgroups per frame tile fill at least half the chip (one utterance: the fp32 128-row kernel is faster)
            const bool x3 = h->precision == 1 && !h->x3_conv.empty();
            const bool x3_pw2 = x3 && lx_tiles * (C / 512) >= h->cus / 2;
            // ... on 64-frame tiles where those still fill the chip: the same weight stream then serves twice the frames
            const long lx_t64 = lx_ragged ? (long)h->cg_n[2] : (long)B * ((T + 63) / 64);
            const int xw = path_opts().x3_wide;
            const bool wide1 = x3 && xw != 0 && (xw == 1 || lx_t64 * (2 * inner / 512) >= h->cus);
            const bool wide2 = x3_pw2 && xw != 0 && (xw == 1 || lx_t64 * (C / 512) >= h->cus);
            auto widen = [&](LxLayerP& q) {                      // the tile bookkeeping of a launch on 64-frame tiles
                q.tiles_per_b = (T + 63) / 64;
                q.inv_tiles_per_b = 1.0f / (float)q.tiles_per_b;
                q.nft = B * q.tiles_per_b;
                if (lx_ragged) { q.cgmap = h->cg_dev[2]; q.ncg = h->cg_n[2]; }
                q.inv_nft = 1.0f / (float)std::max(1, lx_ragged ? q.ncg : q.nft);
            };
            // pw2, fp32: 512-row workgroups (lx_pw2d_kernel, ~130 us per round of one per CU) or 128-row ones (lx_pw2q_kernel, ~40 us
            // per round) - by rounds: between whole rounds of the wide form the narrow one wins (B = 3, 5, 6 at T = 1000: 192 / 320 /
            // 384 wide workgroups for 256 CUs).  DSD_LYNX_PW2Q=0/1 forces.
            bool lx_q_over_d = path_opts().lynx_pw2q == 1;
            if (lx_res2 && path_opts().lynx_pw2q < 0 && lx_pw2q_supported(C, inner)) {
                const long rq = (lx_tiles * (C / 128) + h->cus - 1) / h->cus, rd = (lx_tiles * (C / 512) + h->cus - 1) / h->cus;
                lx_q_over_d = 10 * rq < 33 * rd;
            }
            hipError_t le;
            if (x3) {
                LxLayerP q = p;
                q.A1 = h->blob + h->x3_conv[l];
                if (wide1) widen(q);
                timed_begin(650 + (wide1 ? 1 : 0), lx_fl1 * lx_fr, lx_by1 * lx_fr);
                le = launch_lx_x3(q, 0, C, wide1 ? 4 : 2, st);
                timed_end();
            } else {
                timed_begin(600, lx_fl1 * lx_fr, lx_by1 * lx_fr);
                le = launch_lx_layer(p, 0, C, st);
                timed_end();
            }
            if (le != hipSuccess) return fail(h, DSD_EHIP, "LYNXNet pw1 launch failed: %s", hipGetErrorString(le));
            e = launch_dwconv(h->ubuf, h->vbuf, us, Ts, inner, B, T, h->lens_host.empty() ? nullptr : h->lens_dev,
                              h->blob + h->dw_w[l], h->blob + h->dw_b[l], h->cfg.kernel_size, h->cfg.activation,
                              h->dw_prelu[l] == SIZE_MAX ? nullptr : h->blob + h->dw_prelu[l], st);
            if (e != hipSuccess) return fail(h, DSD_EHIP, "dwconv launch failed: %s", hipGetErrorString(e));
            if (x3_pw2) {
                LxLayerP q = p;
                q.A2 = h->blob + h->x3_out[l];
                if (wide2) widen(q);
                timed_begin(660 + (wide2 ? 1 : 0), lx_fl2 * lx_fr, lx_by2 * lx_fr);
                le = launch_lx_x3(q, 1, C, wide2 ? 4 : 2, st);
                timed_end();
                if (le != hipSuccess) return fail(h, DSD_EHIP, "LYNXNet pw2 (bf16x3) launch failed: %s", hipGetErrorString(le));
            } else if (lx_res2 && !lx_q_over_d) {
                timed_begin(610, lx_fl2 * lx_fr, lx_by2 * lx_fr);
                le = launch_lx_layer(p, 1, C, st);
                timed_end();
                if (le != hipSuccess) return fail(h, DSD_EHIP, "LYNXNet pw2 launch failed: %s", hipGetErrorString(le));
            } else if (path_opts().lynx_pw2q != 0 && lx_pw2q_supported(C, inner)) {
                // one-utterance grids: 128 rows per workgroup, C / 128 workgroups per frame tile (lynx_layer.hip, lx_pw2q_kernel)
                timed_begin(615, lx_fl2 * lx_fr, lx_by2 * lx_fr);
                le = launch_lx_pw2q(p, C, st);
                timed_end();
                if (le != hipSuccess) return fail(h, DSD_EHIP, "LYNXNet pw2 (128-row) launch failed: %s", hipGetErrorString(le));
            } else {
                GemmCall o = make_gemm(h, h->g_pw2[l], h->vbuf, us, Ts, B, T, ST_PLAIN, EP_LYNX_NEXT, 0);
                o.p.act = ACT_NONE;
                o.p.aux = h->xh; o.p.aux_bstride = xs; o.p.aux_rstride = Ts;
                lynx_next(o, l + 1);
                timed_begin(620, lx_fl2 * lx_fr, lx_by2 * lx_fr);
                rc = run_gemm(h, o, st);
                timed_end();
                if (rc) return rc;
            }
            continue;
        }
        GemmCall g = make_gemm(h, h->g_pw1[l], h->xin, xs, Ts, B, T, ST_LN, EP_SWIGLU, 0);
        if ((rc = ln_input(g))) return rc;
        g.p.out = h->ubuf; g.p.o_bstride = us; g.p.o_rstride = Ts;
        timed_begin(630, lx_fl1 * lx_fr, lx_by1 * lx_fr);
        rc = run_gemm(h, g, st);
        timed_end();
        if (rc) return rc;
        e = launch_dwconv(h->ubuf, h->vbuf, us, Ts, inner, B, T, h->lens_host.empty() ? nullptr : h->lens_dev,
                          h->blob + h->dw_w[l], h->blob + h->dw_b[l],
                          h->cfg.kernel_size, h->cfg.activation,
                          h->dw_prelu[l] == SIZE_MAX ? nullptr : h->blob + h->dw_prelu[l], st);
        if (e != hipSuccess) return fail(h, DSD_EHIP, "dwconv launch failed: %s", hipGetErrorString(e));
        GemmCall o = make_gemm(h, h->g_pw2[l], h->vbuf, us, Ts, B, T, ST_PLAIN, EP_LYNX_NEXT, 0);
        o.p.act = ACT_NONE;
        o.p.aux = h->xh; o.p.aux_bstride = xs; o.p.aux_rstride = Ts;
        lynx_next(o, l + 1);
        timed_begin(620, lx_fl2 * lx_fr, lx_by2 * lx_fr);
        rc = run_gemm(h, o, st);
        timed_end();
        if (rc) return rc;
    }
    GemmCall f = make_gemm(h, h->g_out, h->xh, xs, Ts, B, T, ST_LN, EP_LINCOMB, 0);
    if ((rc = ln_input(f))) return rc;
    f.p.nout = nout;
    for (int i = 0; i < nout; ++i) f.p.lo[i] = lo[i];
    f.p.o_bstride = (long)FM * Ts; f.p.o_rstride = Ts;
    return run_gemm(h, f, st);
}

void destroy_graphs(dsd_handle* h) {
    for (auto& kv : h->graphs) {
        if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
        if (kv.second.graph) (void)hipGraphDestroy(kv.second.graph);
    }
    h->graphs.clear();
    h->graph_seen.clear();
}

}  // namespace

// ==========================================================================================
// C ABI
// ==========================================================================================
extern "C" {

int dsd_api_version(void) { return DSD_API_VERSION; }

const char* dsd_last_error(const dsd_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int dsd_create(const dsd_config* cfg, dsd_handle** out) {
    if (!cfg || !out) return fail(nullptr, DSD_EINVAL, "dsd_create: null argument");
    if (cfg->struct_size != (int32_t)sizeof(dsd_config))
        return fail(nullptr, DSD_EINVAL, "dsd_create: struct_size %d != %zu", cfg->struct_size, sizeof(dsd_config));
    if (cfg->backbone != DSD_BACKBONE_WAVENET && cfg->backbone != DSD_BACKBONE_LYNXNET && cfg->backbone != DSD_AUX_CONVNEXT)
        return fail(nullptr, DSD_EINVAL, "dsd_create: unknown backbone %d", cfg->backbone);
    if (cfg->in_dims < 1 || cfg->n_feats < 1 || cfg->num_layers < 1 || cfg->hidden_size < 1)
        return fail(nullptr, DSD_EINVAL, "dsd_create: non-positive dimension");
    if (cfg->backbone == DSD_BACKBONE_WAVENET) {
        // any even count the reference itself can run (SinusoidalPosEmb: two halves of C / 2, exponent / (C / 2 - 1),
        // common_layers.py:275-279); not a multiple of 32: run as the next multiple with zero channels (pad_wavenet_weights)
        if (cfg->num_channels < 4 || cfg->num_channels % 2 != 0)
            return fail(nullptr, DSD_EINVAL, "dsd_create: WaveNet num_channels must be even and >= 4 (got %d)", cfg->num_channels);
    } else if (cfg->num_channels < 32 || cfg->num_channels % 32 != 0) {
        return fail(nullptr, DSD_EINVAL, "dsd_create: num_channels must be a positive multiple of 32 (got %d)",
                    cfg->num_channels);
    }
    if (cfg->backbone == DSD_AUX_CONVNEXT) {
        if (cfg->n_feats != 1) return fail(nullptr, DSD_EINVAL, "dsd_create: the aux decoder has n_feats == 1 (toplevel.py:50)");
        if (cfg->kernel_size < 1 || cfg->kernel_size % 2 == 0 || cfg->kernel_size > 15)
            return fail(nullptr, DSD_EINVAL, "dsd_create: ConvNeXt aux decoder needs an odd kernel_size <= 15");
    } else if (cfg->backbone == DSD_BACKBONE_WAVENET) {
        if (cfg->dilation_cycle_length < 1 || cfg->dilation_cycle_length > 8)
            return fail(nullptr, DSD_EINVAL, "dsd_create: dilation_cycle_length must be in [1, 8]");
    } else {
        if (cfg->expansion_factor < 1 || cfg->kernel_size < 1 || cfg->kernel_size % 2 == 0 || cfg->kernel_size > 63)
            return fail(nullptr, DSD_EINVAL, "dsd_create: LYNXNet needs expansion_factor >= 1 and odd kernel_size <= 63");
        if ((cfg->num_channels * cfg->expansion_factor) % 32 != 0)
            return fail(nullptr, DSD_EINVAL, "dsd_create: num_channels * expansion_factor must be a multiple of 32");
        if (cfg->activation < DSD_ACT_PRELU || cfg->activation > DSD_ACT_RELU)
            return fail(nullptr, DSD_EINVAL, "dsd_create: %d is not a valid activation", cfg->activation);
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, DSD_EHIP, "dsd_create: no HIP device is visible (this library has no CPU path)");
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(nullptr, DSD_EINVAL, "dsd_create: device %d out of range [0, %d)", cfg->device, ndev);
    if (hipSetDevice(cfg->device) != hipSuccess) return fail(nullptr, DSD_EHIP, "dsd_create: hipSetDevice failed");
    hipError_t ie = gemm_init_all();
    if (ie == hipSuccess && cfg->backbone == DSD_BACKBONE_WAVENET) ie = wn_layer_init_all();
    if (ie == hipSuccess && cfg->backbone == DSD_BACKBONE_WAVENET) ie = wn_rowsplit_init_all();
    if (ie == hipSuccess && cfg->backbone == DSD_BACKBONE_WAVENET) ie = wn_rows_init_all();
    if (ie == hipSuccess && cfg->backbone == DSD_BACKBONE_WAVENET) ie = wn_layer_x3_init_all();
    if (ie == hipSuccess && cfg->backbone == DSD_BACKBONE_LYNXNET) ie = lx_x3_init_all();
    if (ie == hipSuccess && cfg->backbone == DSD_BACKBONE_WAVENET) ie = wn_edge_init_all();
    if (ie == hipSuccess && cfg->backbone == DSD_BACKBONE_LYNXNET) ie = lx_layer_init_all();
    if (ie != hipSuccess) return fail(nullptr, DSD_EHIP, "dsd_create: kernel attribute setup failed: %s", hipGetErrorString(ie));
    dsd_handle* h = new dsd_handle();
    h->cfg = *cfg;
    refresh_path_opts();
    h->precision = path_opts().precision == 1 ? 1 : 0;       // DSD_PRECISION=1: split-bf16 layer kernels (dsd_set_precision)
    {
        hipDeviceProp_t prop;
        if (hipGetDeviceProperties(&prop, cfg->device) == hipSuccess && prop.multiProcessorCount > 0) h->cus = prop.multiProcessorCount;
    }
    if (cfg->backbone == DSD_BACKBONE_WAVENET) {      // the kernels run on a multiple of 32 channels (pad_wavenet_weights)
        h->c_user = cfg->num_channels;
        h->cfg.num_channels = (cfg->num_channels + 31) / 32 * 32;
    }
    *out = h;
    return DSD_OK;
}

void dsd_destroy(dsd_handle* h) {
    if (!h) return;
    (void)hipSetDevice(h->cfg.device);
    destroy_graphs(h);
    for (auto& ev : h->ev_pool) {
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
    }
    for (auto& ev : h->cal_pool) {
        (void)hipEventDestroy(ev.first);
        (void)hipEventDestroy(ev.second);
    }
    if (h->blob) (void)hipFree(h->blob);
    if (h->arena) (void)hipFree(h->arena);
    if (h->state) (void)hipFree(h->state);
    if (h->emb_arena) (void)hipFree(h->emb_arena);
    if (h->lens_dev) (void)hipFree(h->lens_dev);
    for (int k = 0; k < 3; ++k)
        if (h->cg_dev[k]) (void)hipFree(h->cg_dev[k]);
    if (h->e_arena) (void)hipFree(h->e_arena);
    if (h->v_arena) (void)hipFree(h->v_arena);
    delete h;
}

int dsd_load_weight(dsd_handle* h, const char* name, const float* data, const int64_t* shape, int32_t ndim,
                    int32_t on_device) {
    if (!h || !name || !data || !shape || ndim < 1 || ndim > 4) return fail(h, DSD_EINVAL, "dsd_load_weight: bad argument");
    const std::string n(name);
    std::vector<int64_t> shp(shape, shape + ndim);
    bool found = false;
    if ((is_enc(h) || is_tok(h)) && n == "encoder.embed_positions._float_tensor")
        return DSD_OK;       // SinusoidalPositionalEmbedding's device/dtype marker buffer (common_layers.py:59): carries no value
    if (n == "diffusion_embedding.freqs" && !is_enc(h) && !is_voc(h) && !is_tok(h)) {
        const int cu = h->c_user ? h->c_user : h->cfg.num_channels;
        if (ndim != 1 || shp[0] != cu / 2)
            return fail(h, DSD_EINVAL, "diffusion_embedding.freqs must have shape [%d]", cu / 2);
        found = true;
    } else {
        for (auto& e : expected_for(h)) {
            if (e.first != n) continue;
            if (e.second != shp) {
                std::string want, got;
                for (auto v : e.second) want += std::to_string(v) + ",";
                for (auto v : shp) got += std::to_string(v) + ",";
                return fail(h, DSD_EINVAL, "size mismatch for %s: expected [%s] got [%s]", name, want.c_str(), got.c_str());
            }
            found = true;
            break;
        }
    }
    if (!found) return fail(h, DSD_ENOTFOUND, "unexpected key in state_dict: %s", name);
    size_t numel = 1;
    for (auto v : shp) numel *= (size_t)v;
    HostTensor t;
    t.shape = shp;
    t.data.resize(numel);
    if (on_device) {
        HIP_OK(h, hipSetDevice(h->cfg.device));
        HIP_OK(h, hipMemcpy(t.data.data(), data, numel * sizeof(float), hipMemcpyDeviceToHost));
    } else {
        memcpy(t.data.data(), data, numel * sizeof(float));
    }
    h->raw[n] = std::move(t);
    h->finalized = false;
    return DSD_OK;
}

int dsd_finalize_weights(dsd_handle* h) {
    if (!h) return DSD_EINVAL;
    std::string missing;
    for (auto& e : expected_for(h))
        if (!h->raw.count(e.first)) missing += (missing.empty() ? "" : ", ") + e.first;
    if (!missing.empty()) return fail(h, DSD_ESTATE, "missing keys in state_dict: %s", missing.c_str());
    HIP_OK(h, hipSetDevice(h->cfg.device));
    int rc = build_packed(h);
    if (rc) return rc;
    destroy_graphs(h);
    if (h->blob) (void)hipFree(h->blob);
    h->blob = nullptr;
    h->blob_floats = h->blob_host.size() + 8192;     // tail guard: the fragment ring reads up to one group (2 x 8 KiB) past the end
    if (hipMalloc(&h->blob, h->blob_floats * sizeof(float)) != hipSuccess)
        return fail(h, DSD_ENOMEM, "hipMalloc(%zu bytes of packed weights) failed", h->blob_floats * 4);
    HIP_OK(h, hipMemset(h->blob, 0, h->blob_floats * sizeof(float)));
    HIP_OK(h, hipMemcpy(h->blob, h->blob_host.data(), h->blob_host.size() * sizeof(float), hipMemcpyHostToDevice));
    std::vector<float>().swap(h->blob_host);
    h->finalized = true;
    h->cond_ready = false;
    return DSD_OK;
}

int dsd_prepare_cond(dsd_handle* h, const float* cond, int32_t B, int32_t T, int64_t stride_b, int64_t stride_h,
                     int64_t stride_t, void* stream) {
    refresh_path_opts();
    if (!h || !cond) return fail(h, DSD_EINVAL, "dsd_prepare_cond: null argument");
    if (is_aux(h)) return fail(h, DSD_ESTATE, "dsd_prepare_cond: this handle is an aux decoder (use dsd_aux_decode)");
    if (is_enc(h) || is_tok(h)) return fail(h, DSD_ESTATE, "dsd_prepare_cond: this handle is an encoder (use dsd_encode / dsd_token_encode)");
    if (is_voc(h)) return fail(h, DSD_ESTATE, "dsd_prepare_cond: this handle is a vocoder (use dsd_vocode)");
    if (!h->finalized) return fail(h, DSD_ESTATE, "dsd_prepare_cond: weights are not finalized");
    if (B < 1 || T < 1) return fail(h, DSD_EINVAL, "dsd_prepare_cond: B and T must be positive (B=%d, T=%d)", B, T);
    if (stride_t != 1 && stride_h != 1)
        return fail(h, DSD_EINVAL, "dsd_prepare_cond: cond must be contiguous along T ([B,H,T]) or along H ([B,T,H])");
    hipStream_t st = (hipStream_t)stream;
    HIP_OK(h, hipSetDevice(h->cfg.device));
    int rc = ensure_workspace(h, B, T, st);
    if (rc) return rc;
    const int H = h->cfg.hidden_size, Ts = h->Ts, L = L_of(h), R = cp_rows(h);
    hipError_t e = launch_pack(cond, stride_b, stride_h, stride_t, h->cond_i, B, H, T, Ts, st);
    if (e != hipSuccess) return fail(h, DSD_EHIP, "pack(cond) launch failed: %s", hipGetErrorString(e));
    GemmCall g = make_gemm(h, h->g_cp, h->cond_i, (long)H * Ts, Ts, B, T, ST_PLAIN, EP_BIAS_ACT, 0);
    g.p.act = ACT_NONE;
    g.p.out = h->cp; g.p.o_bstride = (long)L * R * Ts; g.p.o_rstride = Ts;
    rc = run_gemm(h, g, st);
    if (rc) return rc;
    h->cond_ready = true;
    return DSD_OK;
}

int dsd_encoder_create(const dsd_encoder_config* cfg, dsd_handle** out) {
    if (!cfg || !out) return fail(nullptr, DSD_EINVAL, "dsd_encoder_create: null argument");
    if (cfg->struct_size != (int32_t)sizeof(dsd_encoder_config))
        return fail(nullptr, DSD_EINVAL, "dsd_encoder_create: struct_size %d != %zu", cfg->struct_size, sizeof(dsd_encoder_config));
    if (cfg->vocab_size < 2 || cfg->enc_layers < 1 || cfg->num_heads < 1)
        return fail(nullptr, DSD_EINVAL, "dsd_encoder_create: non-positive dimension");
    if (cfg->hidden_size < 32 || cfg->hidden_size % 32 != 0 || cfg->hidden_size % (2 * cfg->num_heads) != 0)
        return fail(nullptr, DSD_EINVAL, "dsd_encoder_create: hidden_size must be a multiple of 32 and of 2 * num_heads");
    if (cfg->hidden_size / cfg->num_heads > 256 || (cfg->hidden_size / cfg->num_heads) % 8 != 0)
        return fail(nullptr, DSD_EINVAL, "dsd_encoder_create: head dimension must be a multiple of 8, at most 256");
    if (cfg->ffn_kernel_size < 1 || cfg->ffn_kernel_size % 2 == 0 || cfg->ffn_kernel_size > 15)
        return fail(nullptr, DSD_EINVAL, "dsd_encoder_create: enc_ffn_kernel_size must be odd and <= 15");
    if (cfg->num_spk < 0 || cfg->num_lang < 0) return fail(nullptr, DSD_EINVAL, "dsd_encoder_create: negative table size");
    if (cfg->pos_mode < DSD_POS_ROPE || cfg->pos_mode > DSD_POS_SIN) return fail(nullptr, DSD_EINVAL, "dsd_encoder_create: pos_mode must be one of DSD_POS_*");
    if (cfg->ffn_act < DSD_FFN_GELU || cfg->ffn_act > DSD_FFN_SWIGLU) return fail(nullptr, DSD_EINVAL, "dsd_encoder_create: ffn_act must be one of DSD_FFN_*");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, DSD_EHIP, "dsd_encoder_create: no HIP device is visible (this library has no CPU path)");
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(nullptr, DSD_EINVAL, "dsd_encoder_create: device %d out of range [0, %d)", cfg->device, ndev);
    if (hipSetDevice(cfg->device) != hipSuccess) return fail(nullptr, DSD_EHIP, "dsd_encoder_create: hipSetDevice failed");
    hipError_t ie = gemm_init_all();
    if (ie != hipSuccess) return fail(nullptr, DSD_EHIP, "dsd_encoder_create: kernel attribute setup failed: %s", hipGetErrorString(ie));
    dsd_handle* h = new dsd_handle();
    memset(&h->cfg, 0, sizeof(h->cfg));
    h->cfg.struct_size = sizeof(dsd_config);
    h->cfg.backbone = DSD_ENC_FS2_ACOUSTIC;
    h->cfg.in_dims = cfg->vocab_size;
    h->cfg.n_feats = 1;
    h->cfg.num_layers = cfg->enc_layers;
    h->cfg.num_channels = cfg->hidden_size;
    h->cfg.hidden_size = cfg->hidden_size;
    h->cfg.kernel_size = cfg->ffn_kernel_size;
    h->cfg.device = cfg->device;
    h->ecfg = *cfg;
    h->e_ffn_act = cfg->ffn_act;
    *out = h;
    return DSD_OK;
}

// workspace of an encoder pass over (B, L): x, y [H], qkv [3H], mid [4H], nonpad, dur (+ two [Cd] buffers and one [H]
// input buffer for the duration predictor)
static int enc_workspace(dsd_handle* h, int B, int L, int H, int Cd, hipStream_t st) {
    const int Ls = padded_ts(L);
    if (h->e_arena && h->eB == B && h->eL == L && h->dC == Cd) return DSD_OK;
    const size_t per = (size_t)B * Ls;
    size_t off = kGuard;
    auto take = [&](size_t n) {
        size_t o = off;
        off += (n + 63) / 64 * 64 + 64;
        return o;
    };
    const size_t o_x = take(per * H), o_y = take(per * H), o_qkv = take(per * 3 * H),
                 o_mid = take(per * (h->e_ffn_act == DSD_FFN_SWIGLU ? 8 : 4) * H);
    const size_t o_np = take(per), o_dur = take((size_t)B * L);
    const size_t o_da = take(per * Cd), o_db = take(per * Cd), o_di = take(Cd > 0 ? per * H : 0);
    off += kGuard;
    float* a = h->e_arena;      // kept while the new shape fits (every segment of a project has its own token count)
    if (!a || off > h->e_cap) {
        if (h->e_arena) (void)hipFree(h->e_arena);
        h->e_arena = a = nullptr;
        h->e_cap = 0;
        if (hipMalloc(&a, off * sizeof(float)) != hipSuccess)
            return fail(h, DSD_ENOMEM, "hipMalloc of %zu bytes for the encoder workspace failed", off * 4);
        h->e_cap = off;
    }
    if (hipMemsetAsync(a, 0, off * sizeof(float), st) != hipSuccess) return fail(h, DSD_EHIP, "hipMemset(encoder workspace) failed");
    h->e_arena = a;
    h->eB = B; h->eL = L; h->eLs = Ls; h->dC = Cd;
    h->e_x = a + o_x; h->e_y = a + o_y; h->e_qkv = a + o_qkv; h->e_mid = a + o_mid; h->e_nonpad = a + o_np;
    h->e_dur = reinterpret_cast<int*>(a + o_dur);
    h->d_a = a + o_da; h->d_b = a + o_db; h->d_in = a + o_di;
    return DSD_OK;
}

// FastSpeech2Encoder.forward after the embedding (tts_modules.py:412-424): e_x (already masked) -> e_y = LN(x) * nonpad
static int run_fs2_layers(dsd_handle* h, int H, int NL, int heads, int ffn_ks, int B, int L, hipStream_t st) {
    const int Ls = h->eLs;
    const float* blob = h->blob;
    const long xs = (long)H * Ls;
    hipError_t er;
    int rc;
#define ENC_OK(expr, what)                                                                        \
    if ((er = (expr)) != hipSuccess) return fail(h, DSD_EHIP, what " launch failed: %s", hipGetErrorString(er))
    for (int l = 0; l < NL; ++l) {      // EncSALayer.forward  (common_layers.py:236-268)
        ENC_OK(launch_enc_layernorm(h->e_x, h->e_y, blob + h->e_ln1g[l], blob + h->e_ln1b[l], nullptr, H, B, L, Ls, 1e-5f, st),
               "layer_norm1");
        GemmCall q = make_gemm(h, h->g_qkv[l], h->e_y, xs, Ls, B, L, ST_PLAIN, EP_BIAS_ACT, 0);
        q.p.act = ACT_NONE; q.p.out = h->e_qkv; q.p.o_bstride = 3 * xs; q.p.o_rstride = Ls;
        if ((rc = run_gemm(h, q, st))) return rc;
        if (h->e_pos == DSD_POS_ROPE) ENC_OK(launch_enc_rope(h->e_qkv, blob + h->e_freqs, H, H / heads, B, L, Ls, st), "rope");
        ENC_OK(launch_enc_attention(h->e_qkv, h->e_nonpad, h->e_y, H, heads, B, L, Ls, st), "attention");
        GemmCall o = make_gemm(h, h->g_oproj[l], h->e_y, xs, Ls, B, L, ST_PLAIN, EP_BIAS_RES, 0);
        o.p.aux = h->e_x; o.p.aux_bstride = xs; o.p.aux_rstride = Ls;
        o.p.out = h->e_x; o.p.o_bstride = xs; o.p.o_rstride = Ls;
        if ((rc = run_gemm(h, o, st))) return rc;
        ENC_OK(launch_enc_mask(h->e_x, h->e_nonpad, H, B, L, Ls, st), "mask");
        ENC_OK(launch_enc_layernorm(h->e_x, h->e_y, blob + h->e_ln2g[l], blob + h->e_ln2b[l], nullptr, H, B, L, Ls, 1e-5f, st),
               "layer_norm2");
        // TransformerFFNLayer (common_layers.py:142-151): Conv1d(H, 4H, k) * k^-0.5 -> GELU | ReLU | SiLU -> Linear(4H, H);
        // SwiGLU: Conv1d(H, 8H, k), then rows [0, 4H) *= silu(rows [4H, 8H)) in place (common_layers.py:107-117)
        const int fa = h->e_ffn_act;
        const long ms = (fa == DSD_FFN_SWIGLU ? 8 : 4) * xs;
        GemmCall f1 = make_gemm(h, h->g_ffn1[l], h->e_y, xs, Ls, B, L, ST_PLAIN, EP_BIAS_ACT, 1, ffn_ks > 1);
        f1.p.act = fa == DSD_FFN_GELU ? ACT_GELU : fa == DSD_FFN_RELU ? ACT_RELU : fa == DSD_FFN_SWISH ? ACT_SILU : ACT_NONE;
        f1.p.out = h->e_mid; f1.p.o_bstride = ms; f1.p.o_rstride = Ls;
        if ((rc = run_gemm(h, f1, st))) return rc;
        if (fa == DSD_FFN_SWIGLU) ENC_OK(launch_enc_swiglu(h->e_mid, 4 * H, ms, B, L, Ls, st), "swiglu");
        GemmCall f2 = make_gemm(h, h->g_ffn2[l], h->e_mid, ms, Ls, B, L, ST_PLAIN, EP_BIAS_RES, 0);
        f2.p.aux = h->e_x; f2.p.aux_bstride = xs; f2.p.aux_rstride = Ls;
        f2.p.out = h->e_x; f2.p.o_bstride = xs; f2.p.o_rstride = Ls;
        if ((rc = run_gemm(h, f2, st))) return rc;
        ENC_OK(launch_enc_mask(h->e_x, h->e_nonpad, H, B, L, Ls, st), "mask");
    }
    // final LayerNorm * nonpadding  (tts_modules.py:424)
    ENC_OK(launch_enc_layernorm(h->e_x, h->e_y, blob + h->e_lng, blob + h->e_lnb, h->e_nonpad, H, B, L, Ls, 1e-5f, st),
           "final layer_norm");
#undef ENC_OK
    return DSD_OK;
}

int dsd_encode(dsd_handle* h, const int64_t* txt_tokens, const int64_t* mel2ph, const float* f0, int32_t B, int32_t L,
               int32_t T, const dsd_encode_extras* ex, float* cond_out, void* stream) {
    refresh_path_opts();
    if (!h || !txt_tokens || !mel2ph || !f0 || !cond_out) return fail(h, DSD_EINVAL, "dsd_encode: null argument");
    if (!is_enc(h)) return fail(h, DSD_ESTATE, "dsd_encode: this handle is not an encoder");
    if (!h->finalized) return fail(h, DSD_ESTATE, "dsd_encode: weights are not finalized");
    if (B < 1 || L < 1 || T < 1) return fail(h, DSD_EINVAL, "dsd_encode: B, T_txt and T must be positive (%d, %d, %d)", B, L, T);
    if (L > 2048) return fail(h, DSD_EINVAL, "dsd_encode: T_txt = %d tokens exceeds the supported 2048", L);
    const dsd_encoder_config& e = h->ecfg;
    dsd_encode_extras none;
    memset(&none, 0, sizeof(none));
    if (!ex) ex = &none;
    if (e.num_lang > 0 && !ex->languages) return fail(h, DSD_EINVAL, "dsd_encode: use_lang_id model needs `languages`");
    if (e.num_spk > 0 && !ex->spk_embed_id && !ex->spk_mix_embed)
        return fail(h, DSD_EINVAL, "dsd_encode: use_spk_id model needs `spk_embed_id` or `spk_mix_embed`");
    const float* feats[7] = {f0, ex->energy, ex->breathiness, ex->voicing, ex->tension, ex->key_shift, ex->speed};
    for (int k = 1; k < 7; ++k)
        if (lin_present(e, k) && !feats[k]) return fail(h, DSD_EINVAL, "dsd_encode: input for %s is missing", kLinNames[k]);
    hipStream_t st = (hipStream_t)stream;
    HIP_OK(h, hipSetDevice(e.device));
    const int H = e.hidden_size, Ls = padded_ts(L);
    int rc = enc_workspace(h, B, L, H, 0, st);
    if (rc) return rc;
    const float* blob = h->blob;
    hipError_t er;
#define ENC_OK(expr, what)                                                                        \
    if ((er = (expr)) != hipSuccess) return fail(h, DSD_EHIP, what " launch failed: %s", hipGetErrorString(er))
    // mel2ph_to_dur + forward_embedding  (acoustic_encoder.py:89-96, tts_modules.py:385-398)
    ENC_OK(launch_enc_dur((const long long*)mel2ph, B, T, L, h->e_dur, st), "dur");
    ENC_OK(launch_enc_embed((const long long*)txt_tokens, (const long long*)ex->languages, h->e_dur, blob + h->e_txt,
                            e.vocab_size, h->e_lang == SIZE_MAX ? nullptr : blob + h->e_lang, e.num_lang + 1,
                            blob + h->e_durw, blob + h->e_durb, sqrtf((float)H), H, B, L, Ls, h->e_x, h->e_nonpad, st),
           "embed");
    if (e.pos_mode == DSD_POS_REL) {      // x * sqrt(H) + positions, then the padding mask again  (tts_modules.py:390-392,412)
        ENC_OK(launch_enc_relpos(h->e_x, blob + h->e_freqs, H, B, L, Ls, st), "relpos");
        ENC_OK(launch_enc_mask(h->e_x, h->e_nonpad, H, B, L, Ls, st), "mask");
    }
    if (e.pos_mode == DSD_POS_SIN)        // x + positions (zero at padding)  (tts_modules.py:393-395)
        ENC_OK(launch_enc_sinpos(h->e_x, h->e_nonpad, blob + h->e_freqs, H, B, L, Ls, st), "sinpos");
    if ((rc = run_fs2_layers(h, H, e.enc_layers, e.num_heads, e.ffn_kernel_size, B, L, st))) return rc;
    EncExpandArgs a;
    memset(&a, 0, sizeof(a));
    for (int k = 0; k < 7; ++k)
        if (lin_present(e, k)) {
            a.lin_w[k] = blob + h->e_linw[k];
            a.lin_b[k] = blob + h->e_linb[k];
            a.feat[k] = feats[k];
        }
    if (e.num_spk > 0) {
        a.spk_table = blob + h->e_spk;
        a.spk_id = (const long long*)ex->spk_embed_id;
        a.spk_mix = ex->spk_mix_embed;
        a.spk_mix_bstride = ex->spk_mix_bstride;
        a.spk_mix_tstride = ex->spk_mix_tstride;
        a.num_spk = e.num_spk;
    }
    ENC_OK(launch_enc_expand(h->e_y, (const long long*)mel2ph, a, H, B, L, Ls, T, cond_out, st), "expand");
#undef ENC_OK
    return DSD_OK;
}

int dsd_token_encoder_create(const dsd_token_encoder_config* cfg, dsd_handle** out) {
    if (!cfg || !out) return fail(nullptr, DSD_EINVAL, "dsd_token_encoder_create: null argument");
    if (cfg->struct_size != (int32_t)sizeof(dsd_token_encoder_config))
        return fail(nullptr, DSD_EINVAL, "dsd_token_encoder_create: struct_size %d != %zu", cfg->struct_size,
                    sizeof(dsd_token_encoder_config));
    if (cfg->enc_layers < 1 || cfg->num_heads < 1) return fail(nullptr, DSD_EINVAL, "dsd_token_encoder_create: non-positive dimension");
    if (cfg->hidden_size < 32 || cfg->hidden_size % 32 != 0 || cfg->hidden_size % (2 * cfg->num_heads) != 0)
        return fail(nullptr, DSD_EINVAL, "dsd_token_encoder_create: hidden_size must be a multiple of 32 and of 2 * num_heads");
    if (cfg->hidden_size / cfg->num_heads > 256 || (cfg->hidden_size / cfg->num_heads) % 8 != 0)
        return fail(nullptr, DSD_EINVAL, "dsd_token_encoder_create: head dimension must be a multiple of 8, at most 256");
    if (cfg->ffn_kernel_size < 1 || cfg->ffn_kernel_size % 2 == 0 || cfg->ffn_kernel_size > 15)
        return fail(nullptr, DSD_EINVAL, "dsd_token_encoder_create: enc_ffn_kernel_size must be odd and <= 15");
    if (cfg->out_dims < 0 || cfg->dur_layers < 0) return fail(nullptr, DSD_EINVAL, "dsd_token_encoder_create: negative size");
    if (cfg->out_dims > 4 * cfg->hidden_size) return fail(nullptr, DSD_EINVAL, "dsd_token_encoder_create: out_dims above 4 * hidden_size is not supported");
    if (cfg->pos_mode < DSD_POS_ROPE || cfg->pos_mode > DSD_POS_SIN) return fail(nullptr, DSD_EINVAL, "dsd_token_encoder_create: pos_mode must be one of DSD_POS_*");
    if (cfg->dur_layers > 0 && (cfg->dur_chans < 1 || cfg->dur_kernel_size < 1 || cfg->dur_kernel_size % 2 == 0 ||
                                cfg->dur_kernel_size > 15))
        return fail(nullptr, DSD_EINVAL, "dsd_token_encoder_create: duration predictor needs channels >= 1 and an odd kernel size <= 15");
    if (cfg->ffn_act < DSD_FFN_GELU || cfg->ffn_act > DSD_FFN_SWIGLU)
        return fail(nullptr, DSD_EINVAL, "dsd_token_encoder_create: ffn_act must be one of DSD_FFN_*");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, DSD_EHIP, "dsd_token_encoder_create: no HIP device is visible (this library has no CPU path)");
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(nullptr, DSD_EINVAL, "dsd_token_encoder_create: device %d out of range [0, %d)", cfg->device, ndev);
    if (hipSetDevice(cfg->device) != hipSuccess) return fail(nullptr, DSD_EHIP, "dsd_token_encoder_create: hipSetDevice failed");
    hipError_t ie = gemm_init_all();
    if (ie != hipSuccess) return fail(nullptr, DSD_EHIP, "dsd_token_encoder_create: kernel attribute setup failed: %s", hipGetErrorString(ie));
    dsd_handle* h = new dsd_handle();
    memset(&h->cfg, 0, sizeof(h->cfg));
    h->cfg.struct_size = sizeof(dsd_config);
    h->cfg.backbone = DSD_ENC_FS2_TOKENS;
    h->cfg.in_dims = cfg->hidden_size;
    h->cfg.n_feats = 1;
    h->cfg.num_layers = cfg->enc_layers;
    h->cfg.num_channels = cfg->hidden_size;
    h->cfg.hidden_size = cfg->hidden_size;
    h->cfg.kernel_size = cfg->ffn_kernel_size;
    h->cfg.device = cfg->device;
    h->tcfg = *cfg;
    h->e_ffn_act = cfg->ffn_act;
    *out = h;
    return DSD_OK;
}

static int tok_common(dsd_handle* h, const char* who, const void* a, const void* b, const void* c, int B, int L) {
    if (!h || !a || !b || !c) return fail(h, DSD_EINVAL, "%s: null argument", who);
    if (!is_tok(h)) return fail(h, DSD_ESTATE, "%s: this handle is not a token encoder", who);
    if (!h->finalized) return fail(h, DSD_ESTATE, "%s: weights are not finalized", who);
    if (B < 1 || L < 1) return fail(h, DSD_EINVAL, "%s: B and L must be positive (%d, %d)", who, B, L);
    if (L > 2048) return fail(h, DSD_EINVAL, "%s: L = %d tokens exceeds the supported 2048", who, L);
    return DSD_OK;
}

int dsd_token_encode(dsd_handle* h, const float* embed, const uint8_t* padding_mask, int32_t B, int32_t L, float* enc_out,
                     void* stream) {
    refresh_path_opts();
    int rc = tok_common(h, "dsd_token_encode", embed, padding_mask, enc_out, B, L);
    if (rc) return rc;
    const dsd_token_encoder_config& t = h->tcfg;
    hipStream_t st = (hipStream_t)stream;
    HIP_OK(h, hipSetDevice(t.device));
    const int H = t.hidden_size;
    if ((rc = enc_workspace(h, B, L, H, t.dur_layers > 0 ? t.dur_chans : 0, st))) return rc;
    const int Ls = h->eLs;
    hipError_t er;
#define ENC_OK(expr, what)                                                                        \
    if ((er = (expr)) != hipSuccess) return fail(h, DSD_EHIP, what " launch failed: %s", hipGetErrorString(er))
    // x = (embed_scale * main + extra) * nonpadding  (tts_modules.py:401-412), [B, L, H] -> [B][H][Ls]
    ENC_OK(launch_enc_nonpad(padding_mask, B, L, Ls, h->e_nonpad, st), "nonpad");
    ENC_OK(launch_pack(embed, (long)L * H, 1, H, h->e_x, B, H, L, Ls, st), "pack(embed)");
    if (t.pos_mode == DSD_POS_REL) ENC_OK(launch_enc_relpos(h->e_x, h->blob + h->e_freqs, H, B, L, Ls, st), "relpos");
    if (t.pos_mode == DSD_POS_SIN) ENC_OK(launch_enc_sinpos(h->e_x, h->e_nonpad, h->blob + h->e_freqs, H, B, L, Ls, st), "sinpos");
    ENC_OK(launch_enc_mask(h->e_x, h->e_nonpad, H, B, L, Ls, st), "mask");
    if ((rc = run_fs2_layers(h, H, t.enc_layers, t.num_heads, t.ffn_kernel_size, B, L, st))) return rc;
    const float* res = h->e_y;
    int Ho = H;
    if (t.out_dims > 0) {      // MelodyEncoder.out_proj (variance_encoder.py:147)
        GemmCall o = make_gemm(h, h->g_tout, h->e_y, (long)H * Ls, Ls, B, L, ST_PLAIN, EP_BIAS_ACT, 0);
        o.p.act = ACT_NONE; o.p.out = h->e_mid; o.p.o_bstride = (long)t.out_dims * Ls; o.p.o_rstride = Ls;
        if ((rc = run_gemm(h, o, st))) return rc;
        res = h->e_mid;
        Ho = t.out_dims;
    }
    ENC_OK(launch_unpack(res, Ls, enc_out, B, 1, Ho, L, 1, nullptr, nullptr, st), "unpack(enc_out)");
#undef ENC_OK
    return DSD_OK;
}

int dsd_predict_dur(dsd_handle* h, const float* dur_cond, const uint8_t* padding_mask, int32_t B, int32_t L, float* dur_out,
                    void* stream) {
    refresh_path_opts();
    int rc = tok_common(h, "dsd_predict_dur", dur_cond, padding_mask, dur_out, B, L);
    if (rc) return rc;
    const dsd_token_encoder_config& t = h->tcfg;
    if (t.dur_layers < 1) return fail(h, DSD_ESTATE, "dsd_predict_dur: this encoder was created without a duration predictor");
    hipStream_t st = (hipStream_t)stream;
    HIP_OK(h, hipSetDevice(t.device));
    const int H = t.hidden_size, Cd = t.dur_chans;
    if ((rc = enc_workspace(h, B, L, H, Cd, st))) return rc;
    const int Ls = h->eLs;
    const float* blob = h->blob;
    hipError_t er;
#define ENC_OK(expr, what)                                                                        \
    if ((er = (expr)) != hipSuccess) return fail(h, DSD_EHIP, what " launch failed: %s", hipGetErrorString(er))
    ENC_OK(launch_enc_nonpad(padding_mask, B, L, Ls, h->e_nonpad, st), "nonpad");
    ENC_OK(launch_pack(dur_cond, (long)L * H, 1, H, h->d_in, B, H, L, Ls, st), "pack(dur_cond)");
    const float* cur = h->d_in;
    int cin = H;
    for (int l = 0; l < t.dur_layers; ++l) {      // Conv1d -> ReLU -> LayerNorm(channels, eps 1e-12) -> * mask  (tts_modules.py:124-127)
        GemmCall g = make_gemm(h, h->g_dconv[l], cur, (long)cin * Ls, Ls, B, L, ST_PLAIN, EP_BIAS_ACT, 1, t.dur_kernel_size > 1);
        g.p.act = ACT_RELU; g.p.out = h->d_a; g.p.o_bstride = (long)Cd * Ls; g.p.o_rstride = Ls;
        if ((rc = run_gemm(h, g, st))) return rc;
        ENC_OK(launch_enc_layernorm(h->d_a, h->d_b, blob + h->d_lng[l], blob + h->d_lnb[l], h->e_nonpad, Cd, B, L, Ls, 1e-12f, st),
               "dur layer_norm");
        cur = h->d_b;
        cin = Cd;
    }
    ENC_OK(launch_enc_dur_head(cur, blob + h->d_linw, blob + h->d_linb, h->e_nonpad, Cd, B, L, Ls, t.dur_offset, dur_out, st),
           "dur head");
#undef ENC_OK
    return DSD_OK;
}

int dsd_cond_assemble(const dsd_assemble_args* args, float* out, void* stream) {
    if (!args || !out) return fail(nullptr, DSD_EINVAL, "dsd_cond_assemble: null argument");
    if (args->struct_size != (int32_t)sizeof(dsd_assemble_args))
        return fail(nullptr, DSD_EINVAL, "dsd_cond_assemble: struct_size %d != %zu", args->struct_size, sizeof(dsd_assemble_args));
    if (args->B < 1 || args->T < 1 || args->H < 1) return fail(nullptr, DSD_EINVAL, "dsd_cond_assemble: B, T and H must be positive");
    if (args->n_gather < 0 || args->n_gather > DSD_ASSEMBLE_MAX_GATHER || args->n_terms < 0 || args->n_terms > DSD_ASSEMBLE_MAX_TERMS)
        return fail(nullptr, DSD_EINVAL, "dsd_cond_assemble: at most %d gathers and %d terms", DSD_ASSEMBLE_MAX_GATHER, DSD_ASSEMBLE_MAX_TERMS);
    AssembleArgs a;
    memset(&a, 0, sizeof(a));
    a.B = args->B; a.T = args->T; a.H = args->H; a.n_gather = args->n_gather; a.n_terms = args->n_terms;
    for (int g = 0; g < a.n_gather; ++g) {
        if (!args->gather[g].table || !args->gather[g].idx || args->gather[g].rows < 1)
            return fail(nullptr, DSD_EINVAL, "dsd_cond_assemble: gather %d needs a table, an index and rows >= 1", g);
        a.g_table[g] = args->gather[g].table; a.g_bstride[g] = args->gather[g].batch_stride; a.g_rows[g] = args->gather[g].rows;
        a.g_idx[g] = (const long long*)args->gather[g].idx; a.g_off[g] = args->gather[g].idx_offset;
        a.g_scale[g] = args->gather[g].scale; a.g_rowscale[g] = args->gather[g].row_scale;
    }
    for (int k = 0; k < a.n_terms; ++k) {
        if (!args->term[k].v) return fail(nullptr, DSD_EINVAL, "dsd_cond_assemble: term %d has no vector", k);
        a.t_s[k] = args->term[k].s; a.t_v[k] = args->term[k].v;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, DSD_EHIP, "dsd_cond_assemble: no HIP device is visible (this library has no CPU path)");
    if (args->device < 0 || args->device >= ndev) return fail(nullptr, DSD_EINVAL, "dsd_cond_assemble: device %d out of range", args->device);
    if (hipSetDevice(args->device) != hipSuccess) return fail(nullptr, DSD_EHIP, "dsd_cond_assemble: hipSetDevice failed");
    hipError_t er = launch_assemble(a, out, (hipStream_t)stream);
    if (er != hipSuccess) return fail(nullptr, DSD_EHIP, "dsd_cond_assemble: launch failed: %s", hipGetErrorString(er));
    return DSD_OK;
}

static int run_tconv(dsd_handle* h, const PackedTConv& pt, const float* x, float* out, const float* res, int B, int C,
                     int T, int Ts, int dil, float slope_in, int act, hipStream_t st) {
    TConvP p;
    memset(&p, 0, sizeof(p));
    p.W = h->blob + pt.w_off;
    p.bias = h->blob + pt.b_off;
    p.x = x; p.x_bstride = (long)C * Ts; p.x_rstride = Ts;
    p.out = out; p.res = res; p.o_bstride = (long)pt.co_real * Ts; p.o_rstride = Ts;
    p.T = T; p.Ts_out = Ts;
    p.taps = pt.taps; p.dil = dil;
    p.HP = round_up((pt.taps / 2) * dil, 4);
    int SP = 256 + 2 * p.HP;
    while (SP % 32 != 16) SP += 4;
    p.SP = SP;
    p.slope_in = slope_in; p.act = act; p.co_real = pt.co_real;
    p.lds_bytes = tconv_lds_bytes(pt.ci, pt.co, pt.taps, SP);
    if (p.lds_bytes > 160 * 1024) return fail(h, DSD_EINVAL, "few-channel conv needs %d bytes of LDS", p.lds_bytes);
    hipError_t e = launch_tconv(p, pt.ci, pt.co, B, st);
    if (e != hipSuccess) return fail(h, DSD_EHIP, "tconv launch failed: %s", hipGetErrorString(e));
    return DSD_OK;
}

int dsd_vocoder_create(const dsd_vocoder_config* cfg, dsd_handle** out) {
    if (!cfg || !out) return fail(nullptr, DSD_EINVAL, "dsd_vocoder_create: null argument");
    if (cfg->struct_size != (int32_t)sizeof(dsd_vocoder_config))
        return fail(nullptr, DSD_EINVAL, "dsd_vocoder_create: struct_size %d != %zu", cfg->struct_size, sizeof(dsd_vocoder_config));
    if (cfg->num_mels < 1 || cfg->sampling_rate < 1 || cfg->n_ups < 1 || cfg->n_ups > DSD_VOC_MAX_UPS || cfg->n_kernels < 1 ||
        cfg->n_kernels > DSD_VOC_MAX_KERNELS || (cfg->resblock != 1 && cfg->resblock != 2))
        return fail(nullptr, DSD_EINVAL, "dsd_vocoder_create: bad layer counts");
    if (cfg->harmonic_num < 0 || cfg->harmonic_num > 15) return fail(nullptr, DSD_EINVAL, "dsd_vocoder_create: harmonic_num must be in [0, 15]");
    if (!(cfg->noise_sigma >= 0.f)) return fail(nullptr, DSD_EINVAL, "dsd_vocoder_create: noise_sigma must be >= 0");
    if (cfg->mini_nsf && cfg->n_ups < 2) return fail(nullptr, DSD_EINVAL, "dsd_vocoder_create: mini_nsf needs at least two upsampling stages");
    if (cfg->upsample_initial_channel % (1 << cfg->n_ups) != 0 || (cfg->upsample_initial_channel >> cfg->n_ups) < 1)
        return fail(nullptr, DSD_EINVAL, "dsd_vocoder_create: upsample_initial_channel must be divisible by 2^n_ups");
    for (int i = 0; i < cfg->n_ups; ++i) {
        const int u = cfg->upsample_rates[i], k = cfg->upsample_kernel_sizes[i];
        if (u < 1 || k < u || (k - u) % 2 != 0 || k > 8 * u)
            return fail(nullptr, DSD_EINVAL, "dsd_vocoder_create: upsample stage %d (rate %d, kernel %d) is not supported", i, u, k);
    }
    for (int j = 0; j < cfg->n_kernels; ++j) {
        if (cfg->resblock_kernel_sizes[j] < 1 || cfg->resblock_kernel_sizes[j] % 2 == 0 || cfg->resblock_kernel_sizes[j] > 15 ||
            cfg->n_dilations[j] < 1 || cfg->n_dilations[j] > DSD_VOC_MAX_DILS)
            return fail(nullptr, DSD_EINVAL, "dsd_vocoder_create: residual block %d: odd kernel <= 15 and 1..%d dilations", j, DSD_VOC_MAX_DILS);
        for (int d = 0; d < cfg->n_dilations[j]; ++d)
            if (cfg->resblock_dilation_sizes[j][d] < 1 || cfg->resblock_dilation_sizes[j][d] * (cfg->resblock_kernel_sizes[j] / 2) > 48)
                return fail(nullptr, DSD_EINVAL, "dsd_vocoder_create: residual block %d dilation %d reaches beyond 48 frames", j, d);
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(nullptr, DSD_EHIP, "dsd_vocoder_create: no HIP device is visible (this library has no CPU path)");
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(nullptr, DSD_EINVAL, "dsd_vocoder_create: device %d out of range [0, %d)", cfg->device, ndev);
    if (hipSetDevice(cfg->device) != hipSuccess) return fail(nullptr, DSD_EHIP, "dsd_vocoder_create: hipSetDevice failed");
    hipError_t ie = gemm_init_all();
    if (ie == hipSuccess) ie = tconv_init_all();
    if (ie != hipSuccess) return fail(nullptr, DSD_EHIP, "dsd_vocoder_create: kernel attribute setup failed: %s", hipGetErrorString(ie));
    dsd_handle* h = new dsd_handle();
    memset(&h->cfg, 0, sizeof(h->cfg));
    h->cfg.struct_size = sizeof(dsd_config);
    h->cfg.backbone = DSD_VOC_NSF_HIFIGAN;
    h->cfg.in_dims = cfg->num_mels;
    h->cfg.n_feats = 1;
    h->cfg.num_layers = cfg->n_ups;
    h->cfg.num_channels = cfg->upsample_initial_channel;
    h->cfg.hidden_size = cfg->num_mels;
    h->cfg.device = cfg->device;
    h->vcfg = *cfg;
    *out = h;
    return DSD_OK;
}

int dsd_vocode(dsd_handle* h, const float* mel, int32_t B, int32_t T, int64_t stride_b, int64_t stride_m, int64_t stride_t,
               const float* f0, const float* rand_ini, const float* noise, const float* pre_noise, float* wav_out,
               void* stream) {
    refresh_path_opts();
    if (!h || !mel || !f0 || !wav_out) return fail(h, DSD_EINVAL, "dsd_vocode: null argument");
    if (!is_voc(h)) return fail(h, DSD_ESTATE, "dsd_vocode: this handle is not a vocoder");
    if (!h->vcfg.mini_nsf && (!rand_ini || !noise))
        return fail(h, DSD_EINVAL, "dsd_vocode: rand_ini and noise are required (the SineGen source draws them, models.py:145,165)");
    if (h->vcfg.noise_sigma > 0.f && !pre_noise)
        return fail(h, DSD_EINVAL, "dsd_vocode: pre_noise is required when noise_sigma > 0 (models.py:272-273)");
    if (!h->finalized) return fail(h, DSD_ESTATE, "dsd_vocode: weights are not finalized");
    if (B < 1 || T < 1) return fail(h, DSD_EINVAL, "dsd_vocode: B and T must be positive (B=%d, T=%d)", B, T);
    if (stride_t != 1 && stride_m != 1)
        return fail(h, DSD_EINVAL, "dsd_vocode: mel must be contiguous along T ([B,M,T]) or along M ([B,T,M])");
    const dsd_vocoder_config& v = h->vcfg;
    const long upp = voc_upp(v, 0);
    if ((long)T * upp > (1L << 28)) return fail(h, DSD_EINVAL, "dsd_vocode: %ld output samples per utterance is too long", (long)T * upp);
    hipStream_t st = (hipStream_t)stream;
    HIP_OK(h, hipSetDevice(v.device));
    const int NU = v.n_ups, C0 = v.upsample_initial_channel, Ts0 = padded_ts(T);
    // stage lengths
    std::vector<long> len(NU + 1);
    std::vector<int> lts(NU + 1);
    len[0] = T;
    for (int i = 0; i < NU; ++i) len[i + 1] = len[i] * v.upsample_rates[i];
    for (int i = 0; i <= NU; ++i) lts[i] = padded_ts((int)len[i]);
    if (!h->v_arena || h->vB != B || h->vT != T) {
        size_t off = kGuard;
        auto take = [&](size_t n) {
            size_t o = off;
            off += (n + 63) / 64 * 64 + 64;
            return o;
        };
        const size_t o_mel = take((size_t)B * v.num_mels * Ts0), o_pre = take((size_t)B * C0 * Ts0);
        const size_t o_har = take((size_t)B * lts[NU]), o_ph = take((size_t)B * T), o_wav = take((size_t)B * lts[NU]);
        std::vector<size_t> ob;
        for (int i = 0; i < NU; ++i) {
            const size_t n = (size_t)B * voc_stage_channels(v, i) * lts[i + 1];
            for (int k = 0; k < 4; ++k) ob.push_back(take(n));
        }
        off += kGuard;
        float* a = h->v_arena;      // kept while the new shape fits: half a gigabyte per 1000 frames is not re-made per segment
        if (!a || off > h->v_cap) {
            if (h->v_arena) (void)hipFree(h->v_arena);
            h->v_arena = a = nullptr;
            h->v_cap = 0;
            if (hipMalloc(&a, off * sizeof(float)) != hipSuccess)
                return fail(h, DSD_ENOMEM, "hipMalloc of %zu bytes for the vocoder workspace failed", off * 4);
            h->v_cap = off;
        }
        if (hipMemsetAsync(a, 0, off * sizeof(float), st) != hipSuccess) return fail(h, DSD_EHIP, "hipMemset(vocoder workspace) failed");
        h->v_arena = a;
        h->vB = B; h->vT = T;
        h->v_mel = a + o_mel; h->v_pre_out = a + o_pre; h->v_har = a + o_har; h->v_phase = a + o_ph; h->v_wav = a + o_wav;
        h->v_buf.clear();
        for (size_t o : ob) h->v_buf.push_back(a + o);
    }
    const float* blob = h->blob;
    hipError_t er;
    int rc;
#define VOC_OK(expr, what)                                                                        \
    if ((er = (expr)) != hipSuccess) return fail(h, DSD_EHIP, what " launch failed: %s", hipGetErrorString(er))
    // source: harmonic-plus-noise at the output rate (models.py:120-168, 200-203, 266), or - mini_nsf - one interpolated
    // sine at the rate after the second upsampling (models.py:215-217, 251-260, 264)
    const long src_len = v.mini_nsf ? len[NU >= 2 ? 2 : NU] : len[NU];
    if (v.mini_nsf) {
        const int src_upp = (int)(src_len / T);
        VOC_OK(launch_voc_fast_source(f0, B, T, src_upp, (float)v.sampling_rate / (float)voc_upp(v, 2), h->v_phase, lts[NU],
                                      h->v_har, st), "source");
    } else {
        VOC_OK(launch_voc_source(f0, rand_ini, noise, blob + h->v_linw, blob + h->v_linb, B, T, (int)upp, v.harmonic_num + 1,
                                 (float)v.sampling_rate, 0.1f, 0.003f, h->v_phase, lts[NU], h->v_har, st), "source");
    }
    // conv_pre  (models.py:227, 267)
    VOC_OK(launch_pack(mel, stride_b, stride_m, stride_t, h->v_mel, B, v.num_mels, T, Ts0, st), "pack(mel)");
    {
        GemmCall g = make_gemm(h, h->v_pre, h->v_mel, (long)v.num_mels * Ts0, Ts0, B, T, ST_PLAIN, EP_BIAS_ACT, 1, true);
        g.p.act = ACT_NONE; g.p.out = h->v_pre_out; g.p.o_bstride = (long)C0 * Ts0; g.p.o_rstride = Ts0;
        if ((rc = run_gemm(h, g, st))) return rc;
    }
    if (v.noise_sigma > 0.f)      // x += noise_sigma * randn_like(x)  (models.py:272-273)
        VOC_OK(launch_voc_add_noise(h->v_pre_out, pre_noise, B, C0, T, Ts0, v.noise_sigma, st), "pre-noise");
    const float* cur = h->v_pre_out;
    int cur_c = C0;
    for (int i = 0; i < NU; ++i) {
        const int ch = voc_stage_channels(v, i), u = v.upsample_rates[i];
        const int Tin = (int)len[i], Tsi = lts[i], Tq = (int)len[i + 1], Tsq = lts[i + 1];
        float* x = h->v_buf[4 * i + 0];
        float* t1 = h->v_buf[4 * i + 1];
        float* r = h->v_buf[4 * i + 2];
        float* acc = h->v_buf[4 * i + 3];
        const long xs = (long)ch * Tsq;
        {   // leaky_relu -> ConvTranspose1d  (models.py:271-272)
            GemmCall g = make_gemm(h, h->v_ups[i], cur, (long)cur_c * Tsi, Tsi, B, Tin, ST_LRELU, EP_SCATTER, 1, true);
            g.p.in_scale = 0.1f; g.p.C = ch; g.p.up = u;
            g.p.out = x; g.p.o_bstride = xs; g.p.o_rstride = Tsq;
            if ((rc = run_gemm(h, g, st))) return rc;
        }
        if (!v.mini_nsf) {   // + noise_convs[i](har_source)  (models.py:274-276)
            const int sf = (int)voc_upp(v, i + 1);
            const int ksz = i + 1 < NU ? 2 * sf : 1;
            VOC_OK(launch_voc_noise_conv(x, h->v_har, blob + h->v_nw[i], blob + h->v_nb[i], B, ch, Tq, Tsq, sf, ksz,
                                         len[NU], lts[NU], st), "noise conv");
        } else if (i == 1) {  // + source_conv(har_source): k = 1, same rate  (models.py:277-279)
            VOC_OK(launch_voc_noise_conv(x, h->v_har, blob + h->v_nw[i], blob + h->v_nb[i], B, ch, Tq, Tsq, 1, 1, src_len,
                                         lts[NU], st), "source conv");
        }
        for (int j = 0; j < v.n_kernels; ++j) {     // residual blocks  (models.py:280-286; ResBlock1 :62-69, ResBlock2 :92-97)
            const auto& cv = h->v_res[(size_t)i * v.n_kernels + j];
            const auto& tv = h->v_rest[(size_t)i * v.n_kernels + j];
            for (int d = 0; d < v.n_dilations[j]; ++d) {
                const float* src = d == 0 ? x : r;
                const int dil = v.resblock_dilation_sizes[j][d];
                if (!tv.empty()) {          // 16 / 32 channels: time-major convolutions (tconv.hip)
                    if (v.resblock == 1) {
                        if ((rc = run_tconv(h, tv[2 * d], src, t1, nullptr, B, ch, Tq, Tsq, dil, 0.1f, ACT_LRELU, st))) return rc;
                        if ((rc = run_tconv(h, tv[2 * d + 1], t1, r, src, B, ch, Tq, Tsq, 1, 1.f, ACT_NONE, st))) return rc;
                    } else {
                        float* dst = src == x ? r : t1;
                        if ((rc = run_tconv(h, tv[d], src, dst, src, B, ch, Tq, Tsq, dil, 0.1f, ACT_NONE, st))) return rc;
                        if (dst == t1) {
                            float* tmp = r; r = t1; t1 = tmp;
                        }
                    }
                    continue;
                }
                if (v.resblock == 1) {
                    GemmCall c1 = make_gemm(h, cv[2 * d], src, xs, Tsq, B, Tq, ST_LRELU, EP_BIAS_ACT, dil, true);
                    c1.p.in_scale = 0.1f; c1.p.act = ACT_LRELU;
                    c1.p.out = t1; c1.p.o_bstride = xs; c1.p.o_rstride = Tsq;
                    if ((rc = run_gemm(h, c1, st))) return rc;
                    GemmCall c2 = make_gemm(h, cv[2 * d + 1], t1, xs, Tsq, B, Tq, ST_PLAIN, EP_BIAS_RES, 1, true);
                    c2.p.aux = src; c2.p.aux_bstride = xs; c2.p.aux_rstride = Tsq;
                    c2.p.out = r; c2.p.o_bstride = xs; c2.p.o_rstride = Tsq;
                    if ((rc = run_gemm(h, c2, st))) return rc;
                } else {
                    // xt + x with x read both as the (leaky-ReLU'd) conv input and as the residual: the output goes to
                    // the other buffer, the conv's halo reads must not see this launch's own stores
                    float* dst = (src == x || src == t1) ? r : t1;
                    GemmCall c1 = make_gemm(h, cv[d], src, xs, Tsq, B, Tq, ST_LRELU, EP_BIAS_RES, dil, true);
                    c1.p.in_scale = 0.1f;
                    c1.p.aux = src; c1.p.aux_bstride = xs; c1.p.aux_rstride = Tsq;
                    c1.p.out = dst; c1.p.o_bstride = xs; c1.p.o_rstride = Tsq;
                    if ((rc = run_gemm(h, c1, st))) return rc;
                    if (dst == t1) {        // keep the running state in r for the next iteration / the accumulation
                        float* tmp = r; r = t1; t1 = tmp;
                    }
                }
            }
            VOC_OK(launch_voc_accum(acc, r, (long)B * xs, j == 0, j + 1 == v.n_kernels ? (float)v.n_kernels : 1.f, st), "accumulate");
        }
        cur = acc;
        cur_c = ch;
    }
    {   // leaky_relu (default slope 0.01) -> conv_post -> tanh  (models.py:287-289)
        const int ch = voc_stage_channels(v, NU - 1), Tq = (int)len[NU], Tsq = lts[NU];
        if (h->v_postt.valid()) {
            if ((rc = run_tconv(h, h->v_postt, cur, h->v_wav, nullptr, B, ch, Tq, Tsq, 1, 0.01f, ACT_TANH, st))) return rc;
        } else {
            GemmCall g = make_gemm(h, h->v_post, cur, (long)ch * Tsq, Tsq, B, Tq, ST_LRELU, EP_BIAS_ACT, 1, true);
            g.p.in_scale = 0.01f; g.p.act = ACT_TANH;
            g.p.out = h->v_wav; g.p.o_bstride = Tsq; g.p.o_rstride = Tsq;
            if ((rc = run_gemm(h, g, st))) return rc;
        }
        VOC_OK(launch_unpack(h->v_wav, Tsq, wav_out, B, 1, 1, Tq, 0, nullptr, nullptr, st), "unpack(wav)");
    }
#undef VOC_OK
    return DSD_OK;
}

int dsd_aux_decode(dsd_handle* h, const float* cond, int32_t B, int32_t T, int64_t stride_b, int64_t stride_h,
                   int64_t stride_t, float* out, const float* out_scale, const float* out_shift, void* stream) {
    refresh_path_opts();
    if (!h || !cond || !out) return fail(h, DSD_EINVAL, "dsd_aux_decode: null argument");
    if (!is_aux(h)) return fail(h, DSD_ESTATE, "dsd_aux_decode: this handle is a denoiser backbone");
    if (!h->finalized) return fail(h, DSD_ESTATE, "dsd_aux_decode: weights are not finalized");
    if (B < 1 || T < 1) return fail(h, DSD_EINVAL, "dsd_aux_decode: B and T must be positive (B=%d, T=%d)", B, T);
    if (stride_t != 1 && stride_h != 1)
        return fail(h, DSD_EINVAL, "dsd_aux_decode: cond must be contiguous along T ([B,H,T]) or along H ([B,T,H])");
    hipStream_t st = (hipStream_t)stream;
    HIP_OK(h, hipSetDevice(h->cfg.device));
    int rc = ensure_workspace(h, B, T, st);
    if (rc) return rc;
    if ((rc = check_lens(h, "dsd_aux_decode", B, T, st))) return rc;
    RaggedScope ragged_scope(h);
    const int H = h->cfg.hidden_size, Ts = h->Ts, L = L_of(h), C = C_of(h), M = FM_of(h);
    const long xs = (long)C * Ts, us = (long)4 * C * Ts;
    hipError_t e = launch_pack(cond, stride_b, stride_h, stride_t, h->cond_i, B, H, T, Ts, st);
    if (e != hipSuccess) return fail(h, DSD_EHIP, "pack(cond) launch failed: %s", hipGetErrorString(e));
    {   // inconv: Conv1d(H, C, k, padding=(k-1)//2)   convnext.py:63-66,80
        GemmCall g = make_gemm(h, h->g_ain, h->cond_i, (long)H * Ts, Ts, B, T, ST_PLAIN, EP_BIAS_ACT, 1);
        g.p.act = ACT_NONE; g.p.out = h->xh; g.p.o_bstride = xs; g.p.o_rstride = Ts;
        if ((rc = run_gemm(h, g, st))) return rc;
    }
    for (int l = 0; l < L; ++l) {      // ConvNeXtBlock.forward   convnext.py:40-56
        e = launch_dwconv(h->xh, h->xin, xs, Ts, C, B, T, h->lens_host.empty() ? nullptr : h->lens_dev,
                          h->blob + h->dw_w[l], h->blob + h->dw_b[l], 7, 3, nullptr, st);
        if (e != hipSuccess) return fail(h, DSD_EHIP, "dwconv launch failed: %s", hipGetErrorString(e));
        e = launch_lynx_pre(h->xin, nullptr, nullptr, 0, nullptr, 0, 0, 0, xs, Ts, C, B, T, 0, h->stats, Ts, 1e-6f, st);
        if (e != hipSuccess) return fail(h, DSD_EHIP, "LayerNorm stats launch failed: %s", hipGetErrorString(e));
        GemmCall g = make_gemm(h, h->g_pw1[l], h->xin, xs, Ts, B, T, ST_LN, EP_BIAS_ACT, 0);
        g.p.ln_stats = h->stats; g.p.ln_ts = Ts;
        g.p.act = ACT_GELU; g.p.out = h->ubuf; g.p.o_bstride = us; g.p.o_rstride = Ts;
        if ((rc = run_gemm(h, g, st))) return rc;
        GemmCall o = make_gemm(h, h->g_pw2[l], h->ubuf, us, Ts, B, T, ST_PLAIN, EP_BIAS_RES, 0);
        o.p.aux = h->xh; o.p.aux_bstride = xs; o.p.aux_rstride = Ts;
        o.p.out = h->xh; o.p.o_bstride = xs; o.p.o_rstride = Ts;
        if ((rc = run_gemm(h, o, st))) return rc;
    }
    {   // outconv: Conv1d(C, M, k)   convnext.py:73-76,83
        GemmCall g = make_gemm(h, h->g_aout, h->xh, xs, Ts, B, T, ST_PLAIN, EP_BIAS_ACT, 1);
        g.p.act = ACT_NONE; g.p.out = h->io_out; g.p.o_bstride = (long)M * Ts; g.p.o_rstride = Ts;
        if ((rc = run_gemm(h, g, st))) return rc;
    }
    // [B][M][Ts] -> [B,T,M] with the denorm affine of AuxDecoderAdaptor.denorm_spec (aux_decoder/__init__.py:53-56)
    e = launch_unpack(h->io_out, Ts, out, B, 1, M, T, 1, out_scale, out_shift, st);
    if (e != hipSuccess) return fail(h, DSD_EHIP, "unpack launch failed: %s", hipGetErrorString(e));
    return DSD_OK;
}

int dsd_denoise(dsd_handle* h, const float* x, const float* t, int32_t t_len, float* out, void* stream) {
    refresh_path_opts();
    if (!h || !x || !t || !out) return fail(h, DSD_EINVAL, "dsd_denoise: null argument");
    if (!h->cond_ready) return fail(h, DSD_ESTATE, "dsd_denoise: call dsd_prepare_cond first");
    if (x == out) return fail(h, DSD_EINVAL, "dsd_denoise: out must not alias x");
    if (t_len != 1 && t_len != h->B) return fail(h, DSD_EINVAL, "dsd_denoise: t_len must be 1 or B=%d (got %d)", h->B, t_len);
    hipStream_t st = (hipStream_t)stream;
    HIP_OK(h, hipSetDevice(h->cfg.device));
    const int B = h->B, T = h->T, Ts = h->Ts, FM = FM_of(h);
    int rc = check_lens(h, "dsd_denoise", B, T, st);
    if (rc) return rc;
    if ((rc = ensure_emb(h, t_len))) return rc;
    HIP_OK(h, hipMemcpyAsync(h->t_dev, t, sizeof(float) * t_len, hipMemcpyDeviceToDevice, st));
    if ((rc = run_step_tables(h, t_len, st))) return rc;
    hipError_t e = launch_pack(x, (long)FM * T, T, 1, h->io_in, B, FM, T, Ts, st);
    if (e != hipSuccess) return fail(h, DSD_EHIP, "pack(x) launch failed: %s", hipGetErrorString(e));
    LinOut lo;
    memset(&lo, 0, sizeof(lo));
    lo.dst = h->io_out;
    lo.nterms = 1;
    lo.t[0].ptr = nullptr;
    lo.t[0].coef = 1.f;
    rc = run_backbone(h, h->io_in, 0, t_len == 1 ? 0 : 1, &lo, 1, st);
    if (rc) return rc;
    e = launch_unpack(h->io_out, Ts, out, B, h->cfg.n_feats, h->cfg.in_dims, T, 0, nullptr, nullptr, st);
    if (e != hipSuccess) return fail(h, DSD_EHIP, "unpack launch failed: %s", hipGetErrorString(e));
    return DSD_OK;
}

int dsd_sample(dsd_handle* h, const dsd_program* prog, const float* x_init, const float* noise, float* out,
               const float* out_scale, const float* out_shift, uint32_t flags, void* stream) {
    refresh_path_opts();
    if (!h || !prog || !x_init || !out) return fail(h, DSD_EINVAL, "dsd_sample: null argument");
    if (!h->cond_ready) return fail(h, DSD_ESTATE, "dsd_sample: call dsd_prepare_cond first");
    if (prog->n_bufs < 1 || prog->n_bufs > 64 || prog->n_evals < 0 || (prog->n_evals > 0 && !prog->evals))
        return fail(h, DSD_EINVAL, "dsd_sample: malformed program");
    if (prog->result_buf < 0 || prog->result_buf >= prog->n_bufs) return fail(h, DSD_EINVAL, "dsd_sample: bad result_buf");
    if (prog->n_noise > 0 && !noise) return fail(h, DSD_EINVAL, "dsd_sample: program references noise but noise == NULL");
    hipStream_t st = (hipStream_t)stream;
    HIP_OK(h, hipSetDevice(h->cfg.device));
    const int B = h->B, T = h->T, Ts = h->Ts, FM = FM_of(h);
    // validate the program before anything is launched
    for (int i = 0; i < prog->n_evals; ++i) {
        const dsd_eval& ev = prog->evals[i];
        if (ev.x_buf < 0 || ev.x_buf >= prog->n_bufs) return fail(h, DSD_EINVAL, "eval %d: bad x_buf %d", i, ev.x_buf);
        if (ev.n_out < 1 || ev.n_out > DSD_MAX_OUT) return fail(h, DSD_EINVAL, "eval %d: bad n_out %d", i, ev.n_out);
        for (int o = 0; o < ev.n_out; ++o) {
            const dsd_lincomb& lc = ev.out[o];
            if (lc.dst < 0 || lc.dst >= prog->n_bufs) return fail(h, DSD_EINVAL, "eval %d out %d: bad dst %d", i, o, lc.dst);
            if (lc.n_terms < 1 || lc.n_terms > DSD_MAX_TERMS) return fail(h, DSD_EINVAL, "eval %d out %d: bad n_terms", i, o);
            for (int k = 0; k < lc.n_terms; ++k) {
                const int s = lc.terms[k].src;
                const bool ok = (s >= 0 && s < prog->n_bufs) || s == DSD_SRC_MODEL ||
                                (s <= DSD_SRC_NOISE_BASE && DSD_SRC_NOISE_BASE - s < prog->n_noise);
                if (!ok) return fail(h, DSD_EINVAL, "eval %d out %d term %d: bad src %d", i, o, k, s);
            }
        }
    }
    int rc = check_lens(h, "dsd_sample", B, T, st);
    if (rc) return rc;
    if ((rc = ensure_state(h, prog->n_bufs, st))) return rc;
    if (prog->n_evals > 0 && (rc = ensure_emb(h, prog->n_evals))) return rc;

    // x_T (or the shallow-diffusion start) -> state buffer 0
    hipError_t e = launch_pack(x_init, (long)FM * T, T, 1, state_buf(h, 0), B, FM, T, Ts, st);
    if (e != hipSuccess) return fail(h, DSD_EHIP, "pack(x_init) launch failed: %s", hipGetErrorString(e));

    if (prog->n_evals > 0) {
        h->t_host.resize(prog->n_evals);
        for (int i = 0; i < prog->n_evals; ++i) h->t_host[i] = prog->evals[i].t;
        HIP_OK(h, hipMemcpyAsync(h->t_dev, h->t_host.data(), sizeof(float) * prog->n_evals, hipMemcpyHostToDevice, st));
    }

    const long ext_b = (long)FM * T;
    auto body = [&](hipStream_t s) -> int {
        int r = DSD_OK;
        h->edge_xh_src = nullptr;
        if (prog->n_evals > 0 && (r = run_step_tables(h, prog->n_evals, s))) return r;
        for (int i = 0; i < prog->n_evals; ++i) {
            const dsd_eval& ev = prog->evals[i];
            LinOut lo[kMaxOut];
            memset(lo, 0, sizeof(lo));
            for (int o = 0; o < ev.n_out; ++o) {
                lo[o].dst = state_buf(h, ev.out[o].dst);
                lo[o].nterms = ev.out[o].n_terms;
                for (int k = 0; k < ev.out[o].n_terms; ++k) {
                    const dsd_term& tm = ev.out[o].terms[k];
                    LinTerm& lt = lo[o].t[k];
                    lt.coef = tm.coef;
                    if (tm.src == DSD_SRC_MODEL) {
                        lt.ptr = nullptr;
                    } else if (tm.src >= 0) {
                        lt.ptr = state_buf(h, tm.src);
                        lt.bstride = (long)FM * Ts;
                        lt.rstride = Ts;
                    } else {
                        lt.ptr = noise + (long)(DSD_SRC_NOISE_BASE - tm.src) * B * ext_b;
                        lt.bstride = ext_b;
                        lt.rstride = T;
                        lt.ext = 1;
                    }
                }
            }
            const float* next_xin = i + 1 < prog->n_evals ? state_buf(h, prog->evals[i + 1].x_buf) : nullptr;
            if ((r = run_backbone(h, state_buf(h, ev.x_buf), i, 0, lo, ev.n_out, s, next_xin))) return r;
        }
        return r;
    };

    const bool use_graph = (flags & DSD_SAMPLE_GRAPH) && !h->timing && prog->n_evals > 0;
    if (!use_graph) {
        if ((rc = body(st))) return rc;
    } else {
        // key: program bytes + noise pointer (baked into kernel arguments)
        std::string key((const char*)prog->evals, sizeof(dsd_eval) * prog->n_evals);
        key.append((const char*)&prog->n_bufs, sizeof(int32_t));
        key.append((const char*)&noise, sizeof(noise));
        key.append((const char*)&B, sizeof(B));             // batch shape: grids and strides are baked into the launches
        key.append((const char*)&T, sizeof(T));
        key.append((const char*)&path_opts(), sizeof(PathOpts));   // the path switches: a graph is ONE set of launch choices
        key.append((const char*)&h->precision, sizeof(int));
        key.push_back(h->lens_host.empty() ? 'd' : 'r');       // dense / ragged: other kernels, and grids that follow
        for (int v : h->lens_host) key.append((const char*)&v, sizeof(v));      // the lengths (baked into the launches)
        auto it = h->graphs.find(key);
        if (it == h->graphs.end() && (flags & DSD_SAMPLE_GRAPH_LAZY) && !h->graph_seen.count(key)) {
            // DSD_SAMPLE_GRAPH_LAZY, first sight of this (program, batch shape): run it eagerly; the graph is built when the
            // same key comes back
            if (h->graph_seen.size() >= 64) h->graph_seen.clear();
            h->graph_seen.insert(key);
            if ((rc = body(st))) return rc;
        } else {
        if (it == h->graphs.end()) {
            hipStream_t cs = nullptr;
            HIP_OK(h, hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
            hipError_t ce = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
            if (ce != hipSuccess) {
                (void)hipStreamDestroy(cs);
                return fail(h, DSD_EHIP, "hipStreamBeginCapture failed: %s", hipGetErrorString(ce));
            }
            rc = body(cs);
            GraphEntry ge;
            ce = hipStreamEndCapture(cs, &ge.graph);
            (void)hipStreamDestroy(cs);
            if (rc) {
                if (ge.graph) (void)hipGraphDestroy(ge.graph);
                return rc;
            }
            if (ce != hipSuccess) return fail(h, DSD_EHIP, "hipStreamEndCapture failed: %s", hipGetErrorString(ce));
            ce = hipGraphInstantiate(&ge.exec, ge.graph, nullptr, nullptr, 0);
            if (ce != hipSuccess) {
                (void)hipGraphDestroy(ge.graph);
                return fail(h, DSD_EHIP, "hipGraphInstantiate failed: %s", hipGetErrorString(ce));
            }
            if (h->graphs.size() >= 16) destroy_graphs(h);
            it = h->graphs.emplace(key, ge).first;
        }
        HIP_OK(h, hipGraphLaunch(it->second.exec, st));
        }
    }
    e = launch_unpack(state_buf(h, prog->result_buf), Ts, out, B, h->cfg.n_feats, h->cfg.in_dims, T,
                      (flags & DSD_SAMPLE_TRANSPOSE) ? 1 : 0, (flags & DSD_SAMPLE_TRANSPOSE) ? out_scale : nullptr,
                      (flags & DSD_SAMPLE_TRANSPOSE) ? out_shift : nullptr, st);
    if (e != hipSuccess) return fail(h, DSD_EHIP, "unpack launch failed: %s", hipGetErrorString(e));
    return DSD_OK;
}

int dsd_set_precision(dsd_handle* h, int32_t mode) {
    if (!h) return DSD_EINVAL;
    if (mode != DSD_PRECISION_F32 && mode != DSD_PRECISION_BF16X3)
        return fail(h, DSD_EINVAL, "dsd_set_precision: unknown mode %d", mode);
    if (!is_wavenet(h) && h->cfg.backbone != DSD_BACKBONE_LYNXNET)
        return fail(h, DSD_ESTATE, "dsd_set_precision: only denoiser handles have a split-bf16 path");
    if (mode == h->precision) return DSD_OK;
    h->precision = mode;
    if (h->finalized) {                   // the bf16x3 weight streams are built with the packed weights
        h->finalized = false;
        destroy_graphs(h);
        return dsd_finalize_weights(h);
    }
    return DSD_OK;
}

int dsd_set_lengths(dsd_handle* h, const int32_t* lengths, int32_t B, void* stream) {
    if (!h) return DSD_EINVAL;
    if (is_enc(h) || is_tok(h) || is_voc(h))
        return fail(h, DSD_ESTATE, "dsd_set_lengths: only denoiser and aux-decoder handles take ragged batches");
    if (!lengths) {             // back to dense batches
        h->lens_host.clear();
        return DSD_OK;
    }
    if (B < 1) return fail(h, DSD_EINVAL, "dsd_set_lengths: B must be positive (%d)", B);
    for (int b = 0; b < B; ++b)
        if (lengths[b] < 0) return fail(h, DSD_EINVAL, "dsd_set_lengths: lengths[%d] = %d is negative", b, lengths[b]);
    HIP_OK(h, hipSetDevice(h->cfg.device));
    if (h->lens_cap < B) {
        if (h->lens_dev) (void)hipFree(h->lens_dev);
        h->lens_dev = nullptr;
        h->lens_cap = 0;
        if (hipMalloc(&h->lens_dev, sizeof(int) * (size_t)B) != hipSuccess)
            return fail(h, DSD_ENOMEM, "hipMalloc of %d lengths failed", B);
        h->lens_cap = B;
        destroy_graphs(h);      // cached graphs captured the old pointer
    }
    h->lens_host.assign(lengths, lengths + B);
    h->cg_T = -1;               // the valid-tile lists follow the lengths
    // stream-ordered behind earlier launches that still read the old values
    HIP_OK(h, hipMemcpyAsync(h->lens_dev, h->lens_host.data(), sizeof(int) * (size_t)B, hipMemcpyHostToDevice, (hipStream_t)stream));
    return DSD_OK;
}

int dsd_get_stats(const dsd_handle* h, dsd_stats* out) {
    if (!h || !out) return DSD_EINVAL;
    memset(out, 0, sizeof(*out));
    refresh_path_opts();
    const int64_t C = h->c_user ? h->c_user : C_of(h), M = FM_of(h), L = L_of(h);
    out->weight_bytes = (int64_t)h->blob_floats * 4;
    out->workspace_bytes = (int64_t)h->arena_floats * 4;
    if (is_enc(h) || is_tok(h)) {        // per TOKEN per encoder pass (attention excluded: it depends on the sequence length)
        const int64_t H = h->cfg.hidden_size, ks = h->cfg.kernel_size, NL = h->cfg.num_layers;
        out->flops_per_frame_nfe = 2 * NL * (3 * H * H + H * H + ks * H * 4 * H + 4 * H * H);
        out->bytes_per_frame_nfe = 0;
        out->kernels_per_nfe = 3 + 11 * (int)NL + 2;
    } else if (is_aux(h)) {        // one pass per utterance, not per NFE: the "_nfe" fields are per decoder pass here
        const int64_t H = h->cfg.hidden_size, ks = h->cfg.kernel_size;
        out->flops_per_frame_nfe = 2 * (ks * H * C + L * (7 * C + C * 4 * C + 4 * C * C) + ks * C * M);
        out->bytes_per_frame_nfe = 4 * (H + M);
        out->kernels_per_nfe = 2 + 4 * (int)L + 2;
    } else if (is_wavenet(h)) {
        // 2*(M*C + L*(3*C*2C + C*2C) + C*C + C*M); bytes L*24C + 2*4*M
        out->flops_per_frame_nfe = 2 * (M * C + L * (3 * C * 2 * C + C * 2 * C) + C * C + C * M);
        out->bytes_per_frame_nfe = L * 24 * C + 8 * M;
        // around the layers: the edge kernel (skip projection, output projection + solver update, the next evaluation's input
        // projection; wn_edge.hip) or the three GEMMs of gemm.hip
        const int edge_env = edge_choice();
        long t32 = (long)h->B * ((h->T + 31) / 32);
        if (!h->lens_host.empty()) { t32 = 0; for (int v : h->lens_host) t32 += (v + 31) / 32; }
        const bool edge = edge_env != 0 && wn_edge_supported(C_of(h), FM_of(h)) && h->arena && (edge_env == 1 || t32 >= 128);
        std::vector<WnSeg> plan;
        int per_layer = 2;                          // the row-split pair or the two GEMMs
        const long tiles = wn_tiles32(h);
        out->split_tiles = (int32_t)tiles;
        if (h->arena && wn_plan_for(h, plan)) {
            per_layer = 0;
            out->split_tiles = 0;
            for (const WnSeg& sg : plan) {
                per_layer += sg.kind == WN_ROWSPLIT ? 2 : 1;
                (sg.kind == WN_ROWSPLIT ? out->split_tiles : out->fused_tiles) += sg.nt;
                if (sg.kind == WN_FUSED_X3) out->precision = DSD_PRECISION_BF16X3;
            }
        }
        out->layer_launches = per_layer;
        out->kernels_per_nfe = per_layer * (int)L + (edge ? 1 : 3);
    } else {
        const int64_t inner = inner_of(h), ks = h->cfg.kernel_size;
        out->flops_per_frame_nfe = 2 * (M * C + L * (C * 2 * inner + ks * inner + inner * C) + C * M);
        out->bytes_per_frame_nfe = L * 12 * C + 8 * M;
        out->kernels_per_nfe = 1 + 4 * (int)L + 2;
        if (h->precision == 1 && !h->x3_conv.empty()) out->precision = DSD_PRECISION_BF16X3;
    }
    out->graphs_cached = (int)h->graphs.size();
    return DSD_OK;
}

int dsd_kernel_timing(dsd_handle* h, int32_t enable) {
    if (!h) return DSD_EINVAL;
    h->timing = enable != 0;
    h->tclasses.clear();
    h->timing_evals = 0;
    if (const char* e = getenv("DSD_TIMING_STRIDE")) h->timing_stride = std::max(1, atoi(e));
    h->ev_used = 0;
    h->cal_used = 0;
    return DSD_OK;
}

int dsd_kernel_timing_classes(dsd_handle* h, dsd_kernel_time* out, int32_t max_classes, int32_t* n_classes,
                              double* empty_pair_ms) {
    if (!h || !out || !n_classes || max_classes < 1) return DSD_EINVAL;
    HIP_OK(h, hipSetDevice(h->cfg.device));
    int n = 0;
    // largest share of the evaluation first
    std::vector<std::pair<double, const dsd_handle::TimedClass*>> order;
    std::vector<double> means(h->tclasses.size(), 0.0);
    for (size_t i = 0; i < h->tclasses.size(); ++i) {
        const auto& c = h->tclasses[i];
        double sum = 0;
        for (size_t idx : c.evs) {
            HIP_OK(h, hipEventSynchronize(h->ev_pool[idx].second));
            float ms = 0.f;
            HIP_OK(h, hipEventElapsedTime(&ms, h->ev_pool[idx].first, h->ev_pool[idx].second));
            sum += ms;
        }
        means[i] = c.evs.empty() ? 0.0 : sum / (double)c.evs.size();
        order.emplace_back(-means[i] * (double)c.launches, &c);
    }
    std::sort(order.begin(), order.end());
    for (auto& kv : order) {
        if (n == max_classes) break;
        const auto& c = *kv.second;
        if (c.evs.empty()) continue;
        dsd_kernel_time& o = out[n++];
        memset(&o, 0, sizeof(o));
        snprintf(o.name, sizeof(o.name), "%s", c.name.c_str());
        o.mean_ms = means[&c - h->tclasses.data()];
        o.launches_timed = (int64_t)c.evs.size();
        o.launches = (int64_t)c.launches;
        o.evaluations = (int64_t)h->timing_evals;
        o.flops_per_launch = c.flops;
        o.bytes_per_launch = c.bytes;
    }
    *n_classes = n;
    if (empty_pair_ms) {
        double cal = 0;
        for (size_t i = 0; i < h->cal_used; ++i) {
            HIP_OK(h, hipEventSynchronize(h->cal_pool[i].second));
            float ms = 0.f;
            HIP_OK(h, hipEventElapsedTime(&ms, h->cal_pool[i].first, h->cal_pool[i].second));
            cal += ms;
        }
        *empty_pair_ms = h->cal_used ? cal / (double)h->cal_used : 0.0;
    }
    return DSD_OK;
}

int dsd_kernel_timing_read(dsd_handle* h, double* mean_ms, double* empty_pair_ms, int64_t* launches) {
    if (!h || !mean_ms || !empty_pair_ms || !launches) return DSD_EINVAL;
    dsd_kernel_time top;
    int32_t n = 0;
    int rc = dsd_kernel_timing_classes(h, &top, 1, &n, empty_pair_ms);      // the class with the largest share of the pass
    if (rc) return rc;
    *launches = n ? top.launches_timed : 0;
    *mean_ms = n ? top.mean_ms : 0.0;
    h->tclasses.clear();
    h->timing_evals = 0;
    h->ev_used = 0;
    h->cal_used = 0;
    return DSD_OK;
}

}  // extern "C"
