/*
 * libdsdenoise - MI355X-native (gfx950) diffusion denoiser for DiffSinger.
 *
 * C-ABI drop-in boundary for the ONE hot path of hrukalive/DiffSinger inference: the backbone
 * forward (WaveNet / LYNXNet) and the sampling loops that call it once per NFE.  Every entry
 * point replaces (is bound in place of) a reference Python interface, cited per function as
 * `path:line` into the reference tree.  Plain pointers and sizes only: no torch types cross
 * this boundary.  All device pointers are fp32, on the HIP device the handle was created for;
 * `stream` is a `hipStream_t` passed as `void*` (NULL = the null stream).
 *
 * Return value: 0 on success, a negative DSD_E* code on failure; nothing is thrown across the
 * ABI.  `dsd_last_error` returns a human-readable message for the last failure on a handle.
 *
 * Threading: like the reference modules (module-level `noise_list`/`bar`,
 * modules/core/ddpm.py:78,276,324) a handle is NOT re-entrant: calls on one handle are
 * serialised by the caller.  One handle per (model, device, process).
 */
#ifndef DSDENOISE_H_
#define DSDENOISE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DSD_API_VERSION 10

/* error codes */
#define DSD_OK 0
#define DSD_EINVAL (-1)   /* bad argument / shape / enum                        */
#define DSD_ESTATE (-2)   /* call order violated (weights missing, no cond ...) */
#define DSD_EHIP (-3)     /* a HIP runtime call failed                          */
#define DSD_ENOMEM (-4)   /* device allocation failed                           */
#define DSD_ENOTFOUND (-5) /* unknown weight name                                */

typedef struct dsd_handle dsd_handle;

/* modules/backbones/__init__.py:6-9  BACKBONES = {'wavenet': WaveNet, 'lynxnet': LYNXNet} */
enum { DSD_BACKBONE_WAVENET = 0, DSD_BACKBONE_LYNXNET = 1,
       /* modules/aux_decoder/__init__.py:7-9  AUX_DECODERS = {'convnext': ConvNeXtDecoder}: not a denoiser - the
          shallow-diffusion aux decoder that produces the loop's start point (see dsd_aux_decode) */
       DSD_AUX_CONVNEXT = 2,
       /* modules/fastspeech/acoustic_encoder.py:14  FastSpeech2Acoustic: the producer of `cond` (see dsd_encode);
          created with dsd_encoder_create, not dsd_create */
       DSD_ENC_FS2_ACOUSTIC = 3,
       /* modules/nsf_hifigan/models.py:207  Generator (NSF-HiFiGAN vocoder: mel + f0 -> waveform; see dsd_vocode);
          created with dsd_vocoder_create */
       DSD_VOC_NSF_HIFIGAN = 4,
       /* modules/fastspeech/tts_modules.py:353  FastSpeech2Encoder on caller-assembled embeddings (+ DurationPredictor /
          out_proj): the encoders of the variance model; created with dsd_token_encoder_create */
       DSD_ENC_FS2_TOKENS = 5 };
/* modules/backbones/lynxnet.py:38-42  activation_classes */
enum { DSD_ACT_PRELU = 0, DSD_ACT_SILU = 1, DSD_ACT_RELU = 2 };

/*
 * Constructor arguments of the backbone, i.e. what
 *   build_backbone(out_dims, num_feats, backbone_type, backbone_args)   modules/backbones/__init__.py:12-18
 * forwards to WaveNet.__init__ (modules/backbones/wavenet.py:52) or
 * LYNXNet.__init__ (modules/backbones/lynxnet.py:91-92), plus hparams['hidden_size']
 * (wavenet.py:65, lynxnet.py:113).
 */
typedef struct dsd_config {
    int32_t struct_size;            /* sizeof(dsd_config), for forward compatibility        */
    int32_t backbone;               /* DSD_BACKBONE_*                                       */
    int32_t in_dims;                /* M: mel bins / repeat bins (`in_dims`)                */
    int32_t n_feats;                /* F: `n_feats`                                         */
    int32_t num_layers;             /* L                                                    */
    int32_t num_channels;           /* C                                                    */
    int32_t hidden_size;            /* H: encoder hidden size of `cond`                     */
    int32_t dilation_cycle_length;  /* WaveNet only                                         */
    int32_t expansion_factor;       /* LYNXNet only                                         */
    int32_t kernel_size;            /* LYNXNet: depthwise conv (odd); aux decoder: in/out conv */
    int32_t activation;             /* LYNXNet only: DSD_ACT_*                              */
    int32_t strong_cond;            /* LYNXNet only: 0/1                                    */
    int32_t device;                 /* HIP device ordinal                                   */
} dsd_config;

/* Replaces: BACKBONES[backbone_type](out_dims, num_feats, **kwargs)  (backbones/__init__.py:16-18). */
int dsd_create(const dsd_config* cfg, dsd_handle** out);
void dsd_destroy(dsd_handle* h);
/* Message for the last failed call on `h` (h == NULL: last failed dsd_create).  Never NULL. */
const char* dsd_last_error(const dsd_handle* h);
int dsd_api_version(void);

/*
 * Replaces: nn.Module.load_state_dict for the backbone (utils/__init__.py:166-222 load_ckpt ->
 * strict load).  `name` is the reference state_dict key relative to the backbone, e.g.
 * "residual_layers.3.dilated_conv.weight" (wavenet.py:22-31,56-72; lynxnet.py:52-62,71-74,104-124);
 * `shape`/`ndim` must equal the reference parameter's shape.  `data` is read during the call
 * (host pointer if on_device == 0, device pointer otherwise).  One extra, non-parameter entry is
 * accepted: "diffusion_embedding.freqs" [C/2], the SinusoidalPosEmb frequency table
 * (common_layers.py:275-276); if it is not supplied it is computed in fp32 on the host.
 */
int dsd_load_weight(dsd_handle* h, const char* name, const float* data, const int64_t* shape,
                    int32_t ndim, int32_t on_device);
/* Strictness check (every parameter present) + re-layout into MFMA fragment order on the device. */
int dsd_finalize_weights(dsd_handle* h);

/*
 * Hoists ResidualBlock.conditioner_projection / LYNXNetResidualLayer.conditioner_projection
 * (wavenet.py:30,35; lynxnet.py:72,77-82) out of the denoise loop: they depend on `cond` only.
 * The reference's own ONNX exporter does the same hoist (utils/onnx_helper.py:267-349).
 * cond element (b, h, t) is read at cond[b*stride_b + h*stride_h + t*stride_t], so both the
 * backbone's [B,H,T] view (wavenet.py:75-81) and GaussianDiffusion.forward's [B,T,H] `condition`
 * (ddpm.py:353-357) can be passed without a transpose.  Also (re)sizes the workspace for (B, T).
 */
int dsd_prepare_cond(dsd_handle* h, const float* cond, int32_t B, int32_t T, int64_t stride_b,
                     int64_t stride_h, int64_t stride_t, void* stream);

/*
 * Replaces: WaveNet.forward / LYNXNet.forward(spec, diffusion_step, cond)
 * (wavenet.py:75-107, lynxnet.py:128-163) for the cond given to the last dsd_prepare_cond.
 * x, out: [B, F, M, T] contiguous device fp32 (out must not alias x: the reference does not
 * mutate its input either).  t: device fp32, t_len == B or 1 (a [1] step is broadcast,
 * reflow.py:135); integer steps are passed as their float value (common_layers.py:277).
 */
int dsd_denoise(dsd_handle* h, const float* x, const float* t, int32_t t_len, float* out, void* stream);

/*
 * Shallow-diffusion aux decoder (the step right before the loop: it produces the `src_spec` the loop starts
 * from).  A handle created with backbone == DSD_AUX_CONVNEXT takes
 *   in_dims = mel bins M (`out_dims`), n_feats = 1, hidden_size = H (`in_dims` of the decoder),
 *   num_channels / num_layers / kernel_size = ConvNeXtDecoder's keyword arguments
 * and the state_dict of ConvNeXtDecoder (modules/aux_decoder/convnext.py:58-76: inconv.*, conv.N.{gamma,
 * dwconv.*, norm.*, pwconv1.*, pwconv2.*}, outconv.*) through dsd_load_weight / dsd_finalize_weights.
 * Replaces: AuxDecoderAdaptor.forward(condition, infer)  (modules/aux_decoder/__init__.py:58-71) =
 * ConvNeXtDecoder.forward (convnext.py:78-85) + denorm_spec (:53-56).
 *   cond  element (b, h, t) at cond[b*stride_b + h*stride_h + t*stride_t]  ([B,T,H] `condition` or [B,H,T])
 *   out   [B, T, M] contiguous;  out = y * out_scale[m] + out_shift[m]   (NULL, NULL = raw decoder output)
 */
int dsd_aux_decode(dsd_handle* h, const float* cond, int32_t B, int32_t T, int64_t stride_b, int64_t stride_h,
                   int64_t stride_t, float* out, const float* out_scale, const float* out_shift, void* stream);

/*
 * FastSpeech2 acoustic encoder: phoneme tokens + durations + f0 -> `condition` [B, T, H], the tensor every entry
 * point above consumes.  Constructor arguments = what FastSpeech2Acoustic.__init__ reads from hparams
 * (modules/fastspeech/acoustic_encoder.py:15-63) in the reference fork's rotary-embedding configuration
 * (`use_pos_embed: true, use_rope: true`, configs/acoustic.yaml:66; `ffn_act: gelu`, configs/base.yaml:32).
 * Weights: the FastSpeech2Acoustic state_dict (`txt_embed.weight`, `dur_embed.*`, `encoder.layers.N.op.{layer_norm1,
 * self_attn.{in_proj,out_proj}.weight, self_attn.rotary_embed.freqs, layer_norm2, ffn.ffn_1, ffn.ffn_2}.*`,
 * `encoder.layer_norm.*`, `pitch_embed.*`, optional `lang_embed / spk_embed / variance_embeds.X / key_shift_embed /
 * speed_embed`) through dsd_load_weight / dsd_finalize_weights.
 */
#define DSD_EMBED_ENERGY 1u
#define DSD_EMBED_BREATHINESS 2u
#define DSD_EMBED_VOICING 4u
#define DSD_EMBED_TENSION 8u
#define DSD_EMBED_KEY_SHIFT 16u
#define DSD_EMBED_SPEED 32u

/* Positional information of a FastSpeech2Encoder (tts_modules.py:362-364,378-384,390-395) */
enum { DSD_POS_ROPE = 0,   /* use_pos_embed && use_rope: rotary embedding inside the attention (MultiheadSelfAttentionWithRoPE) */
       DSD_POS_REL = 1,    /* use_pos_embed && !use_rope && rel_pos: x * sqrt(H) + RelPositionalEncoding table
                              (espnet_positional_embedding.py:26-47,98-113), torch.nn.MultiheadAttention(bias=False);
                              weights `...self_attn.in_proj_weight` instead of `in_proj.weight` + `rotary_embed.freqs`, plus
                              `encoder.embed_positions.div_term` [H/2] = exp(arange(0, H, 2) * -(ln 10000 / H)) */
       DSD_POS_NONE = 2,   /* !use_pos_embed: no positions; attention and weight names as DSD_POS_REL */
       DSD_POS_SIN = 3 };  /* use_pos_embed && !use_rope && !rel_pos: x + SinusoidalPositionalEmbedding(positions)
                              (common_layers.py:44-99; positions count the non-padding tokens from 1, utils/__init__.py:118-128);
                              attention and weight names as DSD_POS_REL, plus `encoder.embed_positions.freqs` [H/2] =
                              exp(arange(H/2) * -(ln 10000 / (H/2 - 1))); the checkpoint's `encoder.embed_positions._float_tensor`
                              buffer is accepted and ignored */

/* TransformerFFNLayer's activation between ffn_1 and ffn_2 (common_layers.py:126-136): GELU (exact erf), ReLU, SiLU
   ('swish'), or SwiGLU - ffn_1 then has 2 * 4H output channels, `out * silu(gate)` with out = the first half
   (common_layers.py:107-117) */
enum { DSD_FFN_GELU = 0, DSD_FFN_RELU = 1, DSD_FFN_SWISH = 2, DSD_FFN_SWIGLU = 3 };

typedef struct dsd_encoder_config {
    int32_t struct_size;      /* sizeof(dsd_encoder_config)                                          */
    int32_t vocab_size;       /* FastSpeech2Acoustic(vocab_size)                                     */
    int32_t hidden_size;      /* hparams['hidden_size']                                              */
    int32_t enc_layers;       /* hparams['enc_layers']                                               */
    int32_t num_heads;        /* hparams['num_heads']                                                */
    int32_t ffn_kernel_size;  /* hparams['enc_ffn_kernel_size'] (odd)                                */
    int32_t num_spk;          /* hparams['num_spk'] if use_spk_id else 0                             */
    int32_t num_lang;         /* hparams['num_lang'] if use_lang_id else 0 (table has num_lang + 1 rows) */
    uint32_t embed_flags;     /* DSD_EMBED_*: use_energy_embed ... use_speed_embed                   */
    int32_t pos_mode;         /* DSD_POS_*                                                           */
    int32_t device;
    int32_t ffn_act;          /* DSD_FFN_*: hparams['ffn_act'] (TransformerFFNLayer, common_layers.py:120-151) */
} dsd_encoder_config;

/* Optional inputs of FastSpeech2Acoustic.forward (acoustic_encoder.py:82-88); NULL = not given.  All device pointers. */
typedef struct dsd_encode_extras {
    const int64_t* languages;      /* [B, T_txt]   (use_lang_id)                                     */
    const int64_t* spk_embed_id;   /* [B]          (use_spk_id, when spk_mix_embed is NULL)          */
    const float* spk_mix_embed;    /* element (b,t,h) at [b*bstride + t*tstride + h] (use_spk_id)    */
    int64_t spk_mix_bstride, spk_mix_tstride;
    const float* key_shift;        /* [B, T] each                                                    */
    const float* speed;
    const float* energy;
    const float* breathiness;
    const float* voicing;
    const float* tension;
} dsd_encode_extras;

int dsd_encoder_create(const dsd_encoder_config* cfg, dsd_handle** out);
/*
 * Replaces: FastSpeech2Acoustic.forward(txt_tokens, mel2ph, f0, key_shift, speed, spk_embed_id, languages, **kwargs)
 * (acoustic_encoder.py:82-118).  txt_tokens [B, T_txt] int64 (0 = padding), mel2ph [B, T] int64 (1-based token
 * index per frame, 0 = padding frame), f0 [B, T] Hz; cond_out [B, T, H] contiguous.
 */
int dsd_encode(dsd_handle* h, const int64_t* txt_tokens, const int64_t* mel2ph, const float* f0, int32_t B,
               int32_t T_txt, int32_t T, const dsd_encode_extras* extras, float* cond_out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Variance model (modules/toplevel.py:125-309, BASELINE config 5): the pieces between the tokens and the pitch /
 * multi-variance denoisers, which are ordinary dsd_create handles.
 *
 * A "token encoder" is the FastSpeech2Encoder (tts_modules.py:353-428, rotary configuration) on embeddings the caller
 * assembled (dsd_cond_assemble below), with the two heads the variance model hangs on it:
 *   - FastSpeech2Variance (variance_encoder.py:14-99): encoder + DurationPredictor (tts_modules.py:53-134)
 *   - MelodyEncoder (variance_encoder.py:102-148): encoder + out_proj Linear(hidden, out_dims)
 * Weights: `encoder.layers.N.op.*`, `encoder.layer_norm.*` as for dsd_encoder_create; `out_proj.{weight,bias}` when
 * out_dims > 0; `dur_predictor.conv.N.1.{weight,bias}` (Conv1d), `dur_predictor.conv.N.3.{weight,bias}` (LayerNorm over
 * channels, eps 1e-12) and `dur_predictor.linear.{weight,bias}` when dur_layers > 0.
 */
typedef struct dsd_token_encoder_config {
    int32_t struct_size;      /* sizeof(dsd_token_encoder_config)                                     */
    int32_t hidden_size;      /* hparams['hidden_size'] (melody encoder: melody_encoder_args.hidden_size) */
    int32_t enc_layers;
    int32_t num_heads;
    int32_t ffn_kernel_size;  /* odd                                                                  */
    int32_t out_dims;         /* MelodyEncoder.out_proj output size; 0 = no projection                */
    int32_t dur_layers;       /* dur_prediction_args.num_layers; 0 = no duration predictor            */
    int32_t dur_chans;        /* dur_prediction_args.hidden_size                                      */
    int32_t dur_kernel_size;  /* dur_prediction_args.kernel_size (odd)                                */
    float dur_offset;         /* dur_prediction_args.log_offset                                       */
    int32_t pos_mode;         /* DSD_POS_*                                                            */
    int32_t device;
    int32_t ffn_act;          /* DSD_FFN_*                                                            */
} dsd_token_encoder_config;

int dsd_token_encoder_create(const dsd_token_encoder_config* cfg, dsd_handle** out);
/*
 * Replaces: FastSpeech2Encoder.forward(main_embed, extra_embed, padding_mask) (tts_modules.py:400-428) [+ out_proj,
 * variance_encoder.py:147].  embed [B, L, H] = embed_scale * main_embed + extra_embed (tts_modules.py:387-389; no additive
 * positions in the rotary configuration; DSD_POS_REL adds them inside), padding_mask [B, L] bytes (non-zero = padding), enc_out [B, L, H or out_dims].
 */
int dsd_token_encode(dsd_handle* h, const float* embed, const uint8_t* padding_mask, int32_t B, int32_t L,
                     float* enc_out, void* stream);
/*
 * Replaces: DurationPredictor.forward(xs, x_masks, infer=True) (tts_modules.py:112-134): dur_cond [B, L, H] ->
 * dur_out [B, L] = clamp(exp(linear(...)) - offset, min 0), zero at padding.
 */
int dsd_predict_dur(dsd_handle* h, const float* dur_cond, const uint8_t* padding_mask, int32_t B, int32_t L,
                    float* dur_out, void* stream);

/*
 * Gather-and-add assembly of a [B, T, H] tensor: every embedding sum of the variance model
 * (variance_encoder.py:70-96,137-146; toplevel.py:233-236,246-276,289-298) is an instance of
 *   out[b,t,:] = sum_g scale_g * rowscale_g[b,t] * table_g[b*batch_stride_g + idx_g[b,t], :]
 *              + sum_k s_k[b,t] * v_k[:]
 * - an nn.Embedding lookup is a gather with batch_stride 0; `torch.gather(F.pad(x, [0,0,1,0]), 1, mel2ph)` is a gather
 *   from x with idx - 1 (`idx_offset` = -1; a negative row reads as zeros);
 * - Linear(1, H)(x) is the two terms (s = x, v = weight[:, 0]) and (s = 1, v = bias); s = NULL means 1.
 * Terms are added in the order given (gathers first).  No handle: nothing here has weights of its own.
 */
#define DSD_ASSEMBLE_MAX_GATHER 4
#define DSD_ASSEMBLE_MAX_TERMS 16
typedef struct dsd_assemble_args {
    int32_t struct_size;      /* sizeof(dsd_assemble_args) */
    int32_t device;
    int32_t B, T, H;
    int32_t n_gather, n_terms;
    struct {
        const float* table;       /* [rows, H] (batch_stride 0) or [B, rows, H]                     */
        int64_t batch_stride;     /* in elements                                                    */
        int64_t rows;             /* rows per batch item; an index outside [0, rows) reads as zeros */
        const int64_t* idx;       /* [B, T]                                                         */
        int64_t idx_offset;       /* added to every index                                           */
        float scale;
        const float* row_scale;   /* [B, T] or NULL                                                 */
    } gather[DSD_ASSEMBLE_MAX_GATHER];
    struct {
        const float* s;           /* [B, T] or NULL (= 1)                                           */
        const float* v;           /* [H]                                                            */
    } term[DSD_ASSEMBLE_MAX_TERMS];
} dsd_assemble_args;

int dsd_cond_assemble(const dsd_assemble_args* args, float* out, void* stream);

/*
 * NSF-HiFiGAN generator (the step after the loop: mel -> waveform).  The constructor arguments are the fields of the
 * checkpoint's config.json that Generator.__init__ reads (modules/nsf_hifigan/models.py:207-260), `mini_nsf: false`.
 * Weights: the Generator state_dict in its inference form, i.e. after remove_weight_norm() (models.py:292-302):
 * `m_source.l_linear.*`, `noise_convs.N.*`, `conv_pre.*`, `ups.N.*` ([C_in, C_out, K] as ConvTranspose1d stores it),
 * `resblocks.N.convs1.M.* / convs2.M.*` (ResBlock1) or `resblocks.N.convs.M.*` (ResBlock2), `conv_post.*`;
 * with mini_nsf: `source_conv.*` instead of `m_source.*` / `noise_convs.*`.
 */
#define DSD_VOC_MAX_UPS 8
#define DSD_VOC_MAX_KERNELS 8
#define DSD_VOC_MAX_DILS 4
typedef struct dsd_vocoder_config {
    int32_t struct_size;
    int32_t num_mels;
    int32_t sampling_rate;
    int32_t upsample_initial_channel;
    int32_t n_ups;
    int32_t upsample_rates[DSD_VOC_MAX_UPS];
    int32_t upsample_kernel_sizes[DSD_VOC_MAX_UPS];
    int32_t resblock;                                  /* 1 = ResBlock1, 2 = ResBlock2 */
    int32_t n_kernels;
    int32_t resblock_kernel_sizes[DSD_VOC_MAX_KERNELS];
    int32_t n_dilations[DSD_VOC_MAX_KERNELS];
    int32_t resblock_dilation_sizes[DSD_VOC_MAX_KERNELS][DSD_VOC_MAX_DILS];
    int32_t harmonic_num;                              /* SourceModuleHnNSF(harmonic_num=8), models.py:221-224 */
    int32_t mini_nsf;                                  /* h.mini_nsf (models.py:212-225): 1 = fastsinegen source, added
                                                          once through `source_conv` after the second upsampling */
    float noise_sigma;                                 /* h.noise_sigma (models.py:213,272-273): > 0 adds
                                                          noise_sigma * pre_noise after conv_pre; 0 = off */
    int32_t device;
} dsd_vocoder_config;

int dsd_vocoder_create(const dsd_vocoder_config* cfg, dsd_handle** out);
/*
 * Replaces: Generator.forward(x, f0)  (models.py:262-290) with the two random draws of SineGen made explicit:
 *   mel      element (b, m, t) at mel[b*stride_b + m*stride_m + t*stride_t]: natural-log mel, [B, num_mels, T] view
 *            (the wrapper's 2.30259 * log10-mel, vocoders/nsf_hifigan.py:59-64, is the caller's)
 *   f0       [B, T] Hz, 0 = unvoiced
 *   rand_ini [harmonic_num + 1] uniform [0,1) initial phases (torch.rand, models.py:145; element 0 is ignored)
 *   noise    [B, T * prod(upsample_rates), harmonic_num + 1] standard normals (torch.randn_like, models.py:165)
 *            (both may be NULL for a mini_nsf generator: its source is deterministic)
 *   pre_noise [B, upsample_initial_channel, T] standard normals (torch.randn_like(x), models.py:273); required when the
 *            configuration's noise_sigma > 0, ignored (may be NULL) otherwise
 *   wav_out  [B, T * prod(upsample_rates)]
 */
int dsd_vocode(dsd_handle* h, const float* mel, int32_t B, int32_t T, int64_t stride_b, int64_t stride_m,
               int64_t stride_t, const float* f0, const float* rand_ini, const float* noise, const float* pre_noise,
               float* wav_out, void* stream);

/*
 * Ragged batches.  The reference runs one utterance per call (inference/ds_acoustic.py:214-271), because padding a batch
 * changes results near the end of the shorter items: frames beyond an item's end are not zero after the first layer and
 * leak into valid frames through every convolution along time.  With per-item lengths the library treats frames
 * t >= lengths[b] of item b as the convolutions' zero padding - in the dilated convolutions of the WaveNet, the depthwise
 * convolution of LYNXNet and the ConvNeXt aux decoder - so item b of a padded batch comes out as if it had been run alone
 * at T = lengths[b] (outputs at its padded frames are unspecified), and several segments of a project can share one
 * launch.  lengths: HOST array of B values (copied, stream-ordered); applies to the following dsd_prepare_cond / dsd_denoise
 * / dsd_sample / dsd_aux_decode calls at batch size B until changed; NULL restores dense batches.
 */
int dsd_set_lengths(dsd_handle* h, const int32_t* lengths, int32_t B, void* stream);

/* ------------------------------------------------------------------------------------------
 * Sampling programs.  Every sampler of the reference (ddpm.py:149-204,221-351 p_sample /
 * p_sample_ddim / p_sample_plms; dpm_solver_pytorch.py:1171-1213; uni_pc.py:590-672;
 * reflow.py:66-138 euler/rk2/rk4/rk5) is a sequence of backbone evaluations whose results enter
 * the solver state only through linear combinations with scalar coefficients that are known
 * before the loop starts.  A program states exactly that: per evaluation, the state buffer fed
 * to the backbone, the model time, and up to DSD_MAX_OUT linear combinations
 *      dst = sum_k coef_k * src_k,     src_k in { model output, state buffers, injected noise }
 * which the library fuses into the epilogue of the backbone's last GEMM.  The host-side
 * scheduler (diffsinger_amd/schedule.py) computes the coefficients; the device does NFE + axpy.
 * ------------------------------------------------------------------------------------------ */
#define DSD_MAX_TERMS 8
#define DSD_MAX_OUT 3
#define DSD_SRC_MODEL (-1)            /* the backbone output of this evaluation            */
#define DSD_SRC_NOISE_BASE (-1000)    /* src = DSD_SRC_NOISE_BASE - k : k-th injected noise */

typedef struct dsd_term {
    int32_t src;   /* state buffer id >= 0, DSD_SRC_MODEL, or DSD_SRC_NOISE_BASE - k */
    float coef;
} dsd_term;

typedef struct dsd_lincomb {
    int32_t dst;      /* state buffer id */
    int32_t n_terms;  /* 1..DSD_MAX_TERMS */
    dsd_term terms[DSD_MAX_TERMS];
} dsd_lincomb;

typedef struct dsd_eval {
    int32_t x_buf;    /* state buffer holding the backbone input x_t */
    float t;          /* model time fed to SinusoidalPosEmb (same for the whole batch) */
    int32_t n_out;    /* 1..DSD_MAX_OUT */
    dsd_lincomb out[DSD_MAX_OUT];
} dsd_eval;

typedef struct dsd_program {
    int32_t n_bufs;          /* number of state buffers (each [B, F*M, T]); buffer 0 = x_T on entry */
    int32_t result_buf;      /* buffer holding the sample after the last evaluation */
    int32_t n_evals;
    int32_t n_noise;         /* number of injected [B,F,M,T] noise tensors referenced by terms */
    const dsd_eval* evals;
} dsd_program;

#define DSD_SAMPLE_GRAPH 1u      /* replay the whole loop from a cached hipGraph (captured at first use) */
#define DSD_SAMPLE_GRAPH_LAZY 4u /* with DSD_SAMPLE_GRAPH: run a (program, batch shape) eagerly the first time and capture
                                    its graph when it comes back - capture costs about one loop, and the segments of a
                                    project all differ in length (8 segments of 480-1000 frames: 250 ms capturing each,
                                    136 ms lazily) */
#define DSD_SAMPLE_TRANSPOSE 2u  /* out is [B,T,M] (F == 1) or [B,F,T,M] and
                                    out = sample * out_scale[f*M+m] + out_shift[f*M+m]
                                    (x.transpose(2,3).squeeze(1) + denorm_spec, ddpm.py:350,382-383) */

/*
 * Replaces: the loop of GaussianDiffusion.inference (ddpm.py:244-349) / RectifiedFlow.inference
 * (reflow.py:132-136) after x has been initialised, for the cond of the last dsd_prepare_cond.
 *   x_init    [B,F,M,T] device: initial state (x_T, or the shallow-diffusion start)
 *   noise     n_noise x [B,F,M,T] device, or NULL when n_noise == 0 (ancestral DDPM, ddpm.py:153)
 *   out       [B,F,M,T], or the transposed/denormalised form with DSD_SAMPLE_TRANSPOSE
 *   out_scale/out_shift: device [F*M] (only with DSD_SAMPLE_TRANSPOSE; NULL = identity)
 */
int dsd_sample(dsd_handle* h, const dsd_program* prog, const float* x_init, const float* noise,
               float* out, const float* out_scale, const float* out_shift, uint32_t flags, void* stream);

/*
 * Arithmetic of the residual layers' two GEMMs (API v10).  DSD_PRECISION_F32 (default): fp32 operands on
 * v_mfma_f32_16x16x4_f32 - the reference's arithmetic, what every BASELINE number is measured in.  DSD_PRECISION_BF16X3
 * (opt-in; SURVEY.md section 7 "hard parts"): every operand split x = hi + lo into two bf16 values, a product evaluated as
 * hi.hi + hi.lo + lo.hi on v_mfma_f32_16x16x32_bf16 with fp32 accumulation; activations, FiLM, gate, residual and skip
 * arithmetic stay fp32.  Measured against the fp32 oracle: 9.9e-6 on one evaluation, 1.9e-6 on the 50-NFE DPM-Solver++
 * sample (tools/bf16x3_tolerance.py; asserted on the GPU in tests/test_gpu_bf16x3.py at the fp32 tolerances).  Exists for the
 * fused WaveNet layer kernel at C = 256 (batched grids) and for LYNXNet's two pointwise GEMMs at C = 1024 / 512 with
 * expansion 2; other shapes and kernels run fp32 whatever the mode.  Also set for
 * every handle of the process by the environment variable DSD_PRECISION=1 at dsd_create.  May be called at any time; after
 * dsd_finalize_weights it re-packs the weights.
 */
#define DSD_PRECISION_F32 0
#define DSD_PRECISION_BF16X3 1
int dsd_set_precision(dsd_handle* h, int32_t mode);

/* Introspection used by tests, bench.py and the roofline report. */
typedef struct dsd_stats {
    int64_t weight_bytes;        /* packed weights on the device                        */
    int64_t workspace_bytes;     /* current arena size                                  */
    int64_t flops_per_frame_nfe; /* algorithmic FLOPs per mel frame per NFE (hoisted)   */
    int64_t bytes_per_frame_nfe; /* algorithmic HBM bytes per mel frame per NFE         */
    int32_t kernels_per_nfe;     /* kernel launches per backbone evaluation             */
    int32_t graphs_cached;
    /* WaveNet, how a residual layer runs on the current batch shape (API v10): launches per layer (1 = the fused layer
       kernel over every tile, 2 = the row-split pair or the two GEMMs, 3 = a mixed plan: the whole rounds of tiles on the
       fused kernel, the remainder on the row-split pair) and how many 32-frame tiles each form covers */
    int32_t layer_launches;
    int32_t fused_tiles;
    int32_t split_tiles;
    int32_t precision;           /* DSD_PRECISION_* of the fused segments that ran (dsd_set_precision) */
} dsd_stats;
int dsd_get_stats(const dsd_handle* h, dsd_stats* out);

/*
 * Timing hook for bench.py: while enabled, every launch of the dominant kernel (WaveNet: dilated-conv +
 * FiLM + gate GEMM; LYNXNet: the LayerNorm -> C->4C -> SwiGLU GEMM) carries a hipEvent start/stop pair
 * attached to the dispatch itself (hipExtLaunchKernelGGL) on the stream the kernel is launched on; graph
 * replay is bypassed while enabled.  dsd_kernel_timing_read returns the mean kernel duration in milliseconds
 * over the launches recorded since the last reset and their count, plus - for reference - the mean time of
 * an EMPTY hipEventRecord pair on that stream (what a plain event bracket would have added).
 */
int dsd_kernel_timing(dsd_handle* h, int32_t enable);
int dsd_kernel_timing_read(dsd_handle* h, double* mean_ms, double* empty_pair_ms, int64_t* launches);

/*
 * The same pass per kernel CLASS (API v10): the layer kernels of an evaluation are several instantiations (tile halo by
 * dilation, the segments of a mixed plan, the two networks of the variance model each on its own handle), and the roofline
 * report weights them by time.  Classes are returned largest share first; reading does not reset (dsd_kernel_timing does).
 *   name              kernel + template arguments as rocprofv3 prints them, e.g. "wn_layer_kernel<4, 48, 0>"
 *   mean_ms           mean begin -> end time of the launches that carried events (every 7th of the class)
 *   launches          all launches of the class over the pass, evaluations the backbone evaluations of the pass
 *   flops_per_launch  algorithmic FLOPs of one launch over its VALID frames, bytes_per_launch likewise (8(d))
 */
typedef struct dsd_kernel_time {
    char name[96];
    double mean_ms;
    int64_t launches_timed;
    int64_t launches;
    int64_t evaluations;
    double flops_per_launch;
    double bytes_per_launch;
} dsd_kernel_time;
int dsd_kernel_timing_classes(dsd_handle* h, dsd_kernel_time* out, int32_t max_classes, int32_t* n_classes,
                              double* empty_pair_ms);

#ifdef __cplusplus
}
#endif
#endif /* DSDENOISE_H_ */
