"""Deterministic synthetic weights and inputs (numpy only).

There are no checkpoints in the build container or on the GPU box, so every
test, fixture and benchmark regenerates its weights from a seed.  The state
dict names and shapes are exactly those of the reference modules
(`modules/backbones/wavenet.py:22-31,56-72`, `modules/backbones/lynxnet.py:52-62,
71-74,104-124`), so the same dict loads with `strict=True` into the reference
`WaveNet` / `LYNXNet`, into this package's shims and into the numpy oracle.

`output_projection.weight` is zero-initialised by the reference
(`wavenet.py:73`, `lynxnet.py:126`); here it gets fan-in scaled normals like
every other layer, otherwise the network output would be its bias only.
"""
from __future__ import annotations

import hashlib
from collections import OrderedDict

import numpy as np


def wavenet_param_shapes(in_dims, n_feats, num_layers=20, num_channels=256, hidden_size=256):
    c, m, h = num_channels, in_dims * n_feats, hidden_size
    shapes = OrderedDict()
    shapes["input_projection.weight"] = (c, m, 1)
    shapes["input_projection.bias"] = (c,)
    shapes["mlp.0.weight"] = (4 * c, c)
    shapes["mlp.0.bias"] = (4 * c,)
    shapes["mlp.2.weight"] = (c, 4 * c)
    shapes["mlp.2.bias"] = (c,)
    for l in range(num_layers):
        p = f"residual_layers.{l}."
        shapes[p + "dilated_conv.weight"] = (2 * c, c, 3)
        shapes[p + "dilated_conv.bias"] = (2 * c,)
        shapes[p + "diffusion_projection.weight"] = (c, c)
        shapes[p + "diffusion_projection.bias"] = (c,)
        shapes[p + "conditioner_projection.weight"] = (2 * c, h, 1)
        shapes[p + "conditioner_projection.bias"] = (2 * c,)
        shapes[p + "output_projection.weight"] = (2 * c, c, 1)
        shapes[p + "output_projection.bias"] = (2 * c,)
    shapes["skip_projection.weight"] = (c, c, 1)
    shapes["skip_projection.bias"] = (c,)
    shapes["output_projection.weight"] = (m, c, 1)
    shapes["output_projection.bias"] = (m,)
    return shapes


def lynxnet_param_shapes(in_dims, n_feats, num_layers=6, num_channels=512, expansion_factor=2,
                         kernel_size=31, activation="PReLU", hidden_size=256):
    c, m, h = num_channels, in_dims * n_feats, hidden_size
    inner = c * expansion_factor
    shapes = OrderedDict()
    shapes["input_projection.weight"] = (c, m, 1)
    shapes["input_projection.bias"] = (c,)
    shapes["diffusion_embedding.1.weight"] = (4 * c, c)
    shapes["diffusion_embedding.1.bias"] = (4 * c,)
    shapes["diffusion_embedding.3.weight"] = (c, 4 * c)
    shapes["diffusion_embedding.3.bias"] = (c,)
    for l in range(num_layers):
        p = f"residual_layers.{l}."
        shapes[p + "diffusion_projection.weight"] = (c, c, 1)
        shapes[p + "diffusion_projection.bias"] = (c,)
        shapes[p + "conditioner_projection.weight"] = (c, h, 1)
        shapes[p + "conditioner_projection.bias"] = (c,)
        shapes[p + "convmodule.net.0.weight"] = (c,)
        shapes[p + "convmodule.net.0.bias"] = (c,)
        shapes[p + "convmodule.net.2.weight"] = (2 * inner, c, 1)
        shapes[p + "convmodule.net.2.bias"] = (2 * inner,)
        shapes[p + "convmodule.net.4.weight"] = (inner, 1, kernel_size)
        shapes[p + "convmodule.net.4.bias"] = (inner,)
        if activation == "PReLU":
            shapes[p + "convmodule.net.5.weight"] = (inner,)
        shapes[p + "convmodule.net.6.weight"] = (c, inner, 1)
        shapes[p + "convmodule.net.6.bias"] = (c,)
    shapes["norm.weight"] = (c,)
    shapes["norm.bias"] = (c,)
    shapes["output_projection.weight"] = (m, c, 1)
    shapes["output_projection.bias"] = (m,)
    return shapes


def convnext_param_shapes(in_dims, out_dims, num_channels=512, num_layers=6, kernel_size=7, prefix=""):
    """state_dict of modules/aux_decoder/convnext.py:58-76 (ConvNeXtDecoder); `prefix="decoder."` gives the
    AuxDecoderAdaptor layout (aux_decoder/__init__.py:33-37)."""
    c = num_channels
    shapes = OrderedDict()
    shapes[prefix + "inconv.weight"] = (c, in_dims, kernel_size)
    shapes[prefix + "inconv.bias"] = (c,)
    for l in range(num_layers):
        p = f"{prefix}conv.{l}."
        shapes[p + "gamma"] = (c,)
        shapes[p + "dwconv.weight"] = (c, 1, 7)
        shapes[p + "dwconv.bias"] = (c,)
        shapes[p + "norm.weight"] = (c,)
        shapes[p + "norm.bias"] = (c,)
        shapes[p + "pwconv1.weight"] = (4 * c, c)
        shapes[p + "pwconv1.bias"] = (4 * c,)
        shapes[p + "pwconv2.weight"] = (c, 4 * c)
        shapes[p + "pwconv2.bias"] = (c,)
    shapes[prefix + "outconv.weight"] = (out_dims, c, kernel_size)
    shapes[prefix + "outconv.bias"] = (out_dims,)
    return shapes


def fs2_acoustic_param_shapes(vocab_size, hidden_size=256, enc_layers=4, num_heads=2, ffn_kernel_size=3,
                              num_spk=0, num_lang=0, variances=(), key_shift=False, speed=False, rope=True, sinpos=False,
                              ffn_act="gelu"):
    """state_dict of modules/fastspeech/acoustic_encoder.py:14-63 (FastSpeech2Acoustic) in its rotary-embedding
    configuration (`use_rope: true`): tts_modules.py:353-383, common_layers.py:120-234.  The rotary frequency
    table is an (untrained) nn.Parameter of the shared RotaryEmbedding and shows up once per layer.  `rope=False`:
    the pre-rotary layout, torch.nn.MultiheadAttention(bias=False) (common_layers.py:222-226)."""
    h = hidden_size
    shapes = OrderedDict()
    shapes["txt_embed.weight"] = (vocab_size, h)
    if num_lang:
        shapes["lang_embed.weight"] = (num_lang + 1, h)
    shapes["dur_embed.weight"] = (h, 1)
    shapes["dur_embed.bias"] = (h,)
    for l in range(enc_layers):
        p = f"encoder.layers.{l}.op."
        shapes[p + "layer_norm1.weight"] = (h,)
        shapes[p + "layer_norm1.bias"] = (h,)
        if rope:
            shapes[p + "self_attn.in_proj.weight"] = (3 * h, h)
            shapes[p + "self_attn.out_proj.weight"] = (h, h)
            shapes[p + "self_attn.rotary_embed.freqs"] = (h // num_heads // 2,)
        else:
            shapes[p + "self_attn.in_proj_weight"] = (3 * h, h)
            shapes[p + "self_attn.out_proj.weight"] = (h, h)
        shapes[p + "layer_norm2.weight"] = (h,)
        shapes[p + "layer_norm2.bias"] = (h,)
        f1 = (8 if ffn_act == "swiglu" else 4) * h          # SwiGLU: out and gate halves (common_layers.py:132-134)
        shapes[p + "ffn.ffn_1.weight"] = (f1, h, ffn_kernel_size)
        shapes[p + "ffn.ffn_1.bias"] = (f1,)
        shapes[p + "ffn.ffn_2.weight"] = (h, 4 * h)
        shapes[p + "ffn.ffn_2.bias"] = (h,)
    shapes["encoder.layer_norm.weight"] = (h,)
    shapes["encoder.layer_norm.bias"] = (h,)
    if sinpos:       # SinusoidalPositionalEmbedding's marker buffer (common_layers.py:59): a value nobody reads
        shapes["encoder.embed_positions._float_tensor"] = (1,)
    shapes["pitch_embed.weight"] = (h, 1)
    shapes["pitch_embed.bias"] = (h,)
    for v in variances:
        shapes[f"variance_embeds.{v}.weight"] = (h, 1)
        shapes[f"variance_embeds.{v}.bias"] = (h,)
    if key_shift:
        shapes["key_shift_embed.weight"] = (h, 1)
        shapes["key_shift_embed.bias"] = (h,)
    if speed:
        shapes["speed_embed.weight"] = (h, 1)
        shapes["speed_embed.bias"] = (h,)
    if num_spk:
        shapes["spk_embed.weight"] = (num_spk, h)
    return shapes


NSF_HIFIGAN_DEFAULT = dict(       # the openvpi 44.1 kHz / hop 512 NSF-HiFiGAN layout (config.json ships with the checkpoint)
    sampling_rate=44100, num_mels=128, hop_size=512, upsample_rates=[8, 8, 2, 2, 2],
    upsample_kernel_sizes=[16, 16, 4, 4, 4], upsample_initial_channel=512, resblock="1",
    resblock_kernel_sizes=[3, 7, 11], resblock_dilation_sizes=[[1, 3, 5], [1, 3, 5], [1, 3, 5]], mini_nsf=False,
    noise_sigma=0.0)


def nsf_hifigan_param_shapes(h):
    """state_dict of modules/nsf_hifigan/models.py:207-260 (Generator, mini_nsf = False) in its inference form,
    i.e. after `remove_weight_norm()`: plain `weight` / `bias` everywhere."""
    shapes = OrderedDict()
    mini = bool(h.get("mini_nsf", False))
    if not mini:
        shapes["m_source.l_linear.weight"] = (1, 9)
        shapes["m_source.l_linear.bias"] = (1,)
    ch = h["upsample_initial_channel"]
    rates = list(h["upsample_rates"])
    noise = OrderedDict()
    ups = OrderedDict()
    res = OrderedDict()
    nk = len(h["resblock_kernel_sizes"])
    for i, (u, k) in enumerate(zip(rates, h["upsample_kernel_sizes"])):
        ch //= 2
        ups[f"ups.{i}.weight"] = (2 * ch, ch, k)
        ups[f"ups.{i}.bias"] = (ch,)
        for j, (rk, rd) in enumerate(zip(h["resblock_kernel_sizes"], h["resblock_dilation_sizes"])):
            pre = f"resblocks.{i * nk + j}."
            if str(h.get("resblock", "1")) == "1":
                for grp in ("convs1", "convs2"):
                    for d in range(len(rd)):
                        res[f"{pre}{grp}.{d}.weight"] = (ch, ch, rk)
                        res[f"{pre}{grp}.{d}.bias"] = (ch,)
            else:
                for d in range(len(rd)):
                    res[f"{pre}convs.{d}.weight"] = (ch, ch, rk)
                    res[f"{pre}convs.{d}.bias"] = (ch,)
        if mini:
            if i == 1:
                noise["source_conv.weight"] = (ch, 1, 1)
                noise["source_conv.bias"] = (ch,)
        else:
            if i + 1 < len(rates):
                sf = int(np.prod(rates[i + 1:]))
                noise[f"noise_convs.{i}.weight"] = (ch, 1, sf * 2)
            else:
                noise[f"noise_convs.{i}.weight"] = (ch, 1, 1)
            noise[f"noise_convs.{i}.bias"] = (ch,)
    shapes.update(noise)
    shapes["conv_pre.weight"] = (h["upsample_initial_channel"], h["num_mels"], 7)
    shapes["conv_pre.bias"] = (h["upsample_initial_channel"],)
    shapes.update(ups)
    shapes.update(res)
    shapes["conv_post.weight"] = (1, ch, 7)
    shapes["conv_post.bias"] = (1,)
    return shapes


def backbone_param_shapes(kind, in_dims, n_feats, hidden_size=256, **args):
    if kind == "wavenet":
        return wavenet_param_shapes(in_dims, n_feats, num_layers=args.get("num_layers", 20),
                                    num_channels=args.get("num_channels", 256), hidden_size=hidden_size)
    if kind == "lynxnet":
        return lynxnet_param_shapes(in_dims, n_feats, num_layers=args.get("num_layers", 6),
                                    num_channels=args.get("num_channels", 512),
                                    expansion_factor=args.get("expansion_factor", 2),
                                    kernel_size=args.get("kernel_size", 31),
                                    activation=args.get("activation", "PReLU") or "PReLU",
                                    hidden_size=hidden_size)
    raise KeyError(kind)


def synth_state_dict(shapes, seed=42, gain=1.0):
    """name -> float32 ndarray.  Fan-in scaled normals for matrices (times `gain`: < 1 keeps deep residual stacks
    such as the vocoder's in the O(1) range), N(0, 0.1) biases, LayerNorm gains around 1, PReLU slopes around 0.25."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = OrderedDict()
    for name, shape in shapes.items():
        z = rng.standard_normal(shape, dtype=np.float32)
        leaf = name.rsplit(".", 1)[-1]
        is_ln = (name.startswith("norm.") or ".convmodule.net.0." in name or ".norm." in name
                 or ".layer_norm" in name)
        if is_ln and leaf == "weight":
            w = 1.0 + 0.1 * z
        elif ".convmodule.net.5." in name:
            w = 0.25 + 0.05 * z
        elif leaf == "freqs":                      # rotary table: the reference's own values (rotary_embedding_torch.py:119)
            d = 2 * shape[0]
            w = (1.0 / (10000.0 ** (np.arange(0, d, 2, dtype=np.float32) / np.float32(d)))).astype(np.float32)
        elif name.endswith("_embed.weight") and len(shape) == 2 and shape[1] > 1 and "variance" not in name:
            w = z / np.sqrt(np.float32(shape[1]))      # embedding tables: N(0, H^-0.5) (common_layers.py:24)
            if name.startswith(("txt_embed", "lang_embed")):
                w[0] = 0.0                             # padding_idx row
        elif leaf == "gamma":                      # ConvNeXt layer scale (reference init 1e-6; O(1) here so it matters)
            w = 0.5 + 0.1 * z
        elif leaf == "bias":
            w = 0.1 * z
        else:
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
            w = np.float32(gain) * z / np.sqrt(np.float32(fan_in))
        out[name] = np.ascontiguousarray(w, dtype=np.float32)
    return out


def state_dict_digest(sd):
    h = hashlib.sha256()
    for name, arr in sd.items():
        h.update(name.encode())
        h.update(np.ascontiguousarray(arr, dtype=np.float32).tobytes())
    return h.hexdigest()


def synth_normal(shape, seed):
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.standard_normal(shape, dtype=np.float32)
