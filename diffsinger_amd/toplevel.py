"""`DiffSingerAcoustic` (modules/toplevel.py:32-120) on libdsdenoise.

`AcousticDecoder` is everything after the FastSpeech2 encoder has produced `condition` (aux decoder, padding
masks, denoise loop); `DiffSingerAcoustic` adds the encoder (`fs2`) in front.  Same attribute names as the
reference (`fs2`, `aux_decoder`, `diffusion`), so an acoustic checkpoint loads with strict=True; same hparams read
at construction (toplevel.py:44-83) and the same infer-branch behaviour (:84-105).  Inference only.
"""
from __future__ import annotations

from dataclasses import dataclass

import torch
import torch.nn as nn

from .aux_decoder import AuxDecoderAdaptor
from .encoder import FastSpeech2Acoustic
from .diffusion import GaussianDiffusion, RectifiedFlow
from .hparams import hparams


@dataclass
class ShallowDiffusionOutput:          # toplevel.py:26-29
    aux_out: torch.Tensor = None
    diff_out: torch.Tensor = None


def get_backbone_type(hp):             # modules/compat.py: older configs name the backbone differently
    if 'backbone_type' in hp:
        return hp['backbone_type']
    if 'diff_decoder_type' in hp:
        return hp['diff_decoder_type']
    return 'wavenet'


def get_backbone_args(hp, backbone_type):
    args = hp.get('backbone_args')
    if args is not None:
        return args
    if backbone_type == 'wavenet':
        return {'num_layers': hp.get('residual_layers'), 'num_channels': hp.get('residual_channels'),
                'dilation_cycle_length': hp.get('dilation_cycle_length')}
    return None


class AcousticDecoder(nn.Module):
    """`DiffSingerAcoustic` minus `fs2`: aux decoder (shallow diffusion) + denoise loop."""

    def __init__(self, out_dims):
        super().__init__()
        self._init_decoder(out_dims)

    def _init_decoder(self, out_dims):
        self.use_shallow_diffusion = hparams.get('use_shallow_diffusion', False)
        self.shallow_args = hparams.get('shallow_diffusion_args', {})
        if self.use_shallow_diffusion:
            self.aux_decoder = AuxDecoderAdaptor(
                in_dims=hparams['hidden_size'], out_dims=out_dims, num_feats=1,
                spec_min=hparams['spec_min'], spec_max=hparams['spec_max'],
                aux_decoder_arch=self.shallow_args['aux_decoder_arch'],
                aux_decoder_args=self.shallow_args['aux_decoder_args'])
        self.diffusion_type = hparams.get('diffusion_type', 'ddpm')
        self.backbone_type = get_backbone_type(hparams)
        self.backbone_args = get_backbone_args(hparams, self.backbone_type)
        if self.diffusion_type == 'ddpm':
            self.diffusion = GaussianDiffusion(
                out_dims=out_dims, num_feats=1, timesteps=hparams['timesteps'], k_step=hparams['K_step'],
                backbone_type=self.backbone_type, backbone_args=self.backbone_args,
                spec_min=hparams['spec_min'], spec_max=hparams['spec_max'])
        elif self.diffusion_type == 'reflow':
            self.diffusion = RectifiedFlow(
                out_dims=out_dims, num_feats=1, t_start=hparams['T_start'],
                time_scale_factor=hparams['time_scale_factor'],
                backbone_type=self.backbone_type, backbone_args=self.backbone_args,
                spec_min=hparams['spec_min'], spec_max=hparams['spec_max'])
        else:
            raise NotImplementedError(self.diffusion_type)

    def forward(self, condition, mel2ph, gt_mel=None, infer=True, lengths=None, **diffusion_kwargs) -> ShallowDiffusionOutput:
        """toplevel.py:90-105 with `condition = self.fs2(...)` already evaluated.  `lengths` [B] (ragged batch): every
        utterance comes out as if it had been run alone at its own length (dsd_set_lengths)."""
        if not infer:
            raise NotImplementedError("training (toplevel.py:106-120) stays on the reference modules")
        mask = (mel2ph > 0).float()[:, :, None]
        if lengths is not None:
            diffusion_kwargs["lengths"] = lengths
        if self.use_shallow_diffusion:
            aux_mel_pred = self.aux_decoder(condition, infer=True, lengths=lengths)
            aux_mel_pred *= mask
            if gt_mel is not None and self.shallow_args['val_gt_start']:
                src_mel = gt_mel
            else:
                src_mel = aux_mel_pred
        else:
            aux_mel_pred = src_mel = None
        mel_pred = self.diffusion(condition, src_spec=src_mel, infer=True, **diffusion_kwargs)
        mel_pred *= mask
        return ShallowDiffusionOutput(aux_out=aux_mel_pred, diff_out=mel_pred)


class DiffSingerAcoustic(AcousticDecoder):
    """toplevel.py:32-120: `fs2` encoder -> (aux decoder ->) denoise loop, tokens in, mel out."""

    def __init__(self, vocab_size, out_dims):
        nn.Module.__init__(self)
        self.fs2 = FastSpeech2Acoustic(vocab_size=vocab_size)       # registered first, as in the reference
        self._init_decoder(out_dims)

    def forward(self, txt_tokens, mel2ph, f0, key_shift=None, speed=None, spk_embed_id=None, languages=None,
                gt_mel=None, infer=True, noise=None, step_noise=None, lengths=None, **kwargs) -> ShallowDiffusionOutput:
        condition = self.fs2(txt_tokens, mel2ph, f0, key_shift=key_shift, speed=speed, spk_embed_id=spk_embed_id,
                             languages=languages, **kwargs)
        extra = {}
        if noise is not None:
            extra["noise"] = noise
        if step_noise is not None:
            extra["step_noise"] = step_noise
        return AcousticDecoder.forward(self, condition, mel2ph, gt_mel=gt_mel, infer=infer, lengths=lengths, **extra)
