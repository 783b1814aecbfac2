#!/usr/bin/env python3
"""`.ds` project -> waveform with a trained DiffSinger acoustic experiment and an NSF-HiFiGAN vocoder, on the HIP
library only (no DiffSinger checkout needed at run time).

    python examples/ds_to_wav.py checkpoints/my_exp song.ds nsf_hifigan/model.ckpt -o song.wav [--steps 20] [--depth 0.6]
                                 [--batch-size 8] [--seed 42]

checkpoints/my_exp holds what a training run leaves there: config.yaml, model_ckpt_steps_<N>.ckpt, dictionary-<lang>.txt
(or dictionary.txt), and - for multi-speaker / multilingual models - spk_map.json / lang_map.json.
"""
import argparse
import json
import pathlib

from diffsinger_amd import harness
from diffsinger_amd.hparams import hparams, load_config
from diffsinger_amd.toplevel import DiffSingerAcoustic


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("exp", type=pathlib.Path, help="experiment (work) directory")
    ap.add_argument("proj", type=pathlib.Path, help=".ds project")
    ap.add_argument("vocoder", type=pathlib.Path, help="NSF-HiFiGAN generator checkpoint (config.json beside it)")
    ap.add_argument("-o", "--out", type=pathlib.Path, default=None)
    ap.add_argument("--ckpt", type=int, default=None, help="checkpoint step (default: the latest)")
    ap.add_argument("--steps", type=int, default=None, help="sampling steps (scripts/infer.py --steps)")
    ap.add_argument("--depth", type=float, default=None, help="shallow-diffusion depth in (0, 1] (scripts/infer.py --depth)")
    ap.add_argument("--batch-size", type=int, default=8, help="segments per launch of the acoustic model (ragged batch)")
    ap.add_argument("--seed", type=int, default=-1)
    ap.add_argument("--key", type=int, default=0, help="transpose by this many semitones (scripts/infer.py --key)")
    ap.add_argument("--spk", default=None, help='speaker or mix, e.g. "alice" or "alice:0.3|bob" (scripts/infer.py --spk)')
    args = ap.parse_args()

    load_config(args.exp / "config.yaml", overrides=dict(infer=True, work_dir=str(args.exp)))
    harness.apply_depth_steps(hparams, depth=args.depth, steps=args.steps)
    dictionary = harness.load_phoneme_dictionary()
    maps = {}
    for name in ("spk_map", "lang_map"):
        path = args.exp / f"{name}.json"
        maps[name] = json.loads(path.read_text(encoding="utf8")) if path.exists() else {}
    model = DiffSingerAcoustic(len(dictionary), hparams["audio_num_mel_bins"]).cuda().eval()
    ckpt = harness.load_ckpt(model, args.exp, ckpt_steps=args.ckpt, prefix_in_ckpt="model", strict=True)
    print(f"| acoustic model: {ckpt}")
    vocoder = harness.load_vocoder(args.vocoder)
    h = harness.AcousticHarness(model, vocoder, dictionary, spk_map=maps["spk_map"], lang_map=maps["lang_map"], device="cuda")
    out = args.out or args.proj.with_suffix(".wav")
    params = harness.load_ds(args.proj)
    if args.key:
        params = harness.trans_key(params, args.key)
    if args.spk is not None:
        mix = harness.parse_commandline_spk_mix(args.spk)
        for seg in params:
            seg["spk_mix"] = mix
    track = h.run_inference(params, out_path=out, seed=args.seed, batch_size=args.batch_size)
    print(f"| wrote {out}: {track.shape[0] / hparams['audio_sample_rate']:.2f} s")


if __name__ == "__main__":
    main()
