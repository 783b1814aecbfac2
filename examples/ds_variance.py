#!/usr/bin/env python3
"""`.ds` project -> the same project with durations / pitch / variance curves filled in by a trained DiffSinger variance
experiment, on the HIP library only (no DiffSinger checkout needed at run time).

    python examples/ds_variance.py checkpoints/my_variance_exp song.ds -o out_dir [--predict dur pitch energy ...]
                                   [--batch-size 8] [--seed 42]

Without --predict the project is auto-completed: whatever a segment does not already carry is predicted.
"""
import argparse
import json
import pathlib

from diffsinger_amd import harness
from diffsinger_amd.hparams import load_config
from diffsinger_amd.variance import DiffSingerVariance
from diffsinger_amd.variance_harness import VarianceHarness


def main():
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("exp", type=pathlib.Path, help="experiment (work) directory")
    ap.add_argument("proj", type=pathlib.Path, help=".ds project")
    ap.add_argument("-o", "--out", type=pathlib.Path, default=pathlib.Path("."))
    ap.add_argument("--title", default=None)
    ap.add_argument("--ckpt", type=int, default=None, help="checkpoint step (default: the latest)")
    ap.add_argument("--predict", nargs="*", default=[], help="dur / pitch / energy / breathiness / voicing / tension")
    ap.add_argument("--batch-size", type=int, default=8, help="segments per launch (ragged batch)")
    ap.add_argument("--seed", type=int, default=-1)
    args = ap.parse_args()

    load_config(args.exp / "config.yaml", overrides=dict(infer=True, work_dir=str(args.exp)))
    dictionary = harness.load_phoneme_dictionary()
    maps = {}
    for name in ("spk_map", "lang_map"):
        path = args.exp / f"{name}.json"
        maps[name] = json.loads(path.read_text(encoding="utf8")) if path.exists() else {}
    model = DiffSingerVariance(len(dictionary)).cuda().eval()
    ckpt = harness.load_ckpt(model, args.exp, ckpt_steps=args.ckpt, prefix_in_ckpt="model", strict=True)
    print(f"| variance model: {ckpt}")
    h = VarianceHarness(model, dictionary, predictions=set(args.predict), spk_map=maps["spk_map"], lang_map=maps["lang_map"],
                        device="cuda")
    title = args.title or args.proj.stem
    h.run_inference(harness.load_ds(args.proj), out_dir=args.out, title=title, seed=args.seed, batch_size=args.batch_size)
    print(f"| wrote {args.out / (title + '.ds')}")


if __name__ == "__main__":
    main()
