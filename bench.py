#!/usr/bin/env python3
"""Benchmark of the diffusion-denoiser hot path on MI355X (contract: see the task statement / DESIGN.md).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workload (BASELINE.json configs[1]): WaveNet 20 layers x 256 channels (dilation cycle 4, 128 mel bins,
hidden 256), DPM-Solver++ 2M 1000 -> 50 steps (50 NFE), B = 1 utterance of T = 1000 frames per GPU, fp32,
synthetic seeded weights/inputs.  One "step" = one whole sampling loop over the batch: the per-utterance hoist of
every layer's conditioner projection (dsd_prepare_cond - a fresh condition tensor every step, as a real caller
has), 50 NFE with fused solver updates, transposed / denormalised mel; inputs are resident in HBM when the timed
region starts.  With N > 1 ranks each rank denoises its own utterances (weak scaling); the only exchange is the
cond scatter before and the mel gather after the loop (RCCL over xGMI), both inside the timed step, through
buffers allocated once (sharding.Exchange).  --ragged: BASELINE configs[3] as SURVEY 8(d) specifies it - the
utterances' lengths are drawn from {512, 768, 1024, 1280, 1536} (seed 1234), partitioned longest-first over the
ranks and run as ragged batches (dsd_set_lengths); frames = the VALID frames.

value = mel-frames/s per denoise step = (valid frames of all N * B utterances) * NFE / seconds per step.
"""
from __future__ import annotations

import os as _os0
_os0.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
_os0.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL between ranks needs it on this platform
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_*_f32 dense peak
PEAK_BF16_MFMA_TFLOPS = 2500.0     # ... v_mfma_f32_*_bf16 dense peak (--precision bf16x3: three bf16 MFMAs per fp32 one)
PEAK_HBM_GBPS = 8000.0             # MI355X_MICROARCH.md: HBM3E spec
HOP, SR = 512, 44100


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1, help="utterances per GPU")
    ap.add_argument("--frames", type=int, default=1000, help="mel frames per utterance (T)")
    ap.add_argument("--workload", default="wavenet_dpm50", choices=["wavenet_dpm50", "lynxnet_ddim100", "acoustic_default", "acoustic_e2e", "acoustic_wav", "variance_reflow20"])
    ap.add_argument("--ragged", action="store_true",
                    help="utterance lengths drawn from {512,768,1024,1280,1536} (seed 1234), longest-first shards, ragged batches")
    ap.add_argument("--ragged-world", type=int, default=8,
                    help="--ragged: ranks of the partition the shards are taken from (default 8 = BASELINE configs[3]; 0: draw world * B lengths only)")
    ap.add_argument("--precision", default="f32", choices=["f32", "bf16x3"],
                    help="f32: the reference's arithmetic (every BASELINE number); bf16x3: opt-in split-bf16 layer GEMMs "
                         "(three bf16 MFMAs per fp32 one, fp32 accumulation; WaveNet C = 256 fused kernel) - a separate workload")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    return ap.parse_args()


def cpu_baseline(kind, params, bargs, bsz, t_len, bins=128):
    """The numpy oracle (a port of the reference algorithm, parity-pinned by tests/golden) timed on the host
    cores of this box on a bounded sample of the same workload: backbone evaluations (NFE) of the same network at
    the same (B, T) - including, like the reference, the per-NFE conditioner projections.  10-15 s of CPU work."""
    import numpy as np
    from oracle import backbones as ob
    from diffsinger_amd import synth
    from threadpoolctl import threadpool_limits
    try:
        avail = len(os.sched_getaffinity(0))
    except Exception:
        avail = os.cpu_count() or 1
    cores = max(1, min(avail, 16))          # the GPU box gives one GPU a 16-core CPU share
    limiter = threadpool_limits(limits=cores)
    x = synth.synth_normal((bsz, 1, bins, t_len), 1)
    cond = synth.synth_normal((bsz, 256, t_len), 0)
    t = np.full((bsz,), 500.0, np.float32)
    if kind == "wavenet":
        fwd = lambda: ob.wavenet_forward(params, x, t, cond, dilation_cycle_length=bargs["dilation_cycle_length"])  # noqa: E731
    else:
        fwd = lambda: ob.lynxnet_forward(params, x, t, cond, activation=bargs["activation"],  # noqa: E731
                                         strong_cond=bargs["strong_cond"])
    fwd()       # warm-up
    n, t0 = 0, time.perf_counter()
    while True:
        fwd()
        n += 1
        el = time.perf_counter() - t0
        if el > 12.0 or n >= 1000:
            break
    limiter.unregister() if hasattr(limiter, "unregister") else None
    return {"value": round(bsz * t_len * n / el, 1), "unit": "mel-frames/s per denoise step", "cores": int(cores),
            "kind": "port", "sample": f"{n} backbone evaluations (NFE) of the same {kind} at B={bsz}, T={t_len}, "
                                      f"numpy fp32 oracle, {el:.1f} s"}


def main():
    args = parse()
    import numpy as np
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    assert torch.cuda.is_available(), "bench.py needs an MI355X; the product has no CPU path"
    if os.environ.get("DSD_BENCH_SHARE_GPU") == "1":      # rehearsal of the N-rank path on a one-GPU box: ranks share cuda:0
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    use_dist = world > 1 or os.environ.get("DSD_BENCH_FORCE_DIST") == "1"      # the latter: 1-rank RCCL rehearsal
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if os.environ.get("DSD_BENCH_SHARE_GPU") == "1":   # RCCL refuses two ranks on one device: gloo over host staging
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)

    from diffsinger_amd import synth, sharding
    from diffsinger_amd.hparams import hparams
    from diffsinger_amd.diffusion import GaussianDiffusion

    B, T = args.batch, args.frames
    variance = None
    if args.workload == "variance_reflow20":
        # BASELINE config 5 per GPU (configs/variance.yaml:63-72,89-96,101-110): the pitch denoiser (20x256 WaveNet, cycle 5,
        # 64 repeat bins) and the multi-variance denoiser (10x192, cycle 4, 48 bins = energy + breathiness), rectified
        # flow, euler 20 each, on the same condition.  One "denoise step" = one NFE of each.
        kind, bargs = "wavenet", dict(num_layers=20, num_channels=256, dilation_cycle_length=5)
        hp = dict(sampling_algorithm="euler", sampling_steps=20)
        nfe, wname = 20, ("variance: pitch WaveNet 20x256 (cycle 5, 64 bins) + energy/breathiness WaveNet 10x192 "
                          "(cycle 4, 2x24 bins), rectified flow euler 20 each (configs/variance.yaml)")
    elif args.workload == "wavenet_dpm50":
        kind, bargs = "wavenet", dict(num_layers=20, num_channels=256, dilation_cycle_length=4)
        hp = dict(diff_accelerator="dpm-solver", diff_speedup=20, K_step_infer=1000)
        nfe, wname = 50, "WaveNet 20x256 (cycle 4, 128 bins), DPM-Solver++ 2M 1000->50"
    else:
        kind, bargs = "lynxnet", dict(num_layers=6, num_channels=1024, expansion_factor=2, kernel_size=31,
                                      activation="PReLU", strong_cond=True)
        hp = dict(diff_accelerator="ddim", diff_speedup=10, K_step_infer=1000)
        nfe, wname = 100, "LYNXNet 6x1024 (k31, strong_cond), DDIM 1000->100"
    shapes = synth.backbone_param_shapes(kind, 64 if args.workload == "variance_reflow20" else 128, 1, hidden_size=256,
                                         **bargs)
    params = synth.synth_state_dict(shapes, seed=42)
    hparams.clear()
    acoustic = None
    if args.workload in ("acoustic_default", "acoustic_e2e", "acoustic_wav"):
        # configs/acoustic.yaml:61-99 of the reference fork: shallow reflow (euler, 20 steps from t = 0.4) on the
        # LYNXNet above, started from the ConvNeXt aux decoder's mel - DiffSingerAcoustic.forward after the encoder
        from diffsinger_amd.toplevel import AcousticDecoder, DiffSingerAcoustic
        aux_args = dict(num_channels=512, num_layers=6, kernel_size=7, dropout_rate=0.1)
        e2e = args.workload in ("acoustic_e2e", "acoustic_wav")   # + the FastSpeech2 encoder: phoneme tokens in, mel out
        hparams.update(enc_layers=4, enc_ffn_kernel_size=3, ffn_act="gelu", dropout=0.1, num_heads=2, use_pos_embed=True,
                       rel_pos=True, use_rope=True, use_spk_id=False, num_spk=1, use_lang_id=False, num_lang=1)
        hparams.update(hidden_size=256, schedule_type="linear", infer=False, use_shallow_diffusion=True,
                       diffusion_type="reflow", T_start=0.4, T_start_infer=0.4, time_scale_factor=1000,
                       sampling_algorithm="euler", sampling_steps=20, timesteps=1000, K_step=400, K_step_infer=400,
                       backbone_type=kind, backbone_args=bargs, spec_min=[-12.0], spec_max=[0.0],
                       shallow_diffusion_args=dict(aux_decoder_arch="convnext", aux_decoder_args=aux_args,
                                                   val_gt_start=False))
        acoustic = DiffSingerAcoustic(60, 128) if e2e else AcousticDecoder(128)
        sd = dict(acoustic.state_dict())
        if e2e:
            fs2_sd = synth.synth_state_dict(synth.fs2_acoustic_param_shapes(60), seed=44)
            sd.update({"fs2." + k: torch.from_numpy(v) for k, v in fs2_sd.items()})
        sd.update({"diffusion.velocity_fn." + k: torch.from_numpy(v) for k, v in params.items()})
        aux_sd = synth.synth_state_dict(synth.convnext_param_shapes(256, 128, prefix="aux_decoder.decoder."), seed=43)
        sd.update({k: torch.from_numpy(v) for k, v in aux_sd.items()})
        acoustic.load_state_dict(sd, strict=True)
        acoustic = acoustic.to(device).eval()
        d = acoustic.diffusion
        d.denoise_fn = d.velocity_fn          # one name for the backbone below
        nfe, wname = 20, (("FastSpeech2 encoder 4x256 (120 tokens) -> " if e2e else "") +
                          "ConvNeXt aux decoder 6x512 -> shallow reflow euler 20 (t 0.4 -> 1) on LYNXNet 6x1024 "
                          "(configs/acoustic.yaml of the reference fork)" +
                          (" -> NSF-HiFiGAN 44.1 kHz waveform" if args.workload == "acoustic_wav" else ""))
        vocoder = None
        if args.workload == "acoustic_wav":      # tokens in, waveform out: the whole of scripts/infer.py's model side
            from diffsinger_amd.vocoder import Generator, NsfHifiGAN
            vh = dict(synth.NSF_HIFIGAN_DEFAULT)
            vgen = Generator(vh)
            vgen.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(
                synth.nsf_hifigan_param_shapes(vh), seed=45, gain=0.7).items()}, strict=True)
            vocoder = NsfHifiGAN(vgen.to(device).eval())
    elif args.workload == "variance_reflow20":
        from diffsinger_amd.diffusion import MultiVarianceRectifiedFlow, PitchRectifiedFlow
        hparams.update(hidden_size=256, schedule_type="linear", use_shallow_diffusion=False, infer=False, **hp)
        d = PitchRectifiedFlow(vmin=-8.0, vmax=8.0, cmin=-12.0, cmax=12.0, repeat_bins=64, backbone_type=kind,
                               backbone_args=bargs)
        d.velocity_fn.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
        vargs = dict(num_layers=10, num_channels=192, dilation_cycle_length=4)
        variance = MultiVarianceRectifiedFlow(ranges=[(-96.0, -12.0), (-96.0, -20.0)], clamps=[(-96.0, 0.0), (-96.0, 0.0)],
                                              repeat_bins=24, backbone_type=kind, backbone_args=vargs)
        variance.velocity_fn.load_state_dict({k: torch.from_numpy(v) for k, v in synth.synth_state_dict(
            synth.backbone_param_shapes(kind, 24, 2, hidden_size=256, **vargs), seed=46).items()}, strict=True)
        d, variance = d.to(device).eval(), variance.to(device).eval()
        d.denoise_fn = d.velocity_fn
        variance.use_graph = not args.no_graph
    else:
        hparams.update(hidden_size=256, schedule_type="linear", use_shallow_diffusion=False, infer=False, **hp)
        d = GaussianDiffusion(128, 1, timesteps=1000, k_step=1000, backbone_type=kind, backbone_args=bargs,
                              spec_min=[-12.0], spec_max=[0.0])
        d.denoise_fn.load_state_dict({k: torch.from_numpy(v) for k, v in params.items()}, strict=True)
        d = d.to(device).eval()
    d.use_graph = not args.no_graph
    if args.precision != "f32":
        d.denoise_fn.set_precision(args.precision, device)
        if variance is not None:
            variance.velocity_fn.set_precision(args.precision, device)
        wname += " [precision: split-bf16 layer GEMMs (hi.hi + hi.lo + lo.hi), fp32 accumulation - NOT the BASELINE arithmetic]"

    n_utt = world * B
    lengths = None
    ragged_note = ""
    if args.ragged:
        # BASELINE configs[3] as specified: 64 utterances, T drawn per utterance from {512, ..., 1536} (seed 1234),
        # partitioned longest-first by 32-frame tiles over 8 ranks.  With fewer ranks than --ragged-world (default 8) rank r
        # runs shard r of that SAME 8-way partition - per-GPU work stays what it is in the 8-GPU job (weak scaling; --gpus 8 is
        # the configuration itself).  --ragged-world 0: round 2's stand-in - only world * B lengths are drawn (8 utterances at
        # N = 1: mean 736 frames, 184 tiles - not a per-GPU share of the 64-utterance job, whose shards hold 232-240 tiles).
        assert args.workload == "wavenet_dpm50", "--ragged is defined for the WaveNet / DPM-Solver++ workload"
        import random
        rnd = random.Random(1234)
        virt = max(world, args.ragged_world) if args.ragged_world > 0 else world
        lengths_all = [rnd.choice([512, 768, 1024, 1280, 1536]) for _ in range(virt * B)]
        vshards = sharding.shard_longest_first(lengths_all, virt)[:world]
        ids = sorted(i for sh in vshards for i in sh)
        pos = {g: k for k, g in enumerate(ids)}
        shards = [[pos[g] for g in sh] for sh in vshards]
        lengths = [lengths_all[g] for g in ids]
        n_utt = len(ids)
        T = max(lengths)
        ragged_note = (f", shard(s) 0..{world - 1} of the longest-first partition of {virt * B} utterances over {virt} ranks "
                       f"({[sum((lengths_all[g] + 31) // 32 for g in sh) for sh in vshards]} tiles of 32 frames)")
    else:
        shards = [list(r) for r in sharding.shard_ranges(n_utt, world)]
    mine = shards[rank]
    my_lens = [lengths[i] for i in mine] if lengths else None
    # two condition tensors used in turn: every step sees a tensor it has not hoisted yet, as a real caller's does
    cond_bufs = None
    if rank == 0:
        cond_bufs = [torch.from_numpy(synth.synth_normal((n_utt, T, 256), k)).to(device) for k in (0, 7)]
    cond_all = cond_bufs[0] if rank == 0 else None
    noise = sharding.utterance_noise((1, 64 if variance is not None else 128, T), mine, seed=1, device=device)  # x_T, resident
    noise_v = sharding.utterance_noise((2, 24, T), mine, seed=2, device=device) if variance is not None else None
    exchange = sharding.Exchange(shards, T, 256, 3 if variance is not None else 128, device) if use_dist else None

    n_tok = 120
    mel2ph = (torch.arange(T, device=device) * n_tok // T + 1).to(torch.long)[None].expand(len(mine), T).contiguous()
    tokens = (torch.arange(n_tok, device=device) % 59 + 1).to(torch.long)[None].expand(len(mine), n_tok).contiguous()
    f0 = torch.full((len(mine), T), 220.0, device=device)
    voc_ri = voc_noise = None
    if args.workload == "acoustic_wav":          # SineGen's draws, resident like x_T
        voc_ri = torch.rand(9, device=device)
        voc_noise = torch.randn((len(mine), T * 512, 9), device=device)

    def run(c):
        if acoustic is not None:
            with torch.no_grad():       # as DiffSingerAcousticInfer.forward_model does (ds_acoustic.py:136)
                if args.workload == "acoustic_wav":
                    mel = acoustic(tokens, mel2ph, f0, infer=True, noise=noise).diff_out
                    wav = vocoder.spec2wav_torch(mel, f0=f0, rand_ini=voc_ri, noise=voc_noise)
                    return wav.view(len(mine), -1)[:, :T].unsqueeze(-1).expand(-1, -1, 128)   # gather-shaped view
                if args.workload == "acoustic_e2e":     # the condition comes from the encoder on this GPU
                    return acoustic(tokens, mel2ph, f0, infer=True, noise=noise).diff_out
                return acoustic(c, mel2ph, infer=True, noise=noise).diff_out
        if variance is not None:        # [n, T, 3]: delta pitch, energy, breathiness (toplevel.py:262-309 runs them in this order)
            return torch.stack([d(c, infer=True, noise=noise)] + list(variance(c, infer=True, noise=noise_v)), dim=-1)
        if my_lens is not None:
            return d(c, infer=True, noise=noise, lengths=my_lens)
        return d(c, infer=True, noise=noise)

    def step(i):
        if not use_dist:
            return run(cond_bufs[i & 1])
        c = exchange.scatter(cond_bufs[i & 1] if rank == 0 else None)
        return exchange.gather(run(c))

    def sync():
        torch.cuda.synchronize(device)
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize(device)

    for i in range(args.warmup):
        out = step(i)
    sync()
    t0 = time.perf_counter()
    for i in range(args.steps):
        out = step(args.warmup + i)
    torch.cuda.synchronize(device)
    elapsed = time.perf_counter() - t0          # this rank's K steps, from the common start to its own last kernel
    sync()
    dist_info = None
    if use_dist:                                # the job's time = the slowest rank's
        mine_ms = elapsed / max(args.steps, 1) * 1e3
        el = torch.tensor([elapsed], device=sharding._staging(device), dtype=torch.float64)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el.item())
        # what only a working N-rank collective can produce: every rank contributes a one (SUM), its own time and its device
        ones = torch.ones(1, device=sharding._staging(device), dtype=torch.float64)
        dist.all_reduce(ones, op=dist.ReduceOp.SUM)
        per = torch.zeros(world, device=sharding._staging(device), dtype=torch.float64)
        per[rank] = mine_ms
        dist.all_reduce(per, op=dist.ReduceOp.SUM)
        devs = torch.zeros(world, device=sharding._staging(device), dtype=torch.float64)
        devs[rank] = float(torch.cuda.current_device())
        dist.all_reduce(devs, op=dist.ReduceOp.SUM)
        try:
            ver = ".".join(str(v) for v in torch.cuda.nccl.version())
        except Exception:
            ver = None
        dist_info = {"backend": dist.get_backend(), "ranks_seen": int(round(float(ones.item()))),
                     "rccl_version": ver, "ms_per_step_by_rank": [round(float(v), 4) for v in per.tolist()],
                     "device_by_rank": [int(v) for v in devs.tolist()], "exchange_scatters": exchange.scatters,
                     "utterances_by_rank": [len(sh) for sh in shards]}
    if rank == 0:
        assert out is not None and torch.isfinite(out).all()
    sec_per_step = elapsed / max(args.steps, 1)
    utt_frames = sum(lengths) if lengths else n_utt * T
    frames = utt_frames * nfe
    value = frames / sec_per_step

    stats = dict(d.denoise_fn.stats())
    if variance is not None:
        for k, v in variance.velocity_fn.stats().items():
            if k in ("flops_per_frame_nfe", "bytes_per_frame_nfe"):
                stats[k] += v
    result = {
        "metric": "mel-frames/sec (128-bin, 44.1 kHz hop) per denoise step; end-to-end RTF",
        "value": round(value, 1),
        "unit": "mel-frames/s per denoise step",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(sec_per_step * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if args.precision == "f32" else "bf16x3 (f32 accumulate)", "data": "synthetic",
        "config": {"workload": wname + (", utterance lengths drawn from {512,768,1024,1280,1536} (seed 1234)" + ragged_note if lengths else ""),
                   "utterances_per_gpu": B, "frames": (round(utt_frames / n_utt, 1) if lengths else T), "nfe": nfe,
                   "hipgraph": bool(d.use_graph),
                   "sharding": f"{world} rank(s) x {B} utterance(s), " + ("longest-first by length, ragged batches, " if lengths else "")
                               + "cond scatter + mel gather"},
        "rtf": round(sec_per_step / (utt_frames * HOP / SR), 6),
        "ms_per_nfe": round(sec_per_step * 1e3 / nfe, 5),
        "path_tflops": round(stats["flops_per_frame_nfe"] * frames / sec_per_step / 1e12 / world, 3),
        "path_mfma_frac": round(stats["flops_per_frame_nfe"] * frames / sec_per_step / 1e12 / world / PEAK_FP32_MFMA_TFLOPS, 4),      # of the FP32 peak, whatever the precision mode
        "path_hbm_frac": round(stats["bytes_per_frame_nfe"] * frames / sec_per_step / 1e9 / world / PEAK_HBM_GBPS, 5),
    }
    if dist_info is not None:
        result["distributed"] = dist_info

    if rank == 0 and not args.no_roofline:
        # The layer kernels of this workload (WaveNet: the fused layer kernel / the row-split or GEMM pair, per tile halo and per
        # segment of a mixed plan; LYNXNet: the two pointwise GEMMs), each timed by hipEvents attached to the dispatch itself on
        # the stream it is launched on, in one eager pass of the same workload (every 7th launch of a class carries events);
        # the library reports per class the instantiation that ran and its algorithmic FLOPs / bytes per launch.
        import ctypes as C
        from diffsinger_amd import _lib
        handles = [d.denoise_fn._handle] + ([variance.velocity_fn._handle] if variance is not None else [])
        for h in handles:
            _lib.check(h, _lib.lib().dsd_kernel_timing(h, 1), "dsd_kernel_timing")
        t_pass = time.perf_counter()
        run(cond_all[mine] if use_dist else cond_all)
        torch.cuda.synchronize(device)
        t_pass = (time.perf_counter() - t_pass) * 1e3
        classes, empty_us = [], 0.0
        for h in handles:
            arr = (_lib.DsdKernelTime * 8)()
            n, empty_ms = C.c_int32(), C.c_double()
            _lib.check(h, _lib.lib().dsd_kernel_timing_classes(h, arr, 8, C.byref(n), C.byref(empty_ms)), "dsd_kernel_timing_classes")
            empty_us = max(empty_us, empty_ms.value * 1e3)
            for k in arr[:n.value]:
                classes.append({"kernel": k.name.decode(), "launches_per_step": round(k.launches / max(k.evaluations, 1) * nfe),
                                "launches_timed": int(k.launches_timed), "avg_launch_us": k.mean_ms * 1e3,
                                "flops_per_launch": k.flops_per_launch, "bytes_per_launch": k.bytes_per_launch})
            _lib.check(h, _lib.lib().dsd_kernel_timing(h, 0), "dsd_kernel_timing")
        # committed rocprofv3 evidence of this same command (tools/collect_profiles.sh -> tools/summarize_profile.py):
        # per-kernel average duration of the --kernel-trace --stats run and the fabric-side bytes of the --pmc passes
        prof, prof_src = {}, None
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            ent = tj.get(f"{args.workload}{'_ragged' if lengths else ''}{'_' + args.precision if args.precision != 'f32' else ''}/B{B}/T{T}")
            if ent:
                prof_src = ent.get("source")
                prof = ent["kernels"] if "kernels" in ent else {ent["kernel"]: ent}
        except Exception:
            pass

        def rocprof_of(name):
            hits = [v for k, v in prof.items() if ("dsd::" + name + "(") in k]
            return hits[0] if len(hits) == 1 else None

        # split-bf16 kernels execute THREE bf16 MFMA FLOPs per algorithmic fp32 FLOP and are priced against the bf16 peak
        x3 = args.precision == "bf16x3"

        def peak_of(name):
            return PEAK_BF16_MFMA_TFLOPS if "x3" in name else PEAK_FP32_MFMA_TFLOPS

        def mult_of(name):
            return 3.0 if "x3" in name else 1.0

        tot_fl = tot_by = tot_ev = tot_rp = tot_tr = 0.0
        n_launch, cov_rp, cov_tr = 0, 0.0, 0.0
        for c in classes:
            r = rocprof_of(c["kernel"])
            w = c["launches_per_step"]
            c["rocprof_avg_launch_us"] = round(r["rocprof_avg_ns"] / 1e3, 3) if r else None
            c["traffic"] = r.get("traffic_bytes_per_launch") if r else None
            c["mfma_busy_frac_profiled"] = r.get("mfma_busy_frac_profiled") if r else None
            c["frac_events"] = round(mult_of(c["kernel"]) * c["flops_per_launch"] / (c["avg_launch_us"] * 1e-6) / 1e12 / peak_of(c["kernel"]), 4)
            c["frac"] = (round(mult_of(c["kernel"]) * c["flops_per_launch"] / (r["rocprof_avg_ns"] * 1e-9) / 1e12 / peak_of(c["kernel"]), 4) if r else None)
            c["avg_launch_us"] = round(c["avg_launch_us"], 3)
            tot_fl += w * c["flops_per_launch"]
            tot_by += w * c["bytes_per_launch"]
            tot_ev += w * c["avg_launch_us"] * 1e-6
            n_launch += w
            # a kernel below the profile's 1 % cut (the small net's edge kernel ...) enters with its event time / algorithmic bytes
            if r:
                tot_rp += w * r["rocprof_avg_ns"] * 1e-9
                cov_rp += w * c["avg_launch_us"] * 1e-6
            else:
                tot_rp += w * c["avg_launch_us"] * 1e-6
            if r and r.get("traffic_bytes_per_launch") is not None:
                tot_tr += w * r["traffic_bytes_per_launch"]
                cov_tr += w * c["avg_launch_us"] * 1e-6
            else:
                tot_tr += w * c["bytes_per_launch"]
        # `frac` is the figure a reader can recompute from profiles/ (rocprofv3 averages of the committed trace of this
        # command) when that trace exists for every kernel of the set; the dispatch-event figure of THIS run rides along as
        # frac_events (events on the dispatch add ~1-2 us of command-processor time to a 5-70 us kernel).
        all_rp = tot_ev > 0 and cov_rp / tot_ev >= 0.97          # the committed trace covers (nearly) all of the set's time
        all_tr = tot_ev > 0 and cov_tr / tot_ev >= 0.97
        sec = tot_rp if all_rp else tot_ev
        ach = tot_fl / sec / 1e12 if sec > 0 else 0.0
        ach_ev = tot_fl / tot_ev / 1e12 if tot_ev > 0 else 0.0
        peak = PEAK_BF16_MFMA_TFLOPS if x3 else PEAK_FP32_MFMA_TFLOPS
        if x3:      # executed bf16 FLOPs of the split-bf16 kernels (3 per algorithmic FLOP); the fp32 kernels of the set as they are
            fl3 = sum(c["launches_per_step"] * c["flops_per_launch"] * mult_of(c["kernel"]) for c in classes)
            ach, ach_ev = fl3 / sec / 1e12 if sec > 0 else 0.0, fl3 / tot_ev / 1e12 if tot_ev > 0 else 0.0
        result["roofline"] = {
            "bound": "mfma", "achieved": round(ach, 3), "peak": peak, "unit": "TFLOP/s",
            "frac": round(ach / peak, 4),
            "frac_source": ("rocprofv3 --kernel-trace --stats averages of the committed profile: " + str(prof_src)) if all_rp
                           else "hip events on the dispatches of this run (no committed rocprofv3 trace of this configuration)",
            "frac_events": round(ach_ev / peak, 4),
            "traffic": int(round(tot_tr / max(n_launch, 1))) if all_tr else None,
            "traffic_source": prof_src if all_tr else None,
            "scope": "time-weighted over the layer kernels of one step (all launches of `kernels`); per-launch means",
            "kernel": classes[0]["kernel"] if classes else None,
            # the set's largest member on its own (what round 1 / 2 reported as `frac`): 0.55 for the B = 1 conv kernel
            "dominant": ({"kernel": classes[0]["kernel"], "frac": classes[0]["frac"], "frac_events": classes[0]["frac_events"],
                          "share_of_set_time": round(classes[0]["launches_per_step"] * classes[0]["avg_launch_us"] * 1e-6 / tot_ev, 4)}
                         if classes else None),
            "share_of_step": round(tot_ev / sec_per_step, 4),
            "launches_per_step": int(n_launch),
            "avg_launch_us": round(sec / max(n_launch, 1) * 1e6, 3),
            "avg_launch_us_events": round(tot_ev / max(n_launch, 1) * 1e6, 3),
            "algorithmic_flops_per_launch": tot_fl / max(n_launch, 1), "algorithmic_bytes_per_launch": tot_by / max(n_launch, 1),
            "hbm_achieved_GBps": round(tot_by / sec / 1e9, 1) if sec > 0 else 0.0,
            "hbm_frac": round(tot_by / sec / 1e9 / PEAK_HBM_GBPS, 5) if sec > 0 else 0.0,
            "timing_pass_ms": round(t_pass, 2), "empty_event_pair_us": round(empty_us, 3),
            "plan": {k: stats[k] for k in ("kernels_per_nfe", "layer_launches", "fused_tiles", "split_tiles", "precision")},
            "kernels": classes}
        if x3:
            result["roofline"]["note"] = ("split-bf16 mode: `achieved` counts the executed bf16 MFMA FLOPs (3 per algorithmic fp32 FLOP) against "
                                          "the bf16 peak; the kernel is bound by its weight stream from L2 (2 MB per 32-frame tile, hi + lo = 4 "
                                          "bytes per weight, at the ~70 GB/s a CU takes from L2), not by the MFMA pipe - DESIGN.md 4.7")
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(kind, params, bargs, B, T, bins=64 if variance is not None else 128)
    if rank == 0:
        print(json.dumps(result), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
