"""Multi-GPU: independent utterances sharded over the ranks of one node (one process per GPU).

The denoiser path has no cross-utterance operation (no BatchNorm; LayerNorm is per (b, t) column), so the
only exchange is BEFORE the loop (the encoder condition goes from the rank that ran the encoder to the rank
that will denoise that utterance) and AFTER it (mels come back).  Both are single point-to-point style
collectives over RCCL/xGMI (`torch.distributed` backend "nccl" on ROCm) - a scatter and a gather rooted at
rank 0; there is no collective inside the denoise loop and no all-reduce anywhere, so the per-link-bound
ring concern of xGMI does not arise.  The same code runs on `gloo` for the CPU tests.

x_T is drawn per UTTERANCE from `seed + utterance index`, so a result does not depend on the number of ranks.
"""
from __future__ import annotations

from typing import Callable, List, Optional

import torch
import torch.distributed as dist


def shard_ranges(n_utt: int, world: int) -> List[range]:
    """Contiguous, balanced shards: the first n_utt % world ranks get one extra utterance."""
    base, extra = divmod(n_utt, world)
    out, start = [], 0
    for r in range(world):
        n = base + (1 if r < extra else 0)
        out.append(range(start, start + n))
        start += n
    return out


def utterance_noise(shape_per_utt, utt_indices, seed: int, device) -> torch.Tensor:
    """[len(utt_indices), *shape_per_utt] standard normals, one generator per utterance."""
    outs = []
    for u in utt_indices:
        g = torch.Generator(device=device)
        g.manual_seed(int(seed) + int(u))
        outs.append(torch.randn(shape_per_utt, generator=g, device=device))
    if not outs:
        return torch.empty((0,) + tuple(shape_per_utt), device=device)
    return torch.stack(outs)


def _staging(device):
    """Where the collectives' buffers live: the compute device over RCCL; host memory when the process group is `gloo`
    but the tensors are on a GPU (the rehearsal of the N-rank path on a one-GPU box: ranks share the card)."""
    return torch.device("cpu") if dist.get_backend() == "gloo" else device


def scatter_condition(cond_all: Optional[torch.Tensor], n_utt: int, t_len: int, hidden: int, device,
                      src: int = 0) -> torch.Tensor:
    """Rank `src` holds cond_all [n_utt, T, H]; every rank returns its shard [n_local, T, H]."""
    world, rank = dist.get_world_size(), dist.get_rank()
    shards = shard_ranges(n_utt, world)
    n_max = max(len(s) for s in shards)
    stage = _staging(device)
    recv = torch.empty((n_max, t_len, hidden), device=stage, dtype=torch.float32)
    if rank == src:
        pieces = []
        for s in shards:
            p = torch.zeros((n_max, t_len, hidden), device=stage, dtype=torch.float32)
            if len(s):
                p[:len(s)] = cond_all[s.start:s.stop]
            pieces.append(p)
        dist.scatter(recv, pieces, src=src)
    else:
        dist.scatter(recv, None, src=src)
    return recv[:len(shards[rank])].to(device).contiguous()


def gather_mels(mel_local: torch.Tensor, n_utt: int, dst: int = 0) -> Optional[torch.Tensor]:
    """Inverse of scatter_condition for the result [n_local, T, M]; rank `dst` gets [n_utt, T, M]."""
    world, rank = dist.get_world_size(), dist.get_rank()
    shards = shard_ranges(n_utt, world)
    n_max = max(len(s) for s in shards)
    pad = torch.zeros((n_max,) + tuple(mel_local.shape[1:]), device=_staging(mel_local.device), dtype=mel_local.dtype)
    pad[:mel_local.shape[0]] = mel_local
    if rank == dst:
        bufs = [torch.empty_like(pad) for _ in range(world)]
        dist.gather(pad, bufs, dst=dst)
        return torch.cat([b[:len(s)] for b, s in zip(bufs, shards)], dim=0).to(mel_local.device)
    dist.gather(pad, None, dst=dst)
    return None


def sharded_sample(sample_fn: Callable[..., torch.Tensor], cond_all, n_utt: int, t_len: int,
                   hidden: int, noise_shape_per_utt, seed: int, device, lengths=None) -> Optional[torch.Tensor]:
    """scatter cond -> per-rank sampling of its utterances -> gather mels on rank 0.

    sample_fn(cond [n, T, H], x_T [n, F, M, T]) -> mel [n, T, M] is the single-GPU path
    (e.g. `lambda c, z: diffusion(c, infer=True, noise=z)`).
    `lengths` (rank 0: one frame count per utterance, padded to `t_len`): a ragged batch - the per-rank slice is handed on
    as `sample_fn(cond, x_T, lengths=[...])` (`diffusion(..., lengths=...)`, dsd_set_lengths), so every utterance comes out
    as if run alone at its own length whatever the number of ranks."""
    rank, world = dist.get_rank(), dist.get_world_size()
    mine = shard_ranges(n_utt, world)[rank]
    kwargs = {}
    if _any_rank_has(lengths is not None, device):       # every rank takes part in this broadcast
        lens = torch.zeros(n_utt, dtype=torch.int64, device=device)
        if rank == 0:
            lens.copy_(torch.as_tensor(lengths, dtype=torch.int64))
        dist.broadcast(lens, src=0)
        kwargs["lengths"] = [int(v) for v in lens[mine.start:mine.stop].tolist()]
    cond = scatter_condition(cond_all, n_utt, t_len, hidden, device)
    noise = utterance_noise(noise_shape_per_utt, mine, seed, device)
    if len(mine):
        mel = sample_fn(cond, noise, **kwargs)
    else:
        mel = torch.empty((0, t_len, noise_shape_per_utt[-2]), device=device)
    return gather_mels(mel, n_utt)


def _any_rank_has(flag: bool, device) -> bool:
    """True on every rank if `flag` is set on rank 0 (which is the one that holds the project's metadata)."""
    t = torch.tensor([1 if flag else 0], dtype=torch.int64, device=device)
    dist.broadcast(t, src=0)
    return bool(int(t.item()))
