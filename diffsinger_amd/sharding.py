"""Multi-GPU: independent utterances sharded over the ranks of one node (one process per GPU).

The denoiser path has no cross-utterance operation (no BatchNorm; LayerNorm is per (b, t) column), so the
only exchange is BEFORE the loop (the encoder condition goes from the rank that ran the encoder to the rank
that will denoise that utterance) and AFTER it (mels come back).  Both are single point-to-point style
collectives over RCCL/xGMI (`torch.distributed` backend "nccl" on ROCm) - a scatter and a gather rooted at
rank 0; there is no collective inside the denoise loop and no all-reduce anywhere, so the per-link-bound
ring concern of xGMI does not arise.  The same code runs on `gloo` for the CPU tests.

Partition: contiguous and balanced by count (`shard_ranges`), or - for utterances of different lengths -
longest-first bin-packing by frame count (`shard_longest_first`; the reference's own length-aware batching is
utils/__init__.py:64-120 `batch_by_size`, which sorts by length and cuts batches by a token budget).

x_T is drawn per UTTERANCE from `seed + utterance index` and a ragged batch is computed per item as if it ran
alone (dsd_set_lengths), so a result depends neither on the number of ranks nor on the partition.
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

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


def shard_longest_first(lengths: Sequence[int], world: int, tile: int = 32) -> List[List[int]]:
    """Longest-processing-time-first partition of utterances of `lengths` frames over `world` ranks.

    Cost of an utterance = its number of `tile`-frame tiles (what a rank's kernels launch for it).  Utterances are
    taken longest first (ties: lower index first) and given to the rank with the least cost so far (ties: lower
    rank), which bounds the busiest rank by 4/3 - 1/(3 world) of the optimum (Graham 1969) and, more to the point
    here, by the mean load plus ONE utterance.  Deterministic; every rank's list is returned in ascending utterance
    order."""
    cost = [(int(n) + tile - 1) // tile for n in lengths]
    order = sorted(range(len(cost)), key=lambda i: (-cost[i], i))
    load = [0] * world
    shards: List[List[int]] = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        shards[r].append(i)
        load[r] += cost[i]
    return [sorted(s) for s in shards]


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


class Exchange:
    """The two collectives of a sharded run with every buffer allocated ONCE: `scatter(cond_all)` before the loop,
    `gather(mel_local)` after it.  A step of a sharded job then costs the scatter, the gather and - only where the
    shards are not equal contiguous slices of the root's tensors - one copy per shard; no allocation, no fill.

    shards: per rank, the utterance indices it denoises (ranges from `shard_ranges`, lists from
    `shard_longest_first`)."""

    def __init__(self, shards: Sequence[Sequence[int]], t_len: int, hidden: int, out_dims: int, device, src: int = 0):
        self.world, self.rank = dist.get_world_size(), dist.get_rank()
        assert len(shards) == self.world
        self.shards = [list(s) for s in shards]
        self.n_utt = sum(len(s) for s in self.shards)
        self.n_max = max((len(s) for s in self.shards), default=0)
        self.t_len, self.hidden, self.out_dims, self.src = t_len, hidden, out_dims, src
        self.device, self.stage = device, _staging(device)
        # equal contiguous shards in utterance order: the root's tensors ARE the per-rank pieces (views, no copy)
        flat = [i for s in self.shards for i in s]
        self.direct = (flat == list(range(self.n_utt)) and all(len(s) == self.n_max for s in self.shards)
                       and self.stage == device)
        self.mine = self.shards[self.rank]
        self.recv = torch.empty((self.n_max, t_len, hidden), device=self.stage, dtype=torch.float32)
        self.send = torch.zeros((self.n_max, t_len, out_dims), device=self.stage, dtype=torch.float32)
        self.pieces = self.bufs = self.out_all = None
        self.scatters = 0
        if self.rank == src:
            self.out_all = torch.zeros((self.world * self.n_max, t_len, out_dims), device=self.stage, dtype=torch.float32)
            self.bufs = list(self.out_all.view(self.world, self.n_max, t_len, out_dims).unbind(0))
            if not self.direct:
                pool = torch.zeros((self.world, self.n_max, t_len, hidden), device=self.stage, dtype=torch.float32)
                self.pieces = list(pool.unbind(0))
                self._idx = [torch.as_tensor(s, dtype=torch.long, device=self.stage) for s in self.shards]
                self._flat = torch.as_tensor(
                    [r * self.n_max + k for r, s in enumerate(self.shards) for k in range(len(s))], dtype=torch.long,
                    device=self.stage)
                self._inv = torch.empty(self.n_utt, dtype=torch.long, device=self.stage)
                self._inv[torch.as_tensor(flat, dtype=torch.long, device=self.stage)] = self._flat

    def scatter(self, cond_all: Optional[torch.Tensor]) -> torch.Tensor:
        """Rank `src` holds cond_all [n_utt, T, H]; every rank returns its shard [n_local, T, H] - a VIEW of the
        preallocated receive buffer: consume it before the next scatter.  The collective writes the buffer behind
        autograd's back (`dist.scatter` leaves `Tensor._version` alone), so the version counter is bumped here: consumers
        that cache on (data_ptr, _version) - `_NativeBackbone.prepare_cond` keys its conditioner hoist that way - see every
        scatter as the new tensor it is."""
        if self.rank == self.src:
            direct = (self.direct and cond_all.device == self.stage and cond_all.dtype == torch.float32
                      and cond_all.is_contiguous())
            if direct:
                pieces = list(cond_all.view(self.world, self.n_max, self.t_len, self.hidden).unbind(0))
            elif self.direct:       # equal contiguous shards, but the caller's tensor is elsewhere / strided / not fp32
                if self.pieces is None:
                    pool = torch.zeros((self.world, self.n_max, self.t_len, self.hidden), device=self.stage,
                                       dtype=torch.float32)
                    self.pieces = list(pool.unbind(0))
                for r, p in enumerate(self.pieces):
                    p.copy_(cond_all[r * self.n_max:(r + 1) * self.n_max])
                pieces = self.pieces
            else:
                src_t = cond_all.to(device=self.stage, dtype=torch.float32)
                for p, idx in zip(self.pieces, self._idx):
                    if idx.numel():
                        torch.index_select(src_t, 0, idx, out=p[:idx.numel()])
                pieces = self.pieces
            dist.scatter(self.recv, pieces, src=self.src)
        else:
            dist.scatter(self.recv, None, src=self.src)
        torch.autograd.graph.increment_version(self.recv)
        self.scatters += 1
        out = self.recv[:len(self.mine)]
        return out if self.stage == self.device else out.to(self.device)

    def gather(self, mel_local: torch.Tensor) -> Optional[torch.Tensor]:
        """Inverse of `scatter` for the result [n_local, T, M]; rank `src` gets [n_utt, T, M] in utterance order - with
        equal contiguous shards a VIEW of the preallocated result buffer, which the next gather overwrites: copy what
        must outlive it (the one-off `gather_mels` does)."""
        if tuple(mel_local.shape[1:]) != tuple(self.send.shape[1:]) or mel_local.dtype != self.send.dtype:
            raise ValueError(f"Exchange.gather: got {tuple(mel_local.shape)} {mel_local.dtype}, the exchange was built for "
                             f"[n, {self.t_len}, {self.out_dims}] float32 (results of another shape: gather_mels)")
        self.send[:mel_local.shape[0]].copy_(mel_local)
        if self.rank == self.src:
            dist.gather(self.send, self.bufs, dst=self.src)
            torch.autograd.graph.increment_version(self.out_all)
            if self.direct:
                return self.out_all[:self.n_utt]
            res = self.out_all.view(-1, self.t_len, self.out_dims).index_select(0, self._inv)
            return res if self.stage == self.device else res.to(self.device)
        dist.gather(self.send, None, dst=self.src)
        return None


def scatter_condition(cond_all: Optional[torch.Tensor], n_utt: int, t_len: int, hidden: int, device,
                      src: int = 0) -> torch.Tensor:
    """One-off form of Exchange.scatter over contiguous shards (allocates its buffers on every call)."""
    ex = Exchange(shard_ranges(n_utt, dist.get_world_size()), t_len, hidden, 1, device, src)
    return ex.scatter(cond_all).contiguous()


def gather_mels(mel_local: torch.Tensor, n_utt: int, dst: int = 0) -> Optional[torch.Tensor]:
    """One-off gather over contiguous shards of results of ANY trailing shape and dtype ([n, T, M], multi-feature
    [n, F, T, M], ...): rank `dst` gets a fresh [n_utt, ...] tensor in utterance order."""
    world, rank = dist.get_world_size(), dist.get_rank()
    shards = shard_ranges(n_utt, world)
    n_max = max(len(s) for s in shards)
    stage = _staging(mel_local.device)
    send = torch.zeros((n_max,) + tuple(mel_local.shape[1:]), device=stage, dtype=mel_local.dtype)
    send[:mel_local.shape[0]].copy_(mel_local)
    if rank != dst:
        dist.gather(send, None, dst=dst)
        return None
    bufs = [torch.empty_like(send) for _ in range(world)]
    dist.gather(send, bufs, dst=dst)
    res = torch.cat([b[:len(s)] for b, s in zip(bufs, shards)], 0)
    return res if stage == mel_local.device else res.to(mel_local.device)


def sharded_sample(sample_fn: Callable[..., torch.Tensor], cond_all, n_utt: int, t_len: int,
                   hidden: int, noise_shape_per_utt, seed: int, device, lengths=None,
                   partition: str = "contiguous") -> Optional[torch.Tensor]:
    """scatter cond -> per-rank sampling of its utterances -> gather mels on rank 0.

    sample_fn(cond [n, T, H], x_T [n, F, M, T]) -> mel [n, T, M] is the single-GPU path
    (e.g. `lambda c, z: diffusion(c, infer=True, noise=z)`).
    `lengths` (rank 0: one frame count per utterance, padded to `t_len`): a ragged batch - the per-rank slice is handed on
    as `sample_fn(cond, x_T, lengths=[...])` (`diffusion(..., lengths=...)`, dsd_set_lengths), so every utterance comes out
    as if run alone at its own length whatever the number of ranks.
    partition: "contiguous" (balanced by count) or "longest_first" (balanced by frames; needs `lengths`)."""
    rank, world = dist.get_rank(), dist.get_world_size()
    kwargs = {}
    lens_all = None
    if _any_rank_has(lengths is not None, device):       # every rank takes part in this broadcast
        lens = torch.zeros(n_utt, dtype=torch.int64, device=_staging(device))
        if rank == 0:
            lens.copy_(torch.as_tensor(lengths, dtype=torch.int64))
        dist.broadcast(lens, src=0)
        lens_all = [int(v) for v in lens.tolist()]
    if partition == "longest_first":
        if lens_all is None:
            raise ValueError("partition='longest_first' needs the utterance lengths")
        shards = shard_longest_first(lens_all, world)
    elif partition == "contiguous":
        shards = [list(r) for r in shard_ranges(n_utt, world)]
    else:
        raise ValueError(f"unknown partition {partition!r}")
    mine = shards[rank]
    if lens_all is not None:
        kwargs["lengths"] = [lens_all[i] for i in mine]
    ex = Exchange(shards, t_len, hidden, noise_shape_per_utt[-2], device)
    cond = ex.scatter(cond_all)
    noise = utterance_noise(noise_shape_per_utt, mine, seed, device)
    if len(mine):
        mel = sample_fn(cond, noise, **kwargs)
    else:
        mel = torch.empty((0, t_len, noise_shape_per_utt[-2]), device=device)
    return ex.gather(mel)


def _any_rank_has(flag: bool, device) -> bool:
    """True on every rank if `flag` is set on rank 0 (which is the one that holds the project's metadata)."""
    t = torch.tensor([1 if flag else 0], dtype=torch.int64, device=_staging(device))
    dist.broadcast(t, src=0)
    return bool(int(t.item()))
