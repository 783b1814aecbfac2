"""sharding.Exchange feeding the NATIVE backbone, on RCCL (backend "nccl", one rank - what the one-GPU box allows): the receive
buffer is allocated once and every scatter writes it behind autograd's back, while `_NativeBackbone.prepare_cond` keys its
conditioner hoist on (data_ptr, _version, shape, stride).  Round 2's Exchange left `_version` alone: on nccl, where the view of
the receive buffer goes straight to the backbone, every step after the first would have run on the FIRST step's conditioner
projections.  Two different conditions through ONE Exchange must give the two single-GPU results."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist

pytestmark = pytest.mark.gpu

from diffsinger_amd import sharding, synth  # noqa: E402
from gpu_util import dev, make_backbone, set_hp  # noqa: E402


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_exchange_reused_buffer_is_a_new_condition_for_the_backbone():
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    set_hp()
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    device = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=device)
    try:
        args = dict(num_layers=3, num_channels=256, dilation_cycle_length=3)
        net, _ = make_backbone("wavenet", 128, 1, args, 42)
        bsz, t_len = 2, 160
        x = dev(synth.synth_normal((bsz, 1, 128, t_len), 1))
        t = dev(np.array([100.0, 700.0], np.float32))
        conds = [dev(synth.synth_normal((bsz, t_len, 256), k)) for k in (3, 4)]
        with torch.no_grad():
            want = [net(x, t, c.transpose(1, 2)).clone() for c in conds]        # single-GPU: the caller's own tensors
        assert not torch.equal(want[0], want[1])
        ex = sharding.Exchange([list(range(bsz))], t_len, 256, 128, device)
        assert ex.stage == device                                               # RCCL: the collective's buffers are on the GPU
        got, ptrs = [], []
        with torch.no_grad():
            for c in conds + conds[:1]:
                piece = ex.scatter(c)
                ptrs.append(piece.data_ptr())
                got.append(net(x, t, piece.transpose(1, 2)).clone())
        torch.cuda.synchronize()
        assert len(set(ptrs)) == 1                                              # the SAME buffer every time ...
        for i, k in enumerate((0, 1, 0)):
            assert torch.equal(got[i], want[k]), f"scatter {i}: the backbone ran on a stale condition"     # ... new contents
        mel = ex.gather(got[1][:, 0].transpose(1, 2).contiguous())
        assert torch.equal(mel, got[1][:, 0].transpose(1, 2))
        net.release_native()
    finally:
        dist.destroy_process_group()
