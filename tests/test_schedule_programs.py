"""Host logic on the CPU: the sampler programs built by diffsinger_amd/schedule.py, executed with a numpy
program interpreter around the ORACLE backbone, must reproduce the reference's golden sampler outputs
(tests/golden/g5_samplers.npz) and the reference's schedule scalars (g4_schedules.npz)."""
import os

import numpy as np
import pytest
import torch

from diffsinger_amd import schedule, synth
from oracle import backbones as ob
from prog_sim import finish, run_program

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SN_ARGS = dict(num_layers=4, num_channels=64, dilation_cycle_length=2)


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def rel_err(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


@pytest.fixture(scope="module")
def net():
    shapes = synth.backbone_param_shapes("wavenet", 32, 1, hidden_size=256, **SN_ARGS)
    params = synth.synth_state_dict(shapes, seed=45)
    return lambda x, t, c: ob.wavenet_forward(params, x, t, c, dilation_cycle_length=2)


def tables():
    return schedule.DDPMTables(schedule.linear_beta_schedule(1000))


def test_ddpm_tables_bit_exact():
    g = load("g4_schedules")
    tb = tables()
    for k in ("betas", "alphas_cumprod", "alphas_cumprod_prev", "sqrt_alphas_cumprod",
              "sqrt_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
              "posterior_variance", "posterior_log_variance_clipped", "posterior_mean_coef1",
              "posterior_mean_coef2"):
        np.testing.assert_array_equal(getattr(tb, k), g[k], err_msg=k)
    np.testing.assert_array_equal(schedule.DDPMTables(schedule.cosine_beta_schedule(1000)).betas, g["cosine_betas"])


@pytest.mark.parametrize("tag,n_keep,steps", [("full", 1000, 50), ("full20", 1000, 20), ("shallow", 400, 20)])
def test_vp_schedule_scalars_bit_exact(tag, n_keep, steps):
    """Same torch CPU ops in the same order as the reference => the SAME fp32 scalars, not just close ones."""
    g = load("g4_schedules")
    tb = tables()
    ns = schedule.VPSchedule(torch.from_numpy(tb.betas[:n_keep]), clip=True)
    np.testing.assert_array_equal(ns.log_alpha.numpy(), g[f"{tag}_log_alpha_array"])
    np.testing.assert_array_equal(ns.t_array.numpy(), g[f"{tag}_t_array"])
    ts = ns.time_uniform_steps(steps)
    np.testing.assert_array_equal(ts.numpy(), g[f"{tag}_timesteps"])
    np.testing.assert_array_equal(np.array([float(ns.lam(t)) for t in ts], np.float32), g[f"{tag}_lambda"])
    np.testing.assert_array_equal(np.array([float(ns.alpha(t)) for t in ts], np.float32), g[f"{tag}_alpha"])
    np.testing.assert_array_equal(np.array([float(ns.std(t)) for t in ts], np.float32), g[f"{tag}_sigma"])
    np.testing.assert_array_equal(np.array([float(ns.model_time(t)) for t in ts], np.float32), g[f"{tag}_model_t"])
    nu = schedule.VPSchedule(torch.from_numpy(tb.betas[:n_keep]), clip=False)
    np.testing.assert_array_equal(nu.log_alpha.numpy(), g[f"{tag}_unipc_log_alpha_array"])


def _inputs(g, tag, shallow_tb=None, t_max=None):
    bsz, t_len, nseed, n_randn, _, shallow = (int(v) for v in g[f"{tag}_meta"])
    cond = np.ascontiguousarray(np.swapaxes(synth.synth_normal((bsz, t_len, 256), nseed + 500), 1, 2))
    x = synth.synth_normal((bsz, 1, 32, t_len), nseed)
    if shallow:
        src = (synth.synth_normal((bsz, t_len, 32), nseed + 501) * 1.5 - 6.0).astype(np.float32)
        xs = np.swapaxes(((src + 12.0) / 12.0 * 2 - 1).astype(np.float32), 1, 2)[:, None]
        x = (shallow_tb.sqrt_alphas_cumprod[t_max - 1] * xs
             + shallow_tb.sqrt_one_minus_alphas_cumprod[t_max - 1] * x).astype(np.float32)
    return bsz, t_len, nseed, n_randn, cond, x


PROGRAMS = {
    "ddim10": lambda tb: schedule.ddim_program(tb, 1000, 10),
    "ddim100": lambda tb: schedule.ddim_program(tb, 1000, 100),
    "pndm20": lambda tb: schedule.plms_program(tb, 1000, 20),
    "dpm20": lambda tb: schedule.dpm_solver_pp_program(tb.betas, 20),
    "dpm50": lambda tb: schedule.dpm_solver_pp_program(tb.betas, 50),
    "dpm5": lambda tb: schedule.dpm_solver_pp_program(tb.betas, 5),
    "unipc20": lambda tb: schedule.unipc_program(tb.betas, 20),
    "unipc50": lambda tb: schedule.unipc_program(tb.betas, 50),
}


@pytest.mark.parametrize("tag", sorted(PROGRAMS))
def test_gaussian_programs_vs_golden(tag, net):
    g = load("g5_samplers")
    tb = tables()
    prog = PROGRAMS[tag](tb)
    _, _, _, _, cond, x = _inputs(g, tag)
    out = finish(run_program(prog, net, x, cond))
    # 2e-4 of the output range after <= 100 steps (fp32, oracle backbone vs torch backbone included)
    assert rel_err(out, g[f"{tag}_out"]) < 2e-4, tag
    nfe = {"ddim10": 100, "ddim100": 10, "pndm20": 51, "dpm20": 20, "dpm50": 50, "dpm5": 5,
           "unipc20": 20, "unipc50": 50}[tag]
    assert prog.nfe == nfe       # NFE per sampler


def test_shallow_programs_vs_golden(net):
    g = load("g5_samplers")
    tb = tables()
    # ancestral DDPM from K_step_infer = 20 with injected per-step noise (chunked like the product does)
    bsz, t_len, nseed, n_randn, cond, x = _inputs(g, "ddpm_shallow20", tb, 20)
    noise = np.stack([synth.synth_normal((bsz, 1, 32, t_len), nseed + 1 + i) for i in range(n_randn - 1)])
    p1, p2 = schedule.ddpm_ancestral_program(tb, 20, 8), schedule.ddpm_ancestral_program(tb, 8, 0)
    assert p1.n_noise == 12 and p2.n_noise == 8
    x = run_program(p1, net, x, cond, noise[:12])
    x = run_program(p2, net, x, cond, noise[12:])
    assert rel_err(finish(x), g["ddpm_shallow20_out"]) < 1e-4
    _, _, _, _, cond, x = _inputs(g, "dpm_shallow", tb, 400)
    out = finish(run_program(schedule.dpm_solver_pp_program(tb.betas[:400], 20), net, x, cond))
    assert rel_err(out, g["dpm_shallow_out"]) < 2e-4
    _, _, _, _, cond, x = _inputs(g, "ddim_shallow", tb, 200)
    out = finish(run_program(schedule.ddim_program(tb, 200, 10), net, x, cond))
    assert rel_err(out, g["ddim_shallow_out"]) < 1e-4


@pytest.mark.parametrize("tag,algo,nfe", [("rf_euler20", "euler", 20), ("rf_rk2_20", "rk2", 40),
                                          ("rf_rk4_20", "rk4", 80), ("rf_rk5_20", "rk5", 120)])
def test_reflow_programs_vs_golden(tag, algo, nfe, net):
    g = load("g5_samplers")
    _, _, _, _, cond, x = _inputs(g, tag)
    prog = schedule.reflow_program(algo, 20, 0.0, 1000)
    assert prog.nfe == nfe
    out = finish(run_program(prog, net, x, cond))
    assert rel_err(out, g[f"{tag}_out"]) < 5e-5, tag


def test_reflow_shallow_program(net):
    g = load("g5_samplers")
    bsz, t_len, nseed, _, _, _ = (int(v) for v in g["rf_euler_shallow_meta"])
    cond = np.ascontiguousarray(np.swapaxes(synth.synth_normal((bsz, t_len, 256), nseed + 500), 1, 2))
    noise = synth.synth_normal((bsz, 1, 32, t_len), nseed)
    src = (synth.synth_normal((bsz, t_len, 32), nseed + 501) * 1.5 - 6.0).astype(np.float32)
    xs = np.swapaxes(((src + 12.0) / 12.0 * 2 - 1).astype(np.float32), 1, 2)[:, None]
    x = (np.float32(0.4) * xs + np.float32(1 - 0.4) * noise).astype(np.float32)
    out = finish(run_program(schedule.reflow_program("euler", 20, 0.4, 1000), net, x, cond))
    assert rel_err(out, g["rf_euler_shallow_out"]) < 5e-5


def test_program_limits_and_errors():
    tb = tables()
    for prog in (schedule.plms_program(tb, 1000, 20), schedule.unipc_program(tb.betas, 20),
                 schedule.reflow_program("rk5", 3, 0.0, 1000), schedule.dpm_solver_pp_program(tb.betas, 5)):
        for ev in prog.evals:
            assert 1 <= len(ev.outs) <= 3
            for dst, terms in ev.outs:
                assert 0 <= dst < prog.n_bufs and 1 <= len(terms) <= 8
    with pytest.raises(ValueError):
        schedule.reflow_program("nope", 3, 0.0, 1000)
    with pytest.raises(AssertionError):
        schedule.dpm_solver_pp_program(tb.betas, 1)
