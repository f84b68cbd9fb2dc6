"""CPU: the product schedulers' device tables against the oracle schedulers' step() (same closed forms)."""
import torch

from imagharmony_amd import schedulers as hs
from oracle import schedulers as osch
from oracle.detfill import det_randn


def _check(h, o, n=30):
    h.set_timesteps(n)
    o.set_timesteps(n)
    assert torch.equal(h.timesteps.float(), o.timesteps.float())
    tab = h.tables()
    x = det_randn((1, 4, 8, 8), 1) * float(tab["init_noise_sigma"])
    xo = x.clone()
    for i, t in enumerate(o.timesteps):
        eps = det_randn((1, 4, 8, 8), 100 + i)
        xin_o = o.scale_model_input(xo, t)
        xin_h = x if tab["in_scale"] is None else x * tab["in_scale"][i]
        assert torch.allclose(xin_h, xin_o, atol=1e-5, rtol=1e-5)
        xo = o.step(eps, t, xo)[0]
        x = tab["coef"][i, 0] * x + tab["coef"][i, 1] * eps
        assert torch.allclose(x, xo, atol=2e-4, rtol=2e-4), i
    assert abs(float(tab["init_noise_sigma"]) - float(o.init_noise_sigma)) < 1e-5


def test_ddim_tables():
    _check(hs.DDIMScheduler(), osch.DDIMScheduler())
    _check(hs.DDIMScheduler(), osch.DDIMScheduler(), n=50)       # BASELINE.json configs[3]
    _check(hs.DDIMScheduler(), osch.DDIMScheduler(), n=10)       # configs[0]


def test_euler_tables():
    _check(hs.EulerDiscreteScheduler(), osch.EulerDiscreteScheduler())
    _check(hs.EulerDiscreteScheduler(), osch.EulerDiscreteScheduler(), n=10)
    _check(hs.EulerDiscreteScheduler(), osch.EulerDiscreteScheduler(), n=50)
