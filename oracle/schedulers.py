"""Oracle schedulers: DDIM (eta=0) and EulerDiscrete, diffusers==0.30.0 semantics.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  The reference never builds a
scheduler itself: inference uses whatever the loaded SDXL pipeline ships
(test.py:68-72 -> EulerDiscrete for stock SDXL-base) and only calls
``set_timesteps / scale_model_input / step`` (ip_adapter/custom_pipelines.py:250-252,
334,357).  BASELINE.json asks for DDIM.  Both follow SURVEY.md Appendix B; diffusers
cannot be executed here, so **parity with diffusers is unpinned** -- the tests pin the
closed-form properties instead (timestep tables, alpha-bar endpoints, exact
x0-recovery identity).
"""
import numpy as np
import torch


def _alphas_cumprod(n_train=1000, beta_start=0.00085, beta_end=0.012):
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n_train, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0)


class DDIMScheduler:
    """scaled_linear betas, clip_sample=False, set_alpha_to_one=False, steps_offset=1,
    timestep_spacing='leading', prediction_type='epsilon' (the IP-Adapter convention)."""
    order = 1
    init_noise_sigma = 1.0

    def __init__(self, num_train_timesteps=1000):
        self.num_train_timesteps = num_train_timesteps
        self.alphas_cumprod = _alphas_cumprod(num_train_timesteps)
        self.final_alpha_cumprod = self.alphas_cumprod[0]
        self.timesteps = None

    def set_timesteps(self, n, device=None):
        self.num_inference_steps = n
        r = self.num_train_timesteps // n
        ts = (np.arange(0, n) * r).round()[::-1].copy().astype(np.int64) + 1
        self.timesteps = torch.from_numpy(ts)

    def scale_model_input(self, x, t):
        return x

    def step(self, eps, t, x, eta=0.0, return_dict=False, **kw):
        t = int(t)
        prev = t - self.num_train_timesteps // self.num_inference_steps
        a = self.alphas_cumprod[t]
        ap = self.alphas_cumprod[prev] if prev >= 0 else self.final_alpha_cumprod
        x0 = (x - (1 - a) ** 0.5 * eps) / a ** 0.5
        return ((ap ** 0.5) * x0 + ((1 - ap) ** 0.5) * eps,)


class EulerDiscreteScheduler:
    """What stock SDXL-base ships (and therefore what test.py actually runs):
    leading spacing, steps_offset=1, linear sigma interpolation, epsilon prediction."""
    order = 1

    def __init__(self, num_train_timesteps=1000):
        self.num_train_timesteps = num_train_timesteps
        ac = _alphas_cumprod(num_train_timesteps)
        self.all_sigmas = (((1 - ac) / ac) ** 0.5).numpy().astype(np.float64)
        self.timesteps = None

    def set_timesteps(self, n, device=None):
        self.num_inference_steps = n
        r = self.num_train_timesteps // n
        ts = (np.arange(0, n) * r).round()[::-1].copy().astype(np.float32) + 1
        sig = np.interp(ts, np.arange(0, len(self.all_sigmas)), self.all_sigmas)
        self.sigmas = torch.from_numpy(np.concatenate([sig, [0.0]]).astype(np.float32))
        self.timesteps = torch.from_numpy(ts)
        self._i = 0

    @property
    def init_noise_sigma(self):
        # timestep_spacing == 'leading' -> sqrt(sigma_max^2 + 1)
        return float((self.sigmas.max() ** 2 + 1) ** 0.5)

    def scale_model_input(self, x, t):
        s = self.sigmas[self._i]
        return x / ((s ** 2 + 1) ** 0.5)

    def step(self, eps, t, x, return_dict=False, **kw):
        s, sn = self.sigmas[self._i], self.sigmas[self._i + 1]
        x = x.float()
        x0 = x - s * eps.float()
        d = (x - x0) / s
        self._i += 1
        return ((x + d * (sn - s)).to(eps.dtype),)
