"""Schedulers of the denoise loop, restated for a device-resident loop.

The reference calls ``scheduler.set_timesteps / scale_model_input / step`` on whatever scheduler the SDXL
pipeline ships (ip_adapter/custom_pipelines.py:250-252,334,357): EulerDiscrete for stock SDXL-base (what
test.py runs), DDIM eta=0 per BASELINE.json.  Both are LINEAR updates
``x' = cx*x + ce*eps`` with an optional input scale, so each scheduler reduces to three per-step tables
that the HIP kernels read from device memory (csrc/elementwise.hip EW_CONV_IN / EW_CFG_STEP):

    timesteps[i], in_scale[i] (scale_model_input), (cx[i], ce[i]) (step), init_noise_sigma.

Same public surface as diffusers for the calls the reference makes, so they also work as plain
host-side schedulers (``step`` on tensors) in tests.
"""
import numpy as np
import torch


def _alphas_cumprod(n_train=1000, beta_start=0.00085, beta_end=0.012):
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, n_train, dtype=torch.float32) ** 2
    return torch.cumprod(1.0 - betas, dim=0).double()


class _Base:
    order = 1
    num_train_timesteps = 1000

    def tables(self):
        """-> dict(timesteps f32[n], in_scale f32[n] | None, coef f32[n, 2], init_noise_sigma float)"""
        raise NotImplementedError

    def scale_model_input(self, x, t):
        i = self._index(t)
        s = self.tables()["in_scale"]
        return x if s is None else x * float(s[i])

    def step(self, eps, t, x, return_dict=False, **kw):
        i = self._index(t)
        c = self.tables()["coef"][i]
        return ((float(c[0]) * x.float() + float(c[1]) * eps.float()).to(x.dtype),)

    def _index(self, t):
        ts = self.timesteps.tolist()
        return ts.index(float(t) if isinstance(ts[0], float) else int(t))


class DDIMScheduler(_Base):
    """scaled_linear betas, clip_sample=False, set_alpha_to_one=False, steps_offset=1, leading spacing,
    epsilon prediction, eta=0."""
    init_noise_sigma = 1.0

    def __init__(self):
        self.alphas_cumprod = _alphas_cumprod()
        self.timesteps = None

    def set_timesteps(self, n, device=None):
        self.num_inference_steps = n
        r = self.num_train_timesteps // n
        self.timesteps = torch.from_numpy((np.arange(0, n) * r).round()[::-1].copy().astype(np.int64) + 1)
        self._tab = None

    def tables(self):
        if getattr(self, "_tab", None) is None:
            n = self.num_inference_steps
            r = self.num_train_timesteps // n
            ac = self.alphas_cumprod
            coef = torch.zeros(n, 2, dtype=torch.float64)
            for i, t in enumerate(self.timesteps.tolist()):
                a = ac[t]
                ap = ac[t - r] if t - r >= 0 else ac[0]
                cx = (ap / a).sqrt()
                coef[i, 0] = cx
                coef[i, 1] = (1 - ap).sqrt() - cx * (1 - a).sqrt()
            self._tab = dict(timesteps=self.timesteps.float(), in_scale=None, coef=coef.float(), init_noise_sigma=1.0)
        return self._tab


class EulerDiscreteScheduler(_Base):
    """leading spacing, steps_offset=1, linear sigma interpolation, epsilon prediction."""

    def __init__(self):
        ac = _alphas_cumprod()
        self.all_sigmas = (((1 - ac) / ac) ** 0.5).numpy()
        self.timesteps = None

    def set_timesteps(self, n, device=None):
        self.num_inference_steps = n
        r = self.num_train_timesteps // n
        ts = (np.arange(0, n) * r).round()[::-1].copy().astype(np.float32) + 1
        sig = np.interp(ts, np.arange(0, len(self.all_sigmas)), self.all_sigmas)
        self.sigmas = torch.from_numpy(np.concatenate([sig, [0.0]]).astype(np.float32))
        self.timesteps = torch.from_numpy(ts)
        self._tab = None

    @property
    def init_noise_sigma(self):
        return float((self.sigmas.max() ** 2 + 1) ** 0.5)

    def tables(self):
        if getattr(self, "_tab", None) is None:
            s = self.sigmas.double()
            coef = torch.stack([torch.ones(len(s) - 1, dtype=torch.float64), s[1:] - s[:-1]], 1)
            self._tab = dict(timesteps=self.timesteps.float(), in_scale=(1.0 / (s[:-1] ** 2 + 1).sqrt()).float(),
                             coef=coef.float(), init_noise_sigma=self.init_noise_sigma)
        return self._tab
