"""TEST INFRASTRUCTURE ONLY -- a restatement of diffusers.schedulers.DDPMScheduler as the reference scripts configure it
(scripts/evaluate.py:186-202, scripts/train.py:65-88: `trained_betas`, prediction_type="v_prediction", clip_sample=False, the
defaults otherwise: variance_type "fixed_small", timestep_spacing "leading", steps_offset 0).

`diffusers` is a third-party dependency that is absent from the reference tree and from this image, and the reference pins no
version of it: this scheduler is PARITY UNPINNED.  It restates the published DDPM ancestral step (Ho et al. 2020, eq. 6-7, with
the v-parameterisation of Salimans & Ho 2022) and exists only so that the REAL reference class and the product mirror can be
driven by the SAME scheduler object in the golden fixtures and tests (`DiffModernUNet.forward(..., noise_scheduler=...)`).
"""
from types import SimpleNamespace

import torch


class DDPMSchedulerRestated:
    def __init__(self, trained_betas, seed: int = 0):
        self.betas = torch.tensor(list(trained_betas), dtype=torch.float32)
        self.alphas = 1.0 - self.betas
        self.alphas_cumprod = torch.cumprod(self.alphas, dim=0)
        self.num_train_timesteps = len(self.betas)
        self.num_inference_steps = None
        self.timesteps = torch.arange(self.num_train_timesteps - 1, -1, -1)
        self._seed = seed
        self.reseed()

    def reseed(self):
        """variance noise comes from the scheduler's OWN host generator: identical for every model driven with the same seed"""
        self._gen = torch.Generator().manual_seed(self._seed)

    def set_timesteps(self, num_inference_steps: int):
        if num_inference_steps > self.num_train_timesteps:
            raise ValueError("more inference steps than training steps")
        self.num_inference_steps = num_inference_steps
        ratio = self.num_train_timesteps // num_inference_steps
        self.timesteps = (torch.arange(0, num_inference_steps) * ratio).round().flip(0).long()       # "leading" spacing

    def step(self, model_output: torch.Tensor, timestep, sample: torch.Tensor):
        t = int(timestep)
        n = self.num_inference_steps or self.num_train_timesteps
        prev_t = t - self.num_train_timesteps // n
        a_t = self.alphas_cumprod[t].item()
        a_prev = self.alphas_cumprod[prev_t].item() if prev_t >= 0 else 1.0
        b_t, b_prev = 1.0 - a_t, 1.0 - a_prev
        cur_a = a_t / a_prev
        cur_b = 1.0 - cur_a
        x0 = (a_t ** 0.5) * sample - (b_t ** 0.5) * model_output                       # v-prediction
        prev = (a_prev ** 0.5 * cur_b / b_t) * x0 + (cur_a ** 0.5 * b_prev / b_t) * sample
        if t > 0:
            var = max(b_prev / b_t * cur_b, 1e-20)                                     # "fixed_small"
            noise = torch.randn(model_output.shape, generator=self._gen).to(model_output.device)
            prev = prev + (var ** 0.5) * noise
        return SimpleNamespace(prev_sample=prev)
