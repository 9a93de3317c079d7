"""MI355X-native mirror of diffusion/diffusion.py: GaussianDiffusion (schedules, samplers) around the WaveNet denoiser
(SURVEY.md §8f row 2).  Same constructor, buffers (`state_dict` keys) and `forward` signature as the reference.

Inference samplers implemented: plain ancestral sampling (`p_sample`, :155-162), DDIM (`p_sample_ddim`, :143-153),
PNDM/PLMS (`p_sample_plms`, :164-199), DPM-Solver / DPM-Solver++ and UniPC in the configuration this file calls them with
(multistep order 2, :257-303 -> diffusion/dpm_solver_pytorch.py).  Every per-step update is a scalar-coefficient
combination of [B,1,M,T] tensors: the coefficients come from the host copy of the schedule (all batch items share the
step index, so no device gather / sync), the arithmetic runs as svc_ew_f32 launches.  UniPC (:339-371 -> diffusion/uni_pc.py,
variant bh2, multistep order 2) is restated the same way.  Training (`infer=False` -> p_losses, :210-243) runs on the autograd ops of
svc_autograd.py."""
from collections import deque
from functools import partial

import numpy as np
import torch
from torch import nn

import svc_autograd as A
import svc_hip as S


def linear_beta_schedule(timesteps, max_beta=0.02):
    return np.linspace(1e-4, max_beta, timesteps)


def cosine_beta_schedule(timesteps, s=0.008):
    steps = timesteps + 1
    x = np.linspace(0, steps, steps)
    alphas_cumprod = np.cos(((x / steps) + s) / (1 + s) * np.pi * 0.5) ** 2
    alphas_cumprod = alphas_cumprod / alphas_cumprod[0]
    betas = 1 - (alphas_cumprod[1:] / alphas_cumprod[:-1])
    return np.clip(betas, a_min=0, a_max=0.999)


beta_schedule = {"cosine": cosine_beta_schedule, "linear": linear_beta_schedule}


def _lin(a, x, b, y):
    """a*x + b*y on the device (one svc_ew_f32 launch)."""
    return S.ew(S.EW_ADD, x.contiguous(), y.contiguous(), alpha=float(a), beta=float(b))


class GaussianDiffusion(nn.Module):
    def __init__(self, denoise_fn, out_dims=128, timesteps=1000, k_step=1000, max_beta=0.02, spec_min=-12, spec_max=2):
        super().__init__()
        self.denoise_fn = denoise_fn
        self.out_dims = out_dims
        betas = beta_schedule["linear"](timesteps, max_beta=max_beta)
        alphas = 1. - betas
        alphas_cumprod = np.cumprod(alphas, axis=0)
        alphas_cumprod_prev = np.append(1., alphas_cumprod[:-1])
        timesteps, = betas.shape
        self.num_timesteps = int(timesteps)
        self.k_step = k_step if 0 < k_step < timesteps else timesteps
        self.noise_list = deque(maxlen=4)
        to_torch = partial(torch.tensor, dtype=torch.float32)
        self.register_buffer("betas", to_torch(betas))
        self.register_buffer("alphas_cumprod", to_torch(alphas_cumprod))
        self.register_buffer("alphas_cumprod_prev", to_torch(alphas_cumprod_prev))
        self.register_buffer("sqrt_alphas_cumprod", to_torch(np.sqrt(alphas_cumprod)))
        self.register_buffer("sqrt_one_minus_alphas_cumprod", to_torch(np.sqrt(1. - alphas_cumprod)))
        self.register_buffer("log_one_minus_alphas_cumprod", to_torch(np.log(1. - alphas_cumprod)))
        self.register_buffer("sqrt_recip_alphas_cumprod", to_torch(np.sqrt(1. / alphas_cumprod)))
        self.register_buffer("sqrt_recipm1_alphas_cumprod", to_torch(np.sqrt(1. / alphas_cumprod - 1)))
        posterior_variance = betas * (1. - alphas_cumprod_prev) / (1. - alphas_cumprod)
        self.register_buffer("posterior_variance", to_torch(posterior_variance))
        self.register_buffer("posterior_log_variance_clipped", to_torch(np.log(np.maximum(posterior_variance, 1e-20))))
        self.register_buffer("posterior_mean_coef1", to_torch(betas * np.sqrt(alphas_cumprod_prev) / (1. - alphas_cumprod)))
        self.register_buffer("posterior_mean_coef2", to_torch((1. - alphas_cumprod_prev) * np.sqrt(alphas) / (1. - alphas_cumprod)))
        self.register_buffer("spec_min", torch.FloatTensor([spec_min])[None, None, :out_dims])
        self.register_buffer("spec_max", torch.FloatTensor([spec_max])[None, None, :out_dims])
        self._host = None

    # -- host copy of the (fp32) schedule: step coefficients without device round trips -------------------------------
    def _h(self, name, t):
        if self._host is None or self._host[0] != self.betas._version:
            self._host = (self.betas._version, {k: v.detach().cpu().numpy().astype(np.float32).reshape(-1)
                                                for k, v in self.named_buffers(recurse=False)})
        return np.float32(self._host[1][name][int(t)])

    def _tvec(self, i, b, device):
        return torch.full((b,), int(i), device=device, dtype=torch.long)

    def _denoise(self, x, i, cond):
        return self.denoise_fn(x, self._tvec(i, x.shape[0], x.device), cond=cond)

    # -- samplers -------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def p_sample_ddim(self, x, t, interval, cond):
        """Reference :143-153; t: python int step shared by the batch."""
        a_t = self._h("alphas_cumprod", t)
        a_prev = self._h("alphas_cumprod", max(t - interval, 0))
        noise_pred = self._denoise(x, t, cond)
        c1 = np.sqrt(a_prev) / np.sqrt(a_t)
        c2 = np.sqrt(a_prev) * (np.sqrt((1 - a_prev) / a_prev) - np.sqrt((1 - a_t) / a_t))
        return _lin(c1, x, c2, noise_pred)

    @torch.no_grad()
    def p_sample(self, x, t, cond, clip_denoised=True, repeat_noise=False, noise=None):
        """Reference :134-141,155-162."""
        noise_pred = self._denoise(x, t, cond)
        x_recon = _lin(self._h("sqrt_recip_alphas_cumprod", t), x, -self._h("sqrt_recipm1_alphas_cumprod", t), noise_pred)
        x_recon = S.ew(S.EW_CLAMP, x_recon, alpha=-1.0, beta=1.0)
        mean = _lin(self._h("posterior_mean_coef1", t), x_recon, self._h("posterior_mean_coef2", t), x)
        if t == 0:
            return mean
        if noise is None:
            noise = torch.randn(x.shape, device=x.device)
        return _lin(1.0, mean, np.exp(np.float32(0.5) * self._h("posterior_log_variance_clipped", t)), noise)

    @torch.no_grad()
    def p_sample_plms(self, x, t, interval, cond, clip_denoised=True, repeat_noise=False):
        """Reference :164-199 (pseudo linear multi-step)."""
        def get_x_pred(x, noise_t, t):
            a_t = self._h("alphas_cumprod", t)
            a_prev = self._h("alphas_cumprod", max(t - interval, 0))
            a_t_sq, a_prev_sq = np.sqrt(a_t), np.sqrt(a_prev)
            kx = (a_prev - a_t) * (1 / (a_t_sq * (a_t_sq + a_prev_sq)))
            kn = (a_prev - a_t) * (1 / (a_t_sq * (np.sqrt((1 - a_prev) * a_t) + np.sqrt((1 - a_t) * a_prev))))
            return _lin(1 + kx, x, -kn, noise_t)

        nl = self.noise_list
        noise_pred = self._denoise(x, t, cond)
        if len(nl) == 0:
            x_pred = get_x_pred(x, noise_pred, t)
            noise_pred_prev = self._denoise(x_pred, max(t - interval, 0), cond)
            prime = _lin(0.5, noise_pred, 0.5, noise_pred_prev)
        elif len(nl) == 1:
            prime = _lin(1.5, noise_pred, -0.5, nl[-1])
        elif len(nl) == 2:
            prime = _lin(1.0, _lin(23 / 12, noise_pred, -16 / 12, nl[-1]), 5 / 12, nl[-2])
        else:
            prime = _lin(1.0, _lin(55 / 24, noise_pred, -59 / 24, nl[-1]), 1.0, _lin(37 / 24, nl[-2], -9 / 24, nl[-3]))
        x_prev = get_x_pred(x, prime, t)
        nl.append(noise_pred)
        return x_prev

    # -- DPM-Solver / DPM-Solver++ (the reference's default samplers, diffusion/diffusion.py:257-303) -------------------------
    def _sample_dpm_solver(self, x, cond, t, steps, plus):
        """The one configuration the reference calls (diffusion.py:296-302): multistep, order 2, uniform time steps,
        `lower_order_final`, on NoiseScheduleVP('discrete', betas[:t]) — diffusion/dpm_solver_pytorch.py:80-166 (schedule),
        :269-296 (model time / noise model), :425-449 (data prediction), :545-590 (first update), :793-850 (second update),
        :1185-1224 (the multistep driver).  The schedule is scalar fp32 arithmetic on the host (same formulas and operation
        order as the library, which evaluates them on 1-element tensors); every update is a scalar-coefficient combination
        of [B,1,M,T] tensors = svc_ew_f32 launches; the denoiser gets the library's fractional model time."""
        if steps < 2:
            raise ValueError("dpm-solver needs t // infer_speedup >= 2 steps (the library asserts steps >= order)")
        f32 = np.float32
        betas = self._host_arr("betas")[:t]
        log_alphas = (f32(0.5) * np.log(f32(1) - betas)).astype(f32).cumsum(dtype=f32)
        # numerical_clip_alpha (:112-123): drop the tail whose half-logSNR is below -5.1
        lambs = log_alphas - f32(0.5) * np.log(f32(1) - np.exp(f32(2) * log_alphas))
        idx = int(np.searchsorted(lambs[::-1], f32(-5.1)))
        if idx > 0:
            log_alphas = log_alphas[:-idx]
        N = log_alphas.shape[0]
        t_arr = torch.linspace(0., 1., N + 1)[1:].numpy()

        def log_alpha(tc):                                     # interpolate_fn(:1255-1295): piecewise linear, extrapolating
            i = int(np.searchsorted(t_arr, tc, side="left"))
            i0 = 0 if i == 0 else (N - 2 if i == N else i - 1)
            x0, x1, y0, y1 = t_arr[i0], t_arr[i0 + 1], log_alphas[i0], log_alphas[i0 + 1]
            return f32(y0 + (tc - x0) * (y1 - y0) / (x1 - x0))
        std = lambda la: f32(np.sqrt(f32(1) - np.exp(f32(2) * la)))
        lam = lambda la: f32(la - f32(0.5) * np.log(f32(1) - np.exp(f32(2) * la)))
        ts = torch.linspace(1., 1. / N, steps + 1).numpy()    # get_time_steps('time_uniform')
        B = x.shape[0]

        def model(xc, tc):                                     # model_fn: noise model at the library's model time, x0 for '++'
            t_in = torch.full((B,), float((tc - f32(1. / N)) * f32(N)), device=xc.device, dtype=torch.float32)
            noise = self.denoise_fn(xc, t_in, cond=cond)
            if not plus:
                return noise
            la = log_alpha(tc)
            a = f32(np.exp(la))
            return _lin(f32(1) / a, xc, -std(la) / a, noise)   # (x - sigma_t * noise) / alpha_t

        def update(xc, m_prev1, m_prev0, t_prev1, t_prev0, tc, order):
            la0, lat = log_alpha(t_prev0), log_alpha(tc)
            h = lam(lat) - lam(la0)
            if plus:
                c_x = std(lat) / std(la0)
                c_m = -(f32(np.exp(lat)) * f32(np.expm1(-h)))
            else:
                c_x = f32(np.exp(lat - la0))
                c_m = -(std(lat) * f32(np.expm1(h)))
            xt = _lin(c_x, xc, c_m, m_prev0)
            if order == 2:                                     # - 0.5 * coeff * D1_0,  D1_0 = (m0 - m1) / r0
                r0 = (lam(la0) - lam(log_alpha(t_prev1))) / h
                c_d = f32(0.5) * c_m * (f32(1) / r0)
                xt = _lin(1.0, xt, 1.0, _lin(c_d, m_prev0, -c_d, m_prev1))
            return xt

        t_prev = [ts[0]]
        m_prev = [model(x, ts[0])]
        x = update(x, None, m_prev[0], None, ts[0], ts[1], 1)
        t_prev.append(ts[1])
        m_prev.append(model(x, ts[1]))
        for step in range(2, steps + 1):
            order = min(2, steps + 1 - step) if steps < 10 else 2
            x = update(x, m_prev[0], m_prev[1], t_prev[0], t_prev[1], ts[step], order)
            t_prev = [t_prev[1], ts[step]]
            m_prev = [m_prev[1], model(x, ts[step]) if step < steps else None]
        return x

    # -- UniPC (diffusion/diffusion.py:339-371 -> diffusion/uni_pc.py) ------------------------------------------------------
    def _sample_unipc(self, x, cond, t, steps):
        """The one configuration the reference calls (diffusion.py:356-371): UniPC(variant='bh2', data prediction),
        multistep, order 2, uniform time steps, on NoiseScheduleVP('discrete', betas[:t]) — diffusion/uni_pc.py:70-81,103-135
        (schedule; unlike the DPM-Solver library no tail clipping), :172-196 (model time), :287-296 (data prediction),
        :473-590 (the B(h) predictor / corrector) and :592-672 (driver: first step order 1 with corrector, then order-2
        predictor + corrector, last step order 1 without corrector; the model value kept for a time step is the one evaluated
        at the PREDICTED state, so every step costs one denoiser call).  The schedule is fp32 scalar arithmetic on the host —
        evaluated with the same torch CPU ops in the same order as the library does on its 1-element tensors, including
        torch.linalg.solve for the 2x2 corrector system; every state update is a scalar-coefficient combination of [B,1,M,T]
        tensors = svc_ew_f32 launches."""
        if steps < 2:
            raise ValueError("unipc needs t // infer_speedup >= 2 steps (the library asserts steps >= order)")
        betas = torch.from_numpy(np.ascontiguousarray(self._host_arr("betas")[:t], dtype=np.float32))
        log_alphas = 0.5 * torch.log(1 - betas).cumsum(dim=0)
        N = log_alphas.shape[0]
        t_arr = torch.linspace(0., 1., N + 1)[1:]

        def la(tc):                                            # interpolate_fn (:681-721): piecewise linear, extrapolating
            i = int(torch.searchsorted(t_arr, tc.reshape(1), right=False))
            i0 = 0 if i == 0 else (N - 2 if i == N else i - 1)
            return log_alphas[i0] + (tc - t_arr[i0]) * (log_alphas[i0 + 1] - log_alphas[i0]) / (t_arr[i0 + 1] - t_arr[i0])
        std = lambda tc: torch.sqrt(1. - torch.exp(2. * la(tc)))
        lam = lambda tc: la(tc) - 0.5 * torch.log(1. - torch.exp(2. * la(tc)))
        ts = torch.linspace(1., 1. / N, steps + 1)
        B = x.shape[0]

        def model(xc, tc):                                     # data_prediction_fn: x0 = (x - sigma_t * noise) / alpha_t
            t_in = torch.full((B,), float((tc - 1. / N) * N), device=xc.device, dtype=torch.float32)
            noise = self.denoise_fn(xc, t_in, cond=cond)
            a = torch.exp(la(tc))
            return _lin(1. / a, xc, -(std(tc) / a), noise)

        def update(xc, ms, tp, tc, order, use_corrector):
            lam0 = lam(tp[-1])
            h = lam(tc) - lam0
            rk = (lam(tp[-2]) - lam0) / h if order == 2 else None
            rks = torch.tensor([rk, 1.] if order == 2 else [1.])
            hh = -h
            h_phi_1 = torch.expm1(hh)
            h_phi_k = h_phi_1 / hh - 1
            B_h = torch.expm1(hh)                              # variant 'bh2'
            R, bvec, factorial_i = [], [], 1
            for i in range(1, order + 1):
                R.append(torch.pow(rks, i - 1))
                bvec.append((h_phi_k * factorial_i / B_h).reshape(1))
                factorial_i *= (i + 1)
                h_phi_k = h_phi_k / hh - 1 / factorial_i
            alpha_t = torch.exp(la(tc))
            c_x = std(tc) / std(tp[-1])                        # x_t_ = c_x * x + c_m * m0
            c_m = -(alpha_t * h_phi_1)
            g = -(alpha_t * B_h)                               # x_t = x_t_ + g * (sum_k rho_k * D1_k)
            # predictor: order 2 uses rho_p = 0.5 on D1 = (m1 - m0) / rk; order 1 has no correction term
            if order == 2:
                kp = g * 0.5 / rk
                x_t = _lin(1.0, _lin(c_x, xc, c_m - kp, ms[-1]), kp, ms[-2])
            else:
                x_t = _lin(c_x, xc, c_m, ms[-1])
            if not use_corrector:
                return x_t, None
            rhos_c = torch.tensor([0.5]) if order == 1 else torch.linalg.solve(torch.stack(R), torch.cat(bvec))
            model_t = model(x_t, tc)
            kc_t = g * rhos_c[-1]                              # on D1_t = model_t - m0
            if order == 2:
                kc = g * rhos_c[0] / rk                        # on (m1 - m0)
                x_c = _lin(1.0, _lin(c_x, xc, c_m - kc - kc_t, ms[-1]), kc, ms[-2])
            else:
                x_c = _lin(c_x, xc, c_m - kc_t, ms[-1])
            return _lin(1.0, x_c, kc_t, model_t), model_t

        tp, ms = [ts[0]], [model(x, ts[0])]
        x, m = update(x, ms, tp, ts[1], 1, True)
        tp.append(ts[1])
        ms.append(m)
        for step in range(2, steps + 1):
            x, m = update(x, ms, tp, ts[step], min(2, steps + 1 - step), step != steps)
            tp = [tp[1], ts[step]]
            ms = [ms[1], m]
        return x

    def _host_arr(self, name):
        self._h(name, 0)
        return self._host[1][name]

    def q_sample(self, x_start, t, noise=None):
        """Reference :201-206; t python int."""
        if noise is None:
            noise = torch.randn(x_start.shape, device=x_start.device)
        return _lin(self._h("sqrt_alphas_cumprod", t), x_start, self._h("sqrt_one_minus_alphas_cumprod", t), noise)

    def p_losses(self, x_start, t, cond, noise=None, loss_type="l2"):
        """Reference :210-223: x_start [B,1,M,T], t [B] long (per-item steps), cond [B,H,T] -> scalar loss with a HIP
        backward path into the denoiser and `cond`."""
        B, _, M, T = x_start.shape
        if noise is None:
            noise = torch.randn(x_start.shape, device=x_start.device)
        x0, nz = x_start.reshape(B, M, T).float().contiguous(), noise.reshape(B, M, T).float().contiguous()
        a = self.sqrt_alphas_cumprod[t].view(B, 1, 1)                        # extract(): per-item coefficient gathers
        s = self.sqrt_one_minus_alphas_cumprod[t].view(B, 1, 1)
        x_noisy = S.ew(S.EW_ADD, S.ew_bct(S.EW_MUL, x0, a), S.ew_bct(S.EW_MUL, nz, s), alpha=1.0, beta=1.0)   # q_sample :201-206
        x_recon = self.denoise_fn(x_noisy.view(B, 1, M, T), t, cond)
        xr = x_recon.reshape(B, M, T)
        if loss_type == "l1":
            return A.sum_abs_diff(nz, xr) / nz.numel()
        if loss_type == "l2":
            return A.sum_sq_diff(nz, xr) / nz.numel()
        raise NotImplementedError()

    def forward(self, condition, gt_spec=None, infer=True, infer_speedup=10, method="dpm-solver", k_step=300, use_tqdm=True,
                noise=None):
        """Reference :222-390.  condition [B,T,H] (the reference's layout); returns mel [B,T,M] (infer) or the training
        loss (infer=False).  `noise`: optional dict(x_T [B,1,M,T], steps = list of per-step noises for the ancestral
        sampler; for training: t [B] long, noise [B,1,M,T])."""
        if not infer:
            noise = noise or {}
            cond = A._c(condition.transpose(1, 2).float())
            b, device = cond.shape[0], cond.device
            with torch.no_grad():
                spec = self.norm_spec(gt_spec.float())
                t = noise["t"] if "t" in noise else torch.randint(0, self.k_step, (b,), device=device).long()
                norm_spec = spec.transpose(1, 2)[:, None, :, :].contiguous()
            return self.p_losses(norm_spec, t, cond=cond, noise=noise.get("noise"))
        with torch.no_grad():
            return self._sample(condition, gt_spec, infer_speedup, method, k_step, noise)

    def _sample(self, condition, gt_spec, infer_speedup, method, k_step, noise):
        cond = condition.transpose(1, 2).float().contiguous()
        b, device = cond.shape[0], cond.device
        shape = (b, 1, self.out_dims, cond.shape[2])
        noise = noise or {}
        if gt_spec is None:
            t = self.k_step
            x = noise["x_T"] if "x_T" in noise else torch.randn(shape, device=device)
        else:
            t = k_step
            norm_spec = self.norm_spec(gt_spec).transpose(1, 2)[:, None, :, :].contiguous()
            x = self.q_sample(norm_spec, t - 1, noise=noise.get("x_T"))
        if method is not None and infer_speedup > 1:
            if method in ("dpm-solver", "dpm-solver++"):
                x = self._sample_dpm_solver(x, cond, t, t // infer_speedup, plus=(method == "dpm-solver++"))
            elif method == "unipc":
                x = self._sample_unipc(x, cond, t, t // infer_speedup)
            elif method == "pndm":
                self.noise_list = deque(maxlen=4)
                for i in reversed(range(0, t, infer_speedup)):
                    x = self.p_sample_plms(x, i, infer_speedup, cond=cond)
            elif method == "ddim":
                for i in reversed(range(0, t, infer_speedup)):
                    x = self.p_sample_ddim(x, i, infer_speedup, cond=cond)
            else:
                raise NotImplementedError(method)
        else:
            steps = noise.get("steps")
            for n, i in enumerate(reversed(range(0, t))):
                x = self.p_sample(x, i, cond, noise=None if steps is None else steps[n])
        x = x.squeeze(1).transpose(1, 2)
        return self.denorm_spec(x)

    def norm_spec(self, x):
        return (x - self.spec_min) / (self.spec_max - self.spec_min) * 2 - 1

    def denorm_spec(self, x):
        return (x + 1) / 2 * (self.spec_max - self.spec_min) + self.spec_min
