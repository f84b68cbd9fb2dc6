"""The denoise loop of ip_adapter/custom_pipelines.py:249-363 as a device-resident loop.

One denoise step = [UNet forward on the CFG-duplicated latent] + [CFG combine + scheduler.step] +
[step counter += 1], recorded ONCE into a C++ plan and captured into a hipGraph; the 30-step loop is
30 graph replays with no host work in between: timesteps, scheduler coefficients, the Euler input
scale and the per-step IP-scale gate (control_guidance_start/end, custom_pipelines.py:326-329) are
device tables indexed by a device-resident step counter.
"""
import hashlib
import os

import torch

from . import lib as L
from .ctx import Ctx
from .unet import StepState

# XCD cell policy measured once per (device, UNet batch, latent size, dtype) and process: every engine (and fork) of that shape reuses it
_XCD_PICK = {}


class DenoiseEngine:
    def __init__(self, unet, device, dtype=torch.bfloat16, use_graph=True):
        self.unet = unet
        self.device = torch.device(device)
        self.dtype = dtype
        self.use_graph = use_graph
        self.eager = Ctx(self.device, dtype)
        self.st = None
        self.plan = None
        self.key = None
        self.noise_pred = None
        # XCD cell shapes tried when the plan is first recorded (_pick_xcd_cells): the byte-count model, 4 x 2 and 8 x 1 cells, and
        # 4 x 2 with the GEGLU launches left to the model (13); the choice sticks for the life of the engine
        # (IMH_XCD_AUTOTUNE=0: model only)
        self.xcd_candidates = (0, 3, 2, 13) if os.environ.get("IMH_XCD_AUTOTUNE", "1") != "0" else (0,)
        self.xcd_cells = None
        self._is_fork = False
        self._plans = {}         # recorded plans by schedule key (two-stage PNS alternates a preview and a final schedule per image)
        self.max_cached_plans = int(os.environ.get("IMH_MAX_CACHED_PLANS", "3"))      # each pins ~2 GB of activation buffers at 1024^2
        self._sched_key = None

    # -- conditioning (once per image / per PNS run; shared by every candidate seed) --
    @torch.no_grad()
    def set_conditioning(self, prompt_embeds, negative_prompt_embeds, pooled, negative_pooled, height, width,
                         guidance_scale=5.0, guidance_rescale=0.0, original_size=None, crops_coords_top_left=(0, 0),
                         target_size=None, cfg_role=None):
        """prompt_embeds: [S, 77+T, 2048] (IP tokens already concatenated, ip_adapter.py:321-322).
        original_size / crops_coords_top_left / target_size: SDXL micro-conditioning (custom_pipelines.py:277-293),
        default (height, width), (0, 0), (height, width).
        cfg_role = 0 / 1: this engine computes only the unconditional / the conditional half of the CFG pair (UNet batch S
        instead of 2S); the halves are exchanged every step by denoise_cfg_split (two ranks share one candidate's denoise:
        the serial tail of the two-stage PNS schedule)."""
        if cfg_role not in (None, 0, 1):
            raise ValueError("cfg_role must be None, 0 (unconditional half) or 1 (conditional half)")
        if cfg_role != getattr(self, "cfg_role", None):
            self.plan = None
        self._plans = {}                          # every recorded plan points into the previous conditioning's caches
        self.cfg_role = cfg_role
        self.do_cfg = guidance_scale > 1.0                                   # custom_pipelines.py:223
        self.guidance = float(guidance_scale)
        if float(guidance_rescale) != getattr(self, "guidance_rescale", 0.0):
            self.plan = None
        self.guidance_rescale = float(guidance_rescale)                      # :351-354 (only with CFG)
        S = prompt_embeds.shape[0]
        osz, tsz = tuple(original_size or (height, width)), tuple(target_size or (height, width))
        ids = torch.tensor([list(osz) + list(crops_coords_top_left) + list(tsz)], dtype=torch.float32).repeat(S, 1)   # :277-293
        if self.do_cfg and cfg_role is None:                                 # :295-298 -- order [uncond | cond]
            ehs = torch.cat([negative_prompt_embeds, prompt_embeds], 0)
            text = torch.cat([negative_pooled, pooled], 0)
            ids = torch.cat([ids, ids], 0)
        elif self.do_cfg and cfg_role == 0:
            ehs, text = negative_prompt_embeds, negative_pooled
        else:
            ehs, text = prompt_embeds, pooled
        ctx = Ctx(self.device, self.dtype)      # fresh context: the K/V caches must outlive pooled buffers
        st = self.unet.prepare_conditioning(ctx, ehs, text, ids)
        self._cond_ctx = ctx
        self.S, self.H, self.W = S, height // 8, width // 8
        self.T_total = ehs.shape[1]
        old = self.st
        self.st = st
        if old is not None:                      # keep per-run tables
            for k in ("latents", "t_table", "step", "in_scale_tab", "ip_scale_tab", "coef_tab"):
                setattr(st, k, getattr(old, k, None))
        self.plan = None                         # conditioning buffers changed -> re-record
        return st

    # -- schedule tables --
    def set_schedule(self, scheduler, num_inference_steps, control_guidance_start=0.0, control_guidance_end=1.0,
                     denoising_end=None):
        st, dev = self.st, self.device
        from .attention_processor import IPAttnProcessor2_0
        base = next((p.scale for p in self.unet.attn_processors.values() if isinstance(p, IPAttnProcessor2_0)), 1.0)
        # the key carries a fingerprint of the tables themselves (timesteps, coefficients, input scale, init sigma): a scheduler instance
        # configured differently (betas, spacing, prediction type) under the same class name must not reuse another schedule's plan
        scheduler.set_timesteps(num_inference_steps)
        tab = scheduler.tables()
        fp = hashlib.sha1()
        for k in ("timesteps", "coef", "in_scale"):
            v = tab.get(k)
            fp.update(b"-" if v is None else v.detach().to("cpu", torch.float64).contiguous().numpy().tobytes())
        fp.update(repr(float(tab["init_noise_sigma"])).encode())
        key = (type(scheduler).__name__, int(getattr(scheduler, "num_train_timesteps", 1000)), int(num_inference_steps), float(control_guidance_start), float(control_guidance_end), denoising_end, float(base), fp.hexdigest())
        hit = self._plans.get(key)
        if hit is not None:
            # a schedule this engine has run under this conditioning: its tables, time-embedding rows and recorded plan are still there
            # (the plan's launches point at them), so a preview / final alternation re-records nothing
            for k in ("t_table", "coef_tab", "in_scale_tab", "ip_scale_tab", "temb_table"):
                setattr(st, k, hit["st"][k])
            self.steps, self.init_noise_sigma = hit["steps"], hit["init_noise_sigma"]
            self.plan, self.noise_pred, self._temb_ctx = hit["plan"], hit["noise_pred"], hit["temb_ctx"]
            self.plan_tail, self.np_full = hit.get("plan_tail"), hit.get("np_full")
            self._sched_key = key
            return
        n = num_inference_steps
        if denoising_end is not None and isinstance(denoising_end, float) and 0 < denoising_end < 1:
            # custom_pipelines.py:303-311: stop once t falls below the cut-off; the gating window below then counts
            # the truncated list, as upstream does
            cutoff = int(round(1000 - denoising_end * 1000))
            n = int((tab["timesteps"] >= cutoff).sum().item())
        st.t_table = tab["timesteps"].to(dev)
        st.coef_tab = tab["coef"].contiguous().to(dev)
        st.in_scale_tab = tab["in_scale"].to(dev) if tab["in_scale"] is not None else None
        gate = [0.0 if (i / n < control_guidance_start) or ((i + 1) / n > control_guidance_end) else float(base)
                for i in range(n)]                                           # custom_pipelines.py:319-329
        st.ip_scale_tab = torch.tensor(gate, dtype=torch.float32, device=dev)
        if st.step is None:
            st.step = torch.zeros(1, dtype=torch.int32, device=dev)
        self.steps = n
        self.init_noise_sigma = float(tab["init_noise_sigma"])
        self.plan = None                         # table pointers changed
        self._sched_key = key

    def fork(self):
        """A second engine on the SAME weights and the SAME conditioning (K/V caches, aug_emb) with its own latents,
        step counter, activation buffers and plan: lets several PNS candidates be in flight on one GPU (one HIP
        stream each), so that kernels of independent candidates fill the CUs a batch-1 kernel leaves idle."""
        e = DenoiseEngine(self.unet, self.device, self.dtype, self.use_graph)
        e._is_fork = True                        # never tunes: it may record while its parent is running on another stream
        for k in ("do_cfg", "guidance", "guidance_rescale", "S", "H", "W", "T_total", "steps", "init_noise_sigma", "_cond_ctx", "cfg_role",
                  "xcd_candidates", "xcd_cells"):
            setattr(e, k, getattr(self, k))
        st = StepState()
        src = self.st
        st.aug_emb, st.kv = src.aug_emb, src.kv                      # shared, read-only during denoising
        st.t_table, st.coef_tab, st.in_scale_tab, st.ip_scale_tab = src.t_table, src.coef_tab, src.in_scale_tab, src.ip_scale_tab
        st.step = torch.zeros(1, dtype=torch.int32, device=self.device)
        e.st = st
        return e

    def _record(self):
        st = self.st
        if st.latents is None or tuple(st.latents.shape) != (self.S, 4, self.H, self.W):
            st.latents = torch.zeros(self.S, 4, self.H, self.W, dtype=torch.float32, device=self.device)
        # the time-embedding rows of every step of this schedule under this conditioning: once here, not five launches per step
        self._temb_ctx = Ctx(self.device, self.dtype)          # (its pool owns the table for the life of the plan)
        self.unet.precompute_temb(self._temb_ctx, st, st.t_table)
        split = self.do_cfg and getattr(self, "cfg_role", None) is not None
        if not split and self.use_graph and len(self.xcd_candidates) > 1 and self.xcd_cells is None:
            pk = (self.device.index or 0, self.S, self.H, self.W, str(self.dtype), bool(self.do_cfg))
            if pk not in _XCD_PICK and not self._is_fork:
                _XCD_PICK[pk] = self._pick_xcd_cells()
            self.xcd_cells = _XCD_PICK[pk][0] if pk in _XCD_PICK else 0
            if pk in _XCD_PICK:
                self.xcd_times_ms = _XCD_PICK[pk][1]
        rec = Ctx(self.device, self.dtype, record=True)
        rec.xcd_cells = self.xcd_cells or 0
        out = self.unet.emit_forward(rec, st, self.S, self.H, self.W, cfg_dup=self.do_cfg and not split)
        if split:
            # this rank's half of the noise prediction ends the forward plan; the CFG combine + scheduler step + step counter are a
            # second tiny plan over BOTH halves ([uncond | cond] = the layout the fused step reads), run after the per-step exchange
            if self.use_graph:
                rec.capture()
            self.plan, self.noise_pred = rec, out
            self.np_full = torch.empty((2,) + tuple(out.shape), dtype=out.dtype, device=self.device)
            tail = Ctx(self.device, self.dtype, record=True)
            fac = None
            if getattr(self, "guidance_rescale", 0.0) > 0.0:
                fac = tail.new(self.S, dtype=torch.float32)
                tail.ew(L.EW_CFG_RESCALE, fac, a=self.np_full, i=(self.S, self.H * self.W, 0, 0, 0, 0),
                        f=(0.0, 0.0, self.guidance, self.guidance_rescale), descr="cfg.rescale")
            tail.ew(L.EW_CFG_STEP, st.latents, a=self.np_full, w=fac, tab=st.coef_tab, step=st.step,
                    i=(self.S, self.H * self.W, 0, 1, 0, 0), f=(0.0, 0.0, self.guidance, 0.0), descr="cfg+step")
            tail.ew(L.EW_STEP_SET, st.step, i=(0, 0, 0, 0, 0, 0), descr="step++")
            if self.use_graph:
                tail.capture()
            self.plan_tail = tail
            self._remember_plan()
            return
        rec.tag = 70
        fac = None
        if self.do_cfg and getattr(self, "guidance_rescale", 0.0) > 0.0:     # rescale_noise_cfg, custom_pipelines.py:351-354
            fac = rec.new(self.S, dtype=torch.float32)
            rec.ew(L.EW_CFG_RESCALE, fac, a=out, i=(self.S, self.H * self.W, 0, 0, 0, 0),
                   f=(0.0, 0.0, self.guidance, self.guidance_rescale), descr="cfg.rescale")
        rec.ew(L.EW_CFG_STEP, st.latents, a=out, w=fac, tab=st.coef_tab, step=st.step,
               i=(self.S, self.H * self.W, 0, int(self.do_cfg), 0, 0), f=(0.0, 0.0, self.guidance, 0.0), descr="cfg+step")
        rec.ew(L.EW_STEP_SET, st.step, i=(0, 0, 0, 0, 0, 0), descr="step++")
        if self.use_graph:
            rec.capture()
        self.plan = rec
        self.noise_pred = out
        self._remember_plan()

    def _remember_plan(self):
        if self._sched_key is None:
            return
        st = self.st
        self._plans[self._sched_key] = dict(
            st={k: getattr(st, k, None) for k in ("t_table", "coef_tab", "in_scale_tab", "ip_scale_tab", "temb_table")},
            steps=self.steps, init_noise_sigma=self.init_noise_sigma, plan=self.plan, noise_pred=self.noise_pred,
            temb_ctx=self._temb_ctx, plan_tail=getattr(self, "plan_tail", None), np_full=getattr(self, "np_full", None))
        while len(self._plans) > self.max_cached_plans:              # (each plan keeps ~2 GB of activation buffers alive at 1024^2)
            self._plans.pop(next(iter(self._plans)))

    def _pick_xcd_cells(self):
        """Which XCD cell shape this box prefers for the GEMM / conv launches of the forward (imh_gemm_args.xcd): the byte-count model
        (0), or 4 x 2 (3) / 8 x 1 (2) cells over M x N everywhere.  Bit-identical results either way; the faster one differs from box to box
        (fresh MI355X boxes, same build: the model is 0.3 ms per forward ahead on a 19.9-ms box and 0.4 ms behind on 21.5-ms boxes,
        profiles/r04_forward_ab_xcd_cells*.json), so it is measured once per engine: each candidate's forward is recorded,
        captured and replayed a few times; the winner's plan is then recorded for real by _record()."""
        st = self.st
        times = {}
        for cells in self.xcd_candidates:
            rec = Ctx(self.device, self.dtype, record=True)
            rec.xcd_cells = cells
            self.unet.emit_forward(rec, st, self.S, self.H, self.W, cfg_dup=self.do_cfg)
            rec.capture()
            ts = []
            for i in range(11):
                self.eager.ew(L.EW_STEP_SET, st.step, i=(0, 1, 0, 0, 0, 0), descr="step=0")
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); rec.replay(); e1.record()
                torch.cuda.synchronize(self.device)
                if i:                                      # the first replay is the warm-up
                    ts.append(e0.elapsed_time(e1))
            ts.sort()
            times[cells] = ts[len(ts) // 2]                # median of ten: the candidates are 1-2 % apart
            del rec
        self.xcd_times_ms = times
        return min(times, key=times.get), times

    @torch.no_grad()
    def denoise_cfg_split(self, latents, exchange):
        """One candidate's denoise shared by TWO ranks (engines with cfg_role 0 and 1 on the same conditioning and noise): per step
        each runs the UNet on its half of the CFG pair, `exchange(mine [S*HW*4...]) -> (uncond, cond)` swaps the halves (pns.
        pair_exchange: one all_gather of [S, HW, 4] values over xGMI), and both apply the identical combine + scheduler step, so
        the latents stay bit-equal on the two ranks without further traffic.  The two-stage PNS tail (assets/1.png: the judged-best
        noise x the full denoise) is then `steps` batch-S forwards deep instead of batch-2S ones."""
        if getattr(self, "cfg_role", None) is None or not self.do_cfg:
            raise L.ImhError("denoise_cfg_split needs an engine whose conditioning was set with cfg_role = 0 / 1 and guidance > 1")
        if self.plan is None:
            self._record()
        st = self.st
        st.latents.copy_(latents.to(self.device, torch.float32) * self.init_noise_sigma)
        self.eager.ew(L.EW_STEP_SET, st.step, i=(0, 1, 0, 0, 0, 0), descr="step=0")
        for _ in range(self.steps):
            self.plan.replay()
            un, co = exchange(self.noise_pred)
            self.np_full[0].copy_(un); self.np_full[1].copy_(co)
            self.plan_tail.replay()
        return st.latents

    @torch.no_grad()
    def denoise(self, latents, callback=None, callback_steps=1):
        """latents: [S, 4, H/8, W/8] unit-variance noise (CPU or device).  Returns final fp32 latents
        (output_type='latent' of custom_pipelines.py:365-379).  callback(i, t, latents) every ``callback_steps``
        steps (:359-363) is the only thing that makes the host wait inside the loop."""
        if getattr(self, "cfg_role", None) is not None and self.do_cfg:
            raise L.ImhError("this engine holds one half of the CFG pair (cfg_role): use denoise_cfg_split")
        if self.plan is None:
            self._record()
        st = self.st
        st.latents.copy_(latents.to(self.device, torch.float32) * self.init_noise_sigma)     # prepare_latents :255-265
        self.eager.ew(L.EW_STEP_SET, st.step, i=(0, 1, 0, 0, 0, 0), descr="step=0")
        for i in range(self.steps):                                         # :325 -- no host work per step
            self.plan.replay()
            if callback is not None and i % callback_steps == 0:
                torch.cuda.current_stream(self.device).synchronize()
                callback(i, st.t_table[i].item(), st.latents)
        return st.latents
