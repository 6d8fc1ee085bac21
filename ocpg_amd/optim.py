"""AdamW + gradient clipping as three HIP launches (csrc/adamw.hip) behind torch.optim.AdamW's own state layout.

Reference: main.py:76-99 builds a four-group torch.optim.AdamW, engine.py:100-106 clips the total gradient norm and steps.  `ClipAdamW`
IS a torch.optim.AdamW (same param_groups, same `state[p] = {step, exp_avg, exp_avg_sq}`, so `state_dict()` / `load_state_dict()` and LR
schedulers work unchanged and checkpoints stay interchangeable -- SURVEY section 8 row f2); only `step_clip(max_norm)` is new: total
norm, clip coefficient and the update of every parameter without rewriting the gradients (torch: ~10 foreach launches for the norm, one
pass multiplying every gradient, 12 multi-tensor AdamW launches)."""
import torch

from ._lib import check, lib

_CHUNK = 2048


class ClipAdamW(torch.optim.AdamW):
    # torch.amp.GradScaler.step(): "this optimizer handles grad_scale / found_inf itself" -- the scaler then hands both over as device
    # tensors (optimizer.grad_scale / optimizer.found_inf) and does NOT read found_inf back to the host (torch/amp/grad_scaler.py
    # _maybe_opt_step is the .item() this avoids).
    _step_supports_amp_scaling = True

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, foreach=False, fused=False)
        self._tables = None
        self._steps = 0             # steps taken through step_clip since the state was created / loaded (mirrors state[p]["step"])
        self.grad_norm = None       # pre-clip total norm of the last step's (unscaled) gradients, device scalar
        self._amp = None            # fp32[6] on the device once a step ran under a GradScaler: [3] is then THE step count (skipped steps excluded)

    # ---- tables -----------------------------------------------------------------------------------------------------------------
    def _entries(self):
        ent = []
        for gi, group in enumerate(self.param_groups):
            if group.get("amsgrad") or group.get("maximize"):
                raise RuntimeError("ClipAdamW: amsgrad / maximize are not implemented")
            for p in group["params"]:
                if p.grad is None:
                    continue
                ent.append((p, gi))
        return ent

    def _dense_like(self, g, p):
        return g.dtype == torch.float32 and g.device == p.device and tuple(g.shape) == tuple(p.shape) and \
            all(a == b or n == 1 for a, b, n in zip(g.stride(), p.stride(), p.shape))

    def _key(self, ent):
        """Every address the kernels write through: a re-assigned p.data / .grad / moment (memory-format or device moves, a DDP
        rebuild, load_state_dict) must rebuild the tables -- a stale table would be a write into freed memory."""
        st = self.state
        return tuple((p.data_ptr(), p.grad.data_ptr(), st[p]["exp_avg"].data_ptr() if len(st[p]) else 0,
                      st[p]["exp_avg_sq"].data_ptr() if len(st[p]) else 0) for p, _ in ent)

    def _build(self, ent):
        dev = ent[0][0].device
        for p, _ in ent:
            if p.dtype != torch.float32 or not p.is_cuda:
                raise RuntimeError("ClipAdamW serves fp32 GPU parameters")
            st = self.state[p]
            if len(st) == 0:
                st["step"] = torch.tensor(0.0)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        numels = [p.numel() for p, _ in ent]
        prefix = [0]
        for n in numels:
            prefix.append(prefix[-1] + (n + _CHUNK - 1) // _CHUNK)
        t = {"n": len(ent), "chunks": prefix[-1], "params": [p for p, _ in ent], "groups": [gi for _, gi in ent],
             "key": self._key(ent),
             # (through pinned memory: a pageable upload would stall the host until the stream drains)
             "meta": torch.tensor([[p.data_ptr() for p, _ in ent], [p.grad.data_ptr() for p, _ in ent],
                                   [self.state[p]["exp_avg"].data_ptr() for p, _ in ent], [self.state[p]["exp_avg_sq"].data_ptr() for p, _ in ent],
                                   numels, prefix[:-1]], dtype=torch.int64).pin_memory().to(dev, non_blocking=True),
             "partials": torch.empty(max(prefix[-1], 1), dtype=torch.float32, device=dev),
             "norm": torch.zeros(2, dtype=torch.float32, device=dev), "hyper_key": None, "hyper": None}
        return t

    def _hyper(self, t):
        key = tuple((g["lr"], g["weight_decay"]) for g in self.param_groups)
        if t["hyper_key"] != key:                  # the LR scheduler changed a group: refresh the per-tensor arrays
            t["hyper"] = torch.tensor([[self.param_groups[gi]["lr"] for gi in t["groups"]],
                                       [self.param_groups[gi]["weight_decay"] for gi in t["groups"]]], dtype=torch.float32).to(t["meta"].device)
            t["hyper_key"] = key
        return t["hyper"]

    # ---- the step ---------------------------------------------------------------------------------------------------------------
    @torch.no_grad()
    def step_clip(self, max_norm=0.0, grad_scale=None, found_inf=None, amp=False):
        """clip_grad_norm_(all parameters of the groups, max_norm) + step().  Returns the total gradient norm BEFORE clipping (device
        scalar), or None when no parameter has a gradient.

        amp (or a grad_scale / found_inf tensor): the torch.amp.GradScaler form of engine.py:98-106 -- the gradients still carry
        `grad_scale` (device scalar, None = already unscaled); the norm is that of the unscaled gradients, the step is skipped ON THE
        DEVICE when the norm is not finite or `found_inf` is set, and skipped steps do not count (include/ocpg_hip.h,
        ocpg_grad_norm_clip_amp)."""
        amp = amp or grad_scale is not None or found_inf is not None
        ent = self._entries()
        if not ent:
            return None
        for p, _ in ent:
            if not self._dense_like(p.grad, p):    # (does not happen on this path: gradients are born with their parameter's strides)
                p.grad = torch.empty_like(p).copy_(p.grad)
        t = self._tables
        if t is None or len(t["params"]) != len(ent) or any(a is not b for a, (b, _) in zip(t["params"], ent)) or t["key"] != self._key(ent):
            t = self._tables = self._build(ent)
        hyper = self._hyper(t)
        g0 = self.param_groups[0]
        b1, b2 = g0["betas"]
        for g in self.param_groups:
            if g["betas"] != g0["betas"] or g["eps"] != g0["eps"]:
                raise RuntimeError("ClipAdamW: betas / eps must be the same in every group")
        st = torch.cuda.current_stream().cuda_stream
        m = t["meta"]
        if amp:
            if self._amp is None:
                self._amp = torch.zeros(6, dtype=torch.float32, device=m.device)
                self._amp[3] = float(self._steps or int(self.state[t["params"][0]]["step"]))
            for x in (grad_scale, found_inf):
                if x is not None and (x.dtype != torch.float32 or x.device != m.device or x.numel() != 1):
                    raise RuntimeError("ClipAdamW: grad_scale / found_inf must be one-element fp32 tensors on the parameters' device")
            with torch.cuda.device(m.device):
                check(lib().ocpg_grad_norm_clip_amp(m[1].data_ptr(), m[4].data_ptr(), m[5].data_ptr(), t["n"], t["chunks"], float(max_norm),
                                                    t["partials"].data_ptr(), grad_scale.data_ptr() if grad_scale is not None else None,
                                                    found_inf.data_ptr() if found_inf is not None else None, float(b1), float(b2),
                                                    self._amp.data_ptr(), st), "ocpg_grad_norm_clip_amp")
                check(lib().ocpg_adamw_step_amp(m[0].data_ptr(), m[1].data_ptr(), m[2].data_ptr(), m[3].data_ptr(), m[4].data_ptr(), m[5].data_ptr(),
                                                hyper[0].data_ptr(), hyper[1].data_ptr(), t["n"], t["chunks"], self._amp.data_ptr(),
                                                float(b1), float(b2), float(g0["eps"]), st), "ocpg_adamw_step_amp")
            self.grad_norm = self._amp[0]
            return self.grad_norm
        if self._amp is not None:                  # back from scaler-driven steps: the device holds the count
            self._steps, self._amp = int(self._amp[3]), None
        with torch.cuda.device(m.device):
            check(lib().ocpg_grad_norm_clip(m[1].data_ptr(), m[4].data_ptr(), m[5].data_ptr(), t["n"], t["chunks"], float(max_norm),
                                            t["partials"].data_ptr(), t["norm"].data_ptr(), st), "ocpg_grad_norm_clip")
            self._steps = int(self.state[t["params"][0]]["step"]) + 1 if self._steps == 0 else self._steps + 1
            check(lib().ocpg_adamw_step(m[0].data_ptr(), m[1].data_ptr(), m[2].data_ptr(), m[3].data_ptr(), m[4].data_ptr(), m[5].data_ptr(),
                                        hyper[0].data_ptr(), hyper[1].data_ptr(), t["n"], t["chunks"], t["norm"].data_ptr(),
                                        float(b1), float(b2), float(g0["eps"]), self._steps, st), "ocpg_adamw_step")
        self.grad_norm = t["norm"][0]
        return self.grad_norm

    def step(self, closure=None, max_norm=0.0):
        """torch.optim.AdamW.step semantics through the same kernels.  Under `scaler.step(optimizer, max_norm=...)` the scaler has set
        .grad_scale / .found_inf (see _step_supports_amp_scaling): unscale + clip + step in the three launches, no host sync."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        gs, fi = getattr(self, "grad_scale", None), getattr(self, "found_inf", None)
        self.step_clip(max_norm, grad_scale=gs, found_inf=fi)
        return loss

    def steps_taken(self):
        """Optimizer steps actually applied (steps a GradScaler skipped do not count).  Reads the device counter: a host sync."""
        return int(self._amp[3]) if self._amp is not None else self._steps

    def _sync_steps(self):
        """state[p]["step"] of the parameters the kernels step (the tables' members) = the count the kernels were given; parameters
        that never received a gradient keep their own (torch.optim.AdamW does not step them either)."""
        steps = self.steps_taken()
        if steps and self._tables is not None:
            for p in self._tables["params"]:
                st = self.state.get(p)
                if st is not None and "step" in st:
                    st["step"] = torch.tensor(float(steps))

    def state_dict(self):
        self._sync_steps()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._tables, self._steps, self._amp = None, 0, None
