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
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, foreach=False, fused=False)
        self._tables = None
        self._steps = 0             # steps taken through step_clip since the state was created / loaded (mirrors state[p]["step"])

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
    def step_clip(self, max_norm=0.0):
        """clip_grad_norm_(all parameters of the groups, max_norm) + step().  Returns the total gradient norm BEFORE clipping (device
        scalar), or None when no parameter has a gradient."""
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
        with torch.cuda.device(m.device):
            check(lib().ocpg_grad_norm_clip(m[1].data_ptr(), m[4].data_ptr(), m[5].data_ptr(), t["n"], t["chunks"], float(max_norm),
                                            t["partials"].data_ptr(), t["norm"].data_ptr(), st), "ocpg_grad_norm_clip")
            self._steps = int(self.state[t["params"][0]]["step"]) + 1 if self._steps == 0 else self._steps + 1
            check(lib().ocpg_adamw_step(m[0].data_ptr(), m[1].data_ptr(), m[2].data_ptr(), m[3].data_ptr(), m[4].data_ptr(), m[5].data_ptr(),
                                        hyper[0].data_ptr(), hyper[1].data_ptr(), t["n"], t["chunks"], t["norm"].data_ptr(),
                                        float(b1), float(b2), float(g0["eps"]), self._steps, st), "ocpg_adamw_step")
        return t["norm"][0]

    def step(self, closure=None):
        """torch.optim.AdamW.step semantics (no clipping) through the same kernels."""
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self.step_clip(0.0)
        return loss

    def _sync_steps(self):
        """state[p]["step"] of the parameters the kernels step (the tables' members) = the count the kernels were given; parameters
        that never received a gradient keep their own (torch.optim.AdamW does not step them either)."""
        if self._steps and self._tables is not None:
            for p in self._tables["params"]:
                st = self.state.get(p)
                if st is not None and "step" in st:
                    st["step"] = torch.tensor(float(self._steps))

    def state_dict(self):
        self._sync_steps()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._tables, self._steps = None, 0
