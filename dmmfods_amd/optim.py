"""Flat fused Adam over the model's parameter arena (reference: torch.optim.Adam built at
agents/Dense_U_Net_lidar_Agent.py:57-61 and stepped at :265).  One kernel launch updates every parameter;
state_dict()/load_state_dict() use torch.optim.Adam's layout so reference checkpoints round-trip."""
import torch

from . import _lib


class FusedAdam:
    def __init__(self, model, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0, amsgrad=False):
        if amsgrad:
            raise ValueError("amsgrad=True is not supported (reference default False, H:156)")
        self.model = model
        self.defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, amsgrad=False)
        self.param_groups = [dict(self.defaults, params=list(range(sum(1 for _ in model.parameters()))))]
        self.step_count = 0
        self._alloc()

    def _alloc(self):
        p = self.model.param_arena
        self.exp_avg = torch.zeros_like(p)
        self.exp_avg_sq = torch.zeros_like(p)

    def zero_grad(self, set_to_none=False):
        """The HIP backward overwrites the gradient arena, so there is nothing to clear (kept for API parity)."""

    @torch.no_grad()
    def step(self, grad_scale=1.0):
        m = self.model
        if self.exp_avg.device != m.param_arena.device:
            self.exp_avg = self.exp_avg.to(m.param_arena.device)
            self.exp_avg_sq = self.exp_avg_sq.to(m.param_arena.device)
        g = self.param_groups[0]
        self.step_count += 1
        _lib.check(_lib.lib().dmm_adam_step(
            m.param_arena.data_ptr(), m.grad_arena.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
            m.param_arena.numel(), float(g["lr"]), float(g["betas"][0]), float(g["betas"][1]), float(g["eps"]),
            float(g["weight_decay"]), self.step_count, float(grad_scale), _lib.stream_ptr()))

    # ---- torch.optim.Adam-compatible checkpoint format ----
    def state_dict(self):
        state, off = {}, 0
        for i, p in enumerate(self.model.parameters()):
            n = p.numel()
            if self.step_count > 0:
                state[i] = {"step": torch.tensor(float(self.step_count)),
                            "exp_avg": self.exp_avg[off:off + n].view(p.shape).clone(),
                            "exp_avg_sq": self.exp_avg_sq[off:off + n].view(p.shape).clone()}
            off += n
        g = dict(self.param_groups[0])
        g.update(maximize=False, foreach=None, capturable=False, differentiable=False, fused=None)
        return {"state": state, "param_groups": [g]}

    def load_state_dict(self, sd):
        off = 0
        steps = [int(float(s["step"])) for s in sd["state"].values()] or [0]
        self.step_count = max(steps)
        for i, p in enumerate(self.model.parameters()):
            n = p.numel()
            s = sd["state"].get(i)
            if s is not None:
                self.exp_avg[off:off + n].copy_(s["exp_avg"].reshape(-1))
                self.exp_avg_sq[off:off + n].copy_(s["exp_avg_sq"].reshape(-1))
            off += n
        for k in ("lr", "betas", "eps", "weight_decay"):
            if k in sd["param_groups"][0]:
                self.param_groups[0][k] = sd["param_groups"][0][k]
