"""AdamW for the flows' many small parameter tensors: torch.optim.AdamW's update in ceil(n / 72) launches.

The reference trains with `optim.AdamW(filter(requires_grad, model.parameters()), lr=...)` (model.py:289) and calls
`optimizer.step()` after every backward (experiment_cl.py:136, experiment_ad.py:213).  A flow has 135 (cifar10) ... 571 (smap)
parameter tensors of 9 ... 147 K elements; torch's fused multi-tensor kernel spends 4 ... 16 launches of 15 - 43 us on them, 12 - 15 % of
the captured training step at the reference's batch of 256.  `FusedAdamW` is a drop-in `torch.optim.Optimizer` whose `step()`
is one `cf_adamw_step_batch` call per parameter group (the tensor table travels in the kernel arguments): same arithmetic term by
term (decoupled weight decay, lerp form of the first moment, bias corrections; amsgrad off), same `state_dict` layout
(`step`, `exp_avg`, `exp_avg_sq` per parameter - the moments are views of ONE flat buffer per group), capturable (the update
count lives on the device).  fp32 parameters on one GPU per group; anything else raises.
"""
import ctypes

import torch

from .layers import _hip


class FusedAdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2, maximize=False):
        if not 0.0 <= lr or not 0.0 <= eps or not 0.0 <= betas[0] < 1.0 or not 0.0 <= betas[1] < 1.0 or not 0.0 <= weight_decay:
            raise ValueError("FusedAdamW: invalid hyper-parameters lr=%r betas=%r eps=%r weight_decay=%r" % (lr, betas, eps, weight_decay))
        self._flats = []             # per parameter group: (exp_avg flat, exp_avg_sq flat, step) | None - kept out of param_groups
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay, maximize=bool(maximize)))

    def add_param_group(self, param_group):
        super().add_param_group(param_group)
        self._flats.append(self._init_group(self.param_groups[-1]))

    def _init_group(self, group):
        ps = [p for p in group["params"] if p.requires_grad]
        if not ps:
            return None
        dev = ps[0].device
        for p in ps:
            if p.dtype != torch.float32 or p.device != dev or not p.is_cuda:
                raise RuntimeError("FusedAdamW: fp32 parameters on one GPU per group (got %s on %s)" % (p.dtype, p.device))
            if not p.is_contiguous():
                raise RuntimeError("FusedAdamW: parameters must be contiguous")
        offs, o = [], 0
        for p in ps:
            offs.append(o)
            o += (p.numel() + 3) & ~3                            # 16-byte aligned slots: the kernel moves float4s
        m = torch.zeros(max(o, 1), device=dev, dtype=torch.float32)
        v = torch.zeros_like(m)
        step = torch.zeros(1, device=dev, dtype=torch.float32)
        for p, lo in zip(ps, offs):
            self.state[p] = {"step": step[0], "exp_avg": m[lo:lo + p.numel()].view_as(p), "exp_avg_sq": v[lo:lo + p.numel()].view_as(p)}
        return (m, v, step)

    def load_state_dict(self, state_dict):
        """torch puts the loaded moments into fresh tensors: copy them back into the flat buffers the kernel updates."""
        views = {p: dict(self.state[p]) for g in self.param_groups for p in g["params"] if p in self.state}
        super().load_state_dict(state_dict)
        for g, flat in zip(self.param_groups, self._flats):
            if flat is None:
                continue
            loaded_step = None
            for p in g["params"]:
                if p not in views:
                    continue
                new = self.state.get(p, {})
                with torch.no_grad():
                    for k in ("exp_avg", "exp_avg_sq"):
                        if k in new and new[k] is not views[p][k]:
                            views[p][k].copy_(new[k])
                    if "step" in new:
                        loaded_step = float(new["step"])
                self.state[p] = views[p]
            if loaded_step is not None:
                flat[2].fill_(loaded_step)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group, flat in zip(self.param_groups, self._flats):
            if flat is None:
                continue
            ps = [p for p in group["params"] if p.requires_grad and p.grad is not None]
            if not ps:
                continue
            gs = []
            for p in ps:
                g = p.grad
                if g.is_sparse or g.dtype != torch.float32 or g.device != p.device:
                    raise RuntimeError("FusedAdamW: dense fp32 gradients on the parameter's device")
                gs.append(g if g.is_contiguous() else g.contiguous())
            flat[2].add_(1.0)                                    # the update count of THIS step, on the device (capturable)
            n = len(ps)
            numel = (ctypes.c_int64 * n)(*[p.numel() for p in ps])
            A = _hip.ptr_array
            beta1, beta2 = group["betas"]
            _hip.call("cf_adamw_step_batch", n, A(ps), A(gs), A([self.state[p]["exp_avg"] for p in ps]),
                      A([self.state[p]["exp_avg_sq"] for p in ps]), ctypes.cast(numel, ctypes.c_void_p), _hip.p(flat[2]),
                      float(group["lr"]), float(beta1), float(beta2), float(group["eps"]), float(group["weight_decay"]),
                      int(group["maximize"]), _hip.stream())
        return loss
