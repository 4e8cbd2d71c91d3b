"""Adam for the trainer's parameter tensors on the HIP device.

Same update rule, hyper-parameter names, ``param_groups`` and ``state_dict`` layout (``step``, ``exp_avg``,
``exp_avg_sq``) as ``torch.optim.Adam`` without weight decay / amsgrad -- what the reference instantiates at
GaussianPointTrainer.py:126-129 -- so learning-rate schedulers and checkpoints work unchanged.  The step itself is one
streaming HIP kernel per tensor (csrc/gs_optim.hip); there is no CPU path.
"""
from __future__ import annotations

import torch

from ._lib import call, current_stream, ptr


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1):
            raise ValueError("invalid Adam hyper-parameters")
        super().__init__(params, dict(lr=lr, betas=tuple(betas), eps=eps))
        self._scale_regulariser = {}   # id(param) -> (weight, point_invalid_mask)
        self._row_mask = {}            # id(param) -> point_invalid_mask: rows of invalid points are skipped

    def set_row_mask(self, param: torch.Tensor, point_invalid_mask: torch.Tensor) -> None:
        """``param`` is a fixed-capacity [N,3] or [N,56] tensor with one row per point and ``point_invalid_mask`` (int8 [N],
        the live tensor -- it is read at every step) marks the rows without a point: those rows are not touched by the
        step.  Their gradient is zero, so ``torch.optim.Adam`` would only decay their moments and move parameters that
        are overwritten before they are read again; the decay is applied lazily when a row comes back to life
        (``state["last_step"]``), so a row's moments are what ``torch.optim.Adam`` would hold.  With the reference's
        capacity of 10x the initial points (config/tat_truck_every_8_test.yaml:20) most rows are free for most of a
        run."""
        if param.dim() != 2 or param.shape[1] not in (3, 56) or point_invalid_mask.shape != (param.shape[0],):
            raise ValueError("set_row_mask: [N,3] or [N,56] parameter and an [N] mask")
        if point_invalid_mask.dtype != torch.int8:
            raise TypeError("point_invalid_mask must be int8")
        self._row_mask[id(param)] = point_invalid_mask

    def set_scale_regulariser(self, features: torch.Tensor, weight: float, point_invalid_mask: torch.Tensor,
                              local_share=None) -> None:
        """Fuse the gradient of ``weight * mean_live ||exp(features[:, 4:7])||`` (the trainer's scale regulariser,
        LOS:42-54) into the step of the [N,56] parameter ``features``: it is added to the incoming gradient inside the
        Adam kernel (from the pre-step parameters), so neither autograd nor a separate pass has to produce it.
        ``weight = 0`` switches it off.

        ``local_share`` (owner-sharded Gaussians: this tensor holds only the rank's block): a zero-argument callable
        that returns n_live(this block) / n_live(all ranks).  The kernel divides by the live count it finds in the tensor
        it is given; the reference's term is a mean over ALL live Gaussians (LOS:42-54), so the weight handed to the
        kernel is ``weight * local_share()`` -- the same per-point gradient as an un-sharded run."""
        if features.dim() != 2 or features.shape[1] != 56:
            raise ValueError("the scale regulariser applies to the [N,56] feature matrix")
        if weight:
            self._scale_regulariser[id(features)] = (float(weight), point_invalid_mask, local_share)
        else:
            self._scale_regulariser.pop(id(features), None)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            beta1, beta2 = group["betas"]
            for p in group["params"]:
                if p.grad is None:
                    continue
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                    raise RuntimeError("optim.Adam needs contiguous float32 parameters on the HIP device")
                grad = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                state = self.state[p]
                if not state:
                    state["step"] = 0
                    state["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                state["step"] = int(state["step"]) + 1
                reg = self._scale_regulariser.get(id(p))
                mask = self._row_mask.get(id(p))
                if mask is not None:
                    if reg is not None and reg[1] is not mask:
                        raise RuntimeError("set_row_mask and set_scale_regulariser were given different masks")
                    if "last_step" not in state:
                        # moments restored from a state_dict written without row masks (or by torch.optim.Adam) are
                        # current as of the previous step: no lazy decay is owed for them
                        state["last_step"] = torch.full((p.shape[0],), int(state["step"]) - 1, dtype=torch.int32,
                                                        device=p.device)
                    weight = (reg[0] * (float(reg[2]()) if reg[2] is not None else 1.0)) if reg is not None else 0.0
                    ws = torch.empty(256, dtype=torch.int32, device=p.device) if weight else None
                    call("gs_adam_step_rows", ptr(p), ptr(grad), ptr(state["exp_avg"]), ptr(state["exp_avg_sq"]),
                         p.shape[0], p.shape[1], float(group["lr"]), float(beta1), float(beta2), float(group["eps"]),
                         state["step"], ptr(mask), ptr(state["last_step"]), float(weight), ptr(ws),
                         current_stream(p.device))
                    continue
                if reg is not None:
                    weight, mask = reg[0] * (float(reg[2]()) if reg[2] is not None else 1.0), reg[1]
                    ws = torch.empty(256, dtype=torch.int32, device=p.device)
                    call("gs_adam_step_features", ptr(p), ptr(grad), ptr(state["exp_avg"]), ptr(state["exp_avg_sq"]),
                         p.shape[0], float(group["lr"]), float(beta1), float(beta2), float(group["eps"]), state["step"],
                         ptr(mask), weight, ptr(ws), current_stream(p.device))
                    continue
                call("gs_adam_step", ptr(p), ptr(grad), ptr(state["exp_avg"]), ptr(state["exp_avg_sq"]), p.numel(),
                     float(group["lr"]), float(beta1), float(beta2), float(group["eps"]), state["step"],
                     current_stream(p.device))
        return loss
