"""Termination predicates -- host-side mirror of the reference's `algo/mb_utils/terminal_funs.py`.

`get_termination_fn(task)` keeps the reference's name, substring dispatch order (:123-149) and error
behaviour (an unknown task raises TypeError, because the reference executes `raise np.zeros`).
The returned object is callable like the reference's functions -- fn(obs, act, next_obs) -> bool [B,1],
accepting device tensors or arrays -- and carries `.task_id`, the enum the fused kernel
(k_dyn_sample in csrc/dynamics.hip) evaluates on the device so that `step()` needs no D2H sync.
"""
import numpy as np
import torch

from ... import _lib

_DISPATCH = [("halfcheetahvel", "never"), ("halfcheetah", "halfcheetah"), ("hopper", "hopper"), ("antangle", "ant"),
             ("ant", "ant"), ("walker2d", "walker2d"), ("point2denv", "never"), ("point2dwallenv", "never"),
             ("pendulum", "never"), ("humanoid", "humanoid"), ("pen", "pen"), ("door", "never")]


class TerminationFn:
    def __init__(self, kind):
        self.kind = kind
        self.task_id = _lib.TERM_IDS[kind]

    def __call__(self, obs, act, next_obs):
        from ... import ops
        n = next_obs if isinstance(next_obs, torch.Tensor) else torch.as_tensor(np.asarray(next_obs))
        assert n.dim() == 2, "termination functions take [B, S] arrays"      # reference asserts 2-D inputs
        n = n.to(device="cuda", dtype=torch.float32).contiguous()
        return ops.termination(self.task_id, n).cpu().numpy().astype(bool).reshape(-1, 1)

    def __repr__(self):
        return f"TerminationFn({self.kind})"


def get_termination_fn(task):
    for key, kind in _DISPATCH:
        if key in task:
            return TerminationFn(kind)
    raise TypeError("exceptions must derive from BaseException")   # what `raise np.zeros` does in the reference
