"""Adam that leaves elements with a zero gradient untouched and honours a per-parameter learning-rate multiplier --
reference neural_renderer/optimizers.py:9-39.

    opt = Adam(params, alpha=1e-3, beta1=0.9, beta2=0.999, eps=1e-8)     # Chainer's hyper-parameter names
    param.lr = 0.1                                                       # optional multiplier (Mesh.set_lr)

On CUDA float32 parameters the update is the HIP kernel `nr_adam_update` (csrc/nr_optim.hip, one launch per parameter,
the reference's arithmetic literally); other tensors take the equivalent torch.where formulation."""
import math

import torch

from . import _lib


class Adam(torch.optim.Optimizer):
    def __init__(self, params, alpha=0.001, beta1=0.9, beta2=0.999, eps=1e-8):
        super(Adam, self).__init__(params, dict(alpha=alpha, beta1=beta1, beta2=beta2, eps=eps))

    @torch.no_grad()
    def step(self):
        for group in self.param_groups:
            for p in group['params']:
                if p.grad is None:  # optimizers.py:18-19
                    continue
                state = self.state[p]
                if not state:
                    state['t'] = 0
                    state['m'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    state['v'] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                state['t'] += 1
                t, b1, b2 = state['t'], group['beta1'], group['beta2']
                lr = group['alpha'] * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)  # chainer AdamRule.lr
                lr = lr * getattr(p, 'lr', 1.0)                                # optimizers.py:20
                if lr == 0:                                                    # :21
                    continue
                g = p.grad
                if p.is_cuda and p.dtype == torch.float32 and g.dtype == torch.float32 and p.is_contiguous():
                    g = g.contiguous()
                    with torch.cuda.device(p.device):
                        _lib.check(_lib.load().nr_adam_update(
                            p.data_ptr(), g.data_ptr(), state['m'].data_ptr(), state['v'].data_ptr(), p.numel(),
                            lr, 1 - b1, 1 - b2, group['eps'], torch.cuda.current_stream(p.device).cuda_stream),
                            'nr_adam_update')
                    continue
                mask = g != 0  # :26
                m = torch.where(mask, state['m'] + (1 - b1) * (g - state['m']), state['m'])
                v = torch.where(mask, torch.clamp(state['v'] + (1 - b2) * (g * g - state['v']), min=0), state['v'])
                state['m'], state['v'] = m, v
                p.copy_(torch.where(mask, p - lr * m / (torch.sqrt(v) + group['eps']), p))
