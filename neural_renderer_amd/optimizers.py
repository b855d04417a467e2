"""Adam that leaves elements with a zero gradient untouched -- reference neural_renderer/optimizers.py:9-39
(its masked-update CUDA kernel K10 is expressed with torch.where; off the hot path)."""
import math

import torch


class Adam(torch.optim.Optimizer):
    def __init__(self, params, alpha=0.001, beta1=0.9, beta2=0.999, eps=1e-8):
        super(Adam, self).__init__(params, dict(alpha=alpha, beta1=beta1, beta2=beta2, eps=eps))

    @torch.no_grad()
    def step(self):
        for group in self.param_groups:
            for p in group['params']:
                if p.grad is None:
                    continue
                state = self.state[p]
                if not state:
                    state['t'] = 0
                    state['m'] = torch.zeros_like(p)
                    state['v'] = torch.zeros_like(p)
                state['t'] += 1
                t, b1, b2 = state['t'], group['beta1'], group['beta2']
                lr = group.get('lr', group['alpha']) * math.sqrt(1 - b2 ** t) / (1 - b1 ** t)
                g = p.grad
                mask = g != 0  # optimizers.py:26 `if (grad != 0)`
                m = torch.where(mask, state['m'] + (1 - b1) * (g - state['m']), state['m'])
                v = torch.where(mask, state['v'] + (1 - b2) * (g * g - state['v']), state['v'])
                state['m'], state['v'] = m, v
                p.copy_(torch.where(mask, p - lr * m / (torch.sqrt(v) + group['eps']), p))
