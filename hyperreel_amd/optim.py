"""The optimizer of the training loop on the HIP device.

The reference builds `torch.optim.Adam(parameters, lr=..., eps=1e-8, weight_decay=..., betas=(0.9, 0.99))` per parameter group
(utils/__init__.py:49-76, get_optimizer; nlf/__init__.py:504-530, configure_optimizers) and Lightning steps it after every
`training_step`.  `HipAdam` is that optimizer with the same constructor, the same state names (`step`, `exp_avg`, `exp_avg_sq`: its
`state_dict()` loads into `torch.optim.Adam` and back) and the same arithmetic, whose `step()` is ONE launch of `hr_adam_step` over every
parameter of every group -- one pass over memory instead of the eleven of torch's default foreach form."""
import ctypes as C

import torch

from . import lib as _lib


class HipAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        if lr < 0.0 or eps < 0.0 or weight_decay < 0.0 or not (0.0 <= betas[0] < 1.0) or not (0.0 <= betas[1] < 1.0):
            raise ValueError('invalid Adam hyper-parameter')
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        ps, gs, ms, vs, ns, hp = [], [], [], [], [], []
        keep = []
        dev = None
        # validate EVERYTHING before any state is touched: a raise must not leave some parameters' step counts advanced
        for group in self.param_groups:
            if group.get('amsgrad', False) or group.get('maximize', False):
                raise RuntimeError('HipAdam implements plain Adam: amsgrad / maximize are not supported (use torch.optim.Adam for those groups)')
            for p in group['params']:
                if p.grad is None:
                    continue
                if p.device.type != 'cuda':
                    raise RuntimeError('HipAdam steps parameters on the HIP device; there is no CPU path')
                if p.dtype != torch.float32 or p.grad.dtype != torch.float32 or p.grad.is_sparse or not p.is_contiguous():
                    raise RuntimeError('HipAdam: contiguous float32 parameters with dense float32 gradients')
                if dev is None:
                    dev = p.device
                elif p.device != dev:
                    raise RuntimeError('HipAdam: all parameters of one optimizer on one device')
        for group in self.param_groups:
            b1, b2 = group['betas']
            for p in group['params']:
                if p.grad is None:
                    continue
                if p.device.type != 'cuda':
                    raise RuntimeError('HipAdam steps parameters on the HIP device; there is no CPU path')
                if p.dtype != torch.float32 or p.grad.dtype != torch.float32 or p.grad.is_sparse or not p.is_contiguous():
                    raise RuntimeError('HipAdam: contiguous float32 parameters with dense float32 gradients')
                if dev is None:
                    dev = p.device
                elif p.device != dev:
                    raise RuntimeError('HipAdam: all parameters of one optimizer on one device')
                st = self.state[p]
                if len(st) == 0:
                    st['step'] = torch.tensor(0.0, dtype=torch.float32)               # (host, as torch.optim.Adam keeps it when not capturable)
                    st['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st['step'] += 1
                g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                keep.append(g)
                ps.append(p.data_ptr()); gs.append(g.data_ptr()); ms.append(st['exp_avg'].data_ptr()); vs.append(st['exp_avg_sq'].data_ptr())
                ns.append(p.numel())
                hp += [float(group['lr']), float(b1), float(b2), float(group['eps']), float(group['weight_decay']), float(st['step'])]
        if not ps:
            return loss
        k = len(ps)
        PV, NV, HV = C.c_void_p * k, C.c_int64 * k, C.c_double * (6 * k)
        L = _lib.load()
        with torch.cuda.device(dev):
            _lib.check(L.hr_adam_step(PV(*ps), PV(*gs), PV(*ms), PV(*vs), NV(*ns), HV(*hp), k,
                                      C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)), 'hr_adam_step')
        return loss
