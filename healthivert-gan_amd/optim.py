"""Multi-tensor Adam on the HIP path (one launch per optimiser), torch.optim.Adam semantics.

Subclasses torch.optim.Optimizer only so that the reference's LR schedulers (LambdaLR etc.,
reference models/networks.py:39-65, models/base_model.py:124-134) drive it unchanged; the update
itself is hv_adam_step.  Reference: the four Adam optimisers of models/pix2pix_model.py:127-130.
"""
import torch

from . import ops


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0):
        if weight_decay != 0:
            raise NotImplementedError("FusedAdam: weight_decay is not used by the reference and not implemented")
        super().__init__(list(params), dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._table = ops.LayerTable('hv_adam_tensor')
        self._m = self._v = self._step = self._lr = None
        self._lr_host = None

    def _ensure(self):
        ps = [p for g in self.param_groups for p in g['params']]
        if any(p.grad is None for p in ps):
            raise RuntimeError("FusedAdam.step(): a parameter has no gradient")
        key = tuple((p.data_ptr(), p.grad.data_ptr()) for p in ps)
        dev = ps[0].device
        if self._m is None or self._m.device != dev:
            n = sum(p.numel() for p in ps)
            self._m = torch.zeros(n, dtype=torch.float32, device=dev)
            self._v = torch.zeros(n, dtype=torch.float32, device=dev)
            if self._lr is None or self._lr.device != dev:
                self._lr = None
                self.sync_lr()
        if key != self._table.key:
            rows, off = [], 0
            for p in ps:
                n = p.numel()
                rows.append(dict(p=p.data, g=p.grad, m=self._m[off:off + n], v=self._v[off:off + n], n=n))
                off += n
            self._table.update(rows, key, dev)
            self._max = max(p.numel() for p in ps)
        return ps

    def _ensure_state(self):
        """Moments, step count and learning rate on the device before the first step (Pix2PixModel.dp_preflight snapshots them)."""
        ps = [p for g in self.param_groups for p in g['params']]
        dev = ps[0].device
        if self._m is None or self._m.device != dev:
            n = sum(p.numel() for p in ps)
            self._m = torch.zeros(n, dtype=torch.float32, device=dev)
            self._v = torch.zeros(n, dtype=torch.float32, device=dev)
            if self._lr is None or self._lr.device != dev:
                self._lr = None
        self.sync_lr()

    def sync_lr(self):
        """Copy the scheduler's learning rate to the device scalar the kernel reads (outside any graph capture)."""
        lr = float(self.param_groups[0]['lr'])
        if self._lr is None:
            dev = self.param_groups[0]['params'][0].device
            self._step = torch.zeros(8, dtype=torch.float32, device=dev)      # [0] step count; [1..4]: the overflow guard's state (hv_adam_step_guarded)
            self._lr = torch.zeros(1, dtype=torch.float32, device=dev)
            self._lr_host = None
        if lr != self._lr_host:
            self._lr.fill_(lr)
            self._lr_host = lr

    @torch.no_grad()
    def step(self, closure=None, sync_lr=True, guard_flat=None, grad_mul=1.0):
        """guard_flat: the flat gradient buffer behind the parameters' .grad views (fp16 storage mode): the update is skipped on the device when it
        holds an inf / nan (scaled gradients that overflowed an fp16 gradient buffer); skipped_steps() counts those.  grad_mul: the factor that takes the
        loss scale out of the gradients (1 / scale), applied by the same pass that checks them."""
        self._ensure()
        if sync_lr:
            self.sync_lr()
        g = self.param_groups[0]
        ops.adam_step(self._table, self._max, self._lr, g['betas'][0], g['betas'][1], g['eps'], self._step, guard_flat=guard_flat, grad_mul=grad_mul)

    def skipped_steps(self):
        """Steps the overflow guard skipped so far (a host read: call it between steps, not inside them)."""
        return 0 if self._step is None else int(self._step[2].item())

    def zero_grad(self, set_to_none=False):
        """Gradients are (re)assigned by the explicit backward; nothing to clear.  The first parameter is flagged so that the
        nn.Module-API autograd bridges know the next backward assigns and later ones accumulate (networks.grads_are_fresh).  EVERY parameter
        of every group is flagged: one optimiser may span several networks (itertools.chain of their parameters)."""
        for g in self.param_groups:
            for p in g['params']:
                p._hv_fresh = True
        return None
