"""Fused Adam: one HIP launch per step over the model's flat parameter arena.

Counterpart of ``optim.Adam(model.parameters(), lr=...)`` (runners/ivae.py:38, wrappers.py:44,127)
with torch's defaults (betas (.9,.999), eps 1e-8, no weight decay, no amsgrad).  ``lr`` is read
from ``param_groups`` at every step, so ``ReduceLROnPlateau`` keeps working, and
``state_dict()`` has torch.optim.Adam's layout (per-parameter ``step``, ``exp_avg``,
``exp_avg_sq``), so ``utils.checkpoint`` (utils/utils.py:55-64) is unchanged.
"""
import torch

from . import ops


class FusedAdam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8):
        # every key torch.optim.Adam keeps in a param group, at the only values this optimiser implements, so that a
        # state_dict saved here loads into torch.optim.Adam (its load_state_dict takes the SAVED groups) and back
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=0, amsgrad=False, maximize=False,
                                      foreach=None, capturable=False, differentiable=False, fused=None,
                                      decoupled_weight_decay=False))
        self._flat = None        # (params tuple, flat_m, flat_v) of the direct path (step_flat)
        self._flat_key = None    # the param list those were built for
        self._pending = 0        # steps taken by step_flat not yet written to the per-parameter `step` entries

    # -- direct path: the caller holds the flat parameter arena and the flat gradient ----------------------
    def _sync_steps(self):
        if self._pending:
            for p in self._flat[0]:
                self.state[p]['step'] += self._pending
            self._pending = 0

    def step_flat(self, flat_p, flat_g):
        """One Adam update over a whole model whose parameters tile `flat_p` in `param_groups[0]` order, with the
        gradient given as one flat buffer (what `ivae_elbo_fwd_bwd` returns): no per-parameter Python work.  The
        per-parameter state (`step`, `exp_avg`, `exp_avg_sq`) stays what `state_dict()` of torch.optim.Adam holds."""
        if len(self.param_groups) != 1:
            raise RuntimeError('step_flat needs a single parameter group')
        group = self.param_groups[0]
        plist = group['params']
        if self._flat is None or self._flat_key is not plist or len(plist) != len(self._flat[0]):
            ps = tuple(plist)
            self._sync_steps()
            if any(len(self.state[p]) == 0 for p in ps):
                self._init_group_state(list(ps))
            m = self._flat_view([self.state[p]['exp_avg'] for p in ps])
            v = self._flat_view([self.state[p]['exp_avg_sq'] for p in ps])
            if m is None or v is None or m.numel() != flat_p.numel():
                raise RuntimeError('optimizer state is not one flat buffer over these parameters')
            self._flat, self._flat_key = (ps, m, v), plist
            self._step0 = int(self.state[ps[0]]['step'])
        self._pending += 1
        b1, b2 = group['betas']
        ops.adam_step(flat_p, self._flat[1], self._flat[2], flat_g, group['lr'], self._step0 + self._pending, b1, b2,
                      group['eps'])

    def flat_state(self, flat_p):
        """-> (exp_avg, exp_avg_sq, steps_taken, group) as flat buffers over `flat_p`, for callers that run several
        updates in one library call (`ivae_elbo_adam_steps`) and then report them with `advance(n)`."""
        if len(self.param_groups) != 1:
            raise RuntimeError('flat_state needs a single parameter group')
        group = self.param_groups[0]
        plist = group['params']
        if self._flat is None or self._flat_key is not plist or len(plist) != len(self._flat[0]):
            ps = tuple(plist)
            self._sync_steps()
            if any(len(self.state[p]) == 0 for p in ps):
                self._init_group_state(list(ps))
            m = self._flat_view([self.state[p]['exp_avg'] for p in ps])
            v = self._flat_view([self.state[p]['exp_avg_sq'] for p in ps])
            if m is None or v is None or m.numel() != flat_p.numel():
                raise RuntimeError('optimizer state is not one flat buffer over these parameters')
            self._flat, self._flat_key = (ps, m, v), plist
            self._step0 = int(self.state[ps[0]]['step'])
        return self._flat[1], self._flat[2], self._step0 + self._pending, group

    def advance(self, n):
        """Record n updates done through the buffers of `flat_state`."""
        self._pending += int(n)

    def state_dict(self):
        self._sync_steps_keep()
        return super().state_dict()

    def load_state_dict(self, state_dict):
        """torch.optim.Adam's state_dict (what utils/utils.py:55-64 stores under 'optimizer_state_dict') or our own:
        the loaded exp_avg / exp_avg_sq are re-packed into ONE flat buffer per group (a checkpoint loaded with
        map_location='cpu' arrives as separate tensors) and every cache of the direct path is dropped, so step(),
        step_flat() and flat_state() continue from the loaded moments and step count."""
        self._pending = 0                     # steps of the state being replaced
        super().load_state_dict(state_dict)
        self._reset_flat()

    def __setstate__(self, state):
        super().__setstate__(state)
        self.__dict__.setdefault('_pending', 0)
        self._reset_flat()

    def _reset_flat(self):
        self._flat, self._flat_key, self._pending = None, None, 0
        for group in self.param_groups:
            if group.get('weight_decay', 0) != 0 or group.get('amsgrad', False) or group.get('maximize', False):
                raise NotImplementedError('FusedAdam implements torch.optim.Adam with weight_decay=0, amsgrad=False, '
                                          'maximize=False (what the reference uses: runners/ivae.py:38, wrappers.py:44)')
            ps = [p for p in group['params'] if len(self.state.get(p, {}))]
            if not ps:
                continue
            total = sum(p.numel() for p in ps)
            m = torch.empty(total, dtype=torch.float32, device=ps[0].device)
            v = torch.empty_like(m)
            off = 0
            for p in ps:
                n, st = p.numel(), self.state[p]
                m[off:off + n].copy_(st['exp_avg'].reshape(-1))
                v[off:off + n].copy_(st['exp_avg_sq'].reshape(-1))
                st['exp_avg'], st['exp_avg_sq'] = m[off:off + n].view(p.shape), v[off:off + n].view(p.shape)
                st['step'] = torch.tensor(float(st['step']))          # host scalar, as torch.optim.Adam keeps it
                off += n

    def _sync_steps_keep(self):
        if self._pending:
            n = self._pending
            self._sync_steps()
            self._step0 += n

    @staticmethod
    def _flat_view(tensors):
        """If `tensors` tile one contiguous buffer in order, return it as a flat view."""
        if not tensors:
            return None
        first = tensors[0]
        ptr, total = first.data_ptr(), 0
        for t in tensors:
            if not t.is_contiguous() or t.dtype != torch.float32 or t.data_ptr() != ptr + 4 * total:
                return None
            total += t.numel()
        storage_off = first.storage_offset()
        if first.untyped_storage().nbytes() < 4 * (storage_off + total):
            return None
        return torch.as_strided(first, (total,), (1,), storage_off)

    def _init_group_state(self, ps):
        total = sum(p.numel() for p in ps)
        m = torch.zeros(total, dtype=torch.float32, device=ps[0].device)
        v = torch.zeros_like(m)
        off = 0
        for p in ps:
            n = p.numel()
            st = self.state[p]
            st['step'] = torch.tensor(0.)
            st['exp_avg'] = m[off:off + n].view(p.shape)
            st['exp_avg_sq'] = v[off:off + n].view(p.shape)
            off += n

    @torch.no_grad()
    def step(self, closure=None):
        self._sync_steps_keep()
        self._flat = None
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            ps = [p for p in group['params'] if p.grad is not None]
            if not ps:
                continue
            if any(len(self.state[p]) == 0 for p in ps):
                self._init_group_state(ps)
            b1, b2 = group['betas']
            for p in ps:
                self.state[p]['step'] += 1
            step = int(self.state[ps[0]]['step'])
            flat_p = self._flat_view([p.data for p in ps])
            flat_g = self._flat_view([p.grad for p in ps])
            flat_m = self._flat_view([self.state[p]['exp_avg'] for p in ps])
            flat_v = self._flat_view([self.state[p]['exp_avg_sq'] for p in ps])
            if all(t is not None for t in (flat_p, flat_g, flat_m, flat_v)):
                ops.adam_step(flat_p, flat_m, flat_v, flat_g, group['lr'], step, b1, b2, group['eps'])
            else:   # parameters not backed by one arena: one launch per tensor, same kernel
                for p in ps:
                    st = self.state[p]
                    g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
                    ops.adam_step(p.data.view(-1), st['exp_avg'].view(-1), st['exp_avg_sq'].view(-1), g.view(-1),
                                  group['lr'], int(st['step']), b1, b2, group['eps'])
        return loss
