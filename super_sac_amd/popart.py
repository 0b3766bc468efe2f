"""PopArt target normalisation with its state resident in device memory.

Mirror of super_sac/popart.py:8-59.  The statistics update and the (de)normalisation of the
TD target run inside the ``ssac_td_target`` kernel; this class only owns the state block
(struct ssac_popart) and exposes the reference's attributes for inspection / checkpoints.
Like the reference's layer (whose mu/nu/w/b are plain tensors, so ``state_dict()`` is empty and the statistics are
silently dropped from checkpoints), ``state_dict()`` is empty here -- the files stay interchangeable -- and the
statistics are saved through ``stats_dict()`` (Agent.save writes popart{i}_stats.pt).
"""
import ctypes as C

import torch
from torch import nn

from . import _lib


class PopArtLayer(nn.Module):
    def __init__(self, beta=1e-4, min_steps=1000, init_nu=0):
        super().__init__()
        st = _lib.PopArtState(0.0, float(init_nu), 1.0, 0.0, 1, int(min_steps), 0, 0, float(beta))
        raw = torch.frombuffer(bytearray(bytes(st)), dtype=torch.uint8).clone()
        # non-persistent: state_dict() stays EMPTY like the reference layer's (popart.py:8-20 holds plain tensors), so
        # popart{i}.pt files are interchangeable; the statistics travel through stats_dict() / load_stats_dict()
        self.register_buffer("state", raw, persistent=False)

    def stats_dict(self):
        return {"state": self.state.detach().cpu().clone()}

    def load_stats_dict(self, d):
        self.state.copy_(d["state"])

    # ---- host views of the device struct (each read synchronises; not on the update path)
    def _read(self):
        return _lib.PopArtState.from_buffer_copy(bytes(self.state.cpu().numpy().tobytes()))

    def _write(self, st):
        self.state.copy_(torch.frombuffer(bytearray(bytes(st)), dtype=torch.uint8))

    mu = property(lambda self: self._read().mu)
    nu = property(lambda self: self._read().nu)
    w = property(lambda self: self._read().w)
    b = property(lambda self: self._read().b)
    _t = property(lambda self: self._read().t)
    _stable = property(lambda self: bool(self._read().stable))
    beta = property(lambda self: self._read().beta)

    @property
    def min_steps(self):
        return self._read().min_steps

    @min_steps.setter
    def min_steps(self, v):
        st = self._read()
        st.min_steps = int(v)
        self._write(st)

    @property
    def sigma(self):
        st = self._read()
        import math
        s = math.sqrt(st.nu - st.mu ** 2) + 1e-5 if st.nu - st.mu ** 2 >= 0 else float("nan")
        return min(max(s, 1e-4), 1e6)

    @property
    def ptr(self):
        return self.state.data_ptr()

    def __bool__(self):
        return True
