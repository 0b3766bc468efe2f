"""Device-resident replay storage and its sample path (mirror of super_sac/replay.py:10-190).

Layout in HBM: structure-of-arrays ring, one row-major array per field, exactly the fields of
the reference's ``ReplayBufferStorage`` (replay.py:12-23): ``action (size, A) f32``,
``reward (size, 1) f32``, ``done (size, 1) u8`` and per observation label ``s / s1
(size, *shape)`` in the environment's dtype (uint8 pixels stay uint8: 2 x 63.5 KB per DMC
transition, so a 1M-transition pixel buffer is ~127 GB and still fits one MI355X's 288 GB).
Sampling draws indices on the host from the torch CPU generator (bit-exact with
replay.py:122) and gathers rows with the ``ssac_gather_rows`` kernel (uint8 -> fp32 cast fused).
"""
import numpy as np
import torch

from . import engine, rng
from . import device as _default_device
from ._lib import check, lib


class _IndexStager:
    """pinned host buffer -> device copy of the sampled indices, without a sync."""

    def __init__(self, device):
        self.device = device
        self._pinned = {}
        self._dev = {}
        self._events = {}
        self._turn = 0

    def upload(self, cpu_tensor, tag="idx", slots=8):
        key = (tag, tuple(cpu_tensor.shape), cpu_tensor.dtype)
        ring = self._pinned.get(key)
        if ring is None:
            ring = [torch.empty(cpu_tensor.shape, dtype=cpu_tensor.dtype).pin_memory()
                    for _ in range(slots)]
            self._pinned[key] = ring
            self._dev[key] = [torch.empty(cpu_tensor.shape, dtype=cpu_tensor.dtype, device=self.device)
                              for _ in range(slots)]
            self._events[key] = [None] * slots
        self._turn = (self._turn + 1) % slots
        ev = self._events[key][self._turn]
        if ev is not None:
            ev.synchronize()  # the host may run many updates ahead of the device
        ring[self._turn].copy_(cpu_tensor)
        dev = self._dev[key][self._turn]
        dev.copy_(ring[self._turn], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._events[key][self._turn] = ev
        return dev


class ReplayBufferStorage:
    def __init__(self, size, state_example, act_example, device):
        self.device = device
        self.size = size
        self.action_stack = torch.zeros((size,) + tuple(act_example.shape), dtype=torch.float32,
                                        device=device)
        self.reward_stack = torch.zeros((size, 1), dtype=torch.float32, device=device)
        self.done_stack = torch.zeros((size, 1), dtype=torch.uint8, device=device)
        self.s_stack, self.s1_stack, self.s_dtypes = {}, {}, {}
        for label, array in state_example.items():
            dt = torch.uint8 if array.dtype == np.uint8 else torch.float32
            self.s_dtypes[label] = dt
            shape = (size,) + tuple(array.shape)
            self.s_stack[label] = torch.zeros(shape, dtype=dt, device=device)
            self.s1_stack[label] = torch.zeros(shape, dtype=dt, device=device)
        self._next_idx = 0
        self._max_filled = 0

    def __len__(self):
        return self._max_filled

    def _put(self, dst, rows, host, dtype):
        t = torch.from_numpy(np.ascontiguousarray(host)).to(dtype).reshape((len(rows),) + tuple(dst.shape[1:]))
        dst[rows] = t.to(self.device, non_blocking=False)

    def add(self, s, a, r, s1, d):
        a = np.asarray(a)
        num = len(a) if a.ndim > 1 else 1
        R = np.arange(self._next_idx, self._next_idx + num) % self.size
        rows = torch.from_numpy(R).to(self.device)
        for label in s:
            self._put(self.s_stack[label], rows, np.asarray(s[label]), self.s_dtypes[label])
            self._put(self.s1_stack[label], rows, np.asarray(s1[label]), self.s_dtypes[label])
        self._put(self.action_stack, rows, a.astype(np.float32), torch.float32)
        self._put(self.reward_stack, rows, np.asarray(r, dtype=np.float32), torch.float32)
        self._put(self.done_stack, rows, np.asarray(d).astype(np.uint8), torch.uint8)
        self._max_filled = min(max(self._next_idx + num, self._max_filled), self.size)
        self._next_idx = (self._next_idx + num) % self.size
        return R

    def gather_field(self, src, idx_dev, n, dst=None, ld=None, col0=0):
        """rows src[idx] -> fp32 (n, row_elems) (or into `dst` at column col0 with row stride ld)."""
        row_elems = int(np.prod(src.shape[1:])) if src.dim() > 1 else 1
        if dst is None:
            dst = torch.empty((n,) + tuple(src.shape[1:]), dtype=torch.float32, device=self.device)
            ld, col0 = row_elems, 0
        check(lib.ssac_gather_rows(src.data_ptr(), 1 if src.dtype == torch.uint8 else 0, row_elems,
                                   idx_dev.data_ptr(), n, dst.data_ptr(), ld, col0, engine.stream()))
        return dst


class ReplayBuffer:
    def __init__(self, size, alpha=0.6, beta=1.0, device=None):
        assert alpha >= 0
        self._maxsize = size
        self._storage = None
        self.alpha, self.beta = alpha, beta
        self.total_sample_calls = 0
        self.device = torch.device(device) if device is not None else _default_device
        self._stager = None
        self._per = None

    def __len__(self):
        return len(self._storage) if self._storage is not None else 0

    def push(self, state, action, reward, next_state, done, priorities=None, **kwargs):
        engine.require_gpu()
        action = np.asarray(action)
        if self._storage is None:
            if action.ndim > 1:
                act_example = action[0]
                state_example = {x: np.asarray(y)[0] for x, y in state.items()}
            else:
                act_example, state_example = action, {x: np.asarray(y) for x, y in state.items()}
            self._storage = ReplayBufferStorage(self._maxsize, state_example, act_example, self.device)
            self._stager = _IndexStager(self.device)
        R = self._storage.add(state, action, reward, next_state, done)
        if self._per is not None:
            self._per.push_rows(R, priorities)
        return R

    def load_experience(self, s, a, r, s1, d):
        assert len(s) <= self._maxsize, "Experience dataset is larger than the buffer."
        r, d = np.asarray(r), np.asarray(d)
        if r.ndim < 2:
            r = np.expand_dims(r, 1)
        if d.ndim < 2:
            d = np.expand_dims(d, 1)
        self.push(s, a, r, s1, d)

    # ---- sample path -----------------------------------------------------------------------
    def draw_uniform_indices(self, batch_size):
        """(cpu int64 tensor, device int64 tensor) of replay.py:122's torch.randint draw."""
        self.total_sample_calls += 1
        idx = rng.draw_indices(len(self._storage), batch_size)
        return idx, self._stager.upload(idx)

    def gather(self, idx_dev, n):
        st = self._storage
        state = {k: st.gather_field(v, idx_dev, n) for k, v in st.s_stack.items()}
        next_state = {k: st.gather_field(v, idx_dev, n) for k, v in st.s1_stack.items()}
        action = st.gather_field(st.action_stack, idx_dev, n)
        if action.dim() < 2:
            action = action.unsqueeze(1)
        reward = st.gather_field(st.reward_stack, idx_dev, n)
        done = st.gather_field(st.done_stack, idx_dev, n)
        return state, action, reward, next_state, done

    def sample_uniform(self, batch_size):
        idx, idx_dev = self.draw_uniform_indices(batch_size)
        return self.gather(idx_dev, batch_size), idx.numpy()
