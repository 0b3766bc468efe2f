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
from . import _lib
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


class PrioritySampler:
    """Proportional prioritised replay index sampler (mirror of replay.py:140-190, 207-353).

    Host-side by design: the draw consumes numpy's GLOBAL generator (replay.py:166) and the trees are
    float64, so keeping them in numpy makes the index stream and the importance weights bit-identical
    to the reference; only the gather of the selected rows runs on the device.  Sum and min trees are
    stored as implicit binary heaps over the next power of two >= capacity (replay.py:147-152)."""

    def __init__(self, capacity, alpha=0.6, beta=1.0):
        cap = 1
        while cap < capacity:
            cap *= 2
        self.cap, self.alpha, self.beta = cap, alpha, beta
        self.sum_tree = np.zeros(2 * cap, dtype=np.float64)
        self.min_tree = np.full(2 * cap, np.inf, dtype=np.float64)
        self._max_priority = 1.0
        # priority ** alpha: numpy's power, as the reference computes its leaves (replay.py:183-190).  Tests swap in an
        # exactly rounded power to check the device trees bit for bit (numpy's own differs between its SIMD and scalar
        # paths in the last bit; csrc/ssac_per.hip holds the correctly rounded value)
        self.pow_fn = np.power

    def _assign(self, rows, values):
        rows = np.atleast_1d(np.asarray(rows, dtype=np.int64))
        leaves = rows + self.cap
        self.sum_tree[leaves] = values
        self.min_tree[leaves] = values
        nodes = np.unique(leaves >> 1)
        while nodes.size:
            self.sum_tree[nodes] = self.sum_tree[2 * nodes] + self.sum_tree[2 * nodes + 1]
            self.min_tree[nodes] = np.minimum(self.min_tree[2 * nodes], self.min_tree[2 * nodes + 1])
            if nodes[0] <= 1:
                break
            nodes = np.unique(nodes >> 1)

    def push_rows(self, rows, priorities=None):
        pr = self._max_priority if priorities is None else priorities
        self._assign(rows, self.pow_fn(np.asarray(pr, dtype=np.float64), self.alpha))

    def update_priorities(self, idxes, priorities, n_filled):
        priorities = np.asarray(priorities, dtype=np.float64)
        assert len(idxes) == len(priorities)
        assert np.min(priorities) > 0
        assert np.min(idxes) >= 0
        assert np.max(idxes) < n_filled
        self._assign(idxes, self.pow_fn(priorities, self.alpha))
        self._max_priority = max(self._max_priority, float(np.max(priorities)))

    def load_state(self, sum_tree, min_tree, max_priority):
        self.sum_tree[:], self.min_tree[:] = sum_tree, min_tree
        self._max_priority = float(max_priority)

    def _range_sum(self, start, end_exclusive):
        return float(self.sum_tree[self.cap + start: self.cap + end_exclusive].sum())

    def descend(self, mass):
        """largest i with prefix_sum(i) <= mass, for a vector of masses (replay.py:297-336)."""
        mass = np.array(mass, dtype=np.float64)
        node = np.ones(len(mass), dtype=np.int64)
        while True:
            inner = node < self.cap
            if not inner.any():
                break
            left = np.where(inner, 2 * node, node)
            lsum = self.sum_tree[left]
            right = inner & (lsum <= mass)
            mass = np.where(right, mass - lsum, mass)
            node = np.where(inner, left + right.astype(np.int64), node)
        return node - self.cap

    def sample(self, n_filled, batch_size):
        # the reference sums leaves [0, n_filled-2] (SegmentTree.reduce makes `end` exclusive after -1)
        total = self._range_sum(0, n_filled - 1)
        mass = np.random.random(size=batch_size) * total
        idx = self.descend(mass)
        p_min = self.min_tree[1] / self.sum_tree[1]
        max_weight = (p_min * n_filled) ** (-self.beta)
        p_sample = self.sum_tree[self.cap + idx] / self.sum_tree[1]
        weights = (p_sample * n_filled) ** (-self.beta) / max_weight
        return idx, weights


class LazyHost:
    """a device tensor that turns into a numpy array when somebody looks at it (``np.asarray``, indexing, ``len``):
    the indices / weights of a prioritised draw stay on the device along the update path and only cost a
    synchronisation where a caller really reads them on the host (the reference hands out numpy arrays, replay.py:177)"""

    def __init__(self, dev_tensor):
        self.dev = dev_tensor
        self._np = None

    def numpy(self):
        if self._np is None:
            self._np = self.dev.cpu().numpy()
        return self._np

    def __array__(self, dtype=None, copy=None):
        a = self.numpy()
        return a if dtype is None else a.astype(dtype)

    def __len__(self):
        return self.dev.shape[0]

    def __getitem__(self, k):
        return self.numpy()[k]

    def __iter__(self):
        return iter(self.numpy())

    @property
    def shape(self):
        return tuple(self.dev.shape)


class DevicePrioritySampler:
    """PrioritySampler with the trees in HBM (csrc/ssac_per.hip): same layout, float64 trees whose leaves are
    priority^alpha CORRECTLY ROUNDED in the precision the reference computes them in (float32 power for float32
    priorities, float64 otherwise -- numpy's own power differs between its SIMD and scalar paths in the last bit); the draw
    still consumes numpy's GLOBAL generator on the host (replay.py:166) -- B uniforms travel up, nothing comes back.
    ``update_priorities`` takes device tensors (or numpy arrays) and returns without a synchronisation; the reference's
    assertions on the priorities (replay.py:183-187) are evaluated by the kernel and raised at the next call."""

    def __init__(self, capacity, alpha=0.6, beta=1.0, device=None):
        cap = 1
        while cap < capacity:
            cap *= 2
        self.cap, self.alpha, self.beta, self.device = cap, float(alpha), float(beta), device
        self.sum_dev = torch.zeros(2 * cap, dtype=torch.float64, device=device)
        self.min_dev = torch.full((2 * cap,), float("inf"), dtype=torch.float64, device=device)
        # [0] the largest priority seen, [1] 1.0 once that maximum came from a float32 array (the reference's
        # _max_priority is then an np.float32 and a row pushed at max priority takes a float32 power: csrc/ssac_per.hip)
        self.max_dev = torch.tensor([1.0, 0.0], dtype=torch.float64, device=device)
        self.win = torch.full((cap,), -1, dtype=torch.int32, device=device)
        self.err = torch.zeros(16, dtype=torch.int32).pin_memory()
        self._u_ring, self._u_k, self._u_ev = [None] * 4, 0, [None] * 4

    # ---- host views (checkpoints, tests): a synchronising copy
    @property
    def sum_tree(self):
        return self.sum_dev.cpu().numpy()

    @property
    def min_tree(self):
        return self.min_dev.cpu().numpy()

    @property
    def _max_priority(self):
        return float(self.max_dev.cpu()[0])

    @property
    def _max_priority_is_f32(self):
        return bool(self.max_dev.cpu()[1] != 0)

    def load_state(self, sum_tree, min_tree, max_priority, max_is_f32=None):
        self.sum_dev.copy_(torch.from_numpy(np.ascontiguousarray(sum_tree, np.float64)))
        self.min_dev.copy_(torch.from_numpy(np.ascontiguousarray(min_tree, np.float64)))
        if max_is_f32 is None:   # (a reference-built buffer hands its _max_priority over as the object it is)
            max_is_f32 = isinstance(max_priority, np.float32)
        self.max_dev.copy_(torch.tensor([float(max_priority), 1.0 if max_is_f32 else 0.0], dtype=torch.float64))

    def _raise_pending(self):
        code = int(self.err[0])
        if code:
            self.err[0] = 0
            # (the reference asserts synchronously, replay.py:183-187; here the kernel found it one call ago)
            raise AssertionError("update_priorities: " + ("a priority <= 0" if code == 1 else "an index outside the filled rows"))

    def _dev_i64(self, rows):
        if isinstance(rows, LazyHost):
            return rows.dev
        if torch.is_tensor(rows):
            return rows.to(device=self.device, dtype=torch.int64).contiguous()
        return torch.from_numpy(np.ascontiguousarray(np.atleast_1d(np.asarray(rows)), np.int64)).to(self.device)

    def _assign(self, rows, prio, update_max, n_filled):
        rows_d = self._dev_i64(rows)
        n = rows_d.numel()
        pptr, f64 = 0, 0
        if prio is not None:
            if isinstance(prio, LazyHost):
                prio = prio.dev
            if not torch.is_tensor(prio):
                prio = torch.from_numpy(np.broadcast_to(np.asarray(prio, np.float64), (n,)).copy())
            prio = prio.to(self.device).reshape(-1).contiguous()
            assert prio.numel() == n
            if prio.dtype not in (torch.float32, torch.float64):
                prio = prio.to(torch.float64)
            pptr, f64 = prio.data_ptr(), 1 if prio.dtype == torch.float64 else 0
        check(lib.ssac_per_assign(self.sum_dev.data_ptr(), self.min_dev.data_ptr(), self.cap, rows_d.data_ptr(), n, pptr,
                                  f64, self.alpha, self.max_dev.data_ptr(), 1 if update_max else 0, int(n_filled),
                                  self.err.data_ptr(), self.win.data_ptr(), engine.stream()))
        self._keep = (rows_d, prio)

    def push_rows(self, rows, priorities=None):
        self._raise_pending()
        self._assign(rows, priorities, False, self.cap)

    def update_priorities(self, idxes, priorities, n_filled):
        self._raise_pending()
        assert len(idxes) == len(priorities)
        if not torch.is_tensor(priorities) and not isinstance(priorities, LazyHost):
            # host arrays: the reference's assertions, synchronously, as it makes them (replay.py:183-187)
            priorities = np.asarray(priorities, dtype=np.float64)
            assert np.min(priorities) > 0
            if not torch.is_tensor(idxes) and not isinstance(idxes, LazyHost):
                assert np.min(idxes) >= 0
                assert np.max(idxes) < n_filled
        self._assign(idxes, priorities, True, n_filled)

    def sample_device(self, n_filled, batch_size):
        """(int64 indices, float64 weights) on the device; the uniforms come from numpy's global generator"""
        self._raise_pending()
        u = np.random.random(size=batch_size)
        k = self._u_k = (self._u_k + 1) % 4
        if self._u_ev[k] is not None:
            self._u_ev[k].synchronize()
        if self._u_ring[k] is None or self._u_ring[k].numel() < batch_size:
            self._u_ring[k] = torch.empty(max(batch_size, 1024), dtype=torch.float64).pin_memory()
        self._u_ring[k][:batch_size].copy_(torch.from_numpy(u))
        u_dev = self._u_ring[k][:batch_size].to(self.device, non_blocking=True)
        self._u_ev[k] = ev = torch.cuda.Event()
        ev.record()
        idx = torch.empty(batch_size, dtype=torch.int64, device=self.device)
        w = torch.empty(batch_size, dtype=torch.float64, device=self.device)
        check(lib.ssac_per_sample(self.sum_dev.data_ptr(), self.min_dev.data_ptr(), self.cap, int(n_filled),
                                  u_dev.data_ptr(), batch_size, self.beta, idx.data_ptr(), w.data_ptr(), engine.stream()))
        return idx, w

    def sample(self, n_filled, batch_size):
        idx, w = self.sample_device(n_filled, batch_size)
        return idx.cpu().numpy(), w.cpu().numpy()


class _PinnedRing:
    """host staging for small per-update inputs: copy through a ring of pinned buffers straight into a
    FIXED device buffer (graph replay reads it), guarding slot reuse with events."""

    def __init__(self, nbytes, slots=8):
        self.bufs = [torch.empty(nbytes, dtype=torch.uint8).pin_memory() for _ in range(slots)]
        self.events = [None] * slots
        self.k = 0

    def push(self, host_bytes_tensor, dst_dev_u8):
        self.k = (self.k + 1) % len(self.bufs)
        ev = self.events[self.k]
        if ev is not None:
            ev.synchronize()
        buf = self.bufs[self.k]
        buf[:host_bytes_tensor.numel()].copy_(host_bytes_tensor)
        dst_dev_u8.copy_(buf[:dst_dev_u8.numel()], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.events[self.k] = ev


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

    PACKED_MAX_BYTES = 8 << 20  # pushes up to this size travel as ONE pinned staging buffer + ONE scatter launch

    def _fields(self):
        f = self.__dict__.get("_field_list")
        if f is None:
            f = [(self.s_stack[k], self.s_dtypes[k], ("s", k)) for k in self.s_stack]
            f += [(self.s1_stack[k], self.s_dtypes[k], ("s1", k)) for k in self.s1_stack]
            f += [(self.action_stack, torch.float32, ("a", None)), (self.reward_stack, torch.float32, ("r", None)),
                  (self.done_stack, torch.uint8, ("d", None))]
            self._field_list = f
        return f

    def add(self, s, a, r, s1, d):
        """ReplayBufferStorage.add (replay.py:48-60): rows arange(next, next + k) % size.  The k transitions are packed
        field by field into a pinned staging buffer, cross the bus in ONE asynchronous copy and are scattered into the
        ring arrays by ONE launch (ssac_replay_push); the caller's thread never waits for the device."""
        a = np.asarray(a)
        num = len(a) if a.ndim > 1 else 1
        R = np.arange(self._next_idx, self._next_idx + num) % self.size
        fields = self._fields()
        host = {("a", None): a.astype(np.float32, copy=False), ("r", None): np.asarray(r, dtype=np.float32),
                ("d", None): np.asarray(d).astype(np.uint8)}
        for label in s:
            np_dt = np.uint8 if self.s_dtypes[label] == torch.uint8 else np.float32
            host[("s", label)] = np.asarray(s[label]).astype(np_dt, copy=False)
            host[("s1", label)] = np.asarray(s1[label]).astype(np_dt, copy=False)
        row_bytes = [int(np.prod(t.shape[1:])) * t.element_size() for t, _, _ in fields]
        offs, total = [], 0
        for rb in row_bytes:
            offs.append(total)
            total += (num * rb + 15) // 16 * 16
        if total > self.PACKED_MAX_BYTES or len(fields) > 12:
            rows = torch.from_numpy(R).to(self.device)   # bulk loads (load_experience): per-field copies
            for (t, dt, key), _ in zip(fields, row_bytes):
                self._put(t, rows, host[key], dt)
        else:
            stage = self._staging(total)
            sv = stage.numpy()
            for (t, dt, key), rb, off in zip(fields, row_bytes, offs):
                sv[off:off + num * rb] = np.ascontiguousarray(host[key]).reshape(-1).view(np.uint8)[:num * rb]
            dev = self._dev_staging(total)
            dev[:total].copy_(stage[:total], non_blocking=True)
            self._stage_events[self._stage_k] = ev = torch.cuda.Event()
            ev.record()
            tab = (_lib.PushField * len(fields))(*[_lib.PushField(t.data_ptr(), rb, off)
                                                   for (t, _, _), rb, off in zip(fields, row_bytes, offs)])
            check(lib.ssac_replay_push(tab, len(fields), dev.data_ptr(), num, self._next_idx, self.size,
                                       engine.stream()))
        self._max_filled = min(max(self._next_idx + num, self._max_filled), self.size)
        self._next_idx = (self._next_idx + num) % self.size
        return R

    def _staging(self, nbytes):
        """next buffer of a small ring of pinned staging buffers (slot reuse guarded by an event)"""
        ring = self.__dict__.setdefault("_stage_ring", [None] * 4)
        evs = self.__dict__.setdefault("_stage_events", [None] * 4)
        self._stage_k = k = (self.__dict__.get("_stage_k", -1) + 1) % 4
        if evs[k] is not None:
            evs[k].synchronize()
        if ring[k] is None or ring[k].numel() < nbytes:
            ring[k] = torch.empty(max(nbytes, 1 << 16), dtype=torch.uint8).pin_memory()
        return ring[k]

    def _dev_staging(self, nbytes):
        ring = self.__dict__.setdefault("_dev_stage_ring", [None] * 4)
        k = self._stage_k
        if ring[k] is None or ring[k].numel() < nbytes:
            ring[k] = torch.empty(max(nbytes, 1 << 16), dtype=torch.uint8, device=self.device)
        return ring[k]

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
        # prioritised sampling: trees in HBM on a GPU (csrc/ssac_per.hip); the float64 host trees (PrioritySampler) are
        # what a CPU-resident buffer uses and what the tests check the device trees against
        dev = torch.device(self.device) if not isinstance(self.device, torch.device) else self.device
        self._per = (DevicePrioritySampler(size, alpha, beta, dev) if dev.type == "cuda"
                     else PrioritySampler(size, alpha, beta))

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
        if engine.CAPTURE is not None:  # graph capture: fixed-address index buffer, no draw
            return engine.CAPTURE.idx_cpu, engine.CAPTURE.idx_dev
        self.total_sample_calls += 1
        idx = rng.draw_indices(len(self._storage), batch_size)
        return idx, self._stager.upload(idx)

    def draw_per_indices(self, batch_size):
        """prioritised draw of replay.py:163-177 without the gather: (host view of the int64 indices, device indices,
        float64 importance weights).  With the trees on the device nothing synchronises: the host view is lazy and the
        weights are a device tensor."""
        assert self._per is not None, "this buffer was built without prioritised sampling"
        self.total_sample_calls += 1
        if isinstance(self._per, DevicePrioritySampler):
            idx_dev, w_dev = self._per.sample_device(len(self._storage), batch_size)
            return LazyHost(idx_dev), idx_dev, w_dev
        idxes, weights = self._per.sample(len(self._storage), batch_size)
        idx = torch.from_numpy(idxes)
        return idx, self._stager.upload(idx), weights

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

    def sample(self, batch_size):
        """prioritised draw (replay.py:171-177): (batch, float64 importance weights, indices)."""
        self.total_sample_calls += 1
        if isinstance(self._per, DevicePrioritySampler):
            idx_dev, w_dev = self._per.sample_device(len(self._storage), batch_size)
            return self.gather(idx_dev, batch_size), w_dev.cpu(), idx_dev.cpu().numpy()
        idxes, weights = self._per.sample(len(self._storage), batch_size)
        idx_dev = self._stager.upload(torch.from_numpy(idxes))
        return self.gather(idx_dev, batch_size), torch.from_numpy(weights), idxes

    @property
    def per_on_device(self):
        return isinstance(self._per, DevicePrioritySampler)

    def update_priorities(self, idxes, priorities):
        self._per.update_priorities(idxes, priorities, len(self._storage))


class NStepFolder:
    """The n-step fold of the collection loop (main.py:335-369, learning_utils.py:121-153) in front of ``buffer.push``:
    keep the last ``n_step`` raw transitions; once the window is full, pop the oldest, add the discounted rewards of
    the others (``r += gamma ** (i + 1) * r_i`` in double precision, exactly the reference's Python arithmetic), take
    the newest transition's next state and termination flag, and push ONE n-step transition -- which the storage sends
    to the device as one packed asynchronous copy + one scatter launch (ReplayBufferStorage.add).  ``clear()`` at every
    environment reset, as the reference clears its deque."""

    def __init__(self, buffer, n_step, gamma):
        from collections import deque
        assert n_step >= 1
        self.buffer, self.n_step, self.gamma = buffer, int(n_step), float(gamma)
        self.window = deque([], maxlen=self.n_step)

    def clear(self):
        self.window.clear()

    def add(self, state, action, reward, next_state, terminated, done=False):
        """one environment transition; returns the ring rows written (None while the window fills up)"""
        self.window.append((state, action, reward, next_state, terminated))
        if len(self.window) < self.window.maxlen:
            return None
        s, a, r, s1, d = self.window.popleft()
        for i, trans in enumerate(self.window):
            *_, r_i, s1, d = trans
            r = r + (self.gamma ** (i + 1)) * r_i
        return self.buffer.push(s, a, r, s1, done=d, terminate_traj=done)
