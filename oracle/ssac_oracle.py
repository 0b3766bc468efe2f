"""CPU oracle for the off-policy actor-critic update path.  TEST INFRASTRUCTURE ONLY.

This file is a from-scratch CPU restatement (torch-CPU fp32 / numpy) of the update
path of jakegrigsby/super_sac -- the one hot path this repository accelerates.  It
is the *checker* for the HIP kernels and the "port" CPU baseline that ``bench.py``
times; it is never imported by the product package ``super_sac_amd`` (which fails
loudly when its HIP library is missing).  Only ``tests/``, ``__graft_entry__.smoke``
and ``bench.py``'s ``cpu_baseline`` leg may import it.

Parity status: PINNED.  The reference ships no tests or golden vectors
(SURVEY.md section 4), so the pin is made here: ``oracle/gen_golden.py`` imports
the reference *unmodified* in the development container, drives it on seeded
inputs and writes ``tests/golden/*.npz``; ``tests/test_oracle_golden.py`` checks
every function below against those vectors.

Each function cites the reference lines it restates (paths relative to
/root/reference/super_sac).  Weight layout follows ``torch.nn.Linear``:
W is (out, in), y = x W^T + b.
"""
import math
import random as _pyrandom

import numpy as np
import torch
import torch.nn.functional as F

LOG_2PI_HALF = 0.5 * math.log(2.0 * math.pi)
LOG2 = math.log(2.0)


# --------------------------------------------------------------------------------------
# a2: replay index draw.  replay.py:121-126 -> torch.randint(len, (B,)) on the CPU
# default generator (mt19937).  For len < 2**32 torch takes ONE 32-bit output per index
# and reduces it modulo len (ATen random_from_to, 32-bit branch).
# --------------------------------------------------------------------------------------
class MT19937:
    """Matsumoto-Nishimura mt19937, seeded like ``torch.manual_seed`` (init_genrand)."""

    N, M = 624, 397

    def __init__(self, seed):
        st = np.zeros(self.N, dtype=np.uint64)
        st[0] = seed & 0xFFFFFFFF
        for i in range(1, self.N):
            prev = int(st[i - 1])
            st[i] = (1812433253 * (prev ^ (prev >> 30)) + i) & 0xFFFFFFFF
        self.state = st.astype(np.uint32)
        self.pos = self.N

    def _twist(self):
        s = self.state
        upper = np.uint32(0x80000000)
        lower = np.uint32(0x7FFFFFFF)
        mag = np.uint32(0x9908B0DF)
        # the recurrence is sequential over i because s[i+M] may already be updated
        s = s.copy()
        for i in range(self.N):
            y = (s[i] & upper) | (s[(i + 1) % self.N] & lower)
            v = s[(i + self.M) % self.N] ^ (y >> np.uint32(1))
            if y & np.uint32(1):
                v ^= mag
            s[i] = v
        self.state = s
        self.pos = 0

    def raw32(self, n):
        out = np.empty(n, dtype=np.uint32)
        for k in range(n):
            if self.pos >= self.N:
                self._twist()
            y = int(self.state[self.pos])
            self.pos += 1
            y ^= y >> 11
            y ^= (y << 7) & 0x9D2C5680
            y ^= (y << 15) & 0xEFC60000
            y ^= y >> 18
            out[k] = y & 0xFFFFFFFF
        return out


def randint_from_raw32(raw, low, high):
    """torch.randint(low, high, ...) for (high-low) < 2**32: raw % range + low."""
    rng = high - low
    return (raw.astype(np.uint64) % np.uint64(rng)).astype(np.int64) + low


def uniform_indices(seed, buffer_len, batch_size, skip=0):
    """Indices of the (skip/batch_size)-th ``sample_uniform`` call after manual_seed(seed)."""
    g = MT19937(seed)
    raw = g.raw32(skip + batch_size)[skip:]
    return randint_from_raw32(raw, 0, buffer_len)


# --------------------------------------------------------------------------------------
# a1: SoA ring storage.  replay.py:10-95 (ReplayBufferStorage) and :98-137.
# --------------------------------------------------------------------------------------
class ReplayOracle:
    def __init__(self, capacity):
        self.capacity = capacity
        self.obs = None  # label -> (capacity, *shape)
        self.next_obs = None
        self.act = None
        self.rew = None
        self.done = None
        self.next_idx = 0
        self.filled = 0
        self.total_sample_calls = 0

    def __len__(self):
        return self.filled

    def _alloc(self, s, a):
        # replay.py:12-23 -- action/reward f32, done u8, observations keep env dtype
        batched = a.ndim > 1
        a_shape = a.shape[1:] if batched else a.shape
        self.act = np.zeros((self.capacity,) + a_shape, np.float32)
        self.rew = np.zeros((self.capacity, 1), np.float32)
        self.done = np.zeros((self.capacity, 1), np.uint8)
        self.obs, self.next_obs = {}, {}
        for k, v in s.items():
            shp = v.shape[1:] if batched else v.shape
            self.obs[k] = np.zeros((self.capacity,) + shp, v.dtype)
            self.next_obs[k] = np.zeros((self.capacity,) + shp, v.dtype)

    def push(self, s, a, r, s1, d):
        # replay.py:32-60: rows arange(next, next+k) % capacity
        a = np.asarray(a)
        if self.act is None:
            self._alloc(s, a)
        k = len(a) if a.ndim > 1 else 1
        rows = np.arange(self.next_idx, self.next_idx + k) % self.capacity
        for lbl in s:
            self.obs[lbl][rows] = np.asarray(s[lbl]).astype(self.obs[lbl].dtype)
            self.next_obs[lbl][rows] = np.asarray(s1[lbl]).astype(self.obs[lbl].dtype)
        self.act[rows] = a.astype(np.float32)
        self.rew[rows] = np.asarray(r, dtype=np.float32).reshape(k, 1) if k > 1 else np.float32(r)
        self.done[rows] = np.asarray(d).astype(np.uint8).reshape(k, 1) if k > 1 else np.uint8(d)
        self.filled = min(max(self.next_idx + k, self.filled), self.capacity)
        self.next_idx = (self.next_idx + k) % self.capacity
        return rows

    def load_experience(self, s, a, r, s1, d):
        # replay.py:131-137
        r = np.asarray(r)
        d = np.asarray(d)
        if r.ndim < 2:
            r = r[:, None]
        if d.ndim < 2:
            d = d[:, None]
        self.push(s, a, r, s1, d)

    def gather(self, idx):
        # replay.py:66-84 (numpy fancy index), learning_utils.py:186-197 (.float())
        o = {k: torch.from_numpy(v[idx]).float() for k, v in self.obs.items()}
        o1 = {k: torch.from_numpy(v[idx]).float() for k, v in self.next_obs.items()}
        a = torch.from_numpy(self.act[idx]).float()
        if a.dim() < 2:
            a = a.unsqueeze(1)
        r = torch.from_numpy(self.rew[idx]).float()
        d = torch.from_numpy(self.done[idx]).float()
        return o, a, r, o1, d

    def sample_uniform(self, batch_size):
        # replay.py:179-181 + :121-126 (consumes the torch CPU generator)
        self.total_sample_calls += 1
        idx = torch.randint(len(self), (batch_size,))
        return self.gather(idx.numpy()), idx.numpy()


# --------------------------------------------------------------------------------------
# a3 (next): proportional PER.  replay.py:163-190, 285-353.  float64 trees.
# --------------------------------------------------------------------------------------
class PerOracle:
    def __init__(self, capacity, alpha=0.6, beta=1.0):
        cap = 1
        while cap < capacity:
            cap *= 2
        self.cap = cap
        self.alpha, self.beta = alpha, beta
        self.sum = np.zeros(2 * cap, np.float64)
        self.min = np.full(2 * cap, np.inf, np.float64)
        self.max_priority = 1.0

    def _set(self, idx, val):
        idx = np.atleast_1d(np.asarray(idx, dtype=np.int64))
        val = np.broadcast_to(np.asarray(val, np.float64), idx.shape)
        leaf = idx + self.cap
        self.sum[leaf] = val
        self.min[leaf] = val
        node = np.unique(leaf // 2)
        while node.size and node[0] >= 1:
            self.sum[node] = self.sum[2 * node] + self.sum[2 * node + 1]
            self.min[node] = np.minimum(self.min[2 * node], self.min[2 * node + 1])
            if node.size == 1 and node[0] == 1:
                break
            node = np.unique(node // 2)

    def push_rows(self, rows):
        self._set(rows, self.max_priority ** self.alpha)

    def update(self, idx, prios):
        prios = np.asarray(prios, np.float64)
        assert len(idx) == len(prios) and prios.min() > 0
        self._set(idx, prios ** self.alpha)
        self.max_priority = max(self.max_priority, float(prios.max()))

    def prefix_sum(self, n):
        """sum of leaves [0, n-1] (replay.py:164: self._it_sum.sum(0, len-1) is EXCLUSIVE of end)."""
        # SegmentTree.reduce(start, end): end -= 1 -> inclusive end-1.  sum(0, len-1) = leaves [0, len-2].
        return float(self.sum[self.cap: self.cap + n - 1].sum()) if n > 1 else 0.0

    def find_prefixsum_idx(self, mass):
        # replay.py:317-336: vectorised descent from the root
        mass = np.array(mass, np.float64)
        idx = np.ones(len(mass), dtype=np.int64)
        while np.any(idx < self.cap):
            live = idx < self.cap
            left = np.where(live, 2 * idx, idx)
            lv = self.sum[left]
            go_right = live & (lv <= mass)
            mass = np.where(go_right, mass - lv, mass)
            idx = np.where(live, np.where(go_right, left + 1, left), idx)
        return idx - self.cap

    def sample(self, n_filled, batch_size):
        total = self._reduce_sum(0, n_filled - 1)
        mass = np.random.random(size=batch_size) * total
        idx = self.find_prefixsum_idx(mass)
        p_min = self.min[1] / self.sum[1]
        max_w = (p_min * n_filled) ** (-self.beta)
        p = self.sum[self.cap + idx] / self.sum[1]
        w = (p * n_filled) ** (-self.beta) / max_w
        return idx, w

    def _reduce_sum(self, start, end):
        # SegmentTree.reduce semantics (replay.py:251-266): end is exclusive after `end -= 1`
        end -= 1
        return float(self.sum[self.cap + start: self.cap + end + 1].sum())


# --------------------------------------------------------------------------------------
# a5: augmentations.  augmentations.py:20-38 (one randomisation shared by s and s'),
# :214-269 (Drqv2Aug), :165-211 (DrqAug).
# --------------------------------------------------------------------------------------
def drqv2_draw_shift(batch_size, pad=4):
    # augmentations.py:226-231, layout (B,1,1,2) -> [...,0] = x shift, [...,1] = y shift
    return torch.randint(0, 2 * pad + 1, size=(batch_size, 1, 1, 2))


def _linspace_f32(start, end, steps):
    """ATen linspace in fp32: two-sided, and each element is ONE fused multiply-add
    (start + step*i contracts to fma on the CPU build and under nvcc alike)."""
    start = np.float32(start)
    end = np.float32(end)
    step = np.float32((end - start) / np.float32(steps - 1))
    out = np.empty(steps, np.float32)
    half = steps // 2
    for i in range(steps):
        if i < half:
            out[i] = np.float32(np.float64(start) + np.float64(step) * i)
        else:
            out[i] = np.float32(np.float64(end) - np.float64(step) * (steps - 1 - i))
    return out


def drqv2_shift(imgs, shift, pad=4):
    """Drqv2Aug.random_crop restated: replicate pad, fp32 sampling grid, bilinear taps with
    zero padding (F.grid_sample, align_corners=False), then clamp [0,255] (augmentations.py:233-263).
    imgs (B,C,h,h) float32, shift (B,1,1,2) integer tensor."""
    n, c, h, w = imgs.shape
    assert h == w and n == shift.shape[0]
    hp = h + 2 * pad
    x = imgs.numpy().astype(np.float32)
    padded = np.pad(x, ((0, 0), (0, 0), (pad, pad), (pad, pad)), mode="edge")
    # eps = 1.0/(h+2*pad) is a Python double; linspace rounds its end points to fp32
    ar = _linspace_f32(np.float32(-1.0 + 1.0 / hp), np.float32(1.0 - 1.0 / hp), hp)[:h]
    sh = shift.numpy().reshape(n, 2).astype(np.float32) * np.float32(2.0 / hp)
    out = np.zeros((n, c, h, w), np.float32)
    size = np.float32(hp)
    for b in range(n):
        gx = (ar + sh[b, 0]).astype(np.float32)  # along width
        gy = (ar + sh[b, 1]).astype(np.float32)  # along height
        ix = ((gx + np.float32(1.0)) * size - np.float32(1.0)) / np.float32(2.0)
        iy = ((gy + np.float32(1.0)) * size - np.float32(1.0)) / np.float32(2.0)
        ix0 = np.floor(ix)
        iy0 = np.floor(iy)
        wx1 = (ix - ix0).astype(np.float32)
        wy1 = (iy - iy0).astype(np.float32)
        wx0 = (np.float32(1.0) - wx1).astype(np.float32)
        wy0 = (np.float32(1.0) - wy1).astype(np.float32)
        ix0 = ix0.astype(np.int64)
        iy0 = iy0.astype(np.int64)
        acc = np.zeros((c, h, w), np.float32)
        for dy, wy in ((0, wy0), (1, wy1)):
            yy = iy0 + dy
            vy = (yy >= 0) & (yy < hp)
            yyc = np.clip(yy, 0, hp - 1)
            for dx, wx in ((0, wx0), (1, wx1)):
                xx = ix0 + dx
                vx = (xx >= 0) & (xx < hp)
                xxc = np.clip(xx, 0, hp - 1)
                tap = padded[b][:, yyc][:, :, xxc]
                wgt = (wy * vy)[:, None] * (wx * vx)[None, :]
                acc += tap * wgt.astype(np.float32)[None]
        out[b] = acc
    return torch.from_numpy(np.clip(out, 0.0, 255.0))


def drq_draw_offsets(batch_size, pad=4):
    # augmentations.py:184-186 -- w1 first, then h1, exclusive upper bound 2*pad
    w1 = torch.randint(0, pad * 2, (batch_size,))
    h1 = torch.randint(0, pad * 2, (batch_size,))
    return w1, h1


def drq_crop(imgs, w1, h1, pad=4, noise=None):
    """DrqAug.__call__ restated: reflection pad, per-sample integer crop, optional additive
    N(0,1) noise, clamp (augmentations.py:188-204)."""
    n, c, h, w = imgs.shape
    padded = np.pad(imgs.numpy(), ((0, 0), (0, 0), (pad, pad), (pad, pad)), mode="reflect")
    out = np.empty((n, c, h, w), np.float32)
    for i in range(n):
        y0, x0 = int(h1[i]), int(w1[i])
        out[i] = padded[i, :, y0:y0 + h, x0:x0 + w]
    out = torch.from_numpy(out)
    if noise is not None:
        out = out + noise
    return out.clamp(0, 255.0)


class AugOracle:
    """kind in {"identity","drqv2","drq","drq_nonoise"}; one draw per call, shared by all
    batches passed (augmentations.py:25-38)."""

    def __init__(self, kind, batch_size, pad=4):
        self.kind, self.batch_size, self.pad = kind, batch_size, pad
        self.last = None
        self.forced = None  # list of recorded draws to replay instead of the torch generator
        # the reference constructors draw once at build time (augmentations.py:180,221)
        if kind == "drqv2":
            drqv2_draw_shift(batch_size, pad)
        elif kind in ("drq", "drq_nonoise"):
            drq_draw_offsets(batch_size, pad)

    def __call__(self, *obs_dicts, noises=None):
        if self.kind == "identity":
            outs = [{k: v.clone() for k, v in d.items()} for d in obs_dicts]
            return tuple(outs) if len(outs) > 1 else outs[0]
        if self.kind == "drqv2":
            shift = self.forced.pop(0) if self.forced else drqv2_draw_shift(self.batch_size, self.pad)
            self.last = shift
            f = lambda x, nz: drqv2_shift(x, shift, self.pad)
        else:
            w1, h1 = drq_draw_offsets(self.batch_size, self.pad)
            self.last = (w1, h1)
            if self.kind == "drq":
                f = lambda x, nz: drq_crop(x, w1, h1, self.pad,
                                           noise=torch.randn_like(x) if nz is None else nz)
            else:
                f = lambda x, nz: drq_crop(x, w1, h1, self.pad)
        outs = []
        ni = 0
        for d in obs_dicts:
            o = {}
            for k, v in d.items():
                nz = None if noises is None else noises[ni]
                ni += 1
                o[k] = f(v, nz)
            outs.append(o)
        return tuple(outs) if len(outs) > 1 else outs[0]


def sample_move_and_augment(buffer, batch_size, augmenter, aug_mix, idx=None):
    """learning_utils.py:174-214 restated for the uniform path (per=False)."""
    assert len(buffer) >= batch_size
    if idx is None:
        (oo, a, r, oo1, d), idx = buffer.sample_uniform(batch_size)
    else:
        oo, a, r, oo1, d = buffer.gather(idx)
    imp = torch.ones(1)
    k = int(batch_size * aug_mix)
    ao, ao1 = augmenter(oo, oo1)
    o = {x: y.clone() for x, y in oo.items()}
    o1 = {x: y.clone() for x, y in oo1.items()}
    for lbl in o:
        o[lbl][:k] = ao[lbl][:k]
        o1[lbl][:k] = ao1[lbl][:k]
    return {"primary_batch": (o, a, r, o1, d), "augmented_obs": (ao, ao1),
            "original_obs": (oo, oo1), "priority_idxs": idx, "imp_weights": imp}


# --------------------------------------------------------------------------------------
# a6/a8/a9: networks.  nets/mlps.py, nets/cnns.py, nets/distributions.py
# --------------------------------------------------------------------------------------
def mlp3(p, x):
    """relu(fc1) -> relu(fc2) -> out (mlps.py:32-35, 123-129).  Returns (y, h2)."""
    h1 = F.relu(F.linear(x, p["w1"], p["b1"]))
    h2 = F.relu(F.linear(h1, p["w2"], p["b2"]))
    return F.linear(h2, p["w3"], p["b3"]), h2


def critic_q(p, s, a=None):
    """ContinuousCritic: cat(s,a) (mlps.py:123-124); DiscreteCritic: s only (mlps.py:180)."""
    x = s if a is None else torch.cat((s, a), dim=-1)
    return mlp3(p, x)[0]


def tanh_normal_params(out, lo, hi):
    # distributions.py:9-15
    mu, raw = out.chunk(2, dim=-1)
    log_std = lo + 0.5 * (hi - lo) * (torch.tanh(raw) + 1.0)
    return mu, log_std


def tanh_normal_sample(out, lo, hi, eps):
    """a = tanh(mu + sigma*eps); log pi with the CACHED pre-tanh value u
    (distributions.py:64-104; TransformedDistribution.log_prob = base.log_prob(u) - ladj)."""
    mu, log_std = tanh_normal_params(out, lo, hi)
    std = log_std.exp()
    u = mu + std * eps
    a = torch.tanh(u)
    # Normal.log_prob(u) = -((u-mu)^2)/(2 var) - log_std - log(sqrt(2pi))
    base = -((u - mu) ** 2) / (2.0 * std * std) - log_std - LOG_2PI_HALF
    ladj = 2.0 * (LOG2 - u - F.softplus(-2.0 * u))
    logp = (base - ladj).sum(-1, keepdim=True)
    return a, logp


def big_pixel_encoder(p, obs):
    # cnns.py:59-69
    x = obs / 255.0 - 0.5
    x = F.relu(F.conv2d(x, p["c1w"], p["c1b"], stride=2))
    x = F.relu(F.conv2d(x, p["c2w"], p["c2b"]))
    x = F.relu(F.conv2d(x, p["c3w"], p["c3b"]))
    x = F.relu(F.conv2d(x, p["c4w"], p["c4b"]))
    x = x.reshape(x.size(0), -1)
    x = F.linear(x, p["fcw"], p["fcb"])
    x = F.layer_norm(x, (x.shape[-1],), p["lnw"], p["lnb"], 1e-5)
    return torch.tanh(x)


def small_pixel_encoder(p, obs):
    # cnns.py:96-103
    x = obs / 255.0
    x = F.relu(F.conv2d(x, p["c1w"], p["c1b"], stride=4))
    x = F.relu(F.conv2d(x, p["c2w"], p["c2b"], stride=2))
    x = F.relu(F.conv2d(x, p["c3w"], p["c3b"]))
    x = x.reshape(x.size(0), -1)
    return F.linear(x, p["fcw"], p["fcb"])


def encode(enc, obs_dict):
    """enc = {"kind": "identity"|"big"|"small", "key": label, "p": params}."""
    x = obs_dict[enc["key"]]
    if enc["kind"] == "identity":
        return x  # experiments/gym/train_gym.py:18-28
    if enc["kind"] == "big":
        return big_pixel_encoder(enc["p"], x)
    return small_pixel_encoder(enc["p"], x)


# --------------------------------------------------------------------------------------
# a12: PopArt.  popart.py:8-59
# --------------------------------------------------------------------------------------
class PopArtOracle:
    def __init__(self, beta=1e-4, min_steps=1000, init_nu=0.0):
        self.mu = torch.zeros(1)
        self.nu = torch.ones(1) * init_nu
        self.beta = beta
        self.w = torch.ones(1)
        self.b = torch.zeros(1)
        self.t = 1
        self.min_steps = min_steps
        self.stable = False

    @property
    def sigma(self):
        return (torch.sqrt(self.nu - self.mu ** 2) + 1e-5).clamp(1e-4, 1e6)

    def normalize(self, v):
        return (v - self.mu) / self.sigma

    def update_stats(self, v):
        self.t += 1
        old_sigma, old_mu = self.sigma, self.mu
        beta_t = self.beta / (1.0 - (1.0 - self.beta) ** self.t)
        self.mu = (1.0 - beta_t) * self.mu + beta_t * v.mean()
        self.nu = (1.0 - beta_t) * self.nu + beta_t * (v ** 2).mean()
        self.stable = (self.t > self.min_steps) and bool(((1.0 - old_sigma) / self.sigma) <= 0.1)
        if self.stable:
            self.w = self.w * (old_sigma / self.sigma)
            self.b = (old_sigma * self.b + old_mu - self.mu) / self.sigma

    def __call__(self, x, normalized=True):
        y = self.w * x + self.b
        return y if normalized else self.sigma * y + self.mu

    def state(self):
        return dict(mu=float(self.mu), nu=float(self.nu), w=float(self.w), b=float(self.b),
                    t=self.t, sigma=float(self.sigma))


# --------------------------------------------------------------------------------------
# a17: Adam + clip_grad_norm_ (arithmetic of torch.optim.Adam, main.py:188-239)
# --------------------------------------------------------------------------------------
class AdamOracle:
    """torch.optim.Adam (no amsgrad, coupled L2) restated with foreach ops:
    m = lerp(m, g, 1-b1); v = b2 v + (1-b2) g^2;
    p -= (lr / (1-b1^t)) * m / (sqrt(v)/sqrt(1-b2^t) + eps)."""

    def __init__(self, params, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        self.params = list(params)
        self.lr, self.b1, self.b2, self.eps, self.wd = lr, betas[0], betas[1], eps, weight_decay
        self.m = [torch.zeros_like(p) for p in self.params]
        self.v = [torch.zeros_like(p) for p in self.params]
        self.t = [0 for _ in self.params]

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    @torch.no_grad()
    def step(self):
        idx = [i for i, p in enumerate(self.params) if p.grad is not None]
        if not idx:
            return
        ps = [self.params[i] for i in idx]
        gs = [self.params[i].grad for i in idx]
        ms = [self.m[i] for i in idx]
        vs = [self.v[i] for i in idx]
        for i in idx:
            self.t[i] += 1
        if self.wd:
            gs = torch._foreach_add(gs, ps, alpha=self.wd)
        torch._foreach_lerp_(ms, gs, 1.0 - self.b1)
        torch._foreach_mul_(vs, self.b2)
        torch._foreach_addcmul_(vs, gs, gs, 1.0 - self.b2)
        # all params of one optimizer share the step count in this code base
        t = self.t[idx[0]]
        bc1 = 1.0 - self.b1 ** t
        bc2_sqrt = math.sqrt(1.0 - self.b2 ** t)
        denom = torch._foreach_sqrt(vs)
        torch._foreach_div_(denom, bc2_sqrt)
        torch._foreach_add_(denom, self.eps)
        torch._foreach_addcdiv_(ps, ms, denom, -(self.lr / bc1))


def clip_grad_norm(params, max_norm):
    """torch.nn.utils.clip_grad_norm_: coef = max_norm/(total+1e-6), clamped to 1."""
    gs = [p.grad for p in params if p.grad is not None]
    if not gs:
        return torch.zeros(())
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g) for g in gs]))
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in gs:
        g.mul_(coef)
    return total


def grad_norm(params):
    # learning_utils.py:95-106
    tot = 0.0
    for p in params:
        if p.grad is None:
            continue
        tot += float(p.grad.norm(2)) ** 2
    return tot ** 0.5


# --------------------------------------------------------------------------------------
# Agent state container (restates the *data* of agent.py:43-130, not its code)
# --------------------------------------------------------------------------------------
MLP_KEYS = ("w1", "b1", "w2", "b2", "w3", "b3")


def make_mlp(rng, in_dim, hidden, out_dim, w_scale=1.0):
    """Seeded synthetic weights (numpy RandomState -> reproducible everywhere)."""
    def lin(o, i):
        w = (rng.standard_normal((o, i)) * (w_scale / math.sqrt(i))).astype(np.float32)
        b = (rng.standard_normal((o,)) * 0.05).astype(np.float32)
        return torch.from_numpy(w), torch.from_numpy(b)
    w1, b1 = lin(hidden, in_dim)
    w2, b2 = lin(hidden, hidden)
    w3, b3 = lin(out_dim, hidden)
    return {"w1": w1, "b1": b1, "w2": w2, "b2": b2, "w3": w3, "b3": b3}


def make_conv_encoder(rng, kind, channels, emb):
    def conv(o, i, k):
        w = (rng.standard_normal((o, i, k, k)) * (1.0 / math.sqrt(i * k * k))).astype(np.float32)
        b = (rng.standard_normal((o,)) * 0.05).astype(np.float32)
        return torch.from_numpy(w), torch.from_numpy(b)
    p = {}
    if kind == "big":
        p["c1w"], p["c1b"] = conv(32, channels, 3)
        for i in (2, 3, 4):
            p[f"c{i}w"], p[f"c{i}b"] = conv(32, 32, 3)
        flat = 32 * 35 * 35
    else:
        p["c1w"], p["c1b"] = conv(32, channels, 8)
        p["c2w"], p["c2b"] = conv(64, 32, 4)
        p["c3w"], p["c3b"] = conv(64, 64, 3)
        flat = 64 * 7 * 7
    p["fcw"] = torch.from_numpy((rng.standard_normal((emb, flat)) / math.sqrt(flat)).astype(np.float32))
    p["fcb"] = torch.from_numpy((rng.standard_normal((emb,)) * 0.05).astype(np.float32))
    if kind == "big":
        p["lnw"] = torch.from_numpy((1.0 + 0.1 * rng.standard_normal((emb,))).astype(np.float32))
        p["lnb"] = torch.from_numpy((0.05 * rng.standard_normal((emb,))).astype(np.float32))
    return p


class AgentOracle:
    """encoder + E actors + E x N critics + E popart layers (agent.py:86-104)."""

    def __init__(self, *, state_dim, act_dim, hidden, num_critics, ensemble_size=1, discrete=False,
                 actor_kind="stochastic", log_std_low=-10.0, log_std_high=2.0,
                 popart=False, encoder=None, seed=0):
        rng = np.random.RandomState(seed)
        self.discrete = discrete
        self.actor_kind = "discrete" if discrete else actor_kind
        self.lo, self.hi = log_std_low, log_std_high
        self.E, self.N = ensemble_size, num_critics
        self.state_dim, self.act_dim, self.hidden = state_dim, act_dim, hidden
        self.encoder = encoder if encoder is not None else {"kind": "identity", "key": "obs", "p": {}}
        a_out = act_dim if (discrete or actor_kind == "deterministic") else 2 * act_dim
        c_in = state_dim if discrete else state_dim + act_dim
        c_out = act_dim if discrete else 1
        self.actors = [make_mlp(rng, state_dim, hidden, a_out) for _ in range(self.E)]
        self.critics = [[make_mlp(rng, c_in, hidden, c_out) for _ in range(self.N)]
                        for _ in range(self.E)]
        self.popart = [PopArtOracle() if popart else False for _ in range(self.E)]

    # parameter lists in the reference's optimizer order (main.py:188-211)
    def critic_params(self):
        return [c[k] for ens in self.critics for c in ens for k in MLP_KEYS]

    def actor_params(self):
        return [a[k] for a in self.actors for k in MLP_KEYS]

    def encoder_params(self):
        return list(self.encoder["p"].values())

    def requires_grad_(self, flag=True):
        for p in self.critic_params() + self.actor_params() + self.encoder_params():
            p.requires_grad_(flag)
        return self

    def clone(self):
        """copy.deepcopy(agent) for the target network (main.py:321)."""
        import copy
        other = copy.copy(self)
        cp = lambda d: {k: v.detach().clone() for k, v in d.items()}
        other.actors = [cp(a) for a in self.actors]
        other.critics = [[cp(c) for c in ens] for ens in self.critics]
        other.encoder = dict(self.encoder, p=cp(self.encoder["p"]))
        other.popart = [copy.deepcopy(p) for p in self.popart]
        return other


def ensemble_q(critics, s, a, subset_ids=None, return_min=True):
    """agent.Critic.forward (agent.py:22-40).  subset_ids are drawn by the caller with
    ``random.sample(range(N), k)`` (agent.py:29) so the Python RNG stream is explicit."""
    nets = critics if subset_ids is None else [critics[j] for j in subset_ids]
    preds = [critic_q(p, s, a) for p in nets]
    if return_min:
        return torch.stack(preds, dim=0).min(0).values
    return tuple(preds)


def gaussian_exploration_noise(action, scale, clip, noise, low=-1.0, high=1.0, eps=1e-6):
    """GaussianExplorationNoise.sample, torch branch (learning_utils.py:48-59).
    ``noise`` is the standard-normal draw.  Straight-through clamp."""
    n = scale * noise
    if clip is not None:
        n = n.clamp(-clip, clip)
    na = action + n
    clamped = na.clamp(low + eps, high - eps)
    return na - na.detach() + clamped.detach()


# --------------------------------------------------------------------------------------
# a10: TD target.  learning_utils.py:298-354
# --------------------------------------------------------------------------------------
def compute_td_targets(logs, batch, agent, target_agent, i, subset_ids, log_alpha, pop, gamma,
                       eps=None, noise_scale=None, noise_clip=None, noise=None):
    o, a, r, o1, d = batch
    actor = agent.actors[i]
    popart = agent.popart[i]
    with torch.no_grad():
        s1 = encode(target_agent.encoder, o1)
        out = mlp3(actor, s1)[0]
        if agent.discrete:
            logits = out
            q1 = ensemble_q(target_agent.critics[i], s1, None, subset_ids)
            logp = torch.log_softmax(logits, dim=-1)
            probs = torch.softmax(logits, dim=-1)
            bonus = log_alpha.exp() * logp
            val = (probs * (q1 - bonus)).sum(-1, keepdim=True)
            a1 = probs
        else:
            if agent.actor_kind == "deterministic":
                a1 = torch.tanh(out)  # ContinuousDeterministic.sample() == loc (distributions.py:113)
                logp1 = None
            else:
                if eps is None:
                    eps = torch.randn(out.shape[0], out.shape[1] // 2)
                a1, logp1 = tanh_normal_sample(out, agent.lo, agent.hi, eps)
            if noise_scale is not None:
                if noise is None:
                    noise = torch.randn(*a1.shape)
                a1 = gaussian_exploration_noise(a1, noise_scale, noise_clip, noise)
                bonus = torch.zeros(1)
            else:
                if logp1 is None:  # Normal(loc, 1e-4).log_prob(loc)
                    logp1 = torch.full_like(a1, -math.log(1e-4) - LOG_2PI_HALF).sum(-1, keepdim=True)
                bonus = log_alpha.exp() * logp1
            q1 = ensemble_q(target_agent.critics[i], s1, a1, subset_ids)
            val = q1 - bonus
        if popart and pop:
            val = popart(val, normalized=False)
        td = r + gamma * (1.0 - d) * val
        if popart:
            popart.update_stats(td)
            td = popart.normalize(td)
    logs[f"td_targets/mean_td_target_{i}"] = td.mean().item()
    logs[f"td_targets/std_td_target_{i}"] = td.std().item()
    logs[f"td_targets/entropy_bonus_{i}"] = bonus.mean().item()
    return td, (s1, a1)


# --------------------------------------------------------------------------------------
# a11: SUNRISE / softmax backup weights.  learning_utils.py:357-398
# --------------------------------------------------------------------------------------
def compute_backup_weights(logs, batch, agent, target_agent, weight_type, temp, batch_size,
                           eps_list=None, cat_list=None):
    if weight_type is None or temp is None or agent.E == 1:
        return 1.0
    o, a, _, o1, _ = batch
    with torch.no_grad():
        if weight_type == "sunrise":
            s = encode(target_agent.encoder, o)
            if agent.discrete:
                qs = [ensemble_q(c, s, None).gather(-1, a.long()) for c in target_agent.critics]
            else:
                qs = [ensemble_q(c, s, a) for c in target_agent.critics]
            w = torch.sigmoid(-torch.stack(qs, 0).std(0) * temp) + 0.5
        else:
            s1 = encode(target_agent.encoder, o1)
            qs = []
            for k, (actor, crit) in enumerate(zip(agent.actors, agent.critics)):
                out = mlp3(actor, s1)[0]
                if agent.discrete:
                    # a1 = Categorical(logits).sample() (learning_utils.py:388): torch.multinomial stream, so the
                    # fixtures record the sampled actions (cat_list) instead of the raw draws
                    a1 = (torch.distributions.Categorical(logits=out).sample() if cat_list is None
                          else cat_list[k])
                    qs.append(ensemble_q(crit, s1, None).gather(-1, a1.long().view(-1, 1)))
                    continue
                e = torch.randn(out.shape[0], out.shape[1] // 2) if eps_list is None else eps_list[k]
                a1, _ = tanh_normal_sample(out, agent.lo, agent.hi, e)
                qs.append(ensemble_q(crit, s1, a1))
            w = batch_size * F.softmax(-torch.stack(qs, 0).std(0) * temp, dim=0)
    logs["bellman_weights/mean"] = w.mean().item()
    logs["bellman_weights/max"] = w.max().item()
    logs["bellman_weights/min"] = w.min().item()
    logs["bellman_weights/std"] = w.std().item()
    return w


# --------------------------------------------------------------------------------------
# a13: critic update.  learning.py:18-141
# --------------------------------------------------------------------------------------
def critic_update(buffer, agent, target_agent, critic_opt, encoder_opt, log_alphas, batch_size, gamma,
                  critic_clip, encoder_clip, target_n, temp, weight_type, pop, augmenter,
                  aug_mix=0.0, noise_scale=None, noise_clip=None, py_rng=_pyrandom,
                  idx_list=None, eps_list=None, noise_list=None, subset_list=None, dr3_coeff=0.0,
                  bw_eps_list=None, bw_cat_list=None, grad_pick=None, encoder_lambda=0.0):
    """One gradient update of every critic of every ensemble member.
    encoder_lambda > 0 adds the encoder invariance constraint (learning.py:114-117, learning_utils.py:401-409) on the
    LAST member's batch, after the division by E*N: the Frobenius norm of enc(augmented obs) - enc(original obs),
    the latter without gradient.
    dr3_coeff > 0 adds the DR3 feature co-adaptation term (learning.py:100-108): the fc2 features of every critic
    on (s, a) dotted with its features on (s', a'), mean over critics and batch -- inside the member loop, i.e.
    BEFORE the division by E*N.
    Host RNG order per member (matches the reference run): index draw (torch CPU) ->
    augmentation draw (torch CPU) -> action noise (eps) -> REDQ subset (python random)."""
    logs = {}
    loss = 0.0
    dicts = []
    td_error = None
    for i in range(agent.E):
        rd = sample_move_and_augment(buffer, batch_size, augmenter, aug_mix,
                                     idx=None if idx_list is None else idx_list[i])
        o, a, r, o1, d = rd["primary_batch"]
        # eps must be drawn BEFORE the subset to keep the reference's stream order explicit:
        # a_dist.sample() (lu:330) precedes target_critic(..., subset=n) (lu:339).
        eps = None if eps_list is None else eps_list[i]
        if eps is None and not agent.discrete and agent.actor_kind == "stochastic":
            eps = torch.randn(batch_size, agent.act_dim)
        noise = None if noise_list is None else noise_list[i]
        if noise is None and noise_scale is not None:
            noise = torch.randn(batch_size, agent.act_dim)
        subset = (subset_list[i] if subset_list is not None
                  else py_rng.sample(range(agent.N), k=target_n))
        td, (s1, a1) = compute_td_targets(logs, (o, a, r, o1, d), agent, target_agent, i, subset,
                                          log_alphas[i], pop, gamma, eps=eps, noise_scale=noise_scale,
                                          noise_clip=noise_clip, noise=noise)
        w = compute_backup_weights(logs, (o, a, r, o1, d), agent, target_agent, weight_type, temp,
                                   batch_size, eps_list=None if bw_eps_list is None else bw_eps_list[i],
                                   cat_list=None if bw_cat_list is None else bw_cat_list[i])
        s = encode(agent.encoder, o)
        for p in agent.critics[i]:
            q = critic_q(p, s, None if agent.discrete else a)
            if agent.discrete:
                q = q.gather(-1, a.long())
            if agent.popart[i] and pop:
                q = agent.popart[i](q)
            td_error = td - q
            loss = loss + (w * rd["imp_weights"] * td_error ** 2).mean()
        if dr3_coeff > 0:
            xa = s if agent.discrete else torch.cat((s, a), dim=-1)
            x1 = s1 if agent.discrete else torch.cat((s1, a1), dim=-1)
            f_sa = torch.stack([mlp3(p, xa)[1] for p in agent.critics[i]], 0)
            f_s1 = torch.stack([mlp3(p, x1)[1] for p in agent.critics[i]], 0)
            fca = (f_sa * f_s1).sum(-1).mean()
            logs[f"dr3_dotproduct_{i}"] = fca.item()
            loss = loss + dr3_coeff * fca
        rd["subset"] = subset
        rd["td_target"] = td
        dicts.append(rd)
    loss = loss / (agent.E * agent.N)
    if encoder_lambda:
        oo, ao = rd["original_obs"][0], rd["augmented_obs"][0]
        with torch.no_grad():
            os_rep = encode(agent.encoder, oo)
        inv = torch.norm(encode(agent.encoder, ao) - os_rep)
        logs["encoder_constraint_loss"] = inv.item()
        loss = loss + encoder_lambda * inv
    encoder_opt.zero_grad()
    critic_opt.zero_grad()
    loss.backward()
    if critic_clip:
        clip_grad_norm(agent.critic_params(), critic_clip)
    if encoder_clip:
        clip_grad_norm(agent.encoder_params(), encoder_clip)
    encoder_opt.step()
    critic_opt.step()
    logs["losses/last_member_critic_td_error"] = td_error.mean().item()
    logs["losses/critic_overall_loss"] = loss.item()
    # learning.py:135-137: gradient norm (after clipping) of a random.choice'd member's critics, and of the encoder
    pick = py_rng.choice(range(agent.E)) if grad_pick is None else grad_pick
    logs["gradients/critic_random_grad"] = grad_norm([c[k] for c in agent.critics[pick] for k in MLP_KEYS])
    logs["gradients/encoder_criticloss_grad_norm"] = grad_norm(agent.encoder_params())
    return logs, dicts


# --------------------------------------------------------------------------------------
# a14: online actor update.  learning.py:344-421
# --------------------------------------------------------------------------------------
def online_actor_update(agent, actor_opt, log_alphas, dicts, pop, clip, eps_list=None,
                        noise_scale=None, noise_clip=None, noise_list=None, use_baseline=False,
                        base_eps_lists=None, grad_pick=None, py_rng=_pyrandom):
    """use_baseline (learning.py:401): vals = A(s, a) = Q(s, a) - V(s) from the advantage estimator (4 fresh policy
    samples, adv_estimator.py:58-79; its ``pop`` applies the PopArt layer whenever the member has one)."""
    logs = {}
    total = 0.0
    for i in range(agent.E):
        o = dicts[i]["primary_batch"][0]
        with torch.no_grad():
            s = encode(agent.encoder, o)
        out = mlp3(agent.actors[i], s)[0]
        popart = agent.popart[i]
        if agent.discrete:
            probs = torch.softmax(out, dim=-1)
            logp = torch.log_softmax(out, dim=-1)
            with torch.no_grad():
                vals = ensemble_q(agent.critics[i], s, None)
                if popart and pop:
                    vals = popart(vals)
            vals = (probs * vals).sum(-1, keepdim=True)
            bonus = log_alphas[i].exp() * (probs * logp).sum(-1, keepdim=True)
        else:
            e = None if eps_list is None else eps_list[i]
            if e is None:
                e = torch.randn(out.shape[0], agent.act_dim)
            if agent.actor_kind == "deterministic":
                a = torch.tanh(out) + 1e-4 * e  # Normal(loc,1e-4).rsample() (distributions.py:107-111)
                logp = None
            else:
                a, logp = tanh_normal_sample(out, agent.lo, agent.hi, e)
            if noise_scale is not None:
                nz = None if noise_list is None else noise_list[i]
                if nz is None:
                    nz = torch.randn(*a.shape)
                a = gaussian_exploration_noise(a, noise_scale, noise_clip, nz)
                bonus = torch.zeros(1)
            else:
                if logp is None:
                    logp = (-(e ** 2) / 2.0 - math.log(1e-4) - LOG_2PI_HALF).sum(-1, keepdim=True)
                bonus = log_alphas[i].exp() * logp
            if use_baseline:
                vals = advantage(agent, o, a, i, None if base_eps_lists is None else base_eps_lists[i],
                                 grad=True)
            else:
                vals = ensemble_q(agent.critics[i], s, a)  # min over ALL critics (learning.py:402)
                if popart and pop:
                    vals = popart(vals)
        total = total + (vals - bonus).mean()
    loss = -total / agent.E
    actor_opt.zero_grad()
    loss.backward()
    if clip:
        clip_grad_norm(agent.actor_params(), clip)
    actor_opt.step()
    pick = py_rng.choice(range(agent.E)) if grad_pick is None else grad_pick  # learning.py:417-419
    logs["gradients/random_actor_online_grad"] = grad_norm([agent.actors[pick][k] for k in MLP_KEYS])
    logs["losses/actor_pg_loss"] = loss.item()
    return logs


# --------------------------------------------------------------------------------------
# a15: temperature update.  learning.py:222-263 (loss uses log_alpha, not alpha)
# --------------------------------------------------------------------------------------
def alpha_update(agent, alpha_opts, log_alphas, dicts, target_entropy, eps_list=None):
    logs = {}
    for i in range(agent.E):
        o = dicts[i]["primary_batch"][0]
        with torch.no_grad():
            s = encode(agent.encoder, o)
            out = mlp3(agent.actors[i], s)[0]
            if agent.discrete:
                logp = (torch.softmax(out, -1) * torch.log_softmax(out, -1)).sum(-1)
            elif agent.actor_kind == "deterministic":
                logp = torch.full((out.shape[0], 1),
                                  agent.act_dim * (-math.log(1e-4) - LOG_2PI_HALF))
            else:
                e = None if eps_list is None else eps_list[i]
                if e is None:
                    e = torch.randn(out.shape[0], agent.act_dim)
                _, logp = tanh_normal_sample(out, agent.lo, agent.hi, e)
        loss = -(log_alphas[i] * (logp + target_entropy).detach()).mean()
        alpha_opts[i].zero_grad()
        loss.backward()
        alpha_opts[i].step()
        logs[f"losses/alpha_loss_{i}"] = loss.item()
        logs[f"alphas/alpha_{i}"] = log_alphas[i].exp().item()
    return logs


# --------------------------------------------------------------------------------------
# a16: polyak.  learning_utils.py:160-167
# --------------------------------------------------------------------------------------
@torch.no_grad()
def soft_update(target_params, source_params, tau):
    for t, s in zip(target_params, source_params):
        t.copy_(t * (1.0 - tau) + s * tau)


@torch.no_grad()
def hard_update(target_params, source_params):
    for t, s in zip(target_params, source_params):
        t.copy_(s)


# --------------------------------------------------------------------------------------
# advantage-filtered behavioural cloning (AFBC / AWAC actor update) with PER.
# adv_estimator.py:58-79 (continuous, "mean"/"max"), :41-56 (discrete indirect);
# learning_utils.py:241-269 (filtered_bc_loss), :287-295 (adjust_priorities); learning.py:144-219.
# --------------------------------------------------------------------------------------
def tanh_normal_log_prob_data(out, lo, hi, a):
    """log pi(a) of a DATA action: the TanhTransform cache misses, so the pre-tanh value is recovered as
    atanh(clamp(a, -0.99, 0.99)) (distributions.py:74-84) and the jacobian term uses that value."""
    mu, log_std = tanh_normal_params(out, lo, hi)
    std = log_std.exp()
    y = a.clamp(-0.99, 0.99)
    x = 0.5 * (torch.log1p(y) - torch.log1p(-y))
    base = -((x - mu) ** 2) / (2.0 * std * std) - log_std - LOG_2PI_HALF
    ladj = 2.0 * (LOG2 - x - F.softplus(-2.0 * x))
    return (base - ladj).sum(-1, keepdim=True)


def det_normal_log_prob(out, a):
    """ContinuousDeterministic(tanh(out)).log_prob(a) summed over the action dimensions (distributions.py:107-114: a
    Normal with scale 1e-4; torch.distributions.Normal.log_prob in fp32: var = scale ** 2, log_scale = scale.log())."""
    loc = torch.tanh(out)
    scale = torch.tensor(1e-4, dtype=torch.float32)
    var = scale ** 2
    return (-((a - loc) ** 2) / (2 * var) - scale.log() - math.log(math.sqrt(2 * math.pi))).sum(-1, keepdim=True)


def _pop_q(agent, i, s, a):
    """AdvantageEstimator.pop (adv_estimator.py:31-36): min over ALL critics of member i, then popart(q)
    (normalized=True default -> w*q + b) when the member has a PopArt layer."""
    q = ensemble_q(agent.critics[i], s, a)
    return agent.popart[i](q) if agent.popart[i] else q


def advantage(agent, o, a, i, eps_list=None, method="mean", n=4, grad=False):
    """A(s,a) = Q(s,a) - V(s).  Continuous: V from n sampled policy actions (eps_list: the n (B,A) normal
    draws, in order).  Discrete: V = sum_a mean_members(pi)(a) * Q(s)_a, Q(s,a) by gather."""
    with torch.no_grad():
        s = encode(agent.encoder, o)
        if agent.discrete:
            probs = torch.stack([torch.softmax(mlp3(ac, s)[0], dim=-1) for ac in agent.actors], 0).mean(0)
            q_all = _pop_q(agent, i, s, None)
            value = (probs * q_all).sum(-1, keepdim=True)
            return q_all.gather(-1, a.long()) - value
        out = mlp3(agent.actors[i], s)[0]
        qs = []
        for k in range(n):
            if agent.actor_kind == "deterministic":   # .sample() is the loc (distributions.py:113-114): no draw
                act = torch.tanh(out)
            else:
                e = eps_list[k] if eps_list is not None else torch.randn(out.shape[0], agent.act_dim)
                act = tanh_normal_sample(out, agent.lo, agent.hi, e)[0]
            qs.append(_pop_q(agent, i, s, act))
        qs = torch.stack(qs, 0)
        value = qs.mean(0) if method == "mean" else qs.max(0).values
    # Q(s, a) is evaluated OUTSIDE no_grad (adv_estimator.py:76): the use_baseline actor update differentiates it
    if grad:
        return _pop_q(agent, i, s, a) - value
    with torch.no_grad():
        return _pop_q(agent, i, s, a) - value


def offline_actor_update(buffer, per_tree, agent, actor_opt, batch_size, actor_clip, augmenter, aug_mix,
                         per=True, filter_=True, dicts=None, idx_list=None, eps_lists=None,
                         prio_member=None, prio_eps=None, method="mean", update_encoder=False, encoder_opt=None,
                         encoder_clip=None, actor_lambda=0.0, inv_eps_list=None, inv_cat_list=None, grad_pick=0):
    """learning.py:144-219 for identity encoders (update_encoder has nothing to update), actor_lambda 0.
    per_tree: PerOracle over the buffer's rows (None when per is False).
    eps_lists[i]: the 4 normal draws of member i's advantage estimate; prio_member / prio_eps: the
    ``random.choice(range(E))`` result and the 4 draws of adjust_priorities (drawn here when None)."""
    logs = {}
    total = 0.0
    rd = None
    weights = []
    for i in range(agent.E):
        if dicts is not None:
            rd = dicts[i]
        else:
            if per:
                if idx_list is not None:
                    idx = idx_list[i]
                    w = None
                else:
                    buffer.total_sample_calls += 1
                    idx, w = per_tree.sample(len(buffer), batch_size)
                rd = sample_move_and_augment(buffer, batch_size, augmenter, aug_mix, idx=idx)
                rd["imp_weights"] = None if w is None else torch.from_numpy(w)
                weights.append(w)
            else:
                rd = sample_move_and_augment(buffer, batch_size, augmenter, aug_mix,
                                             idx=None if idx_list is None else idx_list[i])
        o, a = rd["primary_batch"][0], rd["primary_batch"][1]
        mask = None
        if filter_:
            adv = advantage(agent, o, a, i, None if eps_lists is None else eps_lists[i], method)
            mask = (adv >= 0.0).float()
        if update_encoder:   # learning_utils.py:255-256: the encoder is trained through the BC loss (BC warm-up)
            s = encode(agent.encoder, o)
        else:
            with torch.no_grad():
                s = encode(agent.encoder, o)
        out = mlp3(agent.actors[i], s)[0]
        if agent.discrete:
            logp = torch.log_softmax(out, dim=-1).gather(-1, a.long())
        elif agent.actor_kind == "deterministic":
            logp = det_normal_log_prob(out, a)
        else:
            logp = tanh_normal_log_prob_data(out, agent.lo, agent.hi, a)
        if filter_:
            logs["losses/adv_weights_mean"] = mask.mean().item()
            logp = logp * mask
        loss_i = -logp.mean()
        logs[f"losses/filterd_bc_loss_{i}"] = loss_i.item()
        total = total + loss_i
        if actor_lambda:
            # action invariance constraint (learning_utils.py:272-285): an action sampled from the actor at the ORIGINAL
            # observation (no gradient) must be as likely at the AUGMENTED one; the second log-probability goes through
            # a fresh distribution object, i.e. through atanh(clamp(a)) for the tanh-normal (distributions.py:74-84).
            # inv_eps_list[i]: the normal draw of o_dist.sample(); inv_cat_list[i]: the sampled classes (discrete)
            oo, ao = rd["original_obs"][0], rd["augmented_obs"][0]
            with torch.no_grad():
                out_o = mlp3(agent.actors[i], encode(agent.encoder, oo))[0]
                if agent.discrete:
                    a_s = (inv_cat_list[i] if inv_cat_list is not None
                           else torch.distributions.Categorical(logits=out_o).sample())
                    olp = torch.log_softmax(out_o, -1).gather(-1, a_s.long().unsqueeze(-1)).squeeze(-1)
                    olp = olp.sum(-1, keepdim=True)   # (the reference sums the (B,) log-probabilities: one number)
                elif agent.actor_kind == "deterministic":
                    a_s = torch.tanh(out_o)
                    olp = det_normal_log_prob(out_o, a_s)
                else:
                    eps = inv_eps_list[i] if inv_eps_list is not None else torch.randn(batch_size, agent.act_dim)
                    a_s, olp = tanh_normal_sample(out_o, agent.lo, agent.hi, eps)
            out_a = mlp3(agent.actors[i], encode(agent.encoder, ao))[0]   # WITH gradient, encoder included
            if agent.discrete:
                alp = torch.log_softmax(out_a, -1).gather(-1, a_s.long().unsqueeze(-1)).squeeze(-1).sum(-1, keepdim=True)
            elif agent.actor_kind == "deterministic":
                alp = det_normal_log_prob(out_a, a_s)
            else:
                alp = tanh_normal_log_prob_data(out_a, agent.lo, agent.hi, a_s)
            total = total + actor_lambda * F.mse_loss(olp, alp)
    loss = total / agent.E
    actor_opt.zero_grad()
    if encoder_opt is not None:
        encoder_opt.zero_grad()
    loss.backward()
    if actor_clip:
        clip_grad_norm(agent.actor_params(), actor_clip)
    if encoder_clip:
        clip_grad_norm(agent.encoder_params(), encoder_clip)
    actor_opt.step()
    if update_encoder:
        encoder_opt.step()
    logs["losses/filtered_bc_overall_loss"] = loss.item()
    # learning.py:209-214 (all AFBC fixtures have one member: the random.choice pick is member 0)
    logs["gradients/actor_offline_grad_norm"] = grad_norm([agent.actors[grad_pick][k] for k in MLP_KEYS])
    logs["gradients/encoder_offline_actorloss_grad_norm"] = grad_norm(agent.encoder_params())
    new_prio = None
    if per:
        import random as _random
        o, a = rd["primary_batch"][0], rd["primary_batch"][1]
        m = prio_member if prio_member is not None else _random.choice(range(agent.E))
        adv = advantage(agent, o, a, m, prio_eps, method)
        new_prio = (F.relu(adv) + 1e-4).squeeze(1).numpy()
        per_tree.update(rd["priority_idxs"], new_prio)
    return logs, rd, new_prio, weights


# --------------------------------------------------------------------------------------
# Markov state-abstraction update.  learning.py:266-341; nets/mlps.py:44-75 (continuous inverse
# model), :96-110 (contrastive model), :152-167 (discrete inverse model); main.py:218-224 (its optimizer).
# --------------------------------------------------------------------------------------
def markov_state_abstraction_update(buffer, agent, inverse, contrastive, opt, batch_size, augmenter, aug_mix,
                                    inverse_coeff, contrastive_coeff, smoothness_coeff, smoothness_max_dist,
                                    grad_clip, idx=None, perm=None, inv_lo=-10.0, inv_hi=2.0):
    """inverse / contrastive: MLP parameter dicts (make_mlp) of the two side models; opt: AdamOracle over
    encoder_params + inverse + contrastive (the reference's chain order).  idx / perm: the replay indices and the
    torch.randperm(batch_size) draw of the reference (drawn here when None)."""
    rd = sample_move_and_augment(buffer, batch_size, augmenter, aug_mix, idx=idx)
    o, a, _, o1, _ = rd["primary_batch"]
    s_rep = encode(agent.encoder, o)          # both WITH gradient (learning.py:294-295)
    s1_rep = encode(agent.encoder, o1)
    out = mlp3(inverse, torch.cat((s_rep, s1_rep), dim=-1))[0]
    if agent.discrete:
        # Categorical(logits).log_prob(a.squeeze(-1)): (B,)
        logp = torch.log_softmax(out, dim=-1).gather(-1, a.long()).squeeze(-1)
        inverse_loss = -logp.mean()
    else:
        # SquashedNormal.log_prob of a DATA action is per dimension (B, A): the mean runs over B x A values
        mu, log_std = tanh_normal_params(out, inv_lo, inv_hi)
        std = log_std.exp()
        y = a.clamp(-0.99, 0.99)
        x = 0.5 * (torch.log1p(y) - torch.log1p(-y))
        base = -((x - mu) ** 2) / (2.0 * std * std) - log_std - LOG_2PI_HALF
        ladj = 2.0 * (LOG2 - x - F.softplus(-2.0 * x))
        inverse_loss = -(base - ladj).mean()
    if perm is None:
        perm = torch.randperm(batch_size)
    s1_neg = s1_rep[perm]
    labels = torch.cat((torch.ones(batch_size, 1), torch.zeros(batch_size, 1)), dim=0)
    s_c = torch.cat((s_rep, s_rep), dim=0)
    s1_c = torch.cat((s1_rep, s1_neg), dim=0)
    pred = torch.sigmoid(mlp3(contrastive, torch.cat((s_c, s1_c), dim=-1))[0])
    contrastive_loss = F.binary_cross_entropy(pred, labels)
    dist = torch.norm(s1_rep - s_rep, dim=-1, p=2) / math.sqrt(s_rep.shape[-1])
    smoothness_loss = F.relu(dist - smoothness_max_dist).square().mean()
    loss = inverse_coeff * inverse_loss + contrastive_coeff * contrastive_loss + smoothness_coeff * smoothness_loss
    opt.zero_grad()
    loss.backward()
    inv_p, con_p = [inverse[k] for k in MLP_KEYS], [contrastive[k] for k in MLP_KEYS]
    if grad_clip is not None:
        clip_grad_norm(agent.encoder_params() + inv_p + con_p, grad_clip)
    opt.step()
    return {"gradients/contrastive_model_grad_norm": grad_norm(con_p),
            "gradients/inverse_model_grad_norm": grad_norm(inv_p),
            "gradients/encoder_markovloss_grad_norm": grad_norm(agent.encoder_params()),
            "losses/markov_loss": loss.item(), "losses/inverse_model_loss": inverse_loss.item(),
            "losses/contrastive_model_loss": contrastive_loss.item(),
            "losses/smoothness_loss": smoothness_loss.item()}, rd
