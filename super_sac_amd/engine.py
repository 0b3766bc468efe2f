"""Host-side plumbing over the C ABI: packed parameter arenas, Adam state, workspace and
the forward/backward launch sequences of a 3-layer ensemble MLP.

PyTorch is used here for device memory (tensors own the HBM allocations), streams and
pinned staging only; all arithmetic on the update path happens in libssac_hip.so.
"""
import ctypes as C
import os
import struct

import torch

from . import _lib
from ._lib import check, lib

SEGS = ("w1", "b1", "w2", "b2", "w3", "b3")
# bench.py's live kernel timing: when PROFILE["tag"] names a forward, its hidden-layer (fc2) launch
# is bracketed by events recorded on the launch stream.
# "reps": the bracketed launch is issued that many times back to back (it is idempotent), so the event pair's own
# overhead (~5 us) is amortised and the per-launch time agrees with a rocprofv3 kernel trace.
PROFILE = {"tag": None, "events": [], "reps": 1}
USE_FUSED = True  # tests flip this to exercise the per-layer kernels on fused-capable shapes


def set_wgrad_variant(variant):
    """form of the merged weight-gradient launch: 0 = automatic (the 32 x 32-tile latency form while all of its workgroups
    are resident at once, else 64 x 64 tiles), 1 = 64 x 64 tiles always, 2 = 32 x 32 tiles whenever the shapes allow.  Both
    forms are parity-tested on every fixture (tests/test_hip_cases.py); recorded launch lists keep the form they were
    recorded with."""
    check(lib.ssac_wgrad_variant(int(variant)))


# launch the head layer's (VALU) weight-gradient kernel as a parallel branch beside the fc2/fc1 GEMM launch
HEAD_BRANCH = False
# head + fc2 + fc1 weight gradients in one launch (the head's VALU workgroups fill CUs the GEMM tiles leave idle)
MERGE_HEAD_WGRAD = True


class CaptureCtx:
    """static device inputs of one captured update (learning.py, graph replay): while CAPTURE is set,
    the host-RNG draw sites hand out these buffers instead of drawing, so the launch sequence that gets
    recorded into a HIP graph reads its per-update inputs from fixed addresses."""

    def __init__(self, idx_cpu, idx_dev, ids, ids_dev, normals, logblk, feed=0):
        self.idx_cpu, self.idx_dev, self.ids, self.ids_dev = idx_cpu, idx_dev, ids, ids_dev
        self.normals = list(normals)
        self.logblk = logblk
        self.feed = feed          # device address of the update's ssac_feed (0: inputs arrive by copy)
        self.published = False    # set once a captured launch has published the log block
        self.tick_ptr = 0         # device address of the update counter (ssac_feed.tick), for in-kernel noise
        self.noise_offset = 0     # draw number of this agent's noise stream at capture time
        self.collective = None    # callable(fn): ends the open recording, runs fn() now, opens the next segment
        self.deferred = None      # ssac_deferred_logs of this recorded update (deferred log finalisation), or None
        self.deferred_chain = False  # the chained launch was issued with the finishing workgroup
        self.deferred_used = False   # ... and the weight-gradient launch left the partials for it
        self.late = None          # ssac_late_polyak of this recorded update (late-bound Polyak), or None
        self.late_target = None   # ... and the target arena its weight-gradient launch carries
        self.late_used = False
        self.defer_begin = False  # the replay gather will also do ssac_begin_update's work (vector buffers)
        self.pending_begin = None # (log block, adam ctl ptr) waiting for that gather


CAPTURE = None


def _ptr(t):
    return 0 if t is None else t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def stream():
    """raw handle of torch's current stream on the current device (asked on every launch site: the fast C accessors
    cost ~0.3 us, torch.cuda.current_stream() ~8 us)"""
    if _raw_stream is not None and _raw_device is not None:
        return _raw_stream(_raw_device())
    return torch.cuda.current_stream().cuda_stream


_side = {}


class side_stream:
    """run the enclosed launches on a second HIP stream, forked from the current one (inside a graph capture
    this becomes a parallel branch): used to overlap independent kernels with the launches of the main
    chain instead of queueing them behind.  The branch is joined back when the block exits, or -- with
    ``defer_join`` -- when the caller invokes ``join()`` later on the main stream."""

    def __init__(self, device, defer_join=False):
        s = _side.get(device)
        if s is None:
            s = _side[device] = torch.cuda.Stream(device=device)
        self.s = s
        self.defer = defer_join

    def __enter__(self):
        self.main = torch.cuda.current_stream()
        self.s.wait_stream(self.main)
        self.ctx = torch.cuda.stream(self.s)
        self.ctx.__enter__()
        return self

    def __exit__(self, *a):
        self.ctx.__exit__(*a)
        if not self.defer:
            self.main.wait_stream(self.s)

    def join(self):
        torch.cuda.current_stream().wait_stream(self.s)


def require_gpu(t=None):
    if not torch.cuda.is_available():
        raise RuntimeError("super_sac_amd runs its update path only on an MI355X/ROCm device; "
                           "no CPU fallback exists (the CPU oracle under oracle/ is test-only)")
    if t is not None and not t.is_cuda:
        raise RuntimeError("expected a device tensor on the update path")


def mlp_layout(in_dim, hidden, out_dim):
    offs = (C.c_int64 * 6)()
    stride = lib.ssac_mlp_layout(in_dim, hidden, out_dim, offs)
    return int(stride), [int(o) for o in offs]


class Workspace:
    """Named scratch tensors, allocated once per (name, shape) so the hot loop never allocates."""

    def __init__(self, device):
        self.device = device
        self._bufs = {}

    def get(self, name, shape, dtype=torch.float32, zero=False):
        key = (name, tuple(shape), dtype)
        buf = self._bufs.get(key)
        if buf is None:
            buf = (torch.zeros if zero else torch.empty)(tuple(shape), dtype=dtype, device=self.device)
            self._bufs[key] = buf
        return buf


class MlpArena:
    """`n_nets` identically shaped 3-layer MLPs packed as one HBM arena (include/ssac_hip.h).

    ``adopt`` re-points the ``.data`` of existing ``nn.Linear`` parameters at views of the
    arena, so the torch modules (state_dict / save / load / deepcopy) and the kernels share
    one copy of the weights."""

    def __init__(self, n_nets, in_dim, hidden, out_dim, device):
        self.n_nets, self.in_dim, self.hidden, self.out_dim = n_nets, in_dim, hidden, out_dim
        self.stride, self.offs = mlp_layout(in_dim, hidden, out_dim)
        self.shapes = [(hidden, in_dim), (hidden,), (hidden, hidden), (hidden,), (out_dim, hidden),
                       (out_dim,)]
        self.params = torch.zeros(n_nets * self.stride, dtype=torch.float32, device=device)
        self.device = device
        # one-launch fused kernels (csrc/ssac_fused.hip) apply to this shape?
        kind = int(lib.ssac_fused_supported(C.byref(self.desc())))  # 1 double-buffered staging fits, 2 single only
        self.fused = bool(kind) and USE_FUSED
        self.fused_dbuf = kind == 1 and USE_FUSED  # eligible as the critic half of the merged launches
        # bf16-operand mode (csrc/ssac_bf16.hip): a bf16 shadow of the arena, None while the arena computes in fp32
        self.shadow = None

    # ---- bf16 shadow ---------------------------------------------------------------------------
    def enable_bf16(self):
        """allocate the bf16 shadow [W1 | W2 | W2^T | W3] per net and fill it from the fp32 masters"""
        if not int(lib.ssac_bf16_supported(C.byref(self.desc()))):
            raise NotImplementedError(f"bf16 mode: shape {self.in_dim}->{self.hidden}->{self.out_dim} is not covered "
                                      "(hidden % 32 == 0, hidden <= 256, out_dim <= 64)")
        if self.shadow is None:
            stride = int(lib.ssac_bf16_layout(self.in_dim, self.hidden, self.out_dim, None))
            self.shadow_stride = stride
            self.shadow = torch.zeros(self.n_nets * stride, dtype=torch.bfloat16, device=self.device)
        self.sync_shadow()
        return self

    def sync_shadow(self):
        """shadow <- bf16(master): after construction, load_state_dict, or any fp32 update of the arena"""
        if self.shadow is not None:
            check(lib.ssac_bf16_sync(C.byref(self.desc()), self.shadow.data_ptr(), stream()))

    def bf_buffers(self, ws, tag, n_rows):
        """transposed bf16 saves of the chained update: H1T, H2T, DZ2uT, DZ1uT (n_nets x hidden x Bp), XT (K1P x Bp);
        zero-initialised once (the kernels never write the pad columns b >= n_rows)"""
        bp = (n_rows + 15) // 16 * 16
        k1p = (self.in_dim + 31) // 32 * 32   # (fragment-major saves: whole 32-row blocks, csrc/ssac_bf16.hip)
        shp = (self.n_nets, self.hidden, bp)
        return {k_: ws.get(f"{tag}.bf.{k_}", shp if k_ != "xt" else (k1p, bp), dtype=torch.bfloat16, zero=True)
                for k_ in ("h1t", "h2t", "dz2t", "dz1t", "xt")}

    def like(self):
        return torch.zeros_like(self.params)

    def desc(self, tensor=None):
        t = self.params if tensor is None else tensor
        return _lib.MlpDesc(t.data_ptr(), self.stride, self.n_nets, self.in_dim, self.hidden,
                            self.out_dim)

    def view(self, net, seg, tensor=None):
        t = self.params if tensor is None else tensor
        k = SEGS.index(seg)
        o = net * self.stride + self.offs[k]
        shp = self.shapes[k]
        n = 1
        for d in shp:
            n *= d
        return t[o:o + n].view(shp)

    def tiles(self, layer):
        return int(lib.ssac_wgrad_tiles(C.byref(self.desc()), layer))

    # ---- adoption of torch modules -------------------------------------------------------
    @staticmethod
    def linear_triples(module):
        """the three Linear layers of an actor/critic module, in order."""
        last = [n for n in ("fc3", "out", "act_p") if hasattr(module, n)]
        assert hasattr(module, "fc1") and hasattr(module, "fc2") and len(last) == 1, \
            f"{type(module).__name__} is not a 3-layer MLP the engine recognises"
        return [module.fc1, module.fc2, getattr(module, last[0])]

    def is_bound(self, modules):
        """cheap per-call validation: every parameter still points at its arena slot."""
        ptrs = self.__dict__.get("_bound_ptrs")
        if ptrs is None or len(ptrs[0]) != 6 * len(modules):
            params, expect = [], []
            for j, m in enumerate(modules):
                for k, lin in enumerate(self.linear_triples(m)):
                    for seg, p in ((SEGS[2 * k], lin.weight), (SEGS[2 * k + 1], lin.bias)):
                        params.append(p)
                        expect.append(self.view(j, seg).data_ptr())
            # (bound methods of the Parameter objects: they follow a re-assigned .data; one list comparison per call)
            ptrs = self.__dict__["_bound_ptrs"] = (params, expect, [p.data_ptr for p in params])
        return [f() for f in ptrs[2]] == ptrs[1]

    @classmethod
    def adopt(cls, modules, device):
        l0 = cls.linear_triples(modules[0])
        in_dim, hidden, out_dim = l0[0].in_features, l0[0].out_features, l0[2].out_features
        arena = cls(len(modules), in_dim, hidden, out_dim, device)
        with torch.no_grad():
            for j, m in enumerate(modules):
                lins = cls.linear_triples(m)
                assert (lins[0].in_features, lins[1].in_features, lins[2].out_features) == \
                    (in_dim, hidden, out_dim), "ensemble members must share one shape"
                for k, lin in enumerate(lins):
                    for seg, p in ((SEGS[2 * k], lin.weight), (SEGS[2 * k + 1], lin.bias)):
                        v = arena.view(j, seg)
                        v.copy_(p.data.to(device=device, dtype=torch.float32))
                        p.data = v
        return arena


def bind_arena(owner, key, modules, device):
    """Arena for `modules`, cached on `owner` and re-validated by pointer each call (a
    deepcopy or load_state_dict that re-allocates parameters triggers a re-pack)."""
    cache = owner.__dict__.setdefault("_ssac_arenas", {})
    arena = cache.get(key)
    if arena is None or arena.n_nets != len(modules) or not arena.is_bound(modules):
        arena = MlpArena.adopt(modules, device)
        cache[key] = arena
        if owner.__dict__.get("_ssac_precision") == "bf16":
            arena.enable_bf16()
    return arena


def set_precision(agent, precision):
    """select the operand type of the matrix products of an agent's networks: "fp32" (the reference's arithmetic, the
    default) or "bf16" (BASELINE.json config 2; fp32 masters + bf16 shadows, csrc/ssac_bf16.hip).  Marks the modules, so
    a later ``copy.deepcopy(agent)`` (the target network) inherits the choice."""
    assert precision in ("fp32", "bf16")
    dev = next(agent.actors[0].parameters()).device
    for owner, key, mods in ([(a, "self", [a]) for a in agent.actors] +
                             [(c, "nets", list(c.nets)) for c in agent.critics]):
        owner.__dict__["_ssac_precision"] = precision
        arena = owner.__dict__.get("_ssac_arenas", {}).get(key)
        if arena is not None:
            if precision == "bf16":
                arena.enable_bf16()
            else:
                arena.shadow = None
    return agent


def sync_shadows(agent):
    """refresh every bf16 shadow of an agent from its fp32 masters (after load / load_state_dict)"""
    for owner in list(agent.actors) + list(agent.critics):
        for arena in owner.__dict__.get("_ssac_arenas", {}).values():
            arena.sync_shadow()


class DeviceStruct:
    """A small C struct living in device memory (Adam control block, PopArt state)."""

    def __init__(self, cstruct, device):
        self.ctype = type(cstruct)
        self.size = C.sizeof(cstruct)
        self.dev = torch.zeros(self.size, dtype=torch.uint8, device=device)
        self.write(cstruct)

    def write(self, cstruct):
        host = torch.frombuffer(bytearray(bytes(cstruct)), dtype=torch.uint8)
        self.dev.copy_(host)

    def read(self):
        raw = bytes(self.dev.cpu().numpy().tobytes())
        return self.ctype.from_buffer_copy(raw)

    @property
    def ptr(self):
        return self.dev.data_ptr()


class AdamGroup:
    """Adam state for one ``torch.optim.Adam`` instance (hyper-parameters are read from its
    param_groups, exactly the values main.py:188-239 passes).  Moments live in arenas of the
    same layout as the parameters they belong to."""

    def __init__(self, optimizer, device):
        g = optimizer.param_groups[0]
        self.lr, (self.b1, self.b2) = float(g["lr"]), g["betas"]
        self.eps, self.wd = float(g["eps"]), float(g["weight_decay"])
        assert not g.get("amsgrad", False), "amsgrad is not supported (the reference never uses it)"
        ctl = _lib.AdamCtl(self.lr, self.b1, self.b2, self.eps, self.wd, 0.0, 1.0, 1.0, 0,
                           (C.c_int32 * 3)(0, 0, 0), self.lr, self.b1, self.b2)
        self.ctl = DeviceStruct(ctl, device)
        self.moments = {}
        self.device = device

    def moments_for(self, key, like):
        mv = self.moments.get(key)
        if mv is None or mv[0].numel() != like.numel():
            mv = (torch.zeros_like(like), torch.zeros_like(like))
            self.moments[key] = mv
        return mv

    def advance(self):
        check(lib.ssac_adam_advance(self.ctl.ptr, stream()))


def adam_group(optimizer, device):
    grp = getattr(optimizer, "_ssac_adam", None)
    if grp is None:
        grp = AdamGroup(optimizer, device)
        optimizer._ssac_adam = grp
    return grp


# ------------------------------------------------------------------------------------------
# launch sequences
# ------------------------------------------------------------------------------------------
def _timed(tag, repeatable=True):
    """context manager recording a (start, end) event pair when bench.py asked for `tag`.  repeatable=False: the launch is
    not idempotent (it steps the optimizer): one issue per event pair whatever PROFILE["reps"] says."""
    class _T:
        def __enter__(self_):
            self_._enter()
            return self_

        def _enter(self_):
            want = PROFILE["tag"]
            self_.on = want is not None and (tag == want or (isinstance(want, tuple) and tag in want))
            self_.reps = max(1, int(PROFILE.get("reps", 1))) if (self_.on and repeatable) else 1
            if self_.on:
                self_.e0 = torch.cuda.Event(enable_timing=True)
                self_.e1 = torch.cuda.Event(enable_timing=True)
                self_.e0.record()
        def __exit__(self_, *a):
            if self_.on:
                self_.e1.record()
                PROFILE["events"].append((self_.e0, self_.e1, tag, self_.reps))
    return _T()


def mlp_forward(arena, X, ldx, x_net_stride, n_rows, ws, tag, net_ids=None, n_sel=None,
                params=None, save=True, force_layers=False):
    """h1 = relu(fc1 x), h2 = relu(fc2 h1), y = out(h2) for every selected net.
    Returns (h1, h2, y) with shapes (n_sel, n_rows, H|H|out); h1/h2 are what autograd would have
    saved for the backward pass (None when `save` is False and the fused kernel kept them in LDS)."""
    n_sel = arena.n_nets if n_sel is None else n_sel
    H, O = arena.hidden, arena.out_dim
    y = ws.get(tag + ".y", (n_sel, n_rows, O))
    d = arena.desc(params)
    ids = _ptr(net_ids)
    st = stream()
    if arena.fused and not force_layers:
        h1 = ws.get(tag + ".h1", (n_sel, n_rows, H)) if save else None
        h2 = ws.get(tag + ".h2", (n_sel, n_rows, H)) if save else None
        check(lib.ssac_mlp3_fwd_fused(C.byref(d), ids, n_sel, X.data_ptr(), ldx, x_net_stride, n_rows,
                                      _ptr(h1), _ptr(h2), y.data_ptr(), st))
        return h1, h2, y
    h1 = ws.get(tag + ".h1", (n_sel, n_rows, H))
    h2 = ws.get(tag + ".h2", (n_sel, n_rows, H))
    check(lib.ssac_mlp_layer_fwd(C.byref(d), 0, ids, n_sel, X.data_ptr(), ldx, x_net_stride, n_rows,
                                 h1.data_ptr(), H, n_rows * H, 1, st))
    with _timed(tag + ".fc2"):
        check(lib.ssac_mlp_layer_fwd(C.byref(d), 1, ids, n_sel, h1.data_ptr(), H, n_rows * H, n_rows,
                                     h2.data_ptr(), H, n_rows * H, 1, st))
    check(lib.ssac_mlp_layer_fwd(C.byref(d), 2, ids, n_sel, h2.data_ptr(), H, n_rows * H, n_rows,
                                 y.data_ptr(), O, n_rows * O, 0, st))
    return h1, h2, y


def weight_grads(arena, X, ldx, x_net_stride, h1, h2, dY, dz2, dz1, n_rows, *, adam=None, adam_key=None,
                 grads=None, sumsq=None, target=None, tau=0.0, net_ids=None, n_sel=None, logs=None,
                 rowscale=None, lossfold=None, actor_fold=None):
    """the weight-gradient launch(es) (+Adam/Polyak in their epilogues, or gradient store).
    logs (critic update, Adam mode, merged launch only): dict(partials, tiles, denom, logs, spec_ptr, td_logs_ptr,
    feed, done) -- the launch's last workgroup then also finalises the update's logs; returns True when it did."""
    n_sel = arena.n_nets if n_sel is None else n_sel
    H, O = arena.hidden, arena.out_dim
    d = arena.desc()
    ids = _ptr(net_ids)
    st = stream()
    m = v = None
    if grads is None or (lossfold is not None and arena.shadow is not None):
        m, v = adam.moments_for(adam_key, arena.params)
    tiles = [arena.tiles(l) for l in range(3)]
    ttot = sum(tiles)
    off = {0: 0, 1: tiles[0], 2: tiles[0] + tiles[1]}
    ctl = 0 if adam is None else adam.ctl.ptr

    def ssp(layer):
        return 0 if sumsq is None else sumsq.data_ptr() + 4 * off[layer]
    if lossfold is not None and arena.shadow is not None:
        # bf16 mode: transposed bf16 saves, Adam on the fp32 masters, shadow refreshed by the epilogue
        f = lossfold
        bf = f["bf"]
        tsh = f.get("target_shadow")
        check(lib.ssac_bf16_wgrad_lossfold(
            C.byref(d), arena.shadow.data_ptr(), bf["xt"].data_ptr(), bf["h1t"].data_ptr(), bf["h2t"].data_ptr(),
            bf["dz2t"].data_ptr(), bf["dz1t"].data_ptr(), f["q"].data_ptr(), f["td_ptr"], f["spec_ptr"], f["weight_ptr"],
            f.get("popart_ptr", 0), int(f.get("pop", 0)),
            float(f["denom"]), f["partials"].data_ptr(), n_rows, m.data_ptr(), v.data_ptr(), ctl, _ptr(grads), _ptr(sumsq),
            bf16_tiles_total(arena), _ptr(target), _ptr(tsh), float(tau),
            C.byref(f["logfold"]) if f.get("logfold") is not None else 0, st))
        return f.get("logfold") is not None
    if lossfold is not None:
        # UNSCALED backward, and the loss gradient dL/dq itself is evaluated inside the launch (per workgroup, in LDS)
        assert O == 1 and net_ids is None and n_sel == arena.n_nets
        f = lossfold
        with _timed("wgrad", repeatable=False):   # (bench.py's live timing; the launch steps Adam)
            check(lib.ssac_mlp_wgrad_all_lossfold(
                C.byref(d), X.data_ptr(), ldx, x_net_stride, h1.data_ptr(), h2.data_ptr(),
                0 if f.get("dz2_from_h2") else dz2.data_ptr(),   # (0: dz2u is rebuilt from h2 and W3 in the operand staging)
                dz1.data_ptr(), _ptr(f.get("w3_snapshot")),
                f["q"].data_ptr(), f["td_ptr"], f["spec_ptr"], f["weight_ptr"], f["popart_ptr"], f["pop"],
                float(f["denom"]), f["partials"].data_ptr(), n_rows, _ptr(m), _ptr(v), ctl, _ptr(grads), ssp(2), ssp(1),
                ssp(0), ttot, _ptr(target), float(tau), C.byref(f["logfold"]) if f.get("logfold") is not None else 0, st))
        return f.get("logfold") is not None
    if rowscale is not None:
        # UNSCALED backward (ssac_target_fwd_critic_bwdu): dL/dq of every (net, row) scales the rows while they load
        assert O == 1
        with _timed("wgrad", repeatable=False):
            check(lib.ssac_mlp_wgrad_all_scaled(C.byref(d), ids, n_sel, X.data_ptr(), ldx, x_net_stride, h1.data_ptr(),
                                                h2.data_ptr(), dz2.data_ptr(), dz1.data_ptr(), rowscale.data_ptr(), n_rows,
                                                _ptr(m), _ptr(v), ctl, _ptr(grads), ssp(2), ssp(1), ssp(0), ttot,
                                                _ptr(target), float(tau), st))
        return
    if actor_fold is not None and actor_fold_applies(arena, grads, sumsq, target, net_ids, n_sel):
        # the online actor update's ONE net: the same merged launch, its last workgroup also finishes the update's two logs
        check(lib.ssac_mlp_wgrad_all_actor(C.byref(d), X.data_ptr(), ldx, h1.data_ptr(), h2.data_ptr(), dz2.data_ptr(),
                                           dz1.data_ptr(), dY.data_ptr(), n_rows, m.data_ptr(), v.data_ptr(), ctl, ssp(2), ssp(1),
                                           ssp(0), ttot, C.byref(actor_fold), st))
        return True
    if O <= 16 and MERGE_HEAD_WGRAD:
        # head (VALU), fc2 and fc1 weight gradients of every selected net: ONE launch
        check(lib.ssac_mlp_wgrad_all(C.byref(d), ids, n_sel, X.data_ptr(), ldx, x_net_stride, h1.data_ptr(),
                                     h2.data_ptr(), dz2.data_ptr(), dz1.data_ptr(), dY.data_ptr(), n_rows, _ptr(m),
                                     _ptr(v), ctl, _ptr(grads), ssp(2), ssp(1), ssp(0), ttot, _ptr(target),
                                     float(tau), st))
        return
    if O <= 16:
        # the head's weight gradient is an independent VALU kernel: run it beside the GEMM launch below
        import contextlib
        with (side_stream(arena.params.device) if HEAD_BRANCH else contextlib.nullcontext()):
            check(lib.ssac_head_wgrad(C.byref(d), ids, n_sel, h2.data_ptr(), dY.data_ptr(), n_rows, _ptr(m),
                                      _ptr(v), ctl, _ptr(grads), ssp(2), ttot, _ptr(target), float(tau),
                                      stream()))
    else:
        check(lib.ssac_mlp_layer_wgrad(C.byref(d), 2, ids, n_sel, h2.data_ptr(), H, n_rows * H,
                                       dY.data_ptr(), O, n_rows * O, n_rows, _ptr(m), _ptr(v), ctl,
                                       _ptr(grads), ssp(2), ttot, _ptr(target), float(tau), st))
    # fc2 and fc1 weight gradients share one launch
    check(lib.ssac_mlp_wgrad_fc12(C.byref(d), ids, n_sel, X.data_ptr(), ldx, x_net_stride, h1.data_ptr(),
                                  dz2.data_ptr(), dz1.data_ptr(), n_rows, _ptr(m), _ptr(v), ctl, _ptr(grads),
                                  ssp(1), ssp(0), ttot, _ptr(target), float(tau), st))


def actor_fold_applies(arena, grads, sumsq, target, net_ids, n_sel):
    """can the merged weight-gradient launch of this call carry the actor update's logs (ssac_mlp_wgrad_all_actor)?"""
    return (MERGE_HEAD_WGRAD and arena.out_dim <= 16 and arena.n_nets == 1 and grads is None and sumsq is not None
            and target is None and net_ids is None and (n_sel is None or n_sel == 1) and arena.shadow is None)


def mlp_backward(arena, dY, X, ldx, x_net_stride, h1, h2, n_rows, ws, tag, *, adam=None,
                 adam_key=None, need_dx=False, grads=None, sumsq=None, target=None, tau=0.0,
                 net_ids=None, n_sel=None, update=True, after_dz2=None):
    """Backward of `mlp_forward` given dL/dy.  With `update`, the weight gradients are consumed
    in-kernel by Adam (or stored to `grads` when clipping needs the global norm first).
    Returns dX (n_sel, n_rows, in_dim) when `need_dx`."""
    n_sel = arena.n_nets if n_sel is None else n_sel
    H, O, I = arena.hidden, arena.out_dim, arena.in_dim
    d = arena.desc()
    ids = _ptr(net_ids)
    st = stream()
    dz2 = ws.get(tag + ".dz2", (n_sel, n_rows, H))
    dz1 = ws.get(tag + ".dz1", (n_sel, n_rows, H))
    check(lib.ssac_mlp_layer_dgrad(C.byref(d), 2, ids, n_sel, dY.data_ptr(), O, n_rows * O,
                                   h2.data_ptr(), H, n_rows * H, n_rows,
                                   dz2.data_ptr(), H, n_rows * H, st))
    if after_dz2 is not None:
        after_dz2(dz2)  # extra gradient into the fc2 pre-activations (DR3)
    check(lib.ssac_mlp_layer_dgrad(C.byref(d), 1, ids, n_sel, dz2.data_ptr(), H, n_rows * H,
                                   h1.data_ptr(), H, n_rows * H, n_rows,
                                   dz1.data_ptr(), H, n_rows * H, st))
    dX = None
    if need_dx:
        dX = ws.get(tag + ".dx", (n_sel, n_rows, I))
        check(lib.ssac_mlp_layer_dgrad(C.byref(d), 0, ids, n_sel, dz1.data_ptr(), H, n_rows * H,
                                       0, 0, 0, n_rows, dX.data_ptr(), I, n_rows * I, st))
    if update:
        weight_grads(arena, X, ldx, x_net_stride, h1, h2, dY, dz2, dz1, n_rows, adam=adam,
                     adam_key=adam_key, grads=grads, sumsq=sumsq, target=target, tau=tau,
                     net_ids=net_ids, n_sel=n_sel)
    return dX


def wgrad_tiles_total(arena):
    return sum(arena.tiles(l) for l in range(3))


def bf16_tiles_total(arena):
    """gradient-norm partial slots per net of the bf16 weight-gradient launch"""
    return int(lib.ssac_bf16_wgrad_tiles(C.byref(arena.desc())))


def pack_f32(*vals):
    return struct.pack("%df" % len(vals), *vals)
