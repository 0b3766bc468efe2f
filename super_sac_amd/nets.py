"""Parameter containers with the reference's class and attribute names (super_sac/nets/).

The modules own ``nn.Linear`` / ``nn.Conv2d`` parameters so ``state_dict`` / ``save`` /
``load`` / optimizers work exactly as with the reference, but they hold no arithmetic: the
update path reads their weights through packed arenas (engine.MlpArena) and runs in HIP.
``forward`` on an MLP module evaluates it with the same HIP layer kernel (no autograd).
"""
import ctypes as C
import math

import torch
from torch import nn

from . import engine
from ._lib import check, lib


def weight_init(m):
    """orthogonal Linear / delta-orthogonal Conv2d, zero bias (nets/__init__.py:4-15)."""
    if isinstance(m, nn.Linear):
        nn.init.orthogonal_(m.weight.data)
        m.bias.data.fill_(0.0)
    elif isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
        assert m.weight.size(2) == m.weight.size(3)
        m.weight.data.fill_(0.0)
        m.bias.data.fill_(0.0)
        mid = m.weight.size(2) // 2
        nn.init.orthogonal_(m.weight.data[:, :, mid, mid], nn.init.calculate_gain("relu"))


class _MLP3(nn.Module):
    """fc1 -> relu -> fc2 -> relu -> <head>; the head attribute name differs per class."""
    HEAD = "out"

    def _build(self, in_dim, hidden, out_dim):
        self.fc1 = nn.Linear(in_dim, hidden)
        self.fc2 = nn.Linear(hidden, hidden)
        setattr(self, self.HEAD, nn.Linear(hidden, out_dim))
        self.apply(weight_init)

    def raw_forward(self, x):
        """head output (n_rows, out) computed by the HIP layer kernel."""
        engine.require_gpu(x)
        x = x.contiguous().float()
        arena = engine.bind_arena(self, "self", [self], x.device)
        ws = self.__dict__.setdefault("_ssac_ws", engine.Workspace(x.device))
        _, h2, y = engine.mlp_forward(arena, x, x.shape[1], 0, x.shape[0], ws, "fwd")
        self.features = h2[0]
        return y[0].clone()


class ContinuousCritic(_MLP3):
    def __init__(self, state_size, action_size, hidden_size=256):
        super().__init__()
        self.features = None
        self._build(state_size + action_size, hidden_size, 1)

    def forward(self, state, action):
        return self.raw_forward(torch.cat((state, action), dim=-1))


class DiscreteCritic(_MLP3):
    def __init__(self, state_size, action_size, hidden_size=300):
        super().__init__()
        self.features = None
        self._build(state_size, hidden_size, action_size)

    def forward(self, state):
        return self.raw_forward(state)


class ContinuousStochasticActor(_MLP3):
    HEAD = "fc3"

    def __init__(self, state_size, action_size, log_std_low=-10.0, log_std_high=2.0,
                 hidden_size=256, dist_impl="pyd"):
        super().__init__()
        assert dist_impl == "pyd", "only the tanh-normal head is on the accelerated path"
        self.log_std_low, self.log_std_high, self.dist_impl = log_std_low, log_std_high, dist_impl
        self.action_size = action_size
        self._build(state_size, hidden_size, 2 * action_size)


class ContinuousDeterministicActor(_MLP3):
    HEAD = "out"

    def __init__(self, state_size, action_size, hidden_size=256, **kwargs):
        super().__init__()
        self.dist_impl = "deterministic"
        self.action_size = action_size
        self._build(state_size, hidden_size, action_size)


class DiscreteActor(_MLP3):
    HEAD = "act_p"

    def __init__(self, state_size, action_size, hidden_size=256):
        super().__init__()
        self.action_size = action_size
        self._build(state_size, hidden_size, action_size)


class ContinuousInverseModel(_MLP3):
    """mlps.py:45-76: (s, s') -> tanh-normal over the action that was taken (Markov state abstraction)."""
    HEAD = "fc3"

    def __init__(self, state_size, action_size, log_std_low=-10.0, log_std_high=2.0, hidden_size=256,
                 dist_impl="pyd"):
        super().__init__()
        assert dist_impl == "pyd", "only the tanh-normal head is on the accelerated path"
        self.log_std_low, self.log_std_high, self.dist_impl = log_std_low, log_std_high, dist_impl
        self.action_size = action_size
        self._build(2 * state_size, hidden_size, 2 * action_size)


class DiscreteInverseModel(_MLP3):
    """mlps.py:153-168: (s, s') -> logits over the action that was taken."""
    HEAD = "act_p"

    def __init__(self, state_size, action_size, hidden_size, **kwargs):
        super().__init__()
        self.action_size = action_size
        self._build(2 * state_size, hidden_size, action_size)


class ContrastiveModel(_MLP3):
    """mlps.py:97-110: (s, s') -> logit of "this is a real transition" (the sigmoid lives in the loss kernel)."""
    HEAD = "out"

    def __init__(self, state_size, hidden_size=256):
        super().__init__()
        self._build(2 * state_size, hidden_size, 1)


class Encoder(nn.Module):
    """nets/__init__.py:21-35: carries a dummy Linear(1,1) so optimizers are never empty."""

    def __init__(self):
        super().__init__()
        self.have_at_least_one_param = nn.Linear(1, 1)

    def forward_rolling(self, obs):
        return self.forward(obs)

    def reset_rolling(self):
        pass

    @property
    def embedding_dim(self):
        raise NotImplementedError


class IdentityEncoder(Encoder):
    """experiments/gym/train_gym.py:18-28: the state vector is the embedding."""

    def __init__(self, dim, key="obs"):
        super().__init__()
        self._dim = dim
        self.ssac_identity_key = key

    @property
    def embedding_dim(self):
        return self._dim

    def forward(self, obs_dict):
        return obs_dict[self.ssac_identity_key]


class _PixelEncoderBase(nn.Module):
    """parameter container of a convolutional encoder; arithmetic lives in conv_encoder.py"""

    def forward(self, obs):
        from . import conv_encoder
        engine.require_gpu(obs)
        eng = conv_encoder.ConvEncoderEngine(self, obs.device) if "_ssac_conv" not in self.__dict__ \
            else self.__dict__["_ssac_conv"]
        self.__dict__["_ssac_conv"] = eng
        out = torch.empty(obs.shape[0], self.embedding_dim, device=obs.device)
        eng.forward(obs.float(), out, self.embedding_dim, save=False)
        return out


def _conv_out(n, k, s):
    return (n - k) // s + 1


class BigPixelEncoder(_PixelEncoderBase):
    """cnns.py:37-69: conv3x3 s2, 3x conv3x3 s1 (32 ch), fc -> LayerNorm -> tanh; input x/255 - 0.5"""

    def __init__(self, obs_shape, out_dim=50):
        super().__init__()
        c, h, w = obs_shape
        self.conv1 = nn.Conv2d(c, 32, kernel_size=3, stride=2)
        self.conv2 = nn.Conv2d(32, 32, kernel_size=3, stride=1)
        self.conv3 = nn.Conv2d(32, 32, kernel_size=3, stride=1)
        self.conv4 = nn.Conv2d(32, 32, kernel_size=3, stride=1)
        h, w = _conv_out(h, 3, 2), _conv_out(w, 3, 2)
        for _ in range(3):
            h, w = _conv_out(h, 3, 1), _conv_out(w, 3, 1)
        self.fc = nn.Linear(h * w * 32, out_dim)
        self.ln = nn.LayerNorm(out_dim)
        self.apply(weight_init)
        self.embedding_dim = out_dim


class SmallPixelEncoder(_PixelEncoderBase):
    """cnns.py:72-103: conv8 s4 (32), conv4 s2 (64), conv3 s1 (64), fc; input x/255"""

    def __init__(self, obs_shape, out_dim=50):
        super().__init__()
        c, h, w = obs_shape
        self.conv1 = nn.Conv2d(c, 32, kernel_size=8, stride=4)
        self.conv2 = nn.Conv2d(32, 64, kernel_size=4, stride=2)
        self.conv3 = nn.Conv2d(64, 64, kernel_size=3, stride=1)
        for k, s in ((8, 4), (4, 2), (3, 1)):
            h, w = _conv_out(h, k, s), _conv_out(w, k, s)
        self.fc = nn.Linear(h * w * 64, out_dim)
        self.apply(weight_init)
        self.embedding_dim = out_dim


class PixelEncoder(Encoder):
    """Encoder wrapper in the shape of the reference scripts' DMCPixelEncoder / AtariEncoder
    (train_dmc_from_pixels.py:15-27, train_atari.py:9-20): obs_dict["obs"] -> conv block."""

    def __init__(self, conv_block, key="obs"):
        super().__init__()
        self.conv_block = conv_block
        self.ssac_obs_key = key

    @property
    def embedding_dim(self):
        return self.conv_block.embedding_dim

    def forward(self, obs_dict):
        return self.conv_block(obs_dict[self.ssac_obs_key])
