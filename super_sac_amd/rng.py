"""Host random streams of the update path, kept on the SAME generators the reference uses
so a run that swaps in this engine consumes them identically:

  replay indices, DrQ shifts      torch CPU default generator   (replay.py:122, augmentations.py:227)
  REDQ target subset, logged net  Python ``random``             (agent.py:29, learning.py:135)
  action noise eps                generator of the compute device (distributions: Normal.sample)

Parity tests replace these functions to replay the draws recorded in tests/golden.
"""
import random

import torch


def draw_indices(n, batch_size):
    return torch.randint(n, (batch_size,))


def draw_subset(num_critics, k):
    return random.sample(range(num_critics), k=k)


def choice(seq):
    return random.choice(seq)


def draw_normal(shape, device):
    return torch.randn(*shape, device=device)


def draw_normal_into(dst):
    """standard-normal draw written in place (same device-generator consumption as torch.randn)."""
    return dst.normal_()


_stock_draw_normal, _stock_draw_normal_into = draw_normal, draw_normal_into


def normal_is_stock():
    """True while no test hook replaces the device-noise draws: only then may a consumer switch to the engine's
    in-kernel Philox stream ("device noise from an engine Philox stream; parity tests inject eps")."""
    return draw_normal is _stock_draw_normal and draw_normal_into is _stock_draw_normal_into


def draw_categorical(logits):
    """Categorical(logits=logits).sample() on the logits' device (learning_utils.py:386-388, softmax backup weights of
    a discrete agent): index plumbing on torch's device generator, as the reference draws it."""
    return torch.distributions.Categorical(logits=logits).sample()


def draw_permutation(n):
    """torch.randperm(n) on the CPU default generator (the negatives of the contrastive Markov loss, learning.py:300)."""
    return torch.randperm(n)


def draw_drqv2_shift(batch_size, pad):
    return torch.randint(0, 2 * pad + 1, size=(batch_size, 1, 1, 2))


def draw_drq_offsets(batch_size, pad):
    w1 = torch.randint(0, pad * 2, (batch_size,))
    h1 = torch.randint(0, pad * 2, (batch_size,))
    return w1, h1
