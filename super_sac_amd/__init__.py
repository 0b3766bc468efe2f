"""super_sac_amd -- MI355X-native update engine behind the super_sac interface.

Same module / function names as the reference package for the one path it replaces:

    super_sac_amd.Agent                         <- super_sac/agent.py
    super_sac_amd.learning.critic_update ...    <- super_sac/learning.py
    super_sac_amd.learning_utils.*              <- super_sac/learning_utils.py
    super_sac_amd.replay.ReplayBuffer           <- super_sac/replay.py (sample path)
    super_sac_amd.augmentations.*               <- super_sac/augmentations.py (DrQ family)
    super_sac_amd.popart.PopArtLayer            <- super_sac/popart.py
    super_sac_amd.nets.*                        <- super_sac/nets/

All arithmetic runs in hand-written HIP kernels (libssac_hip.so, gfx950); importing the
package fails if the library has not been built -- there is no fallback path.
"""
import torch

device = torch.device("cuda") if torch.cuda.is_available() else "cpu"

from . import _lib  # noqa: E402,F401  (raises ImportError when the HIP library is missing)
from . import rng  # noqa: E402,F401
from . import engine  # noqa: E402,F401
from . import nets  # noqa: E402,F401
from . import popart  # noqa: E402,F401
from . import replay  # noqa: E402,F401
from . import augmentations  # noqa: E402,F401
from . import agent  # noqa: E402,F401
from .agent import Agent  # noqa: E402,F401
from . import learning_utils  # noqa: E402,F401
from . import learning  # noqa: E402,F401
from . import adv_estimator  # noqa: E402,F401
from . import checkpoint  # noqa: E402,F401
from . import conv_encoder  # noqa: E402,F401
from . import parallel  # noqa: E402,F401
from . import adopt  # noqa: E402,F401
from .adopt import adopt_agent, adopt_augmenter, adopt_buffer  # noqa: E402,F401
from .engine import set_precision, sync_shadows  # noqa: E402,F401


INSTALLED_AUGMENTATIONS = ("AugmentationSequence", "Drqv2Aug", "DrqAug", "DrqNoNoiseAug", "LargeDrqAug",
                           "LargeDrqNoNoiseAug", "IdentityAug")


def install(reference_package):
    """Rebind the reference's update seam to this engine:

        import super_sac, super_sac_amd
        super_sac_amd.install(super_sac)

    ``super_sac.main.super_sac`` resolves ``learning.*`` / ``lu.*`` through the module objects at call
    time (main.py:18-19), so patching the module attributes is sufficient.

    The two classes the training scripts construct themselves are rebound as well: ``super_sac.replay.ReplayBuffer``
    (resolved at call time: experiments/gym/train_gym.py:84, dmc/train_dmc_from_pixels.py:62, atari/train_atari.py:41)
    and the DrQ-family / identity augmentations with their ``AugmentationSequence`` (main.py:138-141 builds the default
    through the module; a script that did ``from super_sac.augmentations import ...`` BEFORE install() hands over
    reference-built objects, which the update functions adopt in place -- adopt.adopt_augmenter / adopt_buffer)."""
    ref_learning, ref_lu = reference_package.learning, reference_package.learning_utils
    import sys
    ref_name = getattr(reference_package, "__name__", "super_sac")
    ref_replay = getattr(reference_package, "replay", None) or sys.modules.get(ref_name + ".replay")
    ref_aug = getattr(reference_package, "augmentations", None) or sys.modules.get(ref_name + ".augmentations")
    if ref_replay is not None:
        ref_replay.ReplayBuffer = replay.ReplayBuffer
    if ref_aug is not None:
        for name in INSTALLED_AUGMENTATIONS:
            setattr(ref_aug, name, getattr(augmentations, name))
    for name in ("critic_update", "online_actor_update", "alpha_update", "offline_actor_update",
                 "markov_state_abstraction_update"):
        setattr(ref_learning, name, getattr(learning, name))
    for name in ("soft_update", "hard_update", "sample_move_and_augment", "compute_td_targets",
                 "compute_backup_weights", "adjust_priorities", "compute_filter_stats"):
        setattr(ref_lu, name, getattr(learning_utils, name))
    return reference_package
