"""Import harness for the upstream reference (TEST INFRASTRUCTURE, dev container only).

The reference package lives read-only at /root/reference and depends on packages
that are absent from this image (gin, gymnasium, gym, tensorboardX, cv2, skimage,
torchvision).  None of them is touched by the update path, so we register empty
stand-in modules in ``sys.modules`` before importing it *unmodified*.

Only ``oracle/gen_golden.py`` and the opt-in ``tests/test_oracle_vs_reference.py``
use this file; nothing on the product path does, and nothing here ever travels
to the GPU box in a usable form (``/root/reference`` does not exist there).
"""
import sys
import types

REFERENCE_ROOT = "/root/reference"


def _mod(name, **attrs):
    m = types.ModuleType(name)
    for k, v in attrs.items():
        setattr(m, k, v)
    sys.modules[name] = m
    return m


def _configurable(*args, **kwargs):
    # gin.configurable is used both bare (@gin.configurable) and called
    # (@gin.configurable(denylist=[...])).
    if len(args) == 1 and callable(args[0]) and not kwargs:
        return args[0]
    return lambda f: f


class _Dummy:
    def __init__(self, *a, **k):
        pass


def install_stubs():
    if "gin" not in sys.modules:
        _mod("gin", configurable=_configurable, parse_config_file=lambda *a, **k: None,
             REQUIRED=object())
    for base in ("gymnasium", "gym"):
        if base in sys.modules:
            continue
        spaces = _mod(base + ".spaces", Box=_Dummy, Discrete=_Dummy, Dict=_Dummy)
        wrappers = _mod(base + ".wrappers", TimeLimit=_Dummy, AtariPreprocessing=_Dummy,
                        FrameStack=_Dummy)
        _mod(base, Wrapper=_Dummy, ActionWrapper=_Dummy, ObservationWrapper=_Dummy,
             RewardWrapper=_Dummy, Env=_Dummy, spaces=spaces, wrappers=wrappers,
             make=lambda *a, **k: None)
    if "tensorboardX" not in sys.modules:
        _mod("tensorboardX", SummaryWriter=_Dummy)
    if "cv2" not in sys.modules:
        _mod("cv2")
    if "skimage" not in sys.modules:
        tr = _mod("skimage.transform", resize=lambda *a, **k: None)
        shp = _mod("skimage.util.shape", view_as_windows=lambda *a, **k: None)
        ut = _mod("skimage.util", shape=shp)
        _mod("skimage", transform=tr, util=ut)
    if "torchvision" not in sys.modules:
        tv_t = _mod("torchvision.transforms")
        _mod("torchvision", transforms=tv_t)
    if "stable_baselines3" not in sys.modules:
        vec = _mod("stable_baselines3.common.vec_env", SubprocVecEnv=_Dummy, DummyVecEnv=_Dummy)
        com = _mod("stable_baselines3.common", vec_env=vec)
        _mod("stable_baselines3", common=com)


def import_reference():
    """Return the unmodified reference package (``super_sac``)."""
    sys.dont_write_bytecode = True
    install_stubs()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import super_sac  # noqa: F401
    from super_sac import (agent, augmentations, learning, learning_utils, nets,  # noqa: F401
                           popart, replay)
    return super_sac
