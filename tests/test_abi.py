"""CPU: the C-ABI library loads and exports every symbol include/ssac_hip.h declares
(no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "ssac_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ssac_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    for must in ("ssac_mlp_layer_fwd", "ssac_mlp_layer_dgrad", "ssac_mlp_layer_wgrad", "ssac_td_target",
                 "ssac_gather_transition", "ssac_drq_shift", "ssac_polyak", "ssac_alpha_update"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(os.path.join(ROOT, "super_sac_amd", "libssac_hip.so"))
    missing = [s for s in declared_symbols() if not hasattr(lib, s)]
    assert not missing, f"declared in ssac_hip.h but not exported: {missing}"


def test_python_binding_covers_every_symbol():
    from super_sac_amd import _lib
    assert sorted(_lib.SIGNATURES) == declared_symbols()
    import re
    header = open(os.path.join(ROOT, "include", "ssac_hip.h")).read()
    declared = int(re.search(r"#define\s+SSAC_ABI_VERSION\s+(\d+)", header).group(1))
    assert _lib.lib.ssac_abi_version() == _lib.ABI_VERSION == declared


def test_layout_matches_header_contract():
    from super_sac_amd import engine
    stride, offs = engine.mlp_layout(23, 256, 1)
    assert offs == [0, 5888, 6144, 71680, 71936, 72192]
    assert stride == 72196 and stride % 4 == 0  # 72193 params padded to a multiple of 4


def test_struct_sizes_match_the_c_side(tmp_path):
    """every ctypes mirror in _lib.py against sizeof() of the struct in include/ssac_hip.h, as gcc lays it out (the
    header is plain C; a field added on one side only shows up here, on the CPU)"""
    import subprocess
    from super_sac_amd import _lib
    pairs = {"ssac_mlp": _lib.MlpDesc, "ssac_adam_ctl": _lib.AdamCtl, "ssac_popart": _lib.PopArtState,
             "ssac_feed": _lib.Feed, "ssac_rng": _lib.Rng, "ssac_td_spec": _lib.TdSpec,
             "ssac_push_field": _lib.PushField, "ssac_logfold": _lib.LogFold,
             "ssac_deferred_logs": _lib.DeferredLogs, "ssac_gather": _lib.Gather}
    src = tmp_path / "sizes.c"
    src.write_text('#include <stdio.h>\n#include "ssac_hip.h"\nint main(void) {\n' +
                   "".join(f'    printf("{n} %zu\\n", sizeof({n}));\n' for n in pairs) + "    return 0;\n}\n")
    exe = tmp_path / "sizes"
    subprocess.run(["gcc", "-std=c11", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    sizes = dict((ln.split()[0], int(ln.split()[1])) for ln in out.strip().splitlines())
    for name, cls in pairs.items():
        assert ctypes.sizeof(cls) == sizes[name], (name, ctypes.sizeof(cls), sizes[name])
    assert sizes["ssac_adam_ctl"] == 72 and sizes["ssac_popart"] == 40 and sizes["ssac_mlp"] == 32


def test_update_path_refuses_to_run_without_a_gpu():
    import pytest
    import torch
    from super_sac_amd import engine
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError):
        engine.require_gpu()


def test_install_rebinds_the_reference_seam():
    import types
    import super_sac_amd as ssa
    fake = types.SimpleNamespace(learning=types.SimpleNamespace(), learning_utils=types.SimpleNamespace(),
                                 replay=types.SimpleNamespace(), augmentations=types.SimpleNamespace())
    ssa.install(fake)
    assert fake.replay.ReplayBuffer is ssa.replay.ReplayBuffer
    assert fake.augmentations.Drqv2Aug is ssa.augmentations.Drqv2Aug
    assert fake.augmentations.AugmentationSequence is ssa.augmentations.AugmentationSequence
    assert fake.learning.critic_update is ssa.learning.critic_update
    assert fake.learning.alpha_update is ssa.learning.alpha_update
    assert fake.learning_utils.soft_update is ssa.learning_utils.soft_update
