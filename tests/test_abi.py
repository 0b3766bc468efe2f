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
    assert _lib.lib.ssac_abi_version() == 1


def test_layout_matches_header_contract():
    from super_sac_amd import engine
    stride, offs = engine.mlp_layout(23, 256, 1)
    assert offs == [0, 5888, 6144, 71680, 71936, 72192]
    assert stride == 72196 and stride % 4 == 0  # 72193 params padded to a multiple of 4


def test_struct_sizes_match_the_c_side():
    from super_sac_amd import _lib
    assert ctypes.sizeof(_lib.AdamCtl) == 72 and ctypes.sizeof(_lib.PopArtState) == 40
    assert ctypes.sizeof(_lib.MlpDesc) == 32 and ctypes.sizeof(_lib.Feed) == 48


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
    fake = types.SimpleNamespace(learning=types.SimpleNamespace(), learning_utils=types.SimpleNamespace())
    ssa.install(fake)
    assert fake.learning.critic_update is ssa.learning.critic_update
    assert fake.learning.alpha_update is ssa.learning.alpha_update
    assert fake.learning_utils.soft_update is ssa.learning_utils.soft_update
