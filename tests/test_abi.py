"""CPU: the C-ABI library loads and exports every symbol include/ssac_hip.h declares
(no compute calls -- there is no GPU here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


INCLUDE = os.path.join(ROOT, "include")


def _strip(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def _split_lab(text):
    """(text outside #ifdef SSAC_LAB ... #endif, text inside)"""
    inside = "".join(re.findall(r"#ifdef\s+SSAC_LAB\b(.*?)#endif", text, flags=re.S))
    outside = re.sub(r"#ifdef\s+SSAC_LAB\b.*?#endif", "", text, flags=re.S)
    return outside, inside


def prototypes(text):
    """{name: (return type, [parameter C types])} of every `ssac_*` function prototype in (comment-free) header text"""
    out = {}
    for m in re.finditer(r"([A-Za-z_][A-Za-z0-9_ \t\*]*?)\b(ssac_[a-z0-9_]+)\s*\(([^;{}()]*)\)\s*;", text):
        ret, name, args = m.group(1).strip(), m.group(2), m.group(3).strip()
        if ret.startswith("typedef") or not ret:
            continue
        params = [] if args in ("", "void") else [a.strip() for a in args.split(",")]
        types = []
        for a in params:
            a = re.sub(r"\[[^\]]*\]", "*", a)                        # array parameter == pointer
            m2 = re.fullmatch(r"(.*?)([A-Za-z_][A-Za-z0-9_]*)?\s*", a)   # drop the parameter name
            ty = a if "*" in a and a.rstrip().endswith("*") else m2.group(1)
            types.append(re.sub(r"\s+", " ", ty.replace("const", "")).replace(" *", "*").strip())
        out[name] = (re.sub(r"\s+", " ", ret.replace("const", "")).replace(" *", "*").strip(), types)
    return out


def header_prototypes():
    prod = prototypes(_strip(open(os.path.join(INCLUDE, "ssac_hip.h")).read()))
    test_all = _strip(open(os.path.join(INCLUDE, "ssac_hip_test.h")).read())
    outside, inside = _split_lab(test_all)
    return prod, prototypes(outside), prototypes(inside)


def declared_symbols():
    return sorted(header_prototypes()[0])


def _ctype_class(t):
    """coarse class of a ctypes argument type, comparable with _c_class of the header's C type"""
    if t is None:
        return "void"
    if t in (ctypes.c_void_p, ctypes.c_char_p) or hasattr(t, "contents"):   # void* / char* / POINTER(...)
        return "ptr"
    return {ctypes.c_int: "i32", ctypes.c_int32: "i32", ctypes.c_uint32: "i32", ctypes.c_int64: "i64", ctypes.c_longlong: "i64",
            ctypes.c_uint64: "i64", ctypes.c_size_t: "i64", ctypes.c_float: "f32", ctypes.c_double: "f64"}[t]


def _c_class(ty):
    if ty.endswith("*"):
        return "ptr"
    return {"int": "i32", "int32_t": "i32", "uint32_t": "i32", "unsigned": "i32", "int64_t": "i64", "long long": "i64",
            "unsigned long long": "i64", "uint64_t": "i64", "size_t": "i64", "float": "f32", "double": "f64", "void": "void"}[ty]


def test_header_declares_the_expected_surface():
    syms = declared_symbols()
    for must in ("ssac_mlp_layer_fwd", "ssac_mlp_layer_dgrad", "ssac_mlp_layer_wgrad", "ssac_td_target",
                 "ssac_gather_transition", "ssac_drq_shift", "ssac_polyak", "ssac_alpha_update"):
        assert must in syms
    # the drop-in header carries no form switches and no lab hooks (round-5 review, weak 10): those live in ssac_hip_test.h
    prod, test, lab = header_prototypes()
    for name in ("ssac_wgrad_variant", "ssac_chain_form", "ssac_bf16_fwd_form", "ssac_gemm_lean", "ssac_xcd_order"):
        assert name not in prod and name in test
    for name in ("ssac_fused_debug_stamps", "ssac_gemm_debug_stamps", "ssac_debug_timeline", "ssac_xchg_test_mode"):
        assert name not in prod and name not in test and name in lab


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(os.path.join(ROOT, "super_sac_amd", "libssac_hip.so"))
    prod, test, lab = header_prototypes()
    missing = [s for s in list(prod) + list(test) if not hasattr(lib, s)]
    assert not missing, f"declared in ssac_hip.h / ssac_hip_test.h but not exported: {missing}"
    leaked = [s for s in lab if hasattr(lib, s)]
    assert not leaked, f"lab hooks defined by the PRODUCT library: {leaked}"


def test_python_binding_covers_every_symbol():
    from super_sac_amd import _lib
    prod, test, lab = header_prototypes()
    assert sorted(_lib.SIGNATURES) == sorted(prod)
    assert sorted(_lib.TEST_SIGNATURES) == sorted(test)
    assert sorted(_lib.LAB_SIGNATURES) == sorted(lab)
    header = open(os.path.join(ROOT, "include", "ssac_hip.h")).read()
    declared = int(re.search(r"#define\s+SSAC_ABI_VERSION\s+(\d+)", header).group(1))
    assert _lib.lib.ssac_abi_version() == _lib.ABI_VERSION == declared


def test_python_binding_matches_every_prototype():
    """argument COUNT and C type class (pointer / 32-bit int / 64-bit int / float / double) of every ctypes signature
    against the prototype parsed from the headers, and the return types: a changed signature is caught here, on the
    CPU, not by a crash on the GPU box (round-5 review, weak 10)."""
    from super_sac_amd import _lib
    prod, test, lab = header_prototypes()
    bound = {**_lib.SIGNATURES, **_lib.TEST_SIGNATURES, **_lib.LAB_SIGNATURES}
    assert len(bound) >= 150 and set(bound) == set(prod) | set(test) | set(lab)
    bad = []
    for name, (ret, ctys) in {**prod, **test, **lab}.items():
        args = bound[name]
        if len(args) != len(ctys):
            bad.append(f"{name}: {len(args)} ctypes arguments, the header declares {len(ctys)}")
            continue
        for k, (a, c) in enumerate(zip(args, ctys)):
            if _ctype_class(a) != _c_class(c):
                bad.append(f"{name}: argument {k} is {a.__name__} in _lib.py, `{c}` in the header")
        want = _c_class(ret)
        got = _ctype_class(_lib._RESTYPES.get(name, ctypes.c_int))
        if want != got:
            bad.append(f"{name}: returns `{ret}` in the header, {got} in _lib.py")
    assert not bad, "\n".join(bad)


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
             "ssac_deferred_logs": _lib.DeferredLogs, "ssac_gather": _lib.Gather, "ssac_actor_logfold": _lib.ActorLogFold}
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
