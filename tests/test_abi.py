"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/isbfsar.h
declares (no compute calls: there is no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built_lib():
    from isbfsar_amd.build import build
    return build(verbose=False)


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "isbfsar.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(isb_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported(built_lib):
    import ctypes
    lib = ctypes.CDLL(built_lib)
    syms = _declared_symbols()
    assert len(syms) >= 12
    for s in syms:
        assert hasattr(lib, s), f"{s} declared in include/isbfsar.h but not exported"


def test_ctypes_signatures_cover_header(built_lib):
    from isbfsar_amd import _lib
    assert set(_declared_symbols()) == set(_lib.SIGNATURES), set(_declared_symbols()) - set(_lib.SIGNATURES)
    h = _lib.lib()
    for name in _lib.SIGNATURES:           # an entry point without argtypes truncates int pointers
        assert getattr(h, name).argtypes is not None, name
    assert h.isb_version() == 2
    assert isinstance(h.isb_device_count(), int)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from isbfsar_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.IsbError, match="no CPU fallback"):
        _lib.lib()


def test_no_gpu_error_path(built_lib):
    """Without a device, create() must return an error code + message, not crash."""
    import ctypes as C
    from isbfsar_amd import _lib
    h = _lib.lib()
    if h.isb_device_count() > 0:
        pytest.skip("a GPU is visible")
    cfg = _lib.isb_ar_cfg(16, 30, 5, 0, 0, 0)
    out = C.c_void_p()
    rc = h.isb_ar_create(C.byref(cfg), C.byref(out))
    assert rc < 0 and h.isb_last_error()
    bad = _lib.isb_ar_cfg(1, 30, 5, 0, 0, 0)
    assert h.isb_ar_create(C.byref(bad), C.byref(out)) == -1


def test_product_does_not_import_oracle():
    """The product package must never route through oracle/ (or any CPU fallback)."""
    pkg = os.path.join(ROOT, "isbfsar_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), os.path.join(dp, f)


def test_checkpoint_key_migration():
    """state_from_torch accepts the three key dialects the reference has used for DISC.pth: DataParallel's
    `.module` infix (ar.py:18), the pre-RGB `features_extractor.fc1` names migrated by
    utils/rename_torch_layers_and_parameters.py:11, and the current names; the RGB-only `post_resnet.*` is dropped."""
    import numpy as np
    from isbfsar_amd import weights
    ref = weights.make_ar_state(16, 30, seed=3)
    old = {}
    for k, v in ref.items():
        k_old = k.replace("features_extractor.sk.", "features_extractor.")
        k_old = k_old.replace("transformers.0.", "transformers.module.0.", 1) if k_old.startswith("transformers.0.k_linear") else k_old
        old[k_old] = v
    old["post_resnet.l1.weight"] = np.zeros((256, 2048), np.float32)
    old["post_resnet.l1.bias"] = np.zeros(256, np.float32)
    got = weights.state_from_torch(old)
    assert set(got) == set(ref)
    for k in ref:
        np.testing.assert_array_equal(got[k], ref[k])
    # idempotent on current names
    again = weights.state_from_torch(ref)
    assert set(again) == set(ref)


def test_keras_variable_round_trip():
    """effnetv2.state_from_keras: TF layouts (HWIO kernels, [3,3,C,1] depthwise, 1x1 SE convs, BatchNorm statistics)
    -> blob tensors. Round trip through the inverse helper reproduces the state (variance 1-eps, mean 0 make the
    fold exact up to one float32 rounding)."""
    import numpy as np
    from isbfsar_amd import effnetv2
    st = effnetv2.make_state(1)
    kv = effnetv2.to_keras_variables(st)
    hk = np.ascontiguousarray(st["head.weight"].T)[None, None]
    back = effnetv2.state_from_keras({k + ":0": a for k, a in kv.items()}, hk, st["head.bias"])
    assert list(back) == list(st)
    for k in st:
        if k.endswith(".scale") or k.endswith(".shift"):
            np.testing.assert_allclose(back[k], st[k], rtol=2e-7, atol=0)
        else:
            np.testing.assert_array_equal(back[k], st[k])


def test_wspipe_staging_registers_are_private(built_lib):
    """gemm1x1_wspipe_kernel keeps global loads in flight in literally named registers (a[200:255] one wave per SIMD,
    v[228:255] two waves per SIMD; conv_ws.hip, wsp_request / wsp_to_lds). The register allocator only sees them as
    clobbers, so nothing but those requests and their LDS writes may name them in the code the compiler produced. The
    BUILD runs that disassembly check and compiles its verdict into the library (fail closed: unverified = the kernels
    are not selected); here the verdict must agree with a fresh run of the same check -- never skipped."""
    import ctypes
    from isbfsar_amd.build import OBJDUMP, wspipe_registers_private
    why = wspipe_registers_private(built_lib)
    verdict = ctypes.CDLL(built_lib).isb_wsreg_verified()
    assert verdict == (1 if why is None else 0), (verdict, why)
    if os.path.exists(OBJDUMP):
        assert why is None, why                    # this image has the tool: the shipped library must be the verified one


def test_mbfront8_counted_wait_is_guarded_by_the_build(built_lib):
    """mbfront8_kernel's `s_waitcnt vmcnt(5)` assumes exactly five vector-memory operations behind the next sample's
    input requests (conv_mb8.hip). The build checks that in the disassembly and compiles the verdict in (isb::mbf8_verified,
    fail closed: unverified = the expand GEMM + depthwise launches run); the shipped library must be the verified one, and the
    check must reject code it was written to reject (a spill, a sixth store)."""
    import ctypes
    from isbfsar_amd import build
    why = build.mbfront8_wait_counted(built_lib)
    if os.path.exists(build.OBJDUMP):
        assert why is None, why
    lib = ctypes.CDLL(built_lib)
    sym = next((s for s in ("_ZN3isb13mbf8_verifiedEv",) if hasattr(lib, s)), None)
    assert sym, "isb::mbf8_verified() not in the library"
    assert getattr(lib, sym)() == (1 if why is None else 0)
    # the checker on doctored text: one more store behind the requests, and a scratch instruction
    good = build._disassemble(built_lib)
    if good.startswith("\n"):
        lines = good.splitlines()
        orig = build._disassemble
        try:
            first = next(i for i, ln in enumerate(lines) if "mbfront8_kernel" in ln and ln.rstrip().endswith(">:"))
            end = next((i for i in range(first + 1, len(lines)) if lines[i].rstrip().endswith(">:")), len(lines))
            last_dma = max(i for i in range(first, end) if "global_load_lds_dwordx4" in lines[i])
            body_store = next(i for i in range(last_dma, end) if "global_store_dwordx4" in lines[i])      # one of the sample's five stores
            doctored = lines[:body_store] + [lines[body_store]] + lines[body_store:]
            build._disassemble = lambda _p: "\n".join(doctored)
            assert build.mbfront8_wait_counted(built_lib) is not None
            doctored = lines[:body_store] + ["\tscratch_store_dword off, v0, s0"] + lines[body_store:]
            build._disassemble = lambda _p: "\n".join(doctored)
            assert "scratch" in (build.mbfront8_wait_counted(built_lib) or "")
        finally:
            build._disassemble = orig


def test_every_fp16_storing_kernel_saturates(built_lib):
    """T16<true>::pack2 / from_f32 rely on MODE.FP16_OVFL (T16<F16>::enter()) instead of clamping; nothing used to
    enforce that a new fp16 kernel sets it. The build now refuses a library in which a kernel converts f32 -> fp16 without the mode
    bit, without an explicit clamp per conversion, and without an entry (with its reason) in build.FP16_NO_OVERFLOW; the check must
    pass on the shipped library and reject a kernel that lost its s_setreg."""
    from isbfsar_amd import build
    why = build.fp16_conversions_saturate(built_lib)
    if not os.path.exists(build.OBJDUMP):
        assert why is not None
        return
    assert why is None, why
    text = build._disassemble(built_lib)
    lines = text.splitlines()
    # doctor the disassembly: drop the mode write of the first mbfront16_kernel<.., true> instantiation
    start = next(i for i, ln in enumerate(lines) if "mbfront16_kernel" in ln and ln.rstrip().endswith(">:") and "Lb1" in ln)
    drop = next(i for i in range(start, len(lines)) if "s_setreg_imm32_b32" in lines[i])
    assert not any(ln.rstrip().endswith(">:") for ln in lines[start + 1:drop]), "the instantiation sets no mode bit"
    bad = "\n".join(lines[:drop] + lines[drop + 1:])
    why_bad = build.fp16_conversions_saturate(built_lib, text=bad)
    assert why_bad is not None and "mbfront16_kernel" in why_bad, why_bad


def test_hw_queues_are_asked_for_at_load_and_reported(built_lib):
    """isb_hw_queues: engines in flight on their own streams need a hardware queue each; the library asks the HIP
    runtime for eight when it is LOADED (GPU_MAX_HW_QUEUES, read once at the runtime's first call) unless the caller's environment
    already holds a value, and reports which of the two happened. No GPU needed: the call makes no HIP call."""
    import subprocess
    import sys
    prog = ("import ctypes, os, sys\n"
            f"lib = ctypes.CDLL({built_lib!r})\n"
            "src = ctypes.c_int32(-1)\n"
            "lib.isb_hw_queues.argtypes = [ctypes.POINTER(ctypes.c_int32)]\n"
            "print(lib.isb_hw_queues(ctypes.byref(src)), src.value)\n")
    env = {k: v for k, v in os.environ.items() if k != "GPU_MAX_HW_QUEUES"}
    out = subprocess.run([sys.executable, "-c", prog], env=env, capture_output=True, text=True, check=True).stdout.split()
    assert out == ["8", "2"], out                              # unset: the library set 8 at load time
    env["GPU_MAX_HW_QUEUES"] = "6"
    out = subprocess.run([sys.executable, "-c", prog], env=env, capture_output=True, text=True, check=True).stdout.split()
    assert out == ["6", "1"], out                              # the caller's value stands


def test_raw_integer_zero_precision_is_refused_by_the_python_engine():
    """ABI version 2 renumbered ISB_AR_PREC_BF16 from 0 to 3 (0 = the library's default = fp16 operands). A Python caller
    written against version 1 that passes the raw integer 0 must notice instead of silently getting another precision."""
    from isbfsar_amd.engine import ArEngine
    with pytest.raises(ValueError, match="ambiguous"):
        ArEngine(16, 30, 5, precision=0)
    with pytest.raises(ValueError):
        ArEngine(16, 30, 5, precision="fp32")
