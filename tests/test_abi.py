"""The C-ABI library builds for gfx950 on a GPU-less host, loads, and exports every symbol include/orl.h declares.
No compute is attempted here (there is no GPU in the CPU suite); creating a batch must fail loudly."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "orl.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(orl_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from optical_rl_gym_amd import _build, _lib

    lib = _lib.lib()
    raw = ctypes.CDLL(_build.LIB)
    names = _declared()
    assert len(names) >= 20
    for n in names:
        assert hasattr(raw, n), "include/orl.h declares %s but liborlgpu.so does not export it" % n
        assert n in _lib.EXPORTS, "%s has no ctypes prototype in _lib.py" % n
    header = open(os.path.join(ROOT, "include", "orl.h")).read()
    declared_version = int(re.search(r"#define ORL_ABI_VERSION (\d+)", header).group(1))
    assert lib.orl_abi_version() == declared_version == _lib.ABI_VERSION == 2


def test_env_config_binding_matches_the_header_field_for_field():
    """The ctypes mirror of orl_env_config (and the stub shown in INTEGRATION.md) names the header's fields in the header's
    order, so a layout change cannot go unnoticed on either side; a wrong struct_size is refused before anything is read."""
    from optical_rl_gym_amd import _lib

    header = open(os.path.join(ROOT, "include", "orl.h")).read()
    body = re.search(r"typedef struct \{([^}]*)\} orl_env_config;", re.sub(r"/\*.*?\*/", "", header, flags=re.S)).group(1)
    fields = []
    for decl in body.split(";"):
        decl = decl.strip()
        if decl:
            names = decl.split(None, 1)[1] if not decl.startswith("const") else decl.split(None, 2)[2]
            fields += [n.strip().lstrip("*") for n in names.split(",")]
    assert [n for n, _ in _lib.EnvConfig._fields_] == fields
    stub = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    stub_cfg = stub[stub.index("class EnvConfig"):stub.index("def check(rc)")]
    assert re.findall(r'"([a-z_]+)"', stub_cfg) == fields
    # the ctypes layout is the C layout: 13 x 4 bytes, 4 of padding, 2 doubles, 7 pointers, 2 ints, 2 pointers
    assert ctypes.sizeof(_lib.EnvConfig) == 56 + 16 + 56 + 8 + 16
    lib = _lib.lib()
    cfg = _lib.EnvConfig()  # struct_size = 0: a client built against the round-2 header
    out = ctypes.c_void_p()
    fake = ctypes.c_void_p(1)
    assert lib.orl_batch_create_seeded(ctypes.byref(cfg), fake, 1, fake, ctypes.byref(out)) == -1
    assert b"struct_size" in lib.orl_last_error()


def test_no_cpu_fallback():
    """Without a GPU the product must refuse to run rather than fall back to anything."""
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import optical_rl_gym_amd as orl
    from optical_rl_gym_amd._lib import OrlError

    with pytest.raises(OrlError):
        orl.BatchedRMSAEnv("nsfnet_chen", num_envs=2, seeds=[1, 2], load=10, mean_service_holding_time=10)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "optical_rl_gym_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "from oracle" not in src and "import oracle" not in src and "orl_oracle" not in src, f


def test_staleness_hash_covers_every_included_source():
    """A stale .so must never ship with a matching stamp: every file a translation unit #includes (transitively, quoted
    form) is part of the content hash, and so are the flags and the compiler version."""
    from optical_rl_gym_amd import _build

    hashed = {os.path.realpath(p) for p in _build.sources()}
    seen, todo = set(), [os.path.join(_build.CSRC, f) for f in ("orl_api.hip", "orl_kernels.hip")]
    while todo:
        path = os.path.realpath(todo.pop())
        if path in seen:
            continue
        seen.add(path)
        assert path in hashed, "%s is compiled into liborlgpu.so but not hashed" % path
        for inc in re.findall(r'#include\s+"([^"]+)"', open(path).read()):
            todo.append(os.path.join(os.path.dirname(path), inc))
    assert len(seen) >= 8
    h0 = _build.source_hash()
    os.environ["ORL_HIPCC_EXTRA"] = "-DORL_SOMETHING"
    try:
        assert _build.source_hash() != h0
    finally:
        del os.environ["ORL_HIPCC_EXTRA"]
    assert _build.source_hash("alt") != h0
    assert len(h0.split()) == 2  # sources+flags, compiler version


def test_alt_library_exports_the_same_abi():
    from optical_rl_gym_amd import _lib

    assert _lib.lib("default").orl_build_has_alt() == 0
    assert _lib.lib("alt").orl_build_has_alt() == 1


def test_specialisation_flags_carry_the_kernel_form_for_the_batch_size(monkeypatch):
    """orl_spec_flags_for_batch (no device needed): the form of the persistent kernel is part of a specialisation's flags —
    batches of at most 12 288 envs of the single-core families get the two-wavefront form (-DORL_SPEC_RW=1) with everything in
    LDS at 3 waves per SIMD, large batches one wavefront per 8 envs; RMCSA never; QoSConstrainedRA has no persistent kernel."""
    import re

    from bench import WORKLOADS
    from optical_rl_gym_amd import envs

    for k in ("ORL_PERSIST_RW", "ORL_PERSIST_VARIANT", "ORL_PERSIST_INNER", "ORL_STEP_IMPL", "ORL_PERSIST"):
        monkeypatch.delenv(k, raising=False)

    def fields(name, batch):
        fam, topo, kw, _ = WORKLOADS[name]
        flags = envs.ENV_CLASSES[fam].spec_flags(batch=batch, topology=topo, **kw)
        return {k: int(v) for k, v in re.findall(r"-DORL_SPEC_(RW|LDS|WAVES)=(\d+)", flags)}

    for name in ("cfg1", "cfg2", "cfg3", "cfg5"):
        small, large = fields(name, 4096), fields(name, 1 << 20)
        assert small == dict(RW=1, LDS=1, WAVES=3), (name, small)
        assert large["RW"] == 0, (name, large)
        assert fields(name, 12288)["RW"] == (1 if name != "cfg5" else 0)  # (Germany50's window fits a CU four times, not six)
        assert fields(name, 12296)["RW"] == 0
    assert fields("cfg4", 4096)["RW"] == 0
    monkeypatch.setenv("ORL_PERSIST_RW", "0")
    assert fields("cfg2", 4096)["RW"] == 0
    monkeypatch.delenv("ORL_PERSIST_RW")
    qos = envs.ENV_CLASSES["QoSConstrainedRA"].spec_flags(batch=4096, topology="nsfnet_chen", num_spectrum_resources=32,
                                                          num_service_classes=2, classes_arrival_probabilities=[0.5, 0.5],
                                                          classes_reward=[2.0, 1.0])
    assert qos is None
