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
    assert lib.orl_abi_version() == 1


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
