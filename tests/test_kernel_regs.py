"""Code-object checks (no GPU needed: hipcc cross-compiles gfx950 and llvm-readelf reads the kernels' metadata).  The BASELINE
specialisations `__graft_entry__.build()` pre-builds must need no scratch memory at all — no spilled VGPR, private segment 0 — and the
generic library's persistent kernels stay within a bound (round-5 verdict: the headline kernel had crept to 2 spilled VGPRs / 12 B of
scratch, the generic RMCSA 4-wave forms to 34 / 104 B; those forms are not built any more)."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

pytestmark = pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")


def _kernels(path, name):
    import kernel_regs

    out = []
    for k in kernel_regs.kernels(path):
        full = kernel_regs.demangle(k["name"])
        if full.startswith("void " + name + "<") or full.startswith(name + "<"):
            out.append((full.split("(")[0].replace("void ", ""), k))
    return out


def test_baseline_specialisations_need_no_scratch():
    from bench import WORKLOADS
    from optical_rl_gym_amd import _build, envs

    seen = set()
    checked = 0
    for name, (fam, topo, kw, _policy) in WORKLOADS.items():
        for batch in (1 << 20, 4096):
            flags = envs.ENV_CLASSES[fam].spec_flags(batch=batch, topology=topo, **kw)
            if not flags or flags in seen:
                continue
            seen.add(flags)
            lib = _build.build_spec(flags)
            ks = _kernels(lib, "k_persist")
            assert ks, "%s: no k_persist in %s" % (name, lib)
            for full, k in ks:
                assert int(k["vgpr_spill_count"]) == 0 and int(k["private_segment_fixed_size"]) == 0, \
                    "%s (%d envs): %s spills %s VGPRs, %s B of scratch" % (name, batch, full, k["vgpr_spill_count"], k["private_segment_fixed_size"])
                checked += 1
    assert checked >= 8


def test_generic_library_persistent_kernels_spill_little():
    from optical_rl_gym_amd import _build

    lib = _build.build()
    ks = _kernels(lib, "k_persist")
    assert len(ks) >= 40
    worst = max(int(k["vgpr_spill_count"]) for _full, k in ks)
    for full, k in ks:
        # (template arguments: env family, row width, LDS state, waves per SIMD, ...)
        args = [a.strip() for a in full[full.index("<") + 1:full.rindex(">")].split(",")]
        assert not (args[0] == "3" and args[3] == "4"), "RMCSA is not built in the 4-wave forms (25-34 spilled VGPRs there): " + full
        assert int(k["vgpr_spill_count"]) <= 12, "%s: %s spilled VGPRs" % (full, k["vgpr_spill_count"])
    assert worst <= 12
