"""Build liborlgpu.so (HIP, gfx950) in-tree with hipcc.  No JIT cache: the .so sits next to this file so it
travels with the source tree to the GPU box.  Staleness is decided by a content hash of EVERY file under csrc/ plus
include/orl.h, the flags and the compiler version (file times do not survive being copied to another machine);
concurrent builders (one process per GPU) serialise on a file lock.

The library is five translation units compiled in parallel: orl_api.hip (C ABI, W-independent kernels) and
orl_kernels.hip once per row width W in {1, 2, 5, 8} (the env kernels are templates over the number of 64-bit words of
a link row).  `variant="alt"` builds liborlgpu_alt.so with -DORL_ALT_IMPLS: the same library plus the two-kernel form of
the persistent kernel's phases, used only by the cross-implementation tests."""
import fcntl
import glob
import hashlib
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
INCLUDE = os.path.join(HERE, "..", "include", "orl.h")
ROW_WIDTHS = (1, 2, 5, 8)
VARIANTS = {"default": ("liborlgpu.so", []), "alt": ("liborlgpu_alt.so", ["-DORL_ALT_IMPLS"]),
            "timing": ("liborlgpu_timing.so", ["-DORL_TIMING=1"]),  # diagnostic: per-phase shader-clock profile (tools/phase_prof.py)
            "exp": ("liborlgpu_exp.so", [])}  # A/B experiments: built with ORL_HIPCC_EXTRA="-D..." (same value when loading)
LIB = os.path.join(HERE, VARIANTS["default"][0])

# -ffp-contract=off: float64 statistics and the log restatement must round exactly like the reference
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-unused-value"]


def lib_path(variant="default"):
    return os.path.join(HERE, VARIANTS[variant][0])


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP library cannot be built")


def _extra():
    return os.environ.get("ORL_HIPCC_EXTRA", "").split()  # tuning experiments only (e.g. -DORL_STEP_WAVES=6)


def sources():
    """Every file the build reads: all of csrc/ (headers included by the .hip units) and the public header."""
    files = sorted(p for p in glob.glob(os.path.join(CSRC, "*")) if os.path.isfile(p) and p.endswith((".h", ".hip")))
    return files + [INCLUDE]


_HIPCC_VERSION = None


def hipcc_version():
    global _HIPCC_VERSION
    if _HIPCC_VERSION is None:
        try:
            _HIPCC_VERSION = subprocess.run([hipcc_path(), "--version"], capture_output=True, text=True).stdout.strip()
        except Exception as exc:  # no compiler: an existing library with a matching stamp is still usable
            _HIPCC_VERSION = "unavailable: %s" % type(exc).__name__
    return _HIPCC_VERSION


def source_hash(variant="default", with_compiler=True):
    h = hashlib.sha256()
    for s in sources():
        h.update(os.path.basename(s).encode())
        with open(s, "rb") as f:
            h.update(f.read())
    h.update(" ".join(HIPCC_FLAGS + VARIANTS[variant][1] + _extra()).encode())
    text = h.hexdigest()
    if with_compiler:
        text += " " + hashlib.sha256(hipcc_version().encode()).hexdigest()[:16]
    return text


def stale(variant="default"):
    lib = lib_path(variant)
    if not os.path.exists(lib) or not os.path.exists(lib + ".stamp"):
        return True
    have = open(lib + ".stamp").read().split()
    want = source_hash(variant).split()
    if not have or have[0] != want[0]:
        return True
    # the compiler half of the stamp only counts where a compiler exists to rebuild with (the GPU box has the same image)
    return len(have) > 1 and len(want) > 1 and not hipcc_version().startswith("unavailable") and have[1] != want[1]


def build(force=False, verbose=False, variant="default"):
    lib = lib_path(variant)
    if not force and not stale(variant):
        return lib
    os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
    with open(os.path.join(HERE, "build", os.path.basename(lib) + ".lock"), "w") as lock:  # (not beside the library: the package directory holds sources and the built .so only)
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not stale(variant):  # another process built it while we waited
                return lib
            hipcc = hipcc_path()
            flags = HIPCC_FLAGS + VARIANTS[variant][1] + _extra()
            objdir = os.path.join(HERE, "build", variant)
            os.makedirs(objdir, exist_ok=True)
            units = [(os.path.join(CSRC, "orl_api.hip"), os.path.join(objdir, "orl_api.o"), [])]
            for w in ROW_WIDTHS:
                units.append((os.path.join(CSRC, "orl_kernels.hip"), os.path.join(objdir, "orl_kernels_w%d.o" % w), ["-DORL_W=%d" % w]))

            def compile_unit(u):
                src, obj, extra = u
                cmd = [hipcc] + flags + extra + ["-c", src, "-o", obj]
                if verbose:
                    print(" ".join(cmd), flush=True)
                subprocess.check_call(cmd)
                return obj

            with ThreadPoolExecutor(max_workers=min(len(units), os.cpu_count() or 1)) as pool:
                objs = list(pool.map(compile_unit, units))
            tmp = lib + ".tmp.%d" % os.getpid()
            cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", tmp]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            os.replace(tmp, lib)
            with open(lib + ".stamp", "w") as f:
                f.write(source_hash(variant) + "\n")
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return lib


SPEC_DIR = os.path.join(HERE, "build", "spec")


# Code-generation choices of the specialised kernels, measured on MI355X (profiles/r4*).  LLVM's machine-level
# loop-invariant code motion keeps ~35 constants and addresses in registers across the whole step loop: without it the cfg2
# kernel needs 126 VGPRs instead of 161-168 at the same speed, and the 4-wave forms (cfg1 / cfg3 / cfg5, 128-VGPR budget) have
# room for what the 3-wave forms keep in registers — the soon list and the early requests (ORL_PF_WAVES=4): cfg1 +12 %, cfg3
# +7 %, cfg5 +4 %.  RMCSA (3-wave form, 168 VGPRs, nothing to gain from a fourth wave at 16 384 envs) measured 1 % slower
# without the hoisting and keeps it.
SPEC_TUNING = ["-mllvm", "-disable-machine-licm", "-DORL_PF_WAVES=4"]


def spec_tuning(flags):
    if os.environ.get("ORL_SPEC_TUNING", "1") == "0":
        return []
    if "-DORL_SPEC_ENV=3" in flags.split():
        # RMCSA (168-VGPR form): room for 12 release times per round of the rebuild scan — its envs hold ~1 500 pending releases, 24
        # rounds of 8 (round 6, once the rounds really were rounds: cfg4 7.3 -> 7.6e8; the 128-VGPR forms gain nothing and spill at 16)
        return [] if "-DORL_SCAN_BATCH" in flags else ["-DORL_SCAN_BATCH=12"]
    if "-DORL_PF_WAVES" in flags:
        return []
    # (ORL_SPEC_PF_WAVES: experiments with the soon list back in memory in the 4-wave forms)
    return SPEC_TUNING[:2] + ["-DORL_PF_WAVES=%s" % os.environ.get("ORL_SPEC_PF_WAVES", "4")]


def spec_path(flags):
    """Where the specialisation library for these flags (orl_batch_spec_flags) lives: keyed by the flags, the sources and the
    compiler, so a stale one is never picked up."""
    src = source_hash()
    key = hashlib.sha256((flags + " " + " ".join(spec_tuning(flags)) + "|" + src).encode()).hexdigest()[:20]
    return os.path.join(SPEC_DIR, "liborlspec_%s_%s.so" % (src[:8], key))


def prune_specs():
    """Drop the cached specialisations of other source states (they can never be loaded again)."""
    keep = "liborlspec_%s_" % source_hash()[:8]
    if os.path.isdir(SPEC_DIR):
        for f in os.listdir(SPEC_DIR):
            if f.startswith("liborlspec_") and not f.startswith(keep):
                os.unlink(os.path.join(SPEC_DIR, f))


def build_spec(flags, verbose=False):
    """k_persist with one configuration's sizes as compile-time constants: csrc/orl_kernels.hip compiled with the -DORL_SPEC_*
    flags the library wrote for the batch, one kernel instantiation + its launch entry (~15 s of hipcc).  Returns the path of
    the cached library."""
    path = spec_path(flags)
    if os.path.exists(path):
        return path
    os.makedirs(SPEC_DIR, exist_ok=True)
    with open(os.path.join(SPEC_DIR, ".lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if os.path.exists(path):
                return path
            tmp = path + ".tmp.%d" % os.getpid()
            cmd = [hipcc_path()] + HIPCC_FLAGS + _extra() + flags.split() + spec_tuning(flags) + ["-shared", os.path.join(CSRC, "orl_kernels.hip"), "-o", tmp]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            os.replace(tmp, path)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return path


if __name__ == "__main__":
    import sys

    print(build(force=True, verbose=True, variant=sys.argv[1] if len(sys.argv) > 1 else "default"))
