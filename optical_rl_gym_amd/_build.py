"""Build liborlgpu.so (HIP, gfx950) in-tree with hipcc.  No JIT cache: the .so sits next to this file so it
travels with the source tree to the GPU box."""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "liborlgpu.so")
SOURCES = ["orl_gpu.hip", "orl_device.h", "orl_log.h", "orl_log_data.h", os.path.join("..", "..", "include", "orl.h")]

# -ffp-contract=off: float64 statistics and the log restatement must round exactly like the reference
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wno-unused-value"]


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP library cannot be built")


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(os.path.join(CSRC, s)) > t for s in SOURCES)


def build(force=False, verbose=False):
    if not force and not stale():
        return LIB
    extra = os.environ.get("ORL_HIPCC_EXTRA", "").split()  # tuning experiments only (e.g. -DORL_STEP_WAVES=6)
    cmd = [hipcc_path()] + HIPCC_FLAGS + extra + [os.path.join(CSRC, "orl_gpu.hip"), "-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
