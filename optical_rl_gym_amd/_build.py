"""Build liborlgpu.so (HIP, gfx950) in-tree with hipcc.  No JIT cache: the .so sits next to this file so it
travels with the source tree to the GPU box.  Staleness is decided by a content hash of the sources (file times do
not survive being copied to another machine); concurrent builders (one process per GPU) serialise on a file lock."""
import fcntl
import hashlib
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "liborlgpu.so")
STAMP = LIB + ".stamp"
SOURCES = ["orl_gpu.hip", "orl_device.h", "orl_device_g8.h", "orl_log.h", "orl_log_data.h",
           os.path.join("..", "..", "include", "orl.h")]

# -ffp-contract=off: float64 statistics and the log restatement must round exactly like the reference
HIPCC_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-Wno-unused-value"]


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: the HIP library cannot be built")


def _extra():
    return os.environ.get("ORL_HIPCC_EXTRA", "").split()  # tuning experiments only (e.g. -DORL_STEP_WAVES=6)


def source_hash():
    h = hashlib.sha256()
    for s in SOURCES:
        with open(os.path.join(CSRC, s), "rb") as f:
            h.update(f.read())
    h.update(" ".join(HIPCC_FLAGS + _extra()).encode())
    return h.hexdigest()


def stale():
    if not os.path.exists(LIB) or not os.path.exists(STAMP):
        return True
    return open(STAMP).read().strip() != source_hash()


def build(force=False, verbose=False):
    if not force and not stale():
        return LIB
    with open(LIB + ".lock", "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not stale():  # another process built it while we waited
                return LIB
            tmp = LIB + ".tmp.%d" % os.getpid()
            cmd = [hipcc_path()] + HIPCC_FLAGS + _extra() + [os.path.join(CSRC, "orl_gpu.hip"), "-o", tmp]
            if verbose:
                print(" ".join(cmd))
            subprocess.check_call(cmd)
            os.replace(tmp, LIB)
            with open(STAMP, "w") as f:
                f.write(source_hash() + "\n")
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":
    print(build(force=True, verbose=True))
