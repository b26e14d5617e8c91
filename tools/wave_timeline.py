#!/usr/bin/env python3
"""Where a 20-step launch of the persistent kernel spends its wall time (diagnostic build liborlgpu_timing.so): per wavefront the
constant 100 MHz clock at entry, at the first step, after the last step and at the end of the write-back.

    python3 tools/wave_timeline.py [workload] [batch] [steps]
"""
import math
import os
import sys

SPEC = os.environ.get("WT_SPEC", "0") == "1"
if SPEC:
    os.environ["ORL_JIT_SPEC"] = "1"
    os.environ["ORL_SPEC_EXTRA"] = (os.environ.get("ORL_SPEC_EXTRA", "") + " -DORL_TIMING=1").strip()
else:
    os.environ["ORL_LIB_VARIANT"] = "timing"
    os.environ["ORL_JIT_SPEC"] = "0"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import optical_rl_gym_amd as orl  # noqa: E402
from bench import WORKLOADS, workload_load  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
fam, topo, kw, policy = WORKLOADS[name]
env = orl.make(fam, topology=topo, num_envs=B, seeds=[10 + i for i in range(B)], **kw)
if SPEC:
    import ctypes as C
    from optical_rl_gym_amd import _build
    buf = C.create_string_buffer(1024)
    assert env.lib.orl_batch_spec_flags(env._h, buf, 1024) > 0
    spec = C.CDLL(_build.spec_path(buf.value.decode() + " " + os.environ["ORL_SPEC_EXTRA"]))
    spec.orl_spec_prof.argtypes = [C.c_void_p, C.c_int]


def prof(arr, mode):
    env.sync()
    if SPEC:
        assert spec.orl_spec_prof(arr.ctypes.data, mode) == 0
    else:
        env._ck(env.lib.orl_batch_debug_prof(env._h, arr.ctypes.data, mode))


env.run(policy, max(1500, int(math.ceil(5 * workload_load(kw)))))
for _ in range(5):
    st = env.run(policy, steps)
zero = np.zeros(48, np.uint64)
prof(zero, 1)
st = env.run(policy, steps)
raw = np.zeros(16384 * 48, np.uint64)
prof(raw, 3)
ts = np.zeros(16384 * 8, np.uint64)
prof(ts, 2)
waves = (B + 7) // 8
hw = ts.reshape(-1, 8)[:waves, 4:6].astype(np.int64)
t = ts.reshape(-1, 8)[:waves, :4].astype(np.int64)
t0 = t[:, 0].min()
t = (t - t0) / 100.0  # us
print("%s B=%d, %d-step launch: kernel %.1f us (HIP events); %d wavefronts" % (name, B, steps, st.ms_total * 1e3, waves))
print("last wavefront ends at %.1f us after the first one enters" % t[:, 3].max())


def q(a):
    return " ".join("%7.1f" % v for v in np.percentile(a, [0, 1, 10, 50, 90, 99, 100]))


print("                         min      p1     p10     p50     p90     p99     max")
print("entry                %s" % q(t[:, 0]))
print("first step           %s" % q(t[:, 1]))
print("after last step      %s" % q(t[:, 2]))
print("end                  %s" % q(t[:, 3]))
print("load  (entry->first) %s" % q(t[:, 1] - t[:, 0]))
print("loop                 %s" % q(t[:, 2] - t[:, 1]))
print("store (loop->end)    %s" % q(t[:, 3] - t[:, 2]))
first = t[:, 0] < 30.0


def ms(a):
    return (a.mean(), a.std()) if a.size else (0.0, 0.0)


print("wavefronts entering in the first 30 us: %d; loop time of those %.1f (sd %.1f), of the others %.1f (sd %.1f)" %
      ((first.sum(),) + ms(t[first, 2] - t[first, 1]) + ms(t[~first, 2] - t[~first, 1])))
# how many wavefronts are in their loop at time x
edges = np.arange(0, t[:, 3].max() + 20, 20.0)
print("time us: wavefronts loading / in the loop / storing")
for x in edges:
    print("  %6.0f  %5d %5d %5d" % (x, ((t[:, 0] <= x) & (x < t[:, 1])).sum(), ((t[:, 1] <= x) & (x < t[:, 2])).sum(), ((t[:, 2] <= x) & (x < t[:, 3])).sum()))
loop = t[:, 2] - t[:, 1]
pr = raw.reshape(-1, 48)[:waves].astype(np.float64) / 2400.0  # us at 2.4 GHz
print("per-wavefront phase time (us) over the launch: mean, sd, correlation with the wavefront's loop time, regression slope")
for k in np.argsort(-pr.std(axis=0)):
    if pr[:, k].std() > 0.5:
        c = np.corrcoef(pr[:, k], loop)[0, 1]
        print("  slot %2d  mean %7.1f  sd %6.1f  corr %5.2f" % (k, pr[:, k].mean(), pr[:, k].std(), c))
print("sum of the phases: mean %.1f sd %.1f; loop mean %.1f sd %.1f" % (pr.sum(axis=1).mean(), pr.sum(axis=1).std(), loop.mean(), loop.std()))
# position effects: by XCD (blockIdx % 8) and by launch order
for x in range(8):
    m = np.arange(waves) % 8 == x
    print("  blockIdx %% 8 == %d: loop mean %.1f sd %.1f" % (x, loop[m].mean(), loop[m].std()))
wave_id, simd, cu, sh, se = hw[:, 0] & 15, (hw[:, 0] >> 4) & 3, (hw[:, 0] >> 8) & 15, (hw[:, 0] >> 12) & 1, (hw[:, 0] >> 13) & 7
xcc = hw[:, 1] & 15
g1 = t[:, 0] < 30.0
for nm, v in (("wave slot", wave_id), ("simd", simd), ("cu", cu), ("sh", sh), ("se", se), ("xcc", xcc)):
    print("by %s (first generation | second):" % nm)
    for x in np.unique(v):
        a, b = loop[g1 & (v == x)], loop[~g1 & (v == x)]
        print("   %2d: n %5d mean %7.1f sd %5.1f | n %5d mean %7.1f sd %5.1f" % (x, a.size, a.mean() if a.size else 0, a.std() if a.size else 0,
                                                                              b.size, b.mean() if b.size else 0, b.std() if b.size else 0))
# the SIMD as the unit: mean loop time of the first-generation wavefronts per (xcc, se, sh, cu, simd)
key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
for gname, gm in (("first", g1), ("second", ~g1)):
    ks = np.unique(key[gm])
    if not ks.size:
        continue
    means = np.array([loop[gm & (key == k)].mean() for k in ks])
    within = np.array([loop[gm & (key == k)].std() for k in ks])
    print("%s generation: %d SIMDs; sd of the SIMD means %.1f, mean sd within a SIMD %.1f" % (gname, ks.size, means.std(), within.mean()))
keyc = key // 4
ks = np.unique(keyc[g1])
means = np.array([loop[g1 & (keyc == k)].mean() for k in ks])
print("first generation: %d CUs; sd of the CU means %.1f; histogram of the first generation's loop times:" % (ks.size, means.std()))
h, e = np.histogram(loop[g1], bins=20)
for i in range(20):
    print("   %7.1f %5d" % (e[i], h[i]))
env.close()
