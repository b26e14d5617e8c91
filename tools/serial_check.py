import os, sys, time
sys.path.insert(0, os.getcwd())
import optical_rl_gym_amd as orl
from bench import WORKLOADS
fam, topo, kw, policy = WORKLOADS["cfg2"]
B = 65536
for parts in ("1", "2"):
    os.environ["ORL_PERSIST_PARTS"] = parts
    env = orl.make(fam, topology=topo, num_envs=B, seeds=[10 + i for i in range(B)], **kw)
    env.run(policy, 1500)
    s0 = env.lib.orl_batch_debug_serial_count(env._h)
    t0 = time.perf_counter(); st = env.run(policy, 640); dt = time.perf_counter() - t0
    s1 = env.lib.orl_batch_debug_serial_count(env._h)
    print("parts", parts, "serial envs in 640 steps:", s1 - s0, "launches", st.launches, "ms", st.ms_total, "rate %.3e" % (B * 640 / dt))
    env.close()
