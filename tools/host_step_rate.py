"""Host-driven policy()+step() rate for small batches under each step implementation (usage: host_step_rate.py)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import optical_rl_gym_amd as orl
kw = dict(load=300, mean_service_holding_time=25, episode_length=1000, num_spectrum_resources=320)
for B in (1, 64, 1024, 4096):
    for impl in ("64", "2"):
        os.environ["ORL_STEP_IMPL"] = impl
        env = orl.make("RMSA", topology="nsfnet_chen", num_envs=B, seeds=list(range(B)), **kw)
        for _ in range(50):
            env.step(env.policy("SAP_FF"), auto_reset=True)
        t0 = time.perf_counter()
        n = 300
        for _ in range(n):
            env.step(env.policy("SAP_FF"), auto_reset=True)
        dt = time.perf_counter() - t0
        print("B=%d impl=%s: %.0f host steps/s, %.3e env-steps/s" % (B, impl, n / dt, B * n / dt))
        env.close()
