import os, sys, time
sys.path.insert(0, os.getcwd())
import optical_rl_gym_amd as orl
B = 65536
kw = dict(load=300, mean_service_holding_time=25, episode_length=1000, num_spectrum_resources=320, bit_rate_selection="discrete")
env = orl.make("RMSA", topology="nsfnet_chen", num_envs=B, seeds=[10 + i for i in range(B)], **kw)
env.run("SAP_FF", 1500)
for n in (20, 300):
    env.sync(); t0 = time.perf_counter(); env.run("SAP_FF", n); env.sync(); env.run("SAP_FF", n); env.sync()
    dt = (time.perf_counter() - t0) / 2
    print("discrete bit rates, %d-step runs: %.3e env-steps/s" % (n, B * n / dt))
