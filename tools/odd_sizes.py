import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import optical_rl_gym_amd as orl
kw = dict(load=300, mean_service_holding_time=25, episode_length=30, num_spectrum_resources=320)
for fam, pol, extra in (("RMSA", "SAP_FF", {}), ("RWA", "SAP_FF", dict(num_spectrum_resources=80)), ("DeepRMSA", "SAP", None)):
    for B in (1, 7, 9, 100):
        seeds = list(range(50, 50 + B))
        if fam == "DeepRMSA":
            k2 = dict(mean_service_holding_time=7.5, mean_service_inter_arrival_time=1/12., j=1, episode_length=30, num_spectrum_resources=100)
        else:
            k2 = dict(kw, **extra)
        a = orl.make(fam, topology="nsfnet_chen", num_envs=B, seeds=seeds, **k2)
        b = orl.make(fam, topology="nsfnet_chen", num_envs=B, seeds=seeds, **k2)
        a.run(pol, 150)
        for _ in range(150):
            b.step(b.policy(pol), auto_reset=True)
        ok = np.array_equal(a.counters(), b.counters()) and np.array_equal(a.services(), b.services()) and all(
            np.array_equal(a.slots(i), b.slots(i)) and np.array_equal(a.link_stats(i), b.link_stats(i)) and np.array_equal(a.net_stats(i), b.net_stats(i)) for i in (0, B - 1))
        if fam == "DeepRMSA":
            ok = ok and np.array_equal(a.observation(), b.observation())
        print(fam, B, "OK" if ok else "MISMATCH")
        a.close(); b.close()
