"""Randomised configurations: every step implementation must leave EVERY env in the same state (debug / soak aid): the
one-wavefront-per-env kernel, the persistent kernel (generic instantiation; with ORL_JIT_SPEC=1 in the environment also the
one built for the configuration), and — RMSA / DeepRMSA — a loop of agent-driven single steps (k_agent) in the middle of the
run.  usage: fuzz_cross.py [seed] [cases]"""
import os, sys
os.environ.setdefault("ORL_JIT_SPEC", "0")
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import optical_rl_gym_amd as orl

rng = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 1)
n_cases = int(sys.argv[2]) if len(sys.argv) > 2 else 12
bad = 0
for case in range(n_cases):
    fam = rng.choice(["RMSA", "RMSA", "DeepRMSA", "RWA", "RMCSA"])
    topo = rng.choice(["nsfnet_chen", "nsfnet_chen", "germany50", "cost239"])
    B = int(rng.choice([600, 2048, 5000, 20000]))
    steps = int(rng.choice([150, 400, 1200]))
    kw = dict(episode_length=int(rng.choice([7, 40, 1000])), mean_service_holding_time=float(rng.choice([5.0, 25.0])))
    if fam == "RWA":
        kw.update(load=float(rng.choice([50, 450])), num_spectrum_resources=int(rng.choice([17, 80, 130])),
                  allow_rejection=bool(rng.randint(2)))
        policy = str(rng.choice(["SAP_FF", "SP_FF", "LLP_FF", "SAP_LF"]))
    elif fam == "RMCSA":
        kw.update(load=float(rng.choice([300, 1500])), num_spectrum_resources=int(rng.choice([100, 320])),
                  num_spatial_resources=7, allow_rejection=True)
        policy = "SAP_BM_FC_FF"
    elif fam == "DeepRMSA":
        kw.pop("mean_service_holding_time")
        kw.update(mean_service_holding_time=7.5, mean_service_inter_arrival_time=1.0 / float(rng.choice([4, 12])),
                  j=int(rng.choice([1, 2, 4])), num_spectrum_resources=int(rng.choice([100, 200])))
        policy = "SAP"
    else:
        kw.update(load=float(rng.choice([60, 300, 700])), num_spectrum_resources=int(rng.choice([64, 100, 320, 500])),
                  allow_rejection=bool(rng.randint(2)))
        if rng.randint(3) == 0:
            kw.update(bit_rate_selection="discrete")
        policy = str(rng.choice(["SAP_FF", "SP_FF", "LLP_FF"]))
    seeds = [int(s) for s in rng.randint(0, 2**31 - 1, B)]
    out = {}
    try:
        for v in ("64", "1", "2", "rd"):  # ("rd": the rows-deferred form of the persistent kernel where it applies, round 6)
            os.environ["ORL_STEP_IMPL"] = "2" if v == "rd" else v
            os.environ["ORL_AGENT_STEP"] = "1" if v in ("2", "rd") else "0"
            if v == "rd":
                os.environ["ORL_PERSIST_VARIANT"] = str(7 + int(rng.randint(2)))
                os.environ["ORL_PERSIST_RW"] = "0"
            else:
                os.environ.pop("ORL_PERSIST_VARIANT", None)
                os.environ.pop("ORL_PERSIST_RW", None)
            env = orl.make(fam, topology=topo, num_envs=B, seeds=seeds, **kw)
            env.run(policy, steps // 2)
            n_host = 12 if fam in ("RMSA", "DeepRMSA") else 0  # in the middle: host- / agent-driven steps (v == "2": k_agent)
            for _ in range(n_host):
                env.policy(policy, fetch=False)
                env.step(None, auto_reset=True, fetch=False)
            env.run(policy, steps - steps // 2 - n_host)
            env.check()
            out[v] = (env.counters().copy(), env.services().copy(), env.active().copy(), env.flags().copy(),
                      env.net_stats_all().copy(), env.link_stats_all().copy(), env.slots_packed().copy())
            env.close()
    except Exception as exc:  # configuration not supported by the host side: report and go on
        print("case", case, fam, topo, kw, "->", type(exc).__name__, exc)
        continue
    ok = all(np.array_equal(out["64"][k], out[v][k], equal_nan=True) if out["64"][k].dtype.kind == "f" else np.array_equal(out["64"][k], out[v][k])
             for v in ("1", "2", "rd") for k in range(7))
    bad += 0 if ok else 1
    print("case", case, fam, topo, B, steps, policy, kw, "OK" if ok else "MISMATCH", "flags", int(out["64"][3].any()))
print("mismatches:", bad)
