#!/usr/bin/env python3
"""Generate golden vectors by IMPORTING THE REFERENCE in the build container.

TEST INFRASTRUCTURE — runs only where /root/reference exists (never on the GPU box).
Outputs (all *data*, no reference source):
  tests/golden/*.npz                       traces of the reference's step() on fixed inputs
  optical_rl_gym_amd/data/<topology>.npz   flattened topology tables (paths, modulations)

Usage (keeps the read-only reference tree clean):
  cd /tmp && PYTHONDONTWRITEBYTECODE=1 MPLBACKEND=Agg python3 -W ignore /root/repo/oracle/gen_golden.py

What is captured per trace (SURVEY.md §8c G1–G9): the action fed to step(), reward, done,
the info floats, the pending service (arrival, holding, src, dst, bit_rate) before every
step, every integer counter after every step, a CRC32 of the full slot-availability array
after every step, and full snapshots (slot array, per-link statistics, network statistics)
at a few checkpoints.  Action streams come either from the reference's own heuristics or
from a seeded numpy stream generated here and stored in the fixture.
"""
import json
import os
import pickle
import sys
import zlib

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(HERE)
REF = os.environ.get("ORL_REFERENCE", "/root/reference")
sys.path.insert(0, REF)
sys.path.insert(0, os.path.join(HERE, "refshim"))
sys.dont_write_bytecode = True

import numpy as np  # noqa: E402

np.int = int  # the reference still uses the removed numpy alias (rwa_env.py:47, rmcsa_env.py:138)

import gym  # noqa: E402  (the shim)
import optical_rl_gym  # noqa: E402,F401
from optical_rl_gym.envs import deeprmsa_env, rmcsa_env, rmsa_env, rwa_env  # noqa: E402

GOLD = os.path.join(REPO, "tests", "golden")
DATA = os.path.join(REPO, "optical_rl_gym_amd", "data")
TOPO_DIR = os.path.join(REF, "examples", "topologies")

DEEPRMSA_NODE_PROBS = [
    0.01801802, 0.04004004, 0.05305305, 0.01901902, 0.04504505, 0.02402402, 0.06706707,
    0.08908909, 0.13813814, 0.12212212, 0.07607608, 0.12012012, 0.01901902, 0.16916917,
]  # tests/test_deeprmsa.py:30-47 (input data of the reference's own test)


def load_topology(name):
    with open(os.path.join(TOPO_DIR, name + "_5-paths_6-modulations.h5"), "rb") as f:
        return pickle.load(f)


# --------------------------------------------------------------------------------------
# topology flattening
# --------------------------------------------------------------------------------------
def flatten_topology(topo, out_name):
    nodes = list(topo.nodes())
    assert nodes == topo.graph["node_indices"]
    n = len(nodes)
    e = topo.number_of_edges()
    k = topo.graph["k_paths"]
    mods = list(topo.graph["modulations"])
    link_nodes = np.zeros((e, 2), np.int32)
    link_length = np.zeros(e, np.float64)
    link_ids = [None] * e
    edge_iter_order = np.zeros(e, np.int32)
    for it, (a, b) in enumerate(topo.edges()):
        idx = topo[a][b]["index"]
        edge_iter_order[it] = idx
        link_nodes[idx] = (nodes.index(a), nodes.index(b))
        link_length[idx] = topo[a][b]["length"]
        link_ids[idx] = str(topo[a][b]["id"])
    max_hops = max(p.hops for paths in topo.graph["ksp"].values() for p in paths)
    n_paths = np.zeros((n, n), np.int32)
    hops = np.zeros((n, n, k), np.int32)
    links = np.full((n, n, k, max_hops), -1, np.int32)
    length = np.zeros((n, n, k), np.float64)
    path_id = np.full((n, n, k), -1, np.int32)
    best_mod = np.full((n, n, k), -1, np.int32)
    path_nodes = np.full((n, n, k, max_hops + 1), -1, np.int32)
    for (s, d), paths in topo.graph["ksp"].items():
        si, di = nodes.index(s), nodes.index(d)
        n_paths[si, di] = len(paths)
        for ip, p in enumerate(paths):
            hops[si, di, ip] = p.hops
            assert p.hops == len(p.node_list) - 1
            length[si, di, ip] = p.length
            path_id[si, di, ip] = p.path_id
            best_mod[si, di, ip] = mods.index(p.best_modulation)
            for h in range(p.hops):
                links[si, di, ip, h] = topo[p.node_list[h]][p.node_list[h + 1]]["index"]
            for h, nd in enumerate(p.node_list):
                path_nodes[si, di, ip, h] = nodes.index(nd)
    np.savez_compressed(
        os.path.join(DATA, out_name + ".npz"),
        name=np.array(topo.graph["name"]),
        node_names=np.array(nodes),
        k_paths=np.int32(k),
        link_nodes=link_nodes,
        link_length=link_length,
        link_ids=np.array(link_ids),
        edge_iter_order=edge_iter_order,
        n_paths=n_paths,
        path_hops=hops,
        path_links=links,
        path_nodes=path_nodes,
        path_length=length,
        path_id=path_id,
        path_best_mod=best_mod,
        mod_name=np.array([m.name for m in mods]),
        mod_max_length=np.array([m.maximum_length for m in mods], np.float64),
        mod_se=np.array([m.spectral_efficiency for m in mods], np.int32),
        mod_min_osnr=np.array([m.minimum_osnr for m in mods], np.float64),
        mod_inband_xt=np.array([m.inband_xt for m in mods], np.float64),
    )
    print("topology", out_name, "N", n, "E", e, "k", k, "Hmax", max_hops,
          "iter==index", bool((edge_iter_order == np.arange(e)).all()))


# --------------------------------------------------------------------------------------
# trace capture
# --------------------------------------------------------------------------------------
def slots_array(env):
    g = env.topology.graph
    a = g["available_slots"] if "available_slots" in g else g["available_wavelengths"]
    return np.ascontiguousarray(a, dtype=np.uint8)


def svc_tuple(env):
    s = env.current_service
    return (s.arrival_time, s.holding_time, float(s.source_id), float(s.destination_id),
            float(s.bit_rate if s.bit_rate is not None else 0), float(s.service_id))


COUNTERS = ["services_processed", "services_accepted", "episode_services_processed",
            "episode_services_accepted", "bit_rate_requested", "bit_rate_provisioned",
            "episode_bit_rate_requested", "episode_bit_rate_provisioned"]


def counters(env):
    return [int(getattr(env, c, 0)) for c in COUNTERS]


def link_stats(env):
    t = env.topology
    keys = ["utilization", "external_fragmentation", "compactness", "last_update"]
    e = t.number_of_edges()
    out = np.zeros((4, e), np.float64)
    for a, b in t.edges():
        i = t[a][b]["index"]
        for kk, key in enumerate(keys):
            out[kk, i] = t[a][b].get(key, 0.0)
    return out


def net_stats(env):
    g = env.topology.graph
    return np.array([g.get("throughput", 0.0), g.get("compactness", 0.0), g.get("last_update", 0.0),
                     env.current_time], np.float64)


def run_trace(name, env, *, policy=None, actions=None, n_episodes=None, n_steps=None,
              info_keys, meta, obs_fn=None, snapshot_every=500, vec_info_keys=()):
    """Drive `env` the way utils.evaluate_heuristic does (reset() before each episode) and record."""
    rec = dict(actions=[], reward=[], done=[], info=[], svc=[], counters=[], crc=[], reset_before=[],
               n_active=[], obs=[])
    snaps, vec_info = {}, {}
    t = 0
    ep = 0
    episode_rewards = []
    stop = False
    while not stop:
        env.reset()  # soft reset, like evaluate_heuristic (utils.py:114)
        first = True
        done = False
        ep_rew = 0.0
        while not done:
            rec["svc"].append(svc_tuple(env))
            if obs_fn is not None:
                rec["obs"].append(np.asarray(obs_fn(env), np.float64))
            if policy is not None:
                a = policy(env)
            elif meta["env"] == "DeepRMSA":
                a = int(actions[t])
            else:
                a = [int(x) for x in actions[t]]
            a_arr = np.atleast_1d(np.asarray(a, dtype=np.int64))
            _, r, done, info = env.step(a)
            rec["actions"].append(a_arr)
            rec["reward"].append(r)
            rec["done"].append(done)
            rec["info"].append([float(info[k]) for k in info_keys])
            rec["counters"].append(counters(env))
            rec["crc"].append(zlib.crc32(slots_array(env).tobytes()))
            rec["reset_before"].append(first)
            rec["n_active"].append(len(env._events))
            first = False
            ep_rew += r
            t += 1
            if t % snapshot_every == 0:
                snaps[t] = (slots_array(env).copy(), link_stats(env), net_stats(env))
                for vk in vec_info_keys:
                    vec_info[(t, vk)] = np.asarray(info[vk], np.float64)
            if n_steps is not None and t >= n_steps:
                stop = True
                break
        if done:
            episode_rewards.append(ep_rew)
            ep += 1
        if n_episodes is not None and ep >= n_episodes:
            stop = True
    rec["svc"].append(svc_tuple(env))
    if obs_fn is not None:
        rec["obs"].append(np.asarray(obs_fn(env), np.float64))
    snaps[t] = (slots_array(env).copy(), link_stats(env), net_stats(env))
    for vk in vec_info_keys:
        vec_info[(t, vk)] = np.asarray(info[vk], np.float64)
    width = max(len(a) for a in rec["actions"])
    acts = np.full((t, width), -1, np.int64)
    for i, a in enumerate(rec["actions"]):
        acts[i, : len(a)] = a
    meta = dict(meta)
    meta.update(info_keys=list(info_keys), counters=COUNTERS, n_steps=t,
                episode_rewards=episode_rewards, snapshot_steps=sorted(snaps))
    out = dict(
        meta=np.array(json.dumps(meta)),
        actions=acts,
        reward=np.array(rec["reward"], np.float64),
        done=np.array(rec["done"], np.uint8),
        info=np.array(rec["info"], np.float64).reshape(t, len(info_keys)),
        svc=np.array(rec["svc"], np.float64),
        counters=np.array(rec["counters"], np.int64),
        crc=np.array(rec["crc"], np.uint32),
        reset_before=np.array(rec["reset_before"], np.uint8),
        n_active=np.array(rec["n_active"], np.int32),
    )
    if obs_fn is not None:
        out["obs"] = np.array(rec["obs"], np.float64)
    for s, (sl, ls, ns) in snaps.items():
        out["snap%d_slots" % s] = np.packbits(sl, axis=-1, bitorder="little")
        out["snap%d_link_stats" % s] = ls
        out["snap%d_net_stats" % s] = ns
    for (s, vk), v in vec_info.items():
        out["snap%d_%s" % (s, vk)] = v
    np.savez_compressed(os.path.join(GOLD, name + ".npz"), **out)
    print("%-38s steps %5d episodes %3d mean_reward %s" % (
        name, t, len(episode_rewards),
        ("%.4f +- %.4f" % (np.mean(episode_rewards), np.std(episode_rewards))) if episode_rewards else "-"))


RMSA_INFO = ["service_blocking_rate", "episode_service_blocking_rate", "bit_rate_blocking_rate",
             "episode_bit_rate_blocking_rate", "network_compactness", "network_compactness_difference",
             "avg_link_compactness", "avg_link_utilization"]
RMCSA_INFO = RMSA_INFO[:4]
RWA_INFO = RMSA_INFO[:2]


def random_actions(seed, n, k, s, *, extra=()):
    """Seeded action stream that hits every branch of step(): in-range (mostly low slots so that
    many are feasible), out-of-range path, out-of-range slot, the reject action."""
    rs = np.random.RandomState(seed)
    acts = np.zeros((n, 2 + len(extra)), np.int64)
    for i in range(n):
        u = rs.random_sample()
        if u < 0.05:
            p, sl = k, s
        elif u < 0.08:
            p, sl = k, rs.randint(s)
        elif u < 0.11:
            p, sl = rs.randint(k), s
        elif u < 0.60:
            p, sl = rs.randint(k), rs.randint(s)
        else:
            p, sl = rs.randint(min(k, 2)), rs.randint(max(1, s // 3))
        row = [p]
        for hi in extra:
            row.append(rs.randint(hi))
        row.append(sl)
        acts[i] = row
    return acts


def main():
    os.makedirs(GOLD, exist_ok=True)
    os.makedirs(DATA, exist_ok=True)
    nsf = load_topology("nsfnet_chen")
    ger = load_topology("germany50")
    flatten_topology(nsf, "nsfnet_chen_5-paths_6-modulations")
    flatten_topology(ger, "germany50_5-paths_6-modulations")

    # ---- G1: RNG / service-stream known answers for several seeds, all env families --------------
    for seed in (0, 10, 41, 123456789, 2**40 + 7):
        kw = dict(seed=seed, load=300, mean_service_holding_time=25, episode_length=1000,
                  num_spectrum_resources=320)
        env = gym.make("RMSA-v0", topology=nsf, **kw)
        run_trace("g1_rmsa_seed%d" % seed, env, policy=rmsa_env.shortest_available_path_first_fit,
                  n_steps=64, info_keys=RMSA_INFO,
                  meta=dict(env="RMSA", topology="nsfnet_chen", kwargs=kw, policy="SAP_FF"))

    # ---- G2: RMSA cfg2 trace under the reference's KSP-FF heuristic ------------------------------
    kw = dict(seed=10, load=300, mean_service_holding_time=25, episode_length=1000,
              num_spectrum_resources=320, allow_rejection=False)
    env = gym.make("RMSA-v0", topology=nsf, **kw)
    run_trace("g2_rmsa_cfg2_sapff", env, policy=rmsa_env.shortest_available_path_first_fit,
              n_steps=3000, info_keys=RMSA_INFO,
              meta=dict(env="RMSA", topology="nsfnet_chen", kwargs=kw, policy="SAP_FF"))
    for pol, fn in (("SP_FF", rmsa_env.shortest_path_first_fit), ("LLP_FF", rmsa_env.least_loaded_path_first_fit)):
        kw2 = dict(kw, load=400, allow_rejection=True)
        env = gym.make("RMSA-v0", topology=nsf, **kw2)
        run_trace("g2_rmsa_cfg2_%s" % pol.lower().replace("_", ""), env, policy=fn, n_steps=1500,
                  info_keys=RMSA_INFO, meta=dict(env="RMSA", topology="nsfnet_chen", kwargs=kw2, policy=pol))

    # ---- G3: stored random action stream (valid / busy / out-of-range / reject) ------------------
    kw = dict(seed=7, load=250, mean_service_holding_time=25, episode_length=200,
              num_spectrum_resources=320, allow_rejection=True)
    env = gym.make("RMSA-v0", topology=nsf, **kw)
    run_trace("g3_rmsa_random_actions", env, actions=random_actions(1234, 2500, 5, 320), n_steps=2500,
              info_keys=RMSA_INFO, meta=dict(env="RMSA", topology="nsfnet_chen", kwargs=kw, policy="ACTIONS"))

    # ---- G9: the reference's own RMSA test config (discrete bit rates, 64 slots) -----------------
    kw = dict(seed=10, allow_rejection=True, load=50, mean_service_holding_time=25, episode_length=100,
              num_spectrum_resources=64, bit_rate_selection="discrete")
    disc_keys = RMSA_INFO + ["bit_rate_blocking_10", "bit_rate_blocking_40", "bit_rate_blocking_100", "fairness"]
    for pol, fn in (("SP_FF", rmsa_env.shortest_path_first_fit),
                    ("SAP_FF", rmsa_env.shortest_available_path_first_fit),
                    ("LLP_FF", rmsa_env.least_loaded_path_first_fit)):
        env = gym.make("RMSA-v0", topology=nsf, **kw)
        run_trace("g9_rmsa_testcfg_%s" % pol.lower().replace("_", ""), env, policy=fn, n_episodes=10,
                  info_keys=disc_keys, meta=dict(env="RMSA", topology="nsfnet_chen", kwargs=kw, policy=pol))

    # ---- G7: Germany50 (edges() order != index order) --------------------------------------------
    kw = dict(seed=10, load=800, mean_service_holding_time=25, episode_length=1000,
              num_spectrum_resources=320, allow_rejection=False)
    env = gym.make("RMSA-v0", topology=ger, **kw)
    run_trace("g7_rmsa_germany50_sapff", env, policy=rmsa_env.shortest_available_path_first_fit,
              n_steps=1500, info_keys=RMSA_INFO,
              meta=dict(env="RMSA", topology="germany50", kwargs=kw, policy="SAP_FF"))

    # ---- G4: DeepRMSA (observations), j = 1, 2, 3 ------------------------------------------------
    for j in (1, 2, 3):
        kw = dict(seed=10, allow_rejection=False, mean_service_holding_time=7.5,
                  mean_service_inter_arrival_time=1.0 / 12.0, j=j, episode_length=50,
                  node_request_probabilities=DEEPRMSA_NODE_PROBS)
        env = gym.make("DeepRMSA-v0", topology=nsf, **dict(kw, node_request_probabilities=np.array(DEEPRMSA_NODE_PROBS)))
        run_trace("g4_deeprmsa_j%d_sap" % j, env, policy=deeprmsa_env.shortest_available_path_first_fit,
                  n_episodes=10 if j == 1 else 4, info_keys=RMSA_INFO, obs_fn=lambda e: e.observation(),
                  meta=dict(env="DeepRMSA", topology="nsfnet_chen", kwargs=kw, policy="SAP"))
    kw = dict(seed=10, allow_rejection=False, mean_service_holding_time=7.5,
              mean_service_inter_arrival_time=1.0 / 12.0, j=1, episode_length=50,
              node_request_probabilities=DEEPRMSA_NODE_PROBS)
    env = gym.make("DeepRMSA-v0", topology=nsf, **dict(kw, node_request_probabilities=np.array(DEEPRMSA_NODE_PROBS)))
    run_trace("g4_deeprmsa_j1_sp", env, policy=deeprmsa_env.shortest_path_first_fit, n_episodes=10,
              info_keys=RMSA_INFO, obs_fn=lambda e: e.observation(),
              meta=dict(env="DeepRMSA", topology="nsfnet_chen", kwargs=kw, policy="SP"))
    # stored random integer actions incl. the reject action, j = 2, higher load, rejection allowed
    kw = dict(seed=3, allow_rejection=True, mean_service_holding_time=25.0,
              mean_service_inter_arrival_time=0.1, j=2, episode_length=100, num_spectrum_resources=100)
    env = gym.make("DeepRMSA-v0", topology=nsf, **kw)
    rs = np.random.RandomState(99)
    acts = rs.randint(0, 5 * 2 + 1, size=(1200, 1)).astype(np.int64)
    run_trace("g4_deeprmsa_j2_random_actions", env, actions=acts[:, 0], n_steps=1200, info_keys=RMSA_INFO,
              obs_fn=lambda e: e.observation(),
              meta=dict(env="DeepRMSA", topology="nsfnet_chen", kwargs=kw, policy="ACTIONS"))

    # ---- G5: RWA, the reference's own test config ------------------------------------------------
    kw = dict(seed=10, allow_rejection=True, load=450, mean_service_holding_time=25, episode_length=1000)
    for pol, fn in (("SP_FF", rwa_env.shortest_path_first_fit),
                    ("SAP_FF", rwa_env.shortest_available_path_first_fit),
                    ("SAP_LF", rwa_env.shortest_available_path_last_fit),
                    ("LLP_FF", rwa_env.least_loaded_path_first_fit)):
        env = gym.make("RWA-v0", topology=nsf, **kw)
        run_trace("g5_rwa_testcfg_%s" % pol.lower().replace("_", ""), env, policy=fn,
                  n_episodes=10 if pol == "SAP_FF" else 3, info_keys=RWA_INFO, snapshot_every=1000,
                  vec_info_keys=("path_action_probability", "wavelength_action_probability"),
                  meta=dict(env="RWA", topology="nsfnet_chen", kwargs=kw, policy=pol))
    env = gym.make("RWA-v0", topology=nsf, **dict(kw, seed=5, load=600, episode_length=300))
    run_trace("g5_rwa_random_actions", env, actions=random_actions(77, 2000, 5, 80), n_steps=2000,
              info_keys=RWA_INFO, snapshot_every=1000,
              vec_info_keys=("path_action_probability", "wavelength_action_probability"),
              meta=dict(env="RWA", topology="nsfnet_chen", kwargs=dict(kw, seed=5, load=600, episode_length=300),
                        policy="ACTIONS"))

    # ---- G6: RMCSA, the reference's own test config and a 7 x 320 variant ------------------------
    # RMCSAEnv.__init__ mutates the Modulation objects of the topology it is given (+4 dB, rmcsa_env.py:127-129)
    # through the deep copy it owns, so every env gets a freshly loaded topology.
    kw = dict(seed=10, allow_rejection=True, load=250, mean_service_holding_time=25, episode_length=1000,
              num_spectrum_resources=64, num_spatial_resources=7, worst_xt=-84.7)
    env = gym.make("RMCSA-v0", topology=load_topology("nsfnet_chen"), **kw)
    run_trace("g6_rmcsa_testcfg_sapff", env,
              policy=rmcsa_env.shortest_available_path_best_modulation_first_core_first_fit, n_episodes=3,
              info_keys=RMCSA_INFO, snapshot_every=1000,
              meta=dict(env="RMCSA", topology="nsfnet_chen", kwargs=kw, policy="SAP_BM_FC_FF"))
    kw = dict(seed=11, allow_rejection=True, load=1500, mean_service_holding_time=25, episode_length=1000,
              num_spectrum_resources=320, num_spatial_resources=7)
    env = gym.make("RMCSA-v0", topology=load_topology("nsfnet_chen"), **kw)
    run_trace("g6_rmcsa_7x320_sapff", env,
              policy=rmcsa_env.shortest_available_path_best_modulation_first_core_first_fit, n_steps=1500,
              info_keys=RMCSA_INFO, snapshot_every=500,
              meta=dict(env="RMCSA", topology="nsfnet_chen", kwargs=kw, policy="SAP_BM_FC_FF"))
    kw = dict(seed=12, allow_rejection=True, load=400, mean_service_holding_time=25, episode_length=250,
              num_spectrum_resources=100, num_spatial_resources=7)
    env = gym.make("RMCSA-v0", topology=load_topology("nsfnet_chen"), **kw)
    run_trace("g6_rmcsa_random_actions", env, actions=random_actions(4321, 2000, 5, 100, extra=(6, 7)),
              n_steps=2000, info_keys=RMCSA_INFO, snapshot_every=500,
              meta=dict(env="RMCSA", topology="nsfnet_chen", kwargs=kw, policy="ACTIONS"))


if __name__ == "__main__":
    main()
