"""GPU parity: the HIP path (through the C ABI) against (1) the golden traces captured from the reference
and (2) the CPU oracle run live on larger batches.  Integers compare with ==; float64 values compare with ==
too (the device evaluates the reference's expressions in the same order and its log() is glibc's, restated)."""
import numpy as np
import pytest

from tests.helpers import golden_names, load_golden, replay, replay_h, replay_q, replay_w

pytestmark = pytest.mark.gpu

# Device-resident runs go through the persistent kernel (k_persist) wherever it applies, host-driven step() through the
# one-wavefront-per-env kernel (k_step).  The small parity cases run against every implementation by forcing it (the
# variables are read when a batch is created): "wave64" = k_step for everything, "persist" = the default, "split2" = the
# phases of the persistent kernel as two separate launches (liborlgpu_alt.so, the -DORL_ALT_IMPLS build).
IMPLS = ["wave64", "split2", "persist", "persist_global", "persist_lds", "agent8", "persist_pair", "persist_rd"]
IMPL_ENV = {"wave64": dict(ORL_STEP_IMPL="64", ORL_PERSIST="0", ORL_LIB_VARIANT="default", ORL_PERSIST_VARIANT=None, ORL_PERSIST_INNER=None),
            "split2": dict(ORL_STEP_IMPL="2", ORL_PERSIST="0", ORL_LIB_VARIANT="alt", ORL_PERSIST_VARIANT=None, ORL_PERSIST_INNER=None),
            # the persistent kernel in the form the library picks, with all state in global memory, and with slot maps +
            # link statistics + per-core sums resident in LDS (a form only liborlgpu_alt.so carries; where it does not fit
            # the library's default form runs) and the per-row cache of inner free runs switched on (the library uses it only
            # where it costs no wavefront per CU)
            "persist": dict(ORL_STEP_IMPL="2", ORL_PERSIST="1", ORL_LIB_VARIANT="default", ORL_PERSIST_VARIANT=None, ORL_PERSIST_INNER=None),
            "persist_global": dict(ORL_STEP_IMPL="2", ORL_PERSIST="1", ORL_LIB_VARIANT="default", ORL_PERSIST_VARIANT="0", ORL_PERSIST_INNER=None),
            "persist_lds": dict(ORL_STEP_IMPL="2", ORL_PERSIST="1", ORL_LIB_VARIANT="alt", ORL_PERSIST_VARIANT="2", ORL_PERSIST_INNER="2"),
            # host- / agent-driven steps through k_agent (the phases of the persistent kernel for one step, with info) whatever
            # the batch size — the library takes it from 2 048 envs — for all four families (RMSA, DeepRMSA, RWA, RMCSA: every
            # g* / w* / h* fixture of theirs replays through it); device-resident runs as "persist"
            "agent8": dict(ORL_STEP_IMPL="2", ORL_PERSIST="1", ORL_LIB_VARIANT="default", ORL_PERSIST_VARIANT=None, ORL_PERSIST_INNER=None,
                           ORL_AGENT_STEP="1"),
            # the two-wavefront form of the persistent kernel (a control and a row wavefront per 8 envs; the library takes it for
            # batches of at most 12 288 envs of the single-core families) at every batch size: it exists in specialisation libraries
            # only, so one is built for every configuration (RMCSA: the one-wavefront kernel, specialised)
            "persist_pair": dict(ORL_STEP_IMPL="2", ORL_PERSIST="1", ORL_LIB_VARIANT="default", ORL_PERSIST_VARIANT="4", ORL_PERSIST_INNER=None,
                                 ORL_PERSIST_RW="1", ORL_JIT_SPEC="1"),
            # the rows-deferred form (round 6): the loop is slot scan + control phase, which changes the slot maps itself and logs an
            # event per provision / release; k_rowstats replays link statistics and compactness sums after every launch, one lane per
            # link row (single-core families with at most 64 links; elsewhere the library's own choice runs)
            "persist_rd": dict(ORL_STEP_IMPL="2", ORL_PERSIST="1", ORL_LIB_VARIANT="default", ORL_PERSIST_VARIANT="7", ORL_PERSIST_INNER=None,
                               ORL_PERSIST_RW="0")}
for _name, _env in IMPL_ENV.items():
    _env.setdefault("ORL_AGENT_STEP", None)
    _env.setdefault("ORL_PERSIST_RW", None)
    _env.setdefault("ORL_JIT_SPEC", None)


def force_impl(monkeypatch, name):
    for k, v in IMPL_ENV[name].items():
        if v is None:
            monkeypatch.delenv(k, raising=False)
        else:
            monkeypatch.setenv(k, v)


@pytest.fixture(params=IMPLS)
def impl(request, monkeypatch):
    force_impl(monkeypatch, request.param)
    return request.param


# (tests that only take host-driven steps: the forms of the device-resident loop that differ in nothing else are left out)
@pytest.fixture(params=[v for v in IMPLS if v not in ("persist_pair", "persist_rd")])
def impl_host(request, monkeypatch):
    force_impl(monkeypatch, request.param)
    return request.param


def _ran_pair_form(env):
    """The last device-resident run of `env` was launches of the two-wavefront kernel (debug query: 2)."""
    return int(env.lib.orl_batch_debug_persist_spec(env._h)) == 2


# Shards of a multi-device batch: both on the box's only GPU, and — wherever a box has them — on two distinct GPUs (the
# driver's GPU box has one: that case is skipped there and runs on any multi-GPU machine).
DEVICE_PAIRS = [pytest.param((0, 0), id="one_gpu"), pytest.param((0, 1), id="two_gpus")]


def _need_devices(devs):
    from optical_rl_gym_amd import _lib

    n = int(_lib.lib().orl_device_count())
    if max(devs) >= n:
        pytest.skip("needs %d GPUs, this box has %d" % (max(devs) + 1, n))


def _product(meta, num_envs=1, seeds=None, **extra):
    import optical_rl_gym_amd as orl

    kw = dict(meta["kwargs"])
    seed = kw.pop("seed")
    kw.update(extra)
    return orl.make(meta["env"], topology=meta["topology"], num_envs=num_envs,
                    seeds=[seed] if seeds is None else seeds, **kw)


def _exact(name):
    def check(t, what, got, exp):
        got, exp = np.asarray(got), np.asarray(exp)
        if got.dtype.kind == "f" or exp.dtype.kind == "f":
            ok = np.array_equal(got.astype(np.float64), exp.astype(np.float64), equal_nan=True)
        else:
            ok = np.array_equal(got, exp)
        assert ok, "%s: step %d: %s differs\n got %r\n exp %r" % (name, t, what, got, exp)
    return check


@pytest.mark.parametrize("name", golden_names())
def test_hip_reproduces_reference_trace(name, impl_host):
    g = load_golden(name)
    env = _product(g["meta"])
    replay(env, g, _exact(name))
    assert not env.flags().any()
    env.close()


CASES = [
    # (golden whose kwargs to reuse, policy, batch, steps)
    ("g2_rmsa_cfg2_sapff", "SAP_FF", 192, 400),
    ("g2_rmsa_cfg2_llpff", "LLP_FF", 64, 300),
    ("g9_rmsa_testcfg_sapff", "SAP_FF", 64, 300),
    ("g7_rmsa_germany50_sapff", "SAP_FF", 48, 250),
    ("g4_deeprmsa_j2_sap", "SAP", 96, 300),
    ("g5_rwa_testcfg_sapff", "SAP_FF", 64, 400),
    ("g5_rwa_testcfg_saplf", "SAP_LF", 32, 300),
    ("g5_rwa_testcfg_llpff", "LLP_FF", 32, 300),
    ("g6_rmcsa_7x320_sapff", "SAP_BM_FC_FF", 48, 300),
]


@pytest.mark.parametrize("gname,policy,batch,steps", CASES)
def test_hip_matches_oracle_on_batches(gname, policy, batch, steps, impl_host):
    from oracle.oracle import OracleBatch

    meta = load_golden(gname)["meta"]
    kw = dict(meta["kwargs"])
    kw.pop("seed")
    kw["episode_length"] = 60  # many episode boundaries -> auto (soft) reset path
    seeds = [1000 + 7 * i for i in range(batch)]
    ora = OracleBatch(meta["env"], meta["topology"], seeds, **kw)
    dev = _product(dict(meta, kwargs=dict(kw, seed=0)), num_envs=batch, seeds=seeds)
    chk = _exact(gname)
    for t in range(steps):
        a_o = ora.policy(policy)
        a_d = dev.policy(policy)
        chk(t, "actions", a_d, a_o)
        obs_o, r_o, d_o, i_o = ora.step(a_o, auto_reset=True)
        obs_d, r_d, d_d, i_d = dev.step(a_d, auto_reset=True)
        chk(t, "reward", r_d, r_o)
        chk(t, "done", d_d, d_o)
        chk(t, "info", i_d, i_o)
        if obs_o is not None:
            chk(t, "obs", obs_d, obs_o)
        if t % 50 == 49 or t == steps - 1:
            chk(t, "services", dev.services(), ora.services())
            chk(t, "counters", dev.counters(), ora.counters())
            for e in (0, batch // 2, batch - 1):
                chk(t, "slots", dev.slots(e), ora.slots(e))
                chk(t, "link_stats", dev.link_stats(e), ora.link_stats(e))
                chk(t, "net_stats", dev.net_stats(e), ora.net_stats(e))
                chk(t, "n_active", dev.n_active(e), ora.n_active(e))
    assert not dev.flags().any()
    dev.close()


def test_device_resident_run_matches_stepwise(impl):
    """orl_batch_run (policy+step loop on the device, no host round trips) == host-driven policy()/step()."""
    meta = load_golden("g2_rmsa_cfg2_sapff")["meta"]
    kw = dict(meta["kwargs"])
    kw.pop("seed")
    kw["episode_length"] = 80
    seeds = list(range(500, 500 + 128))
    a = _product(dict(meta, kwargs=dict(kw, seed=0)), num_envs=128, seeds=seeds)
    b = _product(dict(meta, kwargs=dict(kw, seed=0)), num_envs=128, seeds=seeds)
    a.run("SAP_FF", 200)
    assert _ran_pair_form(a) or impl != "persist_pair"  # (the library's own choice for a small batch when a specialisation is cached)
    for _ in range(200):
        b.step(b.policy("SAP_FF"), auto_reset=True)
    chk = _exact("run")
    chk(0, "counters", a.counters(), b.counters())
    chk(0, "services", a.services(), b.services())
    for e in (0, 64, 127):
        chk(0, "slots", a.slots(e), b.slots(e))
        chk(0, "link_stats", a.link_stats(e), b.link_stats(e))
    a.close()
    b.close()


def test_gym_front_end_reproduces_reference_script_numbers():
    """tests/test_rmsa.py / test_deeprmsa.py of the reference, run through the product's gym-shaped classes."""
    import optical_rl_gym_amd as orl

    kw = dict(allow_rejection=True, load=50, mean_service_holding_time=25, episode_length=100,
              num_spectrum_resources=64, bit_rate_selection="discrete")
    for heur, mean, std in ((orl.shortest_path_first_fit, 88.7, 7.1281),
                            (orl.shortest_available_path_first_fit, 95.0, 3.2558),
                            (orl.least_loaded_path_first_fit, 95.1, 3.3897)):
        env = orl.RMSAEnv(topology="nsfnet_chen", seed=10, **kw)
        m, s = orl.evaluate_heuristic(env, heur, n_eval_episodes=10)
        assert (round(float(m), 4), round(float(s), 4)) == (mean, std)
        env.close()
    g = load_golden("g4_deeprmsa_j1_sap")
    dkw = dict(g["meta"]["kwargs"])
    dkw.pop("seed")
    env = orl.DeepRMSAEnv(topology="nsfnet_chen", seed=10, **dkw)
    m, s = orl.evaluate_heuristic(env, orl.shortest_available_path_first_fit, n_eval_episodes=10)
    assert (round(float(m), 4), round(float(s), 4)) == (43.2, 4.6)
    env.close()


@pytest.mark.parametrize("workload,batch", [("cfg2", 65536), ("cfg5", 32768), ("cfg1", 4096), ("cfg3", 4096), ("cfg4", 16384),
                                            ("cfg2", 4096)])
def test_full_size_batch_sampled_envs_match_oracle(workload, batch):
    """Every BASELINE.json configuration at its size, with the bench's own kwargs (cfg4: COST239, 7 cores x 320 slots, load
    1500; cfg5: the per-GPU shard of the 8-GPU batch).  Envs are independent, so env i of the big batch must equal a 1-env
    oracle run with seed_i; 24 sampled envs are compared in full (slot map, link statistics, counters, pending service,
    DeepRMSA observation), and cheap invariants are checked on every env."""
    import optical_rl_gym_amd as orl
    from bench import WORKLOADS
    from oracle.oracle import OracleBatch

    fam, topo, kw, policy = WORKLOADS[workload]
    kw = dict(kw, episode_length=100)
    steps = 260
    seeds = [10 + i for i in range(batch)]
    dev = orl.make(fam, topology=topo, num_envs=batch, seeds=seeds, **kw)
    dev.run(policy, steps)
    sample = sorted(set([0, 1, 63, 64, 4095, batch // 2, batch - 1] + list(np.random.RandomState(5).randint(0, batch, 17))))
    ora = OracleBatch(fam, topo, [seeds[i] for i in sample], **kw)
    ora.run(policy, steps)
    chk = _exact(workload)
    cd, sd, ad = dev.counters(), dev.services(), dev.active()
    chk(0, "counters", cd[sample], ora.counters())
    chk(0, "services", sd[sample], ora.services())
    for j, i in enumerate(sample):
        chk(i, "slots", dev.slots(i), ora.slots(j))
        chk(i, "link_stats", dev.link_stats(i), ora.link_stats(j))
        chk(i, "net_stats", dev.net_stats(i), ora.net_stats(j))
        chk(i, "n_active", int(ad[i]), ora.n_active(j))
    if dev.obs_dim:
        chk(0, "observation", dev.observation()[sample], ora.observation())
    assert not dev.flags().any()
    first = 0 if fam == "RWA" else (0 if fam == "RMCSA" else 1)  # RMSA / DeepRMSA count a service when it is created
    assert (cd[:, 0] == steps + first).all() and (cd[:, 1] <= cd[:, 0]).all() and (cd[:, 5] <= cd[:, 4]).all()
    assert (ad >= 0).all() and (ad <= cd[:, 1]).all()
    p, a = dev.totals()
    assert p == int(cd[:, 0].sum()) and a == int(cd[:, 1].sum())
    dev.close()


@pytest.mark.timeout(1800)
@pytest.mark.parametrize("workload,batch,steps", [("cfg2", 65536, 300), ("cfg3", 16384, 150), ("cfg1", 16384, 200),
                                                  # RMCSA: the 24-byte-entry sink and the serial release loop; Germany50: the
                                                  # global-state form (the per-GPU shard of BASELINE's 262 144 envs)
                                                  ("cfg4", 16384, 120), ("cfg5", 32768, 120)])
def test_full_size_batch_every_env_matches_oracle(workload, batch, steps):
    """The headline batch against the ORACLE on every env (not a sample, not another HIP form): the OpenMP build of the oracle
    steps all 65 536 cfg2 envs on the host's cores, and every env's counters, pending service, number of pending releases,
    whole slot map, link statistics and network statistics must equal the device's — so the branches that occur in one
    env-step in 10^4..10^7 (release overflow of a work item, list rebuilds, ties) are checked against the reference
    restatement wherever in the batch they happen."""
    import optical_rl_gym_amd as orl
    from bench import WORKLOADS
    from oracle.oracle import OracleBatch

    fam, topo, kw, policy = WORKLOADS[workload]
    kw = dict(kw, episode_length=100)
    seeds = [10 + i for i in range(batch)]
    dev = orl.make(fam, topology=topo, num_envs=batch, seeds=seeds, **kw)
    dev.run(policy, steps)
    ora = OracleBatch(fam, topo, seeds, omp=True, **kw)
    ora.run(policy, steps)
    chk = _exact(workload + " every env")
    chk(0, "counters", dev.counters(), ora.counters())
    chk(0, "services", dev.services(), ora.services())
    chk(0, "active", dev.active(), ora.active())
    chk(0, "slot maps", dev.slots_packed(), ora.slots_packed())
    chk(0, "link statistics", dev.link_stats_all(), ora.link_stats_all())
    chk(0, "network statistics", dev.net_stats_all(), ora.net_stats_all())
    if dev.obs_dim:
        chk(0, "observation", dev.observation(), ora.observation())
    assert not dev.flags().any()
    # the bulk read-backs are the per-env ones
    for i in (0, batch // 2 + 1, batch - 1):
        W = dev.lib.orl_batch_row_words(dev._h)
        rows = dev.slots(i).reshape(-1, kw.get("num_spectrum_resources", dev.num_spectrum_resources))
        packed = dev.slots_packed()[i][: rows.shape[0] * W].reshape(rows.shape[0], W)
        bits = ((packed[:, :, None] >> np.arange(64, dtype=np.uint64)[None, None, :]) & np.uint64(1)).reshape(rows.shape[0], -1)
        assert np.array_equal(bits[:, : rows.shape[1]].astype(np.uint8), rows)
        chk(i, "link_stats", dev.link_stats_all()[i], dev.link_stats(i))
    dev.close()


@pytest.mark.parametrize("workload,batch,steps,pol", [("cfg2", 65536, 700, None), ("cfg5", 32768, 300, None), ("cfg4", 16384, 300, None),
                                                      ("cfg1", 32768, 300, None), ("cfg3", 32768, 200, None),
                                                      ("cfg2", 16384, 300, "LLP_FF"), ("cfg2", 16384, 300, "SP_FF"),
                                                      ("cfg1", 16384, 300, "SAP_LF"), ("cfg1", 16384, 300, "LLP_FF")])
def test_split_pipeline_equals_wavefront_pipeline_on_every_env(workload, batch, steps, pol, monkeypatch):
    """Tens of millions of env-steps per case: the rare branches of the split pipeline (more releases in one step
    than a work item holds masks for, several rebuild rounds, the serial tail) occur a few hundred times, in envs no
    sample would pick.  The one-wavefront-per-env implementation is pinned to the oracle above; here every env of
    the two implementations must agree on every integer and every float64 of its record."""
    import optical_rl_gym_amd as orl
    from bench import WORKLOADS

    fam, topo, kw, policy = WORKLOADS[workload]
    policy = pol or policy
    kw = dict(kw, episode_length=90)
    seeds = [77 + 3 * i for i in range(batch)]
    out = {}
    for name in IMPLS:
        force_impl(monkeypatch, name)
        env = orl.make(fam, topology=topo, num_envs=batch, seeds=seeds, **kw)
        env.run(policy, steps // 2)  # two calls: the pipeline finishes its pending update between them
        env.run(policy, steps - steps // 2)
        if name == "persist_pair":
            assert _ran_pair_form(env) == (fam != "RMCSA")
        pick = [0, 1, batch // 3, batch - 1]
        if env.obs_dim:  # the observation the run left in the device buffer is the one a fresh evaluation gives
            in_loop = env.device_tensor("obs").cpu().numpy().copy()
            chk0 = _exact(workload)
            chk0(0, name + " obs left by run()", in_loop, env.observation())
        out[name] = dict(counters=env.counters().copy(), services=env.services().copy(), active=env.active().copy(),
                         flags=env.flags().copy(), slots=[env.slots(i).copy() for i in pick],
                         link=[env.link_stats(i).copy() for i in pick], net=[env.net_stats(i).copy() for i in pick],
                         serial=int(env.lib.orl_batch_debug_serial_count(env._h)))
        env.close()
    a = out["wave64"]
    chk = _exact(workload)
    for other in IMPLS[1:]:
        b = out[other]
        for key in ("counters", "services", "active", "flags"):
            chk(0, other + " " + key, b[key], a[key])
        for j in range(4):
            chk(j, other + " slots", b["slots"][j], a["slots"][j])
            chk(j, other + " link_stats", b["link"][j], a["link"][j])
            chk(j, other + " net_stats", b["net"][j], a["net"][j])


def test_split_pipeline_serial_tail_and_tally_pass(monkeypatch):
    """The item form holds 8 releases per link and step; with the limit forced down to 1 the tally pass and the
    serial tail of the release row kernel run thousands of times instead of once per 10^7 env-steps."""
    import optical_rl_gym_amd as orl
    from bench import WORKLOADS

    out = {}
    for workload, batch in (("cfg2", 4096), ("cfg4n", 1024)):
        fam, topo, kw, policy = WORKLOADS[workload]
        kw = dict(kw, episode_length=70)
        seeds = [5 + 11 * i for i in range(batch)]
        for name, v, masks in (("wave64", "wave64", None), ("two", "split2", "1"), ("two2", "split2", "2"),
                               ("persist", "persist", "1"), ("persist2", "persist", "2"), ("persist_g", "persist_global", "1"),
                               ("persist_l", "persist_lds", "1"), ("persist_rd", "persist_rd", "1"), ("persist_rd2", "persist_rd", "2")):
            force_impl(monkeypatch, v)
            if masks:
                monkeypatch.setenv("ORL_ITEM_MASKS", masks)
            else:
                monkeypatch.delenv("ORL_ITEM_MASKS", raising=False)
            env = orl.make(fam, topology=topo, num_envs=batch, seeds=seeds, **kw)
            env.run(policy, 500)
            out[name] = dict(counters=env.counters().copy(), services=env.services().copy(), active=env.active().copy(),
                             slots=env.slots(batch - 1).copy(), link=env.link_stats(batch - 1).copy(),
                             net=env.net_stats(batch - 1).copy(), serial=int(env.lib.orl_batch_debug_serial_count(env._h)))
            assert not env.flags().any()
            env.close()
        chk = _exact(workload)
        for name in ("two", "two2", "persist", "persist2", "persist_g", "persist_l", "persist_rd", "persist_rd2"):
            for key in ("counters", "services", "active", "slots", "link", "net"):
                chk(0, name + " " + key, out[name][key], out["wave64"][key])
        assert out["two"]["serial"] > 100 and out["two2"]["serial"] > 0 and out["persist"]["serial"] > 100


@pytest.mark.parametrize("workload", ["cfg3", "cfg1", "cfg5"])
def test_pair_form_early_exits_parked_and_ordered_services(workload, monkeypatch):
    """The two-wavefront form of small batches with the item form limited to one release per step: wavefronts leave their launches
    early thousands of times — the batch of services on order with the row wavefront is taken over at the exit and parked, groups
    whose batches ran out of phase ask on the spot — and every env must still equal the one-wavefront-per-env kernel (cfg2 is in
    the serial-tail test above; tools/stress_early_exit.py runs the same at 12 288 envs and with two releases per step)."""
    import optical_rl_gym_amd as orl
    from bench import WORKLOADS

    fam, topo, kw, policy = WORKLOADS[workload]
    kw = dict(kw, episode_length=70)
    batch = 4096
    seeds = [5 + 11 * i for i in range(batch)]
    out = {}
    for name, masks in (("wave64", None), ("persist", "1")):
        force_impl(monkeypatch, name)
        if masks:
            monkeypatch.setenv("ORL_ITEM_MASKS", masks)
            monkeypatch.setenv("ORL_JIT_SPEC", "1")
        else:
            monkeypatch.delenv("ORL_ITEM_MASKS", raising=False)
        env = orl.make(fam, topology=topo, num_envs=batch, seeds=seeds, **kw)
        for chunk in (130, 97, 173):
            env.run(policy, chunk)
        if name == "persist":
            assert _ran_pair_form(env) and int(env.lib.orl_batch_debug_serial_count(env._h)) > 10000
        out[name] = [env.counters().copy(), env.services().copy(), env.active().copy(), env.flags().copy(),
                     env.net_stats_all().copy(), env.link_stats_all().copy(), env.slots_packed().copy()]
        env.close()
    chk = _exact(workload + " pair form, early exits")
    for k, (x, y) in enumerate(zip(out["persist"], out["wave64"])):
        chk(k, "item", x, y)


def test_pair_form_hand_over_under_jitter(monkeypatch):
    """The pair form's hand-over (LDS counters, workgroup-scope release / acquire on the local address space) with both wavefronts
    sleeping a pseudo-random time before every signal and after every wait (-DORL_DIAG -DORL_X_JITTER: a library and a specialisation
    of their own), one release per step allowed so that wavefronts also leave early: every env against the one-wavefront-per-env
    kernel of the product library."""
    import optical_rl_gym_amd as orl
    from bench import WORKLOADS

    fam, topo, kw, policy = WORKLOADS["cfg2"]
    kw = dict(kw, episode_length=70)
    batch = 4096
    seeds = [9 + 5 * i for i in range(batch)]
    out = {}
    for name in ("wave64", "persist"):
        force_impl(monkeypatch, name)
        if name == "persist":
            monkeypatch.setenv("ORL_ITEM_MASKS", "2")
            monkeypatch.setenv("ORL_JIT_SPEC", "1")
            monkeypatch.setenv("ORL_HIPCC_EXTRA", "-DORL_DIAG -DORL_X_JITTER")
            monkeypatch.setenv("ORL_LIB_VARIANT", "exp")
        else:
            monkeypatch.delenv("ORL_ITEM_MASKS", raising=False)
        env = orl.make(fam, topology=topo, num_envs=batch, seeds=seeds, **kw)
        for chunk in (140, 61):
            env.run(policy, chunk)
        if name == "persist":
            assert _ran_pair_form(env)
        out[name] = [env.counters().copy(), env.services().copy(), env.active().copy(), env.flags().copy(),
                     env.net_stats_all().copy(), env.link_stats_all().copy(), env.slots_packed().copy()]
        env.close()
    chk = _exact("cfg2 pair form under jitter")
    for k, (x, y) in enumerate(zip(out["persist"], out["wave64"])):
        chk(k, "item", x, y)


@pytest.mark.parametrize("workload,batch", [("cfg2", 4096), ("cfg2", 20000), ("cfg3", 4096), ("cfg5", 2048)])
def test_short_statistics_log_and_early_exits(workload, batch, monkeypatch):
    """The statistics log of a launch shortened to 12 steps (ORL_LOG_CAP; launches are then 6 steps long) and the item form limited
    to one release per step: wavefronts that left early catch up over more steps than a launch can log, stop at the end of the log
    and count as unfinished — with the batch of services the row wavefront of a pair has on order, and with services parked, at
    every kind of exit.  Every env against the one-wavefront-per-env kernel (4 096 / 2 048 envs: the two-wavefront form; 20 000:
    one wavefront per 8 envs)."""
    import optical_rl_gym_amd as orl
    from bench import WORKLOADS

    fam, topo, kw, policy = WORKLOADS[workload]
    kw = dict(kw, episode_length=70)
    seeds = [5 + 11 * i for i in range(batch)]
    out = {}
    # (persist_rd: the rows-deferred form with the same short statistics log; persist_rd_ev: its EVENT log shortened to 40 events per
    # env and launch instead, so that wavefronts stop because that one is full — Germany50 has more links than an event's link bits:
    # the library's own form runs there)
    for name in ("wave64", "persist", "persist_rd", "persist_rd_ev"):
        force_impl(monkeypatch, "persist_rd" if name.startswith("persist_rd") else name)
        monkeypatch.delenv("ORL_ELOG_CAP", raising=False)
        if name != "wave64":
            monkeypatch.setenv("ORL_ITEM_MASKS", "1")
            monkeypatch.setenv("ORL_LOG_CAP", "12")
            monkeypatch.setenv("ORL_JIT_SPEC", "1")
            if name == "persist_rd_ev":
                monkeypatch.setenv("ORL_LOG_CAP", "64")
                monkeypatch.setenv("ORL_ELOG_CAP", "40")
        else:
            monkeypatch.delenv("ORL_ITEM_MASKS", raising=False)
            monkeypatch.delenv("ORL_LOG_CAP", raising=False)
        env = orl.make(fam, topology=topo, num_envs=batch, seeds=seeds, **kw)
        for chunk in (83, 7, 110):
            env.run(policy, chunk)
        if name == "persist":
            assert _ran_pair_form(env) == (batch <= 12288)
        if name.startswith("persist_rd"):
            assert int(env.lib.orl_batch_debug_persist_form(env._h)) == (7 if workload != "cfg5" else 4 if batch <= 12288 else 0)
        out[name] = [env.counters().copy(), env.services().copy(), env.active().copy(), env.flags().copy(),
                     env.net_stats_all().copy(), env.link_stats_all().copy(), env.slots_packed().copy()]
        env.close()
    for name in ("persist", "persist_rd", "persist_rd_ev"):
        chk = _exact("%s %d, short log, %s" % (workload, batch, name))
        for k, (x, y) in enumerate(zip(out[name], out["wave64"])):
            chk(k, "item", x, y)


@pytest.mark.parametrize("workload,batch", [("cfg2", 16384), ("cfg4", 2048), ("cfg5", 4096)])
def test_issue_priority_rotation_changes_no_result(workload, batch, monkeypatch):
    """Round 6: the wavefronts of the persistent kernel rotate their issue priority (`s_setprio` by the constant clock) so that
    the four wavefronts of a SIMD finish together.  It orders instructions, nothing else: the arbiter's own order (ORL_PERSIST_FAIR=0), the
    default period and a fast one leave every env in the same state."""
    import optical_rl_gym_amd as orl
    from bench import WORKLOADS

    fam, topo, kw, policy = WORKLOADS[workload]
    kw = dict(kw, episode_length=45)
    seeds = [5 + 11 * i for i in range(batch)]
    force_impl(monkeypatch, "persist")
    out = {}
    for fair in ("0", "11", "6"):
        monkeypatch.setenv("ORL_PERSIST_FAIR", fair)
        env = orl.make(fam, topology=topo, num_envs=batch, seeds=seeds, **kw)
        env.run(policy, 130)
        env.run(policy, 57)
        out[fair] = dict(counters=env.counters().copy(), services=env.services().copy(), active=env.active().copy(), flags=env.flags().copy(),
                         link=env.link_stats_all().copy(), net=[env.net_stats(i).copy() for i in (0, batch // 2, batch - 1)],
                         slots=env.slots_packed().copy())
        env.close()
    chk = _exact(workload)
    for fair in ("11", "6"):
        for key in ("counters", "services", "active", "flags", "link", "slots"):
            chk(0, "fair=%s %s" % (fair, key), out[fair][key], out["0"][key])
        for j in range(3):
            chk(j, "fair=%s net_stats" % fair, out[fair]["net"][j], out["0"]["net"][j])


def test_specialised_instantiations_are_used_and_equal_the_generic_kernel(monkeypatch):
    """Any configuration gets the persistent kernel with ITS sizes as compile-time constants: a small library built on first
    use from the flags the main library writes for the batch (orl_batch_spec_flags -> _build.build_spec -> orl_batch_load_spec),
    cached, attached after a field-by-field comparison.  BASELINE cfg1 - cfg5, RMSAEnv's default 100-slot spectrum and a
    configuration nobody prepared (300 slots, k = 5, another bit-rate range): the launcher uses the attached instantiation
    (debug query), it leaves the state of the generic kernel (ORL_PERSIST_SPEC=0 forces that one), small batches do not
    trigger a build unless asked to, and a library built for another configuration is refused."""
    import optical_rl_gym_amd as orl
    from bench import WORKLOADS
    from optical_rl_gym_amd import _build

    force_impl(monkeypatch, "persist")
    monkeypatch.setenv("ORL_JIT_SPEC", "1")
    paths = {}
    for workload in ("cfg2", "cfg3", "cfg1", "cfg4", "cfg5", "rmsa100", "odd300"):
        fam, topo, kw, policy = WORKLOADS["cfg2" if workload in ("rmsa100", "odd300") else workload]
        if workload == "rmsa100":  # RMSAEnv's default spectrum
            kw = dict(kw, num_spectrum_resources=100, load=120)
        if workload == "odd300":
            kw = dict(kw, num_spectrum_resources=300, load=250, bit_rate_lower_bound=40, bit_rate_higher_bound=90)
        kw = dict(kw, episode_length=45)
        seeds = [31 + 2 * i for i in range(1024 if workload in ("cfg4", "cfg5") else 4096)]
        out = {}
        for name, spec_env in (("spec", None), ("generic", "0")):
            if spec_env is None:
                monkeypatch.delenv("ORL_PERSIST_SPEC", raising=False)
            else:
                monkeypatch.setenv("ORL_PERSIST_SPEC", spec_env)
            env = orl.make(fam, topology=topo, num_envs=len(seeds), seeds=seeds, **kw)
            assert env.specialised
            env.run(policy, 130)
            env.run(policy, 70)
            # (1: the specialised kernel, 2: its two-wavefront form — batches of at most 12 288 envs of the single-core families)
            pair = fam != "RMCSA"
            assert int(env.lib.orl_batch_debug_persist_spec(env._h)) == ((2 if pair else 1) if spec_env is None else 0)
            out[name] = [env.counters().copy(), env.services().copy(), env.active().copy(), env.slots(9).copy(),
                         env.link_stats(9).copy(), env.net_stats(9).copy()]
            if env.obs_dim:
                out[name].append(env.device_tensor("obs").cpu().numpy().copy())
            buf = __import__("ctypes").create_string_buffer(1024)
            assert env.lib.orl_batch_spec_flags(env._h, buf, 1024) > 0
            paths[workload] = _build.spec_path(buf.value.decode())
            env.close()
        chk = _exact(workload + " specialised")
        for k, (x, y) in enumerate(zip(out["spec"], out["generic"])):
            chk(k, "item", x, y)
    assert len(set(paths.values())) == len(paths)
    # a small batch of a configuration without a cached library runs the generic kernel (no 15-s build behind a unit test) ...
    monkeypatch.delenv("ORL_JIT_SPEC", raising=False)
    monkeypatch.delenv("ORL_PERSIST_SPEC", raising=False)
    fam, topo, kw, policy = WORKLOADS["cfg2"]
    env = orl.make(fam, topology=topo, num_envs=64, seeds=list(range(64)), **dict(kw, num_spectrum_resources=290))
    env.run(policy, 10)
    assert not env.specialised and int(env.lib.orl_batch_debug_persist_spec(env._h)) == 0
    # ... and refuses a library built for another configuration
    rc = env.lib.orl_batch_load_spec(env._h, paths["cfg2"].encode())
    assert rc == -1 and b"another configuration" in env.lib.orl_last_error()
    env.run(policy, 10)
    assert int(env.lib.orl_batch_debug_persist_spec(env._h)) == 0
    env.close()


def test_step_counters_run_on_between_runs_and_wrap(monkeypatch):
    """The persistent kernel's per-workgroup step counters are not cleared between runs (no fill kernel in front of a run):
    they count from a base that is reset only when it would overflow.  With the limit forced down to 100 steps, runs of
    assorted lengths cross it several times and the state equals the per-env kernel's after every run."""
    import optical_rl_gym_amd as orl
    from bench import WORKLOADS

    fam, topo, kw, policy = WORKLOADS["cfg2"]
    kw = dict(kw, episode_length=60)
    seeds = [11 + 5 * i for i in range(2048)]
    lengths = (7, 64, 65, 1, 30, 130, 3, 99, 100, 2)
    out = {}
    for name, v, limit in (("wave64", "wave64", None), ("persist", "persist", "100"), ("persist_plain", "persist", None)):
        force_impl(monkeypatch, v)
        if limit:
            monkeypatch.setenv("ORL_RUN_BASE_LIMIT", limit)
        else:
            monkeypatch.delenv("ORL_RUN_BASE_LIMIT", raising=False)
        env = orl.make(fam, topology=topo, num_envs=len(seeds), seeds=seeds, **kw)
        rec = []
        for n in lengths:
            env.run(policy, n)
            rec.append((env.counters().copy(), env.services().copy(), env.active().copy(), env.slots(77).copy(),
                        env.link_stats(77).copy(), env.net_stats(77).copy()))
        out[name] = rec
        assert not env.flags().any()
        env.close()
    chk = _exact("step counters")
    for name in ("persist", "persist_plain"):
        for t, (a, b) in enumerate(zip(out[name], out["wave64"])):
            for k, (x, y) in enumerate(zip(a, b)):
                chk(t, "%s item %d" % (name, k), x, y)


def test_pending_releases_at_the_end_of_a_run(monkeypatch):
    """A wavefront whose env could not put its releases into items leaves its loop, counts as unfinished, and releases them in
    place at the start of its next launch — also when that happens on the LAST step of a run (the host relaunches until no
    wavefront is left).  With one mask per item that is the common case: DeepRMSA runs of 1, 2, 3, 17, 64 and 65 steps leave
    the state, the observation of the pending service (refreshed after those releases) and the rewards of the per-env
    kernel, and the first envs equal the oracle."""
    import optical_rl_gym_amd as orl
    from oracle.oracle import OracleBatch

    kw = dict(mean_service_holding_time=7.5, mean_service_inter_arrival_time=1.0 / 12.0, j=1, episode_length=50,
              num_spectrum_resources=100)
    seeds = [17 + 3 * i for i in range(1024)]
    lengths = (1, 2, 3, 17, 64, 65, 1)
    out = {}
    for name, v, masks in (("wave64", "wave64", None), ("persist", "persist", "1"), ("persist_l", "persist_lds", "1"),
                           ("persist_g", "persist_global", "1")):
        force_impl(monkeypatch, v)
        if masks:
            monkeypatch.setenv("ORL_ITEM_MASKS", masks)
        else:
            monkeypatch.delenv("ORL_ITEM_MASKS", raising=False)
        env = orl.make("DeepRMSA", topology="nsfnet_chen", num_envs=len(seeds), seeds=seeds, **kw)
        env.run("SAP", 400)
        rec = []
        for n in lengths:
            env.run("SAP", n)
            stored = env.device_tensor("obs").cpu().numpy().copy()  # what the run's last step (or the release path) wrote
            chk0 = _exact(name + " stored observation")
            chk0(n, "obs", stored, env.observation())  # ... equals the observation computed from the state now
            rec.append((stored, env.counters().copy(), env.services().copy(), env.active().copy(),
                        env.slots(5).copy(), env.link_stats(5).copy(), env.net_stats(5).copy()))
        out[name] = rec
        if masks:
            assert int(env.lib.orl_batch_debug_serial_count(env._h)) > 100
        assert not env.flags().any()
        env.close()
    chk = _exact("pending releases")
    for name in ("persist", "persist_l", "persist_g"):
        for t, (a, b) in enumerate(zip(out[name], out["wave64"])):
            for k, (x, y) in enumerate(zip(a, b)):
                chk(t, "%s item %d" % (name, k), x, y)
    ora = OracleBatch("DeepRMSA", "nsfnet_chen", seeds[:4], **kw)
    ora.run("SAP", 400 + sum(lengths))
    chk(0, "oracle observation", out["persist"][-1][0][:4], ora.observation())
    chk(0, "oracle counters", out["persist"][-1][1][:4], ora.counters())


CORNERS = [
    ("RWA", "cost239", "SAP_LF", 3000, dict(load=450.0, mean_service_holding_time=25.0, episode_length=7, num_spectrum_resources=17,
                                            allow_rejection=True)),
    ("RMSA", "nsfnet_chen", "LLP_FF", 1500, dict(load=700.0, mean_service_holding_time=25.0, episode_length=40,
                                                 num_spectrum_resources=500, allow_rejection=True, bit_rate_selection="discrete")),
    ("RMSA", "cost239", "SP_FF", 2048, dict(load=60.0, mean_service_holding_time=5.0, episode_length=1000, num_spectrum_resources=64)),
    ("DeepRMSA", "germany50", "SAP", 1024, dict(mean_service_holding_time=7.5, mean_service_inter_arrival_time=0.25, j=4,
                                                episode_length=9, num_spectrum_resources=200)),
    ("RMCSA", "cost239", "SAP_BM_FC_FF", 700, dict(load=300.0, mean_service_holding_time=5.0, episode_length=40,
                                                   num_spectrum_resources=100, num_spatial_resources=7, allow_rejection=True)),
]


@pytest.mark.parametrize("fam,topo,policy,batch,kw", CORNERS)
def test_corner_configurations_agree_across_step_implementations(fam, topo, policy, batch, kw, monkeypatch):
    """Row widths of 1, 2, 4 and 8 words, slot counts that are no multiple of 64, discrete bit rates, rejection, very short
    episodes, 7 cores, the 88-link topology: every env ends in the same state under all three default-path implementations
    (a sample of tools/fuzz_cross.py, which drew 62 random configurations without a mismatch)."""
    import optical_rl_gym_amd as orl

    seeds = [int(x) for x in np.random.RandomState(len(topo) + batch).randint(0, 2**31 - 1, batch)]
    out = {}
    for v in IMPLS:
        force_impl(monkeypatch, v)
        env = orl.make(fam, topology=topo, num_envs=batch, seeds=seeds, **kw)
        env.run(policy, 130)
        env.run(policy, 170)
        if v == "persist_pair":
            assert _ran_pair_form(env) == (fam != "RMCSA")
        pick = (0, batch // 2, batch - 1)
        out[v] = [env.counters().copy(), env.services().copy(), env.active().copy(), env.flags().copy()] + \
                 [env.slots(i).copy() for i in pick] + [env.link_stats(i).copy() for i in pick] + [env.net_stats(i).copy() for i in pick]
        # one host step on top: its info holds what no read-back shows — the per-rate request / provision histograms of the
        # discrete mode, RWA's action marginals — as the device loop (and the replay of its statistics log) left them
        _, r_h, d_h, i_h = env.step(env.policy(policy).copy(), auto_reset=True)
        out[v] += [np.array(r_h), np.array(d_h), np.array(i_h)]
        env.close()
    chk = _exact(fam + "/" + topo)
    for v in IMPLS[1:]:
        for k, (x, y) in enumerate(zip(out[v], out["wave64"])):
            chk(k, "impl " + v, x, y)
    # ... and that state is the reference's: the first envs against the oracle
    from oracle.oracle import OracleBatch

    ora = OracleBatch(fam, topo, seeds[:5], **kw)
    ora.run(policy, 300)
    ref = out["wave64"]
    chk(0, "oracle counters", ref[0][:5], ora.counters())
    chk(0, "oracle services", ref[1][:5], ora.services())
    chk(0, "oracle slots", ref[4], ora.slots(0))
    chk(0, "oracle link_stats", ref[7], ora.link_stats(0))
    chk(0, "oracle net_stats", ref[10], ora.net_stats(0))


def test_services_wider_than_the_row_items_use_the_per_env_kernel():
    """The row items of the persistent kernel carry a service as (first slot: 9 bits | slots: 6 bits); the library accepts
    services of up to 64 slots: a configuration whose largest service needs exactly 64 runs on the per-env kernel
    (orl_api.hip, pipeline_ok) — with the reference's results, checked against the oracle."""
    import optical_rl_gym_amd as orl
    from oracle.oracle import OracleBatch

    kw = dict(load=60.0, mean_service_holding_time=10.0, episode_length=40, num_spectrum_resources=320,
              bit_rate_selection="discrete", bit_rates=(100, 780), bit_rate_probabilities=(0.7, 0.3))  # 780 Gb/s on BPSK: 63 + 1 slots
    seeds = [3 + 7 * i for i in range(64)]
    env = orl.make("RMSA", topology="nsfnet_chen", num_envs=64, seeds=seeds, **kw)
    st = env.run("SAP_FF", 300)
    assert "k_persist" not in [n for n, _ in st.kernels()]
    ora = OracleBatch("RMSA", "nsfnet_chen", seeds[:6], **kw)
    ora.run("SAP_FF", 300)
    chk = _exact("wide services")
    chk(0, "counters", env.counters()[:6], ora.counters())
    chk(0, "services", env.services()[:6], ora.services())
    for i in range(6):
        chk(i, "slots", env.slots(i), ora.slots(i))
        chk(i, "link_stats", env.link_stats(i), ora.link_stats(i))
        chk(i, "net_stats", env.net_stats(i), ora.net_stats(i))
    assert env.counters()[:, 1].sum() > 0 and not env.flags().any()
    env.close()


def test_run_reports_every_kernel_of_the_step(monkeypatch):
    import optical_rl_gym_amd as orl
    from bench import WORKLOADS

    fam, topo, kw, policy = WORKLOADS["cfg2"]
    for v, names in (("wave64", ["k_step"]), ("split2", ["k_step_a2", "k_rows2", "k_rel_tail"])):
        force_impl(monkeypatch, v)
        env = orl.make(fam, topology=topo, num_envs=2048, seeds=list(range(2048)), **kw)
        st = env.run(policy, 20, time_kernels=1)
        assert [n for n, _ in st.kernels()] == names
        assert all(ms > 0 for _, ms in st.kernels()) and st.launches == 20 * len(names)
        st2 = env.run(policy, 20, time_kernels=2)
        assert st2.ms_policy > 0 and st2.ms_step > 0
        env.close()
    force_impl(monkeypatch, "persist")  # the default: the production run is launches of the persistent kernel
    env = orl.make(fam, topology=topo, num_envs=2048, seeds=list(range(2048)), **kw)
    st = env.run(policy, 20)
    assert [n for n, _ in st.kernels()] == ["k_persist"] and st.kernels()[0][1] > 0
    env.close()


def test_zero_copy_device_tensors_drive_the_batch():
    """An agent on the same GPU: actions written into the batch's device array through torch, step without copies,
    reward / done / info read in place — same numbers as the host-driven path."""
    import torch

    meta = load_golden("g2_rmsa_cfg2_sapff")["meta"]
    kw = dict(meta["kwargs"])
    kw.pop("seed")
    kw["episode_length"] = 40
    seeds = list(range(900, 900 + 96))
    a = _product(dict(meta, kwargs=dict(kw, seed=0)), num_envs=96, seeds=seeds)
    b = _product(dict(meta, kwargs=dict(kw, seed=0)), num_envs=96, seeds=seeds)
    act, rew, done, info = (a.device_tensor(n) for n in ("actions", "reward", "done", "info"))
    assert act.is_cuda and act.shape == (96, 4) and act.dtype == torch.int32 and info.shape == (96, a.n_info)
    for t in range(120):
        acts = b.policy("SAP_FF").copy()
        act.copy_(torch.from_numpy(acts))
        torch.cuda.synchronize()
        a.step(None, auto_reset=True, fetch=False)
        a.sync()
        _, r, d, i = b.step(acts, auto_reset=True)
        assert np.array_equal(rew.cpu().numpy(), r) and np.array_equal(done.cpu().numpy(), d), t
        assert np.array_equal(info.cpu().numpy(), i), t
    a.close()
    b.close()


@pytest.mark.parametrize("B", [4096, 1024])
def test_agent_rollout_in_a_captured_graph_equals_the_eager_loop(B):
    """What the reference's examples/stable_baselines3/DeepRMSA.ipynb:272-302 does through SB3, on the GPU: T = 32 steps of policy
    network forward + sampling + `step(None, fetch=False)` captured ONCE in a torch.cuda.CUDAGraph on the batch's own stream
    (orl_batch_stream: the step call must neither synchronise nor allocate) and replayed, against the same loop queued eagerly:
    the whole batch state — counters, pending services, slot maps, link and network statistics, observation, reward — bit for bit.
    (4 096 envs: k_agent; 1 024: k_step.  The samples come from Gumbel noise drawn beforehand, indexed by a device-side step
    counter, so that both loops consume the same random numbers.)"""
    import torch

    import optical_rl_gym_amd as orl

    T, REPLAYS, WARM = 32, 2, 3
    kw = dict(topology="nsfnet_chen", mean_service_holding_time=7.5, mean_service_inter_arrival_time=1 / 12.0, j=1,
              episode_length=50, num_spectrum_resources=100)
    out = {}
    for mode in ("eager", "graph"):
        env = orl.make("DeepRMSA-v0", num_envs=B, seeds=[3 + 7 * i for i in range(B)], **kw)
        dev = "cuda:%d" % env.device_id
        obs, rew, act = (env.device_tensor(n) for n in ("obs", "reward", "actions"))
        n_actions = env.k_paths * env.j + (1 if env.allow_rejection else 0)
        gen = torch.Generator(device=dev)
        gen.manual_seed(1234)
        net = torch.nn.Sequential(torch.nn.Linear(env.obs_dim, 64), torch.nn.ELU(), torch.nn.Linear(64, n_actions)).to(dev)
        with torch.no_grad():
            for k, prm in enumerate(net.parameters()):
                prm.copy_(torch.randn(prm.shape, generator=gen, device=dev) * 0.3)
        n_steps = WARM + T * REPLAYS
        u = torch.rand((n_steps, B, n_actions), generator=gen, device=dev).clamp_(1e-6, 1 - 1e-6)
        gumbel = -torch.log(-torch.log(u))
        tcount = torch.zeros(1, dtype=torch.long, device=dev)
        env.reset()
        env.observation()
        stream = env.torch_stream()

        def rollout_step():
            with torch.no_grad():
                logits = net(obs.float())
                a = (logits + gumbel.index_select(0, tcount)[0]).argmax(dim=1)
                act[:, 0] = a.int()
                env.step(None, auto_reset=True, fetch=False)
                tcount.add_(1)

        torch.cuda.synchronize()
        with torch.cuda.stream(stream):
            for _ in range(WARM):
                rollout_step()
            if mode == "eager":
                for _ in range(T * REPLAYS):
                    rollout_step()
        torch.cuda.synchronize()
        if mode == "graph":
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g, stream=stream):
                for _ in range(T):
                    rollout_step()
            with torch.cuda.stream(stream):
                for _ in range(REPLAYS):
                    g.replay()
            torch.cuda.synchronize()
        env.check()
        assert int(tcount.item()) == n_steps
        out[mode] = [env.counters().copy(), env.services().copy(), env.active().copy(), env.flags().copy(), env.net_stats_all().copy(),
                     env.link_stats_all().copy(), env.slots_packed().copy(), obs.cpu().numpy().copy(), rew.cpu().numpy().copy()]
        # (a device-resident run on top: the persistent kernel must not trust row caches from before the replays)
        env.run("SAP", 40)
        out[mode] += [env.counters().copy(), env.link_stats_all().copy(), env.slots_packed().copy()]
        env.close()
    assert out["eager"][0][:, 0].min() > 0  # (services were processed)
    chk = _exact("captured rollout, %d envs" % B)
    for k, (x, y) in enumerate(zip(out["graph"], out["eager"])):
        chk(k, "item", x, y)


@pytest.mark.parametrize("workload,B", [("cfg2", 65536), ("cfg3", 65536), ("cfg1", 32768), ("cfg4", 16384)])
def test_agent_in_the_loop_at_full_size(workload, B):
    """The path an RL agent on the same GPU drives (SB3 VecEnv semantics: actions in device memory, auto reset, nothing
    fetched): 65 536 envs stepped through k_agent with actions from the on-device heuristic, against the oracle on sampled
    envs — reward, done, all info floats and the DeepRMSA observation of the last steps, then the whole state."""
    import optical_rl_gym_amd as orl
    from bench import WORKLOADS
    from oracle.oracle import OracleBatch

    fam, topo, kw, policy = WORKLOADS[workload]
    kw = dict(kw, episode_length=40)
    steps = 90
    seeds = [10 + i for i in range(B)]
    dev = orl.make(fam, topology=topo, num_envs=B, seeds=seeds, **kw)
    assert int(dev.lib.orl_batch_debug_step_kernel(dev._h)) == 2  # every family's host / agent steps go through k_agent at this size
    sample = sorted(set([0, 7, 8, 4095, B // 2, B - 1] + list(np.random.RandomState(9).randint(0, B, 10))))
    ora = OracleBatch(fam, topo, [seeds[i] for i in sample], **kw)
    rew, done, info = (dev.device_tensor(n) for n in ("reward", "done", "info"))
    obs = dev.device_tensor("obs") if dev.obs_dim else None
    chk = _exact(workload + " agent loop")
    for t in range(steps):
        dev.policy(policy, fetch=False)
        dev.step(None, auto_reset=True, fetch=False)
        _, r, d, i = ora.step(ora.policy(policy), auto_reset=True)
        if t % 15 == 14 or t >= steps - 3:
            dev.sync()
            chk(t, "reward", rew.cpu().numpy()[sample], r)
            chk(t, "done", done.cpu().numpy()[sample], d)
            chk(t, "info", info.cpu().numpy()[sample], i)
            if obs is not None:
                chk(t, "obs", obs.cpu().numpy()[sample], ora.observation())
    dev.check()
    chk(steps, "counters", dev.counters()[sample], ora.counters())
    chk(steps, "services", dev.services()[sample], ora.services())
    for j, e in enumerate(sample):
        chk(e, "slots", dev.slots(e), ora.slots(j))
        chk(e, "link_stats", dev.link_stats(e), ora.link_stats(j))
        chk(e, "net_stats", dev.net_stats(e), ora.net_stats(j))
    # the device-resident loop continues from that state, and host steps after it
    dev.run(policy, 70)
    ora.run(policy, 70)
    dev.policy(policy, fetch=False)
    dev.step(None, auto_reset=True, fetch=False)
    _, r, d, i = ora.step(ora.policy(policy), auto_reset=True)
    dev.sync()
    chk(-1, "info after run", info.cpu().numpy()[sample], i)
    chk(-1, "counters after run", dev.counters()[sample], ora.counters())
    assert not dev.flags().any()
    dev.close()


def test_terminal_observation_on_device(impl_host):
    """DeepRMSA: the env that reports done gets its observation also as terminal_observation (device array)."""
    meta = load_golden("g4_deeprmsa_j2_sap")["meta"]
    kw = dict(meta["kwargs"])
    kw.pop("seed")
    kw["episode_length"] = 12
    env = _product(dict(meta, kwargs=dict(kw, seed=0)), num_envs=64, seeds=list(range(64)))
    obs, tobs, done = env.device_tensor("obs"), env.device_tensor("terminal_obs"), env.device_tensor("done")
    seen = 0
    for t in range(40):
        env.policy("SAP", fetch=False)
        env.step(None, auto_reset=True, fetch=False)
        env.sync()
        d = done.cpu().numpy().astype(bool)
        if d.any():
            seen += int(d.sum())
            assert np.array_equal(tobs.cpu().numpy()[d], obs.cpu().numpy()[d])
            assert np.array_equal(obs.cpu().numpy(), env.observation())
    assert seen >= 64
    env.close()


def test_device_seeding_equals_cpython():
    """random.Random(seed) expanded on the device (init_by_array) vs CPython's getstate(), incl. negative and >32-bit seeds."""
    import optical_rl_gym_amd as orl
    from oracle.oracle import OracleBatch

    seeds = [0, 1, 10, 41, 2**31 - 1, 2**32, 2**32 + 5, 2**40 + 7, -3, 123456789, 2**62 + 11, 7]
    kw = dict(load=300, mean_service_holding_time=25, episode_length=50, num_spectrum_resources=320)
    dev = orl.BatchedRMSAEnv("nsfnet_chen", num_envs=len(seeds), seeds=seeds, **kw)
    ora = OracleBatch("RMSA", "nsfnet_chen", seeds, **kw)  # CPython expands the seeds here
    chk = _exact("seeding")
    chk(0, "first service", dev.services(), ora.services())
    for t in range(30):
        a = ora.policy("SAP_FF")
        _, r_o, _, _ = ora.step(a, auto_reset=True)
        _, r_d, _, _ = dev.step(a, auto_reset=True)
        chk(t, "reward", r_d, r_o)
    chk(30, "services", dev.services(), ora.services())
    dev.close()


def test_matrix_observation_and_snapshot_restore():
    import optical_rl_gym_amd as orl

    kw = dict(load=200, mean_service_holding_time=25, episode_length=40, num_spectrum_resources=100)
    dev = orl.BatchedRMSAEnv("nsfnet_chen", num_envs=40, seeds=list(range(40)), **kw)
    dev.run("SAP_FF", 150)
    mat = dev.matrix_observation()
    svc = dev.services()
    for e in (0, 17, 39):
        src, dst = int(svc[e, 2]), int(svc[e, 3])
        tau = np.zeros(28, np.uint8)
        tau[min(src, dst)] = 1
        tau[14 + max(src, dst)] = 1
        assert np.array_equal(mat[e, :28], tau)
        assert np.array_equal(mat[e, 28:], dev.slots(e).reshape(-1))
    snap = dev.get_state()
    dev.run("SAP_FF", 60)
    after_a = (dev.counters().copy(), dev.services().copy(), dev.slots(3).copy(), dev.link_stats(3).copy())
    dev.set_state(snap)
    assert np.array_equal(dev.services(), svc)
    dev.run("SAP_FF", 60)
    after_b = (dev.counters(), dev.services(), dev.slots(3), dev.link_stats(3))
    for x, y in zip(after_a, after_b):
        assert np.array_equal(x, y)
    dev.close()


@pytest.mark.parametrize("gname,policy", [("g2_rmsa_cfg2_sapff", "SAP_FF"), ("g4_deeprmsa_j2_sap", "SAP"),
                                          ("g6_rmcsa_7x320_sapff", "SAP_BM_FC_FF"), ("g5_rwa_testcfg_sapff", "SAP_FF")])
def test_device_runs_and_host_steps_alternate_on_one_batch(gname, policy):
    """Default configuration: run() goes through the persistent kernel, host-driven step() of a small batch through the
    one-wavefront-per-env kernel; they alternate on the same state (each leaves what the other needs, caches marked
    unknown) and the result is the oracle's."""
    from oracle.oracle import OracleBatch

    meta = load_golden(gname)["meta"]
    kw = dict(meta["kwargs"])
    kw.pop("seed")
    kw["episode_length"] = 45
    seeds = [3000 + 5 * i for i in range(72)]
    ora = OracleBatch(meta["env"], meta["topology"], seeds, **kw)
    dev = _product(dict(meta, kwargs=dict(kw, seed=0)), num_envs=72, seeds=seeds)
    chk = _exact(gname)
    for rnd in range(4):
        dev.run(policy, 37)
        ora.run(policy, 37)
        for t in range(23):
            a_o, a_d = ora.policy(policy), dev.policy(policy)
            chk(t, "actions", a_d, a_o)
            obs_o, r_o, d_o, i_o = ora.step(a_o, auto_reset=True)
            obs_d, r_d, d_d, i_d = dev.step(a_d, auto_reset=True)
            chk(t, "reward", r_d, r_o)
            chk(t, "done", d_d, d_o)
            chk(t, "info", i_d, i_o)
            if obs_o is not None:
                chk(t, "obs", obs_d, obs_o)
        chk(rnd, "counters", dev.counters(), ora.counters())
        chk(rnd, "services", dev.services(), ora.services())
        for e in (0, 35, 71):
            chk(rnd, "slots", dev.slots(e), ora.slots(e))
            chk(rnd, "link_stats", dev.link_stats(e), ora.link_stats(e))
            chk(rnd, "net_stats", dev.net_stats(e), ora.net_stats(e))
    assert not dev.flags().any()
    dev.close()


def test_snapshot_moves_between_step_implementations(monkeypatch):
    """A snapshot taken from a batch driven by the two-kernel pipeline continues bit-identically in a batch driven by the
    one-wavefront-per-env kernel, and the other way round."""
    import optical_rl_gym_amd as orl

    kw = dict(load=300, mean_service_holding_time=25, episode_length=50, num_spectrum_resources=320)
    seeds = list(range(200, 296))
    envs = {}
    for name, v in (("wave64", "wave64"), ("two", "persist")):
        force_impl(monkeypatch, v)
        envs[name] = orl.BatchedRMSAEnv("nsfnet_chen", num_envs=96, seeds=seeds, **kw)
    envs["two"].run("SAP_FF", 300)
    envs["wave64"].set_state(envs["two"].get_state())
    envs["two"].run("SAP_FF", 200)
    envs["wave64"].run("SAP_FF", 200)
    chk = _exact("snapshot")
    for rnd in range(2):
        a, b = envs["two"], envs["wave64"]
        chk(rnd, "counters", a.counters(), b.counters())
        chk(rnd, "services", a.services(), b.services())
        for e in (0, 50, 95):
            chk(rnd, "slots", a.slots(e), b.slots(e))
            chk(rnd, "link_stats", a.link_stats(e), b.link_stats(e))
            chk(rnd, "net_stats", a.net_stats(e), b.net_stats(e))
        if rnd == 0:  # and back
            envs["two"].set_state(envs["wave64"].get_state())
            envs["two"].run("SAP_FF", 150)
            envs["wave64"].run("SAP_FF", 150)
    for e in envs.values():
        e.close()


def test_gym_front_end_rwa_and_rmcsa_script_numbers():
    """tests/test_rwa.py and tests/test_rmcsa.py of the reference through the product's gym-shaped classes (episode
    rewards as captured from the reference in tests/golden)."""
    import optical_rl_gym_amd as orl

    g = load_golden("g5_rwa_testcfg_sapff")
    kw = dict(g["meta"]["kwargs"])
    kw.pop("seed")
    env = orl.RWAEnv(topology="nsfnet_chen", seed=10, **kw)
    rewards, lengths = orl.evaluate_heuristic(env, orl.shortest_available_path_first_fit, n_eval_episodes=3,
                                              return_episode_rewards=True)
    assert rewards == g["meta"]["episode_rewards"][:3] and lengths == [1000] * 3
    env.close()
    g = load_golden("g6_rmcsa_testcfg_sapff")
    kw = dict(g["meta"]["kwargs"])
    kw.pop("seed")
    env = orl.RMCSAEnv(topology="nsfnet_chen", seed=10, **kw)
    rewards, lengths = orl.evaluate_heuristic(env, orl.shortest_available_path_best_modulation_first_core_first_fit,
                                              n_eval_episodes=3, return_episode_rewards=True)
    assert rewards == g["meta"]["episode_rewards"] and lengths == [999] * 3
    env.close()


def test_vecenv_adapter_on_hip_batch():
    """SB3-VecEnv-shaped adapter over a HIP batch: auto (soft) reset, terminal_observation, Monitor-style episode rows."""
    import optical_rl_gym_amd as orl
    from optical_rl_gym_amd.vec_env import OpticalVecEnv

    g = load_golden("g4_deeprmsa_j1_sap")
    kw = dict(g["meta"]["kwargs"])
    kw.pop("seed")
    batch = orl.BatchedDeepRMSAEnv("nsfnet_chen", num_envs=3, seeds=[10, 11, 12], **kw)
    venv = OpticalVecEnv(batch)
    obs = venv.reset()
    assert obs.shape == (3, 54) and np.array_equal(obs[0], g["obs"][0])
    for _ in range(120):
        obs, rew, done, infos = venv.step(batch.policy("SAP")[:, 0].copy())
        for i in np.flatnonzero(done):
            assert infos[i]["episode"]["l"] == 49 and np.array_equal(infos[i]["terminal_observation"], obs[i])
    assert [r["r"] for r in venv.episode_log][0::3][:2] == g["meta"]["episode_rewards"][:2]
    venv.close()


@pytest.mark.parametrize("name", golden_names("w"))
def test_hip_reproduces_wrapper_and_event_fixtures(name):
    """PathOnlyFirstFitAction on the device (policy PATH_FF), SimpleMatrixObservation (k_matrix_obs), the 2-D action
    histograms, seed() and reset(full) in the middle of a run — against fixtures captured from the reference
    (oracle/gen_golden_wrappers.py)."""
    g = load_golden(name)
    extra = dict(action_histograms=True) if "actions_output" in g else {}
    env = _product(g["meta"], **extra)
    replay_w(env, g, _exact(name))
    assert not env.flags().any() or name.startswith("w3_rmsa")  # (w3_rmsa carries the "reseeded" flag)
    env.close()


@pytest.mark.parametrize("devs", DEVICE_PAIRS)
def test_c_abi_multi_device_group_equals_one_batch(devs):
    """orl_multi_create / orl_multi_run (SURVEY 8e's `n_devices, device_ids` at the C ABI): three shards — on device 0, or
    spread over two GPUs — of 50 envs run their device loops at once from three host threads and leave the state of one
    50-env batch."""
    import ctypes as C

    _need_devices(devs)

    import optical_rl_gym_amd as orl
    from optical_rl_gym_amd import _lib

    kw = dict(load=300, mean_service_holding_time=25, episode_length=40, num_spectrum_resources=320)
    seeds = [90 + i for i in range(50)]
    one = orl.make("RMSA", topology="nsfnet_chen", num_envs=50, seeds=seeds, **kw)
    lib = one.lib
    m = C.c_void_p()
    sd = np.array(seeds, np.int64)
    dev3 = (C.c_int * 3)(devs[0], devs[1], devs[0])
    _lib.check(lib.orl_multi_create(C.byref(one._cfg), C.byref(one._desc), 50, sd.ctypes.data, 3, dev3, C.byref(m)), lib)
    assert lib.orl_multi_n_shards(m) == 3
    stats = (_lib.RunStats * 3)()
    _lib.check(lib.orl_multi_run(m, 1, 130, stats), lib)
    one.run("SAP_FF", 130)
    ref = one.counters()
    lo = 0
    for r, want in enumerate((17, 17, 16)):
        first, n = C.c_int64(), C.c_int64()
        h = lib.orl_multi_shard(m, r, C.byref(first), C.byref(n))
        assert (first.value, n.value) == (lo, want) and stats[r].ms_total > 0
        got = np.zeros((want, 8), np.int64)
        _lib.check(lib.orl_batch_get_counters(h, got.ctypes.data), lib)
        assert np.array_equal(got, ref[lo:lo + want])
        sl = np.zeros((1, one.topology.n_links, 320), np.uint8)
        _lib.check(lib.orl_batch_get_slots(h, want - 1, sl.ctypes.data), lib)
        assert np.array_equal(sl, one.slots(lo + want - 1))
        lo += want
    bad = (C.c_int * 2)(0, 99)
    assert lib.orl_multi_create(C.byref(one._cfg), C.byref(one._desc), 50, sd.ctypes.data, 2, bad, C.byref(C.c_void_p())) == -1
    lib.orl_multi_destroy(m)
    one.close()


def test_episode_log_rearmed_with_a_smaller_capacity():
    """orl_batch_episode_log: the row stride of the log is the armed capacity, whatever an earlier, larger arming allocated
    (round-2 advice: a smaller re-arm overflowed the caller's [n_envs][capacity] buffer)."""
    import optical_rl_gym_amd as orl

    kw = dict(load=300, mean_service_holding_time=25, episode_length=20, num_spectrum_resources=320)
    env = orl.make("RMSA", topology="nsfnet_chen", num_envs=40, seeds=list(range(40)), **kw)
    big, _ = env.evaluate("SAP_FF", 6)
    env2 = orl.make("RMSA", topology="nsfnet_chen", num_envs=40, seeds=list(range(40)), **kw)
    small, lengths = env2.evaluate("SAP_FF", 2)
    assert big.shape == (40, 6) and small.shape == (40, 2) and (lengths == 19).all()
    assert np.array_equal(big[:, :2], small)
    again, _ = env.evaluate("SAP_FF", 2)   # smaller than the first arming of `env`: guard pages would tell, the values do too
    more, _ = env2.evaluate("SAP_FF", 6)   # and growing after a small one
    assert again.shape == (40, 2) and more.shape == (40, 6) and (again >= 0).all() and (again <= 19).all()
    assert np.array_equal(np.concatenate([small, more], 1)[:, :8], np.concatenate([big, again], 1))
    env.close()
    env2.close()


def test_hip_reproduces_rmcsa_4d_action_histograms():
    """RMCSAEnv.actions_output / actions_taken (rmcsa_env.py:145-180, 219, 273, 284-289; a full reset clears them, :437-454)
    kept on the device as an opt-in [2][k+1][M+1][C+1][S+1] array per env: fixture captured from the reference
    (oracle/gen_golden_hist.py), host-driven steps; then the device-resident loop against the oracle on a small batch,
    and the gym-shaped front end's attributes."""
    from oracle.oracle import OracleBatch
    import optical_rl_gym_amd as orl

    g = load_golden("h1_rmcsa_hist4d")
    env = _product(g["meta"], action_histograms=True)
    replay_h(env, g, _exact("h1_rmcsa_hist4d"))
    env.close()
    kw = dict(g["meta"]["kwargs"])
    kw.pop("seed")
    seeds = [700 + i for i in range(24)]
    dev = orl.make("RMCSA", topology="nsfnet_chen", num_envs=24, seeds=seeds, action_histograms=True, **kw)
    ora = OracleBatch("RMCSA", "nsfnet_chen", seeds, **kw)
    dev.run("SAP_BM_FC_FF", 300)
    ora.run("SAP_BM_FC_FF", 300)
    for e in (0, 7, 23):
        for a, b in zip(dev.action_histograms_of(e), ora.action_histograms_of(e)):
            assert a.shape == b.shape and np.array_equal(a, b)
        assert dev.action_histograms_of(e)[0].sum() == 300
    dev.close()
    one = orl.RMCSAEnv(topology="nsfnet_chen", **g["meta"]["kwargs"])
    for t in range(40):
        one.step([int(x) for x in g["actions"][t]])
    assert one.actions_output.shape == tuple(g["meta"]["shape"]) and one.actions_output.sum() == 40 and one.actions_taken.sum() == 40
    assert one.episode_actions_output.shape == one.actions_output.shape and not one.episode_actions_output.any()


@pytest.mark.parametrize("gname,policy", [("g2_rmsa_cfg2_sapff", "SAP_FF"), ("g5_rwa_testcfg_sapff", "SAP_FF"),
                                          ("g4_deeprmsa_j2_sap", "SAP"), ("g6_rmcsa_7x320_sapff", "SAP_BM_FC_FF")])
def test_full_and_masked_resets_match_oracle(gname, policy):
    """reset(only_episode_counters=False) after stepping, for all envs and for a mask of envs, and masked soft resets
    (rmsa_env.py:284-359, rwa_env.py:164-208, rmcsa_env.py:386-483): the batch keeps equal to the oracle."""
    from oracle.oracle import OracleBatch

    meta = load_golden(gname)["meta"]
    kw = dict(meta["kwargs"])
    kw.pop("seed")
    kw["episode_length"] = 55
    n = 80
    seeds = [4000 + 3 * i for i in range(n)]
    ora = OracleBatch(meta["env"], meta["topology"], seeds, **kw)
    dev = _product(dict(meta, kwargs=dict(kw, seed=0)), num_envs=n, seeds=seeds)
    chk = _exact(gname)
    rs = np.random.RandomState(11)

    def compare(tag):
        chk(tag, "counters", dev.counters(), ora.counters())
        chk(tag, "services", dev.services(), ora.services())
        chk(tag, "active", dev.active(), np.array([ora.n_active(i) for i in range(n)]))
        for e in (0, 17, n - 1):
            chk(tag, "slots", dev.slots(e), ora.slots(e))
            chk(tag, "link_stats", dev.link_stats(e), ora.link_stats(e))
            chk(tag, "net_stats", dev.net_stats(e), ora.net_stats(e))
        if dev.obs_dim:
            chk(tag, "obs", dev.observation(), ora.observation())

    dev.run(policy, 130); ora.run(policy, 130)
    mask = (rs.random_sample(n) < 0.4).astype(np.uint8)
    dev.reset(full=True, mask=mask); ora.reset(full=True, mask=mask)
    compare(1)
    dev.run(policy, 90); ora.run(policy, 90)
    compare(2)
    for t in range(40):  # host-driven steps after a masked full reset
        a_o, a_d = ora.policy(policy), dev.policy(policy)
        chk(t, "actions", a_d, a_o)
        _, r_o, d_o, i_o = ora.step(a_o, auto_reset=True)
        _, r_d, d_d, i_d = dev.step(a_d, auto_reset=True)
        chk(t, "reward", r_d, r_o); chk(t, "done", d_d, d_o); chk(t, "info", i_d, i_o)
    dev.reset(full=True); ora.reset(full=True)
    compare(3)
    dev.run(policy, 70); ora.run(policy, 70)
    mask = (rs.random_sample(n) < 0.5).astype(np.uint8)
    dev.reset(full=False, mask=mask); ora.reset(full=False, mask=mask)
    compare(4)
    dev.run(policy, 60); ora.run(policy, 60)
    compare(5)
    assert not dev.flags().any()
    dev.close()


def test_pending_release_overflow_is_reported():
    """A batch whose event_capacity is too small for its load must say so (ORL_E_OVERFLOW -> OverflowError) instead of
    silently dropping the release (the reference's heap is unbounded), in run() and in step()."""
    import optical_rl_gym_amd as orl

    kw = dict(load=300, mean_service_holding_time=25, episode_length=1000, num_spectrum_resources=320)
    for impl in ("persist", "wave64"):
        env = orl.BatchedRMSAEnv("nsfnet_chen", num_envs=64, seeds=list(range(64)), event_capacity=32, **kw)
        with pytest.raises(OverflowError):
            if impl == "persist":
                env.run("SAP_FF", 400)
            else:
                for _ in range(400):
                    env.step(env.policy("SAP_FF"), auto_reset=True)
        assert (env.flags() & 1).any()
        with pytest.raises(OverflowError):
            env.check()  # sticky: those envs have lost a release for good
        env.close()
    ok = orl.BatchedRMSAEnv("nsfnet_chen", num_envs=64, seeds=list(range(64)), **kw)  # the default capacity never overflows
    ok.run("SAP_FF", 1500)
    ok.check()
    ok.close()


def test_out_of_range_actions_raise_like_the_reference():
    """rmsa_env.py:167 / rwa_env.py:103 / rmcsa_env.py:219 raise IndexError on actions_output[...] before anything is
    modified; so does a host-driven step().  Device-resident actions cannot be checked beforehand: the kernel treats them
    as a rejection and the next synchronous call reports them."""
    import torch

    import optical_rl_gym_amd as orl

    kw = dict(load=300, mean_service_holding_time=25, episode_length=100, num_spectrum_resources=320)
    env = orl.BatchedRMSAEnv("nsfnet_chen", num_envs=8, seeds=list(range(8)), **kw)
    env.run("SAP_FF", 30)
    before = (env.counters().copy(), env.services().copy(), env.slots(3).copy())
    good = env.policy("SAP_FF").copy()
    for bad_row in ([6, 0, 0, 0], [0, 321, 0, 0], [-1, 0, 0, 0], [0, -5, 0, 0]):
        a = good.copy()
        a[3] = bad_row
        with pytest.raises(IndexError):
            env.step(a)
        assert np.array_equal(env.counters(), before[0]) and np.array_equal(env.services(), before[1])
        assert np.array_equal(env.slots(3), before[2])
    env.step(good)  # the batch is still usable
    assert (env.counters()[:, 0] == before[0][:, 0] + 1).all()
    # device-resident actions
    act = env.device_tensor("actions")
    a = torch.from_numpy(env.policy("SAP_FF").copy())
    a[5, 0] = 17
    act.copy_(a)
    torch.cuda.synchronize()
    env.step(None, fetch=False)
    with pytest.raises(IndexError):
        env.check()
    env.check()  # reported once
    assert env.counters()[5, 0] == before[0][5, 0] + 2  # it was treated as a rejection: the env moved on
    env.close()
    rwa = orl.BatchedRWAEnv("nsfnet_chen", num_envs=4, seeds=[1, 2, 3, 4], load=100, mean_service_holding_time=10,
                            allow_rejection=False)
    with pytest.raises(IndexError):  # without the reject action the arrays are [k][S] (rwa_env.py:52-58)
        rwa.step(np.array([[5, 0]] * 4))
    rwa.close()


def test_path_only_first_fit_in_the_device_loop(monkeypatch):
    """run("PATH_FF") with a fixed path column: the persistent kernel (all forms) against the one-wavefront-per-env kernel
    and the oracle."""
    import optical_rl_gym_amd as orl
    from oracle.oracle import OracleBatch

    kw = dict(load=320, mean_service_holding_time=25, episode_length=60, num_spectrum_resources=320, allow_rejection=True)
    n = 512
    seeds = [9000 + i for i in range(n)]
    paths = np.random.RandomState(2).randint(0, 6, n)
    out = {}
    for name in ("wave64", "persist", "persist_global", "persist_lds", "split2"):
        force_impl(monkeypatch, name)
        env = orl.BatchedRMSAEnv("nsfnet_chen", num_envs=n, seeds=seeds, **kw)
        env.set_paths(paths)
        env.run("PATH_FF", 220)
        out[name] = (env.counters().copy(), env.services().copy(), env.slots(7).copy(), env.link_stats(7).copy())
        env.close()
    ora = OracleBatch("RMSA", "nsfnet_chen", seeds[:16], **kw)
    ora.set_paths(paths[:16])
    ora.run("PATH_FF", 220)
    chk = _exact("path_ff")
    chk(0, "oracle counters", out["wave64"][0][:16], ora.counters())
    chk(0, "oracle slots", out["wave64"][2], ora.slots(7))
    for name in ("persist", "persist_global", "persist_lds", "split2"):
        for k, (x, y) in enumerate(zip(out[name], out["wave64"])):
            chk(k, name, x, y)


def test_two_shards_on_one_gpu_equal_the_unsharded_batch():
    """Multi-GPU is one process per GPU over contiguous env ranges with seeds base + index and no collective
    (optical_rl_gym_amd/sharding.py): two shards, here as two batches on the same GPU, reproduce the unsharded batch."""
    import optical_rl_gym_amd as orl
    from bench import WORKLOADS
    from optical_rl_gym_amd.sharding import shard_range, shard_seeds

    fam, topo, kw, policy = WORKLOADS["cfg2"]
    kw = dict(kw, episode_length=80)
    total = 3000  # not a multiple of 8 per shard: ragged last wavefront
    whole = orl.make(fam, topology=topo, num_envs=total, seeds=shard_seeds(10, total, 0, 1), **kw)
    whole.run(policy, 200)
    for rank in (0, 1):
        lo, hi = shard_range(total, rank, 2)
        part = orl.make(fam, topology=topo, num_envs=hi - lo, seeds=shard_seeds(10, total, rank, 2), **kw)
        part.run(policy, 200)
        assert np.array_equal(part.counters(), whole.counters()[lo:hi])
        assert np.array_equal(part.services(), whole.services()[lo:hi])
        assert np.array_equal(part.active(), whole.active()[lo:hi])
        for e in (0, (hi - lo) // 2, hi - lo - 1):
            assert np.array_equal(part.slots(e), whole.slots(lo + e))
            assert np.array_equal(part.link_stats(e), whole.link_stats(lo + e))
        part.close()
    whole.close()


def test_batched_evaluate_heuristic_on_device():
    """evaluate_heuristic over a whole batch in one device-resident run (the kernels log every finished episode): per-env
    episode returns equal the oracle's host loop, and env 0 reproduces the reference script's numbers."""
    import optical_rl_gym_amd as orl
    from tests.oracle_backend import OracleBackend

    kw = dict(allow_rejection=True, load=50, mean_service_holding_time=25, episode_length=100,
              num_spectrum_resources=64, bit_rate_selection="discrete")
    seeds = [10 + i for i in range(24)]
    for heur, name, mean, std in ((orl.shortest_available_path_first_fit, "SAP_FF", 95.0, 3.2558),
                                  (orl.least_loaded_path_first_fit, "LLP_FF", 95.1, 3.3897)):
        dev = orl.BatchedRMSAEnv("nsfnet_chen", num_envs=24, seeds=seeds, **kw)
        ora = OracleBackend("RMSA", "nsfnet_chen", seeds, **kw)
        rew_d, len_d = orl.evaluate_heuristic(dev, heur, n_eval_episodes=10, return_episode_rewards=True)
        rew_o, len_o = ora.evaluate(name, 10)
        assert np.array_equal(np.array(rew_d), rew_o) and np.array_equal(np.array(len_d), len_o)
        assert (round(float(np.mean(rew_d[0])), 4), round(float(np.std(rew_d[0])), 4)) == (mean, std)
        assert np.array_equal(dev.counters(), ora.counters())  # the last episode is left un-reset, like the harness does
        dev.close()
    g = load_golden("g4_deeprmsa_j1_sap")
    dkw = dict(g["meta"]["kwargs"])
    dkw.pop("seed")
    dev = orl.BatchedDeepRMSAEnv("nsfnet_chen", num_envs=5, seeds=[10, 11, 12, 13, 14], **dkw)
    mean, std = orl.evaluate_heuristic(dev, "SAP", n_eval_episodes=10)
    assert (round(float(mean[0]), 4), round(float(std[0]), 4)) == (43.2, 4.6)
    dev.close()


def test_vecenv_on_hip_dlpack_and_monitor_file(tmp_path):
    """The SB3-shaped VecEnv over a HIP batch: 2 000 random-policy steps driven through zero-copy DLPack tensors, float32
    observations, and a Monitor file in the format of the reference's examples/heuristics/bkp/rmsa-heu/sap_ff.monitor.csv
    (`#{"t_start": ..., "env_id": ...}` / `r,l,t,<info keywords>` / one row per episode)."""
    import csv
    import json

    import torch

    import optical_rl_gym_amd as orl

    kw = dict(mean_service_holding_time=7.5, mean_service_inter_arrival_time=1.0 / 12.0, j=1, episode_length=50)
    batch = orl.BatchedDeepRMSAEnv("nsfnet_chen", num_envs=16, seeds=list(range(16)), **kw)
    venv = orl.OpticalVecEnv(batch, obs_dtype=np.float32)
    assert venv.observation_space.shape == (54,) and venv.action_space.n == 5
    obs = venv.reset()
    assert obs.dtype == np.float32 and obs.shape == (16, 54)
    rs = np.random.RandomState(0)
    total_done = 0
    held = []  # SB3 keeps the previous observation across one further step: the arrays of the last two steps must stay intact
    for t in range(2000 // 16):
        obs, rew, done, infos = venv.step(rs.randint(0, 5, 16))
        # float32 cast on the device of the float64 observation the batch holds; written into a ring of three host arrays
        assert obs.dtype == np.float32 and np.array_equal(obs, batch.observation().astype(np.float32))
        for prev, copy in held:
            assert np.array_equal(prev, copy)
        held = (held + [(obs, obs.copy())])[-2:]
        total_done += int(done.sum())
        for i in np.flatnonzero(done):
            assert infos[i]["episode"]["l"] == 49 and infos[i]["terminal_observation"].dtype == np.float32
    assert total_done == len(venv.episode_log) == 16 * ((2000 // 16) // 49)
    path = tmp_path / "rnd.monitor.csv"
    venv.save_monitor_csv(str(path), env_id="DeepRMSA-v0")
    lines = path.read_text().splitlines()
    head = json.loads(lines[0][1:])
    assert lines[0][0] == "#" and set(head) == {"t_start", "env_id"}
    assert lines[1] == "r,l,t,episode_service_blocking_rate,episode_bit_rate_blocking_rate"
    rows = list(csv.reader(lines[2:]))
    assert len(rows) == total_done and all(len(r) == 5 and int(r[1]) == 49 for r in rows)
    # the same arrays through DLPack, no copy: an agent on the GPU writes actions and reads results in place
    act = torch.from_dlpack(batch.device_array("actions"))
    rew_t = torch.from_dlpack(batch.device_array("reward"))
    obs_t = torch.from_dlpack(batch.device_array("obs"))
    assert act.is_cuda and act.shape == (16, 4) and rew_t.dtype == torch.float64 and obs_t.shape == (16, 54)
    assert act.data_ptr() == batch.device_tensor("actions").data_ptr()
    a = torch.zeros(16, 4, dtype=torch.int32, device="cuda")
    act.copy_(a)
    torch.cuda.synchronize()
    batch.step(None, auto_reset=True, fetch=False)
    batch.sync()
    assert np.array_equal(obs_t.cpu().numpy(), batch.observation())
    assert venv.get_attr("services_processed") == [int(x) for x in batch.counters()[:, 0]]
    assert venv.env_method("seed", 9, indices=[1, 2]) == [10, 11]
    venv.close()


@pytest.mark.parametrize("devs", DEVICE_PAIRS)
def test_multi_device_wrapper_on_one_gpu(devs):
    """make(..., device_ids=[0, 0] / [0, 1]): two shards behind one batch object (both on the only GPU of the box, or on two
    GPUs), each driven by its own host thread and stream — same results as the single batch; an action outside the action space
    in the SECOND shard's slice is refused before the first shard has queued anything."""
    import optical_rl_gym_amd as orl
    from bench import WORKLOADS

    _need_devices(devs)

    fam, topo, kw, policy = WORKLOADS["cfg2"]
    kw = dict(kw, episode_length=70)
    n = 1000
    seeds = [10 + i for i in range(n)]
    one = orl.make(fam, topology=topo, num_envs=n, seeds=seeds, **kw)
    two = orl.make(fam, topology=topo, num_envs=n, seeds=seeds, device_ids=list(devs), **kw)
    assert isinstance(two, orl.MultiDeviceBatch) and two.num_envs == n and [s.num_envs for s in two.shards] == [500, 500]
    one.run(policy, 150)
    two.run(policy, 150)
    for t in range(20):
        a = one.policy(policy).copy()
        assert np.array_equal(two.policy(policy), a)
        _, r1, d1, i1 = one.step(a, auto_reset=True)
        _, r2, d2, i2 = two.step(a, auto_reset=True)
        assert np.array_equal(r1, r2) and np.array_equal(d1, d2) and np.array_equal(i1, i2)
    for t in range(10):  # the step in two halves: both shards queued from one thread, then collected
        a = one.policy(policy)[:, :2].astype(np.int64)
        two.step_async(a, auto_reset=True, fetch_info=t % 2 == 0)
        _, r1, d1, i1 = one.step(a, auto_reset=True)
        _, r2, d2, i2 = two.step_wait()
        assert np.array_equal(r1, r2) and np.array_equal(d1, d2) and (i2 is None or np.array_equal(i1, i2))
    # a bad action in the second shard's slice: refused with nothing queued or modified on either shard, and the next step works
    before = two.counters().copy()
    bad = one.policy(policy)[:, :2].astype(np.int64)
    bad[777, 0] = 77
    for call in (two.step_async, two.step):
        with pytest.raises(IndexError):
            call(bad, auto_reset=True)
    assert np.array_equal(two.counters(), before)
    a = one.policy(policy)[:, :2].astype(np.int64)
    two.step_async(a, auto_reset=True)
    _, r1, d1, _ = one.step(a, auto_reset=True)
    _, r2, d2, _ = two.step_wait()
    assert np.array_equal(r1, r2) and np.array_equal(d1, d2)
    # policy + step in one launch per shard
    acts, _, r2, d2, i2 = two.policy_step(policy, auto_reset=True)
    a1 = one.policy(policy).copy()
    _, r1, d1, i1 = one.step(None, auto_reset=True)
    assert np.array_equal(acts, a1) and np.array_equal(r1, r2) and np.array_equal(d1, d2) and np.array_equal(i1, i2)
    assert np.array_equal(one.counters(), two.counters()) and np.array_equal(one.services(), two.services())
    for e in (0, 499, 500, 999):
        assert np.array_equal(one.slots(e), two.slots(e)) and np.array_equal(one.link_stats(e), two.link_stats(e))
    assert one.totals() == two.totals()
    one.close()
    two.close()


def test_facade_queries_and_wrappers_on_hip():
    import optical_rl_gym_amd as orl

    env = orl.RMSAEnv(topology="nsfnet_chen", seed=10, load=300, mean_service_holding_time=25, episode_length=1000,
                      num_spectrum_resources=320, allow_rejection=True)
    for _ in range(400):
        env.step(orl.shortest_available_path_first_fit(env))
    svc = env.current_service
    po, mo = orl.PathOnlyFirstFitAction(env), orl.SimpleMatrixObservation(env)
    avail = np.asarray(env.topology.graph["available_slots"])
    for p, path in enumerate(env.k_shortest_paths[svc.source, svc.destination]):
        n = env.get_number_slots(path)
        both = env.get_available_slots(path)
        fit = [s0 for s0 in range(0, 320 - n) if both[s0:s0 + n].all()]
        assert po.action(p) == ((p, fit[0]) if fit else (5, 320))
        assert env.is_path_free(path, fit[0], n) if fit else True
    assert po.action(5) == (5, 320)
    obs = mo.observation()
    assert obs.dtype == np.float64 and obs.shape == (28 + 22 * 320,) and np.array_equal(obs[28:], avail.reshape(-1))
    assert env.actions_output.sum() == 400 and env.actions_taken.sum() == 400
    with pytest.raises(IndexError):
        env.step((0, 400))
    env.close()


# QoSConstrainedRA's host- / agent-driven step: one wavefront per env (k_step; what the library takes below 2 048 envs) and 8 lanes
# per env (k_agent_qos, forced here for any batch size)
@pytest.fixture(params=["wave64", "agent8"])
def qos_impl(request, monkeypatch):
    monkeypatch.setenv("ORL_AGENT_STEP", "1" if request.param == "agent8" else "0")
    return request.param


def _qos_step_kernel_is(env, impl):
    """The library's debug query: 2 = host-driven steps go through the 8-lanes-per-env kernel, 0 = one wavefront per env."""
    return int(env.lib.orl_batch_debug_step_kernel(env._h)) == (2 if impl == "agent8" else 0)


@pytest.mark.parametrize("name", golden_names("q"))
def test_hip_reproduces_qos_fixtures(name, qos_impl):
    """QoSConstrainedRA (qos_constrained_ra.py) on the device against fixtures captured from the reference with its
    constructor repaired at import time: three heuristics and a stored action stream, three service classes."""
    g = load_golden(name)
    env = _product(g["meta"])
    assert _qos_step_kernel_is(env, qos_impl)
    replay_q(env, g, _exact(name))
    assert not env.flags().any()
    env.close()


def test_qos_batches_match_oracle_and_run_equals_stepping(qos_impl):
    import optical_rl_gym_amd as orl
    from oracle.oracle import OracleBatch

    kw = dict(load=900, mean_service_holding_time=25, episode_length=45, num_spectrum_resources=32, num_service_classes=4,
              classes_arrival_probabilities=[0.1, 0.4, 0.3, 0.2], classes_reward=[8.0, 4.0, 2.0, 1.0], allow_rejection=True)
    n = 96
    seeds = [700 + 3 * i for i in range(n)]
    chk = _exact("qos")
    for policy in ("SP_FF", "SAP_FF", "LLP_FF"):
        ora = OracleBatch("QoSConstrainedRA", "nsfnet_chen", seeds, **kw)
        dev = orl.make("QoSConstrainedRA-v0", topology="nsfnet_chen", num_envs=n, seeds=seeds, **kw)
        assert _qos_step_kernel_is(dev, qos_impl)
        dev.run(policy, 160)
        ora.run(policy, 160)
        for t in range(120):
            a_o, a_d = ora.policy(policy), dev.policy(policy)
            chk(t, "actions", a_d, a_o)
            _, r_o, d_o, i_o = ora.step(a_o, auto_reset=True)
            _, r_d, d_d, i_d = dev.step(a_d, auto_reset=True)
            chk(t, "reward", r_d, r_o); chk(t, "done", d_d, d_o); chk(t, "info", i_d, i_o)
        chk(0, "counters", dev.counters(), ora.counters())
        chk(0, "services", dev.services(), ora.services())
        for e in (0, 50, n - 1):
            chk(e, "spectrum", dev.spectrum(e), ora.spectrum(e))
            chk(e, "link_stats", dev.link_stats(e)[[0, 3]], ora.link_stats(e)[[0, 3]])
            chk(e, "n_active", dev.n_active(e), ora.n_active(e))
        mask = (np.arange(n) % 3 == 0).astype(np.uint8)
        dev.reset(full=True, mask=mask); ora.reset(full=True, mask=mask)
        dev.run(policy, 50); ora.run(policy, 50)
        chk(1, "counters", dev.counters(), ora.counters())
        chk(1, "spectrum", dev.spectrum(3), ora.spectrum(3))
        with pytest.raises(IndexError):
            dev.step(np.full((n, 1), 6))
        dev.close()


def test_qos_large_batches_take_the_8_lane_kernel(monkeypatch):
    """From 20 480 envs the library steps QoSConstrainedRA through k_agent_qos by itself; every env must end where the
    one-wavefront-per-env kernel leaves it (agent-driven steps on device-resident actions, a device run in the middle)."""
    import optical_rl_gym_amd as orl

    kw = dict(load=300, mean_service_holding_time=25, episode_length=35, num_spectrum_resources=48, num_service_classes=3,
              classes_arrival_probabilities=[0.2, 0.5, 0.3], classes_reward=[4.0, 2.0, 1.0], allow_rejection=True)
    n = 24576
    seeds = [9 + 5 * i for i in range(n)]
    out = {}
    for name, knob in (("default", None), ("wave64", "0")):
        if knob is None:
            monkeypatch.delenv("ORL_AGENT_STEP", raising=False)
        else:
            monkeypatch.setenv("ORL_AGENT_STEP", knob)
        env = orl.make("QoSConstrainedRA-v0", topology="nsfnet_chen", num_envs=n, seeds=seeds, **kw)
        assert int(env.lib.orl_batch_debug_step_kernel(env._h)) == (2 if knob is None else 0)
        rew = np.zeros(n)
        for phase in range(2):
            for _ in range(60):
                env.policy("SAP_FF", fetch=False)
                env.step(None, auto_reset=True, fetch=False)
                env.sync()  # (the batch steps on its own stream)
                rew += env.device_tensor("reward").cpu().numpy()
            if phase == 0:
                env.run("LLP_FF", 40)
        out[name] = [env.counters().copy(), env.services().copy(), env.active().copy(), env.flags().copy(), env.link_stats_all().copy(),
                     np.stack([env.spectrum(i) for i in (0, n // 2, n - 1)]), rew]
        env.close()
    chk = _exact("qos 8-lane kernel")
    for k, (x, y) in enumerate(zip(out["default"], out["wave64"])):
        chk(k, "item", x, y)


@pytest.mark.parametrize("fam,n,kw,pol", [
    ("RMSA", 2304, dict(load=300, mean_service_holding_time=25, episode_length=30, num_spectrum_resources=320), "SAP_FF"),
    ("DeepRMSA", 2304, dict(mean_service_holding_time=7.5, mean_service_inter_arrival_time=1.0 / 12.0, j=1, episode_length=30), "SAP"),
    ("RWA", 96, dict(load=450, mean_service_holding_time=25, episode_length=30, allow_rejection=True), "SAP_FF")])
def test_step_in_two_halves_equals_step(fam, n, kw, pol):
    """orl_batch_step_async / orl_batch_step_wait (VecEnv.step_async / step_wait): the compact action rows an agent hands over
    (int64 and int32, 1 or 2 columns) are checked and widened by the library, everything is queued, and the results equal those
    of the synchronous step on a twin batch — k_agent (>= 2 048 envs) and k_step; an out-of-range action is refused before
    anything is modified, a second step before the first is collected is refused too."""
    import optical_rl_gym_amd as orl
    from optical_rl_gym_amd._lib import OrlError

    seeds = [40 + i for i in range(n)]
    a = orl.make(fam, topology="nsfnet_chen", num_envs=n, seeds=seeds, **kw)
    b = orl.make(fam, topology="nsfnet_chen", num_envs=n, seeds=seeds, **kw)
    width = 1 if fam == "DeepRMSA" else 2
    chk = _exact(fam + " async")
    obs32 = a.host_array((n, a.obs_dim), np.float32) if a.obs_dim else None
    for t in range(70):
        acts = b.policy(pol)[:, :width].copy()
        compact = acts.astype(np.int64) if t % 2 else np.ascontiguousarray(acts, np.int32)
        a.step_async(compact, auto_reset=True, obs_out=obs32 if (obs32 is not None and t % 3 == 0) else None, fetch_info=t % 4 != 1)
        if t == 5:
            with pytest.raises(OrlError):
                a.step_async(compact, auto_reset=True)
        o_a, r_a, d_a, i_a = a.step_wait()
        o_b, r_b, d_b, i_b = b.step_sync_abi(acts, auto_reset=True)  # orl_batch_step itself, as a C caller uses it
        chk(t, "reward", r_a, r_b); chk(t, "done", d_a, d_b)
        if i_a is not None:
            chk(t, "info", i_a, i_b)
        if o_b is not None:
            chk(t, "obs", o_a, o_b.astype(o_a.dtype))
    before = a.counters().copy()
    bad = np.zeros((n, width), np.int64)
    bad[n // 2, 0] = 10 ** 6 if fam == "DeepRMSA" else 77
    if fam != "DeepRMSA":  # (DeepRMSA takes any integer: deeprmsa_env.py:48-58)
        with pytest.raises(IndexError):
            a.step_async(bad, auto_reset=True)
        chk(0, "untouched", a.counters(), before)
    with pytest.raises(OrlError):
        a.step_wait()  # nothing pending
    chk(0, "counters", a.counters(), b.counters())
    chk(0, "services", a.services(), b.services())
    a.close(); b.close()


def test_qos_evaluate_on_device_equals_the_harness(qos_impl):
    """evaluate_heuristic (utils.py:103-141) for QoSConstrainedRA on the device: the reward of an accepted service is its
    class's reward (qos_constrained_ra.py:131-136), so the kernels keep each episode's float64 reward sum in step order;
    against the harness's accounting played on the oracle (reset, steps until done, += reward), with rewards that do not sum
    exactly in another order."""
    import optical_rl_gym_amd as orl
    from oracle.oracle import OracleBatch

    kw = dict(load=700, mean_service_holding_time=25, episode_length=37, num_spectrum_resources=24, num_service_classes=3,
              classes_arrival_probabilities=[0.2, 0.5, 0.3], classes_reward=[0.7, 0.1, 1.0 / 3.0], allow_rejection=True)
    n, n_ep = 40, 4
    seeds = [900 + 5 * i for i in range(n)]
    for policy in ("SP_FF", "SAP_FF"):
        dev = orl.make("QoSConstrainedRA-v0", topology="nsfnet_chen", num_envs=n, seeds=seeds, **kw)
        ora = OracleBatch("QoSConstrainedRA", "nsfnet_chen", seeds, **kw)
        dev.run(policy, 55); ora.run(policy, 55)  # somewhere inside an episode
        rewards, lengths = dev.evaluate(policy, n_ep)
        ora.reset(full=False)
        exp = np.zeros((n, n_ep))
        for ep in range(n_ep):
            for t in range(kw["episode_length"]):
                last = ep == n_ep - 1 and t == kw["episode_length"] - 1
                _, r, d, _ = ora.step(ora.policy(policy), auto_reset=not last)
                exp[:, ep] += r
            assert d.all()
        assert np.array_equal(rewards, exp) and (lengths == kw["episode_length"]).all()
        assert len(np.unique(rewards)) > n  # class rewards, not counts
        _exact("qos evaluate")(0, "counters", dev.counters(), ora.counters())
        again, _ = dev.evaluate(policy, 2)  # re-armed with a smaller capacity
        assert again.shape == (n, 2) and (again > 0).all()
        dev.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n_nodes,chords", [(5, 0), (6, 2), (4, 2)])  # 5, 8 and 6 links: fewer than a group's 8 lanes, exactly 8, a tail of 6
@pytest.mark.parametrize("fam", ["RMSA", "DeepRMSA", "RWA"])
def test_tiny_topologies_through_every_step_form(n_nodes, chords, fam, impl, tmp_path):
    """Rings of 4-6 nodes (with chords): the link loops of every kernel — numpy's pairwise mean over the links in info (8-lane
    partial sums only from 8 links on, then a tail), the cumulative tables of random.choices with 3-5 entries, the row phase
    with fewer links than lanes — against the oracle, for host-driven steps (info, reward, done, observation) and a
    device-resident run, in every step implementation."""
    import optical_rl_gym_amd as orl
    from optical_rl_gym_amd.topology_io import build_topology, save_topology
    from oracle.oracle import OracleBatch

    if impl == "persist_pair" and (n_nodes, chords) != (6, 2):
        pytest.skip("the two-wavefront form needs a specialisation library per configuration: one ring")
    links = [(i + 1, (i + 1) % n_nodes + 1, 400 + 150 * i) for i in range(n_nodes)]
    links += [(1 + c, 1 + (c + 2) % n_nodes, 900 + 100 * c) for c in range(chords)]
    raw = tmp_path / "ring.txt"
    raw.write_text("# ring\n%d\n%d\n%s\n" % (n_nodes, len(links), "\n".join("%d %d %d" % l for l in links)))
    npz = str(tmp_path / "ring_5-paths_6-modulations.npz")
    save_topology(build_topology(str(raw), name="ring", k_paths=3), npz)
    kw = dict(episode_length=25, mean_service_holding_time=8.0)
    if fam == "DeepRMSA":
        kw.update(mean_service_inter_arrival_time=0.25, j=2, num_spectrum_resources=64)
    elif fam == "RWA":
        kw.update(load=20, num_spectrum_resources=16, allow_rejection=True)
    else:
        kw.update(load=60, num_spectrum_resources=70, allow_rejection=True)
    policy = "SAP" if fam == "DeepRMSA" else "SAP_FF"
    B = 24
    seeds = [77 + i for i in range(B)]
    dev = orl.make(fam, topology=npz, num_envs=B, seeds=seeds, **kw)
    ora = OracleBatch(fam, npz, seeds, **kw)
    chk = _exact("ring %d+%d %s %s" % (n_nodes, chords, fam, impl))
    for t in range(60):
        a = ora.policy(policy)
        chk(t, "policy", dev.policy(policy), a)
        o1, r1, d1, i1 = dev.step(a, auto_reset=True)
        o2, r2, d2, i2 = ora.step(a, auto_reset=True)
        chk(t, "reward", r1, r2)
        chk(t, "done", d1, d2)
        chk(t, "info", i1, i2)
        if dev.obs_dim:
            chk(t, "obs", dev.observation(), ora.observation())
    dev.run(policy, 150)
    ora.run(policy, 150)
    assert _ran_pair_form(dev) or impl != "persist_pair"
    chk(0, "counters", dev.counters(), ora.counters())
    chk(0, "services", dev.services(), ora.services())
    for i in range(B):
        chk(i, "slots", dev.slots(i), ora.slots(i))
        chk(i, "link_stats", dev.link_stats(i), ora.link_stats(i))
        chk(i, "net_stats", dev.net_stats(i), ora.net_stats(i))
    dev.check()
    dev.close()


@pytest.mark.gpu
@pytest.mark.parametrize("fam,kw,policy", [("RMSA", dict(load=300, mean_service_holding_time=25, num_spectrum_resources=320), "SAP_FF"),
                                            ("DeepRMSA", dict(mean_service_holding_time=7.5, mean_service_inter_arrival_time=0.1, j=1), "SAP")])
@pytest.mark.parametrize("devs", DEVICE_PAIRS)
def test_vecenv_sparse_info_equals_the_whole_info_array(fam, kw, policy, devs):
    """OpticalVecEnv leaves info on the device and reads the rows of the envs that finished an episode
    (orl_batch_get_info_rows): the episode rows must be the ones the whole [n_envs][info_dim] array gives — same batch, same
    seeds, stepped once through the VecEnv and once through step() with everything fetched — in both the k_agent range
    (4 096 envs) and the one-wavefront-per-env range (300 envs); the row reader is also checked against the array directly."""
    import optical_rl_gym_amd as orl

    _need_devices(devs)
    for B in (300, 4096):
        seeds = [5 + i for i in range(B)]
        a = orl.make(fam, topology="nsfnet_chen", num_envs=B, seeds=seeds, episode_length=9 + (B % 7), **kw)
        b = orl.make(fam, topology="nsfnet_chen", num_envs=B, seeds=seeds, episode_length=9 + (B % 7), **kw)
        venv = orl.OpticalVecEnv(a)
        assert venv._sparse_info
        venv.reset()
        # ... and the same through two shards behind one VecEnv (both on device 0 here): obs_out slices, per-shard info rows
        m = orl.make(fam, topology="nsfnet_chen", num_envs=B, seeds=seeds, episode_length=9 + (B % 7), device_ids=list(devs), **kw)
        mvenv = orl.OpticalVecEnv(m, obs_dtype=np.float32)
        assert mvenv._sparse_info and mvenv._direct_obs
        mvenv.reset()
        keys = venv.info_keywords
        cols = [a.info_keys.index(k) for k in keys]
        n_rows = 0
        for t in range(30):
            act = b.policy(policy).copy()
            o1, rew, done, infos = venv.step(act if fam != "DeepRMSA" else act[:, 0].copy())
            o3, r3, d3, infos3 = mvenv.step(act if fam != "DeepRMSA" else act[:, 0].copy())
            _, r2, d2, i2 = b.step(act, auto_reset=True)
            assert np.array_equal(rew, r2) and np.array_equal(done, d2.astype(bool))
            assert np.array_equal(r3, r2) and np.array_equal(d3, d2.astype(bool))
            if o1 is not None:
                assert o3.dtype == np.float32 and np.array_equal(o3, o1.astype(np.float32))
            for i in np.flatnonzero(d2):
                n_rows += 1
                assert [infos[i]["episode"][k] for k in keys] == [float(i2[i, j]) for j in cols]
                assert [infos3[i]["episode"][k] for k in keys] == [float(i2[i, j]) for j in cols]
            pick = np.random.RandomState(t).randint(0, B, 17)
            assert np.array_equal(b.info_rows(pick), i2[pick], equal_nan=True)
            assert b.info_rows([]).shape == (0, b.n_info)
        assert n_rows >= 2 * B
        with pytest.raises(orl._lib.OrlError):
            b.info_rows([B])
        venv.close()
        mvenv.close()
        b.close()


@pytest.mark.gpu
def test_agent_on_the_batchs_own_stream_needs_no_synchronisation():
    """orl_batch_stream / env.torch_stream(): a torch "agent" whose kernels are queued on the batch's stream — actions computed
    from the observation on the device, written into the action array, step(None, fetch=False), no host synchronisation in
    the loop — must leave exactly the state of the same loop with a synchronisation around every step."""
    import torch

    import optical_rl_gym_amd as orl

    kw = dict(mean_service_holding_time=7.5, mean_service_inter_arrival_time=0.1, j=1, episode_length=30)
    B = 4096
    seeds = [3 + i for i in range(B)]
    envs = [orl.make("DeepRMSA", topology="nsfnet_chen", num_envs=B, seeds=seeds, **kw) for _ in range(2)]
    assert envs[0].stream_ptr() != 0 and envs[0].stream_ptr() != envs[1].stream_ptr()
    n_act = envs[0].k_paths * envs[0].j

    def agent(obs):  # any deterministic function of the observation that keeps the device busy for a while
        x = obs.float()
        for _ in range(6):
            x = torch.tanh(x @ torch.ones(x.shape[1], x.shape[1], device=x.device) * 1e-3 + x)
        return (x.abs().sum(1) * 1000).long() % n_act

    for env in envs:
        env.reset()
        env.observation()
    torch.cuda.synchronize()
    e = envs[0]
    obs, act = e.device_tensor("obs"), e.device_tensor("actions")
    with torch.cuda.stream(e.torch_stream()):
        for t in range(120):
            act[:, 0] = agent(obs).int()
            e.step(None, auto_reset=True, fetch=False)
    e.sync()
    e = envs[1]
    obs, act = e.device_tensor("obs"), e.device_tensor("actions")
    for t in range(120):
        act[:, 0] = agent(obs).int()
        torch.cuda.synchronize()
        e.step(None, auto_reset=True, fetch=False)
        e.sync()
    chk = _exact("agent on the batch's stream")
    chk(0, "counters", envs[0].counters(), envs[1].counters())
    chk(0, "services", envs[0].services(), envs[1].services())
    chk(0, "slots", envs[0].slots_packed(), envs[1].slots_packed())
    chk(0, "obs", envs[0].device_tensor("obs").cpu().numpy(), envs[1].device_tensor("obs").cpu().numpy())
    assert envs[0].counters()[:, 1].sum() > 0
    for env in envs:
        env.check()
        env.close()


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["cfg2", "cfg3"])
def test_row_caches_kept_between_launches_follow_every_change_of_the_slot_maps(workload, monkeypatch):
    """The persistent kernel leaves its per-row caches with the state and the next launch loads them instead of rebuilding them
    from the slot maps — valid only while nothing else has written slot maps.  Short runs interleaved with everything that
    does: agent-driven steps (k_agent), host steps of a few envs through the one-wavefront kernel is not possible on the same
    batch size, so: k_agent steps, a masked full reset, set_state back to an earlier snapshot, forced deferrals (releases in
    place at the start of the next launch).  Against the oracle on every env after each stage, and against the same sequence
    with the caches rebuilt at every launch."""
    import optical_rl_gym_amd as orl
    from bench import WORKLOADS
    from oracle.oracle import OracleBatch

    fam, topo, kw, policy = WORKLOADS[workload]
    kw = dict(kw, episode_length=60)
    B = 4096
    seeds = [21 + i for i in range(B)]
    chk = _exact(workload + " kept row caches")

    def sequence(dev, ora):
        def same(tag):
            chk(0, tag + " counters", dev.counters(), ora.counters())
            chk(0, tag + " slots", dev.slots_packed(), ora.slots_packed())
            chk(0, tag + " link stats", dev.link_stats_all(), ora.link_stats_all())
            chk(0, tag + " net stats", dev.net_stats_all(), ora.net_stats_all())
        for n in (20, 20, 7, 150, 20):  # launches that load what the previous one stored (150: two launches in one run)
            dev.run(policy, n); ora.run(policy, n)
        same("runs")
        for _ in range(3):  # agent-driven steps change the maps behind the stored caches
            a = ora.policy(policy)
            dev.step(a, auto_reset=True); ora.step(a, auto_reset=True)
        dev.run(policy, 20); ora.run(policy, 20)
        same("after agent steps")
        snap = dev.get_state()
        mask = (np.arange(B) % 3 == 0).astype(np.uint8)
        dev.reset(full=True, mask=mask); ora.reset(full=True, mask=mask)
        dev.run(policy, 25); ora.run(policy, 25)
        same("after a masked full reset")
        after = dev.counters().copy()
        dev.set_state(snap)
        dev.run(policy, 25)
        assert not np.array_equal(dev.counters(), after)  # (a different state than the reset one)
        dev2 = orl.make(fam, topology=topo, num_envs=B, seeds=seeds, **kw)
        dev2.set_state(snap)
        dev2.run(policy, 25)
        chk(0, "set_state: counters", dev.counters(), dev2.counters())
        chk(0, "set_state: slots", dev.slots_packed(), dev2.slots_packed())
        chk(0, "set_state: link stats", dev.link_stats_all(), dev2.link_stats_all())
        dev2.close()

    dev = orl.make(fam, topology=topo, num_envs=B, seeds=seeds, **kw)
    sequence(dev, OracleBatch(fam, topo, seeds, omp=True, **kw))
    dev.close()
    # forced deferrals: releases done in place at the start of the next launch invalidate that wavefront's stored caches
    monkeypatch.setenv("ORL_ITEM_MASKS", "1")
    dev = orl.make(fam, topology=topo, num_envs=B, seeds=seeds, **kw)
    ora = OracleBatch(fam, topo, seeds, omp=True, **kw)
    for n in (20, 20, 20, 130):
        dev.run(policy, n); ora.run(policy, n)
    chk(0, "deferrals: counters", dev.counters(), ora.counters())
    chk(0, "deferrals: slots", dev.slots_packed(), ora.slots_packed())
    chk(0, "deferrals: link stats", dev.link_stats_all(), ora.link_stats_all())
    dev.close()


@pytest.mark.gpu
@pytest.mark.parametrize("gname,policy,n,agent", [
    ("g2_rmsa_cfg2_sapff", "SAP_FF", 2304, None), ("g2_rmsa_cfg2_llpff", "LLP_FF", 96, "1"), ("g2_rmsa_cfg2_sapff", "SP_FF", 96, "0"),
    ("g4_deeprmsa_j2_sap", "SAP", 2304, None), ("g4_deeprmsa_j2_sap", "SP", 64, "1"),
    ("g5_rwa_testcfg_saplf", "SAP_LF", 96, "1"), ("g5_rwa_testcfg_llpff", "LLP_FF", 2304, None),
    ("g6_rmcsa_7x320_sapff", "SAP_BM_FC_FF", 64, "1"), ("g7_rmsa_germany50_sapff", "SAP_FF", 64, "1")])
def test_policy_and_step_in_one_launch_equals_policy_then_step(gname, policy, n, agent, monkeypatch):
    """orl_batch_policy_step: the heuristic's slot scan as the first phase of the step kernel (k_agent<..., FUSED>; all four
    families, every policy form) against policy() followed by step() on a twin batch — actions, reward, done, every info float,
    the DeepRMSA observation per step, then counters, services and sampled slot maps; with ORL_AGENT_STEP=0 the entry point is
    the two launches (k_policy, k_step) and must agree too."""
    import optical_rl_gym_amd as orl

    if agent is None:
        monkeypatch.delenv("ORL_AGENT_STEP", raising=False)
    else:
        monkeypatch.setenv("ORL_AGENT_STEP", agent)
    meta = load_golden(gname)["meta"]
    kw = dict(meta["kwargs"])
    kw.pop("seed")
    kw["episode_length"] = 45
    seeds = [700 + 3 * i for i in range(n)]
    a = orl.make(meta["env"], topology=meta["topology"], num_envs=n, seeds=seeds, **kw)
    b = orl.make(meta["env"], topology=meta["topology"], num_envs=n, seeds=seeds, **kw)
    if agent != "0":
        assert int(a.lib.orl_batch_debug_step_kernel(a._h)) == 2
    chk = _exact(gname + " fused")
    for t in range(160):
        act_a, o_a, r_a, d_a, i_a = a.policy_step(policy, auto_reset=True)
        act_b = b.policy(policy).copy()
        o_b, r_b, d_b, i_b = b.step(None, auto_reset=True)
        chk(t, "actions", act_a, act_b)
        chk(t, "reward", r_a, r_b); chk(t, "done", d_a, d_b); chk(t, "info", i_a, i_b)
        if o_b is not None:
            chk(t, "obs", o_a, o_b)
        if t == 80:  # nothing fetched: the calls only queue work
            for _ in range(5):
                a.policy_step(policy, auto_reset=True, fetch=False)
                b.policy(policy, fetch=False)
                b.step(None, auto_reset=True, fetch=False)
    a.check(); b.check()
    chk(0, "counters", a.counters(), b.counters())
    chk(0, "services", a.services(), b.services())
    for e in (0, n // 2, n - 1):
        chk(e, "slots", a.slots(e), b.slots(e))
        chk(e, "link_stats", a.link_stats(e), b.link_stats(e))
        chk(e, "net_stats", a.net_stats(e), b.net_stats(e))
    a.close(); b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("fam,kw,policy", [("RMSA", dict(load=300, mean_service_holding_time=25, num_spectrum_resources=320), "SAP_FF"),
                                            ("DeepRMSA", dict(mean_service_holding_time=7.5, mean_service_inter_arrival_time=0.1, j=1), "SAP")])
def test_info_mode_rates_only_changes_nothing_but_the_skipped_entries(fam, kw, policy):
    """orl_batch_set_info_mode(1): the 8-lanes-per-env step kernel leaves out the compactness entries and the two link means of
    info (what OpticalVecEnv asks for when SB3 reads blocking rates only); rewards, dones, the four blocking rates, observations
    and the whole state stay those of the full mode."""
    import optical_rl_gym_amd as orl

    n = 2304
    seeds = [300 + i for i in range(n)]
    a = orl.make(fam, topology="nsfnet_chen", num_envs=n, seeds=seeds, episode_length=25, **kw)
    b = orl.make(fam, topology="nsfnet_chen", num_envs=n, seeds=seeds, episode_length=25, **kw)
    assert int(a.lib.orl_batch_debug_step_kernel(a._h)) == 2
    a.set_info_mode(True)
    chk = _exact(fam + " info mode")
    for t in range(60):
        act = b.policy(policy).copy()
        o_a, r_a, d_a, i_a = a.step(act, auto_reset=True)
        o_b, r_b, d_b, i_b = b.step(act, auto_reset=True)
        chk(t, "reward", r_a, r_b); chk(t, "done", d_a, d_b); chk(t, "rates", i_a[:, :4], i_b[:, :4])
        assert np.isnan(i_a[:, 4:8]).all(), "the skipped info columns read NaN in the rates-only mode, not stale values"
        if o_b is not None:
            chk(t, "obs", o_a, o_b)
    a.set_info_mode(False)
    act = b.policy(policy).copy()
    _, _, _, i_a = a.step(act, auto_reset=True)
    _, _, _, i_b = b.step(act, auto_reset=True)
    chk(61, "info, full mode again", i_a, i_b)
    chk(0, "counters", a.counters(), b.counters())
    for e in (0, n // 2, n - 1):
        chk(e, "link_stats", a.link_stats(e), b.link_stats(e))
        chk(e, "net_stats", a.net_stats(e), b.net_stats(e))
    # the VecEnv face: full info unless the caller opts in (the reference's step() always fills every entry, rmsa_env.py:228-264);
    # with rates_only_info the batch goes back to the full mode when the VecEnv is closed
    from optical_rl_gym_amd.vec_env import OpticalVecEnv

    width = a.N_ACTION
    v = OpticalVecEnv(a)
    v.reset()
    v.step(b.policy(policy)[:, :width].copy())
    assert np.isfinite(v.device_tensors()["info"].cpu().numpy()[:, 4:8]).all()
    v2 = OpticalVecEnv(b, rates_only_info=True)
    v2.reset()
    v2.step(a.policy(policy)[:, :width].copy())
    i_v = v2.device_tensors()["info"].cpu().numpy()
    assert np.isnan(i_v[:, 4:8]).all() and np.isfinite(i_v[:, :4]).all()
    v2._rates_only and v2.batch.set_info_mode(False)  # (what close() does before it closes the batch)
    v2._rates_only = False
    _, _, _, i_b = b.step(a.policy(policy), auto_reset=True)
    assert np.isfinite(i_b[:, 4:8]).all()
    a.close(); b.close()
