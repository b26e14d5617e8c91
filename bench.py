#!/usr/bin/env python3
"""Headline benchmark: env-steps/s of RMSA-v0 on NSFNET (320 slots, k=5, load 300 Erlang), batch 65 536
envs per MI355X, on-device KSP-FF policy (BASELINE.json `metric`; SURVEY.md §8d cfg2 at B = 65 536).

One "step" = one batched policy + env.step() over the whole batch, entirely on the device: the persistent kernel
k_persist — one wavefront owns 8 envs for the whole run and alternates a control phase (slot scan + all per-env control +
release detection -> work items) with a row phase (one lane per touched link row); the K timed steps are ceil(K/64) launches.
Inputs are resident in HBM before the timed region.  N > 1: one process per GPU (torchrun), every rank owns its own 65 536 envs (weak scaling, no
collective on the data path; torch.distributed is used only for the barrier and the max-over-ranks time).

    python bench.py --gpus 1 --steps 300 --warmup 1500
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is what a copy achieves

WORKLOADS = {
    # name: (env family, topology, kwargs, policy)
    "cfg2": ("RMSA", "nsfnet_chen", dict(load=300, mean_service_holding_time=25, episode_length=1000,
                                         num_spectrum_resources=320, allow_rejection=False), "SAP_FF"),
    "cfg5": ("RMSA", "germany50", dict(load=800, mean_service_holding_time=25, episode_length=1000,
                                       num_spectrum_resources=320, allow_rejection=False), "SAP_FF"),
    "cfg3": ("DeepRMSA", "nsfnet_chen", dict(mean_service_holding_time=7.5, mean_service_inter_arrival_time=1.0 / 12.0,
                                             j=1, episode_length=50), "SAP"),
    "cfg4": ("RMCSA", "cost239", dict(load=1500, mean_service_holding_time=25, episode_length=1000,
                                          num_spectrum_resources=320, num_spatial_resources=7,
                                          allow_rejection=True), "SAP_BM_FC_FF"),
    "cfg4n": ("RMCSA", "nsfnet_chen", dict(load=1500, mean_service_holding_time=25, episode_length=1000,
                                           num_spectrum_resources=320, num_spatial_resources=7,
                                           allow_rejection=True), "SAP_BM_FC_FF"),
    "cfg1": ("RWA", "nsfnet_chen", dict(load=450, mean_service_holding_time=25, episode_length=1000,
                                        allow_rejection=True), "SAP_FF"),
}


def algorithmic_bytes(env, mean_hops, active):
    """bytes an env-step has to move, per env.
    scan = C*E*W*8 + 24 (one read of the packed link x slot map + 16 B service descriptor + 8 B action)
    step = scan + 32*H + 128*H + 64*ceil(log2 A) + 116 + 128 + 81
    Returned per kernel name (DESIGN.md §5 says which term each kernel of the split pipeline carries)."""
    C, E, S = env.num_spatial_resources, env.topology.n_links, env.num_spectrum_resources
    W = (S + 63) // 64
    scan = C * E * W * 8 + 24
    lg = 64 * math.ceil(math.log2(max(active, 2)))
    fixed = 116 + 128 + 81 + (8 * env.obs_dim if env.obs_dim else 0)
    rest = 32 * mean_hops + 128 * mean_hops + lg + fixed
    return {
        "k_policy": scan,                                   # slot scan alone
        "k_step": rest, "k_step8": rest,                    # monolithic step kernels
        # split pipeline: validation reads of the chosen path + RNG/env record/outputs + the release push
        "k_policy_ctrl_a": scan + 32 * mean_hops + fixed + lg / 2,
        "k_ctrl_a": 32 * mean_hops + fixed / 2 + lg / 2,
        "k_ctrl_b1": fixed / 2,
        "k_rows(provision)": 64 * mean_hops,               # link rows + per-link statistics, read and written
        "k_rows(release)": 64 * mean_hops,
        "k_ctrl_b2": lg / 2,                                # due-release detection
        "k_rel_tail": 0.0,
        # two-kernel pipeline: all per-env control in one kernel; provision + release rows in one launch
        "k_step_a2": scan + 32 * mean_hops + fixed + lg,
        "k_rows2": 128 * mean_hops,
        "k_obs": 8 * env.obs_dim if env.obs_dim else 0.0,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=1500)
    ap.add_argument("--batch", type=int, default=65536, help="envs per GPU")
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", help="torch.distributed backend for the barrier / max-over-ranks (nccl = RCCL)")
    ap.add_argument("--device", type=int, default=None, help="override the GPU index (default LOCAL_RANK)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch

    dev_index = local_rank if args.device is None else args.device
    dist = None
    backend = args.dist_backend
    if world > 1:
        import torch.distributed as dist

        torch.cuda.set_device(dev_index)
        try:
            if backend == "nccl":
                dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
            else:
                dist.init_process_group(backend)
        except Exception as exc:  # the data path needs no collective: fall back to gloo for the barrier only
            print("rank %d: %s backend unavailable (%s); using gloo for the barrier" % (rank, backend, exc), file=sys.stderr)
            backend = "gloo"
            dist.init_process_group("gloo")
    elif torch.cuda.is_available():
        torch.cuda.set_device(dev_index)

    import optical_rl_gym_amd as orl

    fam, topo, kw, policy = WORKLOADS[args.workload]
    B = args.batch
    seeds = [10 + rank * B + i for i in range(B)]  # seed_i = 10 + global env index
    env = orl.make(fam, topology=topo, num_envs=B, seeds=seeds, device_id=dev_index, **kw)

    def barrier():
        if dist is not None:
            dist.barrier()
        env.sync()
        if torch.cuda.is_available():
            torch.cuda.synchronize()

    env.run(policy, args.warmup)  # untimed: brings every env to its steady-state occupancy
    barrier()
    t0 = time.perf_counter()
    st_run = env.run(policy, args.steps)   # EXACTLY K steps of policy + step, entirely on the device
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-kernel durations of the same launches, each bracketed by HIP events on the stream it runs on (one stream,
    # whole batch), and the stand-alone slot-scan kernel (the form orl_batch_policy launches for host-driven agents)
    n_t = min(args.steps, 200)
    st = env.run(policy, n_t, time_kernels=1)
    st2 = env.run(policy, n_t, time_kernels=2)
    active = float(env.active().mean())
    processed, accepted = env.totals()
    t = env.topology
    h0 = t.path_hops[:, :, 0]
    mean_hops = float(h0[h0 > 0].mean())
    alg = algorithmic_bytes(env, mean_hops, active)
    kernels = {}
    step_names = [n for n, _ in st.kernels()]
    for name, ms in st.kernels():
        bpe = alg.get(name, 0.0)
        if name in ("k_step", "k_step8") and "k_policy" not in step_names:
            bpe += alg["k_policy"]  # the device loop of the per-env kernel runs the slot scan inside k_step
        kernels[name] = dict(ms=ms, bytes=bpe * B)
    kernels.setdefault("k_policy", dict(ms=st2.ms_policy, bytes=alg["k_policy"] * B, standalone=True))
    persistent = st_run.n_kernels == 1 and st_run.kernels()[0][0] == "k_persist"
    if persistent:
        # the timed run was ONE launch of the persistent kernel covering all K steps (live HIP-event duration of that
        # launch); the kernels above are the same work as separate launches (the form time_kernels=1 runs), kept as a breakdown
        for k in kernels.values():
            k["breakdown"] = True
        spl = args.steps / max(int(st_run.launches), 1)  # the run is cut into launches of 64 steps (the last one shorter)
        kernels["k_persist"] = dict(ms=st_run.kernels()[0][1], bytes=(alg["k_policy"] + alg["k_step"]) * B * spl,
                                    steps_per_launch=round(spl, 2))
    # HBM traffic per launch from rocprofv3 PMC passes (FETCH_SIZE x calibration + WRITE_SIZE), collected offline with
    # tools/pmc_traffic.py for exactly this workload/batch and committed under profiles/; null otherwise.
    traffic, requests = {}, {}
    tpath = os.path.join(ROOT, "profiles", "traffic_cfg2.json")
    if os.path.exists(tpath):
        tj = json.load(open(tpath))
        if tj.get("workload") == args.workload and tj.get("batch") == B:
            traffic = {k: v["hbm_bytes_per_launch"] for k, v in tj["kernels"].items()}
            requests = {k: v.get("dram_requests_per_launch") for k, v in tj["kernels"].items()}
            for k, v in tj["kernels"].items():  # kernels whose launch spans several steps: per-step figures
                if v.get("steps_per_launch"):
                    traffic[k + "_per_step"] = v["hbm_bytes_per_launch"] / v["steps_per_launch"]
                    requests[k + "_per_step"] = (v.get("dram_requests_per_launch") or 0) / v["steps_per_launch"] or None
    roof = {}
    for name, k in kernels.items():
        ach = k["bytes"] / (k["ms"] * 1e-3) / 1e9 if k["ms"] > 0 else 0.0
        roof[name] = dict(bound="hbm", achieved=round(ach, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                          frac=round(ach / HBM_PEAK_GBS, 5), traffic=traffic.get(name.split("(")[0]), us_per_launch=round(k["ms"] * 1e3, 2),
                          algorithmic_bytes_per_launch=int(k["bytes"]))
        if k.get("standalone"):
            roof[name]["note"] = "stand-alone slot-scan kernel (orl_batch_policy); the device loop runs the scan inside its step kernel"
        elif k.get("breakdown"):
            roof[name]["note"] = "separate-launch form of the same work (one stream, whole batch): breakdown only"
        if k.get("steps_per_launch"):
            roof[name]["steps_per_launch"] = k["steps_per_launch"]
            t_ps = traffic.get(name + "_per_step")
            roof[name]["traffic"] = None if t_ps is None else int(t_ps * k["steps_per_launch"])
    dominant = "k_persist" if persistent else max((n for n in kernels if not kernels[n].get("standalone")), key=lambda n: kernels[n]["ms"])
    # The bound these scattered-access kernels actually run into: L2<->fabric requests per batched step
    # (PMC, profiles/) against the random 64-byte-line access rate measured with tools/micro/gather_bench.hip
    req_roof = None
    if persistent:
        step_req = [requests.get("k_persist_per_step")]
    else:
        step_req = [requests.get(n.split("(")[0]) for n in kernels if not kernels[n].get("standalone") and n != "k_rel_tail"]
    if step_req and all(step_req):
        RANDOM_ACCESS_PEAK = 43.7e9  # read+write mix; 53.6e9 read-only (profiles/r1f_gather_bench.txt)
        per_step = float(sum(step_req))
        req_roof = dict(bound="dram_requests", requests_per_step=int(per_step), peak=RANDOM_ACCESS_PEAK, unit="requests/s")

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle.oracle import OracleBatch

        n_cpu, warm, timed = 128, 1500, 3500  # ~14 s of one host core
        ora = OracleBatch(fam, topo, seeds[:n_cpu], **kw)
        ora.run(policy, warm)
        c0 = time.perf_counter()
        ora.run(policy, timed)
        cdt = time.perf_counter() - c0
        cpu = dict(value=round(n_cpu * timed / cdt, 1), unit="env-steps/s", cores=1, kind="port",
                   sample="%d envs x %d steps after %d warm-up steps, same workload and seeds, oracle/orl_oracle.c, 1 thread"
                          % (n_cpu, timed, warm))
        try:  # the same restatement over all host cores (OpenMP over env ranges), a bounded sample too
            n_mt = 64 * (os.cpu_count() or 1)
            omt = OracleBatch(fam, topo, [10 + i for i in range(n_mt)], omp=True, **kw)
            omt.run(policy, 300)
            c0 = time.perf_counter()
            omt.run(policy, 1200)
            cpu["all_cores"] = dict(value=round(n_mt * 1200 / (time.perf_counter() - c0), 1), unit="env-steps/s",
                                    cores=os.cpu_count(), sample="%d envs x 1200 steps after 300, OpenMP" % n_mt)
        except Exception as exc:  # the OpenMP build of the oracle is optional
            cpu["all_cores"] = dict(error=str(exc))

    host = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        # SURVEY 8(d) second variant: host-supplied uniform-random actions, reward/done/info fetched every step — the
        # PCIe-inclusive rate of an agent on the host (never `value`)
        import numpy as np

        rng = np.random.RandomState(3)
        n_host = 30
        acts = [np.stack([rng.randint(0, env.k_paths + 1, B), rng.randint(0, env.num_spectrum_resources + 1, B)], 1).astype(np.int32)
                for _ in range(4)]
        env.step(acts[0], auto_reset=True)
        h0 = time.perf_counter()
        for s_ in range(n_host):
            env.step(acts[s_ % 4], auto_reset=True)
        hdt = time.perf_counter() - h0
        host = dict(value=round(B * n_host / hdt, 1), unit="env-steps/s", steps=n_host,
                    note="host-driven step(): 16 B/env of actions in, reward+done+info (%d B/env) out per step over PCIe, "
                         "synchronous; four-kernel form with info" % (8 + 1 + 8 * env.n_info))

    if rank == 0:
        total_steps = B * world * args.steps
        out = {
            "metric": "env-steps/sec RMSA-v0 NSFNET batch 65536" if args.workload == "cfg2" and B == 65536
                      else "env-steps/sec %s" % args.workload,
            "value": round(total_steps / elapsed, 1),
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64 bitmaps + f64 statistics",
            "data": "synthetic",
            "config": {"workload": "%s: %s-v0 %s, %d slots, k=%d, batch %d envs/GPU, on-device %s policy, seeds 10+i"
                                   % (args.workload, fam, topo, env.num_spectrum_resources, env.k_paths, B, policy),
                       "envs_per_gpu": B,
                       "step_kernels": ["k_persist (%d launches for the %d steps)" % (int(st_run.launches), args.steps)] if persistent else [n for n, _ in st.kernels()]},
            "roofline": dict(roof[dominant], kernel=dominant),
            "roofline_by_kernel": roof,
            "request_roofline": None if req_roof is None else dict(
                req_roof, achieved=round(req_roof["requests_per_step"] / elapsed * args.steps, 1),
                frac=round(req_roof["requests_per_step"] / elapsed * args.steps / req_roof["peak"], 4)),
            "cpu_baseline": cpu,
            "host_driven": host,
            "state": {"mean_active_services": round(active, 1),
                      "blocking": round(1.0 - accepted / max(processed, 1), 5)},
        }
        print(json.dumps(out))
    env.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
