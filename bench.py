#!/usr/bin/env python3
"""Headline benchmark: env-steps/s of RMSA-v0 on NSFNET (320 slots, k=5, load 300 Erlang), batch 65 536
envs per MI355X, on-device KSP-FF policy (BASELINE.json `metric`; SURVEY.md §8d cfg2 at B = 65 536).

One "step" = one batched policy + env.step() over the whole batch, entirely on the device: the persistent kernel
k_persist — one wavefront owns 8 envs for the whole run and alternates a control phase (slot scan + per-env control +
release detection -> work items) with a row phase (one lane per touched link row); K steps are ceil(K/128) launches, each
followed by k_stats, which replays the launch's per-env bookkeeping (counters, running averages) one lane per env.
Inputs are resident in HBM before the timed region.

What is timed, however the script is invoked:
  1. state preparation (untimed, mandatory): every env is stepped to its steady-state occupancy — max(1500, 5 x load)
     steps, plus --warmup more; the mean number of active services is then checked (cfg2: 288 +- 5 %) and reported as
     `state_before`.  --warmup adds to this preparation, it never replaces it.
  2. the timed block: EXACTLY --steps steps, bracketed by a barrier + device synchronisation on both sides, max over ranks.
     The block is repeated until >= 5 s have been timed (--min-timed-s; at least 3 blocks); `value` is the median block, `timed_region_s` the
     sum.  `roofline.achieved` comes from the HIP-event duration of the same launches.
N > 1: one process per GPU, every rank owns its own 65 536 envs (weak scaling, no collective on the data path; the barrier
and the max-over-ranks time go through a gloo group — the data path needs no RCCL).  Started under torchrun the script is one
rank; started by hand with --gpus N > 1 it launches the N ranks itself (fresh child processes, before any GPU call) and
refuses to run when fewer than N GPUs are visible.  `n_gpus` in the output is the number of ranks that ran; `per_rank` lists
each one's own rate.

    python bench.py --gpus 1 --steps 300 --warmup 0
    python bench.py --gpus 8 --steps 300              # starts 8 ranks itself
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W
"""
import argparse
import json
import math
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6.3 TB/s is what a copy achieves
RANDOM_ACCESS_PEAK = 43.7e9  # independent random 64-B-line accesses/s, read+write mix, measured on this GPU with
#                              tools/micro/gather_bench.hip (profiles/r1f_gather_bench.txt); 53.6e9 read-only

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
# steady-state mean number of active services per env (load x (1 - blocking)); the state check before timing
EXPECTED_ACTIVE = {"cfg2": 288.0}
# the reference itself (Python), measured in the build container: BASELINE.md section 2, 1 core, policy included
REFERENCE_PYTHON = {"cfg2": 196, "cfg1": 6064, "cfg3": 229, "cfg4n": 413, "cfg5": 60}


def workload_load(kw):
    if "load" in kw:
        return float(kw["load"])
    return kw["mean_service_holding_time"] / kw["mean_service_inter_arrival_time"]  # deeprmsa_env.py:25


def algorithmic_bytes(env, mean_hops, active):
    """bytes one env-step has to move.
    scan = C*E*W*8 + 24 (one read of the packed link x slot map + 16 B service descriptor + 8 B action)
    step = scan + 32*H + 128*H + 64*ceil(log2 A) + 116 + 128 + 81 (+ 8*obs_dim for DeepRMSA)"""
    C, E, S = env.num_spatial_resources, env.topology.n_links, env.num_spectrum_resources
    W = (S + 63) // 64
    scan = C * E * W * 8 + 24
    lg = 64 * math.ceil(math.log2(max(active, 2)))
    fixed = 116 + 128 + 81 + (8 * env.obs_dim if env.obs_dim else 0)
    return dict(scan=scan, step=scan + 32 * mean_hops + 128 * mean_hops + lg + fixed)


def free_port():
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def visible_gpus():
    """Number of GPUs a rank will see, counted in a throwaway child: the launcher itself must make no GPU call at all (on
    ROCm `torch.cuda.device_count()` may fall back to hipGetDeviceCount, which loads the HSA runtime and opens /dev/kfd —
    and the ranks are started from this process)."""
    import subprocess

    try:
        out = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                             capture_output=True, text=True, timeout=600)
        return int(out.stdout.strip().splitlines()[-1])
    except Exception:
        return None  # unknown: every rank checks its own device before the rendezvous


def launch_ranks(args, argv):
    """`bench.py --gpus N` started by hand (no torchrun): this process makes NO GPU call; it starts N fresh children — one
    rank per GPU, RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* set, the same command line — supervises them (the first rank
    that fails ends its siblings: a rank waiting in a gloo barrier for a dead peer would sit there for half an hour),
    forwards rank 0's JSON line and exits non-zero when any rank did."""
    import subprocess
    import tempfile

    n = args.gpus
    if args.device is None and not args.launch_check:
        have = visible_gpus()
        if have is not None and have < n:
            raise SystemExit("bench: --gpus %d but only %d GPU(s) visible; refusing to measure fewer devices than asked for "
                             "(--device D puts every rank on GPU D: a functional check, not a scaling number)" % (n, have))
    port = free_port()
    procs, outs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        outs.append(tempfile.TemporaryFile())  # (a file, not a pipe: nobody has to drain it while the ranks run)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env, stdout=outs[-1]))
    codes = [None] * n
    failed = False
    while any(c is None for c in codes):
        for r, p in enumerate(procs):
            if codes[r] is None:
                codes[r] = p.poll()
        if not failed and any(c not in (None, 0) for c in codes):
            failed = True
            for r, p in enumerate(procs):  # exactly the children started above
                if codes[r] is None:
                    p.terminate()
        time.sleep(0.05)
    outs[0].seek(0)
    sys.stdout.write(outs[0].read().decode())
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        raise SystemExit("bench: rank(s) %s failed (exit codes %s)" % ([r for r, _ in bad], [c for _, c in bad]))


def _hip_clock_khz(device):
    """hipDeviceAttributeClockRate (peak engine clock, kHz) from the HIP runtime this process already uses (the copy torch
    bundles — the same file path, so no second runtime is loaded); 0 when it cannot be asked."""
    try:
        import ctypes

        import torch

        lib = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        hip = ctypes.CDLL(lib if os.path.exists(lib) else "libamdhip64.so")
        v = ctypes.c_int(0)
        if hip.hipDeviceGetAttribute(ctypes.byref(v), 5, int(device)) == 0:  # 5 = hipDeviceAttributeClockRate
            return int(v.value)
    except Exception:
        pass
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=0, help="extra untimed steps on top of the mandatory state preparation")
    ap.add_argument("--batch", type=int, default=65536, help="envs per GPU (--scaling weak) or in total (--scaling strong)")
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"),
                    help="weak: every rank owns --batch envs (the default; BASELINE cfg5 = --gpus 8 --workload cfg5 --batch 32768); "
                         "strong: --batch envs in total, cut into contiguous blocks of env indices over the ranks "
                         "(optical_rl_gym_amd.sharding.shard_range) — the headline batch of 65 536 at 1/2/4/8 GPUs")
    ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
    ap.add_argument("--min-timed-s", type=float, default=5.0, help="repeat the timed block until this much has been timed")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--device", type=int, default=None,
                    help="put EVERY rank on this GPU index (default: rank r on GPU LOCAL_RANK); with --gpus N > 1 this is a "
                         "functional check of the N-rank path on one device, not a scaling measurement")
    ap.add_argument("--launch-check", action="store_true",
                    help="only the N-rank plumbing: ranks rendezvous (gloo), barrier, report; no GPU work (CPU test)")
    args = ap.parse_args()
    if args.steps < 1:
        ap.error("--steps must be >= 1")
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return launch_ranks(args, sys.argv[1:])
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench: --gpus %d but WORLD_SIZE=%d: launch one rank per GPU (python bench.py --gpus N starts them itself)"
                         % (args.gpus, world))
    import torch

    dev_index = local_rank if args.device is None else args.device
    if not args.launch_check:  # before the rendezvous: a rank without its device fails at once, and the launcher ends the others
        if not torch.cuda.is_available():
            raise SystemExit("bench: no GPU visible; the HIP path is the only path")
        if dev_index >= torch.cuda.device_count():
            raise SystemExit("bench: rank %d wants GPU %d but only %d GPU(s) are visible" % (rank, dev_index, torch.cuda.device_count()))
    dist = None
    if world > 1:
        import datetime

        import torch.distributed as dist

        # envs are independent: no collective on the data path, so no RCCL — a gloo group carries the barrier and the max
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=600))
    if args.launch_check:
        ranks = [None] * world
        if dist is not None:
            dist.barrier()
            dist.all_gather_object(ranks, dict(rank=rank, device=dev_index, pid=os.getpid()))
            dist.destroy_process_group()
        else:
            ranks = [dict(rank=rank, device=dev_index, pid=os.getpid())]
        if rank == 0:
            print(json.dumps({"launch_check": True, "n_gpus": len(ranks), "ranks": ranks}))
        return
    torch.cuda.set_device(dev_index)

    import optical_rl_gym_amd as orl
    from optical_rl_gym_amd import _build

    fam, topo, kw, policy = WORKLOADS[args.workload]
    from optical_rl_gym_amd.sharding import shard_range

    if args.scaling == "strong":  # the batch is the job's: rank r owns env indices [lo, hi)
        lo, hi = shard_range(args.batch, rank, world)
        if hi - lo < 8:
            raise SystemExit("bench: --scaling strong leaves rank %d with %d envs" % (rank, hi - lo))
    else:
        lo, hi = rank * args.batch, (rank + 1) * args.batch
    B = hi - lo
    seeds = [10 + i for i in range(lo, hi)]  # seed_i = 10 + global env index
    env = orl.make(fam, topology=topo, num_envs=B, seeds=seeds, device_id=dev_index, **kw)

    def device_sync():
        env.sync()
        torch.cuda.synchronize()

    def barrier():
        device_sync()
        if dist is not None:
            dist.barrier()

    # ---- 1. state preparation: untimed and mandatory ---------------------------------------------------------------
    prep_steps = max(1500, int(math.ceil(5 * workload_load(kw)))) + max(args.warmup, 0)
    env.run(policy, prep_steps)
    active_before = float(env.active().mean())
    expected = EXPECTED_ACTIVE.get(args.workload)
    if expected is not None and abs(active_before - expected) > 0.05 * expected:
        raise SystemExit("bench: state preparation did not reach the steady state (mean active services %.1f, expected %.1f)"
                         % (active_before, expected))

    # ---- 2. timed blocks of EXACTLY --steps steps ----------------------------------------------------------------
    blocks = []  # (wall seconds max over ranks, RunStats)
    own = []     # this rank's own block times (reported per rank; `value` uses the max over ranks of every block)
    timed = 0.0
    while len(blocks) < 3 or timed < args.min_timed_s:
        barrier()
        t0 = time.perf_counter()
        st_run = env.run(policy, args.steps)   # policy + step, entirely on the device
        device_sync()
        elapsed = time.perf_counter() - t0     # this rank's K steps, device idle again
        own.append(elapsed)
        if dist is not None:
            # the closing barrier, then the MAX over ranks of the per-rank times: the slowest rank's K steps.  (The clock is
            # read before the gloo barrier — a TCP round trip among N processes takes as long as several of the 20-step
            # blocks' steps and is not part of any rank's work; all ranks left the opening barrier together.)
            dist.barrier()
            t = torch.tensor([elapsed], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        blocks.append((elapsed, st_run.ms_total, int(st_run.launches), st_run.n_kernels == 1 and st_run.kernels()[0][0] == "k_persist"))
        timed += elapsed
        if len(blocks) >= 20000:
            break
    per_rank = [dict(rank=rank, device=dev_index, envs=B, env_steps_per_s=round(B * args.steps / statistics.median(own), 1))]
    if dist is not None:
        gathered = [None] * world
        dist.all_gather_object(gathered, per_rank[0])
        per_rank = gathered
    walls = sorted(b[0] for b in blocks)
    elapsed = statistics.median(walls)
    med = min(blocks, key=lambda b: abs(b[0] - elapsed))
    active_after = float(env.active().mean())
    if abs(active_after - active_before) > 0.05 * max(active_before, 1.0):
        raise SystemExit("bench: the state drifted during the timed blocks (%.1f -> %.1f active services)" % (active_before, active_after))

    # ---- roofline of the dominant kernel, live: algorithmic bytes of the steps a launch covers / HIP-event duration ---
    processed, accepted = env.totals()
    t = env.topology
    h0 = t.path_hops[:, :, 0]
    mean_hops = float(h0[h0 > 0].mean())
    alg = algorithmic_bytes(env, mean_hops, active_after)
    persistent = med[3]
    launches = max(med[2], 1) if persistent else args.steps
    ms_launch = med[1] / launches                      # HIP events on the batch's stream around the timed block
    steps_per_launch = args.steps / launches
    bytes_per_launch = alg["step"] * B * steps_per_launch
    ach = bytes_per_launch / (ms_launch * 1e-3) / 1e9
    kernel = "k_persist" if persistent else "k_step"
    # HBM traffic and L2<->fabric requests from rocprofv3 PMC passes (tools/pmc_traffic.py, FETCH_SIZE x calibration +
    # WRITE_SIZE; TCC_EA0_RDREQ + WRREQ): used only when they were collected for exactly this build, workload, batch and
    # state — otherwise null
    # which instantiation of the persistent kernel the timed launches were (debug query of the library)
    kernel_form = None
    if persistent:
        kernel_form = {0: "generic", 1: "specialised for this configuration (one wavefront per 8 envs)",
                       2: "specialised for this configuration, two-wavefront form (a control and a row wavefront per 8 envs: batches of at "
                          "most 12 288 envs; per-wavefront-step counter figures are per 8-env workgroup)"}.get(
                              int(env.lib.orl_batch_debug_persist_spec(env._h)))
    traffic = req_roof = valu = traffic_gbs = traffic_scaled = None
    tpath = os.path.join(ROOT, "profiles", "traffic_%s_%d.json" % (args.workload, B))  # (a batch other than the BASELINE size)
    if not os.path.exists(tpath):
        tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.workload)
    if os.path.exists(tpath):
        tj = json.load(open(tpath))
        # counters taken over launches of the SAME length as the timed ones where there are such (cfg2: ten 20-step launches for
        # the driver's invocation), else over 128-step launches, scaled per step (then `traffic_scaled_from_steps` says so: the
        # window fill / write-back of a launch is amortised over 128 steps there)
        k = tj.get("kernels", {}).get("%s_steps%d" % (kernel, int(round(args.steps / max(med[2], 1)))) if med[3] else kernel) \
            or tj.get("kernels", {}).get(kernel)
        same_state = abs(tj.get("mean_active_services", -1e9) - active_after) <= 0.05 * active_after
        if k and tj.get("workload") == args.workload and tj.get("batch") == B and same_state and \
                tj.get("source_hash") == _build.source_hash(with_compiler=False):
            per_step = k["hbm_bytes_per_launch"] / k["steps_per_launch"]
            traffic = int(per_step * steps_per_launch)
            traffic_scaled = None if abs(k["steps_per_launch"] - steps_per_launch) < 0.5 else k["steps_per_launch"]
            traffic_gbs = traffic / (ms_launch * 1e-3) / 1e9  # bytes that really crossed the HBM interface / launch time
            sq = k.get("sq_per_launch") or {}
            if sq.get("SQ_ACTIVE_INST_VALU"):
                # issue side of the same launches: a wave64 VALU instruction occupies its SIMD for 4 cycles
                n_simd = 4 * torch.cuda.get_device_properties(dev_index).multi_processor_count
                # the engine clock the runtime reports for this device (hipDeviceProp_t::clockRate, kHz); the MI355X peak of
                # MI355X_MICROARCH.md only where the property is missing
                clock_khz = getattr(torch.cuda.get_device_properties(dev_index), "clock_rate", 0) or _hip_clock_khz(dev_index)
                clock_hz = clock_khz * 1e3 if clock_khz > 0 else 2.4e9
                clock_src = "hipDeviceProp_t.clockRate" if clock_khz > 0 else "assumed (MI355X peak engine clock)"
                per_step_c = {c: v / k["steps_per_launch"] for c, v in sq.items()}
                cycles = (elapsed / args.steps) * clock_hz
                busy = per_step_c["SQ_ACTIVE_INST_VALU"] * 4.0 / (cycles * n_simd)
                valu = dict(bound="valu_issue", frac=round(busy, 4), simds=n_simd, clock_ghz=clock_hz / 1e9, clock_source=clock_src,
                            valu_insts_per_wavefront_step=round(per_step_c.get("SQ_INSTS_VALU", 0) / ((B + 7) // 8), 1),
                            salu_insts_per_wavefront_step=round(per_step_c.get("SQ_INSTS_SALU", 0) / ((B + 7) // 8), 1),
                            wait_frac_of_wave_lifetime=round(sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"], 4) if sq.get("SQ_WAVE_CYCLES") else None,
                            lds_bank_conflict_cycles_per_lds_active_cycle=round(sq["SQ_LDS_BANK_CONFLICT"] / sq["SQ_ACTIVE_INST_LDS"], 3)
                            if sq.get("SQ_ACTIVE_INST_LDS") and sq.get("SQ_LDS_BANK_CONFLICT") is not None else None,
                            note="SQ_ACTIVE_INST_VALU x 4 / (cycles of one batched step x SIMDs), counters from profiles/traffic_%s.json "
                                 "(same build, workload, batch, state)" % args.workload)
            if k.get("dram_requests_per_launch"):
                rps = k["dram_requests_per_launch"] / k["steps_per_launch"]
                rate = rps / (elapsed / args.steps)
                req_roof = dict(bound="dram_requests", requests_per_step=int(rps), peak=RANDOM_ACCESS_PEAK, unit="requests/s",
                                achieved=round(rate, 1), frac=round(rate / RANDOM_ACCESS_PEAK, 4))
    # SURVEY 8(d) "report both": beside the 8 TB/s spec peak, what this GPU's HBM delivers to the library's own plain streaming
    # kernels, measured now (orl_debug_stream_peak: 1 GiB, 16 bytes per lane, HIP events, best of 10): a read — the kernel the
    # counter calibration uses — and a copy (read + write bytes).  `peak_measured` is the higher of the two.
    peak_meas = peak_read = peak_copy = None
    try:
        import ctypes as C_

        rd, cp = C_.c_double(), C_.c_double()
        if env.lib.orl_debug_stream_peak(dev_index, 1 << 30, 10, C_.byref(rd), C_.byref(cp)) == 0 and rd.value > 0:
            peak_read, peak_copy = rd.value, cp.value
            peak_meas = max(peak_read, peak_copy)
    except Exception:  # (a GPU too full for the two buffers: the spec peak alone)
        peak_meas = None
    roofline = dict(bound="hbm", achieved=round(ach, 2), peak=HBM_PEAK_GBS, unit="GB/s", frac=round(ach / HBM_PEAK_GBS, 5),
                    peak_measured=None if peak_meas is None else round(peak_meas, 1),
                    peak_measured_read=None if peak_read is None else round(peak_read, 1),
                    peak_measured_copy=None if peak_copy is None else round(peak_copy, 1),
                    frac_of_measured=None if peak_meas is None else round(ach / peak_meas, 5),
                    # (the library's own streaming kernels stay 15-20 % below the guide's float4 copy: the fraction of THAT figure too)
                    peak_guide_copy=6290.0, frac_of_guide_copy=round(ach / 6290.0, 5),
                    traffic=traffic, achieved_traffic_gbs=None if traffic_gbs is None else round(traffic_gbs, 2),
                    traffic_frac=None if traffic_gbs is None else round(traffic_gbs / HBM_PEAK_GBS, 5),
                    traffic_scaled_from_steps=traffic_scaled, valu=valu,
                    kernel=kernel, us_per_launch=round(ms_launch * 1e3, 2),
                    steps_per_launch=round(steps_per_launch, 2), algorithmic_bytes_per_env_step=round(alg["step"], 1),
                    algorithmic_bytes_per_launch=int(bytes_per_launch))

    # the stand-alone slot-scan kernel (what orl_batch_policy launches for a host-side agent), on the same steady-state
    # slot maps.  The production loop above does NOT launch it: there the scan is the first phase of k_persist.
    n_t = 50
    st2 = env.run(policy, n_t, time_kernels=2)
    scan_ach = alg["scan"] * B / (st2.ms_policy * 1e-3) / 1e9
    # ... and on the bytes the kernel's lanes really request: the rows of the k paths of the pending pair (links shared by
    # several paths are requested once per path), the 32-byte path records, the descriptor in and the action out
    hops_sum = t.path_hops.reshape(t.n_nodes, t.n_nodes, -1).sum(-1)
    np_pair = t.n_paths.reshape(t.n_nodes, t.n_nodes)
    off = ~np.eye(t.n_nodes, dtype=bool)
    W_ = (env.num_spectrum_resources + 63) // 64
    W_ = 1 if W_ <= 1 else (2 if W_ <= 2 else (5 if W_ <= 5 else 8))
    req_env = float(hops_sum[off].mean()) * W_ * 8 * env.num_spatial_resources + 32.0 * float(np_pair[off].mean()) + 8 + 16
    scan_req = req_env * B / (st2.ms_policy * 1e-3) / 1e9
    slot_scan = dict(kernel="k_policy", bound="hbm", achieved=round(scan_ach, 2), peak=HBM_PEAK_GBS, unit="GB/s",
                     frac=round(scan_ach / HBM_PEAK_GBS, 5), us_per_launch=round(st2.ms_policy * 1e3, 2),
                     algorithmic_bytes_per_launch=int(alg["scan"] * B),
                     requested_bytes_per_env=round(req_env, 1), requested_gbs=round(scan_req, 2),
                     requested_frac=round(scan_req / HBM_PEAK_GBS, 5),
                     note="stand-alone launch (host-side agents); not part of the timed loop, whose scan runs inside " + kernel)

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle.oracle import OracleBatch

        n_cpu, warm, timed_c = 128, 1500, 3500  # ~14 s of one host core
        ora = OracleBatch(fam, topo, seeds[:n_cpu], **kw)
        ora.run(policy, warm)
        c0 = time.perf_counter()
        ora.run(policy, timed_c)
        cdt = time.perf_counter() - c0
        cpu = dict(value=round(n_cpu * timed_c / cdt, 1), unit="env-steps/s", cores=1, kind="port",
                   sample="%d envs x %d steps after %d warm-up steps, same workload and seeds, oracle/orl_oracle.c, 1 thread"
                          % (n_cpu, timed_c, warm))
        try:  # the same restatement over all host cores (OpenMP over env ranges), a bounded sample too
            n_mt = 64 * (os.cpu_count() or 1)
            omt = OracleBatch(fam, topo, [10 + i for i in range(n_mt)], omp=True, **kw)
            omt.run(policy, 300)
            c0 = time.perf_counter()
            omt.run(policy, 1200)
            cpu["all_cores"] = dict(value=round(n_mt * 1200 / (time.perf_counter() - c0), 1), unit="env-steps/s",
                                    cores=os.cpu_count(), sample="%d envs x 1200 steps after 300, OpenMP" % n_mt,
                                    note="OpenMP over env ranges on every hardware thread of the host; the restatement is a chain of "
                                         "dependent cache misses per env and scales ~10x on 128 cores x 2 threads: a weak multi-core "
                                         "baseline, reported for what it is")
        except Exception as exc:  # the OpenMP build of the oracle is optional
            cpu["all_cores"] = dict(error=str(exc))
        if args.workload in REFERENCE_PYTHON:
            cpu["reference_python"] = dict(value=REFERENCE_PYTHON[args.workload], unit="env-steps/s", cores=1,
                                           hw="Xeon Ice Lake 2.6 GHz (build container), Python 3.10.12, numpy 2.2.6",
                                           source="BASELINE.md section 2: the reference's own step() + heuristic, imported and "
                                                  "timed in the build container; it cannot travel to the GPU box")

    host = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and fam in ("RMSA", "RWA"):
        # SURVEY 8(d) second variant: host-supplied uniform-random actions, reward/done/info fetched every step — the
        # PCIe-inclusive rate of an agent on the host (never `value`)
        rng = np.random.RandomState(3)
        n_host = 30
        rej = 1 if kw.get("allow_rejection") else 0
        hi_p, hi_s = (env.k_paths + 1, env.num_spectrum_resources + 1) if fam == "RMSA" else (env.k_paths + rej, env.num_spectrum_resources + rej)
        acts = [np.stack([rng.randint(0, hi_p, B), rng.randint(0, hi_s, B)], 1).astype(np.int32) for _ in range(4)]
        env.step(acts[0], auto_reset=True)
        h0_ = time.perf_counter()
        for s_ in range(n_host):
            env.step(acts[s_ % 4], auto_reset=True)
        hdt = time.perf_counter() - h0_
        host = dict(value=round(B * n_host / hdt, 1), unit="env-steps/s", steps=n_host,
                    note="host-driven step(): 16 B/env of actions in, reward+done+info (%d B/env) out per step over PCIe, "
                         "synchronous, uniform-random actions (most are rejected); k_agent (RMSA / DeepRMSA from 2 048 envs) "
                         "or the one-wavefront-per-env kernel" % (8 + 1 + 8 * env.n_info))
        # the loop an agent on the same GPU drives: actions stay in device memory, nothing is fetched, one sync at the end.
        # The representative figure applies FRESH actions every step — here a heuristic's, with its slot scan as the first phase of
        # the step kernel (orl_batch_policy_step: one launch per step); re-issuing one stale action set (most of it rejected from
        # the second step on: no provision to apply) is the step kernel's floor, reported beside it.
        n_loop = 100

        def loop(call):
            env.policy(policy, fetch=False)
            env.sync()
            l0 = time.perf_counter()
            for _ in range(n_loop):
                call()
            env.sync()
            return time.perf_counter() - l0

        ldt = loop(lambda: env.policy_step(policy, auto_reset=True, fetch=False))
        sdt = loop(lambda: env.step(None, auto_reset=True, fetch=False))
        host["agent_loop_zero_copy"] = dict(value=round(B * n_loop / ldt, 1), unit="env-steps/s", us_per_step=round(ldt / n_loop * 1e6, 2),
                                            note="policy_step(%s, auto_reset=True, fetch=False) x %d: fresh actions every step from the "
                                                 "on-device heuristic, scan + step in ONE launch of k_agent per step, info / reward / done "
                                                 "written in place" % (policy, n_loop),
                                            stale_actions_floor=dict(value=round(B * n_loop / sdt, 1), us_per_step=round(sdt / n_loop * 1e6, 2),
                                                                     note="step(None, ...) x %d on one stale action set (mostly rejected)" % n_loop))

    if rank == 0:
        total_envs = sum(p["envs"] for p in per_rank)
        total_steps = total_envs * args.steps
        out = {
            "metric": "env-steps/sec RMSA-v0 NSFNET batch 65536" if args.workload == "cfg2" and args.batch == 65536
                      else "env-steps/sec %s" % args.workload,
            "value": round(total_steps / elapsed, 1),
            "unit": "env-steps/s",
            "n_gpus": len(per_rank),  # the ranks that ran and reported
            "per_rank": per_rank,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "timed_region_s": round(timed, 4),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "u64 bitmaps + f64 statistics",
            "data": "synthetic",
            "config": {"workload": "%s: %s-v0 %s, %d slots, k=%d, batch %s, on-device %s policy, seeds 10+i"
                                   % (args.workload, fam, topo, env.num_spectrum_resources, env.k_paths,
                                      "%d envs/GPU" % B if args.scaling == "weak" else "%d envs over %d GPU(s)" % (total_envs, len(per_rank)), policy),
                       "envs_per_gpu": B if args.scaling == "weak" else [p["envs"] for p in per_rank],
                       "envs_total": total_envs,
                       "step_kernels": ["%s (%d launches per %d-step block)" % (kernel, launches, args.steps)] +
                                       (["k_stats (the launch's deferred bookkeeping, one lane per env, behind every k_persist launch)"]
                                        if persistent else []),
                       "kernel_form": kernel_form},
            "timing": {"blocks": len(blocks), "timed_region_s": round(timed, 4), "block_s_median": round(elapsed, 6),
                       "block_s_min": round(walls[0], 6), "block_s_max": round(walls[-1], 6),
                       "state_preparation_steps": prep_steps},
            "state_before": {"mean_active_services": round(active_before, 1)},
            "state": {"mean_active_services": round(active_after, 1),
                      "blocking": round(1.0 - accepted / max(processed, 1), 5)},
            "roofline": roofline,
            "slot_scan_standalone": slot_scan,
            "request_roofline": req_roof,
            "cpu_baseline": cpu,
            "host_driven": host,
        }
        print(json.dumps(out))
    env.close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
