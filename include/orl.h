/*
 * orl.h — C ABI of liborlgpu.so: batched optical-network RL environments on MI355X (gfx950).
 *
 * This is the drop-in boundary for ONE hot path of optical-rl-gym: the per-env
 * reset()/step()/observation()/heuristic loop of RWAEnv / RMSAEnv / DeepRMSAEnv / RMCSAEnv.
 * The reference is pure Python (no FFI of its own); each entry point below names the
 * reference interface it stands in for (file:line relative to the reference repo).  The
 * ctypes binding a maintainer would add is shown in INTEGRATION.md and implemented in
 * optical_rl_gym_amd/_lib.py.
 *
 * Conventions
 *   - every function returns 0 on success or a negative ORL_E_* code; orl_last_error() gives text
 *   - nothing throws across the ABI; no torch / HIP types appear in any signature
 *   - host buffers are caller-owned, C-contiguous, valid only for the duration of the call
 *   - device memory is owned by the handles; one batch lives on ONE GPU (multi-GPU = one process and
 *     one batch per GPU; envs are independent, there is no collective)
 *   - a handle is not thread-safe: one host thread drives it (the reference env is single-threaded too)
 *   - `n_envs` independent envs; env i behaves exactly like a reference env constructed with seed_i
 */
#ifndef ORL_H
#define ORL_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: orl_env_config starts with struct_size and carries the QoSConstrainedRA fields; ORL_E_INTERNAL; orl_multi_* (round 3).
 * A client compiled against another layout is refused by orl_batch_create (struct_size) and by the loader (version). */
#define ORL_ABI_VERSION 2

#define ORL_OK 0
#define ORL_E_INVALID (-1)   /* bad argument / unsupported configuration */
#define ORL_E_HIP (-2)       /* HIP runtime error (no GPU, out of memory, launch failure) */
#define ORL_E_ACTION (-3)    /* an action was out of the action space (reference: IndexError, rmsa_env.py:167) */
#define ORL_E_OVERFLOW (-4)  /* an env exceeded its pending-release capacity */
#define ORL_E_INTERNAL (-5)  /* a C++ exception (std::bad_alloc, ...) was stopped at the boundary: nothing throws across the ABI */

/* env families (optical_rl_gym/__init__.py:3-26 registry ids) */
#define ORL_ENV_RMSA 0      /* "RMSA-v0"      optical_rl_gym/envs/rmsa_env.py:18 */
#define ORL_ENV_DEEPRMSA 1  /* "DeepRMSA-v0"  optical_rl_gym/envs/deeprmsa_env.py:9 */
#define ORL_ENV_RWA 2       /* "RWA-v0"       optical_rl_gym/envs/rwa_env.py:15 */
#define ORL_ENV_RMCSA 3     /* "RMCSA-v0"     optical_rl_gym/envs/rmcsa_env.py:18 */
#define ORL_ENV_QOS 4       /* "QoSConstrainedRA-v0"  optical_rl_gym/envs/qos_constrained_ra.py:13 (its constructor raises
                             * upstream, :32-41; semantics as the class is written, see oracle/gen_golden_qos.py).  Actions: the
                             * path index (Discrete(k + reject)); policies SP_FF / SAP_FF / LLP_FF = shortest_path /
                             * shortest_available_path / least_loaded_path (:408-450); info: the two service blocking rates. */

/* on-device heuristics.  RMSA: rmsa_env.py:747-803; DeepRMSA: deeprmsa_env.py:135-155 (SP=0, SAP=1);
 * RWA: rwa_env.py:425-502; RMCSA: rmcsa_env.py:882-911 (id 1). */
#define ORL_POLICY_SP_FF 0
#define ORL_POLICY_SAP_FF 1 /* k-shortest-path first-fit (KSP-FF) */
#define ORL_POLICY_LLP_FF 2
#define ORL_POLICY_SAP_LF 3 /* RWA only */
/* PathOnlyFirstFitAction (rmsa_env.py:840-874, rwa_env.py:505-536): the agent chose the path (Discrete(k + reject)), the
 * device finds the first fitting slot on it.  The per-env path column is set with orl_batch_set_paths() or written into
 * the ORL_BUF_PATHS device array.  RMSA and RWA. */
#define ORL_POLICY_PATH_FF 4

/* Flattened topology: what the reference keeps in topology.graph["ksp"] / ["modulations"] and in the
 * networkx edge attributes (examples/create_topology.py:96-147).  All arrays are copied. */
typedef struct {
  int32_t n_nodes, n_links, k_paths, max_hops, n_modulations;
  const int32_t* n_paths;         /* [n_nodes*n_nodes]            paths available for (src,dst) */
  const int32_t* path_hops;       /* [n_nodes*n_nodes*k]          Path.hops */
  const int32_t* path_links;      /* [n_nodes*n_nodes*k*max_hops] edge "index" per hop, -1 padded */
  const double* path_length;      /* [n_nodes*n_nodes*k]          Path.length (km) */
  const int32_t* path_modulation; /* [n_nodes*n_nodes*k]          index of the modulation the heuristics use */
  const int32_t* edge_iter_order; /* [n_links]                    link index of the i-th edge of topology.edges() */
} orl_topology_desc;

/* Traffic + environment parameters: the reference constructors' kwargs after their own derivations
 * (optical_network_env.py:14-94, rmsa_env.py:29-161, deeprmsa_env.py:10-46, rwa_env.py:19-94,
 * rmcsa_env.py:29-207).  Float tables are computed on the host with the reference's own expressions. */
typedef struct {
  uint32_t struct_size;        /* sizeof(orl_env_config) as the caller compiled it; checked by orl_batch_create */
  int32_t env_type;            /* ORL_ENV_* */
  int32_t num_spectrum_resources;
  int32_t num_spatial_resources; /* cores; 1 unless RMCSA */
  int32_t episode_length;
  int32_t allow_rejection;
  int32_t j;                   /* DeepRMSA: blocks per path */
  int32_t bit_rate_mode;       /* 0 = continuous randint(lo, hi), 1 = discrete choices(bit_rates, probs) */
  int32_t bit_rate_lo, bit_rate_hi;
  int32_t n_bit_rates;         /* rows of the bit-rate tables: hi-lo+1 (continuous) or len(bit_rates) */
  int32_t event_capacity;      /* pending releases per env (0 = derive from load) */
  int32_t action_histograms;   /* != 0: keep actions_output / actions_taken per env: [2][k+1][S+1] (rmsa_env.py:126-137,
                                * rwa_env.py:52-58); RMCSA [2][k+1][M+1][C+1][S+1] (rmcsa_env.py:145-180; 863 KB per env at 7 x 320) */
  double lambda_arrival;       /* 1 / mean_service_inter_arrival_time  (rmsa_env.py:548-550) */
  double lambda_holding;       /* 1 / mean_service_holding_time        (rmsa_env.py:553) */
  const double* cum_src;       /* [n_nodes]          accumulate(node_request_probabilities) */
  const double* cum_dst;       /* [n_nodes*n_nodes]  per src: accumulate(probs with src zeroed, renormalised) */
  const int32_t* bit_rates;    /* [n_bit_rates]      bit rate of table row i */
  const double* cum_bit_rate;  /* [n_bit_rates]      discrete mode: accumulate(bit_rate_probabilities) */
  const uint8_t* n_slots;      /* [n_bit_rates*n_modulations] get_number_slots (rmsa_env.py:610-621) */
  const double* lmax_snr;      /* [n_modulations*n_bit_rates] RMCSA reach limit, eq.(1) (rmcsa_env.py:366-375); may be NULL */
  const double* lmax_xt;       /* [n_modulations]             RMCSA reach limit, eq.(2) (rmcsa_env.py:377-379); may be NULL */
  /* QoSConstrainedRA only (qos_constrained_ra.py:21-23); ignored (may be 0 / NULL) for the other families */
  int32_t n_service_classes;   /* num_service_classes */
  int32_t reserved;
  const double* cum_class;     /* [n_service_classes] accumulate(classes_arrival_probabilities) */
  const double* class_reward;  /* [n_service_classes] classes_reward */
} orl_env_config;

typedef struct orl_topology orl_topology;
typedef struct orl_batch orl_batch;

/* per-env integer counters, in this order (optical_network_env.py:29-34, rmsa_env.py:73-76) */
#define ORL_N_COUNTERS 8
/* current service record: arrival_time, holding_time, source_id, destination_id, bit_rate, service_id */
#define ORL_N_SERVICE 6

#define ORL_MAX_STEP_KERNELS 12
typedef struct {
  double ms_total;   /* wall time of the whole call on the device (HIP events) */
  double ms_policy;  /* time_kernels == 2: average duration of one stand-alone slot-scan (policy) launch */
  double ms_step;    /* time_kernels == 2: average duration of the launches of one step() */
  int64_t launches;  /* kernel launches issued */
  /* time_kernels == 0 with the persistent kernel: n_kernels = 1, kernel_name[0] = "k_persist", launches = number of its
   * launches (chunks of 128 steps), ms_kernel[0] = their average duration (ms_total / launches).
   * time_kernels == 1: the kernels one policy+step of the device loop launches, in launch order, each bracketed by HIP
   * events on the stream it runs on; average duration per launch */
  int32_t n_kernels;
  int32_t reserved;
  double ms_kernel[ORL_MAX_STEP_KERNELS];
  char kernel_name[ORL_MAX_STEP_KERNELS][32];
} orl_run_stats;

int orl_abi_version(void);
const char* orl_last_error(void);
int orl_device_count(void);

int orl_topology_create(const orl_topology_desc* desc, int device_id, orl_topology** out);
void orl_topology_destroy(orl_topology* t);

/* n_envs envs on device `device_id`.  mt_state: [n_envs][625] uint32 = random.Random(seed_i).getstate()[1]
 * (624 state words + index), i.e. optical_network_env.py:205-210 done by the caller.  Construction ends with
 * the reference's reset(only_episode_counters=False) (rmsa_env.py:160-161): every env holds its first service. */
int orl_batch_create(const orl_env_config* cfg, const orl_topology* topo, int64_t n_envs, const uint32_t* mt_state,
                     orl_batch** out);
/* Same, but the device runs random.Random(seed_i) itself (CPython random_seed -> init_by_array on abs(seed)). */
int orl_batch_create_seeded(const orl_env_config* cfg, const orl_topology* topo, int64_t n_envs, const int64_t* seeds,
                            orl_batch** out);
void orl_batch_destroy(orl_batch* b);

int orl_batch_info_dim(const orl_batch* b); /* floats per env in the info row */
int orl_batch_obs_dim(const orl_batch* b);  /* DeepRMSA observation length, else 0 */

/* reset(only_episode_counters = !full) (rmsa_env.py:284-359, rwa_env.py:164-208, rmcsa_env.py:386-483).
 * env_mask: NULL = all envs, else [n_envs] bytes. */
int orl_batch_reset(orl_batch* b, int full, const uint8_t* env_mask);

/* seed(seed) (optical_network_env.py:205-210): env i (of the envs selected by env_mask, NULL = all) continues with the
 * random stream of random.Random(seeds[i]); nothing else of its state changes.  seeds: [n_envs] (entries of unselected
 * envs are ignored). */
int orl_batch_reseed(orl_batch* b, const int64_t* seeds, const uint8_t* env_mask);

/* evaluate_heuristic on the device (utils.py:103-141: reset, loop until done, sum the rewards, n episodes).  Arm a per-env
 * log of `capacity` episodes (0 = disarm); from then on every env that finishes an episode appends its
 * episode_services_accepted — the episode's reward sum for RMSA / RWA / RMCSA (reward 1 per accepted service), and
 * 2 * accepted - steps for DeepRMSA (reward +1 / -1) — before the auto (soft) reset.  Read it back with
 * orl_batch_get_episode_log: counts[n_envs], accepted[n_envs][capacity]. */
int orl_batch_episode_log(orl_batch* b, int32_t capacity);
int orl_batch_get_episode_log(orl_batch* b, int32_t* counts, int32_t* accepted);
/* QoSConstrainedRA (reward = the accepted service's class reward, qos_constrained_ra.py:131-136): the harness's episode_reward of
 * every logged episode — the float64 sum of the step rewards in step order, from 0.0 (utils.py:125-131) — [n_envs][capacity]. */
int orl_batch_get_episode_rewards(orl_batch* b, double* rewards);

/* ORL_POLICY_PATH_FF: the path index chosen for every env, [n_envs] int32 (>= k_paths = reject). */
int orl_batch_set_paths(orl_batch* b, const int32_t* paths);

/* Heuristic decision for the pending service of every env.  actions_out: [n_envs][4] int32 or NULL to keep the
 * result on the device for the next orl_batch_step(actions = NULL).
 * Columns: RMSA/RWA (path, slot, -, -); DeepRMSA (action, -, -, -); RMCSA (path, modulation, core, slot). */
int orl_batch_policy(orl_batch* b, int policy_id, int32_t* actions_out);

/* step() for every env (rmsa_env.py:163-282, deeprmsa_env.py:48-58, rwa_env.py:101-162, rmcsa_env.py:209-339).
 * actions: [n_envs][4] int32, or NULL = use the device-resident result of the last orl_batch_policy().
 * auto_reset != 0: an env that returns done is soft-reset right away (what SB3's VecEnv does).
 * Any output pointer may be NULL (result stays on the device).  reward_out/[n_envs] double, done_out/[n_envs] u8,
 * info_out/[n_envs][info_dim] double, obs_out/[n_envs][obs_dim] double (DeepRMSA).
 * Synchronous on return when any output pointer is given.
 * Errors, mirroring the reference's exceptions:
 *   ORL_E_ACTION    host-supplied `actions` hold an index outside the reference's actions_output array (IndexError at
 *                   rmsa_env.py:167, rwa_env.py:103, rmcsa_env.py:219): returned BEFORE anything is modified.  With
 *                   device-resident actions (actions = NULL) the kernel treats such an action as a rejection and the
 *                   error is reported by the next synchronous call (or orl_batch_check).
 *   ORL_E_OVERFLOW  an env needed more than event_capacity pending releases: that env's state is invalid from then on
 *                   (the reference's heap is unbounded); recreate the batch with a larger event_capacity. */
int orl_batch_step(orl_batch* b, const int32_t* actions, int auto_reset, double* obs_out, double* reward_out,
                   uint8_t* done_out, double* info_out);

/* step() in two halves, for a caller that has work of its own between issuing a step and needing its results (SB3's
 * VecEnv.step_async / step_wait, examples/stable_baselines3/DeepRMSA.ipynb:272-302 drives the env through them).
 * actions: [n_envs][action_width] (1..4 columns: the family's action, in the column order of orl_batch_step) of int32 or int64
 * (action_elem_bytes 4 / 8), C-contiguous — the array an agent hands to VecEnv.step as it is — or NULL for the device-resident
 * ones; checked and widened in one pass into the library's own page-locked staging rows.
 * orl_batch_step_async: the checks of orl_batch_step (ORL_E_ACTION before anything is modified), then everything is QUEUED on the
 * batch's stream — actions in, the step kernel, the requested results out — and the call returns.  The output buffers (any may
 * be NULL; obs_f32_out receives the observation cast to float32 on the device) must stay valid and untouched until
 * orl_batch_step_wait returns; give page-locked ones (orl_host_alloc) or the copies are staged synchronously.
 * orl_batch_step_wait: waits for that step and reports what orl_batch_step would have (ORL_E_OVERFLOW, flagged device-resident
 * actions).  One step may be pending per batch; any other call on the batch in between is ordered behind it by the stream. */
int orl_batch_step_async(orl_batch* b, const void* actions, int action_width, int action_elem_bytes, int auto_reset,
                         double* obs_out, float* obs_f32_out, double* reward_out, uint8_t* done_out, double* info_out);
int orl_batch_step_wait(orl_batch* b);

/* A heuristic's decision and the step on it in one call: what orl_batch_policy(b, policy_id, ...) followed by
 * orl_batch_step(b, NULL, auto_reset, ...) do — the heuristic of rmsa_env.py:747-803 / deeprmsa_env.py:135-155 / rwa_env.py:403-502 /
 * rmcsa_env.py:882-911, then step() — with the slot scan as the first phase of the step kernel where the 8-lanes-per-env step
 * kernel serves the batch (ONE launch per step instead of two; k_paths <= 8), else as the two launches.  For agents on the same
 * GPU that imitate, warm-start from or are compared with a heuristic.  actions_out: [n_envs][4] int32 or NULL (the actions also
 * stay in ORL_BUF_ACTIONS); the other outputs and the error behaviour as orl_batch_step with device-resident actions.  With no
 * output buffer the call only queues work on the batch's stream. */
int orl_batch_policy_step(orl_batch* b, int policy_id, int auto_reset, int32_t* actions_out, double* obs_out, double* reward_out,
                          uint8_t* done_out, double* info_out);

/* Which info entries the 8-lanes-per-env step kernel writes per step (RMSA / DeepRMSA; rmsa_env.py:228-264): 0 (default) = all of
 * them; 1 = the blocking rates only (service / episode_service / bit_rate / episode_bit_rate blocking and the per-rate entries of
 * the discrete mode) — network_compactness, network_compactness_difference, avg_link_compactness and avg_link_utilization, the
 * last two a read of every link record of every env per step, keep whatever an earlier step wrote.  For consumers that read the
 * rates only (SB3's Monitor with the reference's info_keywords, examples/stable_baselines3/DeepRMSA.ipynb:279-288).  The
 * one-wavefront-per-env kernel always writes everything. */
int orl_batch_set_info_mode(orl_batch* b, int mode);

/* DeepRMSAEnv.observation() for the pending service (deeprmsa_env.py:60-121). */
int orl_batch_observation(orl_batch* b, double* obs_out);
/* The observation array the last orl_batch_step / orl_batch_observation / orl_batch_reset left on the device (ORL_BUF_OBS), cast
 * to float32 ON THE DEVICE and copied out — half the PCIe bytes and no conversion pass on the host for agents that work in
 * float32 (SB3's default); the float64 values stay the parity-checked ones.  obs_out: [n_envs][obs_dim] float. */
int orl_batch_get_obs_f32(orl_batch* b, float* obs_out);
/* Rows env_index[0..n) of the info array the last orl_batch_step left on the device (ORL_BUF_INFO), gathered on the device and
 * copied out: info_out [n][info_dim] double.  What a VecEnv needs of info is the rows of the envs that just finished an episode
 * (SB3 reads `info["episode"]` there and nothing elsewhere, DeepRMSA.ipynb:272-302 with Monitor's info_keywords): a handful of
 * rows instead of the whole [n_envs][info_dim] array (4 MB per step at 65 536 RMSA envs) over PCIe.  Synchronous. */
int orl_batch_get_info_rows(orl_batch* b, const int64_t* env_index, int64_t n, double* info_out);

/* n_steps x { policy ; step(auto_reset) } entirely on the device (the loop of utils.evaluate_heuristic,
 * utils.py:113-128, with VecEnv-style auto reset).  time_kernels: 0 = production run (the persistent kernel where it
 * applies, else the per-env kernel with the slot scan inside), 1 = the separate-launch form with every kernel bracketed by
 * HIP events (stats->ms_kernel), 2 = stand-alone slot-scan kernel + per-env step launches (ms_policy, ms_step).
 * Returns ORL_E_OVERFLOW / ORL_E_ACTION like orl_batch_step. */
int orl_batch_run(orl_batch* b, int policy_id, int64_t n_steps, int time_kernels, orl_run_stats* stats);

int orl_batch_sync(orl_batch* b);
/* Synchronises and returns ORL_E_OVERFLOW / ORL_E_ACTION if any env flagged it since the last report (see orl_batch_step). */
int orl_batch_check(orl_batch* b);

/* Page-locked host memory for the buffers handed to orl_batch_step / orl_batch_policy: copies to and from it run at the
 * full PCIe rate (pageable numpy memory is staged through a bounce buffer by the runtime). */
int orl_host_alloc(size_t bytes, void** out);
int orl_host_free(void* p);

/* Zero-copy access for an agent that lives on the same GPU (the reference hands numpy arrays to SB3,
 * DeepRMSA.ipynb:272-302).  Device pointers of the batch's I/O arrays: write actions there, call
 * orl_batch_step(b, NULL, auto_reset, NULL, NULL, NULL, NULL) (no copies, no synchronisation: the launches are queued
 * on the batch's stream), orl_batch_sync(b), read reward / done / info / obs in place.  The pointers stay valid until
 * orl_batch_destroy.  `which`: */
#define ORL_BUF_ACTIONS 0  /* int32 [n_envs][4] */
#define ORL_BUF_REWARD 1   /* f64   [n_envs] */
#define ORL_BUF_DONE 2     /* u8    [n_envs] */
#define ORL_BUF_INFO 3     /* f64   [n_envs][info_dim] */
#define ORL_BUF_OBS 4      /* f64   [n_envs][obs_dim] (DeepRMSA) */
#define ORL_BUF_TERM_OBS 5 /* f64   [n_envs][obs_dim]: observation before an auto reset */
#define ORL_BUF_PATHS 6    /* int32 [n_envs]: path column of ORL_POLICY_PATH_FF */
int orl_batch_device_buffer(orl_batch* b, int which, void** device_ptr, int64_t* n_elements);
/* The HIP stream (hipStream_t) the batch queues its launches on.  An agent on the same GPU that queues ITS kernels on this
 * stream too (torch: `torch.cuda.ExternalStream(ptr)`) needs no synchronisation between its network and orl_batch_step: the
 * step kernel runs after the kernels that wrote the actions, the network's next forward pass after the step kernel that wrote
 * the observation.  (orl_batch_run also uses a second stream internally and returns synchronised.) */
int orl_batch_stream(orl_batch* b, void** hip_stream_out);

/* state read-back (parity tests, Python attribute surface) */
int orl_batch_get_counters(orl_batch* b, int64_t* out /*[n_envs][ORL_N_COUNTERS]*/);
int orl_batch_get_services(orl_batch* b, double* out /*[n_envs][ORL_N_SERVICE]*/);
int orl_batch_get_slots(orl_batch* b, int64_t env, uint8_t* out /*[cores][links][slots] 0/1*/);
int orl_batch_get_link_stats(orl_batch* b, int64_t env, double* out /*[4][links]: utilization, external_fragmentation, compactness, last_update*/);
/* the same for every env at once: the slot maps as the device keeps them — bit s of 64-bit word s / 64 of a (core, link) row
 * set = slot s free, rows of orl_batch_row_words() words, orl_batch_map_words() words per env (cores * links * row words, padded
 * to an even number) — and [n_envs][4][links] link statistics, [n_envs][4] network statistics */
int orl_batch_row_words(const orl_batch* b);
int orl_batch_map_words(const orl_batch* b);
int orl_batch_get_slots_packed(orl_batch* b, uint64_t* out /*[n_envs][map_words]*/);
int orl_batch_get_link_stats_all(orl_batch* b, double* out /*[n_envs][4][links]*/);
int orl_batch_get_net_stats_all(orl_batch* b, double* out /*[n_envs][4]*/);
/* QoSConstrainedRA: topology.graph["available_spectrum"] of one env (free units per link, optical_network_env.py:189-193) */
int orl_batch_get_spectrum(orl_batch* b, int64_t env, int32_t* out /*[links]*/);
int orl_batch_get_net_stats(orl_batch* b, int64_t env, double* out /*[4]: throughput, compactness, last_update, current_time*/);
int orl_batch_get_active(orl_batch* b, int32_t* out /*[n_envs] pending releases*/);
int orl_batch_get_flags(orl_batch* b, int32_t* out /*[n_envs] bit0 event overflow, bit1 bad action*/);
/* actions_output and actions_taken of one env (batch created with action_histograms): int32 [2][k_paths+1][slots+1]
 * (rmsa_env.py:126-137, 167, 201, 211-212; rwa_env.py:52-58, 103, 125, 132-133 — RWA uses the top-left
 * [k+reject][slots+reject] corner); RMCSA: int32 [2][k_paths+1][modulations+1][cores+1][slots+1] (rmcsa_env.py:145-180,
 * 219, 273, 284-289) */
int orl_batch_get_action_histograms(orl_batch* b, int64_t env, int32_t* out);
/* The pending releases of one env (the reference's heap `_events`, optical_network_env.py:143-154, unordered): returns
 * their number; fills up to `capacity` entries of time_out[] (release time) and rec_out[][6] = (src*n_nodes+dst, path
 * index, initial slot, number of slots, core, bit rate) when both pointers are given. */
int orl_batch_get_pending(orl_batch* b, int64_t env, int32_t capacity, double* time_out, int32_t* rec_out);
/* summed over envs: services_processed, services_accepted (for throughput/blocking reports) */
int orl_batch_totals(orl_batch* b, int64_t* processed, int64_t* accepted);

/* SimpleMatrixObservation (rmsa_env.py:806-837, rmcsa_env.py:914-947) for every env:
 * uint8 [n_envs][2*n_nodes + cores*links*slots] = one-hot(min(src,dst)), one-hot(max(src,dst)), slot map. */
int orl_batch_matrix_obs_dim(const orl_batch* b);
int orl_batch_matrix_observation(orl_batch* b, uint8_t* out);

/* Several GPUs of one node from one process (the `n_devices, device_ids` of the batch constructor).  Envs are
 * independent, so the group is nothing but contiguous shards of the env index range — shard r holds envs
 * [first_env, first_env + n) on device_ids[r], seeds[first_env ...] — each an ordinary orl_batch bound to its device:
 * use orl_multi_shard() with every per-batch entry point above (scatter actions / gather results per shard).  No collective,
 * no peer access.  orl_multi_run drives the device-resident loop of all shards at once, one host thread per shard;
 * stats: [n_shards] or NULL; returns the first non-zero shard status.  The same device may be listed more than once. */
typedef struct orl_multi orl_multi;
int orl_multi_create(const orl_env_config* cfg, const orl_topology_desc* topo, int64_t n_envs, const int64_t* seeds,
                     int n_devices, const int* device_ids, orl_multi** out);
int orl_multi_n_shards(const orl_multi* m);
orl_batch* orl_multi_shard(orl_multi* m, int shard, int64_t* first_env /*nullable*/, int64_t* n_envs /*nullable*/);
int orl_multi_run(orl_multi* m, int policy_id, int64_t n_steps, orl_run_stats* stats);
void orl_multi_destroy(orl_multi* m);

/* Snapshot / restore of the complete simulation state of the batch (slot maps, pending releases, RNG, statistics,
 * counters).  The reference has no equivalent (no checkpointing); used for long PPO runs. */
int64_t orl_batch_state_bytes(orl_batch* b);
int orl_batch_get_state(orl_batch* b, void* out);
int orl_batch_set_state(orl_batch* b, const void* in);

/* Profiling aid: reads the whole slot-map array once with 8-B (width16 = 0) or 16-B (1) loads per lane and returns the
 * number of bytes read, so that rocprofv3's FETCH_SIZE can be calibrated on a known byte count. */
int64_t orl_batch_debug_stream_read(orl_batch* b, int width16);
/* Measurement aid (bench.py, roofline.peak_measured): what the HBM of GPU `device` delivers to plain streaming kernels of this
 * library, measured now with HIP events, best of `reps` launches over `bytes` bytes (>= 1 MiB): *read_gbs — 16 bytes per lane,
 * read only (the kernel of orl_batch_debug_stream_read); *copy_gbs — 16 bytes per lane, copied (read + write bytes counted). */
int orl_debug_stream_peak(int device, int64_t bytes, int reps, double* read_gbs, double* copy_gbs);
/* Specialisation: the persistent kernel with ONE configuration's sizes (topology, spectrum, traffic model, kernel form) as
 * compile-time constants — same source, same results, 5-12 % faster.  It is a small shared library of its own, built on first
 * use from csrc/orl_kernels.hip with the -D flags orl_batch_spec_flags() writes (hipcc --offload-arch=gfx950 -O3 -std=c++17
 * -ffp-contract=off -fPIC <flags> -shared; optical_rl_gym_amd/_build.py build_spec caches it under build/spec/ keyed by the
 * flags and the source hash) and attached with orl_batch_load_spec(), which compares every field with the batch and refuses
 * a mismatch.  orl_spec_flags_for() gives the same flags without a device (pre-building) for a large batch,
 * orl_spec_flags_for_batch() for a batch of n_envs: the kernel form is part of the flags, and batches of at most 12 288 envs of the
 * single-core families take the two-wavefront form (a control and a row wavefront per 8 envs).  All return the
 * string length, 0 when the configuration does not run the persistent kernel.  ORL_PERSIST_SPEC=0 in the environment keeps the
 * generic kernel. */
int orl_batch_spec_flags(orl_batch* b, char* buf, int capacity);
int orl_spec_flags_for(const orl_env_config* cfg, const orl_topology_desc* topo, char* buf, int capacity);
int orl_spec_flags_for_batch(const orl_env_config* cfg, const orl_topology_desc* topo, int64_t n_envs, char* buf, int capacity);
int orl_batch_load_spec(orl_batch* b, const char* so_path);
/* Whether the last orl_batch_run used the attached specialisation — 1: one wavefront per 8 envs, 2: its two-wavefront form (a control
 * and a row wavefront per 8 envs: batches of at most 12 288 envs) — or the generic persistent kernel (0); -1 = another step form. */
int orl_batch_debug_persist_spec(orl_batch* b);
/* The form of the persistent kernel the last orl_batch_run launched (what lives in its LDS window / how the row statistics are
 * done): 0-6 the forms with the row phase in the loop, 7 / 8 the rows-deferred forms (round 6: the loop logs events,
 * k_rowstats replays the link statistics after the launch); -1 = no device-resident run through the persistent kernel yet. */
int orl_batch_debug_persist_form(orl_batch* b);
/* Which kernel orl_batch_step launches for this batch: 2 = k_agent (8 lanes per env, the persistent kernel's phases for one
 * step; QoSConstrainedRA: k_agent_qos, from 20 480 envs), 0 = k_step (one wavefront per env). */
int orl_batch_debug_step_kernel(orl_batch* b);
/* Statistics: env-steps whose releases took the serial tail (more than 8 of one step meeting on one link). */
int64_t orl_batch_debug_serial_count(orl_batch* b);
/* Diagnostic builds with -DORL_TIMING only (zeros otherwise): shader-clock cycles per phase of the persistent kernel, 48
 * slots summed over wavefronts (0-15 control phase, 16-31 release detection, 32-47 row phase). */
int orl_batch_debug_prof(orl_batch* b, uint64_t* out48, int reset);
/* 1 when the library was built with -DORL_ALT_IMPLS (the two-kernel form of the persistent kernel's phases, used by
 * the cross-implementation tests), else 0. */
int orl_build_has_alt(void);

#ifdef __cplusplus
}
#endif
#endif /* ORL_H */
