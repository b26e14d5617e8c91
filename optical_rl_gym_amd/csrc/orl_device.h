// orl_device.h — CDNA4 (gfx950) device code of the batched optical-network environments.
//
// Execution model: ONE 64-lane wavefront per environment instance.  Everything an env
// decides is wave-uniform; lanes are spent on the naturally parallel axes of the state:
//   * (path)           k-shortest-path availability AND-reduce + first-fit run search
//   * (hop of a path)  slot provision / release and the per-link statistics that follow
//   * (event slot)     pending-release scan (64 release times per load instruction)
//   * (MT word)        32 Mersenne-Twister outputs per refill, twisted in place
//   * (node), (link)   weighted node draw (cumulative-table ballot) and link means
// The link x slot availability map is bit-packed (1 = free), W 64-bit words per link
// row, and is staged through LDS once per kernel so that the per-path reductions and
// the provision/release read-modify-writes never go back to HBM.
//
// Reference semantics restated here (file:line relative to the reference repo):
//   optical_rl_gym/envs/rmsa_env.py        step 163-282, reset 284-359, _provision_path 364-415,
//                                          _release_path 417-437, stats 439-543, _next_service 545-597,
//                                          get_number_slots 610-621, is_path_free 623-636,
//                                          get_available_blocks 667-697, compactness 699-744,
//                                          heuristics 747-803
//   optical_rl_gym/envs/deeprmsa_env.py    step 48-58, observation 60-121, reward 123-124, heuristics 135-155
//   optical_rl_gym/envs/rwa_env.py         step 101-162, _next_service 258-288, stats 365-383, heuristics 403-502
//   optical_rl_gym/envs/rmcsa_env.py       step 209-339, crosstalk 341-384, _next_service 690-739, heuristic 882-911
//   optical_rl_gym/envs/optical_network_env.py  _add_release 143-154, _get_node_pair 156-173
// plus CPython's random.Random (MT19937, random(), expovariate, choices, randint) and
// numpy's pairwise float64 summation, which the reference calls on this path.
//
// Compile with -ffp-contract=off: every float64 expression below is written in the
// reference's evaluation order and must not be fused.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "orl_log.h"

namespace orl {

typedef unsigned long long u64;
typedef long long i64;
typedef unsigned int u32;

enum { ENV_RMSA = 0, ENV_DEEPRMSA = 1, ENV_RWA = 2, ENV_RMCSA = 3,
       ENV_QOS = 4 };  // QoSConstrainedRA (qos_constrained_ra.py): per-link spectrum counters instead of slot maps, service classes
enum { POL_SP_FF = 0, POL_SAP_FF = 1, POL_LLP_FF = 2, POL_SAP_LF = 3,
       POL_PATH_FF = 4 };  // PathOnlyFirstFitAction (rmsa_env.py:840-874, rwa_env.py:505-536): first fit on the path the agent chose

// scalar-record slots (one 8-byte word each; 32 per env = one 256-B line, lane l owns word l)
enum {
  SC_NOW = 0, SC_AT, SC_HT, SC_GTHR, SC_GCOMP, SC_GLAST,
  SC_SP, SC_SA, SC_ESP, SC_ESA, SC_BRQ, SC_BRP, SC_EBRQ, SC_EBRP, SC_SBR, SC_SNH,
  SC_SRC_DST, SC_BR_IDX, SC_ID_MTPOS, SC_EV, SC_FLAGS,
  SC_NEXTREL,  // f64 lower bound of every pending release time (-inf = unknown, +inf = none pending)
  SC_HINT,     // low 32 bits: number of entries on the free-slot stack (SC_FREE0..3); 0 = none known
  SC_ACC,      // split pipeline: low 32 = bit 0 action provisioned, bit 1 g_comp update pending, bit 16 serial releases; high 32 = core
  SC_NOWA,     // split pipeline: clock of the provision phase (the row kernel of phase A reads it)
  SC_GC_A,     // split pipeline: last_compactness * last_update of the pending network-compactness update
  SC_GC_TD,    // split pipeline: its time_diff
  SC_TSOON,    // f64: every pending release earlier than this is in the env's soon list (-inf = list invalid)
  SC_FREE0, SC_FREE1, SC_FREE2, SC_FREE3,  // 16 x u16: known-empty pending-release slots (a stack; 8-lane kernels)
  SC_COUNT
};
#define ORL_SCAL_WORDS 32
// the records of a wavefront's 8 envs in its LDS window lie 34 words apart: at 32 (256 bytes = 64 banks' worth) word k of all
// eight records falls into the same bank, and every record load / store of the control phase — ~44 LDS instructions per step,
// 8 lanes per env on the same word — was an 8-way bank conflict
#define ORL_SCAL_LDS_WORDS 34
#define ORL_FREE_SLOTS 16
#define ORL_IMASKS 8  // masks one work item can carry = releases of one step that may meet on one link (orl_device_split.h)
#define ORL_FLAG_EV_OVERFLOW 1
#define ORL_FLAG_BAD_ACTION 2
#define ORL_FLAG_MT2 4  // the env was reseeded: its bit rates keep coming from the stream it was constructed with (mt2)

// soon list (split pipeline): the earliest pending releases of an env, ORL_SOON_PER_LANE per lane of its 8-lane group.
// 5: the most the 3-wave forms of the persistent kernel hold in registers without spilling (cfg2: 167 of 168 VGPRs); 4 -> 5
// measured +1.5 % (cfg2) ... +6 % (cfg4): fewer rebuild scans.
#ifndef ORL_SOON_PER_LANE
#define ORL_SOON_PER_LANE 5
#endif
#define ORL_SOON (8 * ORL_SOON_PER_LANE)

#define ORL_SLOG_ROW_WORDS 3  // statistics log of the persistent kernel: words per env-step (orl_device_split.h, ctrl_d)
struct DevParams {
  int env_type, N, E, K, H, M, S, W, C, episode_length, allow_rejection, J;
  int bit_rate_mode, br_lo, n_br, rand_n, rand_bits;
  int ev_cap, bm_words, n_info, obs_dim, lds_bytes, cs_words;
  double lambda_a, lambda_h;
  double pf_window;  // releases due within this time of the clock are worth an early request of their info word (4 mean inter-arrival times)
  i64 B;
  // shared, read-only (L2-resident) topology / traffic tables
  const int* n_paths;               // [N*N]
  const double* path_length;        // [N*N*K]
  const int* edge_iter_order;       // [E]
  const int* link_pos;              // [E]      inverse permutation: position of link l in topology.edges()
  const double* cum_src;            // [N]      accumulate(node_request_probabilities)
  const double* cum_dst;            // [N*N]    accumulate(renormalised probs with src zeroed)
  const int* bit_rates;             // [n_br]   discrete mode values
  const double* cum_br;             // [n_br]   accumulate(bit_rate_probabilities)
  const unsigned char* nslots;      // [n_br*M] ceil(bit_rate/(se*channel_width))+1
  const double* lmax_snr;           // [M*n_br] RMCSA reach limits
  const double* lmax_xt;            // [M]
  const unsigned char* path_rec;    // [N*N*K][32]  {hops, modulation, link[30]}: one 32-B record per path
  const unsigned char* nslots_path; // [N*N*K][n_br] slots needed on that path (its best modulation) per bit rate
  const double* cum_class;          // [n_classes]  QoSConstrainedRA: accumulate(classes_arrival_probabilities)
  const double* class_reward;       // [n_classes]  classes_reward
  int n_classes;
  // per-env state (struct-of-arrays over envs)
  u64* bitmap;      // [B][bm_words]      bm_words = C*E*W rounded up to a multiple of 2
  double* ev_time;  // [B][ev_cap]        +inf = empty slot
  u64* ev_info;     // [B][ev_cap]        packed {pair_path:24 | slot:12 | n:8 | core:5 | bit_rate:15}
  u32* mt;          // [B][624]           update-behind MT19937 state
  u32* mt2;         // [B][624] or null   after seed(): the stream the env was constructed with, which the reference keeps drawing
                    //                    bit rates from (functools.partial bound in __init__, rmsa_env.py:85-87, 97-99); position in
                    //                    the high half of SC_HINT
  double* lstat;    // [B][E][4]          utilization, external_fragmentation, compactness, last_update
  u64* scal;        // [B][32]
  u64* svc_desc;    // [B]  pending service for the slot-scan kernel: pair_base:32 | br_idx:16 | n_paths:8
  // control phase -> row phase (orl_device_split.h): work items, one per link a step touches
  ulonglong2* q_a;  // [q_cap] mixed items of this step, one region of q_wave slots per control wavefront (8 envs)
  u32* q_cnt_a;     // [ceil(B/8)] items each control wavefront put into its region
  int q_wave;       // item slots per wavefront region
  int persist_ic;   // persistent kernel: keep the per-row cache of inner free runs in LDS (set per launch by the host)
  int persist_evl;  // two-wavefront form: the 8 envs' pending release times in LDS behind the pair's areas (set per launch by the host)
  int persist_fair;    // persistent kernel: rotate the wavefronts' issue priority (s_setprio) over the slots of their SIMD, period 2^persist_fair x 10 ns per level; 0: off
  int persist_finish;  // this launch ends a run: every wavefront that reaches the target finishes its envs' pending
                       // network-compactness update and reports their flags itself (what k_finish2 does in a launch of its own)
  // the row caches of the persistent kernel travel with the state from launch to launch instead of being rebuilt from the
  // slot maps at the start of every launch (~0.3 step's worth of instructions per wavefront: 3 % of a 20-step launch)
  u32* row_cache;        // [ceil(B/8)][2][row_cache_words] per wavefront: inner-run words, then occ / free-block words
  int* row_cache_stamp;  // [ceil(B/8)] the key the wavefront's copy was written under (0 = none)
  int row_cache_words;   // u32 words per cache level and wavefront (8 envs x E links, padded to 16 bytes)
  int row_cache_key;     // this launch's key: changes whenever anything but the persistent kernel may have touched the slot maps,
                         // or the form / cache level differs (set per launch by the host; 0 = do not use stored caches)
  int item_masks;   // releases of one step that may meet on one link before the env falls back to the serial tail (<= 8)
  int rel_limit;    // compact sink of the persistent kernel: releases of one env-step its mask table takes (<= 31; test knob)
  int pipeline2;    // core_sums[2C..4C) accumulates what the current step's releases add to the sums (persistent kernel)
  i64 q_def_stride; // second q_def buffer (the steps of the two-kernel form alternate)
  u32* q_def;       // [0] = number of envs whose releases this step do not fit the item form, [16..] = their indices
  u32* q_stat;      // [1] env-steps that took the serial release path (statistics)
  // service look-ahead of the persistent kernel (orl_device_split.h, svc_generate): what a wavefront held when it left its loop,
  // [ceil(B/8)][64] each, indexed by wavefront and lane.  Between runs every count says "empty".
  double* svc_q;    // inter-arrival time
  double* svc_ht;   // holding time
  u32* svc_pk;      // source | destination << 10 | bit-rate index << 20
  int* svc_cnt;     // services in the group's batch << 8 | next one to take
  // deferred statistics of the persistent kernel (orl_device_split.h: ctrl_d logs, k_stats replays lane-per-env after the launch):
  // three words per env-step, [slot][word][env] so that the replay reads consecutive envs
  u64* slog;        // [log_cap + 1][3][log_stride]; slot s = the s-th step a wavefront ran in this launch
  int* log_n;       // [ceil(B/8)] steps the wavefront logged in this launch | (it finished the run's state itself) << 16
  i64 log_stride;   // envs per row of the log: the whole batch, whichever view of it a launch works on
  int log_cap;      // steps one launch can log per wavefront
  // rows-deferred form of the persistent kernel (round 6; orl_device_split.h ctrl_d<..., RD>, orl_kernels.hip k_rowstats): the loop
  // changes the slot maps itself and logs one 32-byte event per provision / release — {first slot:9 | slots:6 | step of the
  // launch:9 | provision:1, bit mask of the links of the path}, {clock of the event, 0} — in the order the reference applies
  // them; the per-link statistics and the compactness sums are replayed from them after the launch, one lane per link ROW
  ulonglong2* elog; // [B][elog_cap][2]
  int* elog_n;      // [B] events the env logged in this launch
  int elog_cap;     // events per env and launch
  u64* bitmap0;     // [B][bm_words] the slot maps as the launch found them (written when a wavefront fills its window): where the
                    // replay starts from
  u32* ssum;        // [log_cap + 1][log_stride] what the replay hands k_stats: the compactness sums right after each step's provision,
                    // (occupied range sum << 16) | free blocks inside — the fields the in-loop forms put into log word w1
  double* soon_t;   // [B][ORL_SOON] release times of the soon list (+inf = free slot); lane l of the env's group owns l, l+8, ...
  u32* soon_i;      // [B][ORL_SOON] their slots in ev_time / ev_info
  i64 q_cap;
  int* core_sums;   // [B][cs_words]      [2*C] per core: sum(lambda_max-lambda_min), sum(free blocks inside); [2*C] the part
                    //                    of them this step's releases added (pipelines; same 64-byte line)
                    //                    then [C*E] per (core, link): that row's own contribution, (occ << 16) | fb
  i64* br_hist;     // [B][2*n_br]        discrete mode: requested / provisioned histograms
  i64* act_hist;    // [B][(K+1)+(S+1)]   RWA: marginals of actions_output
  int* act2d;       // [B][2][(K+1)*(S+1)] opt-in: actions_output, actions_taken (rmsa_env.py:126-137, rwa_env.py:52-58); else null
  int act2d_words;  // 2*(K+1)*(S+1)
  int* path_col;    // [B] POL_PATH_FF: the path index each env's agent chose (Discrete(k + reject) action)
  int* ep_log;      // [B][ep_cap] or null: episode_services_accepted of every finished episode (evaluate_heuristic on the device)
  int* ep_count;    // [B] episodes finished since the log was armed
  int ep_cap;
  // QoSConstrainedRA: the reward is the accepted service's class reward (qos_constrained_ra.py:131-136), so the harness's
  // episode_reward (utils.py:125-131: += reward per step, from 0.0) is kept as a float64 sum in step order
  double* ep_rew;      // [B][ep_cap] or null: sum of rewards of every finished episode
  double* ep_rew_acc;  // [B] the running episode's sum
  int info_mode;    // k_agent: 0 = every info entry of step() is written; 1 = the blocking rates only (entries 0..3 and the per-rate
                    // ones) — the network-compactness entries and the two means over the links, a read of every link record of
                    // every env per step, are left as they were (orl_batch_set_info_mode: SB3's Monitor reads the rates)
  // I/O (device resident; the C-ABI copies to/from host buffers)
  int* actions;            // [B][4]
  double* reward;          // [B]
  unsigned char* done;     // [B]
  double* info;            // [B][n_info]
  double* obs;             // [B][obs_dim]
  double* term_obs;        // [B][obs_dim]  observation before the auto reset (same values: soft reset keeps the service)
};

// ---------------------------------------------------------------------------------------------
// wave helpers
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63); }

__device__ __forceinline__ u32 rdlane(u32 v, int l) { return (u32)__builtin_amdgcn_readlane((int)v, l); }
__device__ __forceinline__ u64 rdlane64(u64 v, int l) {
  u32 lo = rdlane((u32)v, l), hi = rdlane((u32)(v >> 32), l);
  return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ double rdlane_f64(double v, int l) { return __longlong_as_double((i64)rdlane64((u64)__double_as_longlong(v), l)); }


// compiler-level ordering of LDS/global accesses between phases of one wave (no instructions emitted)
__device__ __forceinline__ void wave_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }

// ---- cross-lane moves as DPP modifiers (VALU, a few cycles) instead of ds_bpermute (LDS pipe, ~100+ cycles of
// dependent latency per step).  PMC on the first version showed ~800 LDS-pipe shuffles per wavefront-step, almost
// all of them the 3-step reductions over an 8-lane row group.  All lanes of the reading group are always active.
#define ORL_DPP_XOR1 0xB1         // quad_perm [1,0,3,2]
#define ORL_DPP_XOR2 0x4E         // quad_perm [2,3,0,1]
#define ORL_DPP_XOR3 0x1B         // quad_perm [3,2,1,0]
#define ORL_DPP_HALF_MIRROR 0x141 // lane i <-> 7-i inside each group of 8
#define ORL_DPP_MIRROR 0x140      // lane i <-> 15-i inside each row of 16
#define ORL_DPP_SHR1 0x111        // row_shr:1: lane i reads lane i-1 of its 16-lane row
// (bound_ctrl = 1 with full row / bank masks: a source lane that is out of range or disabled reads as 0 — what `old` = 0 gave
// with bound_ctrl = 0 — but the destination need not be initialised first: one v_mov_b32_dpp instead of v_mov + v_mov_dpp, and
// foldable into the consuming instruction)
template <int CTRL> __device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, true); }
template <int CTRL> __device__ __forceinline__ double dpp_d(double v) {
  i64 b = __double_as_longlong(v);
  u32 lo = (u32)dpp_i<CTRL>((int)(u32)b), hi = (u32)dpp_i<CTRL>((int)(u32)((u64)b >> 32));
  return __longlong_as_double((i64)(((u64)hi << 32) | lo));
}
__device__ __forceinline__ int wave_sum(int v) {
  v += dpp_i<ORL_DPP_XOR1>(v); v += dpp_i<ORL_DPP_XOR2>(v); v += dpp_i<ORL_DPP_HALF_MIRROR>(v); v += dpp_i<ORL_DPP_MIRROR>(v);
  return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) + __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}
__device__ __forceinline__ int wave_max(int v) {
  int t;
  t = dpp_i<ORL_DPP_XOR1>(v); v = t > v ? t : v;
  t = dpp_i<ORL_DPP_XOR2>(v); v = t > v ? t : v;
  t = dpp_i<ORL_DPP_HALF_MIRROR>(v); v = t > v ? t : v;
  t = dpp_i<ORL_DPP_MIRROR>(v); v = t > v ? t : v;
  int a = __builtin_amdgcn_readlane(v, 0), b = __builtin_amdgcn_readlane(v, 16), c = __builtin_amdgcn_readlane(v, 32), d = __builtin_amdgcn_readlane(v, 48);
  a = a > b ? a : b; c = c > d ? c : d;
  return a > c ? a : c;
}

// ---------------------------------------------------------------------------------------------
// bit-packed slot rows: W 64-bit words, bit s of the row = slot s is free; bits >= S are always 0
// ---------------------------------------------------------------------------------------------
template <int W> struct Row { u64 w[W]; };

template <int W> __device__ __forceinline__ Row<W> row_load(const u64* p) {
  Row<W> r;
#pragma unroll
  for (int i = 0; i < W; i++) r.w[i] = p[i];
  return r;
}
template <int W> __device__ __forceinline__ void row_store(u64* p, const Row<W>& r) {
#pragma unroll
  for (int i = 0; i < W; i++) p[i] = r.w[i];
}
template <int W> __device__ __forceinline__ Row<W> row_and(const Row<W>& a, const Row<W>& b) {
  Row<W> r;
#pragma unroll
  for (int i = 0; i < W; i++) r.w[i] = a.w[i] & b.w[i];
  return r;
}
template <int W> __device__ __forceinline__ Row<W> row_andn(const Row<W>& a, const Row<W>& b) {  // a & ~b
  Row<W> r;
#pragma unroll
  for (int i = 0; i < W; i++) r.w[i] = a.w[i] & ~b.w[i];
  return r;
}
template <int W> __device__ __forceinline__ Row<W> row_or(const Row<W>& a, const Row<W>& b) {
  Row<W> r;
#pragma unroll
  for (int i = 0; i < W; i++) r.w[i] = a.w[i] | b.w[i];
  return r;
}
template <int W> __device__ __forceinline__ bool row_any(const Row<W>& a) {
  u64 o = 0;
#pragma unroll
  for (int i = 0; i < W; i++) o |= a.w[i];
  return o != 0;
}
template <int W> __device__ __forceinline__ int row_popc(const Row<W>& a) {
  int c = 0;
#pragma unroll
  for (int i = 0; i < W; i++) c += __popcll(a.w[i]);
  return c;
}
// index of the lowest set bit, 64*W if none
template <int W> __device__ __forceinline__ int row_ctz(const Row<W>& a) {
  int r = 64 * W;
#pragma unroll
  for (int i = W - 1; i >= 0; i--)
    if (a.w[i]) r = 64 * i + (int)__builtin_ctzll(a.w[i]);
  return r;
}
// index of the highest set bit + 1, 0 if none
template <int W> __device__ __forceinline__ int row_bitlen(const Row<W>& a) {
  int r = 0;
#pragma unroll
  for (int i = 0; i < W; i++)
    if (a.w[i]) r = 64 * i + 64 - (int)__builtin_clzll(a.w[i]);
  return r;
}
// bits [0, n)
template <int W> __device__ __forceinline__ Row<W> row_mask_lo(int n) {
  Row<W> r;
#pragma unroll
  for (int i = 0; i < W; i++) {
    int c = n - 64 * i;
    r.w[i] = c >= 64 ? ~0ull : (c <= 0 ? 0ull : ((1ull << c) - 1ull));
  }
  return r;
}
// bits [s, s+n)
template <int W> __device__ __forceinline__ Row<W> row_range(int s, int n) {
  Row<W> r;
#pragma unroll
  for (int i = 0; i < W; i++) {
    int lo = s - 64 * i, hi = s + n - 64 * i;
    lo = lo < 0 ? 0 : lo;
    hi = hi > 64 ? 64 : hi;
    int c = hi - lo;
    u64 m = c >= 64 ? ~0ull : (c <= 0 ? 0ull : ((1ull << c) - 1ull));
    r.w[i] = c <= 0 ? 0ull : (m << lo);
  }
  return r;
}
// logical shift right by st, 0 < st < 64
template <int W> __device__ __forceinline__ Row<W> row_shr_small(const Row<W>& a, int st) {
  Row<W> r;
#pragma unroll
  for (int i = 0; i < W; i++) {
    u64 hi = (i + 1 < W) ? a.w[i + 1] : 0ull;
    r.w[i] = (a.w[i] >> st) | (hi << (64 - st));
  }
  return r;
}
// the same for 0 < st < 32 (the log-step run search of services of at most 63 slots shifts by at most 31), one v_alignbit_b32 per
// 32-bit half instead of two 64-bit shifts and two ORs per word
template <int W> __device__ __forceinline__ Row<W> row_shr_lt32(const Row<W>& a, int st) {
  Row<W> r;
  u32 d[2 * W + 1];
#pragma unroll
  for (int i = 0; i < W; i++) { d[2 * i] = (u32)a.w[i]; d[2 * i + 1] = (u32)(a.w[i] >> 32); }
  d[2 * W] = 0u;
#pragma unroll
  for (int i = 0; i < W; i++) {
    const u32 lo = __builtin_amdgcn_alignbit(d[2 * i + 1], d[2 * i], (u32)st);
    const u32 hi = __builtin_amdgcn_alignbit(d[2 * i + 2], d[2 * i + 1], (u32)st);
    r.w[i] = ((u64)hi << 32) | lo;
  }
  return r;
}
// x & ~(x << 1): one bit per run of ones (its first position)
template <int W> __device__ __forceinline__ Row<W> row_starts(const Row<W>& a) {
  Row<W> r;
#pragma unroll
  for (int i = 0; i < W; i++) {
    u64 carry = (i > 0) ? (a.w[i - 1] >> 63) : 0ull;
    r.w[i] = a.w[i] & ~((a.w[i] << 1) | carry);
  }
  return r;
}
// bit s set iff slots s .. s+n-1 are all ones in m (1 <= n <= 64).  Log-step shift-AND.
template <int W> __device__ __forceinline__ Row<W> row_runs_ge(const Row<W>& m, int n) {
  Row<W> r = m;
  int have = 1;
  while (have < n) {
    int st = n - have;
    st = st < have ? st : have;
    if (st < 32) r = row_and<W>(r, row_shr_lt32<W>(r, st));  // (the usual case; a real branch: the other side is skipped)
    else r = row_and<W>(r, row_shr_small<W>(r, st));
    have += st;
  }
  return r;
}
// the same, straight-line: the masks of run starts of >= 1, 2, 4, 8, 16, 32 ones, the largest power of two that still has one,
// then five refinement steps — for the row phase, where 40-odd lanes search words with different longest runs and the two
// data-dependent loops of word_longest_run() ran as many rounds as the lane with the longest one, under exec masks.  For x with at
// least one zero bit (runs of at most 63: the row phase searches words without their boundary runs); checked against the loop
// version and a bit-by-bit count on 2e7 words
__device__ __forceinline__ int word_longest_run_flat(u64 x) {
  const u64 r2 = x & (x >> 1), r4 = r2 & (r2 >> 2), r8 = r4 & (r4 >> 4), r16 = r8 & (r8 >> 8), r32 = r16 & (r16 >> 16);
  int L = x ? 1 : 0;
  u64 r = x;
  if (r2) { L = 2; r = r2; }
  if (r4) { L = 4; r = r4; }
  if (r8) { L = 8; r = r8; }
  if (r16) { L = 16; r = r16; }
  if (r32) { L = 32; r = r32; }
  const int p2 = L;  // the power of two found: the steps halve IT
#pragma unroll
  for (int k = 1; k <= 5; k++) {
    const int st = p2 >> k;  // (0 once the steps are used up: r & r, L + 0)
    const u64 t = r & (r >> st);
    const bool ok = (t != 0ull) & (st > 0);
    r = ok ? t : r;
    L += ok ? st : 0;
  }
  return L;
}
// longest run of ones inside one 64-bit word (binary search on run length)
__device__ __forceinline__ int word_longest_run(u64 x) {
  if (!x) return 0;
  int L = 1;
  u64 r = x;
  for (;;) {
    u64 t = (L < 64) ? (r & (r >> L)) : 0ull;
    if (!t) break;
    r = t;
    L <<= 1;
  }
  for (int st = L >> 1; st > 0; st >>= 1) {
    u64 t = r & (r >> st);
    if (t) { r = t; L += st; }
  }
  return L;
}
// longest run of ones across the row (runs continue across word boundaries)
template <int W> __device__ __forceinline__ int row_longest_run(const Row<W>& a) {
  int best = 0, carry = 0;
#pragma unroll
  for (int i = 0; i < W; i++) {
    u64 x = a.w[i];
    if (x == ~0ull) { carry += 64; continue; }
    int lead = (int)__builtin_ctzll(~x);
    int c = carry + lead;
    best = c > best ? c : best;
    int inner = word_longest_run(x);
    best = inner > best ? inner : best;
    carry = (x >> 63) ? (int)__builtin_clzll(~x) : 0;
  }
  return carry > best ? carry : best;
}

// Per-link integer summary that feeds the network spectrum compactness (rmsa_env.py:699-744):
// if the row holds >= 2 used blocks: occ = lambda_max - lambda_min, fb = free blocks strictly inside.
template <int W> __device__ __forceinline__ void link_summary(const Row<W>& a, int S, int& occ, int& fb) {
  Row<W> used = row_andn<W>(row_mask_lo<W>(S), a);
  occ = 0;
  fb = 0;
  if (row_popc<W>(row_starts<W>(used)) > 1) {
    int lo = row_ctz<W>(used), hi = row_bitlen<W>(used);
    occ = hi - lo;
    Row<W> in = row_and<W>(a, row_range<W>(lo, hi - lo));
    fb = row_popc<W>(row_starts<W>(in));
  }
}

// ---------------------------------------------------------------------------------------------
// per-wave working state (wave-uniform values; the compiler keeps most of it in SGPRs)
// ---------------------------------------------------------------------------------------------
struct Env {
  double now, at, ht, g_thr, g_comp, g_last;
  i64 sp, sa, esp, esa, brq, brp, ebrq, ebrp, s_br, s_nh;
  int src, dst, bit_rate, br_idx, id, mt_pos, ev_hwm, ev_cnt, new_service, flags;
  int mt2_pos;  // position in the construction-time stream (ORL_FLAG_MT2)
  // freshly pushed release event of this kernel invocation (kept in registers: the store to
  // ev_time is not guaranteed visible to the other lanes' loads within the same kernel)
  double push_t;
  int push_idx;
  // pointers
  u64* bm;        // LDS: [C*E*W]
  double* ls;     // LDS: [4][E]
  double* scratch;  // LDS: [E]
  double* obs_l;  // LDS: [obs_dim]
  int* cs;        // LDS: [2*C]
  double* ev_time;
  u64* ev_info;
  double* evl;    // LDS copy of ev_time[0, ev_hwm) when the kernel staged it (else nullptr)
  u32* mt;
  i64 env;
};

// the load and the unpacking are separate so that a kernel can put other independent loads between them
__device__ __forceinline__ u64 env_fetch(const DevParams& P, i64 env, int lane) {
  return (lane < ORL_SCAL_WORDS) ? P.scal[env * ORL_SCAL_WORDS + lane] : 0ull;
}
__device__ __forceinline__ void env_unpack(const DevParams& P, Env& e, i64 env, int lane, u64 v) {
#define F64(slot) __longlong_as_double((i64)rdlane64(v, slot))
#define I64(slot) ((i64)rdlane64(v, slot))
  e.now = F64(SC_NOW); e.at = F64(SC_AT); e.ht = F64(SC_HT);
  e.g_thr = F64(SC_GTHR); e.g_comp = F64(SC_GCOMP); e.g_last = F64(SC_GLAST);
  e.sp = I64(SC_SP); e.sa = I64(SC_SA); e.esp = I64(SC_ESP); e.esa = I64(SC_ESA);
  e.brq = I64(SC_BRQ); e.brp = I64(SC_BRP); e.ebrq = I64(SC_EBRQ); e.ebrp = I64(SC_EBRP);
  e.s_br = I64(SC_SBR); e.s_nh = I64(SC_SNH);
  u64 t;
  t = rdlane64(v, SC_SRC_DST); e.src = (int)(u32)t; e.dst = (int)(t >> 32);
  t = rdlane64(v, SC_BR_IDX); e.bit_rate = (int)(u32)t; e.br_idx = (int)(t >> 32);
  t = rdlane64(v, SC_ID_MTPOS); e.id = (int)(u32)t; e.mt_pos = (int)(t >> 32);
  t = rdlane64(v, SC_EV); e.ev_hwm = (int)(u32)t; e.ev_cnt = (int)(t >> 32);
  t = rdlane64(v, SC_FLAGS); e.new_service = (int)(u32)t; e.flags = (int)(t >> 32);
  t = rdlane64(v, SC_HINT); e.mt2_pos = (int)(t >> 32);
#undef F64
#undef I64
  e.push_idx = -1;
  e.push_t = 0.0;
  e.evl = nullptr;
  e.env = env;
  e.ev_time = P.ev_time + env * P.ev_cap;
  e.ev_info = P.ev_info + env * P.ev_cap;
  e.mt = P.mt + env * 624;
}
__device__ __forceinline__ void env_load(const DevParams& P, Env& e, i64 env, int lane) {
  env_unpack(P, e, env, lane, env_fetch(P, env, lane));
}

__device__ __forceinline__ u64 pack2(int lo, int hi) { return ((u64)(u32)hi << 32) | (u64)(u32)lo; }

__device__ __forceinline__ void env_store(const DevParams& P, const Env& e, int lane) {
  u64 v = 0;
#define PUTF(slot, x) if (lane == slot) v = (u64)__double_as_longlong(x);
#define PUTI(slot, x) if (lane == slot) v = (u64)(x);
  PUTF(SC_NOW, e.now) PUTF(SC_AT, e.at) PUTF(SC_HT, e.ht) PUTF(SC_GTHR, e.g_thr) PUTF(SC_GCOMP, e.g_comp) PUTF(SC_GLAST, e.g_last)
  PUTI(SC_SP, e.sp) PUTI(SC_SA, e.sa) PUTI(SC_ESP, e.esp) PUTI(SC_ESA, e.esa)
  PUTI(SC_BRQ, e.brq) PUTI(SC_BRP, e.brp) PUTI(SC_EBRQ, e.ebrq) PUTI(SC_EBRP, e.ebrp)
  PUTI(SC_SBR, e.s_br) PUTI(SC_SNH, e.s_nh)
  PUTI(SC_SRC_DST, pack2(e.src, e.dst)) PUTI(SC_BR_IDX, pack2(e.bit_rate, e.br_idx))
  PUTI(SC_ID_MTPOS, pack2(e.id, e.mt_pos)) PUTI(SC_EV, pack2(e.ev_hwm, e.ev_cnt)) PUTI(SC_FLAGS, pack2(e.new_service, e.flags))
  PUTF(SC_NEXTREL, -__builtin_inf()) PUTI(SC_HINT, pack2(0, e.mt2_pos)) PUTF(SC_TSOON, -__builtin_inf())  // caches of the 8-lane kernels: unknown
#undef PUTF
#undef PUTI
  if (lane < SC_COUNT) P.scal[e.env * ORL_SCAL_WORDS + lane] = v;
  if (lane == 0) {
    u64 np_ = (u64)(u32)P.n_paths[e.src * P.N + e.dst];
    P.svc_desc[e.env] = (u64)(u32)((e.src * P.N + e.dst) * P.K) | ((u64)(u32)e.br_idx << 32) | (np_ << 48);
  }
}

// stage the env's slot map, link statistics and per-core sums into this wave's LDS window
__device__ __forceinline__ void stage_in(const DevParams& P, Env& e, u64* lds, int lane) {
  e.bm = lds;
  e.ls = (double*)(lds + P.bm_words);
  e.scratch = e.ls + 4 * P.E;
  e.obs_l = e.scratch + P.E;
  e.cs = (int*)(e.obs_l + P.obs_dim);
  const ulonglong2* g = (const ulonglong2*)(P.bitmap + e.env * P.bm_words);
  ulonglong2* l = (ulonglong2*)lds;
  for (int i = lane; i < P.bm_words / 2; i += 64) l[i] = g[i];
  const double* gs = P.lstat + e.env * 4 * P.E;
  for (int i = lane; i < 4 * P.E; i += 64) e.ls[i] = gs[i];
  for (int i = lane; i < P.cs_words; i += 64) e.cs[i] = P.core_sums[e.env * P.cs_words + i];
  wave_fence();
}
__device__ __forceinline__ void stage_out(const DevParams& P, Env& e, int lane) {
  wave_fence();
  ulonglong2* g = (ulonglong2*)(P.bitmap + e.env * P.bm_words);
  const ulonglong2* l = (const ulonglong2*)e.bm;
  for (int i = lane; i < P.bm_words / 2; i += 64) g[i] = l[i];
  double* gs = P.lstat + e.env * 4 * P.E;
  for (int i = lane; i < 4 * P.E; i += 64) gs[i] = e.ls[i];
  for (int i = lane; i < P.cs_words; i += 64) P.core_sums[e.env * P.cs_words + i] = e.cs[i];
}

// ---------------------------------------------------------------------------------------------
// MT19937, CPython-compatible stream, "update-behind" in-place form.
//
// CPython regenerates all 624 words at once when the index reaches 624
// (Modules/_randommodule.c genrand_uint32).  Here array position i holds the NEXT generation's
// word if i < pos and the current generation's word if i >= pos: right after word i is handed
// out, its successor-generation value is computed from (M[i], M[i+1], M[i+397]) and written in
// place — the same recurrence, evaluated lazily, so the output sequence is identical but a
// refill is 32 lanes x (3 loads + 1 store) with no 2.5-KB regeneration burst.
// ---------------------------------------------------------------------------------------------
// values a kernel may request at entry so that their latency overlaps the staging of the env state
struct Prefetch {
  bool have_rec;   // rec / nslots belong to the action's path
  int pidx;
  u64 rq0, rq1, rq2, rq3;
  int nslots;
  bool have_cum;   // cum_my = cum_src[min(lane, N-1)] (only when N <= 64)
  double cum_my;
};

struct Rng {
  u32 out;   // lane j < 32: tempered output number (consumed + j) of the stream
  u32 nxt;   // lane j < 32: next-generation value for that array position
  int used;  // words handed out from the current window
};

__device__ __forceinline__ void rng_fill(Env& e, Rng& r, int lane) {
  int i = e.mt_pos + (lane & 31);
  int i0 = i >= 624 ? i - 624 : i;
  int i1 = i0 + 1 >= 624 ? i0 + 1 - 624 : i0 + 1;
  int im = i0 + 397 >= 624 ? i0 + 397 - 624 : i0 + 397;
  u32 cur = e.mt[i0], nx = e.mt[i1], far = e.mt[im];
  u32 y = (cur & 0x80000000u) | (nx & 0x7fffffffu);
  r.nxt = far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
  u32 t = cur;
  t ^= (t >> 11);
  t ^= (t << 7) & 0x9d2c5680u;
  t ^= (t << 15) & 0xefc60000u;
  t ^= (t >> 18);
  r.out = t;
  r.used = 0;
}
__device__ __forceinline__ void rng_commit(Env& e, Rng& r, int lane) {
  if (lane < r.used) {
    int i = e.mt_pos + lane;
    e.mt[i >= 624 ? i - 624 : i] = r.nxt;
  }
  int p = e.mt_pos + r.used;
  e.mt_pos = p >= 624 ? p - 624 : p;
  r.used = 0;
}
__device__ __forceinline__ u32 rng_u32(Env& e, Rng& r, int lane) {
  if (r.used == 32) {
    rng_commit(e, r, lane);
    rng_fill(e, r, lane);
  }
  u32 v = rdlane(r.out, r.used);
  r.used++;
  return v;
}
// random.random(): (a>>5, b>>6) -> 53-bit double
__device__ __forceinline__ double rng_random(Env& e, Rng& r, int lane) {
  u32 a = rng_u32(e, r, lane) >> 5, b = rng_u32(e, r, lane) >> 6;
  return ((double)a * 67108864.0 + (double)b) * (1.0 / 9007199254740992.0);
}
// random.expovariate(lambd) = -log(1.0 - random()) / lambd
__device__ __forceinline__ double rng_expovariate(Env& e, Rng& r, int lane, double lambd) {
  return -orl_log(1.0 - rng_random(e, r, lane)) / lambd;
}
// random.choices(pop, weights)[0] = bisect_right(cum, random() * (cum[-1] + 0.0), 0, n - 1)
//  = number of i in [0, n-2] with cum[i] <= x (cum is non-decreasing)
__device__ __forceinline__ int rng_choice(Env& e, Rng& r, int lane, const double* cum, int n) {
  double x = rng_random(e, r, lane) * (cum[n - 1] + 0.0);
  int cnt = 0;
  for (int base = 0; base < n - 1; base += 64) {
    int i = base + lane;
    bool le = (i < n - 1) && (cum[i] <= x);
    cnt += (int)__popcll(__ballot(le));
  }
  return cnt;
}

// 32-byte path record: byte 0 hops, byte 1 modulation, bytes 2.. link per hop
struct PathRec { u64 q[4]; };
__device__ __forceinline__ PathRec path_rec_load(const DevParams& P, int pidx) {
  const ulonglong2* r = (const ulonglong2*)(P.path_rec + (size_t)pidx * 32);
  ulonglong2 a = r[0], b = r[1];
  PathRec o;
  o.q[0] = a.x; o.q[1] = a.y; o.q[2] = b.x; o.q[3] = b.y;
  return o;
}
__device__ __forceinline__ int path_rec_byte(const PathRec& r, int i) {
  int w = i >> 3;
  u64 v = w == 0 ? r.q[0] : (w == 1 ? r.q[1] : (w == 2 ? r.q[2] : r.q[3]));
  return (int)((v >> ((i & 7) * 8)) & 0xffull);
}
template <int W>
__device__ __forceinline__ Row<W> path_and_rec(const PathRec& r, const u64* bm, int E, int S, int core) {
  Row<W> m = row_mask_lo<W>(S);
  const int hops = path_rec_byte(r, 0);
  const u64* base = bm + (size_t)core * E * W;
  for (int h = 0; h < hops; h += 2) {  // two link rows in flight per round
    const Row<W> r0 = row_load<W>(base + path_rec_byte(r, 2 + h) * W);
    const Row<W> r1 = row_load<W>(base + path_rec_byte(r, (h + 1 < hops) ? 3 + h : 2 + h) * W);
    m = row_and<W>(m, row_and<W>(r0, r1));
  }
  return m;
}

// same draw when lane i already holds cum[min(i, n-1)] in a register (n <= 64)
__device__ __forceinline__ int rng_choice_pre(Env& e, Rng& r, int lane, double cum_my, int n) {
  double x = rng_random(e, r, lane) * (rdlane_f64(cum_my, n - 1) + 0.0);
  return (int)__popcll(__ballot(lane < n - 1 && cum_my <= x));
}

// ---------------------------------------------------------------------------------------------
// shared helpers on the env state
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int pair_base(const DevParams& P, int src, int dst) { return (src * P.N + dst) * P.K; }

// _get_network_compactness from the running integer sums
__device__ __forceinline__ double net_compactness(const DevParams& P, const Env& e, int core) {
  int occ = e.cs[2 * core], fb = e.cs[2 * core + 1];
  if (fb > 0) return ((double)occ / (double)e.s_nh) * ((double)P.E / (double)fb);
  return 1.0;
}

// ---- (row, word) lane layout -------------------------------------------------------------------
// Provision / release touch up to 8 link rows at once: lane = 8*r + w holds 64-bit word w of the r-th
// link row of the path, so every per-row quantity (popcounts, run starts, lambda_min/max, longest free
// run) is a handful of single-word instructions plus a 3-step reduction over the 8 lanes of the row.
__device__ __forceinline__ int g8_sum(int v) { v += dpp_i<ORL_DPP_XOR1>(v); v += dpp_i<ORL_DPP_XOR2>(v); v += dpp_i<ORL_DPP_HALF_MIRROR>(v); return v; }
__device__ __forceinline__ int g8_min(int v) {
  int t;
  t = dpp_i<ORL_DPP_XOR1>(v); v = t < v ? t : v;
  t = dpp_i<ORL_DPP_XOR2>(v); v = t < v ? t : v;
  t = dpp_i<ORL_DPP_HALF_MIRROR>(v); v = t < v ? t : v;
  return v;
}
__device__ __forceinline__ int g8_max(int v) {
  int t;
  t = dpp_i<ORL_DPP_XOR1>(v); v = t > v ? t : v;
  t = dpp_i<ORL_DPP_XOR2>(v); v = t > v ? t : v;
  t = dpp_i<ORL_DPP_HALF_MIRROR>(v); v = t > v ? t : v;
  return v;
}
// value held by the previous lane of the 8-lane row group (lane w-1); `dflt` for w == 0
__device__ __forceinline__ int g8_prev(int v, int w, int dflt) { int t = dpp_i<ORL_DPP_SHR1>(v); return w == 0 ? dflt : t; }
__device__ __forceinline__ u64 word_mask_lo(int c) { return c >= 64 ? ~0ull : (c <= 0 ? 0ull : ((1ull << c) - 1ull)); }
// bits [lo, hi) of a 64-bit word, lo/hi relative to the word and unclamped
__device__ __forceinline__ u64 word_range(int lo, int hi) {
  lo = lo < 0 ? 0 : lo;
  hi = hi > 64 ? 64 : hi;
  return hi <= lo ? 0ull : (word_mask_lo(hi - lo) << lo);
}

struct RowStat { int free_, nf, nu, lo, hi, occ, fb; };

// a = word w of the row (0 for w >= W).  All 8 lanes of the row group return the same RowStat.
template <int W, bool FULL>
__device__ __forceinline__ void row_stat(u64 a, int w, int S, RowStat& st) {
  const u64 maskw = (w < W) ? word_mask_lo(S - 64 * w) : 0ull;
  const u64 used = ~a & maskw;
  const int prev_top = g8_prev((int)(a >> 63), w, 1);  // bit 63 of word w-1 of `a`; w == 0: "used" to the left of slot 0
  const u64 carry_a = (w == 0) ? 0ull : (u64)prev_top;
  const u64 carry_u = (w == 0) ? 0ull : (u64)(prev_top ^ 1);  // slot 64w-1 < S for every w < W
  const u64 st_u = used & ~((used << 1) | carry_u);
  st.nu = g8_sum(__popcll(st_u));
  int lo_l = used ? 64 * w + (int)__builtin_ctzll(used) : (1 << 20);
  int hi_l = used ? 64 * w + 64 - (int)__builtin_clzll(used) : 0;
  st.lo = g8_min(lo_l);
  st.hi = g8_max(hi_l);
  const bool two = st.nu > 1;
  st.occ = two ? st.hi - st.lo : 0;
  // free blocks strictly inside [lambda_min, lambda_max)
  const u64 in = two ? (a & word_range(st.lo - 64 * w, st.hi - 64 * w)) : 0ull;
  const u64 carry_i = (w > 0 && prev_top && (64 * w - 1 >= st.lo) && (64 * w - 1 < st.hi)) ? 1ull : 0ull;
  st.fb = g8_sum(__popcll(in & ~((in << 1) | carry_i)));
  if (FULL) {
    st.free_ = g8_sum(__popcll(a));
    st.nf = g8_sum(__popcll(a & ~((a << 1) | carry_a)));
  }
}

// longest run of ones across the row (runs continue across word boundaries)
template <int W>
__device__ __forceinline__ int row_longest_run8(u64 a, int w) {
  const bool full = (a == ~0ull);
  const int lead = full ? 64 : (int)__builtin_ctzll(~a);
  const int trail = full ? 64 : (int)__builtin_clzll(~a);
  const int inner = word_longest_run(a);
  const int pf = g8_prev(full ? 1 : 0, w, 0), pt = g8_prev(trail, w, 0);
  int c = 0;  // length of the run of ones that ends exactly at the boundary below word w
#pragma unroll
  for (int it = 1; it < W; it++) {
    int pc = g8_prev(c, w, 0);
    c = (w == 0) ? 0 : (pf ? pc + 64 : pt);
  }
  int cand = full ? c + 64 : ((c + lead) > inner ? (c + lead) : inner);
  if (w >= W) cand = 0;
  return g8_max(cand);
}

// set (release) or clear (provision) slots [s0, s0+n) on every link of the path, then do what the
// reference's per-link loop does (_provision_path rmsa_env.py:381-396, _release_path :417-436):
// _update_link_stats (rmsa_env.py:464-543) per touched link, plus the integer sums behind
// _get_network_compactness.  Returns the path's hop count.
// QoSConstrainedRA: the env's "slot map" is one counter per link, topology.graph["available_spectrum"] (e.bm[link], free
// units).  _provision_path / _release_path (qos_constrained_ra.py:296-338): every link of the path -/+ number_slots, then
// _update_link_stats (:355-372, utilisation only).  Lane h owns hop h (the links of a path are distinct).
__device__ __forceinline__ int qos_path_apply(const DevParams& P, Env& e, int lane, const PathRec rec, bool release) {
  const int hops = path_rec_byte(rec, 0);
  if (lane < hops) {
    const int link = path_rec_byte(rec, 2 + lane);
    const i64 avail = (i64)e.bm[link] + (release ? 1 : -1);
    e.bm[link] = (u64)avail;
    const double last_update = e.ls[4 * link + 3];
    const double time_diff = e.now - last_update;
    if (e.now > 0) {
      const double cur_util = (double)((i64)P.S - avail) / (double)P.S;
      e.ls[4 * link] = ((e.ls[4 * link] * last_update) + (cur_util * time_diff)) / e.now;
    }
    e.ls[4 * link + 3] = e.now;
  }
  wave_fence();
  return hops;
}
// is_path_free (qos_constrained_ra.py:381-392) for number_slots = 1
__device__ __forceinline__ bool qos_path_free(const DevParams& P, const Env& e, int lane, const PathRec rec) {
  const int hops = path_rec_byte(rec, 0);
  const bool busy = lane < hops && (i64)e.bm[path_rec_byte(rec, 2 + lane)] < 1;
  return P.S >= 1 && __ballot(busy) == 0ull;
}

template <int ENV, int W>
__device__ __forceinline__ int path_apply(const DevParams& P, Env& e, int lane, const PathRec rec, int core, int s0, int n, bool release) {
  if (ENV == ENV_QOS) return qos_path_apply(P, e, lane, rec, release);
  const int hops = path_rec_byte(rec, 0);
  const int r = lane >> 3, w = lane & 7;
  const int E = P.E, S = P.S;
  int d_occ = 0, d_fb = 0;
  for (int h0 = 0; h0 < hops; h0 += 8) {
    const int h = h0 + r;
    const bool rowv = h < hops;
    const bool v = rowv && (w < W);
    const int link = rowv ? path_rec_byte(rec, 2 + h) : 0;
    u64* wp = e.bm + (core * E + link) * W + (w < W ? w : 0);
    u64 a = v ? *wp : 0ull;
    RowStat before, after;
    // the row's contribution to the compactness sums before the change (a cached copy used to save this summary; it
    // tied this kernel to being the only one that steps the batch — the pipelines do not maintain such a cache)
    if (ENV != ENV_RWA) row_stat<W, false>(a, w, S, before);
    const u64 m = word_range(s0 - 64 * w, s0 + n - 64 * w);
    a = release ? (a | m) : (a & ~m);
    if (v) *wp = a;
    if (ENV != ENV_RWA) {
      row_stat<W, true>(a, w, S, after);
      if (rowv && w == 0) { d_occ += after.occ - before.occ; d_fb += after.fb - before.fb; }
    } else {
      after.free_ = g8_sum(__popcll(a));
    }
    // _update_link_stats: time-weighted running averages, evaluated in the reference's operation order
    double last_update = e.ls[4 * link + 3];
    double time_diff = e.now - last_update;
    if (e.now > 0) {
      const int free_ = after.free_;
      double cur_util = (double)(S - free_) / (double)S;
      double util = ((e.ls[4 * link] * last_update) + (cur_util * time_diff)) / e.now;
      double frag = 0.0, comp = 0.0;
      if (ENV != ENV_RWA) {
        double cur_frag = 0.0, cur_comp = 0.0;
        const int top = (S - 1) - 64 * w;  // bit of slot S-1 inside this lane's word, if it is here
        const int edge = g8_sum(((w == 0 && (a & 1ull)) ? 1 : 0) + ((top >= 0 && top < 64 && ((a >> top) & 1ull)) ? 1 : 0));
        const int max_empty = row_longest_run8<W>(a, w);
        if (free_ > 0) {
          int me = (after.nf > 1 && !(after.nf == 2 && edge == 2)) ? max_empty : 0;
          cur_frag = 1.0 - ((double)me / (double)free_);
          if (after.nu > 1) cur_comp = ((double)(after.hi - after.lo) / (double)(S - free_)) * (1.0 / (double)after.nu);
          else cur_comp = 1.0;
        }
        frag = ((e.ls[4 * link + 1] * last_update) + (cur_frag * time_diff)) / e.now;
        comp = ((e.ls[4 * link + 2] * last_update) + (cur_comp * time_diff)) / e.now;
      }
      if (rowv && w == 0) {
        e.ls[4 * link] = util;
        if (ENV != ENV_RWA) { e.ls[4 * link + 1] = frag; e.ls[4 * link + 2] = comp; }
      }
    }
    if (rowv && w == 0) e.ls[4 * link + 3] = e.now;
    wave_fence();
  }
  if (ENV != ENV_RWA) {
    d_occ = wave_sum(d_occ);
    d_fb = wave_sum(d_fb);
    if (lane == 0) {
      e.cs[2 * core] += d_occ;
      e.cs[2 * core + 1] += d_fb;
    }
  }
  wave_fence();
  return hops;
}

// is_path_free: all links of the path free on [s0, s0+n)
template <int W>
__device__ __forceinline__ bool path_is_free(const DevParams& P, const Env& e, int lane, const PathRec rec, int core, int s0, int n) {
  if (s0 + n > P.S) return false;
  const int hops = path_rec_byte(rec, 0);
  const int r = lane >> 3, w = lane & 7;
  bool busy = false;
  for (int h0 = 0; h0 < hops; h0 += 8) {
    const int h = h0 + r;
    if (h < hops && w < W) {
      int link = path_rec_byte(rec, 2 + h);
      u64 a = e.bm[(core * P.E + link) * W + w];
      busy = busy || ((word_range(s0 - 64 * w, s0 + n - 64 * w) & ~a) != 0ull);
    }
  }
  return __ballot(busy) == 0ull;
}

// AND of the link rows of a path (lane-private: every lane may call it with its own pidx/core)
template <int W>
__device__ __forceinline__ Row<W> path_and(const DevParams& P, const Env& e, int pidx, int core) {
  return path_and_rec<W>(path_rec_load(P, pidx), e.bm, P.E, P.S, core);
}

// pending-release storage: unordered slots, +inf = empty.  Push = lowest empty slot.
// EVL: the kernel staged ev_time[0, hwm) into LDS (e.evl); scans run there, updates go to both copies.
template <bool EVL>
__device__ __forceinline__ double ev_read(const Env& e, int i) { return EVL ? e.evl[i] : e.ev_time[i]; }

template <bool EVL>
__device__ __forceinline__ void ev_push(const DevParams& P, Env& e, int lane, double t, u64 info) {
  int idx = -1;
  for (int base = 0; base < e.ev_hwm; base += 64) {
    int i = base + lane;
    bool empty = (i < e.ev_hwm) && (ev_read<EVL>(e, i) == __builtin_inf());
    u64 b = __ballot(empty);
    if (b) { idx = base + (int)__builtin_ctzll(b); break; }
  }
  if (idx < 0) {
    if (e.ev_hwm >= P.ev_cap) { e.flags |= ORL_FLAG_EV_OVERFLOW; return; }
    idx = e.ev_hwm++;
  }
  if (lane == 0) {
    e.ev_time[idx] = t;
    e.ev_info[idx] = info;
    if (EVL) e.evl[idx] = t;
  }
  wave_fence();
  e.ev_cnt++;
  e.push_idx = idx;
  e.push_t = t;
}

__device__ __forceinline__ u64 ev_pack(int pidx, int s0, int n, int core, int bit_rate) {
  return (u64)(u32)pidx | ((u64)(u32)s0 << 24) | ((u64)(u32)n << 36) | ((u64)(u32)core << 44) | ((u64)(u32)bit_rate << 49);
}

// release every pending service with release_time <= now, in increasing time order
// (the reference pops its heap until the top is in the future: rmsa_env.py:590-597).
// Each round finds the smallest (time, slot) strictly after the previous one, so the result
// does not depend on whether this kernel's own ev_time stores are visible to the loads yet.
template <int ENV, int W, bool EVL>
__device__ __forceinline__ void release_due(const DevParams& P, Env& e, int lane) {
  double prev_t = -__builtin_inf();
  int prev_i = -1;
  for (;;) {
    double bt = __builtin_inf();
    int bi = 0x7fffffff;
    for (int base = 0; base < e.ev_hwm; base += 64) {
      int i = base + lane;
      if (i < e.ev_hwm) {
        double t = (i == e.push_idx) ? e.push_t : ev_read<EVL>(e, i);
        bool after = (t > prev_t) || (t == prev_t && i > prev_i);
        if (after && (t < bt || (t == bt && i < bi))) { bt = t; bi = i; }
      }
    }
#define ORL_MIN_STEP(CTRL) { double ot = dpp_d<CTRL>(bt); int oi = dpp_i<CTRL>(bi); if (ot < bt || (ot == bt && oi < bi)) { bt = ot; bi = oi; } }
    ORL_MIN_STEP(ORL_DPP_XOR1) ORL_MIN_STEP(ORL_DPP_XOR2) ORL_MIN_STEP(ORL_DPP_HALF_MIRROR) ORL_MIN_STEP(ORL_DPP_MIRROR)
#undef ORL_MIN_STEP
    {  // every lane of a 16-lane row now holds the row minimum: combine the four rows through readlane
      double t0 = rdlane_f64(bt, 0), t1 = rdlane_f64(bt, 16), t2 = rdlane_f64(bt, 32), t3 = rdlane_f64(bt, 48);
      int i0 = __builtin_amdgcn_readlane(bi, 0), i1 = __builtin_amdgcn_readlane(bi, 16), i2 = __builtin_amdgcn_readlane(bi, 32), i3 = __builtin_amdgcn_readlane(bi, 48);
      if (t1 < t0 || (t1 == t0 && i1 < i0)) { t0 = t1; i0 = i1; }
      if (t3 < t2 || (t3 == t2 && i3 < i2)) { t2 = t3; i2 = i3; }
      if (t2 < t0 || (t2 == t0 && i2 < i0)) { t0 = t2; i0 = i2; }
      bt = t0; bi = i0;
    }
    if (!(bt <= e.now)) break;
    u64 info = e.ev_info[bi];
    int pidx = (int)(info & 0xffffffu), s0 = (int)((info >> 24) & 0xfffu), n = (int)((info >> 36) & 0xffu);
    int core = (int)((info >> 44) & 0x1fu), br = (int)((info >> 49) & 0x7fffu);
    if (lane == 0) { e.ev_time[bi] = __builtin_inf(); if (EVL) e.evl[bi] = __builtin_inf(); }
    if (bi == e.push_idx) e.push_t = __builtin_inf();
    e.ev_cnt--;
    int hops_r = path_apply<ENV, W>(P, e, lane, path_rec_load(P, pidx), core, s0, n, true);
    e.s_br -= br;
    e.s_nh -= (i64)n * hops_r;
    prev_t = bt;
    prev_i = bi;
  }
  // shrink the scan window when its tail is empty
  while (e.ev_hwm > 0) {
    int i = e.ev_hwm - 1;
    double t = (i == e.push_idx) ? e.push_t : ev_read<EVL>(e, i);
    // entries released above read +inf or a stale finite time <= now; both mean "empty"
    if (t == __builtin_inf() || t <= e.now) e.ev_hwm--; else break;
  }
}

// _next_service
template <int ENV, int W, bool EVL>
__device__ __forceinline__ void next_service(const DevParams& P, Env& e, int lane, const Rng* prefilled, const Prefetch* pf = nullptr) {
  if (e.new_service) return;
  Rng r;
  if (prefilled) r = *prefilled;  // window loaded at kernel entry so its latency hides behind the step logic
  else rng_fill(e, r, lane);
  double at = e.now + rng_expovariate(e, r, lane, P.lambda_a);
  e.now = at;
  double ht = rng_expovariate(e, r, lane, P.lambda_h);
  int src = (pf && pf->have_cum) ? rng_choice_pre(e, r, lane, pf->cum_my, P.N) : rng_choice(e, r, lane, P.cum_src, P.N);
  int dst = rng_choice(e, r, lane, P.cum_dst + src * P.N, P.N);
  int bit_rate = 0, br_idx = 0;
  if (ENV == ENV_QOS) {  // the service class (qos_constrained_ra.py:262-265) rides in the bit-rate fields
    br_idx = rng_choice(e, r, lane, P.cum_class, P.n_classes);
    bit_rate = br_idx;
  }
  if (ENV != ENV_RWA && ENV != ENV_QOS) {
    // After seed() the reference draws the bit rate from the Random object it bound at construction (functools.partial,
    // rmsa_env.py:85-87 / 97-99) while everything else uses the new one: a second stream for envs that were reseeded.
    const bool second = P.mt2 && (e.flags & ORL_FLAG_MT2);
    u32* mt_main = e.mt;
    int pos_main = 0;
    Rng r2;
    if (second) {
      rng_commit(e, r, lane);
      pos_main = e.mt_pos;
      e.mt = P.mt2 + e.env * 624;
      e.mt_pos = e.mt2_pos;
      rng_fill(e, r2, lane);
    }
    Rng& rb = second ? r2 : r;
    if (P.bit_rate_mode == 0) {  // randint(lo, hi) = lo + _randbelow(hi + 1 - lo)
      u32 v = rng_u32(e, rb, lane) >> (32 - P.rand_bits);
      while ((int)v >= P.rand_n) v = rng_u32(e, rb, lane) >> (32 - P.rand_bits);
      br_idx = (int)v;
      bit_rate = P.br_lo + br_idx;
    } else {
      br_idx = rng_choice(e, rb, lane, P.cum_br, P.n_br);
      bit_rate = P.bit_rates[br_idx];
    }
    if (second) {
      rng_commit(e, r2, lane);
      e.mt2_pos = e.mt_pos;
      e.mt = mt_main;
      e.mt_pos = pos_main;
      r.used = 0;  // (already committed)
    }
  }
  rng_commit(e, r, lane);
  if (ENV == ENV_RWA || ENV == ENV_RMCSA || ENV == ENV_QOS) release_due<ENV, W, EVL>(P, e, lane);
  e.id = (int)e.esp;
  e.src = src; e.dst = dst; e.at = at; e.ht = ht; e.bit_rate = bit_rate; e.br_idx = br_idx;
  e.new_service = 1;
  if (ENV == ENV_RMSA || ENV == ENV_DEEPRMSA) { e.sp += 1; e.esp += 1; }
  if (ENV != ENV_RWA && ENV != ENV_QOS) {
    e.brq += bit_rate;
    e.ebrq += bit_rate;
    if (P.bit_rate_mode == 1 && lane == 0) P.br_hist[e.env * 2 * P.n_br + br_idx] += 1;
  }
  if (ENV == ENV_RMSA || ENV == ENV_DEEPRMSA) release_due<ENV, W, EVL>(P, e, lane);
}

// soft reset (reset(only_episode_counters=True)): the pending service is counted again
template <int ENV> __device__ __forceinline__ void soft_reset(Env& e) {
  e.ebrq = 0; e.ebrp = 0; e.esp = 0; e.esa = 0;
  if (ENV != ENV_RWA && ENV != ENV_QOS && e.new_service) { e.esp += 1; e.ebrq += e.bit_rate; }
}

// get_available_blocks (rmsa_env.py:667-697): the free runs of m with length >= n, low to high — for the action decode of
// DeepRMSAEnv.step (deeprmsa_env.py:48-58), which needs only the start of block number `want - 1`: no arrays (indexed by a run-time block number they lived in scratch memory: the 64 bytes of private
// segment and a dozen "spilled" registers of every DeepRMSA kernel).  Returns how many of the first `want` blocks exist;
// `start`: the first slot of the last one found.
template <int W>
__device__ __forceinline__ int nth_block(const Row<W>& m, int S, int n, int want, int& start) {
  Row<W> r = row_runs_ge<W>(m, n);
  Row<W> zeros = row_andn<W>(row_mask_lo<W>(S), m);
  int found = 0;
  while (found < want && row_any<W>(r)) {
    const int s = row_ctz<W>(r);
    Row<W> z = row_andn<W>(zeros, row_mask_lo<W>(s));
    const int end = row_any<W>(z) ? row_ctz<W>(z) : S;
    start = s;
    found++;
    r = row_andn<W>(r, row_mask_lo<W>(end));
  }
  return found;
}

// DeepRMSAEnv.observation (deeprmsa_env.py:60-121); lanes = paths, lane 0 writes the header
template <int W>
__device__ __forceinline__ void deep_observation(const DevParams& P, const Env& e, int lane, double* obs_out, double* obs_out2) {
  const int N = P.N, J = P.J, S = P.S, WD = 2 * J + 3;
  double* obs = e.obs_l;  // assembled in LDS, written out coalesced
  for (int i = lane; i < P.obs_dim; i += 64) obs[i] = (i >= 1 + 2 * N) ? -1.0 : 0.0;
  wave_fence();
  int mn = e.src < e.dst ? e.src : e.dst, mx = e.src < e.dst ? e.dst : e.src;
  if (lane == 0) {
    obs[0] = (double)e.bit_rate / 100;
    obs[1 + mn] = 1.0;
    obs[1 + N + mx] = 1.0;
  }
  int np_ = P.n_paths[e.src * N + e.dst];
  if (lane < np_) {
    int pidx = pair_base(P, e.src, e.dst) + lane;
    Row<W> m = path_and<W>(P, e, pidx, 0);
    int n = P.nslots_path[(size_t)pidx * P.n_br + e.br_idx];
    double* sp = obs + 1 + 2 * N + lane * WD;
    Row<W> r = row_runs_ge<W>(m, n);
    Row<W> zeros = row_andn<W>(row_mask_lo<W>(S), m);
    for (int b = 0; b < J && row_any<W>(r); b++) {
      int s = row_ctz<W>(r);
      Row<W> z = row_andn<W>(zeros, row_mask_lo<W>(s));
      int end = row_any<W>(z) ? row_ctz<W>(z) : S;
      sp[2 * b] = 2 * ((double)s - 0.5 * (double)S) / (double)S;
      sp[2 * b + 1] = (double)(end - s - 8) / 8;
      r = row_andn<W>(r, row_mask_lo<W>(end));
    }
    sp[2 * J] = ((double)n - 5.5) / 3.5;
    int tot = row_popc<W>(m);
    sp[2 * J + 1] = 2 * ((double)tot - 0.5 * (double)S) / (double)S;
    int nruns = row_popc<W>(row_starts<W>(m));
    if (nruns > 0) sp[2 * J + 2] = ((double)tot / (double)nruns - 4) / 4;
  }
  wave_fence();
  for (int i = lane; i < P.obs_dim; i += 64) {
    double v = obs[i];
    obs_out[i] = v;
    if (obs_out2) obs_out2[i] = v;
  }
  wave_fence();
}

// np.mean over the per-link values taken in topology.edges() order: numpy pairwise sum then / E
__device__ __forceinline__ double link_mean(const DevParams& P, const double* vals /*LDS [E]*/, double* scratch /*LDS [E]*/, int lane) {
  const int E = P.E;
  for (int i = lane; i < E; i += 64) scratch[i] = vals[4 * P.edge_iter_order[i]];
  wave_fence();
  double res;
  if (E < 8) {
    res = 0.;
    for (int i = 0; i < E; i++) res += scratch[i];
  } else {  // E <= 128 (checked on the host)
    double r = (lane < 8) ? scratch[lane] : 0.0;
    int i;
    for (i = 8; i < E - (E % 8); i += 8)
      if (lane < 8) r += scratch[i + lane];
    double r0 = rdlane_f64(r, 0), r1 = rdlane_f64(r, 1), r2 = rdlane_f64(r, 2), r3 = rdlane_f64(r, 3);
    double r4 = rdlane_f64(r, 4), r5 = rdlane_f64(r, 5), r6 = rdlane_f64(r, 6), r7 = rdlane_f64(r, 7);
    res = ((r0 + r1) + (r2 + r3)) + ((r4 + r5) + (r6 + r7));
    for (; i < E; i++) res += scratch[i];
  }
  wave_fence();
  return res / (double)E;
}

// evaluate_heuristic on the device (utils.py:103-141): what an episode's reward sum derives from, logged when it ends
__device__ __forceinline__ void episode_log(const DevParams& P, i64 env, i64 accepted) {
  const int idx = P.ep_count[env];
  if (idx < P.ep_cap) P.ep_log[env * P.ep_cap + idx] = (int)accepted;
  P.ep_count[env] = idx + 1;
}

// opt-in action histograms (rmsa_env.py:126-137, 167, 201, 211-212; rwa_env.py:52-58, 103, 125, 132-133;
// rmcsa_env.py:145-180, 219, 273, 284-289): actions_output[action] counts every action, actions_taken[action] the accepted
// ones and its last element ([k, S], RMCSA [k, M, C, S]) the rejections.  `idx`: the action's flat index in one array.
__device__ __forceinline__ void act_hist_count(const DevParams& P, i64 env, int idx, bool accepted) {
  int* h = P.act2d + env * P.act2d_words;
  const int half = P.act2d_words >> 1;
  h[idx] += 1;
  h[half + (accepted ? idx : half - 1)] += 1;
}
__device__ __forceinline__ void act2d_count(const DevParams& P, i64 env, int path, int slot, bool accepted) {
  act_hist_count(P, env, path * (P.S + 1) + slot, accepted);
}
__device__ __forceinline__ void act4d_count(const DevParams& P, i64 env, int path, int mod, int core, int slot, bool accepted) {
  act_hist_count(P, env, ((path * (P.M + 1) + mod) * (P.C + 1) + core) * (P.S + 1) + slot, accepted);
}

// ---------------------------------------------------------------------------------------------
// heuristics (the policy side) — the slot-scan.
//
// GS lanes cooperate on one env (GS = 8: eight envs per wavefront, lanes = paths; GS = 64: one env per
// wavefront, lanes = (path, core) pairs for RMCSA).  Each lane ANDs the link rows of its path out of the
// LDS-staged slot map, detects runs of >= n free slots by log-step shift-AND and reports its first fit;
// a group ballot picks the path.  `bm` is this env's map in LDS.
// ---------------------------------------------------------------------------------------------
template <int GS> __device__ __forceinline__ u64 group_ballot(bool p, int lane) {
  u64 b = __ballot(p);
  if (GS == 64) return b;
  return (b >> ((lane / GS) * GS)) & ((1ull << (GS & 63)) - 1ull);
}
template <int GS> __device__ __forceinline__ int group_max(int v) { return GS == 8 ? g8_max(v) : wave_max(v); }
template <int GS> __device__ __forceinline__ int group_get(int v, int src, int lane) { return __shfl(v, (lane & ~(GS - 1)) + src, 64); }

// What the scan's winning lane knows of the chosen path, handed to a control phase that runs in the same kernel on the same
// state (k_persist): the slots the service needs on it and the path record — reading them again after the scan was a global
// round trip in the step's dependent chain.  Only the first `words` 8-byte words of the record (hops, modulation and the first
// 8 x words - 2 links) are fetched from the winning lane: the caller asks for what the longest path needs.
struct ScanHand { int n; u64 q[4]; int words; };
template <int GS> __device__ __forceinline__ u64 group_get64(u64 v, int src, int lane) {
  const u32 lo = (u32)__shfl((int)(u32)v, (lane & ~(GS - 1)) + src, 64), hi = (u32)__shfl((int)(u32)(v >> 32), (lane & ~(GS - 1)) + src, 64);
  return ((u64)hi << 32) | lo;
}
template <int ENV, int W, int GS>
__device__ __forceinline__ void policy_g(const DevParams& P, const u64* bm, bool valid, int pb, int br_idx, int np_,
                                         int lane, int pol, int pcol, int* a, ScanHand* hand = nullptr) {
  const int K = P.K, S = P.S;
  const int p = lane & (GS - 1);
  a[0] = a[1] = a[2] = a[3] = 0;
  if (ENV == ENV_QOS) {
    // shortest_path / shortest_available_path / least_loaded_path (qos_constrained_ra.py:408-450); br_idx = service class.
    // Lane = path: free on every link (>= 1 unit), hop count, capacity = min over the links.
    bool free_ = false;
    int hops = 0;
    i64 cap = 0;
    if (valid && p < np_) {
      const PathRec rec = path_rec_load(P, pb + p);
      hops = path_rec_byte(rec, 0);
      cap = (i64)1 << 40;
      for (int h = 0; h < hops; h++) { const i64 v = (i64)bm[path_rec_byte(rec, 2 + h)]; cap = v < cap ? v : cap; }
      free_ = S >= 1 && cap >= 1;
    }
    if (pol == POL_SP_FF) {
      a[0] = (group_ballot<GS>(free_, lane) & 1ull) ? 0 : K;
    } else if (br_idx == 0) {
      a[0] = 0;  // high-priority services only accept the shortest path
    } else if (pol == POL_SAP_FF) {  // fewest hops among the free paths, the earlier one on ties (strict <)
      const int mh = -group_max<GS>(free_ ? -hops : -(1 << 20));
      const u64 bb = group_ballot<GS>(free_ && hops == mh, lane);
      a[0] = bb ? (int)__builtin_ctzll(bb) : K;
    } else {  // largest capacity, the earlier path on ties (strict >, starting from np.finfo(0.0).min: free or not)
      const int mx = group_max<GS>((valid && p < np_) ? (int)cap : -1);
      const u64 bb = group_ballot<GS>(valid && p < np_ && (int)cap == mx, lane);
      a[0] = bb ? (int)__builtin_ctzll(bb) : K;
    }
    return;
  }
  if (ENV == ENV_RMSA) {
    // KSP first-fit incl. the reference's off-by-one: start slots 0 .. S-n-1 only (rmsa_env.py:774-776)
    a[0] = K; a[1] = S;
    int slot = -1, freec = 0;
    int limit = (pol == POL_SP_FF) ? 1 : np_;
    // PathOnlyFirstFitAction: only the chosen path is scanned (a choice >= k, or beyond the pair's paths, rejects)
    const bool mine = (pol == POL_PATH_FF) ? (p == pcol && pcol < K) : true;
    PathRec rec;
    rec.q[0] = rec.q[1] = rec.q[2] = rec.q[3] = 0;
    int n = 1;
    if (valid && p < limit && mine) {
      int pidx = pb + p;
      rec = path_rec_load(P, pidx);
      n = P.nslots_path[(size_t)pidx * P.n_br + br_idx];
      Row<W> m = path_and_rec<W>(rec, bm, P.E, S, 0);
      Row<W> cand = row_and<W>(row_runs_ge<W>(m, n), row_mask_lo<W>(S - n));
      if (row_any<W>(cand)) { slot = row_ctz<W>(cand); if (pol == POL_LLP_FF) freec = row_popc<W>(m); }
    }
    u64 fit = group_ballot<GS>(slot >= 0, lane);
    int best = fit ? (int)__builtin_ctzll(fit) : -1;
    if (pol == POL_LLP_FF) {  // most free slots on the AND-row, first path wins ties (strict >)
      int mx = group_max<GS>(slot >= 0 ? freec : -1);
      u64 bb = group_ballot<GS>(slot >= 0 && freec == mx, lane);
      best = (bb && mx > 0) ? (int)__builtin_ctzll(bb) : -1;
    }
    int bslot = group_get<GS>(slot, best < 0 ? 0 : best, lane);
    if (best >= 0) { a[0] = best; a[1] = bslot; }
    if (hand) {
      const int src = best < 0 ? 0 : best;
      hand->n = group_get<GS>(n, src, lane);
#pragma unroll
      for (int i = 0; i < 4; i++) hand->q[i] = (i < hand->words) ? group_get64<GS>(rec.q[i], src, lane) : 0ull;
    }
  } else if (ENV == ENV_DEEPRMSA) {
    // a[0]: the action (route * j + block 0); a[1], a[2]: what DeepRMSAEnv.step decodes it to on this very slot map — the
    // route and the first slot of its first block, (k, S) when the action rejects — so that a control phase fed by this scan
    // in the same kernel need not walk the blocks again (deeprmsa_env.py:48-58)
    a[0] = K * P.J; a[1] = K; a[2] = S;
    bool has = false;
    int first = -1;
    int limit = (pol == POL_SP_FF) ? 1 : np_;
    PathRec rec;
    rec.q[0] = rec.q[1] = rec.q[2] = rec.q[3] = 0;
    int n = 1;
    if (valid && p < limit) {
      int pidx = pb + p;
      rec = path_rec_load(P, pidx);
      n = P.nslots_path[(size_t)pidx * P.n_br + br_idx];
      Row<W> m = path_and_rec<W>(rec, bm, P.E, S, 0);
      const Row<W> r = row_runs_ge<W>(m, n);
      has = row_any<W>(r);
      if (has) first = row_ctz<W>(r);  // the lowest start of n free slots is the start of the first block of >= n
    }
    u64 fit = group_ballot<GS>(has, lane);
    const int route = (pol == POL_SP_FF) ? 0 : (fit ? (int)__builtin_ctzll(fit) : 0);
    const int rslot = group_get<GS>(first, route, lane);
    if (pol == POL_SP_FF) a[0] = (!P.allow_rejection || fit) ? 0 : K * P.J;
    else if (fit) a[0] = route * P.J;
    if ((fit >> route) & 1ull) { a[1] = route; a[2] = rslot; }
    if (hand) {
      hand->n = group_get<GS>(n, route, lane);
#pragma unroll
      for (int i = 0; i < 4; i++) hand->q[i] = (i < hand->words) ? group_get64<GS>(rec.q[i], route, lane) : 0ull;
    }
  } else if (ENV == ENV_RWA) {
    a[0] = K; a[1] = S;
    int slot = -1, cap = 0, hops = 0;
    int limit = (pol == POL_SP_FF) ? 1 : np_;
    const bool mine = (pol == POL_PATH_FF) ? (p == pcol && pcol < K) : true;
    PathRec rec;
    rec.q[0] = rec.q[1] = rec.q[2] = rec.q[3] = 0;
    if (valid && p < limit && mine) {
      int pidx = pb + p;
      rec = path_rec_load(P, pidx);
      Row<W> m = path_and_rec<W>(rec, bm, P.E, S, 0);
      hops = path_rec_byte(rec, 0);
      cap = row_popc<W>(m);
      if (pol == POL_SAP_LF) {  // range(S-1, 0, -1): wavelength 0 is never tried (rwa_env.py:473)
        Row<W> c = row_andn<W>(m, row_mask_lo<W>(1));
        if (row_any<W>(c)) slot = row_bitlen<W>(c) - 1;
      } else if (row_any<W>(m)) {
        slot = row_ctz<W>(m);
      }
    }
    int best = -1;
    if (pol == POL_SP_FF) {
      best = (group_ballot<GS>(slot >= 0, lane) & 1ull) ? 0 : -1;
    } else if (pol == POL_PATH_FF) {
      const u64 fit = group_ballot<GS>(slot >= 0, lane);
      best = fit ? (int)__builtin_ctzll(fit) : -1;
    } else if (pol == POL_LLP_FF) {
      // cap > best_load with best_load = -DBL_MAX initially: the first path with the largest positive capacity
      int mx = group_max<GS>(slot >= 0 ? cap : -1);
      u64 bb = group_ballot<GS>(slot >= 0 && cap == mx, lane);
      if (mx > 0 && bb) best = (int)__builtin_ctzll(bb);
    } else {
      // fewest hops among the paths that have a free wavelength; earlier path wins ties (strict <)
      int mh = -group_max<GS>(slot >= 0 ? -hops : -(1 << 20));
      u64 bb = group_ballot<GS>(slot >= 0 && hops == mh, lane);
      if (bb) best = (int)__builtin_ctzll(bb);
    }
    int bslot = group_get<GS>(slot, best < 0 ? 0 : best, lane);
    if (best >= 0) { a[0] = best; a[1] = bslot; }
    if (hand) {
      const int src = best < 0 ? 0 : best;
      hand->n = 1;
#pragma unroll
      for (int i = 0; i < 4; i++) hand->q[i] = (i < hand->words) ? group_get64<GS>(rec.q[i], src, lane) : 0ull;
    }
  } else if (ENV == ENV_RMCSA) {
    // lanes = (path, core) pairs in the reference's loop order: path-major, then core (rmcsa_env.py:889-906)
    a[0] = K; a[1] = P.M; a[2] = P.C; a[3] = S;
    int best = -1, bslot = 0;
    for (int base = 0; base < np_ * P.C && best < 0; base += GS) {
      int q = base + p, slot = -1;
      if (valid && q < np_ * P.C) {
        int pth = q / P.C, core = q - pth * P.C;
        int pidx = pb + pth;
        PathRec rec = path_rec_load(P, pidx);
        int n = P.nslots_path[(size_t)pidx * P.n_br + br_idx];
        Row<W> m = path_and_rec<W>(rec, bm, P.E, S, core);
        Row<W> cand = row_and<W>(row_runs_ge<W>(m, n), row_mask_lo<W>(S - n));
        if (row_any<W>(cand)) slot = row_ctz<W>(cand);
      }
      u64 fit = group_ballot<GS>(slot >= 0, lane);
      int l = fit ? (int)__builtin_ctzll(fit) : 0;
      int sl = group_get<GS>(slot, l, lane);
      if (fit) { best = base + l; bslot = sl; }
    }
    if (best >= 0) {
      int pth = best / P.C;
      a[0] = pth; a[1] = P.path_rec[(size_t)(pb + pth) * 32 + 1]; a[2] = best - pth * P.C; a[3] = bslot;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// step(): everything between receiving the action and handing back (reward, done, info)
// ---------------------------------------------------------------------------------------------
template <int ENV, int W, bool EVL>
__device__ __forceinline__ void step(const DevParams& P, Env& e, int lane, const int* act, int auto_reset,
                                     double* reward_out, unsigned char* done_out, double* info_out, double* obs_out,
                                     double* term_obs_out, const Rng* prefilled, const Prefetch* pf = nullptr) {
  const int K = P.K, S = P.S, rej = P.allow_rejection ? 1 : 0;
  if (ENV == ENV_QOS) {  // QoSConstrainedRA.step (qos_constrained_ra.py:100-157)
    int a = act[0];
    const bool badq = a < 0 || a >= K + rej;  // actions_output[action] += 1 raises IndexError
    if (badq) { e.flags |= ORL_FLAG_BAD_ACTION; a = K; }
    const int clazz = e.bit_rate, np_ = P.n_paths[e.src * P.N + e.dst];
    bool accepted = false;
    if (!badq && ((clazz == 0 && a == 0) || (clazz != 0 && a < np_))) {
      const int pidx = pair_base(P, e.src, e.dst) + a;
      const PathRec prec = path_rec_load(P, pidx);
      if (qos_path_free(P, e, lane, prec)) {
        qos_path_apply(P, e, lane, prec, false);
        e.sa += 1;
        e.esa += 1;
        accepted = true;
        ev_push<EVL>(P, e, lane, e.at + e.ht, ev_pack(pidx, 0, 1, 0, 0));
      }
    }
    e.sp += 1;
    e.esp += 1;
    const double rew = accepted ? P.class_reward[clazz] : 0.0;
    if (info_out && lane == 0) {
      info_out[0] = (double)(e.sp - e.sa) / (double)e.sp;
      info_out[1] = (double)(e.esp - e.esa) / (double)e.esp;
    }
    e.new_service = 0;
    next_service<ENV, W, EVL>(P, e, lane, prefilled, pf);
    const bool doneq = (e.esp == (i64)P.episode_length);
    if (P.ep_log && P.ep_rew && lane == 0) {
      const double acc = P.ep_rew_acc[e.env] + rew;
      if (doneq) {
        const int idx = P.ep_count[e.env];
        if (idx < P.ep_cap) P.ep_rew[e.env * P.ep_cap + idx] = acc;
      }
      P.ep_rew_acc[e.env] = doneq ? 0.0 : acc;
    }
    if (doneq && P.ep_log && lane == 0) episode_log(P, e.env, e.esa);
    if (doneq && auto_reset) soft_reset<ENV>(e);
    if (lane == 0) {
      if (reward_out) *reward_out = rew;
      if (done_out) *done_out = doneq ? 1 : 0;
    }
    return;
  }
  int path, slot, mod = 0, core = 0;
  bool bad = false;
  if (ENV == ENV_DEEPRMSA) {  // deeprmsa_env.py:48-58
    int aa = act[0];
    path = K; slot = S;
    if (aa >= 0 && aa < K * P.J) {
      int route = aa / P.J, block = aa - route * P.J;
      int start = 0;
      int pidx = pair_base(P, e.src, e.dst) + route;
      int nb = 0;
      if (route < P.n_paths[e.src * P.N + e.dst]) {
        Row<W> m = path_and<W>(P, e, pidx, 0);
        nb = nth_block<W>(m, S, P.nslots_path[(size_t)pidx * P.n_br + e.br_idx], block + 1, start);
      }
      if (block < nb) { path = route; slot = start; }
    }
  } else if (ENV == ENV_RMCSA) {
    path = act[0]; mod = act[1]; core = act[2]; slot = act[3];
    bad = path < 0 || path > K || mod < 0 || mod > P.M || core < 0 || core > P.C || slot < 0 || slot > S;
  } else if (ENV == ENV_RWA) {
    path = act[0]; slot = act[1];
    bad = path < 0 || path >= K + rej || slot < 0 || slot >= S + rej;
  } else {
    path = act[0]; slot = act[1];
    bad = path < 0 || path > K || slot < 0 || slot > S;
  }
  if (bad) {  // the reference raises IndexError on actions_output[...]; flag it and treat as a rejection
    e.flags |= ORL_FLAG_BAD_ACTION;
    path = K; slot = S; mod = P.M; core = P.C;
  }
  const int path0 = path, slot0 = slot, mod0 = mod, core0 = core;
  double prev_comp = 0.0, cur_comp = 0.0;
  if ((ENV == ENV_RMSA || ENV == ENV_DEEPRMSA) && info_out) prev_comp = net_compactness(P, e, 0);
  bool accepted = false;
  bool in_range = (ENV == ENV_RMCSA) ? (path < K && mod < P.M && core < P.C && slot < S) : (path < K && slot < S);
  if (in_range && path < P.n_paths[e.src * P.N + e.dst]) {
    int pidx = pair_base(P, e.src, e.dst) + path;
    int n = 1;
    const bool hit = pf && pf->have_rec && pf->pidx == pidx;  // the record requested at kernel entry is this path's
    PathRec prec;
    if (hit) { prec.q[0] = pf->rq0; prec.q[1] = pf->rq1; prec.q[2] = pf->rq2; prec.q[3] = pf->rq3; }
    else prec = path_rec_load(P, pidx);
    if (ENV == ENV_RMCSA) n = P.nslots[e.br_idx * P.M + mod];
    else if (ENV != ENV_RWA) n = hit ? pf->nslots : (int)P.nslots_path[(size_t)pidx * P.n_br + e.br_idx];
    bool ok = path_is_free<W>(P, e, lane, prec, core, slot, n);
    if (ok && ENV == ENV_RMCSA) {  // _crosstalk_is_acceptable: two reach limits
      double len = P.path_length[pidx];
      ok = (len < P.lmax_xt[mod]) && (len < P.lmax_snr[mod * P.n_br + e.br_idx]);
    }
    if (ok) {
      int hops_p = path_apply<ENV, W>(P, e, lane, prec, core, slot, n, false);
      e.s_br += e.bit_rate;
      e.s_nh += (i64)n * hops_p;
      if (ENV != ENV_RWA) {  // _update_network_stats
        double last_update = e.g_last, time_diff = e.now - last_update;
        if (e.now > 0) {
          double cur_thr = (double)e.s_br;
          e.g_thr = ((e.g_thr * last_update) + (cur_thr * time_diff)) / e.now;
          e.g_comp = ((e.g_comp * last_update) + (net_compactness(P, e, core) * time_diff)) / e.now;
        }
        e.g_last = e.now;
        e.brp += e.bit_rate;
        e.ebrp += e.bit_rate;
        if (P.bit_rate_mode == 1 && lane == 0) P.br_hist[e.env * 2 * P.n_br + P.n_br + e.br_idx] += 1;
      }
      e.sa += 1;
      e.esa += 1;
      accepted = true;
      ev_push<EVL>(P, e, lane, e.at + e.ht, ev_pack(pidx, slot, n, core, e.bit_rate));
    }
  }
  if (ENV == ENV_RWA) { e.sp += 1; e.esp += 1; }
  if (ENV == ENV_RMCSA) { e.sp += 1; e.esp += 1; e.brq += e.bit_rate; e.ebrq += e.bit_rate; }
  if ((ENV == ENV_RMSA || ENV == ENV_DEEPRMSA) && info_out) cur_comp = net_compactness(P, e, 0);
  if (ENV != ENV_RMCSA && P.act2d && !bad && lane == 0) act2d_count(P, e.env, path0, slot0, accepted);
  if (ENV == ENV_RMCSA && P.act2d && !bad && lane == 0) act4d_count(P, e.env, path0, mod0, core0, slot0, accepted);

  if (ENV == ENV_RWA) {
    // actions_output marginals (rwa_env.py:103, 148-151).  Each lane owns histogram entries, applies this
    // step's increment itself and derives its info value from the updated count (total = services_processed).
    i64* h = P.act_hist + e.env * ((K + 1) + (S + 1));
    const int npa = K + rej, nsa = S + rej;
    for (int base = 0; base < npa + nsa; base += 64) {
      int i = base + lane;
      if (i < npa + nsa) {
        int hi = (i < npa) ? i : (K + 1) + (i - npa);
        bool hit = !bad && ((i < npa) ? (i == path0) : (i - npa == slot0));
        i64 v = h[hi] + (hit ? 1 : 0);
        if (hit) h[hi] = v;
        if (info_out) info_out[2 + i] = (double)v / (double)e.sp;
      }
    }
  }
  double reward = accepted ? 1.0 : (ENV == ENV_DEEPRMSA ? -1.0 : 0.0);
  if (info_out) {
    double i0 = (double)(e.sp - e.sa) / (double)e.sp;
    double i1 = (double)(e.esp - e.esa) / (double)e.esp;
    if (lane == 0) { info_out[0] = i0; info_out[1] = i1; }
    if (ENV != ENV_RWA) {
      double i2 = (double)(e.brq - e.brp) / (double)e.brq;
      double i3 = (double)(e.ebrq - e.ebrp) / (double)e.ebrq;
      if (lane == 0) { info_out[2] = i2; info_out[3] = i3; }
    }
    if (ENV == ENV_RMSA || ENV == ENV_DEEPRMSA) {
      double* scratch = e.scratch;
      double mc = link_mean(P, e.ls + 2, scratch, lane);
      double mu = link_mean(P, e.ls, scratch, lane);
      if (lane == 0) { info_out[4] = cur_comp; info_out[5] = prev_comp - cur_comp; info_out[6] = mc; info_out[7] = mu; }
      if (P.bit_rate_mode == 1 && lane == 0) {  // rmsa_env.py:217-227, 268-273
        const i64* rq = P.br_hist + e.env * 2 * P.n_br;
        const i64* pv = rq + P.n_br;
        double mxv = -__builtin_inf(), mnv = __builtin_inf();
        for (int i = 0; i < P.n_br; i++) {
          double bl = 0.0;
          if (rq[i] > 0) bl = (double)(rq[i] - pv[i]) / (double)rq[i];
          info_out[8 + i] = bl;
          mxv = bl > mxv ? bl : mxv;
          mnv = bl < mnv ? bl : mnv;
        }
        info_out[8 + P.n_br] = mxv - mnv;
      }
    }
  }
  e.new_service = 0;
  next_service<ENV, W, EVL>(P, e, lane, prefilled, pf);
  bool done = (e.esp == (i64)P.episode_length);
  if (done && P.ep_log && lane == 0) episode_log(P, e.env, e.esa);
  if (ENV == ENV_DEEPRMSA && obs_out) {
    deep_observation<W>(P, e, lane, obs_out, (done && term_obs_out) ? term_obs_out : nullptr);
  }
  if (done && auto_reset) soft_reset<ENV>(e);
  if (lane == 0) {
    if (reward_out) *reward_out = reward;
    if (done_out) *done_out = done ? 1 : 0;
  }
}

}  // namespace orl
