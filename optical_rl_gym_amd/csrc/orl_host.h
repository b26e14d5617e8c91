// orl_host.h — host-side structures shared by the translation units of liborlgpu.so.
//
// The library is built from one W-independent unit (orl_api.hip: the C ABI of include/orl.h, batch construction, the small
// utility kernels) and one unit per row width W in {1, 2, 5, 8} 64-bit words (orl_kernels.hip compiled with -DORL_W=W: the
// env kernels, which keep a whole link row in registers and are therefore templates over W).  The units compile in
// parallel (optical_rl_gym_amd/_build.py); the API unit reaches the kernels through the launchers declared below.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "../../include/orl.h"
#include "orl_device.h"

// Deferred statistics of the persistent kernel (orl_device_split.h ctrl_d, orl_kernels.hip k_stats): the per-env bookkeeping of a launch is
// logged and replayed lane-per-env afterwards (RMSA, DeepRMSA, RWA, RMCSA).  -DORL_PERSIST_DS=0 keeps it in the loop.
#ifndef ORL_PERSIST_DS
#define ORL_PERSIST_DS 1
#endif
#ifndef ORL_PERSIST_SVC
#define ORL_PERSIST_SVC 1
#endif
static inline bool orl_persist_deferred(int env_type) {
  return ORL_PERSIST_DS != 0 && ORL_PERSIST_SVC != 0 &&
         (env_type == orl::ENV_RMSA || env_type == orl::ENV_DEEPRMSA || env_type == orl::ENV_RWA || env_type == orl::ENV_RMCSA);
}

struct orl_topology {
  int device;
  int N, E, K, H, M;
  std::vector<int32_t> h_hops, h_links, h_mod;  // host copies (per-batch derived tables are built from them)
  int* n_paths;
  double* path_length;
  int* edge_iter_order;
  int* link_pos;
};

// per-kernel timing (orl_batch_run, time_kernels == 1): an event after every launch
struct TkRec { std::vector<hipEvent_t> ev; std::vector<const char*> name; };

struct orl_batch {
  orl::DevParams P;
  TkRec* tk = nullptr;
  int parity = 0;        // two-kernel form (ORL_ALT_IMPLS): which deferred-env buffer the next step writes
  int persist = 0;       // device-resident runs go through the persistent kernel (k_persist)
  int lds_state = 0;     // ... with the slot maps and link statistics of a wavefront's envs resident in LDS
  int agent_step = 0;    // host- / agent-driven steps with auto reset go through k_agent (the phases of k_persist) instead of k_step
  int two_kernel = 0;    // ORL_ALT_IMPLS builds, ORL_STEP_IMPL=2 ORL_PERSIST=0: the phases of k_persist as separate launches
  int64_t persist_launches = 0;
  int persist_spec = 0;            // 1: the last launch of k_persist used the instantiation built for this configuration
  int persist_form_last = -1;      // the form (index into kPersistForms) of the last launch of k_persist
  // a specialisation library attached by orl_batch_load_spec: k_persist with this batch's sizes as compile-time constants
  void* spec_handle = nullptr;
  void (*spec_launch)(const orl::DevParams*, unsigned, size_t, hipStream_t, int, int, int*, unsigned int*, unsigned int*) = nullptr;
  void (*spec_agent_launch)(const orl::DevParams*, unsigned, size_t, hipStream_t, int, int) = nullptr;  // k_agent of the same library
  int spec_lds = -1, spec_waves = -1;
  int* d_wg_step = nullptr;        // [ceil(B/8)] steps each workgroup of the persistent kernel has completed since run_base was 0
  int64_t run_base = 0;            // ... all of them, between runs (no per-run clearing of d_wg_step)
  bool wg_dirty = true;            // a run did not complete (or none has run yet): clear d_wg_step and run_base first
  bool run_abandoned = false;      // a device-resident run returned early (HIP error): services may be parked, envs at different steps
  int un_slot[2] = {0, 0};         // per half of the batch: the slot of d_unfinished its next launch counts into
  // per half p and slot s, at 8 p + 4 s: [0] straggler workgroups of a persistent launch, [1] OR of the env flag words
  // (k_finish2); a launch counts into one slot and clears the other for the launch after it (no memset between launches);
  // [16..17]: the same pair for report_flags
  unsigned int* d_unfinished = nullptr;
  int device = 0, wt = 0;
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;   // second half of the batch in device-resident runs (see orl_batch_run)
  hipEvent_t ev_half = nullptr;
  int n_cu = 256;
  std::vector<void*> allocs;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  unsigned long long* d_totals = nullptr;
  int32_t* h_actions = nullptr;    // page-locked [B][4]: where orl_batch_step_async expands the caller's compact action rows
  int step_pending = 0;            // orl_batch_step_async queued a step that orl_batch_step_wait has not collected yet
  unsigned int* h_tail = nullptr;  // page-locked: where the straggler count / flag words of a run land (a pageable target is staged)
  int log_cap = 0;                 // deferred statistics: steps one launch of k_persist can log per wavefront (0: no log)
  int cache_epoch = 1;             // bumped by every call that may change slot maps outside the persistent kernel (DevParams::row_cache_key)
  long long* gather_idx = nullptr;  // orl_batch_get_info_rows: row indices and gathered rows on the device, grown on demand
  double* gather_out = nullptr;
  int64_t gather_cap = 0;
  float* obs_f32 = nullptr;        // device copy of the observation array in float32 (orl_batch_get_obs_f32), allocated on first use
  int* ep_buf = nullptr;           // episode log buffer (orl_batch_episode_log): [B][ep_alloc] ints, armed with stride P.ep_cap <= ep_alloc
  int ep_alloc = 0;
  double* ep_rew_buf = nullptr;    // QoSConstrainedRA: the float64 reward sums beside it, same shape
};

#define ORL_TK(B_, NAME)                                                                 \
  do {                                                                                   \
    if ((B_)->tk) {                                                                      \
      hipEvent_t e_;                                                                     \
      hipEventCreate(&e_);                                                               \
      hipEventRecord(e_, (B_)->stream);                                                  \
      (B_)->tk->ev.push_back(e_);                                                        \
      (B_)->tk->name.push_back(NAME);                                                    \
    }                                                                                    \
  } while (0)

// ---- launchers, one explicit instantiation per W (orl_kernels.hip) ----------------------------------------------------
namespace orl_launch {
template <int W> void reset(orl_batch* b, int full, const unsigned char* dmask);
template <int W> void policy(orl_batch* b, int pol);                       // stand-alone slot scan -> P.actions
template <int W> void step64(orl_batch* b, int auto_reset, int want_info, int fused_policy);  // one wavefront per env
template <int W> void obs(orl_batch* b, int with_terminal);                // DeepRMSA observation
// k_persist over the env range of view VP up to step `target` of this run, then k_rel_tail, on stream st
template <int W> void persist(orl_batch* b, const orl::DevParams& VP, hipStream_t st, int pol, int target, int* wg_step, unsigned int* unfinished,
                              unsigned int* clear_next, int finish);  // finish: this launch ends the run (DevParams::persist_finish)
template <int W> int persist_resident(orl_batch* b, int n_cu);              // wavefronts of k_persist the GPU holds at once
template <int W> int persist_uses_lds(orl_batch* b);                       // 1: the persistent kernel keeps slot maps / link statistics in LDS
template <int W> int prof_read(unsigned long long* out48, int reset);      // -DORL_TIMING builds: per-phase cycle sums
template <int W> void step2(orl_batch* b, int pol);
template <int W> void agent_step(orl_batch* b, int auto_reset, int pol);
template <int W> void persist_form(const orl::DevParams& VP, int* lds_state, int* waves);  // the form persist() takes for this configuration                            // k_agent: one step, actions in P.actions, info / obs written                        // ORL_ALT_IMPLS: k_step_a2 ; k_rows2 ; k_rel_tail
}  // namespace orl_launch

#define ORL_DISPATCH_W(B_, CALL)      \
  switch ((B_)->wt) {                 \
    case 1: CALL(1); break;           \
    case 2: CALL(2); break;           \
    case 5: CALL(5); break;           \
    default: CALL(8); break;          \
  }
