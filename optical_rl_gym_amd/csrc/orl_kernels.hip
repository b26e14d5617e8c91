// orl_kernels.hip — the env kernels of liborlgpu.so for ONE row width W (compiled once per -DORL_W=1|2|5|8; gfx950 only).
//
//   k_reset     full reset (clear network, draw first service) or soft reset (episode counters)     one wavefront per env
//   k_policy    stand-alone slot scan: AND the link rows of each of the k paths, log-step run detection, first fit -> action
//               [the HBM-streaming kernel; 8 lanes per env, lane = path]
//   k_step      host-driven step(): apply action, statistics, reward/info, next service, observation   one wavefront per env
//   k_persist   device-resident loop: one wavefront owns 8 envs for a whole run and alternates a control phase (slot scan,
//               validation, counters, next service, due releases -> work items) and a row phase (one lane per touched link)
//   k_rel_tail  two-kernel test form: the rare envs whose releases of a step did not fit the item form release them in place
//               (k_persist does that itself at the start of the owning wavefront's next launch)
//   k_obs8/k_obs DeepRMSA observation of the pending service
// Launchers (orl_launch::*<W>) are explicitly instantiated at the end; orl_api.hip dispatches on the batch's W.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#include "orl_host.h"
#include "orl_device_g8.h"
#include "orl_device_split.h"

#ifndef ORL_W
#error "compile with -DORL_W=1|2|5|8"
#endif

using namespace orl;

extern __shared__ __attribute__((aligned(16))) unsigned char orl_lds_raw[];

template <int ENV, int W>
__global__ void __launch_bounds__(64) k_reset(DevParams P, int full, const unsigned char* mask) {
  const i64 env = blockIdx.x;
  const int lane = lane_id();
  if (mask && !mask[env]) return;
  // (services the persistent kernel drew ahead for a run that was abandoned — an overflow error — are dropped with the reset)
  if (lane < 8) P.svc_cnt[(env >> 3) * 64 + (env & 7) * 8 + lane] = 0;
#ifndef ORL_X_NO_STAMP
  if (full && lane == 8 && P.row_cache_stamp) P.row_cache_stamp[env >> 3] = 0;  // (the slot maps change: stored row caches are stale)
#endif
  Env e;
  env_load(P, e, env, lane);
  if (!full) {
    soft_reset<ENV>(e);
    env_store(P, e, lane);
    return;
  }
  u64* lds = (u64*)orl_lds_raw;
  e.bm = lds;
  e.ls = (double*)(lds + P.bm_words);
  e.scratch = e.ls + 4 * P.E;
  e.obs_l = e.scratch + P.E;
  e.cs = (int*)(e.obs_l + P.obs_dim);
  // available_slots = ones (rmsa_env.py:337-339); bits >= S stay 0
  for (int i = lane; i < P.bm_words; i += 64) {
    int w = i % W;
    int c = P.S - 64 * w;
    u64 v = c >= 64 ? ~0ull : (c <= 0 ? 0ull : ((1ull << c) - 1ull));
    if (ENV == ENV_QOS) v = (u64)P.S;  // available_spectrum = num_spectrum_resources per link (optical_network_env.py:189-193)
    lds[i] = (i < P.C * P.E * W) ? v : 0ull;
  }
  for (int i = lane; i < 4 * P.E; i += 64) e.ls[i] = 0.0;
  for (int i = lane; i < P.cs_words; i += 64) e.cs[i] = 0;
  for (int i = lane; i < P.ev_cap; i += 64) e.ev_time[i] = __builtin_inf();
  if (P.br_hist) for (int i = lane; i < 2 * P.n_br; i += 64) P.br_hist[env * 2 * P.n_br + i] = 0;
  if (P.act_hist) for (int i = lane; i < (P.K + 1) + (P.S + 1); i += 64) P.act_hist[env * ((P.K + 1) + (P.S + 1)) + i] = 0;
  // RWAEnv.reset and RMCSAEnv.reset clear actions_output / actions_taken (rwa_env.py:194-203, rmcsa_env.py:437-454); RMSAEnv.reset
  // never does (rmsa_env.py:284-359)
  if (P.act2d && (ENV == ENV_RWA || ENV == ENV_RMCSA)) for (int i = lane; i < P.act2d_words; i += 64) P.act2d[env * P.act2d_words + i] = 0;
  wave_fence();
  e.now = 0; e.at = 0; e.ht = 0; e.g_thr = 0; e.g_comp = 0; e.g_last = 0;
  e.sp = e.sa = e.esp = e.esa = e.brq = e.brp = e.ebrq = e.ebrp = e.s_br = e.s_nh = 0;
  e.src = e.dst = e.bit_rate = e.br_idx = e.id = 0;
  e.ev_hwm = 0; e.ev_cnt = 0; e.new_service = 0; e.flags &= ORL_FLAG_MT2;  // (a full reset keeps the Random objects)
  next_service<ENV, W, false>(P, e, lane, nullptr);
  stage_out(P, e, lane);
  env_store(P, e, lane);
}

// Slot-scan kernel.  GS lanes per env: a wavefront serves 64/GS envs whose slot maps are contiguous in HBM.  Link rows are
// read straight from global memory (each lane the 8W-byte rows of its own path).  A/B on MI355X, cfg2, B = 65 536: 17.2 us per
// launch vs 19.4 us with the maps staged through LDS first — the 28 KB/workgroup LDS window capped residency at 5
// workgroups/CU and cost a second dispatch round, while the 7-KB-per-wave footprint stays L2/TCP resident.
#ifndef ORL_POLICY_WAVES
#define ORL_POLICY_WAVES 4  // wavefronts per workgroup in the slot-scan kernel
#endif
template <int ENV, int W, int GS>
__global__ void __launch_bounds__(64 * ORL_POLICY_WAVES) k_policy(DevParams P, int pol) {
  constexpr int EPW = 64 / GS;  // envs per wavefront
  const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
  const i64 env0 = ((i64)blockIdx.x * ORL_POLICY_WAVES + wave) * EPW;
  if (env0 >= P.B) return;
  const u64* maps = P.bitmap + env0 * P.bm_words;
  const int grp = lane / GS;
  const i64 env = env0 + grp;
  const bool valid = env < P.B;
  u64 d = valid ? P.svc_desc[env] : 0ull;
  int a[4];
  policy_g<ENV, W, GS>(P, maps + (size_t)grp * P.bm_words, valid, (int)(u32)d, (int)((d >> 32) & 0xffffu), (int)((d >> 48) & 0xffu),
                       lane, pol, valid ? P.path_col[env] : 0, a);
  // (DeepRMSA: columns 1-2 of the scan result are its decoded route / slot for a control phase in the same kernel; the action is column 0)
  if (valid && (lane & (GS - 1)) == 0) *(int4*)(P.actions + env * 4) = (ENV == ENV_DEEPRMSA) ? make_int4(a[0], 0, 0, 0) : make_int4(a[0], a[1], a[2], a[3]);
}

#ifndef ORL_STEP_WAVES
#define ORL_STEP_WAVES 5  // waves per SIMD the register allocator must leave room for (measured: 4 -> 428 us, 5 -> 389 us, 6 -> 436 us)
#endif
template <int ENV, int W, bool EVL>
__global__ void __launch_bounds__(64, ORL_STEP_WAVES) k_step(DevParams P, int auto_reset, int want_info, int pol) {
  const i64 env = blockIdx.x;
  const int lane = lane_id();
#ifndef ORL_X_NO_STAMP
  if (lane == 0 && P.row_cache_stamp) P.row_cache_stamp[env >> 3] = 0;  // (as k_agent: the persistent kernel's row caches of these envs are stale)
#endif
  Env e;
  // Round trip 1: everything addressed by the env index alone is requested before anything is waited for — the
  // scalar record, the action, the slot map / link statistics / per-core sums (into LDS), the source-node table.
  const u64 sv = env_fetch(P, env, lane);
  int4 av = (pol < 0) ? *(const int4*)(P.actions + env * 4) : make_int4(0, 0, 0, 0);
  Prefetch pf;
  pf.have_cum = P.N <= 64;
  pf.cum_my = pf.have_cum ? P.cum_src[lane < P.N - 1 ? lane : P.N - 1] : 0.0;
  e.env = env;
  stage_in(P, e, (u64*)orl_lds_raw, lane);
  env_unpack(P, e, env, lane, sv);
  if (pol >= 0) {
    // device-resident loop of the batches the persistent kernel does not take: the slot scan runs right here on the LDS copy
    // of the slot map (lanes = paths, or (path, core) pairs) — one launch and one read of the map per policy + step
    int a[4];
    policy_g<ENV, W, 64>(P, e.bm, true, pair_base(P, e.src, e.dst), e.br_idx, P.n_paths[e.src * P.N + e.dst], lane, pol,
                         P.path_col[env], a);
    av = (ENV == ENV_DEEPRMSA) ? make_int4(a[0], 0, 0, 0) : make_int4(a[0], a[1], a[2], a[3]);
    if (lane == 0) *(int4*)(P.actions + env * 4) = av;
  }
  // Round trip 2: what the scalar record addresses — the MT window of the next service, the pending release times
  // (EVL: into LDS for all scans) and the path record + slot count of the action's path.
  Rng pre;
  rng_fill(e, pre, lane);
  {
    const int route = (ENV == ENV_DEEPRMSA) ? (av.x >= 0 ? av.x / P.J : P.K) : av.x;  // (QoS: the action is the path)
    pf.have_rec = route >= 0 && route < P.K;
    pf.pidx = pair_base(P, e.src, e.dst) + (pf.have_rec ? route : 0);
    const PathRec r0 = path_rec_load(P, pf.pidx);
    pf.rq0 = r0.q[0]; pf.rq1 = r0.q[1]; pf.rq2 = r0.q[2]; pf.rq3 = r0.q[3];
    pf.nslots = P.nslots_path[(size_t)pf.pidx * P.n_br + e.br_idx];
  }
  if (EVL) {
    e.evl = (double*)((unsigned char*)orl_lds_raw + P.lds_bytes);
    for (int i = lane; i < e.ev_hwm; i += 64) e.evl[i] = e.ev_time[i];
    wave_fence();
  }
  int act[4] = {av.x, av.y, av.z, av.w};
  step<ENV, W, EVL>(P, e, lane, act, auto_reset, P.reward + env, P.done + env,
                    want_info ? P.info + env * P.n_info : nullptr,
                    P.obs_dim ? P.obs + env * P.obs_dim : nullptr,
                    P.obs_dim ? P.term_obs + env * P.obs_dim : nullptr, &pre, &pf);
  stage_out(P, e, lane);
  env_store(P, e, lane);
}

template <int W>
__device__ __forceinline__ void obs8_env(const DevParams& P, const u64* bm, const u64* rec, i64 env, int lane, int terminal);
template <int W>
__device__ __forceinline__ void obs8_env_w(const DevParams& P, const u64* bm, u64 sd, u64 br, i64 env, int lane, int terminal);

#ifdef ORL_ALT_IMPLS
// ---- two-kernel form of the persistent kernel's phases (cross-checks, per-kernel timing): k_step_a2 ; k_rows2 ------------
template <int ENV, int W, bool FUSED_POLICY>
__global__ void __launch_bounds__(256) k_step_a2(DevParams P, int pol, int parity) {
  __shared__ u32 s_tally[32 * 32];
  const int lane = lane_id();
  const i64 env = (i64)blockIdx.x * 32 + (threadIdx.x >> 3);
  const bool valid = env < P.B;
  sp::Prof prof;
  const sp::Wmem M = sp::wmem_global(P);
  sp::CtrlOpts O;
  O.persistent = false; O.write_io = true; O.trusted = false; O.emit_queue = true; O.prefetch = false; O.auto_reset = true; O.rank_pairs = false;
  ORL_PROFA_BEGIN();
  if (FUSED_POLICY) {
    const i64 env0 = env - ((lane >> 3));  // first env of this wavefront
    u64 d = valid ? P.svc_desc[env] : 0ull;
    int a[4];
    policy_g<ENV, W, 8>(P, P.bitmap + env0 * P.bm_words + (size_t)(lane >> 3) * P.bm_words, valid, (int)(u32)d,
                        (int)((d >> 32) & 0xffffu), (int)((d >> 48) & 0xffu), lane, pol, valid ? P.path_col[env] : 0, a);
    const int4 av = make_int4(a[0], a[1], a[2], a[3]);
    ORL_PROFA(1);
    O.trusted = true;
    sp::ctrl_a<ENV, W>(P, M, O, env, valid, lane, prof, &av, s_tally, (sp::SinkEntry*)orl_lds_raw, parity, nullptr, nullptr);
  } else {
    sp::ctrl_a<ENV, W>(P, M, O, env, valid, lane, prof, nullptr, s_tally, (sp::SinkEntry*)orl_lds_raw, parity, nullptr, nullptr);
  }
  ORL_PROFA_END();
}
// One lane per mixed item; one 192-thread workgroup per control workgroup (32 envs, ~140 items): a thread handles at most one
// item in all but one launch in 10^3, so the kernel's duration is one item's dependent chain.
#define ORL_ROWS2_THREADS 192
template <int ENV, int W>
__global__ void __launch_bounds__(ORL_ROWS2_THREADS) k_rows2(DevParams P, int parity) {
  constexpr int NR = 4;
  sp::Prof prof;
  ORL_PROFR_BEGIN();
  const u32 r0 = blockIdx.x * NR;
  const u32 n_regions = (u32)((P.B + 31) / 32) * 4u;
  const u32* cnt = P.q_cnt_a + r0;
  u32 cum[NR + 1];
  cum[0] = 0;
#pragma unroll
  for (int j = 0; j < NR; j++) cum[j + 1] = cum[j] + ((r0 + j < n_regions) ? cnt[j] : 0u);
  const ulonglong2* q = P.q_a + (size_t)r0 * P.q_wave * 2;
  for (u32 idx = threadIdx.x; idx < cum[NR]; idx += ORL_ROWS2_THREADS) {
    u32 j = 0, base = 0;
#pragma unroll
    for (int t = 1; t < NR; t++)
      if (idx >= cum[t]) { j = (u32)t; base = cum[t]; }
    const size_t at = (size_t)j * P.q_wave + (idx - base);
    ORL_PROFR(1);
    sp::Item it;
    it.a = q[2 * at];
    it.b = q[2 * at + 1];
    ORL_PROFR(2);
    sp::row_item_lane<ENV, W>(P, sp::wmem_global(P), it, prof);  // (the general form: one lane walks all masks of the item)
  }
  ORL_PROFR(8);
  ORL_PROFR_END();
}
#endif  // ORL_ALT_IMPLS

// ---- persistent kernel -------------------------------------------------------------------------------------------------
// Envs never interact, so a wavefront can own its 8 envs for a whole launch (128 steps): control phase -> row phase over the
// items the wavefront itself just collected (one lane per touched link) -> next step, with no kernel boundary and no
// grid-wide tail between the phases; the wavefronts of a launch drift out of phase and keep the memory system uniformly busy.
// LDS forms (the 8 envs' slot maps, records and per-core sums fit the wavefront's share of the CU's LDS): that state is
// loaded once per launch, every scan / validation / row update of the launch works on LDS, and it is written back at the
// end — the slot map is read ~14 times and rewritten ~5 times per env-step, none of which reaches memory any more.  Work
// items never leave LDS either (the sink table is read in place through a dense index list).
// A wavefront in which an env's releases did not fit the item form (one env-step in 10^7) leaves the loop after that step's
// row phase and counts as unfinished; at the start of its next launch it releases them in place and resumes from its own
// step count.
struct PersistLds {  // byte offsets into the workgroup's dynamic LDS window (all multiples of 16)
  int tab, mtab, tally, tw, list, clk, misc, bm, ls, cs, csw, sc, ic, mini, total;  // (ic: inner-run cache, then the occ / fb cache)
};
// state: 0 = only the per-step tables, 1 = + slot maps, per-core sums and env records, 2 = + link statistics, 3 = slot maps and
// per-core sums but the env records stay in global memory (the window of the 4-wave forms: cfg2 8 832 B, 16 per CU);
// compact: the bit-word sink of the single-core families (4 bytes per link and env + a mask table per env);
// inner: 0 = no row caches, 1 = the per-word longest-run cache of every row, 2 = + every row's contribution to the compactness
// sums, (occ << 16) | free blocks (4 bytes per row each)
// mini: the eight record words the deferred-statistics control phase works on, for the forms whose records stay in global memory
// rd: the rows-deferred form (round 6) — no row phase in the loop, hence no sink table, mask table, item list, clock pairs, row
// caches or per-core sums: the window is the slot maps, the record words the control phase works on, and 16 bytes
__host__ __device__ inline PersistLds persist_lds_layout(int E, int H, int bm_words, int C, int state, bool compact, int inner, bool mini = false,
                                                         bool rd = false) {
  PersistLds L;
  int o = 0;
  L.tab = o; if (!rd) o += (8 * E * (int)(compact ? sizeof(sp::SinkEntryC) : sizeof(sp::SinkEntry)) + 15) & ~15;
  L.mtab = o; if (compact && !rd) o += 8 * ORL_MTAB * 2;
  L.tw = compact ? 0 : (E + 3) >> 2;
  L.tally = o; o += 8 * L.tw * 4;
  // one entry per touched link and env, a second one where the step's provision meets a release (at most its hops)
  L.list = o; if (!rd) o += ((8 * (E + (H < E ? H : E)) * 2) + 15) & ~15;
  // {provision clock, step clock} of the 8 envs for the row phase (the deferred-statistics control phase never writes SC_NOWA,
  // which the replay owns, so the forms with the records in LDS have the pair too; the full-LDS test form reads the records)
  L.clk = o; if (state != 2 && !rd) o += 8 * 2 * 8;
  L.misc = o; o += 16;
  L.bm = o; if (state >= 1) o += 8 * bm_words * 8;
  L.csw = (4 * C + 3) & ~3;  // sums + their release part, ints per env
  L.cs = o; if ((state >= 1 || mini) && !rd) o += 8 * L.csw * 4;  // (the global-state form of the deferred-statistics kernel keeps them in LDS too)
  L.sc = o; if (state == 1 || state == 2) o += 8 * ORL_SCAL_LDS_WORDS * 8;
  L.ic = o; if (state >= 1 && inner && !rd) o += inner * ((8 * E * 4 + 15) & ~15);
  L.ls = o; if (state == 2) o += 8 * E * 32;
  L.mini = o; if (mini && (state == 0 || state == 3)) o += (8 * ORL_MINI_STRIDE * 8 + 15) & ~15;
  L.total = o;
  return L;
}
template <int ENV, int LDS> struct PersistCompact { static constexpr bool value = ENV != ENV_RMCSA; };
static inline bool persist_compact(int env_type, int state) { return env_type != ENV_RMCSA; }
// (rows of one or two words: searching both costs less than the bookkeeping — cfg3 measured 1.20e9 without, 1.01e9 with)
template <int ENV, int W, int LDS> struct PersistInner { static constexpr bool value = LDS >= 1 && W >= 3 && W <= 5 && (ENV == ENV_RMSA || ENV == ENV_DEEPRMSA); };
static inline bool persist_inner(int env_type, int W, int state) { return state >= 1 && W >= 3 && W <= 5 && (env_type == ENV_RMSA || env_type == ENV_DEEPRMSA); }
// The LDS window of a wavefront's 8 envs, filled from their (contiguous) global arrays.  Everything is REQUESTED before
// anything is waited for (blocks of 8 x 64 pieces of 16 bytes; the records and the sums with the first block).  A copy loop
// `l[i] = g[i]` compiles to load - wait - store per iteration: 13 dependent memory round trips at the start of every
// wavefront of every launch, all wavefronts of a generation at once — a fixed ~45 us per generation, which made a 20-step
// launch 15 % slower per step than a 300-step run.
typedef unsigned long long orl_u64x2 __attribute__((ext_vector_type(2)));  // (a native vector: selects stay in registers)
typedef int orl_i32x4 __attribute__((ext_vector_type(4)));
// SNAP (rows-deferred forms): the slot maps also go to DevParams::bitmap0 as they are read — the state the replay starts from
template <bool REC, bool SNAP = false>
__device__ __forceinline__ void persist_fill_window(const DevParams& P, i64 env0, int nenv, int lane, void* l_bm_, void* l_rec_, void* l_cs_,
                                                    int q /* 16-byte pieces of the sums per env */) {
  const orl_u64x2* g = (const orl_u64x2*)(P.bitmap + env0 * P.bm_words);
  orl_u64x2* g0 = SNAP ? (orl_u64x2*)(P.bitmap0 + env0 * P.bm_words) : nullptr;
  const orl_u64x2* gr = (const orl_u64x2*)(P.scal + env0 * ORL_SCAL_WORDS);
  orl_u64x2* l_bm = (orl_u64x2*)l_bm_;
  orl_u64x2* l_rec = (orl_u64x2*)l_rec_;
  orl_i32x4* l_cs = (orl_i32x4*)l_cs_;
  const int n_bm = nenv * (P.bm_words / 2), n_rec = nenv * (ORL_SCAL_WORDS / 2);
  for (int base = 0; base < n_bm; base += 64 * 8) {
    const int i0 = base + lane;
    orl_u64x2 b0 = 0, b1 = 0, b2 = 0, b3 = 0, b4 = 0, b5 = 0, b6 = 0, b7 = 0, r0 = 0, r1 = 0;
    orl_i32x4 vc = 0;
    if (i0 < n_bm) b0 = g[i0];
    if (i0 + 64 < n_bm) b1 = g[i0 + 64];
    if (i0 + 128 < n_bm) b2 = g[i0 + 128];
    if (i0 + 192 < n_bm) b3 = g[i0 + 192];
    if (i0 + 256 < n_bm) b4 = g[i0 + 256];
    if (i0 + 320 < n_bm) b5 = g[i0 + 320];
    if (i0 + 384 < n_bm) b6 = g[i0 + 384];
    if (i0 + 448 < n_bm) b7 = g[i0 + 448];
    if (base == 0) {
      if (REC) {
        if (lane < n_rec) r0 = gr[lane];
        if (lane + 64 < n_rec) r1 = gr[lane + 64];
      }
      if (lane < nenv * q) vc = ((const orl_i32x4*)(P.core_sums + (env0 + lane / q) * P.cs_words))[lane % q];
    }
    if (i0 < n_bm) l_bm[i0] = b0;
    if (i0 + 64 < n_bm) l_bm[i0 + 64] = b1;
    if (i0 + 128 < n_bm) l_bm[i0 + 128] = b2;
    if (i0 + 192 < n_bm) l_bm[i0 + 192] = b3;
    if (i0 + 256 < n_bm) l_bm[i0 + 256] = b4;
    if (i0 + 320 < n_bm) l_bm[i0 + 320] = b5;
    if (i0 + 384 < n_bm) l_bm[i0 + 384] = b6;
    if (i0 + 448 < n_bm) l_bm[i0 + 448] = b7;
    if constexpr (SNAP) {
      if (i0 < n_bm) g0[i0] = b0;
      if (i0 + 64 < n_bm) g0[i0 + 64] = b1;
      if (i0 + 128 < n_bm) g0[i0 + 128] = b2;
      if (i0 + 192 < n_bm) g0[i0 + 192] = b3;
      if (i0 + 256 < n_bm) g0[i0 + 256] = b4;
      if (i0 + 320 < n_bm) g0[i0 + 320] = b5;
      if (i0 + 384 < n_bm) g0[i0 + 384] = b6;
      if (i0 + 448 < n_bm) g0[i0 + 448] = b7;
    }
    if (base == 0) {
      if (REC) {
        // (16 pieces of 16 bytes per record, records ORL_SCAL_LDS_WORDS apart)
        if (lane < n_rec) l_rec[(lane >> 4) * (ORL_SCAL_LDS_WORDS / 2) + (lane & 15)] = r0;
        if (lane + 64 < n_rec) l_rec[((lane + 64) >> 4) * (ORL_SCAL_LDS_WORDS / 2) + (lane & 15)] = r1;
      }
      if (lane < nenv * q) l_cs[lane] = vc;
      for (int i = lane + 64; i < nenv * q; i += 64)  // (more than 8 cores: the rest of the sums)
        l_cs[i] = ((const orl_i32x4*)(P.core_sums + (env0 + i / q) * P.cs_words))[i % q];
    }
  }
}

// Deferred statistics (orl_device_split.h, ctrl_d): the persistent kernel of the single-core families leaves the per-env
// bookkeeping to k_stats below.  -DORL_PERSIST_DS=0 builds keep it in the loop (ctrl_a; A/B measurements, cross-checks).
// (the macros and orl_persist_deferred(): orl_host.h)
template <int ENV, int LDS> struct PersistDeferred {
  static constexpr bool value = ORL_PERSIST_DS != 0 && ORL_PERSIST_SVC != 0 && LDS != 2 &&
                                (ENV == ENV_RMSA || ENV == ENV_DEEPRMSA || ENV == ENV_RWA || ENV == ENV_RMCSA);
};

// The bookkeeping of the steps a launch of k_persist ran, one LANE per env: counters, bit-rate sums, the running averages of
// network throughput and compactness, episode ends — what ctrl_a / service_part (orl_device_split.h) do inside the step,
// statement for statement and in the reference's order (rmsa_env.py:163-282 decision and counters, 439-462
// _update_network_stats, 545-597 _next_service, 284-359 soft reset; rwa_env.py:101-162), from the three log words per step.
// Leaves the record exactly as that code would: a pending network-compactness update stays pending (SC_ACC bit 1 with its
// stashed factors) unless the wavefront finished the run's state (log_n bit 16), in which case it is finished here from the
// sums the wavefront logged after its last row phase, as k_finish2 does.
#ifndef ORL_STATS_BATCH
#define ORL_STATS_BATCH 8  // log slots requested per round trip
#endif
// (one lane per env, 64 envs per wavefront: spreading a batch over four times the wavefronts — 16 busy lanes each — measured
// slower, 143 us instead of 90 us behind a 128-step launch of 65 536 envs: the replay is bound by memory requests, and a
// request of 16 lanes carries a quarter of the bytes)
#define ORL_STATS_LANES 64
// RD: the launch ran a rows-deferred form — the compactness sums come from DevParams::ssum (k_rowstats), (occ << 16) | fb per step.
// (A template parameter: as a run-time switch the conditional loads of ssum among the unrolled log loads changed the results of the
// other forms in a few wavefronts per 4 096-env Germany50 batch — tools/pair_diff2.py — with nothing of ssum reaching a result.)
// hist_lds (RWA): the marginals of actions_output (rwa_env.py:103) are counted in LDS for the launch — one row of (k + 1) + (S + 1)
// counters per lane — and added to the env's histogram once at the end, with coalesced requests.  Two scattered 8-byte global updates
// per env-step (read-modify-writes until round 6, then atomics) ran at ~10 G updates/s: 0.83 ms behind a 128-step launch of 32 768
// envs, a quarter of cfg1's GPU time, where the other families' replay takes a tenth of that.
template <int ENV, bool RD = false>
__global__ void __launch_bounds__(64) k_stats(DevParams P, int hist_lds) {
  constexpr bool rd = RD;
  if (threadIdx.x >= ORL_STATS_LANES) return;
  const i64 env_raw = (i64)blockIdx.x * ORL_STATS_LANES + (i64)threadIdx.x;
  const int ln = env_raw < P.B ? P.log_n[env_raw >> 3] : 0;
  const int n = ln & 0xffff;
  const bool fin = ((ln >> 16) & 1) != 0;
  const bool lds_hist = ENV == ENV_RWA && hist_lds != 0;
  const int HW = (P.K + 1) + (P.S + 1);
  u32* s_h = (u32*)orl_lds_raw;  // [64][HW] (lds_hist)
  if (lds_hist) {
    for (int i = (int)threadIdx.x; i < ORL_STATS_LANES * HW; i += ORL_STATS_LANES) s_h[i] = 0u;
    wave_fence();
  }
  // (lanes without an env or without logged steps: the other families' leave here; RWA's stay for the cooperative flush below)
  if (!(ENV == ENV_RWA && lds_hist) && n == 0) return;
  // discrete bit rates (rmsa_env.py:217-227: requested / provisioned counts per rate): the same for the 2 n_br counters of a lane's
  // env — counted in the lane's own LDS row, added once per launch
  const bool br_lds = ENV != ENV_RWA && P.bit_rate_mode == 1 && hist_lds != 0;
  const int HB = 2 * P.n_br;
  if (br_lds)
    for (int i = 0; i < HB; i++) s_h[(int)threadIdx.x * HB + i] = 0u;
  const bool live = n > 0;
  const i64 env = live ? env_raw : 0;
  u64* s = P.scal + env * ORL_SCAL_WORDS;
#define F64(slot) __longlong_as_double((i64)s[slot])
  double g_thr = F64(SC_GTHR), g_comp = F64(SC_GCOMP), g_last = F64(SC_GLAST);
  double gc_a = F64(SC_GC_A), gc_td = F64(SC_GC_TD), now_a = F64(SC_NOWA);
#undef F64
  i64 sp = (i64)s[SC_SP], sa = (i64)s[SC_SA], esp = (i64)s[SC_ESP], esa = (i64)s[SC_ESA];
  i64 brq = (i64)s[SC_BRQ], brp = (i64)s[SC_BRP], ebrq = (i64)s[SC_EBRQ], ebrp = (i64)s[SC_EBRP];
  i64 s_br = (i64)s[SC_SBR], s_nh = (i64)s[SC_SNH];
  u64 acc = s[SC_ACC];
  const u64 acc_keep = acc & (1ull << 16);  // (serial releases pending: set by the loop, cleared by rel_serial)
  int id = (int)(u32)s[SC_ID_MTPOS];
  const u64* lg = P.slog + env;
  const size_t st = (size_t)P.log_stride;
  // the clock the first logged step is decided at and the bit rate of the service it serves: the record has moved on to the
  // launch's last step — the wavefront logged them when it began (header row, slot log_cap)
  double now = __longlong_as_double((i64)lg[(size_t)(3 * P.log_cap) * st]);
  int br_idx = (int)(u32)lg[(size_t)(3 * P.log_cap + 2) * st];
  int bit_rate = (ENV == ENV_RWA) ? 0 : ((P.bit_rate_mode == 0) ? P.br_lo + br_idx : P.bit_rates[br_idx]);
  bool done = false;
  for (int t0 = 0; t0 < n; t0 += ORL_STATS_BATCH) {
    u64 w0[ORL_STATS_BATCH], w1[ORL_STATS_BATCH], w2[ORL_STATS_BATCH];
    u32 w3[ORL_STATS_BATCH];
#pragma unroll
    for (int k = 0; k < ORL_STATS_BATCH; k++) {
      const int t = (t0 + k < n) ? t0 + k : n - 1;
      w0[k] = lg[(size_t)(3 * t) * st];
      w1[k] = lg[(size_t)(3 * t + 1) * st];
      w2[k] = lg[(size_t)(3 * t + 2) * st];
      if constexpr (rd && ENV != ENV_RWA) w3[k] = P.ssum[(size_t)t * st + (size_t)env];
      else w3[k] = 0u;
    }
#pragma unroll
    for (int k = 0; k < ORL_STATS_BATCH; k++) {
      if (t0 + k < n) {
        const u64 a1 = w1[k], a2 = w2[k];
        if (ENV != ENV_RWA && ((u32)acc & 2u)) {
          // the previous accepted step's network-compactness update, from the sums right after its provision (ctrl_a)
          const int c0 = (int)((acc >> 32) & 31);
          (void)c0;
          const i64 s_nh_prov = (i64)(acc >> 37);
          const int occ = rd ? (int)(w3[k] >> 16) : (int)((a1 >> 25) & 0x1ffffu), fb = rd ? (int)(w3[k] & 0xffffu) : (int)((a1 >> 42) & 0xffffu);
          const double cmp = (fb > 0) ? sp::div_pos((double)occ, (double)s_nh_prov) * sp::div_pos((double)P.E, (double)fb) : 1.0;
          g_comp = sp::div_pos(gc_a + (cmp * gc_td), now_a);
        }
        const bool accepted = (a1 & 1ull) != 0ull;
        const int core = (int)((a1 >> 58) & 31u), n_hops = (int)((a1 >> 1) & 0xfffu), br_new = (int)((a1 >> 13) & 0xfffu);
        const int d_nh = (int)(a2 & 0xfffffu), d_br = (int)((a2 >> 20) & 0xffffffu);
        if (accepted) {
          s_br += bit_rate;
          s_nh += n_hops;
          if (ENV != ENV_RWA) {
            brp += bit_rate;
            ebrp += bit_rate;
            if (P.bit_rate_mode == 1) {
              if (br_lds) s_h[(int)threadIdx.x * HB + P.n_br + br_idx] += 1u;
              else P.br_hist[env * 2 * P.n_br + P.n_br + br_idx] += 1;
            }
          }
          sa += 1;
          esa += 1;
        }
        if (ENV == ENV_RMCSA) {  // counted at the decision — and the bit rate requested a second time (rmcsa_env.py:294-295, 730-731)
          sp += 1; esp += 1;
          brq += bit_rate; ebrq += bit_rate;
        }
        if (ENV == ENV_RWA) {
          sp += 1; esp += 1;
          // actions_output marginals (rwa_env.py:103): the scan's action is never out of range
          const int path0 = (int)((a1 >> 25) & 15u), slot0 = (int)((a1 >> 29) & 1023u), rej = P.allow_rejection ? 1 : 0;
          // (atomics without a return value: a read-modify-write per marginal made the replay a chain of 2 n dependent global round
          // trips per lane — 80 us behind a 20-step launch of 65 536 RWA envs, three times the other families' replay)
          if (lds_hist) {
            u32* hl = s_h + (int)threadIdx.x * HW;
            if (path0 < P.K + rej) atomicAdd(hl + path0, 1u);
            if (slot0 < P.S + rej) atomicAdd(hl + (P.K + 1) + slot0, 1u);
          } else {
            unsigned long long* h = (unsigned long long*)(P.act_hist + env * ((P.K + 1) + (P.S + 1)));
            if (path0 < P.K + rej) atomicAdd(h + path0, 1ull);
            if (slot0 < P.S + rej) atomicAdd(h + (P.K + 1) + slot0, 1ull);
          }
        }
        acc = pack2(accepted ? 1 : 0, core);
        now_a = now;
        if (accepted && ENV != ENV_RWA) {  // _update_network_stats (rmsa_env.py:439-462), as service_part
          const double last_update = g_last, time_diff = now - last_update;
          if (now > 0) {
            const double cur_thr = (double)s_br;
            g_thr = sp::div_pos((g_thr * last_update) + (cur_thr * time_diff), now);
            gc_a = g_comp * last_update;
            gc_td = time_diff;
            acc = 3ull | ((u64)(u32)core << 32) | ((u64)s_nh << 37);
          }
          g_last = now;
        }
        // _next_service: the clock moves to the new arrival; the service is counted (RMSA / DeepRMSA)
        now = __longlong_as_double((i64)w0[k]);
        br_idx = (ENV != ENV_RWA) ? br_new : 0;
        bit_rate = (ENV == ENV_RWA) ? 0 : ((P.bit_rate_mode == 0) ? P.br_lo + br_idx : P.bit_rates[br_idx]);
        id = (int)esp;
        if (ENV == ENV_RMSA || ENV == ENV_DEEPRMSA) { sp += 1; esp += 1; }
        if (ENV != ENV_RWA) {
          brq += bit_rate;
          ebrq += bit_rate;
          if (P.bit_rate_mode == 1) {
            if (br_lds) s_h[(int)threadIdx.x * HB + br_idx] += 1u;
            else P.br_hist[env * 2 * P.n_br + br_idx] += 1;
          }
        }
        done = (esp == (i64)P.episode_length);
        if (done) {
          if (P.ep_log) episode_log(P, env, esa);
          ebrq = 0; ebrp = 0; esp = 0; esa = 0;
          if (ENV != ENV_RWA) { esp += 1; ebrq += bit_rate; }
        }
        // the step's releases
        s_br -= d_br;
        s_nh -= d_nh;
      }
    }
  }
  if (fin && ENV != ENV_RWA && ((u32)acc & 2u)) {
    // the run ends here: the update the last step left pending, from the sums after its row phase (k_finish2's expressions)
    int occ, fb;
    if constexpr (rd) {
      const u32 a3 = P.ssum[(size_t)n * st + (size_t)env];
      occ = (int)(a3 >> 16); fb = (int)(a3 & 0xffffu);
    } else {
      const u64 a1 = lg[(size_t)(3 * n + 1) * st];
      occ = (int)((a1 >> 25) & 0x1ffffu); fb = (int)((a1 >> 42) & 0xffffu);
    }
    const i64 s_nh_prov = (i64)(acc >> 37);
    const double cmp = (fb > 0) ? ((double)occ / (double)s_nh_prov) * ((double)P.E / (double)fb) : 1.0;
    g_comp = (gc_a + (cmp * gc_td)) / now_a;
    acc &= ~2ull;
  }
  if (live) {
#define PF(slot, x) s[slot] = (u64)__double_as_longlong(x);
    PF(SC_GTHR, g_thr) PF(SC_GCOMP, g_comp) PF(SC_GLAST, g_last) PF(SC_GC_A, gc_a) PF(SC_GC_TD, gc_td) PF(SC_NOWA, now_a)
#undef PF
    s[SC_SP] = (u64)sp; s[SC_SA] = (u64)sa; s[SC_ESP] = (u64)esp; s[SC_ESA] = (u64)esa;
    s[SC_BRQ] = (u64)brq; s[SC_BRP] = (u64)brp; s[SC_EBRQ] = (u64)ebrq; s[SC_EBRP] = (u64)ebrp;
    s[SC_SBR] = (u64)s_br; s[SC_SNH] = (u64)s_nh;
    s[SC_ACC] = acc | acc_keep;
    ((u32*)(s + SC_ID_MTPOS))[0] = (u32)id;
  }
  (void)now; (void)br_idx;
  if (br_lds && live) {
    unsigned long long* hb = (unsigned long long*)(P.br_hist + env * HB);
    for (int i = 0; i < HB; i++) {
      const u32 c = s_h[(int)threadIdx.x * HB + i];
      if (c) atomicAdd(hb + i, (unsigned long long)c);
    }
  }
  if (lds_hist) {
    // the launch's counts to the envs' histograms: the 64 lanes walk one env's row together (consecutive 8-byte counters)
    wave_fence();
    for (int e = 0; e < ORL_STATS_LANES; e++) {
      const i64 env_e = (i64)blockIdx.x * ORL_STATS_LANES + e;
      if (env_e >= P.B) break;
      // (atomics without a return value on consecutive counters: nothing waits for them — as read-modify-writes the 128 rounds
      // of this loop were a chain of dependent global round trips, 60 us per launch)
      unsigned long long* h = (unsigned long long*)(P.act_hist + env_e * HW);
      for (int i = (int)threadIdx.x; i < HW; i += ORL_STATS_LANES) {
        const u32 c = s_h[e * HW + i];
        if (c) atomicAdd(h + i, (unsigned long long)c);
      }
    }
  }
}

// ---- rows-deferred form: the replay of the row statistics (round 6) ----------------------------------------------------------
// In the rows-deferred forms of k_persist the loop is slot scan + control phase only: the control phase changes the slot maps
// itself and logs one event per provision / release (sp::ctrl_d<..., RD>).  Nothing the row phase computed feeds a decision —
// the per-link running averages of _update_link_stats (rmsa_env.py:464-543; rwa_env.py:365-383) and the integer sums behind
// _get_network_compactness (rmsa_env.py:699-744) are state the loop only ever wrote.  This kernel replays them after the launch
// with one LANE per link ROW: the lane starts from the row as the launch found it (DevParams::bitmap0, written when the wavefront
// filled its window) and walks the events that touch its link in order — mask, row summary, float64 update at the event's clock:
// the expressions of sp::row_item_lane1 — with the row, the link's 32-byte record, the row's inner-run cache and its contribution
// to the sums in registers for the whole launch: no provision / release lane pairs, no sink tables, one read and one write of
// the link record per launch instead of per touch.
// A workgroup owns G whole envs and works through their events in WINDOWS of 128 events per env:
//   A  all threads: the window's events into LDS (step, slots, clock: a provision happens at the clock its step was decided at —
//      the previous step's new clock, or the launch's header row — a release at the step's new clock, log word w0) and, per event
//      and link of its path, a bit into the touch words of that link's row (LDS atomics: ~2.4 per event instead of a test per
//      event and row)
//   B  (first window) the rows sorted by their number of touches, heaviest first: lanes of one wavefront then run about the same
//      number of rounds — with rows in link order a wavefront ran as many rounds as its busiest row, at 4.4 touches per row on
//      average and ~11 at most in a 20-step launch (lanes busy 45 %)
//   C  every lane walks the touches of ITS row: everything it needs comes from LDS
//   D  one lane per env walks the window's events in order and turns the touches' deltas (LDS atomics, a cell per event) into the
//      sums right after each step's provision — what the in-loop forms log into word w1 — stored to DevParams::ssum for k_stats;
//      when the launch's events are done it leaves DevParams::core_sums as the in-loop row phase would have.
#define ORL_ROWSTATS_THREADS 256
#ifdef ORL_RS_PROF  // diagnostic builds: shader-clock cycles per phase of k_rowstats, summed over wavefronts (tools/rs_prof.py)
#define ORL_RSP_WAVES 32768
__device__ unsigned long long g_rs_prof[ORL_RSP_WAVES * 16];  // (a slot per wavefront: atomics on shared counters perturb what is measured)
#define ORL_RSP_SLOT(k) g_rs_prof[((size_t)((blockIdx.x * 4 + (threadIdx.x >> 6)) % ORL_RSP_WAVES)) * 16 + (k)]
#define ORL_RSP(k) do { const long long n_ = clock64(); if ((threadIdx.x & 63) == 0) ORL_RSP_SLOT(k) += (unsigned long long)(n_ - rsp_t); rsp_t = n_; } while (0)
#define ORL_RSP_CNT(k, v) do { if ((threadIdx.x & 63) == 0) ORL_RSP_SLOT(k) += (unsigned long long)(v); } while (0)
#else
#define ORL_RSP(k) do { } while (0)
#define ORL_RSP_CNT(k, v) do { } while (0)
#endif
#define ORL_RS_WIN 64   // events per env and window (a 20-step launch logs ~40 per env: one window)
#define ORL_RS_GMAX 16  // envs per workgroup at most
#ifndef ORL_RS_WAVES
#define ORL_RS_WAVES 5  // waves per SIMD the register allocator leaves room for
#endif
// one touch of a row in the replay: the event's mask applied, the summary brought up to date (sp::row_inc_apply), the link's running
// averages updated at the event's clock (_update_link_stats, rmsa_env.py:464-543; the expressions of sp::row_item_lane1); returns what
// the row's contribution to the compactness sums changed by, (occupied range << 16) + free blocks inside
struct RsLink { double util, frag, comp, last_update; u32 cw; };  // (cw: rows of 3-5 words — the per-word cache of inner free runs, 63 = unknown)
// Rows of 3-5 words (cfg2) are summarised word by word with that cache, as the in-loop row phase does (sp::row_stat_lane): the
// incremental summary issues MORE instructions there — 1.37 against 1.15e8 VALU wavefront-instructions per 20-step replay — because
// some lane of a wavefront re-searches the longest run at almost every touch; rows of one or two words take the incremental one
// (cfg3's replay 286 -> 188 us).
template <int W> struct RsIncremental { static constexpr bool value = !(W >= 3 && W <= 5); };
template <bool RWA, int W>
__device__ __forceinline__ void rs_row_init(const u64 (&a)[W], int S, sp::RowInc& ri, int& occ0, int& fb0) {
  if (!RWA && !RsIncremental<W>::value) {
    sp::row_occ_fb<W>(a, S, occ0, fb0);  // (the word-by-word summary needs the row's contribution to the sums only)
    ri.free_ = 0; ri.nu = 0; ri.lo = 1 << 20; ri.hi = 0; ri.me = 0;
  } else if (!RWA) {  // the row's summary as the launch found it: once per row and launch
    RowStat st0;
    int me0 = 0, edge0 = 0;
    sp::row_stat_lane<W>(a, S, st0, me0, edge0);
    ri.free_ = st0.free_; ri.nu = st0.nu; ri.lo = st0.lo; ri.hi = st0.hi; ri.me = me0;
    occ0 = st0.occ; fb0 = st0.fb;
  } else {
    int f = 0;
#pragma unroll
    for (int w = 0; w < W; w++) f += __popcll(a[w]);
    ri.free_ = f; ri.nu = 0; ri.lo = 1 << 20; ri.hi = 0; ri.me = 0;
    occ0 = 0; fb0 = 0;
  }
}
template <bool RWA, int W>
__device__ __forceinline__ int rs_touch(u64 (&a)[W], sp::RowInc& ri, RsLink& rl, int& occ0, int& fb0, u32 meta, double clock, int S) {
  const int s0 = (int)(meta & 0x1ffu), n = (int)((meta >> 9) & 63u);
  const bool prov = ((meta >> 24) & 1u) != 0u;
  int nu = 0, lo = 0, hi = 0, nf = 0, occ = 0, fb = 0, max_empty = 0, edge = 0;
  if (!RWA && !RsIncremental<W>::value) {
    const sp::Mask2 mm = sp::mask2(s0, n);
#pragma unroll
    for (int w = 0; w < W; w++) a[w] ^= sp::mask2_word(mm, w);  // (a provision clears free slots, a release sets taken ones)
    RowStat after;
    sp::row_stat_lane<W, !RsIncremental<W>::value>(a, S, after, max_empty, edge, &rl.cw, sp::mask_words(s0, n), true);
    ri.free_ = after.free_;
    nu = after.nu; lo = after.lo; hi = after.hi; nf = after.nf; occ = after.occ; fb = after.fb;
  } else if (!RWA) {
    // (incremental: the two free runs next to the mask give everything the summary changes by)
    sp::row_inc_apply<W>(a, S, s0, n, prov, ri);
    const int tw = (S - 1) >> 6, tb = (S - 1) & 63;
    edge = (int)(a[0] & 1ull) + (int)((sp::row_word<W>(a, tw) >> tb) & 1ull);
    const bool two = ri.nu > 1;
    nu = ri.nu; lo = ri.lo; hi = ri.hi; nf = ri.nu - 1 + edge;
    occ = two ? ri.hi - ri.lo : 0; fb = two ? ri.nu - 1 : 0;
    max_empty = ri.me;
  } else {
    ri.free_ += prov ? -n : n;  // (RWA: the utilization is all a link keeps)
  }
  const int free_ = ri.free_;
  const double cur_util = sp::div_pos((double)(S - free_), (double)S);
  double cur_frag = 0.0, cur_comp = 0.0;
  if (!RWA && free_ > 0) {
    const int me = (nf > 1 && !(nf == 2 && edge == 2)) ? max_empty : 0;
    cur_frag = 1.0 - sp::div_pos((double)me, (double)free_);
    if (nu > 1) cur_comp = sp::div_pos((double)(hi - lo), (double)(S - free_)) * sp::div_pos(1.0, (double)nu);
    else cur_comp = 1.0;
  }
  if (clock > 0) {
    // (a link touched again at the same clock: time_diff == 0, i.e. new = ((old * now) + (cur * 0.0)) / now with a finite
    // cur >= 0 — the reference's own expression, and what the in-loop row phase computes for the further releases of a step)
    const double time_diff = clock - rl.last_update;
    const sp::Recip rc = sp::recip_of(clock);
    rl.util = sp::div_by((rl.util * rl.last_update) + (cur_util * time_diff), rc);
    if (!RWA) {
      rl.frag = sp::div_by((rl.frag * rl.last_update) + (cur_frag * time_diff), rc);
      rl.comp = sp::div_by((rl.comp * rl.last_update) + (cur_comp * time_diff), rc);
    }
  }
  rl.last_update = clock;
  int d = 0;
  if (!RWA) {
    d = ((occ - occ0) << 16) + (fb - fb0);
    occ0 = occ; fb0 = fb;
  }
  return d;
}
struct RowstatsLds { int bits, meta, clk, delta, hist, perm, nev, sc, total; };
__host__ __device__ inline RowstatsLds rowstats_lds_layout(int G) {
  RowstatsLds L;
  int o = 0;
  L.clk = o; o += G * ORL_RS_WIN * 8;
  L.bits = o; o += (ORL_RS_WIN / 32) * ORL_ROWSTATS_THREADS * 4;
  L.meta = o; o += G * ORL_RS_WIN * 4;
  L.delta = o; o += G * ORL_RS_WIN * 4;
  L.hist = o; o += 2 * 16 * 4;
  L.perm = o; o += ORL_ROWSTATS_THREADS * 2;
  L.nev = o; o += ORL_RS_GMAX * 4;
  L.sc = o; o += (ORL_RS_GMAX * 8 + 4) * 4;
  L.total = (o + 15) & ~15;
  return L;
}
template <int ENV, int W>
__global__ void __launch_bounds__(ORL_ROWSTATS_THREADS) __attribute__((amdgpu_waves_per_eu(ORL_RS_WAVES, ORL_RS_WAVES)))
k_rowstats(DevParams P, int G) {
  constexpr bool RWA = ENV == ENV_RWA;
  constexpr int NWORD = ORL_RS_WIN / 32;
  const int E = P.E, S = P.S;
  const int tid = (int)threadIdx.x;
  const RowstatsLds L = rowstats_lds_layout(G);
  u32* s_bits = (u32*)(orl_lds_raw + L.bits);      // [NWORD][256]: touch bits of row r, word w at [w][r]
  double* s_clk = (double*)(orl_lds_raw + L.clk);  // [G][WIN]
  u32* s_meta = (u32*)(orl_lds_raw + L.meta);      // [G][WIN]
  int* s_delta = (int*)(orl_lds_raw + L.delta);    // [G][WIN]
  int* s_hist = (int*)(orl_lds_raw + L.hist);      // [16] rows per touch count (15 = that many or more), [16] cursors
  unsigned short* s_perm = (unsigned short*)(orl_lds_raw + L.perm);
  int* s_nev = (int*)(orl_lds_raw + L.nev);
  const i64 envb = (i64)blockIdx.x * G;
#ifdef ORL_RS_PROF
  long long rsp_t = clock64();
#endif
  // the env scan of phase D keeps its state in LDS (whichever wavefront finishes the workgroup's last window runs it): per env the
  // running totals, the sums pending for the step after the last event's, that step, the steps logged, finished-the-run, on / off
  int* s_sc = (int*)(orl_lds_raw + L.sc);  // [GMAX][8]
  int* s_done = s_sc + ORL_RS_GMAX * 8;
  if (tid < ORL_RS_GMAX) {
    int n = 0, sc_occ = 0, sc_fb = 0, sp_occ = 0, sp_fb = 0, sc_ns = 0, sc_fin = 0, sc_on = 0;
    if (tid < G && envb + tid < P.B) {
      const int ln = P.log_n[(envb + tid) >> 3];
      sc_ns = ln & 0xffff;
      sc_fin = (ln >> 16) & 1;
      n = sc_ns ? P.elog_n[envb + tid] : 0;
      sc_on = (!RWA && sc_ns > 0) ? 1 : 0;
      if (sc_on) {
        const int* cs = P.core_sums + (envb + tid) * P.cs_words;
        sc_occ = cs[0]; sc_fb = cs[1];
        sp_occ = sc_occ - cs[2 * P.C]; sp_fb = sc_fb - cs[2 * P.C + 1];  // (what the previous launch's last releases added is not part of it)
      }
    }
    s_nev[tid] = n;
    int* q = s_sc + 8 * tid;
    q[0] = sc_occ; q[1] = sc_fb; q[2] = sp_occ; q[3] = sp_fb; q[4] = -1; q[5] = sc_ns; q[6] = sc_fin; q[7] = sc_on;
    if (tid == 0) *s_done = 0;
  }
  if (tid >= 32 && tid < 64) s_hist[tid - 32] = 0;
  __syncthreads();
  int nev_max = s_nev[tid & (ORL_RS_GMAX - 1)];
#pragma unroll
  for (int o = 1; o < ORL_RS_GMAX; o <<= 1) { const int t_ = __shfl_xor(nev_max, o, 64); nev_max = t_ > nev_max ? t_ : nev_max; }
  // this lane's row (assigned after the first window's sort) and its state
  int r = tid, g_r = 0, link = 0;
  bool have = false, touched_any = false;
  u64 a[W];
#pragma unroll
  for (int w = 0; w < W; w++) a[w] = 0ull;
  RsLink rl;
  rl.util = 0.0; rl.frag = 0.0; rl.comp = 0.0; rl.last_update = 0.0; rl.cw = 0u;
#pragma unroll
  for (int w = 0; w < (W <= 5 ? W : 0); w++) rl.cw |= 63u << (6 * w);  // (nothing known about the row's inner runs yet)
  int occ0 = 0, fb0 = 0;
  sp::RowInc ri;
  ri.free_ = 0; ri.nu = 0; ri.lo = 1 << 20; ri.hi = 0; ri.me = 0;
  double* ls = nullptr;
  // the steps behind the last event, the slot behind the last step (k_stats finishes the run's last pending update from it), and
  // core_sums as the in-loop row phase leaves them: the totals, and what the last step's releases added (cleared when the
  // wavefront finished the run's state: DevParams::persist_finish)
  auto finish_envs = [&](int l) {
    if (l >= G) return;
    const int* q = s_sc + 8 * l;
    if (!q[7]) return;
    const int sc_occ = q[0], sc_fb = q[1], sp_occ = q[2], sp_fb = q[3], sc_t = q[4], sc_ns = q[5];
    u32* out = P.ssum + (size_t)(envb + l);
    const size_t st = (size_t)P.log_stride;
    out[(size_t)(sc_t + 1) * st] = ((u32)sp_occ << 16) | (u32)sp_fb;
    for (int s2 = sc_t + 2; s2 <= sc_ns; s2++) out[(size_t)s2 * st] = ((u32)sc_occ << 16) | (u32)sc_fb;
    // (the sums of slot ns: after the last step's provision — the pending ones when that step had events, else the totals)
    const bool last_had = sc_t == sc_ns - 1;
    const int e_occ = last_had ? sp_occ : sc_occ, e_fb = last_had ? sp_fb : sc_fb;
    int* cs = P.core_sums + (envb + l) * P.cs_words;
    cs[0] = sc_occ; cs[1] = sc_fb;
    cs[2 * P.C] = q[6] ? 0 : sc_occ - e_occ;
    cs[2 * P.C + 1] = q[6] ? 0 : sc_fb - e_fb;
  };
  ORL_RSP(0);
  for (int j0 = 0; j0 < nev_max; j0 += ORL_RS_WIN) {
    // ---- A: the window's events ------------------------------------------------------------------------------------------------
#pragma unroll
    for (int w = 0; w < NWORD; w++) s_bits[w * ORL_ROWSTATS_THREADS + tid] = 0u;
    // (the cells of a window: G x the smallest power of two that holds its events)
    const int nwin = nev_max - j0 < ORL_RS_WIN ? nev_max - j0 : ORL_RS_WIN;
    const int sh = nwin <= 8 ? 3 : (nwin <= 16 ? 4 : (nwin <= 32 ? 5 : 6));
    ulonglong2 ea[(ORL_RS_GMAX * ORL_RS_WIN + ORL_ROWSTATS_THREADS - 1) / ORL_ROWSTATS_THREADS], eb[(ORL_RS_GMAX * ORL_RS_WIN + ORL_ROWSTATS_THREADS - 1) / ORL_ROWSTATS_THREADS];
#pragma unroll
    for (int q = 0; q < (ORL_RS_GMAX * ORL_RS_WIN + ORL_ROWSTATS_THREADS - 1) / ORL_ROWSTATS_THREADS; q++) {  // (every request of the window first)
      const int i = tid + q * ORL_ROWSTATS_THREADS, g = i >> sh, k = i & ((1 << sh) - 1);
      ea[q] = make_ulonglong2(0ull, 0ull); eb[q] = ea[q];
      if (g < G && j0 + k < s_nev[g]) {
        const ulonglong2* ev = P.elog + ((envb + g) * (i64)P.elog_cap + (j0 + k)) * 2;
        ea[q] = ev[0]; eb[q] = ev[1];
      }
    }
    __syncthreads();  // (the touch words are clear)
#pragma unroll
    for (int q = 0; q < (ORL_RS_GMAX * ORL_RS_WIN + ORL_ROWSTATS_THREADS - 1) / ORL_ROWSTATS_THREADS; q++) {
      const int i = tid + q * ORL_ROWSTATS_THREADS, g = i >> sh, k = i & ((1 << sh) - 1);
      if (g < G) {
        const int c = g * ORL_RS_WIN + k;
        s_delta[c] = 0;
        if (j0 + k < s_nev[g]) {
          s_clk[c] = __longlong_as_double((i64)eb[q].x);
          s_meta[c] = (u32)ea[q].x;
          const u32 bit = 1u << (k & 31);
          u32* bw = s_bits + (k >> 5) * ORL_ROWSTATS_THREADS + g * E;
          for (u64 lm = ea[q].y; lm; lm &= lm - 1ull) atomicOr(bw + (int)__builtin_ctzll(lm), bit);
        }
      }
    }
    ORL_RSP(1);
    __syncthreads();
    ORL_RSP(2);
    // ---- B: rows sorted by touches (first window), then each lane fetches its row ---------------------------------------------------
    if (j0 == 0) {
      int cnt = 0;
#pragma unroll
      for (int w = 0; w < NWORD; w++) cnt += __popc(s_bits[w * ORL_ROWSTATS_THREADS + tid]);
      const int bin = 15 - (cnt < 15 ? cnt : 15);  // (heaviest first)
      atomicAdd(&s_hist[bin], 1);
      __syncthreads();
      int base = 0;
#pragma unroll
      for (int b = 0; b < 15; b++) base += (b < bin) ? s_hist[b] : 0;
      s_perm[base + atomicAdd(&s_hist[16 + bin], 1)] = (unsigned short)tid;
      __syncthreads();
      // (which quarter of the sorted rows a wavefront takes rotates with the workgroup: wavefront w of every workgroup runs on the
      // same SIMD of its CU, and the heaviest quarter everywhere on SIMD 0 left the other three SIMDs half idle)
      r = (int)s_perm[(((tid >> 6) + (int)blockIdx.x) & 3) * 64 + (tid & 63)];
      g_r = r / E;
      link = r - g_r * E;
      have = g_r < G && s_nev[g_r] > 0;
      if (have) {
        const i64 env = envb + g_r;
        const u64* row = P.bitmap0 + env * P.bm_words + (size_t)link * W;
#pragma unroll
        for (int w = 0; w < W; w++) a[w] = row[w];
        ls = P.lstat + env * 4 * E + 4 * link;
        const double2 ls01 = *(const double2*)ls, ls23 = *(const double2*)(ls + 2);
        rl.util = ls01.x; rl.frag = ls01.y; rl.comp = ls23.x; rl.last_update = ls23.y;
        rs_row_init<RWA, W>(a, S, ri, occ0, fb0);
      }
    }
    ORL_RSP(3);
    int rounds_ = 0;
    // ---- C: this lane's touches of the window ---------------------------------------------------------------------------------------
    if (have) {
      const double* clk_g = s_clk + g_r * ORL_RS_WIN;
      const u32* meta_g = s_meta + g_r * ORL_RS_WIN;
      int wi = 0;
      u32 word = s_bits[r];
      for (;;) {
        while (word == 0u && wi + 1 < NWORD) { wi++; word = s_bits[wi * ORL_ROWSTATS_THREADS + r]; }
        if (word == 0u) break;
        const int k = 32 * wi + (int)__builtin_ctz(word);
        word &= word - 1u;
        touched_any = true;
        rounds_++;
        const int d = rs_touch<RWA, W>(a, ri, rl, occ0, fb0, meta_g[k], clk_g[k], S);
        if (!RWA && d) atomicAdd(&s_delta[g_r * ORL_RS_WIN + k], d);
      }
    }
    ORL_RSP(4);
#ifdef ORL_RS_PROF
    { int mx = rounds_, sm = rounds_;
      for (int o = 32; o > 0; o >>= 1) { const int t_ = __shfl_xor(mx, o, 64); mx = t_ > mx ? t_ : mx; sm += __shfl_xor(sm, o, 64); }
      ORL_RSP_CNT(8, mx); ORL_RSP_CNT(9, sm); ORL_RSP_CNT(10, 1); }
#endif
    (void)rounds_;
    // ---- D: the sums after each step's provision, one lane per env ------------------------------------------------------------------
    // (lane l < G of the wavefront that runs it; state in LDS)
    auto scan_window = [&](int l) {
      int* q = s_sc + 8 * l;
      if (!q[7]) return;
      int sc_occ = q[0], sc_fb = q[1], sp_occ = q[2], sp_fb = q[3], sc_t = q[4];
      const int nk = s_nev[l] - j0 < ORL_RS_WIN ? s_nev[l] - j0 : ORL_RS_WIN;
      u32* out = P.ssum + (size_t)(envb + l);
      const size_t st = (size_t)P.log_stride;
      for (int k = 0; k < nk; k++) {
        const u32 meta = s_meta[l * ORL_RS_WIN + k];
        const int t = (int)((meta >> 15) & 0x1ffu);
        if (t != sc_t) {
          // the step before is complete: its pending sums are those of step sc_t + 1; steps without events in between see the totals
          out[(size_t)(sc_t + 1) * st] = ((u32)sp_occ << 16) | (u32)sp_fb;
          for (int s2 = sc_t + 2; s2 <= t; s2++) out[(size_t)s2 * st] = ((u32)sc_occ << 16) | (u32)sc_fb;
          sc_t = t;
          sp_occ = sc_occ; sp_fb = sc_fb;
        }
        const int d = s_delta[l * ORL_RS_WIN + k];
        const int d_fb = (int)(short)(d & 0xffff);
        sc_occ += (d - d_fb) >> 16; sc_fb += d_fb;
        if ((meta >> 24) & 1u) { sp_occ = sc_occ; sp_fb = sc_fb; }  // (right after the provision)
      }
      q[0] = sc_occ; q[1] = sc_fb; q[2] = sp_occ; q[3] = sp_fb; q[4] = sc_t;
    };
    if (j0 + ORL_RS_WIN < nev_max) {  // (more windows follow: the tables are overwritten, everybody waits)
      __syncthreads();
      ORL_RSP(5);
      if (tid < G) scan_window(tid);
      ORL_RSP(6);
      __syncthreads();
      ORL_RSP(7);
    } else {
      // The workgroup's last window: a wavefront whose rows are done stores their link records and LEAVES — no barrier: the four
      // wavefronts take sorted quarters of the rows, and the lighter three waited a quarter of their lifetime for the heaviest
      // (tools/rs_prof.py) while holding registers another workgroup could use.  The last one to get here runs the env scan.
      if (have && touched_any) {
        *(double2*)ls = make_double2(rl.util, rl.frag);
        *(double2*)(ls + 2) = make_double2(rl.comp, rl.last_update);
      }
      int arrived = 0;
      if ((tid & 63) == 0) arrived = __hip_atomic_fetch_add(s_done, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
      arrived = __builtin_amdgcn_readfirstlane(arrived);
      ORL_RSP(5);
      if (arrived != ORL_ROWSTATS_THREADS / 64 - 1) return;
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
      if ((tid & 63) < G) scan_window(tid & 63);
      ORL_RSP(6);
      finish_envs(tid & 63);
      return;
    }
  }
  // (no events at all in this workgroup's envs: the steps they logged still get their sums)
  if (tid < ORL_RS_GMAX) finish_envs(tid);
}

// LDS: 0 = the state stays in global memory, 1 = slot maps + per-core sums + env records in LDS, 2 = + link statistics,
// 3 = slot maps + per-core sums in LDS, env records in global memory
// PF: early requests of the Mersenne-Twister window and the link statistics (more live registers: the 3-wave forms)
// SVC: services drawn 8 steps ahead, one per lane of an env's group (sp::svc_generate)
// ---- the two-wavefront form (RW; round 5, small batches) -----------------------------------------------------------------------
// A batch of a few thousand envs leaves most SIMDs idle, and a lone wavefront's step is a chain of dependent latencies: the row
// phase is a quarter of the chain, and nothing it computes — summaries, running averages, sums — feeds the next decision;
// only the slot maps do.  In this form a workgroup is a PAIR of wavefronts sharing the LDS window.  The control wavefront runs
// everything but the row phase and changes the slot maps itself, as it appends a mask to the sink (sp::row_apply_mask: LDS
// atomics without a return value); the row wavefront waits for a step's item list, reads the rows — final already — takes the
// step's masks back to get the states the statistics are defined on (row_item_lane1<EARLY>) and does the statistics while the
// control wavefront is in its next step.  The control wavefront waits twice per step, normally for nothing: before it clears the
// tables, for the row wavefront to have read them; before it logs the sums, for the previous step's statistics.  Single-core
// families, slot maps in the LDS window (everything the two wavefronts share is in LDS).  Four words behind the window: [0]
// steps whose items are listed, [1] steps whose tables and rows have been read, [2] steps whose statistics are complete, [3] the
// control wavefront has left its loop.
// Behind the four counters, two channels of four words each — [4..7] the batch asked for a step ahead, [8..11] one asked for on
// the spot: batches asked for, batches drawn, services wanted in the batch asked for, which groups draw (bit per env) — then
// the two staging areas, one slot per lane of the control wavefront (ORL_RW_STAGE_BYTES).
// The service look-ahead (sp::svc_generate: a window of Mersenne-Twister words, two logarithms, three table searches for 8 services
// per env at once, every 8th step) is 9 % of the control wavefront's chain and open-loop like the row statistics: the control
// wavefront asks for the next batch when its own holds ONE more service, behind a step that was not cut short — so at any exit
// it holds either services of its own and nothing is on order, or none and the batch on order replaces them — and finds the batch
// in LDS a step later.  A group that is out of phase with the others (its window of generator words once held fewer services than
// asked for) and runs dry while their batch is on order gets one on the spot, into the second staging area.  Only the row
// wavefront touches the generator's state.
#define ORL_RW_SYNC_WORDS 12
#define ORL_RW_STAGE_BYTES (2 * 64 * 24)  // two batches: the one asked for a step ahead, and one asked for on the spot
#define ORL_RW_EXTRA_BYTES (ORL_RW_SYNC_WORDS * 4 + ORL_RW_STAGE_BYTES)
__device__ __forceinline__ u32 rw_load(const u32* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// (Everything the pair shares is in LDS.  Round 6: the hand-over is a workgroup-scope release / acquire on the LOCAL address space
// only — sp::rw_release_lds / rw_acquire_lds: the release drains the wavefront's LDS operations (s_waitcnt lgkmcnt(0)) before the
// counter is stored and orders the compiler, inside the memory model.  A plain workgroup-scope release would also wait for every
// global store the wavefront has in flight — log words and event records that nobody in the pair reads, a global round trip in the
// step's chain — which is what round 5 avoided with wavefront-scope fences and the assumption that a CU's LDS executes a wavefront's
// instructions in issue order.)
__device__ __forceinline__ void rw_wait(const u32* p, u32 want) {
  while ((u32)__builtin_amdgcn_readfirstlane((int)rw_load(p)) < want) __builtin_amdgcn_s_sleep(1);
  sp::rw_acquire_lds();
  ORL_DIAG_JITTER();
}
__device__ __forceinline__ void rw_signal(u32* p, u32 v, int lane) {
  ORL_DIAG_JITTER();
  sp::rw_release_lds();
  if (lane == 0) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <int ENV, int W, int LDS>
__device__ __forceinline__ void persist_row_wave(const DevParams& P, const PersistLds& L, int ICL, u32* sync) {
  static_assert(LDS == 1 || LDS == 3, "the two-wavefront form shares the slot maps through the LDS window");
  const int lane = lane_id();
  const i64 env0 = (i64)blockIdx.x * 8;
  const sp::SinkEntryC* s_tab = (const sp::SinkEntryC*)(orl_lds_raw + L.tab);
  const unsigned short* s_mtab = (const unsigned short*)(orl_lds_raw + L.mtab);
  unsigned short* s_list = (unsigned short*)(orl_lds_raw + L.list);
  u32* s_list_n = (u32*)(orl_lds_raw + L.misc + 8);
  sp::Wmem M = sp::wmem_global(P);
  M.clk = (double*)(orl_lds_raw + L.clk);
  M.clk_env0 = env0;
  M.bm0 = (u64*)(orl_lds_raw + L.bm);
  M.env0 = env0;
  M.cs0 = (int*)(orl_lds_raw + L.cs);
  M.cenv0 = env0;
  M.cs_lds = true;
  M.cs_stride = L.csw;
  if (ICL >= 1) M.ic0 = (u32*)(orl_lds_raw + L.ic);
  if (ICL >= 2) M.oc0 = (u32*)(orl_lds_raw + L.ic + ((8 * P.E * 4 + 15) & ~15));
  else if (ICL == 1 && P.row_cache_key != 0) M.ocg = P.row_cache + (size_t)blockIdx.x * 2 * P.row_cache_words + P.row_cache_words;
  if (LDS == 1) {  // (the env records are in the window: the generator's position is a word of them)
    M.sc0 = (u64*)(orl_lds_raw + L.sc);
    M.scenv0 = env0;
    M.sc_stride = ORL_SCAL_LDS_WORDS;
  }
  sp::Prof prof;
  ORL_PROF_BEGIN();
  u32 k = 0u, drawn0 = 0u, drawn1 = 0u;
  for (;;) {
    const u32 gone = (u32)__builtin_amdgcn_readfirstlane((int)rw_load(sync + 3));
    const u32 asked0 = (u32)__builtin_amdgcn_readfirstlane((int)rw_load(sync + 4));
    const u32 asked1 = (u32)__builtin_amdgcn_readfirstlane((int)rw_load(sync + 8));
    const u32 listed = (u32)__builtin_amdgcn_readfirstlane((int)rw_load(sync + 0));
    // a batch of services for the control wavefront: one it is waiting for at once, one for its next step when there are no
    // statistics to do
    if (asked1 > drawn1 || (listed <= k && asked0 > drawn0)) {
      {
        const int ch = (asked1 > drawn1) ? 1 : 0;
        u32* chan = sync + 4 + 4 * ch;
        sp::rw_acquire_lds();
        const int n_want = __builtin_amdgcn_readfirstlane((int)rw_load(chan + 2));
        const u32 groups = (u32)__builtin_amdgcn_readfirstlane((int)rw_load(chan + 3));
        double* stg_q = (double*)((char*)(sync + ORL_RW_SYNC_WORDS) + ch * (ORL_RW_STAGE_BYTES / 2));
        double* stg_ht = stg_q + 64;
        u32* stg_pk = (u32*)(stg_ht + 64);
        int* stg_cnt = (int*)(stg_pk + 64);
        const i64 env = env0 + (lane >> 3);
        const bool active = env < P.B && ((groups >> (lane >> 3)) & 1u) != 0u;
        sp::SvcBuf sb;
        sb.q = 0.0; sb.ht = 0.0; sb.pk = 0u; sb.cnt = 0;
        sp::svc_generate<ENV>(P, sp::wm_scal(P, M, active ? env : M.scenv0), P.mt + (active ? env : 0) * 624, lane, n_want, sb, active);
        stg_q[lane] = sb.q; stg_ht[lane] = sb.ht; stg_pk[lane] = sb.pk; stg_cnt[lane] = sb.cnt;
        if (ch) drawn1++; else drawn0++;
        rw_signal(chan + 1, ch ? drawn1 : drawn0, lane);
        ORL_PROFR(11);
        continue;
      }
    }
    if (listed <= k) {
      if (gone) break;
      __builtin_amdgcn_s_sleep(2);
      continue;
    }
    sp::rw_acquire_lds();
    ORL_PROFR(8);  // (idle)
    k++;
    // the step's work items, from the sink table the control wavefront filled (its own ctrl_d skips this in the pair form)
    sp::sink_compact(s_tab, P.E, lane, s_list, s_list_n);
    wave_fence();
    const int n_items = __builtin_amdgcn_readfirstlane((int)*s_list_n);
    for (int base = 0; base < n_items; base += 64) {
      const int idx = base + lane;
      if (idx < n_items) {
        const int code = (int)s_list[idx];
        const int el = (code >> 8) & 7, link = code & 0xff, second = code >> 15;
        sp::row_item_lane1<ENV, W, true>(P, M, env0 + el, link, s_tab[P.E * el + link].bits, s_mtab + ORL_MTAB * el, second, prof, true, nullptr,
                                         (base + 64 >= n_items) ? sync + 1 : nullptr, k);
      }
    }
    if (n_items == 0) rw_signal(sync + 1, k, lane);
    rw_signal(sync + 2, k, lane);
    ORL_PROFR(10);
  }
  ORL_PROF_END();
}

// RW: the two-wavefront form above (128 threads per workgroup)
// LDS_ 4 / 5: the rows-deferred forms (round 6) — the window of state 3 / 1 without everything the row phase needed; the loop is
// slot scan + ctrl_d<..., RD> (which changes the slot maps itself and logs events), k_rowstats replays the rest after the launch
#ifdef ORL_TIMING
// (diagnostic builds) per wavefront of the last launch: the constant 100 MHz clock at entry, at the first step, behind the last step and
// at the end of the write-back, HW_ID, XCC_ID — tools/wave_timeline.py reads them through the profile call (mode 2; mode 3: the
// per-wavefront phase sums, unsummed)
static __device__ unsigned long long g_wts[16384 * 8];
#endif
// Fair issue priority.  A SIMD's arbiter picks the OLDEST ready wavefront first: of the four wavefronts of this kernel that share a
// SIMD for a whole launch, the one in slot 0 ran 7 % faster than the mean and the one in slot 3 8 % slower (tools/wave_timeline.py,
// cfg2: 2 227 / 2 325 / 2 443 / 2 599 us for the same 80 steps) — and a launch ends when its slowest wavefront does.  Each wavefront
// therefore sets its user priority (which the arbiter looks at before the age) to (its slot + the constant 100 MHz clock >> shift)
// mod 4 at a few points of every step: the four take turns at each level, and they finish together.
__device__ __forceinline__ void wave_prio_rotate(int slot, int shift) {
  const int p = (slot + (int)(wall_clock64() >> shift)) & 3;
  if (p == 0) __builtin_amdgcn_s_setprio(0);
  else if (p == 1) __builtin_amdgcn_s_setprio(1);
  else if (p == 2) __builtin_amdgcn_s_setprio(2);
  else __builtin_amdgcn_s_setprio(3);
}
template <int ENV, int W, int LDS_, bool PF, bool RW = false>
__device__ __forceinline__ void persist_body(const DevParams& P, int pol, int target, int* wg_step, u32* n_unfinished) {
  constexpr bool RD = (LDS_ == 4 || LDS_ == 5);
  constexpr int LDS = (LDS_ == 4) ? 3 : ((LDS_ == 5) ? 1 : LDS_);
  constexpr bool CP = PersistCompact<ENV, LDS>::value;
  static_assert(!RD || (CP && !RW), "rows deferred: single-core families, one wavefront per 8 envs");
  static_assert(!RW || (CP && (LDS == 1 || LDS == 3)), "two-wavefront form: single-core families, slot maps in LDS");
// (one wavefront per workgroup: a barrier is an ordering point of the wavefront; the control wavefront of a pair must not wait at
// one for the row wavefront, which is in its own loop)
#ifdef ORL_DIAG_WAVE_SYNC  // (A/B: the one-wavefront forms with wavefront-scope ordering points as well)
#define ORL_SYNC() do { wave_fence(); __builtin_amdgcn_wave_barrier(); } while (0)
#else
#define ORL_SYNC() do { if constexpr (RW) { wave_fence(); __builtin_amdgcn_wave_barrier(); } else __syncthreads(); } while (0)
#endif
  constexpr bool SVC = ORL_PERSIST_SVC != 0;
  constexpr bool DS = PersistDeferred<ENV, LDS>::value;  // bookkeeping logged for k_stats (ctrl_d) instead of done in the loop
#ifdef ORL_DIAG_NO_MINI  // (A/B: the control phase's record words stay in the global records)
  constexpr bool MINI = false;
#else
  constexpr bool MINI = DS && (LDS == 0 || LDS == 3);    // ... and the control phase's record words in the LDS window (sp::mrec)
#endif
  const int ICL = (PersistInner<ENV, W, LDS>::value && !RD) ? P.persist_ic : 0;  // (the host decides: only where it costs no wavefront)
  const bool IC = ICL >= 1, OC = ICL >= 2;
  constexpr bool SR = PF;  // soon list in registers: the forms with registers to spare
  const PersistLds L = persist_lds_layout(P.E, P.H, P.bm_words, P.C, LDS, CP, ICL, PersistDeferred<ENV, 0>::value, RD);
  typename sp::SinkEntryOf<CP>::type* s_tab = (typename sp::SinkEntryOf<CP>::type*)(orl_lds_raw + L.tab);
  u32* s_tally = (u32*)(orl_lds_raw + L.tally);
  unsigned short* s_mtab = (unsigned short*)(orl_lds_raw + L.mtab);
  unsigned short* s_list = (unsigned short*)(orl_lds_raw + L.list);
  int* s_deferred = (int*)(orl_lds_raw + L.misc);  // [2], alternating by step
  u32* s_list_n = (u32*)(orl_lds_raw + L.misc + 8);
  u32* rw_sync = (u32*)(orl_lds_raw + L.total);  // (RW: the pair's counters behind the window, then the staged batch of services)
  u32 rw_k = 0u;                                  // (RW: steps of this launch whose items have been listed)
  u32 rw_asked0 = 0u, rw_asked1 = 0u;             // (RW: batches of services asked of the row wavefront so far, per channel)
  u32 rw_early = 0u, rw_early_groups = 0u;        // (RW: the number of the batch asked for a step ahead and not yet taken, 0: none; its groups)
  // (area 0: the batch asked for a step ahead; area 1: one asked for on the spot)
#define ORL_RW_TAKE(AREA, LANE)                                                                                           \
  do {                                                                                                                   \
    const double* q_ = (const double*)((const char*)(rw_sync + ORL_RW_SYNC_WORDS) + (AREA) * (ORL_RW_STAGE_BYTES / 2));    \
    svb.q = q_[LANE]; svb.ht = q_[64 + (LANE)]; svb.pk = ((const u32*)(q_ + 128))[LANE]; svb.cnt = ((const int*)(q_ + 128))[64 + (LANE)]; \
  } while (0)
#define ORL_RW_ASK(AREA, BALLOT, NWANT, LANE)                                                                             \
  do {                                                                                                                   \
    u32 groups_ = 0u;                                                                                                    \
    for (int g_ = 0; g_ < 8; g_++) groups_ |= (u32)(((BALLOT) >> (8 * g_)) & 1ull) << g_;                               \
    if ((LANE) == 0) { rw_sync[6 + 4 * (AREA)] = (u32)(NWANT); rw_sync[7 + 4 * (AREA)] = groups_; }                       \
    if ((AREA) == 0) { rw_asked0++; rw_early = rw_asked0; rw_early_groups = groups_; }                                   \
    else rw_asked1++;                                                                                                    \
    rw_signal(rw_sync + 4 + 4 * (AREA), (AREA) ? rw_asked1 : rw_asked0, (LANE));                                         \
  } while (0)
  if constexpr (RW) {
    if (threadIdx.x < ORL_RW_SYNC_WORDS) rw_sync[threadIdx.x] = 0u;
    __syncthreads();
    if (threadIdx.x >= 64) {
      persist_row_wave<ENV, W, LDS>(P, L, ICL, rw_sync);
      return;
    }
  }
  const int lane = lane_id();
  const i64 env0 = (i64)blockIdx.x * 8;
  const i64 env = env0 + (threadIdx.x >> 3);
  const int nenv = (int)(P.B - env0 < 8 ? P.B - env0 : 8);
  int step = wg_step[blockIdx.x];
  const int fair = RW ? 0 : P.persist_fair;  // (the pair form's two wavefronts depend on each other: left to the arbiter)
  const int wslot = (int)__builtin_amdgcn_s_getreg(4 | (0 << 6) | (3 << 11));  // HW_ID.WAVE_ID: the wavefront's slot on its SIMD
  sp::Prof prof;
#ifdef ORL_TIMING
  unsigned long long wts0 = wall_clock64(), wts1 = 0, wts2 = 0;
#endif
  // An env whose releases did not fit the item form in the last step of the previous launch (flag in its record; the
  // wavefront left its loop right after that step's row phase): they are released in place, by the wavefront that owns
  // the env, before anything of this launch uses the state.  (A kernel of its own after every launch — round 1, and
  // the two-kernel form still — had to wait for a free CU while the other half's launch filled the GPU: 0.5-1 ms on the
  // stream between two 3 ms launches, rocprofv3 kernel trace r2n.)  The flag is requested here and looked at after the
  // LDS window has been requested too (one memory round trip instead of two at the start of every wavefront); in the rare
  // case the window is filled again afterwards.
  const bool pend = (env < P.B) && ((P.scal[env * ORL_SCAL_WORDS + SC_ACC] >> 16) & 1ull) != 0ull;
  const bool valid = env < P.B;
  // carried from step to step in registers: the pending service's descriptor and (SR) this lane's entries of the env's soon
  // list.  Requested together with the LDS window, before the row caches are built from it (their latency hides behind
  // that); requested again in the rare case that pending releases are done first.
  u64 desc = 0ull;
  sp::SoonRegs soon_c;
  soon_c.dirty = 0;
  sp::SvcBuf svb;
  svb.q = 0.0; svb.ht = 0.0; svb.pk = 0u; svb.cnt = 0;
  int esp_c = 0;        // DS: the env's episode step counter (SC_ESP; the replay keeps the record's copy)
  int prev_core = 0;    // DS, RMCSA: the core of the env's last accepted provision (the high half of SC_ACC)
  int ecur = 0;         // RD: events this lane's env has logged in this launch
  u64 now0_w = 0ull;    // DS: the clock the launch starts at (logged for the replay)
#define ORL_LOAD_CARRIED()                                                                                  \
  do {                                                                                                      \
    desc = valid ? P.svc_desc[env] : 0ull;                                                                  \
    if (DS && valid && step < target) {                                                                     \
      esp_c = (int)P.scal[env * ORL_SCAL_WORDS + SC_ESP];                                                   \
      if (ENV == ENV_RMCSA) prev_core = (int)((P.scal[env * ORL_SCAL_WORDS + SC_ACC] >> 32) & 31ull);       \
      now0_w = P.scal[env * ORL_SCAL_WORDS + SC_NOW];                                                       \
    }                                                                                                       \
    if (SVC && valid && step < target) {                                                                    \
      const size_t sl_ = (size_t)blockIdx.x * 64 + (size_t)lane;                                            \
      svb.cnt = P.svc_cnt[sl_];                                                                             \
      if (!sp::svc_empty(svb)) { svb.q = P.svc_q[sl_]; svb.ht = P.svc_ht[sl_]; svb.pk = P.svc_pk[sl_]; }   \
    }                                                                                                       \
    _Pragma("unroll") for (int k = 0; k < ORL_SOON_PER_LANE; k++) {                                         \
      const bool ld = SR && valid && step < target;                                                         \
      soon_c.t[k] = ld ? P.soon_t[env * ORL_SOON + (lane & 7) + 8 * k] : __builtin_inf();                   \
      soon_c.i[k] = ld ? (int)P.soon_i[env * ORL_SOON + (lane & 7) + 8 * k] : 0;                            \
    }                                                                                                       \
  } while (0)
  sp::Wmem M = sp::wmem_global(P);
  constexpr bool REC = (LDS == 1 || LDS == 2);  // the env records are in the LDS window
  if (LDS != 2 && !RD) {
    M.clk = (double*)(orl_lds_raw + L.clk);
    M.clk_env0 = env0;
  }
  if (LDS >= 1) {  // the wavefront's envs are contiguous in every array: coalesced 16-byte loads
    M.bm0 = (u64*)(orl_lds_raw + L.bm);
    M.env0 = env0;
    M.cs0 = (int*)(orl_lds_raw + L.cs);
    M.cenv0 = env0;
    M.cs_lds = true;
    M.cs_stride = L.csw;
    if (REC) {
      M.sc0 = (u64*)(orl_lds_raw + L.sc);
      M.scenv0 = env0;
      M.sc_stride = ORL_SCAL_LDS_WORDS;
    }
#define ORL_FILL_WINDOW() persist_fill_window<REC, RD>(P, env0, nenv, lane, orl_lds_raw + L.bm, orl_lds_raw + L.sc, orl_lds_raw + L.cs, RD ? 0 : L.csw / 4)
    if (step < target) ORL_FILL_WINDOW();
    ORL_LOAD_CARRIED();
    // the row caches this wavefront left with the state at the end of its previous launch are still good when nothing but the
    // persistent kernel has touched the slot maps since (the host's key) and no release is pending in place
    bool cache_stored = false;
    if (IC && P.row_cache_key != 0) {
      const int stamp = P.row_cache_stamp[blockIdx.x];
      cache_stored = (stamp == P.row_cache_key) && __ballot(pend) == 0ull;
    }
    if (__ballot(pend) != 0ull) {
      if (IC && P.row_cache_key != 0 && threadIdx.x == 0) P.row_cache_stamp[blockIdx.x] = 0;  // (rel_serial writes slot maps)
      if (pend) {
        sp::rel_serial<ENV, W>(P, env, lane);
        if (ENV == ENV_DEEPRMSA && P.obs_dim) {
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
          obs8_env<W>(P, P.bitmap + env * P.bm_words, P.scal + env * ORL_SCAL_WORDS, env, lane, P.done[env]);
        }
      }
      __threadfence();
      ORL_SYNC();
      if (step < target) ORL_FILL_WINDOW();
      ORL_LOAD_CARRIED();
    }
#undef ORL_FILL_WINDOW
    if (IC) {
      M.ic0 = (u32*)(orl_lds_raw + L.ic);
      const u32* g_cache = P.row_cache + (size_t)blockIdx.x * 2 * P.row_cache_words;
      if (step < target) {
        ORL_SYNC();  // the rows are in LDS
        if (cache_stored) {  // (both levels with one batch of requests)
          for (int i = lane; i < nenv * P.E; i += 64) {
            const u32 a = g_cache[i], c = OC ? g_cache[P.row_cache_words + i] : 0u;
            M.ic0[i] = a;
            if (OC) ((u32*)(orl_lds_raw + L.ic + ((8 * P.E * 4 + 15) & ~15)))[i] = c;
          }
        } else {
          for (int i = lane; i < nenv * P.E; i += 64) M.ic0[i] = sp::row_inner_cache<W>(M.bm0 + (size_t)(i / P.E) * P.bm_words + (size_t)(i % P.E) * W);
        }
      }
      // what every row contributes to the compactness sums, as the launch finds it (kept exact by the row phase): in the LDS
      // window at cache level 2; at level 1 — no room for them there — the wavefront's level-2 words of DevParams::row_cache are
      // used in place, in global memory (round 5: a load and a store of 4 bytes per work item beside its link record instead of
      // ~90 instructions recomputing the value from the five words of the row)
#ifdef ORL_DIAG_NO_OCG
      const bool OCG = false;
#else
      const bool OCG = !OC && P.row_cache_key != 0;
#endif
      if (OC) M.oc0 = (u32*)(orl_lds_raw + L.ic + ((8 * P.E * 4 + 15) & ~15));
      if (OCG) M.ocg = P.row_cache + (size_t)blockIdx.x * 2 * P.row_cache_words + P.row_cache_words;
      if ((OC || OCG) && step < target && !cache_stored)
        for (int i = lane; i < nenv * P.E; i += 64) {
          const u64* row = M.bm0 + (size_t)(i / P.E) * P.bm_words + (size_t)(i % P.E) * W;
          u64 a[W];
#pragma unroll
          for (int w = 0; w < W; w++) a[w] = row[w];
          int occ, fb;
          sp::row_occ_fb<W>(a, P.S, occ, fb);
          if (OC) M.oc0[i] = ((u32)occ << 16) | (u32)fb;
          else M.ocg[i] = ((u32)occ << 16) | (u32)fb;
        }
    }
  }
  if (LDS == 0) {
    if (__ballot(pend) != 0ull) {
      if (pend) {
        sp::rel_serial<ENV, W>(P, env, lane);
        if (ENV == ENV_DEEPRMSA && P.obs_dim) {
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
          obs8_env<W>(P, P.bitmap + env * P.bm_words, P.scal + env * ORL_SCAL_WORDS, env, lane, P.done[env]);
        }
      }
      __threadfence();
      ORL_SYNC();
    }
    ORL_LOAD_CARRIED();
  }
#undef ORL_LOAD_CARRIED
  // Global-state form (round 5): the per-core sums of the 8 envs — 16 bytes each for a single core — in the LDS window for the
  // launch.  In global memory every work item of the row phase updated them with L2 atomics, and the control phase read them
  // back through L2 with four RETURNING atomics per step, a global round trip in the step's dependent chain.
  constexpr bool CS0 = (LDS == 0) && PersistDeferred<ENV, 0>::value && !RD;
  if constexpr (CS0) {
    M.cs0 = (int*)(orl_lds_raw + L.cs);
    M.cenv0 = env0;
    M.cs_lds = true;
    M.cs_stride = L.csw;
    if (step < target) {
      const int q = L.csw / 4;
      for (int i = lane; i < nenv * q; i += 64)
        ((orl_i32x4*)M.cs0)[i] = ((const orl_i32x4*)(P.core_sums + (env0 + i / q) * P.cs_words))[i % q];
    }
    wave_fence();
  }
  if (LDS == 2) {
    M.ls0 = (double*)(orl_lds_raw + L.ls);
    M.senv0 = env0;
    if (step < target) {
      const double2* gs = (const double2*)(P.lstat + env0 * 4 * P.E);
      double2* ls = (double2*)M.ls0;
      for (int i = lane; i < nenv * 2 * P.E; i += 64) ls[i] = gs[i];
    }
  }
  if (threadIdx.x == 0) s_deferred[0] = s_deferred[1] = 0;
  const int first_step = step;
  bool left_pending = false;  // the loop ended on a step whose releases are still to be done
  if (DS && valid && step < target && (lane & 7) < 2) {
    // header row of the log: the clock the first step is decided at, the bit-rate index of the service it serves
    // (words 0 and 2 of that row: its word 1 is where a wavefront that logs log_cap steps leaves its last sums)
    P.slog[(size_t)(3 * P.log_cap + 2 * (lane & 7)) * (size_t)P.log_stride + (size_t)env] = (lane & 7) == 0 ? now0_w : ((desc >> 32) & 0xffffull);
  }
  if (MINI) {
    M.mini = (u64*)(orl_lds_raw + L.mini);
    M.mini_env0 = env0;
    if (valid && step < target && (lane & 7) < ORL_MINI_WORDS)
      M.mini[(lane >> 3) * ORL_MINI_STRIDE + (lane & 7)] = P.scal[env * ORL_SCAL_WORDS + sp::mini_slot(lane & 7)];
    wave_fence();
  }
  // Small batches (the two-wavefront form, when the host found room: persist_evl): the pending release times of the 8 envs in LDS
  // for the launch — the rebuild scan of the release detection walks them in five dependent rounds, global round trips of
  // 1 000-2 000 cycles each on a GPU this empty
  const bool EVL = RW && P.persist_evl != 0;
  double* s_ev = (double*)(orl_lds_raw + L.total + ORL_RW_EXTRA_BYTES);
  if (EVL) {
    if (step < target) {
      const double* g = P.ev_time + env0 * P.ev_cap;
      const int n = nenv * P.ev_cap;
      for (int base = 0; base < n; base += 64 * 8) {
        double v[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { const int i = base + 64 * k + lane; v[k] = (i < n) ? g[i] : 0.0; }
#pragma unroll
        for (int k = 0; k < 8; k++) { const int i = base + 64 * k + lane; if (i < n) s_ev[i] = v[k]; }
      }
      wave_fence();
    }
    M.evl0 = s_ev;
  }
  ORL_PROF_BEGIN();
#ifdef ORL_TIMING
  wts1 = wall_clock64();
#endif
  // (DS: a wavefront that caught up over more steps than a launch can log stops there and counts as unfinished)
  while (step < target && (!DS || step - first_step < P.log_cap)) {
    ORL_SYNC();  // (one wavefront: an ordering point) the previous row phase's writes are done
    if (threadIdx.x == 0) s_deferred[(step + 1) & 1] = 0;
    // per-iteration opaque copies: without them the compiler hoists every per-lane address out of the loop and keeps
    // them all live across both phases
    int env_lo = (int)env, lane_i = lane;
    asm volatile("" : "+v"(env_lo), "+v"(lane_i));
    const i64 env_i = (i64)env_lo;
    const bool valid_i = env_i < P.B;
    if (fair) wave_prio_rotate(wslot, fair);
    // (ORL_DIAG_*: hook points of the diagnostic builds, orl_diag.h)
    if (SVC) {  // a group whose batch of services is used up draws the next one: as many as the launch has steps left, 8 at most
      const bool need = valid_i && sp::svc_empty(svb);
#ifdef ORL_DIAG_INSTEAD_OF_SERVICES
      ORL_DIAG_INSTEAD_OF_SERVICES
#else
      if (__ballot(need) != 0ull) {
        const int left = target - step;
        if constexpr (RW) {
          // the batch the row wavefront drew while this one was in its previous step (asked for below), or — the first step of a
          // launch, groups out of phase — one asked for now
          bool want = need;
          const bool early_mine = ((rw_early_groups >> (lane_i >> 3)) & 1u) != 0u;
          if (rw_early != 0u && __ballot(want && early_mine) != 0ull) {  // (its groups run dry together: all of them take it now)
            rw_wait(rw_sync + 5, rw_early);
            if (want && early_mine) { ORL_RW_TAKE(0, lane_i); want = false; }
            rw_early = 0u;
            rw_early_groups = 0u;
          }
          if (__ballot(want) != 0ull) {  // the first step of a launch, or a group out of phase with the others: on the spot
            ORL_RW_ASK(1, __ballot(want), (left < 8 ? left : 8), lane_i);
            rw_wait(rw_sync + 9, rw_asked1);
            if (want) ORL_RW_TAKE(1, lane_i);
          }
          wave_fence();  // (the slots are read before a later request can overwrite them)
        } else {
          sp::svc_generate<ENV>(P, sp::wm_scal(P, M, valid_i ? env_i : M.scenv0), P.mt + (valid_i ? env_i : 0) * 624, lane_i, left < 8 ? left : 8, svb, need);
        }
      }
#endif
    }
    ORL_PROFA(0);
    sp::CtrlOpts O;
    O.persistent = true;
    O.write_io = (step + 1 == target);  // what the host can see after the run: the last step's action / reward / done
    O.trusted = true;
    O.emit_queue = false;
    O.prefetch = PF;
    O.auto_reset = true;
    O.rank_pairs = SR;
    int done_i = 0;
    {
      int a[4];
      // (deferred-statistics control phase: the top of the env's free-slot stack is requested before the scan and the chosen
      // path's slot count and record come out of it — nothing the decision needs is fetched behind the scan)
      ScanHand hand;
      hand.n = 1; hand.q[0] = hand.q[1] = hand.q[2] = hand.q[3] = 0ull;
      hand.words = (P.H + 2 + 7) >> 3;
#ifdef ORL_DIAG_INSTEAD_OF_SCAN
      constexpr bool HAND = false;
      int pop_pre = -2;
      ORL_DIAG_INSTEAD_OF_SCAN
#else
      constexpr bool HAND = DS && ENV != ENV_RMCSA;
      int pop_pre = -2;
      if (HAND) {
        pop_pre = -1;
        if (valid_i) {
          u64* rec_i = sp::wm_scal(P, M, env_i);
          const int nfree = (int)(u32)*sp::mrec<MINI>(M, rec_i, env_i, SC_HINT);
          if (nfree > 0) pop_pre = (int)((const unsigned short*)(rec_i + SC_FREE0))[nfree - 1];
        }
      }
      policy_g<ENV, W, 8>(P, sp::wm_bm(P, M, valid_i ? env_i : M.env0), valid_i, (int)(u32)desc, (int)((desc >> 32) & 0xffffu),
                          (int)((desc >> 48) & 0xffu), lane_i, pol, (pol == POL_PATH_FF && valid_i) ? P.path_col[env_i] : 0, a,
                          HAND ? &hand : nullptr);
#endif
      const int4 av = make_int4(a[0], a[1], a[2], a[3]);
      ORL_PROFA(1);
      if (fair) wave_prio_rotate(wslot, fair);
      if constexpr (DS) {
        u64* slog_s = P.slog + (size_t)(step - first_step) * ORL_SLOG_WORDS * (size_t)P.log_stride + (size_t)(valid_i ? env_i : 0);
        desc = sp::ctrl_d<ENV, W, CP, MINI, RW, RD>(P, M, O, env_i, valid_i, lane_i, prof, av, desc, s_tab, s_tally, L.tw, &s_deferred[step & 1], &done_i,
                                              s_list, s_list_n, SR ? &soon_c : nullptr, s_mtab, svb, esp_c, prev_core, slog_s,
                                              HAND ? &hand : nullptr, pop_pre, rw_sync, rw_k, &ecur, step - first_step);
      } else {
        desc = sp::ctrl_a<ENV, W, CP>(P, M, O, env_i, valid_i, lane_i, prof, &av, s_tally, s_tab, 0, &s_deferred[step & 1], &done_i,
                                      s_list, s_list_n, L.tw, SR ? &soon_c : nullptr, s_mtab, nullptr, SVC ? &svb : nullptr);
      }
    }
    ORL_SYNC();  // sink table + item list, clocks, env records
    if (fair) wave_prio_rotate(wslot, fair);
    if constexpr (RW) {  // the row wavefront takes the items (the slot maps are up to date: ctrl_d applied the masks); this one goes on
      rw_k++;
      rw_signal(rw_sync + 0, rw_k, lane_i);
      if constexpr (SVC) {
        // the next batch of services, asked for when this one holds one more: the next step — certain to run: this one was not cut
        // short and the launch has room — takes it, the one after finds the new batch drawn
        const bool one_left = valid_i && ((svb.cnt & 0xff) + 1 == (svb.cnt >> 8));
        const int lim = DS ? first_step + P.log_cap : target;
        const int after = target - step - 2;  // steps of the run left once the last service of this batch is taken
        if (s_deferred[step & 1] == 0 && step + 1 < (lim < target ? lim : target) && after >= 1 && rw_early == 0u) {
          const u64 bw = __ballot(one_left);
          if (bw != 0ull) ORL_RW_ASK(0, bw, (after < 8 ? after : 8), lane_i);
        }
      }
      ORL_PROFA(12);
    } else if constexpr (!RD) {
#ifdef ORL_DIAG_NO_ROWS
      const int n_items = 0;
#else
      const int n_items = (int)*s_list_n;
#endif
      ORL_PROFA(12);
      for (int idx = lane_i; idx < n_items; idx += 64) {
        const int code = (int)s_list[idx];
        const int el = (code >> 8) & 7, link = code & 0xff, second = code >> 15;
        if constexpr (!CP) {  // RMCSA: 24-byte entries with a core per mask, the general loop
          if (!second) sp::row_item_lane<ENV, W>(P, M, sp::item_from_sink(env0 + el, link, s_tab[P.E * el + link]), prof);
        } else {
          sp::row_item_lane1<ENV, W>(P, M, env0 + el, link, s_tab[P.E * el + link].bits, s_mtab + ORL_MTAB * el, second, prof, PF);
        }
      }
      ORL_PROFA(13);
    }
    const bool deferred = s_deferred[step & 1] != 0;  // set before the barrier in front of the row phase
    if (ENV == ENV_DEEPRMSA && (O.write_io || deferred)) {  // the observation of the new pending service, from the rows as they are now
      ORL_SYNC();
      if (valid_i) {
        if (MINI) {  // (the pending service's words are the descriptor's)
          u64 sd, br;
          sp::svc_words<ENV>(P, desc, sd, br);
          obs8_env_w<W>(P, sp::wm_bm(P, M, env_i), sd, br, env_i, lane_i, done_i);
        } else {
          obs8_env<W>(P, sp::wm_bm(P, M, env_i), sp::wm_scal(P, M, env_i), env_i, lane_i, done_i);
        }
      }
    }
    step++;
    if (deferred) { left_pending = true; break; }
    // (RD: an env whose event log could not take another step's provision and releases: the wavefront stops here and counts as
    // unfinished, like one that has used up the statistics log)
    if constexpr (RD) { if (__ballot(valid_i && ecur + 1 + P.rel_limit > P.elog_cap) != 0ull) break; }
  }
  ORL_PROF_END();
#ifdef ORL_TIMING
  wts2 = wall_clock64();
#endif
  if constexpr (RW) {
    rw_wait(rw_sync + 2, rw_k);
    if (SVC && rw_early != 0u) {  // a batch on order: its groups' own are used up (see above); parked below like any other
      rw_wait(rw_sync + 5, rw_early);
      if (valid && ((rw_early_groups >> (lane >> 3)) & 1u) != 0u) ORL_RW_TAKE(0, lane);
    }
    rw_signal(rw_sync + 3, 1u, lane);
  }
  const bool finished_run = P.persist_finish && step > first_step && !left_pending;
  if (DS) {
    // the sums after the last row phase (the replay finishes the last step's pending network-compactness update from them when
    // the run ends here), and how many steps this wavefront logged
    if (step > first_step) {
      ORL_SYNC();
      int tid_t = (int)threadIdx.x;
      asm volatile("" : "+v"(tid_t));
      const i64 env_t = env0 + (tid_t >> 3);
      if (ENV != ENV_RWA && !RD && env_t < P.B && (tid_t & 7) == 0) {  // (RD: k_rowstats writes them)
        int* cs = sp::wm_cs(P, M, env_t);
        int* rs = cs + 2 * P.C;
        const int pc = (ENV == ENV_RMCSA) ? prev_core : 0;  // (this lane's group is env_t's)
        int occ, fb;
        if (!M.cs_lds) { occ = atomicAdd(cs + 2 * pc, 0) - atomicAdd(rs + 2 * pc, 0); fb = atomicAdd(cs + 2 * pc + 1, 0) - atomicAdd(rs + 2 * pc + 1, 0); }
        else { occ = cs[2 * pc] - rs[2 * pc]; fb = cs[2 * pc + 1] - rs[2 * pc + 1]; }
        P.slog[(size_t)(3 * (step - first_step) + 1) * (size_t)P.log_stride + (size_t)env_t] = sp::slog_w1(false, 0, 0, occ, fb);
      }
    }
    if (threadIdx.x == 0) P.log_n[blockIdx.x] = (step - first_step) | (finished_run ? (1 << 16) : 0);
    if constexpr (RD) {
      int tid_e = (int)threadIdx.x;
      asm volatile("" : "+v"(tid_e));
      const i64 env_e = env0 + (tid_e >> 3);
      if (env_e < P.B && (tid_e & 7) == 0) P.elog_n[env_e] = ecur;
    }
  } else if (PersistDeferred<ENV, 0>::value && threadIdx.x == 0) {
    P.log_n[blockIdx.x] = 0;  // (a form that keeps the bookkeeping in the loop: nothing for k_stats)
  }
  if (P.persist_finish && step > first_step && !left_pending) {
    // the end of a run: what k_finish2 (orl_api.hip) does for every env in a launch of its own — the network-compactness update
    // the last step left pending (rmsa_env.py:439-462 with the sums right after that step's provision), the release part of
    // the sums cleared, the env's flag word reported — on the records and sums where this wavefront has them
    ORL_SYNC();  // the last row phase is done
    u32 f = 0u;
    // (the env index recomputed from the thread index behind an opaque copy: kept live across the step loop for this block it
    // cost the 128-VGPR forms a spilled register)
    int tid_f = (int)threadIdx.x;
    asm volatile("" : "+v"(tid_f));
    const i64 env_f = env0 + (tid_f >> 3);
    if (env_f < P.B && (tid_f & 7) == 0) {
      u64* s = sp::wm_scal(P, M, env_f);
      int* cs = sp::wm_cs(P, M, env_f);
      int* rs = cs + 2 * P.C;
      const u64 acc = DS ? 0ull : s[SC_ACC];  // (DS: the replay finishes the pending update, from the sums logged above)
      if ((u32)acc & 2u) {
        const int c0 = (int)((acc >> 32) & 31);
        const i64 s_nh_prov = (i64)(acc >> 37);
        int occ, fb;
        if (!M.cs_lds) {  // (global sums are updated by L2 atomics: read through L2 as well)
          occ = atomicAdd(cs + 2 * c0, 0) - atomicAdd(rs + 2 * c0, 0);
          fb = atomicAdd(cs + 2 * c0 + 1, 0) - atomicAdd(rs + 2 * c0 + 1, 0);
        } else {
          occ = cs[2 * c0] - rs[2 * c0];
          fb = cs[2 * c0 + 1] - rs[2 * c0 + 1];
        }
        const double a0 = __longlong_as_double((i64)s[SC_GC_A]), td = __longlong_as_double((i64)s[SC_GC_TD]);
        const double now_a = __longlong_as_double((i64)s[SC_NOWA]);
        const double cmp = (fb > 0) ? ((double)occ / (double)s_nh_prov) * ((double)P.E / (double)fb) : 1.0;
        s[SC_GCOMP] = (u64)__double_as_longlong((a0 + (cmp * td)) / now_a);
        s[SC_ACC] = acc & ~2ull;
      }
      if constexpr (!RD) {  // (RD: the sums are k_rowstats' — it clears the release part when the run ends here)
        for (int i = 0; i < 2 * P.C; i++) {
          if (!M.cs_lds) atomicExch(rs + i, 0);
          else rs[i] = 0;
        }
      }
      const u64 v = s[SC_FLAGS];
      f = (u32)(v >> 32);
      if (f & ORL_FLAG_BAD_ACTION) s[SC_FLAGS] = v & ~((u64)ORL_FLAG_BAD_ACTION << 32);
    }
    for (int o = 32; o > 0; o >>= 1) f |= (u32)__shfl_xor((int)f, o, 64);
    if ((tid_f & 63) == 0 && f) atomicOr(n_unfinished + 1, f);
  }
  if (EVL && step > first_step) {  // (the release times go back)
    wave_fence();
    double* g = P.ev_time + env0 * P.ev_cap;
    const int n = nenv * P.ev_cap;
    for (int i = lane; i < n; i += 64) g[i] = s_ev[i];
  }
  if (CS0 && step > first_step) {
    wave_fence();
    const int q = L.csw / 4;
    for (int i = lane; i < nenv * q; i += 64)
      ((int4*)(P.core_sums + (env0 + i / q) * P.cs_words))[i % q] = ((const int4*)M.cs0)[i];
  }
  if (LDS >= 1 && step > first_step) {
    ORL_SYNC();
    ulonglong2* g = (ulonglong2*)(P.bitmap + env0 * P.bm_words);
    const ulonglong2* l = (const ulonglong2*)M.bm0;
    for (int i = lane; i < nenv * (P.bm_words / 2); i += 64) g[i] = l[i];
    if constexpr (!RD) {  // (RD: the sums are k_rowstats')
      const int q = L.csw / 4;
      for (int i = lane; i < nenv * q; i += 64)
        ((int4*)(P.core_sums + (env0 + i / q) * P.cs_words))[i % q] = ((const int4*)M.cs0)[i];
    }
    if (IC && P.row_cache_key != 0) {  // the row caches go with the state; the stamp says under which key they were written
      u32* g_cache = P.row_cache + (size_t)blockIdx.x * 2 * P.row_cache_words;
      for (int i = lane; i < nenv * P.E; i += 64) {
        g_cache[i] = M.ic0[i];
        if (OC) g_cache[P.row_cache_words + i] = M.oc0[i];
      }
      if (threadIdx.x == 0) P.row_cache_stamp[blockIdx.x] = left_pending ? 0 : P.row_cache_key;
    }
    if (REC) {
      ulonglong2* gr = (ulonglong2*)(P.scal + env0 * ORL_SCAL_WORDS);
      const ulonglong2* lr = (const ulonglong2*)M.sc0;
      for (int i = lane; i < nenv * (ORL_SCAL_WORDS / 2); i += 64) gr[i] = lr[(i >> 4) * (ORL_SCAL_LDS_WORDS / 2) + (i & 15)];
    }
  }
  if (LDS == 2 && step > first_step) {
    double2* gs = (double2*)(P.lstat + env0 * 4 * P.E);
    const double2* ls = (const double2*)M.ls0;
    for (int i = lane; i < nenv * 2 * P.E; i += 64) gs[i] = ls[i];
  }
  // (the env index of the write-backs below recomputed from the thread index behind an opaque copy: kept live across the step loop —
  // 64 bits that nothing in the loop reads — it was the one value the 128-VGPR specialisations spilled, 2 VGPRs / 12 B of scratch)
  int tid_w = (int)threadIdx.x;
  asm volatile("" : "+v"(tid_w));
  const i64 env_w = env0 + (tid_w >> 3);
  const bool valid_w = env_w < P.B;
  if (MINI && step > first_step && valid_w) {
    // the record words the control phase kept in the window go back to the records (SC_AT: the pending service arrived at the clock)
    wave_fence();
    const int k = lane & 7;
    if (k < ORL_MINI_WORDS) {
      const u64 v = M.mini[(lane >> 3) * ORL_MINI_STRIDE + k];
      P.scal[env_w * ORL_SCAL_WORDS + sp::mini_slot(k)] = v;
      if (k == 0) P.scal[env_w * ORL_SCAL_WORDS + SC_AT] = v;
    } else if (k == ORL_MINI_WORDS) {
      u64 sd, br;
      sp::svc_words<ENV>(P, desc, sd, br);
      P.scal[env_w * ORL_SCAL_WORDS + SC_SRC_DST] = sd;
      P.scal[env_w * ORL_SCAL_WORDS + SC_BR_IDX] = br;
    }
  }
  if (SVC && step > first_step && valid_w) {
    // what the group drew ahead and did not use: nothing when the loop ran to the launch's target; a wavefront that left early
    // (releases to be done in place) parks it for its next launch
    const size_t sl = (size_t)blockIdx.x * 64 + (size_t)lane;
    P.svc_cnt[sl] = svb.cnt;
    if (!sp::svc_empty(svb)) { P.svc_q[sl] = svb.q; P.svc_ht[sl] = svb.ht; P.svc_pk[sl] = svb.pk; }
  }
  if (SR && step > first_step && valid_w) {
#pragma unroll
    for (int k = 0; k < ORL_SOON_PER_LANE; k++) {
      P.soon_t[env_w * ORL_SOON + (lane & 7) + 8 * k] = soon_c.t[k];
      P.soon_i[env_w * ORL_SOON + (lane & 7) + 8 * k] = (u32)soon_c.i[k];
    }
  }
  if (step > first_step && step < target && valid_w) {
    // leaving early (deferred releases): the descriptor the next launch / the stand-alone scan reads; action, reward and done
    // of an unfinished run are not host-visible
    if ((lane & 7) == 0) P.svc_desc[env_w] = desc;
  }
  if (threadIdx.x == 0) {
    wg_step[blockIdx.x] = step;
    if (step < target || left_pending) atomicAdd(n_unfinished, 1u);  // (pending releases: the next launch starts with them)
  }
#ifdef ORL_TIMING
  if (threadIdx.x == 0 && blockIdx.x < 16384) {
    __builtin_amdgcn_s_waitcnt(0);
    g_wts[blockIdx.x * 8 + 0] = wts0; g_wts[blockIdx.x * 8 + 1] = wts1; g_wts[blockIdx.x * 8 + 2] = wts2; g_wts[blockIdx.x * 8 + 3] = wall_clock64();
    g_wts[blockIdx.x * 8 + 4] = (unsigned long long)__builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
    g_wts[blockIdx.x * 8 + 5] = (unsigned long long)__builtin_amdgcn_s_getreg((20) | (0 << 6) | (31 << 11));
  }
#endif
}
#undef ORL_SYNC
#undef ORL_RW_TAKE
#undef ORL_RW_ASK
// Register budgets: WAVES waves/SIMD -> 512 / WAVES VGPRs.  Global state: 4 (128 VGPRs) for NSFNET-sized RMSA / RWA /
// DeepRMSA, 3 (168, no spills) for the heavier RMCSA and Germany50 steps.  LDS state: the LDS window decides the residency
// (orl_launch::persist), the kernel is built for 2 or 3.
// Specialisations.  The sizes of a topology and of the traffic model reach the kernel as ~20 uniform scalars of DevParams
// (kept in SGPRs for the whole loop — or spilled to VGPR lanes — and multiplied, compared and looped over at run time).
// With SPEC = 1 they are compile-time constants: the compiler folds the address arithmetic, unrolls the per-word loops and
// halves the SGPR spills (cfg2 +6 %, cfg3 +12 %; same source, same results).  Round 2 carried a table of the BASELINE
// configurations inside the library; now ANY configuration gets its own instantiation, built on first use: this file
// compiled with -DORL_SPEC_ONLY and the sizes as -DORL_SPEC_* macros (optical_rl_gym_amd/_build.py build_spec, the flags
// from orl_batch_spec_flags) into a small shared library of its own — one k_persist instantiation and the launch entry
// orl_spec_launch — which orl_batch_load_spec attaches to the batch after comparing every field.  The main library holds the
// generic kernels only.
struct PersistSpec { int env, W, lds, waves, N, E, K, H, M, S, C, J, bit_rate_mode, br_lo, n_br, rand_n, rand_bits, ev_cap, bm_words, cs_words, obs_dim, n_info; };
#ifdef ORL_SPEC_ONLY
#ifndef ORL_SPEC_RW
#define ORL_SPEC_RW 0  // 1: the two-wavefront form (small batches)
#endif
static constexpr PersistSpec kPersistSpec = {ORL_SPEC_ENV, ORL_W, ORL_SPEC_LDS, ORL_SPEC_WAVES + 16 * ORL_SPEC_RW, ORL_SPEC_N, ORL_SPEC_E, ORL_SPEC_K, ORL_SPEC_H,
                                             ORL_SPEC_M, ORL_SPEC_S, ORL_SPEC_C, ORL_SPEC_J, ORL_SPEC_BRMODE, ORL_SPEC_BRLO, ORL_SPEC_NBR,
                                             ORL_SPEC_RANDN, ORL_SPEC_RANDBITS, ORL_SPEC_EVCAP, ORL_SPEC_BMWORDS, ORL_SPEC_CSWORDS,
                                             ORL_SPEC_OBSDIM, ORL_SPEC_NINFO};
#else
static constexpr PersistSpec kPersistSpec = {};  // (the generic library never instantiates SPEC > 0)
#endif
template <int SPEC> __device__ __forceinline__ void persist_spec_apply(DevParams& P) {
  if constexpr (SPEC > 0) {
    constexpr PersistSpec s = kPersistSpec;
    P.N = s.N; P.E = s.E; P.K = s.K; P.H = s.H; P.M = s.M; P.S = s.S; P.W = s.W; P.C = s.C; P.J = s.J;
    P.bit_rate_mode = s.bit_rate_mode; P.br_lo = s.br_lo; P.n_br = s.n_br; P.rand_n = s.rand_n; P.rand_bits = s.rand_bits;
    P.ev_cap = s.ev_cap; P.bm_words = s.bm_words; P.cs_words = s.cs_words; P.obs_dim = s.obs_dim; P.n_info = s.n_info;
  }
}

// ORL_PERSIST_VGPR (specialisation libraries of the rows-deferred forms, experiments): an explicit register budget below the one
// the waves-per-SIMD figure gives — 4 x 104 leave room for a wavefront of k_rowstats (96) beside the four of this kernel
#ifdef ORL_PERSIST_VGPR
#define ORL_PERSIST_VGPR_ATTR __attribute__((amdgpu_num_vgpr(ORL_PERSIST_VGPR)))
#else
#define ORL_PERSIST_VGPR_ATTR
#endif
template <int ENV, int W, int LDS, int WAVES, int SPEC = 0, bool RW = false>
__global__ void __launch_bounds__(RW ? 128 : 64) __attribute__((amdgpu_waves_per_eu(WAVES, WAVES))) ORL_PERSIST_VGPR_ATTR
k_persist(DevParams P, int pol, int target, int* wg_step, u32* n_unfinished, u32* clear_next) {
  // the counters the NEXT launch of this half of the batch uses (it starts after this one has ended)
  if (blockIdx.x == 0 && threadIdx.x == 0) { clear_next[0] = 0u; clear_next[1] = 0u; }
  persist_spec_apply<SPEC>(P);
  #ifndef ORL_PF_WAVES
#define ORL_PF_WAVES 3  // forms of at most this many waves per SIMD keep the soon list in registers and request early
#endif
  persist_body<ENV, W, LDS, (WAVES <= ORL_PF_WAVES), RW>(P, pol, target, wg_step, n_unfinished);
}


// ---- one step for an agent in the loop -----------------------------------------------------------------------------------
// step() of every env with the actions in P.actions (written by the host, or by an agent on the same GPU), as SB3's VecEnv
// drives it (auto reset): the control phase and the row phase of k_persist for ONE step — 8 lanes per env, 8 envs per
// wavefront, all state in global memory, the per-step tables in LDS — plus what a host-visible step() owes beyond the
// device-resident loop: the action is validated (is_path_free), reward / done / info / observation are written, the
// network-compactness update is finished in the same launch, and releases that do not fit the item form are done in place
// right away, so that every launch leaves final state.  All four families, both bit-rate modes (the discrete mode's per-rate
// blocking entries and their spread, io[8..], are written by the control phase; tests: the g9 discrete-bit-rate fixtures under
// the forced `agent8` form); the one-wavefront-per-env kernel k_step serves batches below 2 048 envs and reseeded batches
// (305 us per 65 536 cfg2 envs against ~84 us here); QoSConstrainedRA has k_agent_qos below.  RWA's info carries the action probabilities (written
// beside the histogram update of the control phase), RMCSA's the four blocking rates.
// info (rmsa_env.py:234-264): the four blocking rates from the counters before the next service is counted (control
// phase); network_compactness after the provision = (totals - what this step's releases added) over the occupied-slot sum at
// provision time, the difference to its value before the provision; the two link averages over topology.edges() in numpy's
// pairwise order, on the values AFTER the provision and BEFORE the step's releases — a lane of the row phase that applies a
// release leaves the link's values from before its update in an LDS stash.
#ifndef ORL_AGENT_WAVES
#define ORL_AGENT_WAVES 4  // waves per SIMD the register allocator leaves room for (cfg2 65 536 envs: 3 -> 114 us, 4 -> 108 us, 5 -> 141 us, 8 -> 193 us per launch)
#endif
// SPEC: the configuration's sizes as compile-time constants (the instantiation a specialisation library carries beside k_persist)
// FUSED (round 5): the action comes from one of the library's heuristics — the slot scan runs as this kernel's first phase, on
// the same slot maps, exactly as k_persist does (policy_g; the action is then trusted: is_path_free holds by construction and
// DeepRMSA's block walk is skipped) — instead of from a launch of k_policy in front of this one: one launch per agent step
// instead of two (cfg2 65 536 envs: 109 us -> ~85 us), actions written to P.actions as the stand-alone scan would have.
template <int ENV, int W, int SPEC = 0, bool FUSED = false>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(ORL_AGENT_WAVES, ORL_AGENT_WAVES))) k_agent(DevParams P, int auto_reset, int pol) {
  persist_spec_apply<SPEC>(P);
  constexpr bool CP = ENV != ENV_RMCSA;                                  // RMCSA: sink entries with a core per mask, the general row loop
  constexpr bool LINK_INFO = (ENV == ENV_RMSA || ENV == ENV_DEEPRMSA);   // info carries network compactness and the two link averages
  const PersistLds L = persist_lds_layout(P.E, P.H, P.bm_words, P.C, 0, CP, 0, true);
  typename sp::SinkEntryOf<CP>::type* s_tab = (typename sp::SinkEntryOf<CP>::type*)(orl_lds_raw + L.tab);
  u32* s_tally = (u32*)(orl_lds_raw + L.tally);
  unsigned short* s_mtab = (unsigned short*)(orl_lds_raw + L.mtab);
  unsigned short* s_list = (unsigned short*)(orl_lds_raw + L.list);
  int* s_deferred = (int*)(orl_lds_raw + L.misc);
  u32* s_list_n = (u32*)(orl_lds_raw + L.misc + 8);
  double* s_stash = (double*)(orl_lds_raw + L.total);  // [8][E][2] (LINK_INFO)
  const int lane = lane_id(), gl = lane & 7, el = lane >> 3;
  const i64 env0 = (i64)blockIdx.x * 8, env = env0 + el;
  const bool valid = env < P.B;
  const int nenv = (int)(P.B - env0 < 8 ? P.B - env0 : 8);
  sp::Prof prof;
#ifdef ORL_TIMING  // (tools/agent_timeline.py: the wavefront's clocks at entry and end, its cycles in the rebuild scan and in the release loop)
  const unsigned long long ag_t0 = wall_clock64();
  ORL_PROF_BEGIN_();
#endif
  sp::Wmem M = sp::wmem_global(P);
  M.clk = (double*)(orl_lds_raw + L.clk);
  M.clk_env0 = env0;
  // the per-core sums of the 8 envs in LDS for the launch (round 5): the control phase reads them three times — compactness before
  // the provision, the previous step's pending update, the end of the step — and in global memory every read was a RETURNING L2
  // atomic behind the row phase's L2 atomics: three global round trips of the step's dependent chain
  M.cs0 = (int*)(orl_lds_raw + L.cs);
  M.cenv0 = env0;
  M.cs_lds = true;
  M.cs_stride = L.csw;
  {
    const int q = L.csw / 4;
    for (int i = lane; i < nenv * q; i += 64)
      ((orl_i32x4*)M.cs0)[i] = ((const orl_i32x4*)(P.core_sums + (env0 + i / q) * P.cs_words))[i % q];
  }
  if (threadIdx.x == 0) s_deferred[0] = 0;
  // (this launch changes slot maps outside the persistent kernel: the row caches that kernel left with the state — keyed by a host
  // counter the launch of a captured graph does not advance — no longer describe them)
#ifndef ORL_X_NO_STAMP
  if (threadIdx.x == 0 && P.row_cache_stamp) P.row_cache_stamp[blockIdx.x] = 0;
#endif
  wave_fence();
  sp::CtrlOpts O;
  O.persistent = true; O.write_io = true; O.trusted = false; O.emit_queue = false; O.prefetch = true; O.auto_reset = auto_reset != 0;
  O.rank_pairs = false;
  int done_i = 0;
  sp::InfoCarry ic;
  ic.prev_comp = 1.0; ic.s_nh_prov = 0;
  // (the env's soon list requested with its record and kept in registers through the control phase, as the 3-wave forms of
  // k_persist do — two dependent memory round trips less, but at 128 VGPRs it costs more than it saves: cfg2 86 -> 101 us
  // per launch, cfg1 the same; DeepRMSA / RMCSA spill with it.  Off.)
#ifndef ORL_AGENT_SOONR
#define ORL_AGENT_SOONR 0
#endif
  constexpr bool SOONR = ORL_AGENT_SOONR != 0;
  sp::SoonRegs soon_c;
  soon_c.dirty = 0;
  if constexpr (SOONR) {
#pragma unroll
    for (int k = 0; k < ORL_SOON_PER_LANE; k++) {
      soon_c.t[k] = valid ? P.soon_t[env * ORL_SOON + gl + 8 * k] : __builtin_inf();
      soon_c.i[k] = valid ? (int)P.soon_i[env * ORL_SOON + gl + 8 * k] : 0;
    }
  }
  int4 av = make_int4(0, 0, 0, 0);
  if constexpr (FUSED) {
    const u64 desc = valid ? P.svc_desc[env] : 0ull;
    int a[4];
    policy_g<ENV, W, 8>(P, P.bitmap + (valid ? env : 0) * P.bm_words, valid, (int)(u32)desc, (int)((desc >> 32) & 0xffffu),
                        (int)((desc >> 48) & 0xffu), lane, pol, (pol == POL_PATH_FF && valid) ? P.path_col[env] : 0, a);
    av = make_int4(a[0], a[1], a[2], a[3]);
    O.trusted = true;
  }
  sp::ctrl_a<ENV, W, CP>(P, M, O, env, valid, lane, prof, FUSED ? &av : nullptr, s_tally, s_tab, 0, s_deferred, &done_i, s_list, s_list_n, L.tw,
                         SOONR ? &soon_c : nullptr, s_mtab, &ic);
  if constexpr (SOONR) {
    if (valid) {
#pragma unroll
      for (int k = 0; k < ORL_SOON_PER_LANE; k++) {
        P.soon_t[env * ORL_SOON + gl + 8 * k] = soon_c.t[k];
        P.soon_i[env * ORL_SOON + gl + 8 * k] = (u32)soon_c.i[k];
      }
    }
  }
  __syncthreads();  // sink table + item list, clocks
  {
    const int n_items = (int)*s_list_n;
    for (int idx = lane; idx < n_items; idx += 64) {
      const int code = (int)s_list[idx];
      const int iel = (code >> 8) & 7, link = code & 0xff, second = code >> 15;
      if constexpr (!CP) {
        if (!second) sp::row_item_lane<ENV, W>(P, M, sp::item_from_sink(env0 + iel, link, s_tab[P.E * iel + link]), prof);
      } else {
        sp::row_item_lane1<ENV, W>(P, M, env0 + iel, link, s_tab[P.E * iel + link].bits, s_mtab + ORL_MTAB * iel, second, prof, true,
                                   (LINK_INFO && P.info_mode == 0) ? s_stash + 2 * P.E * iel : nullptr);
      }
    }
  }
  __syncthreads();
  const bool deferred = s_deferred[0] != 0;
  u64* rec = P.scal + env * ORL_SCAL_WORDS;
  // the sums right after the provision = totals minus what the step's releases added; then the release part is cleared for the
  // next step and the sums go back to global memory (before the rare in-place releases below, which work there)
  int occ_p = 0, fb_p = 0;
  u64 acc_p = 0ull;
  if (ENV != ENV_RWA) {
    if (valid) {
      int* cs = sp::wm_cs(P, M, env);
      int* rs = cs + 2 * P.C;
      acc_p = rec[SC_ACC];
      const int c0 = (int)((acc_p >> 32) & 31);
      occ_p = cs[2 * c0] - rs[2 * c0];
      fb_p = cs[2 * c0 + 1] - rs[2 * c0 + 1];
      wave_fence();
      for (int i = gl; i < 2 * P.C; i += 8) rs[i] = 0;
    }
    wave_fence();
    const int q = L.csw / 4;
    for (int i = lane; i < nenv * q; i += 64)
      ((int4*)(P.core_sums + (env0 + i / q) * P.cs_words))[i % q] = ((const int4*)M.cs0)[i];
  }
  double mean_comp = 0.0, mean_util = 0.0;
  if constexpr (LINK_INFO) {
    if (valid && P.info_mode == 0) {
      // np.mean over the links in topology.edges() order (numpy pairwise sum, optical_rl_gym_amd/csrc/orl_device.h link_mean):
      // lane j of the group accumulates x[j], x[8 + j], ...; ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)); the tail one by one.
      // Every round's link indices are requested together, then the values of all of them (the loop that fetched index and
      // value entry by entry cost NSFNET's 22 links sixteen dependent memory round trips at the end of every wavefront).
      const int E = P.E;
      const double* ls = P.lstat + env * 4 * E;
      const double* st = s_stash + 2 * E * el;
      auto value = [&](int link, double& u, double& c) {
        bool rel = false;  // a release of this step touched the link: the values from before it
        if constexpr (CP) rel = (s_tab[E * el + link].bits >> 1) != 0u;
        const double gu = ls[4 * link], gc = ls[4 * link + 2], lu = st[2 * link], lc = st[2 * link + 1];
        u = rel ? lu : gu;
        c = rel ? lc : gc;
      };
      const int nfull = (E < 8) ? 0 : E - (E % 8), ntail = E - nfull;
      const int tl = (gl < ntail) ? P.edge_iter_order[nfull + gl] : 0;  // lane t holds tail entry t
      double su = 0.0, sc = 0.0, tu = 0.0, tc = 0.0;
      for (int b = 0; b < nfull; b += 32) {
        int lk[4];
        double uu[4], cc[4];
#pragma unroll
        for (int k = 0; k < 4; k++) lk[k] = (b + 8 * k < nfull) ? P.edge_iter_order[b + 8 * k + gl] : 0;
        if (b == 0 && gl < ntail) value(tl, tu, tc);
#pragma unroll
        for (int k = 0; k < 4; k++) { uu[k] = 0.0; cc[k] = 0.0; if (b + 8 * k < nfull) value(lk[k], uu[k], cc[k]); }
#pragma unroll
        for (int k = 0; k < 4; k++) {
          if (b + 8 * k < nfull) {
            if (b == 0 && k == 0) { su = uu[0]; sc = cc[0]; } else { su += uu[k]; sc += cc[k]; }
          }
        }
      }
      if (nfull == 0 && gl < ntail) value(tl, tu, tc);
      if (nfull > 0) {
        su += dpp_d<ORL_DPP_XOR1>(su); sc += dpp_d<ORL_DPP_XOR1>(sc);
        su += dpp_d<ORL_DPP_XOR2>(su); sc += dpp_d<ORL_DPP_XOR2>(sc);
        su += dpp_d<ORL_DPP_HALF_MIRROR>(su); sc += dpp_d<ORL_DPP_HALF_MIRROR>(sc);
      }
#pragma unroll
      for (int t = 0; t < 7; t++)
        if (t < ntail) { su += g8::gget(tu, t, lane); sc += g8::gget(tc, t, lane); }
      mean_util = su / (double)E;
      mean_comp = sc / (double)E;
    }
  }
  if (deferred) {  // (a few env-steps in 10^7) releases that did not fit the item form: in place, now
    __threadfence();
    __syncthreads();
    if (valid) {
      sp::rel_serial<ENV, W>(P, env, lane);
      if (ENV != ENV_RWA) {  // (what they added to the release part of the sums is not the next step's business either)
        int* rs_g = P.core_sums + env * P.cs_words + 2 * P.C;
        for (int i = gl; i < 2 * P.C; i += 8) rs_g[i] = 0;
      }
    }
    __threadfence();
    __syncthreads();
  }
  if (valid) {
    if constexpr (ENV != ENV_RWA) {
      // network compactness right after the provision: the totals minus what the step's releases added (row phase: L2
      // atomics), over the occupied-slot sum at provision time; the pending average of _update_network_stats
      // (rmsa_env.py:439-462) is finished with it, as k_finish2 does at the end of a device-resident run
      const u64 acc = acc_p;
      const int occ = occ_p, fb = fb_p;
      const double cur = (fb > 0) ? ((double)occ / (double)ic.s_nh_prov) * ((double)P.E / (double)fb) : 1.0;
      if (gl == 0 && ((u32)acc & 2u)) {
        const double a0 = __longlong_as_double((i64)rec[SC_GC_A]), td = __longlong_as_double((i64)rec[SC_GC_TD]);
        const double now_a = __longlong_as_double((i64)rec[SC_NOWA]);
        rec[SC_GCOMP] = (u64)__double_as_longlong((a0 + (cur * td)) / now_a);
        rec[SC_ACC] = (deferred ? rec[SC_ACC] : acc) & ~2ull;  // (the in-place releases cleared their own flag in the word)
      }
      if (LINK_INFO && gl == 0 && P.info_mode == 0) {
        double* io = P.info + env * P.n_info;
        io[4] = cur;
        io[5] = ic.prev_comp - cur;
        io[6] = mean_comp;
        io[7] = mean_util;
      }
    }
    if (ENV == ENV_DEEPRMSA && P.obs_dim) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      obs8_env<W>(P, P.bitmap + env * P.bm_words, rec, env, lane, done_i);
    }
  }
#ifdef ORL_TIMING
  if (threadIdx.x == 0 && blockIdx.x < 16384) {
    __builtin_amdgcn_s_waitcnt(0);
    g_wts[blockIdx.x * 8 + 0] = ag_t0; g_wts[blockIdx.x * 8 + 1] = wall_clock64(); g_wts[blockIdx.x * 8 + 2] = prof.acc[16 + 5];
    g_wts[blockIdx.x * 8 + 3] = prof.acc[16 + 6] + prof.acc[16 + 7];
    g_wts[blockIdx.x * 8 + 4] = (unsigned long long)__builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));
  }
#endif
}

// serial tail, one small workgroup per launch: the envs whose releases of this step did not fit the item form (about
// one env-step in 10^7; the control phase appends them to q_def) release them in place, 8 lanes per env.  A launch of its
// own because inlined into the row phase this code cost it half its occupancy.  DeepRMSA: the observation the persistent
// kernel wrote for such an env predates these releases and is written again.
template <int ENV, int W>
__global__ void __launch_bounds__(256) k_rel_tail(DevParams P, int buffer) {
  u32* dq = P.q_def + (size_t)buffer * P.q_def_stride;
  const u32 nd = dq[0];
  for (u32 d = threadIdx.x >> 3; d < nd; d += 32u) {
    const i64 env = (i64)dq[16 + d];
    sp::rel_serial<ENV, W>(P, env, lane_id());
    if (ENV == ENV_DEEPRMSA && P.obs_dim) {
      __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");
      obs8_env<W>(P, P.bitmap + env * P.bm_words, P.scal + env * ORL_SCAL_WORDS, env, lane_id(), P.done[env]);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) dq[0] = 0u;  // a following launch of any form starts from an empty list
}

template <int ENV, int W>
__global__ void __launch_bounds__(64) k_obs(DevParams P, int with_terminal) {
  const i64 env = blockIdx.x;
  const int lane = lane_id();
  Env e;
  env_load(P, e, env, lane);
  stage_in(P, e, (u64*)orl_lds_raw, lane);
  // after a step: an env that just finished its episode also gets the observation as `terminal_observation` (the soft
  // reset keeps the pending service, so the values are the same; SB3 VecEnv convention)
  if (ENV == ENV_DEEPRMSA)
    deep_observation<W>(P, e, lane, P.obs + env * P.obs_dim, (with_terminal && P.done[env]) ? P.term_obs + env * P.obs_dim : nullptr);
}

// DeepRMSAEnv.observation (deeprmsa_env.py:60-121) with 8 lanes per env, lane = path (k <= 8), 8 envs per wavefront: rows
// are read straight from global memory (the form of the slot scan), every lane writes its own path block.  The
// one-wavefront-per-env k_obs above staged the whole slot map in LDS and took 26.7 us per 32 768-env launch (cfg3).
// bm: the env's slot map (global or LDS); terminal: also write the observation as `terminal_observation` (the env just
// finished its episode: the soft reset keeps the pending service, so the values are the same; SB3 VecEnv convention)
// (sd, br: the record's SC_SRC_DST and SC_BR_IDX words)
template <int W>
__device__ __forceinline__ void obs8_env_w(const DevParams& P, const u64* bm, u64 sd, u64 br, i64 env, int lane, int terminal) {
  const int gl = lane & 7;
  const int src = (int)(u32)sd, dst = (int)(sd >> 32);
  const int bit_rate = (int)(u32)br, br_idx = (int)(br >> 32);
  const int N = P.N, J = P.J, S = P.S, WD = 2 * J + 3;
  double* o = P.obs + env * P.obs_dim;
  double* o2 = terminal ? P.term_obs + env * P.obs_dim : nullptr;
  const int mn = src < dst ? src : dst, mx = src < dst ? dst : src;
  for (int i = gl; i < 1 + 2 * N; i += 8) {
    const double v = (i == 0) ? (double)bit_rate / 100 : ((i == 1 + mn || i == 1 + N + mx) ? 1.0 : 0.0);
    o[i] = v;
    if (o2) o2[i] = v;
  }
  if (gl < P.K) {
    // this lane's path block: -1.0 everywhere first, then the values that exist (stores of one lane to one address keep their
    // order) — a register array of 2 j + 3 <= 19 doubles, written at the end, was 38 live VGPRs in the middle of the
    // persistent kernel's loop
    double* sp = o + 1 + 2 * N + gl * WD;
    double* sp2 = o2 ? o2 + 1 + 2 * N + gl * WD : nullptr;
    for (int i = 0; i < WD; i++) { sp[i] = -1.0; if (sp2) sp2[i] = -1.0; }
    if (gl < P.n_paths[src * N + dst]) {
      const int pidx = (src * N + dst) * P.K + gl;
      const Row<W> m = path_and_rec<W>(path_rec_load(P, pidx), bm, P.E, S, 0);
      const int n = P.nslots_path[(size_t)pidx * P.n_br + br_idx];
      Row<W> r = row_runs_ge<W>(m, n);
      const Row<W> zeros = row_andn<W>(row_mask_lo<W>(S), m);
      for (int b = 0; b < J && row_any<W>(r); b++) {
        const int st = row_ctz<W>(r);
        const Row<W> z = row_andn<W>(zeros, row_mask_lo<W>(st));
        const int end = row_any<W>(z) ? row_ctz<W>(z) : S;
        const double f0 = 2 * ((double)st - 0.5 * (double)S) / (double)S, f1 = (double)(end - st - 8) / 8;
        sp[2 * b] = f0; sp[2 * b + 1] = f1;
        if (sp2) { sp2[2 * b] = f0; sp2[2 * b + 1] = f1; }
        r = row_andn<W>(r, row_mask_lo<W>(end));
      }
      const double fn = ((double)n - 5.5) / 3.5;
      const int tot = row_popc<W>(m);
      const double ft = 2 * ((double)tot - 0.5 * (double)S) / (double)S;
      const int nruns = row_popc<W>(row_starts<W>(m));
      const double fr = (nruns > 0) ? ((double)tot / (double)nruns - 4) / 4 : -1.0;
      sp[2 * J] = fn; sp[2 * J + 1] = ft; sp[2 * J + 2] = fr;
      if (sp2) { sp2[2 * J] = fn; sp2[2 * J + 1] = ft; sp2[2 * J + 2] = fr; }
    }
  }
}
template <int W>
__device__ __forceinline__ void obs8_env(const DevParams& P, const u64* bm, const u64* rec, i64 env, int lane, int terminal) {
  obs8_env_w<W>(P, bm, rec[SC_SRC_DST], rec[SC_BR_IDX], env, lane, terminal);
}
template <int W>
__global__ void __launch_bounds__(256) k_obs8(DevParams P, int with_terminal) {
  const i64 env = (i64)blockIdx.x * 32 + (threadIdx.x >> 3);
  if (env >= P.B) return;
  obs8_env<W>(P, P.bitmap + env * P.bm_words, P.scal + env * ORL_SCAL_WORDS, env, lane_id(), with_terminal && P.done[env]);
}

// ---- QoSConstrainedRA for an agent in the loop: 8 lanes per env (round 5) --------------------------------------------------------
// QoSConstrainedRA.step (qos_constrained_ra.py:100-157) with the layout of k_agent — 8 envs per wavefront, all state in global
// memory — for batches of at least 20 480 envs (k_step, one wavefront per env, served every batch size until round 5; it stays for
// the smaller ones, where its 64 lanes find an env's due releases in one round trip, and is the other implementation in the
// parity tests).  The env has per-link spectrum counters instead of slot maps
// and no row statistics beyond a utilization average per link, so there is no row phase: the hops of a path are spread over the
// group's 8 lanes (g8::qos_path_free / qos_path_apply), releases are done in place (g8::release_due), the next service comes from
// the 8-lane generator helpers (g8::next_service draws the service class into the bit-rate fields).
#ifndef ORL_SPEC_ONLY
template <int W>  // (this file is compiled once per row width: the env's rows are one word — instantiated for W == 1 only)
__global__ void __launch_bounds__(64) k_agent_qos(DevParams P, int auto_reset) {
  const int lane = lane_id(), gl = lane & 7;
  const i64 env = (i64)blockIdx.x * 8 + (lane >> 3);
  if (env >= P.B) return;  // (whole groups: the helpers exchange values within a group only)
  g8::EnvG e;
  g8::env_load(P, e, env);
  g8::RngG rng;
  g8::rng_fill(e, rng, gl);
  e.t_soon = -__builtin_inf();  // (no soon list is kept for this family: ev_push must not try to maintain one)
  const int K = P.K, rej = P.allow_rejection ? 1 : 0;
  int a = P.actions[env * 4];
  const bool badq = a < 0 || a >= K + rej;  // actions_output[action] += 1 raises IndexError
  if (badq) { e.flags |= ORL_FLAG_BAD_ACTION; a = K; }
  const int clazz = e.bit_rate, np_ = P.n_paths[e.src * P.N + e.dst];
  bool accepted = false;
  if (!badq && ((clazz == 0 && a == 0) || (clazz != 0 && a < np_))) {
    const int pidx = (e.src * P.N + e.dst) * K + a;
    const PathRec prec = path_rec_load(P, pidx);
    if (g8::qos_path_free(P, e, lane, prec)) {
      g8::qos_path_apply(P, e, lane, prec, false);
      e.sa += 1;
      e.esa += 1;
      accepted = true;
      g8::ev_push(P, e, lane, e.at + e.ht, ev_pack(pidx, 0, 1, 0, 0));
    }
  }
  e.sp += 1;
  e.esp += 1;
  const double rew = accepted ? P.class_reward[clazz] : 0.0;
  if (gl == 0) {
    double* info = P.info + env * P.n_info;
    info[0] = (double)(e.sp - e.sa) / (double)e.sp;
    info[1] = (double)(e.esp - e.esa) / (double)e.esp;
  }
  e.new_service = 0;
  g8::next_service<ENV_QOS, 1>(P, e, lane, rng);
  g8::rng_commit_stores(e, rng, gl);
  g8::release_due<ENV_QOS, 1>(P, e, lane);  // (qos_constrained_ra.py:245-251: every service that left before the new arrival)
  const bool doneq = (e.esp == (i64)P.episode_length);
  if (P.ep_log && P.ep_rew && gl == 0) {
    const double acc = P.ep_rew_acc[env] + rew;
    if (doneq) {
      const int idx = P.ep_count[env];
      if (idx < P.ep_cap) P.ep_rew[env * P.ep_cap + idx] = acc;
    }
    P.ep_rew_acc[env] = doneq ? 0.0 : acc;
  }
  if (doneq && P.ep_log && gl == 0) episode_log(P, env, e.esa);
  if (doneq && auto_reset) { e.ebrq = 0; e.ebrp = 0; e.esp = 0; e.esa = 0; }  // (soft reset: this family counts at the decision)
  if (gl == 0) {
    P.reward[env] = rew;
    P.done[env] = doneq ? 1 : 0;
  }
  g8::env_store(P, e, gl);
}
#endif

#if defined(ORL_RS_PROF) && !defined(ORL_SPEC_ONLY)
#if ORL_W == 5
extern "C" int orl_debug_rs_prof(unsigned long long* out16, int reset) {
  std::vector<unsigned long long> h((size_t)ORL_RSP_WAVES * 16);
  if (hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_rs_prof), h.size() * 8) != hipSuccess) return -1;
  for (int k = 0; k < 16; k++) out16[k] = 0;
  for (size_t w = 0; w < ORL_RSP_WAVES; w++)
    for (int k = 0; k < 16; k++) out16[k] += h[w * 16 + k];
  if (reset) { std::fill(h.begin(), h.end(), 0ull); if (hipMemcpyToSymbol(HIP_SYMBOL(g_rs_prof), h.data(), h.size() * 8) != hipSuccess) return -1; }
  return 0;
}
#endif
#endif
#ifdef ORL_SPEC_ONLY
// ---- the whole of a specialisation library: one instantiation and its launch entry ----------------------------------------
extern "C" int orl_spec_struct_bytes(void) { return (int)sizeof(DevParams); }
extern "C" void orl_spec_describe(int* out /*[22]*/) {
  const PersistSpec& s = kPersistSpec;
  const int v[22] = {s.env, s.W, s.lds, s.waves, s.N, s.E, s.K, s.H, s.M, s.S, s.C, s.J, s.bit_rate_mode, s.br_lo, s.n_br, s.rand_n, s.rand_bits,
                     s.ev_cap, s.bm_words, s.cs_words, s.obs_dim, s.n_info};
  for (int i = 0; i < 22; i++) out[i] = v[i];
}
extern "C" void orl_spec_launch(const DevParams* VP, unsigned grid, size_t lds, hipStream_t st, int pol, int target, int* wg_step,
                                unsigned int* unfinished, unsigned int* clear_next) {
  if (lds > 48 * 1024)
    hipFuncSetAttribute((const void*)k_persist<ORL_SPEC_ENV, ORL_W, ORL_SPEC_LDS, ORL_SPEC_WAVES, 1, ORL_SPEC_RW != 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((k_persist<ORL_SPEC_ENV, ORL_W, ORL_SPEC_LDS, ORL_SPEC_WAVES, 1, ORL_SPEC_RW != 0>), dim3(grid), dim3(ORL_SPEC_RW ? 128 : 64), lds, st, *VP, pol,
                     target, wg_step, unfinished, clear_next);
}
// (diagnostic builds, -DORL_TIMING: the per-phase cycle sums of this library's kernel — tools/pair_prof.py)
extern "C" int orl_spec_prof(unsigned long long* out48, int reset) {
#ifdef ORL_TIMING
  if (reset == 2) return hipMemcpyFromSymbol(out48, HIP_SYMBOL(g_wts), 16384 * 8 * 8) == hipSuccess ? 0 : -1;
  if (reset == 3) return hipMemcpyFromSymbol(out48, HIP_SYMBOL(sp::g_prof), (size_t)ORL_PROF_WAVES * ORL_PROF_SLOTS * 8) == hipSuccess ? 0 : -1;
#endif
  for (int k = 0; k < ORL_PROF_SLOTS; k++) out48[k] = 0;
#ifdef ORL_TIMING
  std::vector<unsigned long long> h((size_t)ORL_PROF_WAVES * ORL_PROF_SLOTS);
  if (hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(sp::g_prof), h.size() * 8) != hipSuccess) return -1;
  for (size_t w = 0; w < ORL_PROF_WAVES; w++)
    for (int k = 0; k < ORL_PROF_SLOTS; k++) out48[k] += h[w * ORL_PROF_SLOTS + k];
  if (reset) {
    std::fill(h.begin(), h.end(), 0ull);
    if (hipMemcpyToSymbol(HIP_SYMBOL(sp::g_prof), h.data(), h.size() * 8) != hipSuccess) return -1;
  }
#else
  (void)reset;
#endif
  return 0;
}
extern "C" void orl_spec_agent_launch(const DevParams* VP, unsigned grid, size_t lds, hipStream_t st, int auto_reset, int pol) {
  if (pol >= 0) hipLaunchKernelGGL((k_agent<ORL_SPEC_ENV, ORL_W, 1, true>), dim3(grid), dim3(64), lds, st, *VP, auto_reset, pol);
  else hipLaunchKernelGGL((k_agent<ORL_SPEC_ENV, ORL_W, 1, false>), dim3(grid), dim3(64), lds, st, *VP, auto_reset, -1);
}
#else
// =============================================================================================
// launchers
// =============================================================================================
#define ORL_FOR_ENV(B_, MACRO)                              \
  switch ((B_)->P.env_type) {                               \
    case ENV_RMSA: { MACRO(ENV_RMSA) } break;               \
    case ENV_DEEPRMSA: { MACRO(ENV_DEEPRMSA) } break;       \
    case ENV_RWA: { MACRO(ENV_RWA) } break;                 \
    default: { MACRO(ENV_RMCSA) } break;                    \
  }

// the host-driven kernels also serve QoSConstrainedRA (one counter per link: row width 1 only)
#define ORL_FOR_ENV_ALL(B_, MACRO)                                                  \
  if ((B_)->P.env_type == ENV_QOS) { if constexpr (W == 1) { MACRO(ENV_QOS) } }     \
  else ORL_FOR_ENV(B_, MACRO)

namespace orl_launch {

template <int W> void reset(orl_batch* b, int full, const unsigned char* dmask) {
  const DevParams& VP = b->P;
  dim3 g((unsigned)VP.B), blk(64);
  size_t lds = VP.lds_bytes;
#define PER_ENV(E_) hipLaunchKernelGGL((k_reset<E_, W>), g, blk, lds, b->stream, VP, full, dmask);
  ORL_FOR_ENV_ALL(b, PER_ENV)
#undef PER_ENV
}

template <int W> void policy(orl_batch* b, int pol) {
  const DevParams& VP = b->P;
  // RMCSA scans (path, core) pairs: one env per wavefront.  The other families put 8 envs on a wavefront when k <= 8.
  const bool wide = (VP.env_type == ENV_RMCSA) || VP.K > 8;
  const i64 per_wg = ORL_POLICY_WAVES * (wide ? 1 : 8);
  dim3 g((unsigned)((VP.B + per_wg - 1) / per_wg)), blk(64 * ORL_POLICY_WAVES);
#define PER_ENV(E_)                                                                             \
  if (wide) hipLaunchKernelGGL((k_policy<E_, W, 64>), g, blk, 0, b->stream, VP, pol);           \
  else hipLaunchKernelGGL((k_policy<E_, W, 8>), g, blk, 0, b->stream, VP, pol);
  ORL_FOR_ENV_ALL(b, PER_ENV)
#undef PER_ENV
  ORL_TK(b, "k_policy");
}

template <int W> void step64(orl_batch* b, int auto_reset, int want_info, int fused_policy) {
  const DevParams& VP = b->P;
  dim3 g((unsigned)VP.B), blk(64);
  // stage the pending release times through LDS when the per-env window stays small enough for 5 waves/SIMD
  const size_t ev_bytes = (size_t)VP.ev_cap * 8;
  const bool evl = (VP.lds_bytes + ev_bytes) <= 8 * 1024;
  size_t lds = VP.lds_bytes + (evl ? ev_bytes : 0);
#define PER_ENV(E_)                                                                                                          \
  if (evl) hipLaunchKernelGGL((k_step<E_, W, true>), g, blk, lds, b->stream, VP, auto_reset, want_info, fused_policy);       \
  else hipLaunchKernelGGL((k_step<E_, W, false>), g, blk, lds, b->stream, VP, auto_reset, want_info, fused_policy);
  ORL_FOR_ENV_ALL(b, PER_ENV)
#undef PER_ENV
  ORL_TK(b, "k_step");
}

template <int W> void obs(orl_batch* b, int with_terminal) {
  const DevParams& VP = b->P;
  if (VP.K <= 8 && VP.J <= 8) {
    hipLaunchKernelGGL((k_obs8<W>), dim3((unsigned)((VP.B + 31) / 32)), dim3(256), 0, b->stream, VP, with_terminal);
  } else {
    hipLaunchKernelGGL((k_obs<ENV_DEEPRMSA, W>), dim3((unsigned)VP.B), dim3(64), (size_t)VP.lds_bytes, b->stream, VP, with_terminal);
  }
  ORL_TK(b, "k_obs");
}

// Forms of the persistent kernel: (what lives in LDS, waves per SIMD the registers are budgeted for).  The LDS window decides
// how many wavefronts a CU holds.  The hardware allocates LDS in pieces of 1 280 bytes (tools/probe/lds_resident.hip,
// measured on MI355X: 12 workgroups share a CU's 160 KiB up to 12 800 B each, 11 up to 14 080, 16 up to 10 240 —
// hipOccupancyMaxActiveBlocksPerMultiprocessor says 12 up to 13 648).
struct PersistForm { int lds, waves; };
// (round 6: 7 and 8 are the rows-deferred forms — state 4 / 5 = the window of state 3 / 1 without the row phase's tables)
static const PersistForm kPersistForms[] = {{0, 4}, {0, 3}, {2, 2}, {2, 3}, {1, 3}, {1, 4}, {3, 4}, {4, 4}, {5, 4}};
constexpr int kNumPersistForms = 9;
static inline bool persist_rd_state(int state) { return state == 4 || state == 5; }
static int lds_wgs_per_cu(size_t lds) {
  if (lds == 0) return 1 << 20;
  const size_t alloc = (lds + 1279) / 1280 * 1280;
  return (int)((size_t)(160 * 1024) / alloc);
}
struct PersistChoice { int form; size_t lds; int inner; int rw; int evl; };
static size_t persist_window(const DevParams& VP, int state, int inner) {
  const bool rd = persist_rd_state(state);
  const int base = state == 4 ? 3 : (state == 5 ? 1 : state);
  return (size_t)persist_lds_layout(VP.E, VP.H, VP.bm_words, VP.C, base, persist_compact(VP.env_type, base), rd ? 0 : inner,
                                    orl_persist_deferred(VP.env_type), rd).total;
}
// the rows-deferred forms: single-core families with the statistics deferred, a bit per link in a 64-bit event word, services of
// at most 63 slots in a 9-bit first slot (the compact sink's own limits), and an event log to write to
static bool persist_rd_possible(const DevParams& VP) {
  return orl_persist_deferred(VP.env_type) && VP.env_type != ENV_RMCSA && VP.E <= 64 && VP.S <= 512;
}
// `tuned`: the choice for a specialisation library (built without machine-level LICM and with the soon list in registers in the
// 4-wave forms, _build.py SPEC_TUNING) — for the flags such a library is built with, and at launch when one is attached
static PersistChoice persist_choose(const DevParams& VP, bool tuned = false) {
  // Measured on MI355X, env-steps/s: cfg2 65 536 envs: form 0 (global state, 4 waves) 8.3e8, form 4 (LDS
  // state, 3 waves) 1.02e9 at 11 wavefronts per CU with the inner-run cache, 1.05e9 at 12 without it — 69 MB of HBM traffic
  // and 0.85 M L2<->fabric requests per batched step against 206 MB / 2.63 M; cfg1 65 536: form 4 1.13e9, form 5 (LDS state,
  // 4 waves) 1.18e9; cfg3: 1.25e9 / 1.27e9.  A wavefront more per CU is worth 3-5 %: the 4-wave form is taken if its
  // window keeps 16 on a CU, the 3-wave form down to 10, and the inner-run cache (+2.5 %) only where it costs no wavefront.
  const bool can_inner = persist_inner(VP.env_type, ORL_W, 1);
  // the row caches (level 1: inner free runs, +2.5 %; level 2: + each row's occ / free-block contribution, +2 %) are taken at the
  // highest level that costs no wavefront per CU
  auto level = [&](int state, int cap) {
    if (!can_inner) return 0;
    const int r_none = lds_wgs_per_cu(persist_window(VP, state, 0));
    for (int lv = 2; lv >= 1; lv--) {
      const int r = lds_wgs_per_cu(persist_window(VP, state, lv));
      if ((r < cap ? r : cap) == (r_none < cap ? r_none : cap)) return lv;
    }
    return 0;
  };
  const size_t l0 = persist_window(VP, 1, 0), g0 = persist_window(VP, 3, 0);  // (g: records in global memory)
  const int r0 = lds_wgs_per_cu(l0);
  PersistChoice c;
  // (round 3, cfg2 with the 4-byte sink entries: form 4 with the cache 1.23e9; form 6 — 4 waves per SIMD, 16 per CU, but the
  // records in global memory, the soon list in memory and 9 spilled VGPRs — 1.14e9: what a wavefront keeps next to itself is
  // worth more than a fourth wavefront per SIMD.  Form 6 is taken only where the 3-wave window does not fit at all.)
  if (r0 >= 16) { c.form = 5; c.inner = level(1, 16); }
  // (round 4: a tuned instantiation of form 6 needs 128 VGPRs with the soon list in registers and no spills, and 16 wavefronts
  // per CU are 4 096 resident = exactly two generations of a 65 536-env batch: cfg2 20-step launches 1.135e9 -> 1.190e9, 300-step
  // runs 1.467e9 -> 1.474e9 against form 4)
  // ... for batches of more wavefronts than form 4 keeps resident (12 per CU x 256 CUs); below that no generation is cut short, and
  // the records in LDS are a dependent round trip per step less: 4 096 envs +1.7 %, 8 192 +2.1 %, 16 384 +4.1 % for form 4
  else if (tuned && VP.env_type != ENV_RMCSA && r0 >= 10 && lds_wgs_per_cu(g0) >= 16 && (VP.B + 7) / 8 > 12 * 256) { c.form = 6; c.inner = level(3, 16); }
  else if (r0 >= 10) { c.form = 4; c.inner = level(1, 12); }
  else if (lds_wgs_per_cu(g0) >= 16) { c.form = 6; c.inner = level(3, 16); }
  // (global state: the 4-wave form except for RMCSA — round 3, with the 4-byte sink entries: cfg5 Germany50 32 768 envs 5.6e8 at 4
  // waves per SIMD, 5.2e8 at 3; cfg4 RMCSA 5.0e8 / 5.3e8)
  else { c.form = (VP.env_type == ENV_RMCSA) ? 1 : 0; c.inner = 0; }
  // Small batches (round 5): at most 1 536 workgroups — 6 pairs per CU, all resident at 3 waves per SIMD — need the window to fit
  // at most six times — form 4
  // (everything in LDS, 3 waves per SIMD: soon list in registers) for every single-core configuration whose window fits a
  // workgroup's 64 KiB, in its two-wavefront form (below).  4 096 envs, form 4 as a pair against the form chosen above alone:
  // cfg2 +16 %, cfg3 +7 %, cfg1 +2 %, cfg5 (Germany50, global state above) +19 %.
  const i64 n_wg = (VP.B + 7) / 8;
  bool small_pair = false;
  if (tuned && VP.env_type != ENV_RMCSA && n_wg <= 1536) {
    const size_t w = persist_window(VP, 1, can_inner ? 2 : 0) + ORL_RW_EXTRA_BYTES;
    if (w <= 64 * 1024 && lds_wgs_per_cu(w) >= (int)((n_wg + 255) / 256)) { c.form = 4; small_pair = true; }
  }
  if (const char* e = getenv("ORL_PERSIST_VARIANT")) {  // A/B measurements and cross-checks
    const int f = atoi(e);
    bool built = f >= 0 && f < kNumPersistForms;
#ifndef ORL_ALT_IMPLS
    built = built && f != 2 && f != 3;
#endif
    if (f >= 7) built = built && persist_rd_possible(VP);
    if (built && persist_window(VP, kPersistForms[f].lds, 0) <= 64 * 1024 && f != c.form) {
      c.form = f;
      const int st = kPersistForms[f].lds;
      c.inner = (st >= 1 && st <= 3 && persist_inner(VP.env_type, ORL_W, st)) ? level(st, 4 * kPersistForms[f].waves) : 0;
    }
  }
  // RMCSA (24-byte sink entries, a core per mask, the general row loop) does not fit the 128-VGPR budget of the 4-wave forms — 25-32
  // spilled VGPRs, measured slower than its 3-wave forms wherever both fit — and is not built in them: routed to the 3-wave form
  // with the same state (global: 1; maps + records in LDS: 4, where that window fits a workgroup; else global)
  if (VP.env_type == ENV_RMCSA && kPersistForms[c.form].waves == 4) {
    const bool lds_ok = kPersistForms[c.form].lds != 0 && persist_window(VP, 1, 0) <= 64 * 1024 && lds_wgs_per_cu(persist_window(VP, 1, 0)) >= 4;
    c.form = lds_ok ? 4 : 1;
    c.inner = 0;
  }
  if (const char* e = getenv("ORL_PERSIST_INNER")) {  // A/B and cross-checks: 0 = no row caches, 1 = inner runs, 2 = + occ / free blocks
    const int v = atoi(e);
    c.inner = (v >= 0 && v <= 2 && !persist_rd_state(kPersistForms[c.form].lds) && persist_inner(VP.env_type, ORL_W, kPersistForms[c.form].lds)) ? v : 0;
  }
  // The two-wavefront form (k_persist<..., RW>, specialisation libraries only): batches whose pairs are all resident at once
  // (measured: +20 % at 10 240 and 12 288 envs of cfg2, -20 % at 14 336, where a second generation starts).  LDS is no constraint
  // there: both row caches, and ORL_RW_EXTRA_BYTES for the pair's counters and the staged batch of services.  ORL_PERSIST_RW=0/1: A/B measurements and cross-checks at
  // any batch size.
  c.rw = 0;
  if (tuned && VP.env_type != ENV_RMCSA && kPersistForms[c.form].lds == 1) {
    c.rw = small_pair ? 1 : 0;
    if (const char* e = getenv("ORL_PERSIST_RW")) c.rw = atoi(e) != 0 ? 1 : 0;
  }
  if (c.rw && !getenv("ORL_PERSIST_INNER")) c.inner = can_inner ? 2 : 0;
  c.lds = persist_window(VP, kPersistForms[c.form].lds, c.inner) + (c.rw ? ORL_RW_EXTRA_BYTES : 0);
  // ... and, where it still fits a workgroup's 64 KiB and the batch's workgroups a CU, the 8 envs' pending release times (cfg2:
  // 36 KiB: two workgroups per CU, batches of at most 4 096 envs)
  c.evl = 0;
  if (c.rw) {
    const size_t w = c.lds + (size_t)8 * VP.ev_cap * 8;
    if (w <= 64 * 1024 && lds_wgs_per_cu(w) >= (int)(((VP.B + 7) / 8 + 255) / 256)) c.evl = 1;
    if (const char* e = getenv("ORL_PERSIST_EVL")) c.evl = (atoi(e) != 0 && w <= 64 * 1024) ? 1 : 0;
    if (c.evl) c.lds = w;
  }
  return c;
}
static int persist_variant(const DevParams& VP, size_t* lds_bytes, bool tuned = false) {
  const PersistChoice c = persist_choose(VP, tuned);
  *lds_bytes = c.lds;
  return c.form;
}
// the form the launcher takes for this configuration: what is in the LDS window, waves per SIMD (the key of a specialisation)
template <int W> void persist_form(const DevParams& VP, int* lds_state, int* waves) {
  const PersistChoice c = persist_choose(VP, true);
  *lds_state = kPersistForms[c.form].lds;
  *waves = kPersistForms[c.form].waves + 16 * c.rw;  // (bit 4: the two-wavefront form)
}
template <int W> int persist_uses_lds(orl_batch* b) {
  size_t lds;
  return kPersistForms[persist_variant(b->P, &lds, b->spec_launch != nullptr)].lds;
}
// Workgroups per CU the form allows (LDS window, register budget).  ORL_PERSIST_WGS_PER_CU=r lowers the residency by padding
// the LDS request (experiments).
static int persist_max_per_cu(int v, size_t lds) {
  int per_cu = 4 * kPersistForms[v].waves;
  if (lds_wgs_per_cu(lds) < per_cu) per_cu = lds_wgs_per_cu(lds);
  return per_cu < 1 ? 1 : per_cu;
}
static size_t persist_tuned_lds(int v, size_t lds) {
  const int rmax = persist_max_per_cu(v, lds);
  int want_r = rmax;
  if (const char* e = getenv("ORL_PERSIST_WGS_PER_CU")) { int f = atoi(e); if (f >= 1 && f <= rmax) want_r = f; }
  if (want_r == rmax) return lds;
  const size_t want = ((size_t)(160 * 1024) / (size_t)want_r) / 1280 * 1280;  // the largest window that still fits want_r times
  return want > lds ? want : lds;
}
// forms 2 and 3 (link statistics and sums in LDS too) measured slower everywhere: they are built only into the
// -DORL_ALT_IMPLS library, as one more independent form for the cross-implementation tests
#ifdef ORL_ALT_IMPLS
#define ORL_FULL_LDS_CASES(E_) case 2: LAUNCH(E_, 2, 2); break; case 3: LAUNCH(E_, 2, 3); break;
#else
#define ORL_FULL_LDS_CASES(E_)
#endif
template <int W> void persist(orl_batch* b, const DevParams& VP0, hipStream_t st, int pol, int target, int* wg_step, unsigned int* unfinished,
                              unsigned int* clear_next, int finish) {
  DevParams VP = VP0;
  VP.persist_finish = finish;
  dim3 gc((unsigned)((VP.B + 7) / 8)), blk(64);
  bool use_spec = b->spec_launch != nullptr;
  if (const char* e = getenv("ORL_PERSIST_SPEC")) { if (atoi(e) == 0) use_spec = false; }
  // the form is chosen for the WHOLE batch (its wavefront count decides between the 3- and the 4-wave form, and a specialisation
  // library is built for that choice): a run in two halves launches the same kernel on both views
  DevParams VC = VP;
  VC.B = b->P.B;
  const PersistChoice ch = persist_choose(VC, use_spec);
  const int v = ch.form;
  VP.persist_ic = ch.inner;
  VP.persist_evl = ch.evl;
  VP.persist_fair = 11;  // (20 us per priority level: about one step)
  if (const char* e = getenv("ORL_PERSIST_FAIR")) { const int f = atoi(e); if (f >= 0 && f <= 30) VP.persist_fair = f; }  // A/B: 0 = the arbiter's oldest-first
  VP.row_cache_key = VP.row_cache ? ((b->cache_epoch << 8) | (v << 4) | ch.inner) : 0;
  if (const char* e = getenv("ORL_ROW_CACHE_KEEP")) { if (atoi(e) == 0) VP.row_cache_key = 0; }  // A/B: rebuild at every launch
  size_t lds_a = persist_tuned_lds(v, ch.lds);
#define LAUNCH(E_, LDS_, WV_)                                                                                                 \
  do {                                                                                                                       \
    if (lds_a > 48 * 1024) hipFuncSetAttribute((const void*)k_persist<E_, W, LDS_, WV_>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a); \
    hipLaunchKernelGGL((k_persist<E_, W, LDS_, WV_>), gc, blk, lds_a, st, VP, pol, target, wg_step, unfinished, clear_next); \
  } while (0)
  // an instantiation built for this very configuration (orl_batch_load_spec)?  (ORL_PERSIST_SPEC=0: the generic kernel)
  bool spec = b->spec_launch != nullptr && b->spec_lds == kPersistForms[v].lds && b->spec_waves == kPersistForms[v].waves + 16 * ch.rw;
  if (const char* e = getenv("ORL_PERSIST_SPEC")) { if (atoi(e) == 0) spec = false; }
  b->persist_spec = spec ? (ch.rw ? 2 : 1) : 0;  // (debug query: 2 = the two-wavefront form)
  b->persist_form_last = v;
  if (spec) {
    b->spec_launch(&VP, gc.x, lds_a, st, pol, target, wg_step, unfinished, clear_next);
  } else {
#define PER_ENV(E_)                                                                                                          \
  switch (v) {                                                                                                               \
    case 0: if constexpr (E_ != ENV_RMCSA) LAUNCH(E_, 0, 4); break;                                                          \
    case 1: LAUNCH(E_, 0, 3); break;                                                                                         \
    ORL_FULL_LDS_CASES(E_)                                                                                                   \
    case 4: LAUNCH(E_, 1, 3); break;                                                                                         \
    case 6: if constexpr (E_ != ENV_RMCSA) LAUNCH(E_, 3, 4); break;                                                          \
    case 7: if constexpr (E_ != ENV_RMCSA) LAUNCH(E_, 4, 4); break;                                                          \
    case 8: if constexpr (E_ != ENV_RMCSA) LAUNCH(E_, 5, 4); break;                                                          \
    default: if constexpr (E_ != ENV_RMCSA) LAUNCH(E_, 1, 4); break;                                                         \
  }
    ORL_FOR_ENV(b, PER_ENV)
#undef PER_ENV
  }
#undef LAUNCH
  // rows-deferred forms: the link statistics and the compactness sums of the launch's events, one lane per link row, behind the
  // launch on its stream and in front of k_stats (which reads the sums it puts into the statistics log)
  if (persist_rd_state(kPersistForms[v].lds)) {
    // (this form writes slot maps without keeping the other forms' row caches: a stamp they left must not match again)
    if (++b->cache_epoch >= (1 << 22)) {
      if (b->P.row_cache_stamp) hipMemsetAsync(b->P.row_cache_stamp, 0, (size_t)((b->P.B + 7) / 8) * sizeof(int), st);
      b->cache_epoch = 1;
    }
    int G = ORL_ROWSTATS_THREADS / VP.E;
    G = G > ORL_RS_GMAX ? ORL_RS_GMAX : G;
    const size_t lds_r = (size_t)rowstats_lds_layout(G).total;
    dim3 gr((unsigned)((VP.B + G - 1) / G)), br(ORL_ROWSTATS_THREADS);
#define ROWSTATS(E_) hipLaunchKernelGGL((k_rowstats<E_, W>), gr, br, lds_r, st, VP, G)
    switch (VP.env_type) {
      case ENV_RMSA: ROWSTATS(ENV_RMSA); break;
      case ENV_DEEPRMSA: ROWSTATS(ENV_DEEPRMSA); break;
      default: ROWSTATS(ENV_RWA); break;
    }
#undef ROWSTATS
  }
  // the bookkeeping of the steps this launch ran, one lane per env (deferred statistics: ctrl_d logged it), behind the launch
  // on its stream; the forms that keep it in the loop logged nothing
  if (orl_persist_deferred(VP.env_type) && VP.slog) {
    dim3 gs((unsigned)((VP.B + ORL_STATS_LANES - 1) / ORL_STATS_LANES));
    const bool rd = persist_rd_state(kPersistForms[v].lds);
    // (discrete bit rates: the per-rate counts of a launch in LDS, 2 n_br counters per lane)
    const int br_lds = (VP.bit_rate_mode == 1 && VP.br_hist && (size_t)ORL_STATS_LANES * 2 * VP.n_br * 4 <= 48 * 1024) ? 1 : 0;
    const size_t brb = br_lds ? (size_t)ORL_STATS_LANES * 2 * VP.n_br * 4 : 0;
    switch (VP.env_type) {
      case ENV_RMSA:
        if (rd) hipLaunchKernelGGL((k_stats<ENV_RMSA, true>), gs, blk, brb, st, VP, br_lds);
        else hipLaunchKernelGGL((k_stats<ENV_RMSA>), gs, blk, brb, st, VP, br_lds);
        break;
      case ENV_DEEPRMSA:
        if (rd) hipLaunchKernelGGL((k_stats<ENV_DEEPRMSA, true>), gs, blk, brb, st, VP, br_lds);
        else hipLaunchKernelGGL((k_stats<ENV_DEEPRMSA>), gs, blk, brb, st, VP, br_lds);
        break;
      case ENV_RMCSA: hipLaunchKernelGGL((k_stats<ENV_RMCSA>), gs, blk, brb, st, VP, br_lds); break;
      default: {
        // (RWA: the action marginals of the launch counted in LDS while the row of counters per lane fits 48 KiB)
        const size_t hb = (size_t)ORL_STATS_LANES * (size_t)((VP.K + 1) + (VP.S + 1)) * 4;
        const int hist_lds = (hb <= 48 * 1024 && VP.act_hist) ? 1 : 0;
        hipLaunchKernelGGL((k_stats<ENV_RWA>), gs, blk, hist_lds ? hb : 0, st, VP, hist_lds);
      } break;
    }
  }
}
// one host- or agent-driven step through the phases of the persistent kernel
// pol >= 0: the heuristic's slot scan as the kernel's first phase (k_agent<..., FUSED>); -1: the actions in P.actions
template <int W> void agent_step(orl_batch* b, int auto_reset, int pol) {
  const DevParams& VP = b->P;
  dim3 g((unsigned)((VP.B + 7) / 8)), blk(64);
  if (VP.env_type == ENV_QOS) {  // (no slot maps, no row phase: a kernel of its own; a heuristic's scan is a launch in front of it)
    if constexpr (W == 1) {
      if (pol >= 0) policy<W>(b, pol);
      hipLaunchKernelGGL((k_agent_qos<W>), g, blk, 0, b->stream, VP, auto_reset);
      ORL_TK(b, "k_agent_qos");
    }
    return;
  }
  const size_t lds = (size_t)persist_lds_layout(VP.E, VP.H, VP.bm_words, VP.C, 0, VP.env_type != ENV_RMCSA, 0, true).total + (size_t)8 * VP.E * 16;
  // the instantiation built for this configuration, when a specialisation library is attached (ORL_PERSIST_SPEC=0: generic)
  bool spec = b->spec_agent_launch != nullptr;
  if (const char* e = getenv("ORL_PERSIST_SPEC")) { if (atoi(e) == 0) spec = false; }
  if (spec) {
    b->spec_agent_launch(&VP, g.x, lds, b->stream, auto_reset, pol);
    ORL_TK(b, "k_agent");
    return;
  }
#define PER_ENV(E_)                                                                                                 \
  if (pol >= 0) hipLaunchKernelGGL((k_agent<E_, W, 0, true>), g, blk, lds, b->stream, VP, auto_reset, pol);         \
  else hipLaunchKernelGGL((k_agent<E_, W, 0, false>), g, blk, lds, b->stream, VP, auto_reset, -1);
  ORL_FOR_ENV(b, PER_ENV)
#undef PER_ENV
  ORL_TK(b, "k_agent");
}

// wavefronts of the persistent kernel a GPU of `n_cu` CUs holds at once for this batch (LDS window and register budget)
template <int W> int persist_resident(orl_batch* b, int n_cu) {
  size_t lds = 0;
  const int v = persist_variant(b->P, &lds, b->spec_launch != nullptr);
  return persist_max_per_cu(v, lds) * n_cu;
}

template <int W> void step2(orl_batch* b, int pol) {
#ifdef ORL_ALT_IMPLS
  const DevParams& VP = b->P;
  const bool wide = VP.K > 8;
  if (wide) policy<W>(b, pol);  // k > 8: the one-env-per-wavefront slot scan stays a launch of its own
  dim3 gc((unsigned)((VP.B + 31) / 32)), blk(256), blk_r(ORL_ROWS2_THREADS);
  const size_t lds_a = (size_t)32 * VP.E * sizeof(sp::SinkEntry);
  int& par = b->parity;
#define PER_ENV(E_)                                                                                                          \
  if (wide) {                                                                                                                \
    if (lds_a > 48 * 1024) hipFuncSetAttribute((const void*)k_step_a2<E_, W, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a); \
    hipLaunchKernelGGL((k_step_a2<E_, W, false>), gc, blk, lds_a, b->stream, VP, pol, par);                                  \
  } else {                                                                                                                   \
    if (lds_a > 48 * 1024) hipFuncSetAttribute((const void*)k_step_a2<E_, W, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a); \
    hipLaunchKernelGGL((k_step_a2<E_, W, true>), gc, blk, lds_a, b->stream, VP, pol, par);                                   \
  }                                                                                                                          \
  ORL_TK(b, "k_step_a2");                                                                                                    \
  hipLaunchKernelGGL((k_rows2<E_, W>), gc, blk_r, 0, b->stream, VP, par);                                                    \
  ORL_TK(b, "k_rows2");                                                                                                      \
  hipLaunchKernelGGL((k_rel_tail<E_, W>), dim3(1), blk, 0, b->stream, VP, par);                                              \
  ORL_TK(b, "k_rel_tail");
  ORL_FOR_ENV(b, PER_ENV)
#undef PER_ENV
  par ^= 1;
  if (VP.obs_dim) obs<W>(b, 1);
#else
  (void)b; (void)pol;
#endif
}

// diagnostic builds (-DORL_TIMING): per-phase cycle sums of this unit's persistent kernels; zeros otherwise
template <int W> int prof_read(unsigned long long* out48, int reset) {
#ifdef ORL_TIMING
  if (reset == 2) return hipMemcpyFromSymbol(out48, HIP_SYMBOL(g_wts), 16384 * 8 * 8) == hipSuccess ? 0 : -1;
  if (reset == 3) return hipMemcpyFromSymbol(out48, HIP_SYMBOL(sp::g_prof), (size_t)ORL_PROF_WAVES * ORL_PROF_SLOTS * 8) == hipSuccess ? 0 : -1;
#endif
  for (int k = 0; k < ORL_PROF_SLOTS; k++) out48[k] = 0;
#ifdef ORL_TIMING
  std::vector<unsigned long long> h((size_t)ORL_PROF_WAVES * ORL_PROF_SLOTS);
  if (hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(sp::g_prof), h.size() * 8) != hipSuccess) return -1;
  for (size_t w = 0; w < ORL_PROF_WAVES; w++)
    for (int k = 0; k < ORL_PROF_SLOTS; k++) out48[k] += h[w * ORL_PROF_SLOTS + k];
  if (reset) {
    std::fill(h.begin(), h.end(), 0ull);
    if (hipMemcpyToSymbol(HIP_SYMBOL(sp::g_prof), h.data(), h.size() * 8) != hipSuccess) return -1;
  }
#else
  (void)reset;
#endif
  return 0;
}

template void reset<ORL_W>(orl_batch*, int, const unsigned char*);
template int prof_read<ORL_W>(unsigned long long*, int);
template void policy<ORL_W>(orl_batch*, int);
template void step64<ORL_W>(orl_batch*, int, int, int);
template void obs<ORL_W>(orl_batch*, int);
template void persist<ORL_W>(orl_batch*, const DevParams&, hipStream_t, int, int, int*, unsigned int*, unsigned int*, int);
template int persist_resident<ORL_W>(orl_batch*, int);
template int persist_uses_lds<ORL_W>(orl_batch*);
template void step2<ORL_W>(orl_batch*, int);
template void agent_step<ORL_W>(orl_batch*, int, int);
template void persist_form<ORL_W>(const DevParams&, int*, int*);

}  // namespace orl_launch
#endif  // ORL_SPEC_ONLY
