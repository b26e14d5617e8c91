// orl_gpu.hip — kernels + C ABI (include/orl.h) of liborlgpu.so.  gfx950 only.
//
// Kernels (one 64-lane wavefront = one workgroup = one env):
//   k_init_mt   CPython MT19937 state -> in-place "update-behind" form (orl_device.h, Rng)
//   k_reset     full reset (clear network, draw first service) or soft reset (episode counters)
//   k_policy    slot-scan: AND the link rows of each of the k paths out of LDS, log-step run
//               detection, first fit -> action            [the HBM-streaming kernel of the roofline]
//   k_step      apply action, statistics, reward/info, next service (RNG, releases), observation
//   k_obs       DeepRMSA observation of the pending service
// Host side: plain HIP runtime, one stream per batch, no torch types.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <string>
#include <vector>

#include "../../include/orl.h"
#include "orl_device.h"
#include "orl_device_g8.h"
#include "orl_device_split.h"

using namespace orl;

// =============================================================================================
// kernels
// =============================================================================================
extern __shared__ __attribute__((aligned(16))) unsigned char orl_lds_raw[];

__global__ void __launch_bounds__(64) k_init_mt(DevParams P, const u32* raw) {
  const i64 env = blockIdx.x;
  const int lane = lane_id();
  u32* m = (u32*)orl_lds_raw;
  const u32* src = raw + env * 625;
  for (int i = lane; i < 624; i += 64) m[i] = src[i];
  int p0 = (int)src[624];
  if (p0 > 624) p0 = 624;
  wave_fence();
  // positions < p0 were already handed out by CPython: advance them to the next generation
  for (int base = 0; base < p0; base += 64) {
    int i = base + lane;
    u32 nw = 0;
    if (i < p0) {
      int i1 = i + 1 >= 624 ? i + 1 - 624 : i + 1;
      int im = i + 397 >= 624 ? i + 397 - 624 : i + 397;
      u32 y = (m[i] & 0x80000000u) | (m[i1] & 0x7fffffffu);
      nw = m[im] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    }
    wave_fence();
    if (i < p0) m[i] = nw;
    wave_fence();
  }
  u32* dst = P.mt + env * 624;
  for (int i = lane; i < 624; i += 64) dst[i] = m[i];
  u64 v = 0;
  if (lane == SC_ID_MTPOS) v = pack2(0, p0 >= 624 ? 0 : p0);
  if (lane < ORL_SCAL_WORDS) P.scal[env * ORL_SCAL_WORDS + lane] = v;
}

// random.Random(seed) on the device: CPython's random_seed() takes abs(seed), splits it into 32-bit words
// (little-endian, at least one) and calls init_by_array (Modules/_randommodule.c).  One thread per env; the
// resulting 624 words + index 624 go to the same raw buffer k_init_mt converts.
__global__ void k_seed_mt(const long long* seeds, i64 n, u32* raw) {
  i64 env = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= n) return;
  u32* mt = raw + env * 625;
  long long sv = seeds[env];
  u64 a = sv < 0 ? (u64)(-(sv + 1)) + 1ull : (u64)sv;
  u32 key[2] = {(u32)a, (u32)(a >> 32)};
  const int klen = key[1] ? 2 : 1;
  mt[0] = 19650218u;
  for (int i = 1; i < 624; i++) mt[i] = 1812433253u * (mt[i - 1] ^ (mt[i - 1] >> 30)) + (u32)i;
  int i = 1, j = 0;
  for (int k = 624; k; k--) {
    mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1664525u)) + key[j] + (u32)j;
    i++; j++;
    if (i >= 624) { mt[0] = mt[623]; i = 1; }
    if (j >= klen) j = 0;
  }
  for (int k = 623; k; k--) {
    mt[i] = (mt[i] ^ ((mt[i - 1] ^ (mt[i - 1] >> 30)) * 1566083941u)) - (u32)i;
    i++;
    if (i >= 624) { mt[0] = mt[623]; i = 1; }
  }
  mt[0] = 0x80000000u;
  mt[624] = 624u;
}

// SimpleMatrixObservation (rmsa_env.py:806-837, rmcsa_env.py:914-947): [one-hot(min(src,dst)) | one-hot(max(src,dst)) |
// available_slots.flatten()] as uint8, one workgroup per env, bits unpacked 8 per thread-iteration.
__global__ void k_matrix_obs(DevParams P, unsigned char* out) {
  const i64 env = blockIdx.x;
  const int N = P.N, S = P.S, rows = P.C * P.E;
  const int dim = 2 * N + rows * S;
  unsigned char* o = out + env * dim;
  const u64 sd = P.scal[env * ORL_SCAL_WORDS + SC_SRC_DST];
  const int src = (int)(u32)sd, dst = (int)(sd >> 32);
  const int mn = src < dst ? src : dst, mx = src < dst ? dst : src;
  for (int i = threadIdx.x; i < 2 * N; i += blockDim.x) o[i] = (i == mn || i == N + mx) ? 1 : 0;
  const u64* bm = P.bitmap + env * P.bm_words;
  for (int i = threadIdx.x; i < rows * S; i += blockDim.x) {
    int r = i / S, sl = i - r * S;
    o[2 * N + i] = (unsigned char)((bm[r * P.W + (sl >> 6)] >> (sl & 63)) & 1ull);
  }
}

template <int ENV, int W>
__global__ void __launch_bounds__(64) k_reset(DevParams P, int full, const unsigned char* mask) {
  const i64 env = blockIdx.x;
  const int lane = lane_id();
  if (mask && !mask[env]) return;
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
    lds[i] = (i < P.C * P.E * W) ? v : 0ull;
  }
  for (int i = lane; i < 4 * P.E; i += 64) e.ls[i] = 0.0;
  for (int i = lane; i < P.cs_words; i += 64) e.cs[i] = 0;
  for (int i = lane; i < P.ev_cap; i += 64) e.ev_time[i] = __builtin_inf();
  if (P.br_hist) for (int i = lane; i < 2 * P.n_br; i += 64) P.br_hist[env * 2 * P.n_br + i] = 0;
  if (P.act_hist) for (int i = lane; i < (P.K + 1) + (P.S + 1); i += 64) P.act_hist[env * ((P.K + 1) + (P.S + 1)) + i] = 0;
  wave_fence();
  e.now = 0; e.at = 0; e.ht = 0; e.g_thr = 0; e.g_comp = 0; e.g_last = 0;
  e.sp = e.sa = e.esp = e.esa = e.brq = e.brp = e.ebrq = e.ebrp = e.s_br = e.s_nh = 0;
  e.src = e.dst = e.bit_rate = e.br_idx = e.id = 0;
  e.ev_hwm = 0; e.ev_cnt = 0; e.new_service = 0; e.flags = 0;
  next_service<ENV, W, false>(P, e, lane, nullptr);
  stage_out(P, e, lane);
  env_store(P, e, lane);
}

// Slot-scan kernel.  GS lanes per env: a wavefront serves 64/GS envs whose slot maps are contiguous in HBM,
// so staging them into LDS is one fully coalesced stream of 16-B-per-lane loads (1 KiB per instruction).
// Workgroup = 4 wavefronts, each with its own LDS window; no workgroup-level synchronisation is needed.
#ifndef ORL_POLICY_LDS
#define ORL_POLICY_LDS 0  // 1 = stage the slot maps through LDS in k_policy (kept for A/B measurements)
#endif
#ifndef ORL_POLICY_WAVES
#define ORL_POLICY_WAVES 4  // wavefronts per workgroup in the slot-scan kernel
#endif
template <int ENV, int W, int GS>
__global__ void __launch_bounds__(64 * ORL_POLICY_WAVES) k_policy(DevParams P, int pol) {
  constexpr int EPW = 64 / GS;  // envs per wavefront
  const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
  const i64 env0 = ((i64)blockIdx.x * ORL_POLICY_WAVES + wave) * EPW;
  if (env0 >= P.B) return;
#if ORL_POLICY_LDS == 0
  // Link rows are read straight from global memory (each lane 40-B rows of its own path).  A/B on MI355X, cfg2,
  // B = 65 536: 17.2 us per launch vs 19.4 us with LDS staging — the 28 KB/workgroup LDS window caps residency at
  // 5 workgroups/CU and costs a second dispatch round, while the 7-KB-per-wave footprint stays L2/TCP resident.
  const u64* lds = P.bitmap + env0 * P.bm_words;
#else
  u64* lds = (u64*)orl_lds_raw + (size_t)wave * EPW * P.bm_words;
  {
    i64 nenv = P.B - env0 < EPW ? P.B - env0 : EPW;
    const int n16 = (int)(nenv * (P.bm_words / 2));
    const ulonglong2* g = (const ulonglong2*)(P.bitmap + env0 * P.bm_words);
    ulonglong2* l = (ulonglong2*)lds;
    for (int i = lane; i < n16; i += 64) l[i] = g[i];
  }
#endif
  const int grp = lane / GS;
  const i64 env = env0 + grp;
  const bool valid = env < P.B;
  u64 d = valid ? P.svc_desc[env] : 0ull;
  wave_fence();
  int a[4];
  policy_g<ENV, W, GS>(P, lds + (size_t)grp * P.bm_words, valid, (int)(u32)d, (int)((d >> 32) & 0xffffu), (int)((d >> 48) & 0xffu),
                       lane, pol, a);
  if (valid && (lane & (GS - 1)) == 0) *(int4*)(P.actions + env * 4) = make_int4(a[0], a[1], a[2], a[3]);
}

#ifndef ORL_STEP_WAVES
#define ORL_STEP_WAVES 5  // waves per SIMD the register allocator must leave room for (measured: 4 -> 428 us, 5 -> 389 us, 6 -> 436 us)
#endif
template <int ENV, int W, bool EVL>
__global__ void __launch_bounds__(64, ORL_STEP_WAVES) k_step(DevParams P, int auto_reset, int want_info, int pol) {
  const i64 env = blockIdx.x;
  const int lane = lane_id();
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
    // device-resident loop: the slot scan runs right here on the LDS copy of the slot map (lanes = paths, or (path, core)
    // pairs) — one launch and one read of the map per policy + step instead of two
    int a[4];
    policy_g<ENV, W, 64>(P, e.bm, true, pair_base(P, e.src, e.dst), e.br_idx, P.n_paths[e.src * P.N + e.dst], lane, pol, a);
    av = make_int4(a[0], a[1], a[2], a[3]);
    if (lane == 0) *(int4*)(P.actions + env * 4) = av;
  }
  // Round trip 2: what the scalar record addresses — the MT window of the next service, the pending release times
  // (EVL: into LDS for all scans) and the path record + slot count of the action's path.
  Rng pre;
  rng_fill(e, pre, lane);
  {
    const int route = (ENV == ENV_DEEPRMSA) ? (av.x >= 0 ? av.x / P.J : P.K) : av.x;
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

// step(), 8 lanes per env / 8 envs per wavefront (orl_device_g8.h): state is updated in place in HBM, no LDS
#ifndef ORL_STEP8_WAVES
#define ORL_STEP8_WAVES 1
#endif
template <int ENV, int W>
__global__ void __launch_bounds__(256, ORL_STEP8_WAVES) k_step8(DevParams P, int auto_reset, int want_info) {
  const int lane = lane_id(), wave = (int)(threadIdx.x >> 6);
  const i64 env0 = ((i64)blockIdx.x * 4 + wave) * 8;
  if (env0 >= P.B) return;
  // The 8 envs of a wavefront have contiguous slot maps and link statistics in HBM: stage both into this wave's
  // LDS window with 16-B-per-lane loads, work there, write back at the end.
  const int per_env = P.bm_words + 4 * P.E;  // u64 words
  u64* win = (u64*)orl_lds_raw + (size_t)wave * 8 * per_env;
  const i64 nenv = P.B - env0 < 8 ? P.B - env0 : 8;
  {
    const ulonglong2* g = (const ulonglong2*)(P.bitmap + env0 * P.bm_words);
    ulonglong2* l = (ulonglong2*)win;
    for (int i = lane; i < (int)(nenv * (P.bm_words / 2)); i += 64) l[i] = g[i];
    const double* gs = P.lstat + env0 * 4 * P.E;
    double* ls = (double*)(win + 8 * P.bm_words);
    for (int i = lane; i < (int)(nenv * 4 * P.E); i += 64) ls[i] = gs[i];
  }
  const i64 env = env0 + (lane >> 3);
  wave_fence();
  if (env < P.B) {
    g8::EnvG e;
    g8::env_load(P, e, env);
    e.bm = win + (size_t)(lane >> 3) * P.bm_words;
    e.ls = (double*)(win + 8 * P.bm_words) + (size_t)(lane >> 3) * 4 * P.E;
    int4 av = *(const int4*)(P.actions + env * 4);
    int act[4] = {av.x, av.y, av.z, av.w};
    g8::step<ENV, W>(P, e, lane, act, auto_reset, want_info != 0);
    g8::env_store(P, e, lane & 7);
  }
  wave_fence();
  {
    ulonglong2* g = (ulonglong2*)(P.bitmap + env0 * P.bm_words);
    const ulonglong2* l = (const ulonglong2*)win;
    for (int i = lane; i < (int)(nenv * (P.bm_words / 2)); i += 64) g[i] = l[i];
    double* gs = P.lstat + env0 * 4 * P.E;
    const double* ls = (const double*)(win + 8 * P.bm_words);
    for (int i = lane; i < (int)(nenv * 4 * P.E); i += 64) gs[i] = ls[i];
  }
}

// ---- split pipeline (orl_device_split.h): 32 envs per workgroup in the control kernels, 32 items per workgroup-pass
// in the row kernel ---------------------------------------------------------------------------------------------
template <int ENV, int W>
__global__ void __launch_bounds__(256) k_ctrl_a(DevParams P, int want_info) {
  const i64 env = (i64)blockIdx.x * 32 + (threadIdx.x >> 3);
  sp::Prof prof;
  sp::ctrl_a<ENV, W, 0>(P, env, env < P.B, lane_id(), want_info != 0, prof);
}
// device-policy loop: slot-scan and control kernel A in one launch (same 8-lanes-per-env layout; the action never
// leaves the registers, the link rows the scan just read are still in cache for the validation)
template <int ENV, int W>
__global__ void __launch_bounds__(256) k_policy_ctrl_a(DevParams P, int pol) {
  const int lane = lane_id();
  const i64 env = (i64)blockIdx.x * 32 + (threadIdx.x >> 3);
  const bool valid = env < P.B;
  const i64 env0 = env - ((lane >> 3));  // first env of this wavefront
  sp::Prof prof;
  ORL_PROFA_BEGIN();
  u64 d = valid ? P.svc_desc[env] : 0ull;
  int a[4];
  policy_g<ENV, W, 8>(P, P.bitmap + env0 * P.bm_words + (size_t)(lane >> 3) * P.bm_words, valid, (int)(u32)d,
                      (int)((d >> 32) & 0xffffu), (int)((d >> 48) & 0xffu), lane, pol, a);
  const int4 av = make_int4(a[0], a[1], a[2], a[3]);
  if (valid && (lane & 7) == 0) *(int4*)(P.actions + env * 4) = av;
  ORL_PROFA(1);
  sp::ctrl_a<ENV, W, 1>(P, env, valid, lane, false, prof, &av);
  ORL_PROFA_END();
}
#ifndef ORL_ROWS1_GROUPS
#define ORL_ROWS1_GROUPS 3
#endif
// ---- two-kernel pipeline (step_impl 2): k_step_a2 ; k_rows2 -------------------------------------------------------
// slot scan + everything of step() that is per-env control: validation, counters, release push, next service, the due
// releases of the step; output = one queue of mixed work items (orl_device_split.h, ctrl_a<MERGE = 2>)
template <int ENV, int W, bool FUSED_POLICY>
__global__ void __launch_bounds__(256) k_step_a2(DevParams P, int pol, int parity) {
  __shared__ u32 s_tally[32 * 32];
  const int lane = lane_id();
  const i64 env = (i64)blockIdx.x * 32 + (threadIdx.x >> 3);
  const bool valid = env < P.B;
  sp::Prof prof;
  ORL_PROFA_BEGIN();
  if (FUSED_POLICY) {
    const i64 env0 = env - ((lane >> 3));  // first env of this wavefront
    u64 d = valid ? P.svc_desc[env] : 0ull;
    int a[4];
    policy_g<ENV, W, 8>(P, P.bitmap + env0 * P.bm_words + (size_t)(lane >> 3) * P.bm_words, valid, (int)(u32)d,
                        (int)((d >> 32) & 0xffffu), (int)((d >> 48) & 0xffu), lane, pol, a);
    const int4 av = make_int4(a[0], a[1], a[2], a[3]);
    if (valid && (lane & 7) == 0) *(int4*)(P.actions + env * 4) = av;
    ORL_PROFA(1);
    sp::ctrl_a<ENV, W, 2>(P, env, valid, lane, false, prof, &av, s_tally, (sp::SinkEntry*)orl_lds_raw, parity);
  } else {
    sp::ctrl_a<ENV, W, 2>(P, env, valid, lane, false, prof, nullptr, s_tally, (sp::SinkEntry*)orl_lds_raw, parity);
  }
  ORL_PROFA_END();
}
// one lane per mixed item
// One workgroup per control workgroup (32 envs, ~140 mixed items, 192 threads): a thread handles at most one item in
// all but one launch in 10^3, so the kernel's duration is one item's dependent chain.  Measured, us per 65 536-env launch
// between events: 1 group x 192 threads 55.9, x 256 57.8, x 128 66.2; 2 groups x 256 62.6, 3: 63.8, 4: 70.5, 6: 80.1.  (A
// static thread -> slot map that requests the item together with the fill count was slower: 62.9.)
#ifndef ORL_ROWS2_GROUPS
#define ORL_ROWS2_GROUPS 1
#endif
#ifndef ORL_ROWS2_THREADS
#define ORL_ROWS2_THREADS 192
#endif
template <int ENV, int W>
__global__ void __launch_bounds__(ORL_ROWS2_THREADS) k_rows2(DevParams P, int parity) {
  constexpr int NR = 4 * ORL_ROWS2_GROUPS;
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
#if defined(ORL_TIMING) && ORL_TIMING == 4
    if (it.a.x == 0x123456789abcdefull) return;  // (keeps the item loads ahead of the timing mark)
#endif
    ORL_PROFR(2);
    sp::row_item_lane<ENV, W, true>(P, it, SC_NOW, prof);
  }
  ORL_PROFR(8);
  ORL_PROFR(9);
  ORL_PROFR_END();
}
// ---- persistent form of the two-kernel pipeline (step_impl 3) ----------------------------------------------------------
// Envs never interact, so a workgroup can own its 32 envs for a whole run: control phase (the body of k_step_a2) ->
// barrier -> row phase over the items its own wavefronts just emitted (the body of k_rows2) -> barrier -> next step, with
// no kernel boundary and no grid-wide tail between the phases; the workgroups of a launch drift out of phase and keep
// the memory system uniformly busy (what only a 10^6-env batch achieves with separate launches).  Data written in one
// phase and read in the next stays within the workgroup: same CU, same L1.  A workgroup in which an env's releases did
// not fit the item form (one env-step in 10^7) leaves the loop after that step's row phase; the host runs k_rel_tail
// and relaunches, every workgroup resuming from its own step count.
template <int W>
__device__ __forceinline__ void obs8_env(const DevParams& P, i64 env, int lane, int with_terminal);
#ifndef ORL_PERSIST_WAVES
#define ORL_PERSIST_WAVES 4   // waves per SIMD the register allocator must leave room for
#endif
#ifndef ORL_PERSIST_WG
#define ORL_PERSIST_WG 1      // wavefronts (8 envs each) per workgroup of the persistent kernel
#endif
template <int ENV, int W>
__device__ __forceinline__ void persist_body(const DevParams& P, int pol, int target, int* wg_step, u32* n_unfinished) {
  constexpr int NW = ORL_PERSIST_WG;
  __shared__ u32 s_tally[32 * 8 * NW];
  __shared__ int s_deferred[2];  // alternating by step: a flag is cleared only after every thread has passed the next barrier
  const int lane = lane_id();
  const i64 env = (i64)blockIdx.x * (8 * NW) + (threadIdx.x >> 3);
  int step = wg_step[blockIdx.x];
  sp::Prof prof;
  if (threadIdx.x == 0) s_deferred[0] = s_deferred[1] = 0;
  while (step < target) {
    __syncthreads();  // the previous row phase's writes are visible to every wavefront of the workgroup
    if (threadIdx.x == 0) s_deferred[(step + 1) & 1] = 0;
    // per-iteration opaque copies: without them the compiler hoists every per-lane address out of the loop and keeps
    // them all live across both phases (199 VGPRs instead of ~128)
    int env_lo = (int)env, lane_i = lane;
    asm volatile("" : "+v"(env_lo), "+v"(lane_i));
    const i64 env_i = (i64)env_lo;
    const bool valid_i = env_i < P.B;
    const i64 env0_i = env_i - (lane_i >> 3);
    {
      u64 d = valid_i ? P.svc_desc[env_i] : 0ull;
      int a[4];
      policy_g<ENV, W, 8>(P, P.bitmap + env0_i * P.bm_words + (size_t)(lane_i >> 3) * P.bm_words, valid_i, (int)(u32)d,
                          (int)((d >> 32) & 0xffffu), (int)((d >> 48) & 0xffu), lane_i, pol, a);
      const int4 av = make_int4(a[0], a[1], a[2], a[3]);
      if (valid_i && (lane_i & 7) == 0) *(int4*)(P.actions + env_i * 4) = av;
      sp::ctrl_a<ENV, W, 2>(P, env_i, valid_i, lane_i, false, prof, &av, s_tally, (sp::SinkEntry*)orl_lds_raw, 0, &s_deferred[step & 1]);
    }
    __syncthreads();  // items, fill counts, env records
    {
      u32 r0 = blockIdx.x * (u32)NW, tid = threadIdx.x;
      asm volatile("" : "+s"(r0), "+v"(tid));
      const u32* cnt = P.q_cnt_a + r0;
      u32 cum[NW + 1];
      cum[0] = 0;
#pragma unroll
      for (int j = 0; j < NW; j++) cum[j + 1] = cum[j] + cnt[j];
      const ulonglong2* q = P.q_a + (size_t)r0 * P.q_wave * 2;
      for (u32 idx = tid; idx < cum[NW]; idx += 64 * NW) {
        u32 j = 0, base = 0;
#pragma unroll
        for (int t = 1; t < NW; t++)
          if (idx >= cum[t]) { j = (u32)t; base = cum[t]; }
        const size_t at = (size_t)j * P.q_wave + (idx - base);
        sp::Item it;
        it.a = q[2 * at];
        it.b = q[2 * at + 1];
        sp::row_item_lane<ENV, W, true>(P, it, SC_NOW, prof);
      }
    }
    if (ENV == ENV_DEEPRMSA) {  // the observation of the new pending service, from the rows as they are now
      __syncthreads();
      if (valid_i) obs8_env<W>(P, env_i, lane_i, 1);
    }
    step++;
    if (s_deferred[(step - 1) & 1]) break;  // set before the barrier in front of the row phase
  }
  if (threadIdx.x == 0) {
    wg_step[blockIdx.x] = step;
    if (step < target) atomicAdd(n_unfinished, 1u);
  }
}
// Two register budgets of the same body: 4 waves/SIMD (128 VGPRs, a few spills) is the better trade for NSFNET-sized
// RMSA / RWA / DeepRMSA (cfg2 6.7e8 vs 6.2e8, cfg1 7.1e8 vs 6.1e8, cfg3 7.1e8 vs 5.8e8), 3 waves/SIMD (168 VGPRs, no spills)
// for the heavier RMCSA and Germany50 steps (cfg4 5.0e8 vs 4.6e8, cfg5 3.4e8 vs 3.2e8).
template <int ENV, int W>
__global__ void __launch_bounds__(64 * ORL_PERSIST_WG) __attribute__((amdgpu_waves_per_eu(ORL_PERSIST_WAVES, ORL_PERSIST_WAVES)))
k_persist(DevParams P, int pol, int target, int* wg_step, u32* n_unfinished) {
  persist_body<ENV, W>(P, pol, target, wg_step, n_unfinished);
}
template <int ENV, int W>
__global__ void __launch_bounds__(64 * ORL_PERSIST_WG) __attribute__((amdgpu_waves_per_eu(3, 3)))
k_persist3(DevParams P, int pol, int target, int* wg_step, u32* n_unfinished) {
  persist_body<ENV, W>(P, pol, target, wg_step, n_unfinished);
}

// serial tail, one small workgroup per launch: the envs whose releases of this step did not fit the item form (about
// one env-step in 10^7; control kernels append them to q_def) release them in place, 8 lanes per env.  A launch of its
// own because inlined into the row kernels this code cost them half their occupancy.
template <int ENV, int W>
__global__ void __launch_bounds__(256) k_rel_tail(DevParams P, int buffer) {
  const u32* dq = P.q_def + (size_t)buffer * P.q_def_stride;
  const u32 nd = dq[0];
  for (u32 d = threadIdx.x >> 3; d < nd; d += 32u) sp::rel_serial<ENV, W>(P, (i64)dq[16 + d], lane_id());
}
// end of a device-resident run: the network-compactness update the last step left pending (one thread per env), so
// that every host-visible state is final
__global__ void k_finish2(DevParams P) {
  const i64 env = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= P.B) return;
  u64* s = P.scal + env * ORL_SCAL_WORDS;
  int* rs = P.core_sums + env * P.cs_words + 2 * P.C;
  const u64 acc = s[SC_ACC];
  if ((u32)acc & 2u) {
    const int* cs = P.core_sums + env * P.cs_words;
    const int c0 = (int)((acc >> 32) & 31);
    const i64 s_nh_prov = (i64)(acc >> 37);
    const int occ = cs[2 * c0] - rs[2 * c0], fb = cs[2 * c0 + 1] - rs[2 * c0 + 1];
    const double a0 = __longlong_as_double((i64)s[SC_GC_A]), td = __longlong_as_double((i64)s[SC_GC_TD]);
    const double now_a = __longlong_as_double((i64)s[SC_NOWA]);
    const double cmp = (fb > 0) ? ((double)occ / (double)s_nh_prov) * ((double)P.E / (double)fb) : 1.0;
    s[SC_GCOMP] = (u64)__double_as_longlong((a0 + (cmp * td)) / now_a);
    s[SC_ACC] = acc & ~2ull;
  }
  for (int i = 0; i < 2 * P.C; i++) rs[i] = 0;
}

template <int ENV, int W>
__global__ void __launch_bounds__(256) k_ctrl_b1(DevParams P, int auto_reset, int want_info) {
  const i64 env = (i64)blockIdx.x * 32 + (threadIdx.x >> 3);
  sp::ctrl_b1<ENV, W>(P, env, env < P.B, lane_id(), auto_reset, want_info != 0);
}
template <int ENV, int W>
__global__ void __launch_bounds__(256) k_ctrl_b2(DevParams P) {
  __shared__ u32 s_tally[32 * 32];
  const i64 env = (i64)blockIdx.x * 32 + (threadIdx.x >> 3);
  sp::ctrl_b2<ENV, W>(P, env, env < P.B, lane_id(), s_tally, (sp::SinkEntry*)orl_lds_raw);  // 32 x E entries
}
// lane-per-item row kernel: one workgroup covers ORL_ROWS1_GROUPS control workgroups (32 envs each, ~75 items) and maps
// its threads onto the dense item index over their 4 * ORL_ROWS1_GROUPS wavefront regions
template <int ENV, int W>
__global__ void __launch_bounds__(256) k_rows1(DevParams P, int phase) {
  constexpr int NR = 4 * ORL_ROWS1_GROUPS;
  sp::Prof prof;
  ORL_PROFR_BEGIN();
  const u32 r0 = blockIdx.x * NR;
  const u32 n_regions = (u32)((P.B + 31) / 32) * 4u;  // regions the control kernels of this launch wrote
  const u32* cnt = (phase ? P.q_cnt_b : P.q_cnt_a) + r0;
  u32 cum[NR + 1];
  cum[0] = 0;
#pragma unroll
  for (int j = 0; j < NR; j++) cum[j + 1] = cum[j] + ((r0 + j < n_regions) ? cnt[j] : 0u);
  const ulonglong2* q = (phase ? P.q_b : P.q_a) + (size_t)r0 * P.q_wave * 2;
  for (u32 idx = threadIdx.x; idx < cum[NR]; idx += 256) {
    u32 j = 0, base = 0;
#pragma unroll
    for (int t = 1; t < NR; t++)
      if (idx >= cum[t]) { j = (u32)t; base = cum[t]; }
    const size_t at = (size_t)j * P.q_wave + (idx - base);
    ORL_PROFR(1);
    sp::Item it;
    it.a = q[2 * at];
    it.b = q[2 * at + 1];
#if defined(ORL_TIMING) && ORL_TIMING == 4
    if (it.a.x == 0x123456789abcdefull) return;  // (keeps the item loads ahead of the timing mark)
#endif
    ORL_PROFR(2);
    sp::row_item_lane<ENV, W, false>(P, it, phase ? SC_NOW : SC_NOWA, prof);
  }
  ORL_PROFR(8);
  ORL_PROFR(9);
  ORL_PROFR_END();
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
template <int W>
__device__ __forceinline__ void obs8_env(const DevParams& P, i64 env, int lane, int with_terminal) {
  const int gl = lane & 7;
  const u64* s = P.scal + env * ORL_SCAL_WORDS;
  u64 t = s[SC_SRC_DST];
  const int src = (int)(u32)t, dst = (int)(t >> 32);
  t = s[SC_BR_IDX];
  const int bit_rate = (int)(u32)t, br_idx = (int)(t >> 32);
  const int N = P.N, J = P.J, S = P.S, WD = 2 * J + 3;
  double* o = P.obs + env * P.obs_dim;
  double* o2 = (with_terminal && P.done[env]) ? P.term_obs + env * P.obs_dim : nullptr;
  const int mn = src < dst ? src : dst, mx = src < dst ? dst : src;
  for (int i = gl; i < 1 + 2 * N; i += 8) {
    const double v = (i == 0) ? (double)bit_rate / 100 : ((i == 1 + mn || i == 1 + N + mx) ? 1.0 : 0.0);
    o[i] = v;
    if (o2) o2[i] = v;
  }
  if (gl < P.K) {
    double f[19];  // 2 * J + 3 <= 19
#pragma unroll
    for (int i = 0; i < 19; i++) f[i] = -1.0;
    if (gl < P.n_paths[src * N + dst]) {
      const int pidx = (src * N + dst) * P.K + gl;
      const Row<W> m = path_and_rec<W>(path_rec_load(P, pidx), P.bitmap + env * P.bm_words, P.E, S, 0);
      const int n = P.nslots_path[(size_t)pidx * P.n_br + br_idx];
      Row<W> r = row_runs_ge<W>(m, n);
      const Row<W> zeros = row_andn<W>(row_mask_lo<W>(S), m);
#pragma unroll
      for (int b = 0; b < 8; b++) {
        if (b < J && row_any<W>(r)) {
          const int st = row_ctz<W>(r);
          const Row<W> z = row_andn<W>(zeros, row_mask_lo<W>(st));
          const int end = row_any<W>(z) ? row_ctz<W>(z) : S;
          f[2 * b] = 2 * ((double)st - 0.5 * (double)S) / (double)S;
          f[2 * b + 1] = (double)(end - st - 8) / 8;
          r = row_andn<W>(r, row_mask_lo<W>(end));
        }
      }
      const double fn = ((double)n - 5.5) / 3.5;
      const int tot = row_popc<W>(m);
      const double ft = 2 * ((double)tot - 0.5 * (double)S) / (double)S;
      const int nruns = row_popc<W>(row_starts<W>(m));
      const double fr = (nruns > 0) ? ((double)tot / (double)nruns - 4) / 4 : -1.0;
      // f[2J], f[2J+1], f[2J+2] with a run-time J: written below by position
#pragma unroll
      for (int i = 0; i < 19; i++) f[i] = (i == 2 * J) ? fn : (i == 2 * J + 1) ? ft : (i == 2 * J + 2) ? fr : f[i];
    }
    double* sp = o + 1 + 2 * N + gl * WD;
#pragma unroll
    for (int i = 0; i < 19; i++)
      if (i < WD) { sp[i] = f[i]; if (o2) o2[1 + 2 * N + gl * WD + i] = f[i]; }
  }
}
template <int W>
__global__ void __launch_bounds__(256) k_obs8(DevParams P, int with_terminal) {
  const i64 env = (i64)blockIdx.x * 32 + (threadIdx.x >> 3);
  if (env >= P.B) return;
  obs8_env<W>(P, env, lane_id(), with_terminal);
}

// Counter calibration: streams the whole slot-map array once with a known byte count (FETCH_SIZE on gfx950 is
// documented to under-report wide coalesced reads; this gives the factor for our own access widths).
__global__ void k_calib_read(const u64* __restrict__ src, i64 n_words, int width16, u64* sink) {
  i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  u64 acc = 0;
  if (width16) {
    const ulonglong2* s2 = (const ulonglong2*)src;
    for (i64 j = i; j < n_words / 2; j += (i64)gridDim.x * blockDim.x) { ulonglong2 v = s2[j]; acc ^= v.x ^ v.y; }
  } else {
    for (i64 j = i; j < n_words; j += (i64)gridDim.x * blockDim.x) acc ^= src[j];
  }
  if (acc == 0x123456789abcdefull) *sink = acc;  // keep the loads alive
}

// sums of services_processed / services_accepted over the batch (two atomics per wave)
__global__ void k_totals(DevParams P, unsigned long long* out) {
  i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  i64 sp = 0, sa = 0;
  if (i < P.B) { sp = (i64)P.scal[i * ORL_SCAL_WORDS + SC_SP]; sa = (i64)P.scal[i * ORL_SCAL_WORDS + SC_SA]; }
  for (int o = 32; o > 0; o >>= 1) { sp += __shfl_xor(sp, o, 64); sa += __shfl_xor(sa, o, 64); }
  if ((threadIdx.x & 63) == 0) { atomicAdd(out, (unsigned long long)sp); atomicAdd(out + 1, (unsigned long long)sa); }
}

// =============================================================================================
// host side
// =============================================================================================
static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}
#define HIPCHK(x)                                                                                     \
  do {                                                                                                \
    hipError_t _e = (x);                                                                              \
    if (_e != hipSuccess) return fail(ORL_E_HIP, "%s failed: %s", #x, hipGetErrorString(_e));         \
  } while (0)

struct orl_topology {
  int device;
  int N, E, K, H, M;
  std::vector<int32_t> h_hops, h_links, h_mod;  // host copies (per-batch derived tables are built from them)
  int* n_paths;
  double* path_length;
  int* edge_iter_order;
  int* link_pos;
};

struct TkRec;
struct orl_batch {
  DevParams P;
  TkRec* tk = nullptr;  // per-kernel timing of orl_batch_run(time_kernels = 1)
  int parity[66] = {0};  // two-kernel pipeline: which deferred-env buffer the next step of view k writes (0 = whole batch)
  int persist = 0;       // device-resident runs through the persistent kernel (k_persist)
  int host_wave64 = 0;   // host-driven step() launches the one-wavefront-per-env kernel (small batches)
  int64_t persist_launches = 0;
  int* d_wg_step = nullptr;        // [ceil(B/32)] steps each workgroup of the persistent kernel has completed in this run
  unsigned int* d_unfinished = nullptr;
  int device, wt;
  int step_impl;  // 64 = one wavefront per env, 8 = eight lanes per env (monolithic), 1 = four-kernel split pipeline, 2 = two-kernel pipeline
  hipStream_t stream;
  std::vector<void*> allocs;
  hipEvent_t ev0, ev1;
  unsigned long long* d_totals;
  // sub-batches for the device-resident run loop: contiguous env ranges, each driven on its own stream so that the
  // short bandwidth-bound slot-scan of one range overlaps the long issue-bound step kernel of another
  std::vector<DevParams> subs;
  std::vector<hipStream_t> sub_streams;    // stream of sub-batch k (several sub-batches may share one)
  std::vector<hipStream_t> owned_streams;  // the distinct streams behind sub_streams
  const DevParams* view;
  hipStream_t view_stream;
};

template <typename T, typename S>
static int upload_conv(T** out, const S* src, size_t n, std::vector<void*>* track) {
  std::vector<T> tmp(n);
  for (size_t i = 0; i < n; i++) tmp[i] = (T)src[i];
  HIPCHK(hipMalloc((void**)out, n * sizeof(T) + 16));
  if (track) track->push_back(*out);
  HIPCHK(hipMemcpy(*out, tmp.data(), n * sizeof(T), hipMemcpyHostToDevice));
  return 0;
}

extern "C" int orl_abi_version(void) { return ORL_ABI_VERSION; }
extern "C" const char* orl_last_error(void) { return g_err.c_str(); }
extern "C" int orl_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

extern "C" int orl_topology_create(const orl_topology_desc* d, int device_id, orl_topology** out) {
  if (!d || !out) return fail(ORL_E_INVALID, "null argument");
  if (d->n_nodes < 2 || d->n_nodes > 512 || d->n_links < 1 || d->n_links > 128 || d->k_paths < 1 || d->k_paths > 64 ||
      d->max_hops < 1 || d->max_hops > 64 || d->n_modulations < 1 || d->n_modulations > 255)
    return fail(ORL_E_INVALID, "topology out of supported range (N<=512, E<=128, k<=64, hops<=64)");
  HIPCHK(hipSetDevice(device_id));
  orl_topology* t = new orl_topology();
  t->n_paths = nullptr;
  t->path_length = nullptr; t->edge_iter_order = nullptr; t->link_pos = nullptr;
  t->device = device_id;
  t->N = d->n_nodes; t->E = d->n_links; t->K = d->k_paths; t->H = d->max_hops; t->M = d->n_modulations;
  size_t nn = (size_t)t->N * t->N, npk = nn * t->K;
  for (size_t i = 0; i < npk; i++)
    if (d->path_hops[i] > t->H || (d->path_modulation[i] >= t->M)) { delete t; return fail(ORL_E_INVALID, "bad path table entry %zu", i); }
  for (size_t i = 0; i < npk * t->H; i++)
    if (d->path_links[i] >= t->E) { delete t; return fail(ORL_E_INVALID, "bad link index in path table"); }
  if (t->H > 30) { delete t; return fail(ORL_E_INVALID, "max_hops must be <= 30"); }
  int rc = 0;
  std::vector<int32_t> mod(npk);
  for (size_t i = 0; i < npk; i++) mod[i] = d->path_modulation[i] < 0 ? 0 : d->path_modulation[i];
  t->h_hops.assign(d->path_hops, d->path_hops + npk);
  t->h_links.assign(d->path_links, d->path_links + npk * t->H);
  t->h_mod = mod;
  rc |= upload_conv(&t->n_paths, d->n_paths, nn, nullptr);
  rc |= upload_conv(&t->path_length, d->path_length, npk, nullptr);
  rc |= upload_conv(&t->edge_iter_order, d->edge_iter_order, (size_t)t->E, nullptr);
  {
    std::vector<int32_t> pos((size_t)t->E, 0);
    for (int i = 0; i < t->E; i++) pos[(size_t)d->edge_iter_order[i]] = i;
    rc |= upload_conv(&t->link_pos, pos.data(), (size_t)t->E, nullptr);
  }
  if (rc) { delete t; return ORL_E_HIP; }
  *out = t;
  return ORL_OK;
}

extern "C" void orl_topology_destroy(orl_topology* t) {
  if (!t) return;
  hipSetDevice(t->device);
  hipFree(t->n_paths);
  hipFree(t->path_length); hipFree(t->edge_iter_order); hipFree(t->link_pos);
  delete t;
}

// ---- launch dispatch over (env family, words per row) ------------------------------------------
#define ORL_FOR_W(CALL)                 \
  switch (b->wt) {                      \
    case 1: CALL(1); break;             \
    case 2: CALL(2); break;             \
    case 5: CALL(5); break;             \
    default: CALL(8); break;            \
  }
#define ORL_FOR_ENV(MACRO)                                  \
  switch (b->P.env_type) {                                  \
    case ENV_RMSA: { MACRO(ENV_RMSA) } break;               \
    case ENV_DEEPRMSA: { MACRO(ENV_DEEPRMSA) } break;       \
    case ENV_RWA: { MACRO(ENV_RWA) } break;                 \
    default: { MACRO(ENV_RMCSA) } break;                    \
  }

// per-kernel timing (orl_batch_run, time_kernels == 1): an event after every launch
struct TkRec { std::vector<hipEvent_t> ev; std::vector<const char*> name; };
#define ORL_TK(NAME) do { if (b->tk) { hipEvent_t e_; hipEventCreate(&e_); hipEventRecord(e_, VS); b->tk->ev.push_back(e_); b->tk->name.push_back(NAME); } } while (0)

static void launch_reset(orl_batch* b, int full, const unsigned char* dmask) {
  const DevParams& VP = b->view ? *b->view : b->P;
  hipStream_t VS = b->view ? b->view_stream : b->stream;
  dim3 g((unsigned)VP.B), blk(64);
  size_t lds = VP.lds_bytes;
#define CALLW(WW) hipLaunchKernelGGL((k_reset<EE, WW>), g, blk, lds, VS, VP, full, dmask)
#define PER_ENV(E_) { constexpr int EE = E_; ORL_FOR_W(CALLW) }
  ORL_FOR_ENV(PER_ENV)
#undef PER_ENV
#undef CALLW
}
static void launch_policy(orl_batch* b, int pol) {
  const DevParams& VP = b->view ? *b->view : b->P;
  hipStream_t VS = b->view ? b->view_stream : b->stream;
  // RMCSA scans (path, core) pairs: one env per wavefront.  The other families put 8 envs on a wavefront when k <= 8.
  const bool wide = (VP.env_type == ENV_RMCSA) || VP.K > 8;
  const int epw = wide ? 1 : 8;
  const i64 per_wg = ORL_POLICY_WAVES * epw;
  dim3 g((unsigned)((VP.B + per_wg - 1) / per_wg)), blk(64 * ORL_POLICY_WAVES);
#if ORL_POLICY_LDS == 0
  size_t lds = 0;
#else
  size_t lds = (size_t)per_wg * VP.bm_words * 8;
#endif
#define CALLW(WW) \
  do { if (wide) hipLaunchKernelGGL((k_policy<EE, WW, 64>), g, blk, lds, VS, VP, pol); \
       else hipLaunchKernelGGL((k_policy<EE, WW, 8>), g, blk, lds, VS, VP, pol); } while (0)
#define PER_ENV(E_) { constexpr int EE = E_; ORL_FOR_W(CALLW) }
  ORL_FOR_ENV(PER_ENV)
#undef PER_ENV
#undef CALLW
  ORL_TK("k_policy");
}
static void launch_obs(orl_batch* b, int with_terminal = 1);
static void launch_step(orl_batch* b, int auto_reset, int want_info, int fused_policy = -1);
// two-kernel pipeline: (slot scan +) all per-env control -> one queue of mixed items -> row kernel
static void launch_step2(orl_batch* b, int pol, bool wide) {
  const DevParams& VP = b->view ? *b->view : b->P;
  hipStream_t VS = b->view ? b->view_stream : b->stream;
  int& par = b->parity[b->view ? (int)(b->view - b->subs.data()) + 1 : 0];
  if (wide) launch_policy(b, pol);  // RMCSA / k > 8: the one-env-per-wavefront slot scan stays a launch of its own
  dim3 gc((unsigned)((VP.B + 31) / 32)), blk(256);
  dim3 gr((gc.x + ORL_ROWS2_GROUPS - 1) / ORL_ROWS2_GROUPS), blk_r(ORL_ROWS2_THREADS);
  const size_t lds_a = (size_t)32 * VP.E * sizeof(sp::SinkEntry);
#define CALLW(WW)                                                                                                   \
  do {                                                                                                              \
    if (wide) {                                                                                                     \
      if (lds_a > 48 * 1024) hipFuncSetAttribute((const void*)k_step_a2<EE, WW, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a); \
      hipLaunchKernelGGL((k_step_a2<EE, WW, false>), gc, blk, lds_a, VS, VP, pol, par);                              \
    } else {                                                                                                        \
      if (lds_a > 48 * 1024) hipFuncSetAttribute((const void*)k_step_a2<EE, WW, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a); \
      hipLaunchKernelGGL((k_step_a2<EE, WW, true>), gc, blk, lds_a, VS, VP, pol, par);                               \
    }                                                                                                               \
    ORL_TK("k_step_a2");                                                                                            \
    hipLaunchKernelGGL((k_rows2<EE, WW>), gr, blk_r, 0, VS, VP, par);                                               \
    ORL_TK("k_rows2");                                                                                              \
    hipLaunchKernelGGL((k_rel_tail<EE, WW>), dim3(1), blk, 0, VS, VP, par);                                         \
    ORL_TK("k_rel_tail");                                                                                           \
  } while (0)
#define PER_ENV(E_) { constexpr int EE = E_; ORL_FOR_W(CALLW) }
  ORL_FOR_ENV(PER_ENV)
#undef PER_ENV
#undef CALLW
  par ^= 1;
  if (VP.obs_dim) launch_obs(b);
}
// after the last step of a device-resident run (every view): nothing pending is left for the host to see
static void launch_finish2(orl_batch* b) {
  const DevParams& VP = b->view ? *b->view : b->P;
  hipStream_t VS = b->view ? b->view_stream : b->stream;
  hipLaunchKernelGGL(k_finish2, dim3((unsigned)((VP.B + 255) / 256)), dim3(256), 0, VS, VP);
}
// policy + step of the device-resident loop; the split pipeline fuses the slot-scan with its first control kernel
static void launch_policy_step(orl_batch* b, int pol) {
  const DevParams& VP = b->view ? *b->view : b->P;
  const bool wide = (VP.env_type == ENV_RMCSA) || VP.K > 8;
  // (the pipelines scan with 8 lanes per env also for RMCSA: the (path, core) pairs are walked 8 at a time in the
  // reference's order and the walk stops at the first batch that fits — usually the first)
  if (b->step_impl == 2) { launch_step2(b, pol, VP.K > 8); return; }
  if (b->step_impl == 1 && !wide) { launch_step(b, 1, 0, pol); return; }
  if (b->step_impl == 64 && !getenv("ORL_UNFUSED_POLICY")) { launch_step(b, 1, 0, pol); return; }  // slot scan inside k_step
  launch_policy(b, pol);
  launch_step(b, 1, 0);
}
static void launch_step(orl_batch* b, int auto_reset, int want_info, int fused_policy) {
  const DevParams& VP = b->view ? *b->view : b->P;
  hipStream_t VS = b->view ? b->view_stream : b->stream;
  if ((b->step_impl == 1 || b->step_impl == 2) && !(b->host_wave64 && fused_policy < 0)) {  // host-driven steps: the four-kernel form
    dim3 gc((unsigned)((VP.B + 31) / 32)), blk(256);
    dim3 gr((gc.x + ORL_ROWS1_GROUPS - 1) / ORL_ROWS1_GROUPS);  // lane-per-item row kernel
    const size_t lds_b2 = (size_t)32 * VP.E * sizeof(sp::SinkEntry);
#define CALLW(WW)                                                                                      \
  do {                                                                                                 \
    if (fused_policy >= 0) { hipLaunchKernelGGL((k_policy_ctrl_a<EE, WW>), gc, blk, 0, VS, VP, fused_policy); ORL_TK("k_policy_ctrl_a"); } \
    else { hipLaunchKernelGGL((k_ctrl_a<EE, WW>), gc, blk, 0, VS, VP, want_info); ORL_TK("k_ctrl_a"); } \
    hipLaunchKernelGGL((k_rows1<EE, WW>), gr, blk, 0, VS, VP, 0);                                       \
    ORL_TK("k_rows(provision)");                                                                       \
    if (fused_policy < 0) { hipLaunchKernelGGL((k_ctrl_b1<EE, WW>), gc, blk, 0, VS, VP, auto_reset, want_info); ORL_TK("k_ctrl_b1"); } \
    if (lds_b2 > 48 * 1024) hipFuncSetAttribute((const void*)k_ctrl_b2<EE, WW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b2); \
    hipLaunchKernelGGL((k_ctrl_b2<EE, WW>), gc, blk, lds_b2, VS, VP);                                     \
    ORL_TK("k_ctrl_b2");                                                                               \
    hipLaunchKernelGGL((k_rows1<EE, WW>), gr, blk, 0, VS, VP, 1);                                       \
    ORL_TK("k_rows(release)");                                                                         \
    hipLaunchKernelGGL((k_rel_tail<EE, WW>), dim3(1), blk, 0, VS, VP, 0);                              \
    ORL_TK("k_rel_tail");                                                                              \
  } while (0)
#define PER_ENV(E_) { constexpr int EE = E_; ORL_FOR_W(CALLW) }
    ORL_FOR_ENV(PER_ENV)
#undef PER_ENV
#undef CALLW
    if (VP.obs_dim) launch_obs(b);
    return;
  }
  const size_t lds8 = (size_t)32 * (VP.bm_words + 4 * VP.E) * 8;  // 4 wavefronts x 8 envs per workgroup
  if (b->step_impl == 8 && lds8 <= 64 * 1024) {
    dim3 g((unsigned)((VP.B + 31) / 32)), blk(256);
#define CALLW(WW) hipLaunchKernelGGL((k_step8<EE, WW>), g, blk, lds8, VS, VP, auto_reset, want_info)
#define PER_ENV(E_) { constexpr int EE = E_; ORL_FOR_W(CALLW) }
    ORL_FOR_ENV(PER_ENV)
#undef PER_ENV
#undef CALLW
    ORL_TK("k_step8");
    if (VP.obs_dim) launch_obs(b);  // DeepRMSA observation of the new pending service
    return;
  }
  dim3 g((unsigned)VP.B), blk(64);
  // stage the pending release times through LDS when the per-env window stays small enough for 5 waves/SIMD
  const size_t ev_bytes = (size_t)VP.ev_cap * 8;
  const bool evl = (VP.lds_bytes + ev_bytes) <= 8 * 1024;
  size_t lds = VP.lds_bytes + (evl ? ev_bytes : 0);
#define CALLW(WW) \
  do { if (evl) hipLaunchKernelGGL((k_step<EE, WW, true>), g, blk, lds, VS, VP, auto_reset, want_info, fused_policy); \
       else hipLaunchKernelGGL((k_step<EE, WW, false>), g, blk, lds, VS, VP, auto_reset, want_info, fused_policy); } while (0)
#define PER_ENV(E_) { constexpr int EE = E_; ORL_FOR_W(CALLW) }
  ORL_FOR_ENV(PER_ENV)
#undef PER_ENV
#undef CALLW
  ORL_TK("k_step");
}
static void launch_obs(orl_batch* b, int with_terminal) {
  const DevParams& VP = b->view ? *b->view : b->P;
  hipStream_t VS = b->view ? b->view_stream : b->stream;
  dim3 g((unsigned)VP.B), blk(64);
  size_t lds = VP.lds_bytes;
  const bool g8form = VP.K <= 8 && VP.J <= 8;
  dim3 g8((unsigned)((VP.B + 31) / 32)), blk8(256);
#define CALLW(WW) do { if (g8form) hipLaunchKernelGGL((k_obs8<WW>), g8, blk8, 0, VS, VP, with_terminal); \
                       else hipLaunchKernelGGL((k_obs<ENV_DEEPRMSA, WW>), g, blk, lds, VS, VP, with_terminal); } while (0)
  ORL_FOR_W(CALLW)
#undef CALLW
  ORL_TK("k_obs");
}

template <typename T> static int dalloc(orl_batch* b, T** p, size_t n) {
  HIPCHK(hipMalloc((void**)p, n * sizeof(T) + 64));
  b->allocs.push_back(*p);
  return 0;
}

static int batch_create_impl(const orl_env_config* c, const orl_topology* t, int64_t n_envs, const uint32_t* mt_state,
                             const int64_t* seeds, orl_batch** out) {
  if (!c || !t || !out || (!mt_state && !seeds) || n_envs < 1) return fail(ORL_E_INVALID, "null/invalid argument");
  if (c->env_type < 0 || c->env_type > 3) return fail(ORL_E_INVALID, "unknown env_type %d", c->env_type);
  const int S = c->num_spectrum_resources, C = c->num_spatial_resources;
  if (S < 2 || S > 512) return fail(ORL_E_INVALID, "num_spectrum_resources must be in [2, 512]");
  if (C < 1 || C > 31 || (c->env_type != ORL_ENV_RMCSA && C != 1)) return fail(ORL_E_INVALID, "bad num_spatial_resources");
  if (c->env_type == ORL_ENV_DEEPRMSA && (c->j < 1 || c->j > 8)) return fail(ORL_E_INVALID, "j must be in [1, 8]");
  if (c->n_bit_rates < 1 || c->n_bit_rates > 4096) return fail(ORL_E_INVALID, "bad n_bit_rates");
  if (!c->cum_src || !c->cum_dst) return fail(ORL_E_INVALID, "node probability tables missing");
  if (c->env_type != ORL_ENV_RWA && (!c->n_slots || !c->bit_rates)) return fail(ORL_E_INVALID, "bit-rate tables missing");
  if (c->env_type == ORL_ENV_RMCSA && (!c->lmax_snr || !c->lmax_xt)) return fail(ORL_E_INVALID, "RMCSA reach tables missing");
  if (c->bit_rate_mode == 1 && !c->cum_bit_rate) return fail(ORL_E_INVALID, "cum_bit_rate missing");
  if (c->env_type != ORL_ENV_RWA) {
    for (int i = 0; i < c->n_bit_rates * t->M; i++)
      if (c->n_slots[i] < 1 || c->n_slots[i] > 64) return fail(ORL_E_INVALID, "n_slots entries must be in [1, 64]");
    for (int i = 0; i < c->n_bit_rates; i++)
      if (c->bit_rates[i] < 0 || c->bit_rates[i] > 32767) return fail(ORL_E_INVALID, "bit rates must be < 32768");
  }
  HIPCHK(hipSetDevice(t->device));
  orl_batch* b = new orl_batch();
  memset(&b->P, 0, sizeof b->P);
  b->device = t->device;
  b->d_totals = nullptr;
  b->view = nullptr;
  b->view_stream = nullptr;
  {
    // Measured on MI355X (tools/compare_impls.sh, env-steps/s of the device loop; two-kernel pipeline / four-kernel split /
    // one wavefront per env): cfg2 65 536 envs 5.3e8 / 4.4e8 / 2.5e8; cfg5 32 768: 2.6e8 / 2.1e8 / 1.2e8; RMCSA 16 384:
    // 2.0e8 / 1.6e8 / 1.0e8.  The pipeline wins once the batch fills the chip; the heavier the env, the earlier:
    // RMSA NSFNET 12 288 envs 1.74e8 vs 1.37e8, 10 240: 1.53e8 vs 1.62e8, 8 192: 1.28e8 vs 1.39e8; RWA 8 192: 1.78e8 vs
    // 1.42e8; Germany50 8 192: 0.96e8 vs 0.89e8; RMCSA 4 096: 0.70e8 vs 0.59e8.  ORL_STEP_IMPL = 64 | 8 | 1 | 2 overrides.
    const char* impl = getenv("ORL_STEP_IMPL");
    int64_t from = 12288;
    if (c->env_type == ORL_ENV_RWA || t->E >= 64) from = 8192;
    if (c->env_type == ORL_ENV_RMCSA) from = 4096;
    // The persistent kernel (k_persist) wins at every batch size where it applies (slot scan with 8 lanes per env):
    // cfg2 64 envs 2.6e6 vs 2.1e6; 4 096: 1.6e8 vs 7.6e7; 32 768: 6.3e8 vs 4.0e8; RWA 4 096: 2.4e8 vs 8.3e7.
    const bool persist_ok = t->K <= 8;
    b->step_impl = impl ? atoi(impl) : ((persist_ok || n_envs >= from) ? 2 : 64);
    // Host-driven step() of small batches: two launches of the per-env kernel beat the six of the split form (1 env: 18 500 vs
    // 11 200 steps/s; 4 096 envs: 3.3e7 vs 2.7e7 env-steps/s).  Every kernel leaves the state in the common layout (the
    // 8-lane kernels' caches are marked unknown by the per-env kernel), so the forms can alternate on one batch.
    b->host_wave64 = (!impl && n_envs < from) ? 1 : 0;
    if (b->step_impl != 8 && b->step_impl != 1 && b->step_impl != 2) b->step_impl = 64;
  }
  DevParams& P = b->P;
  P.env_type = c->env_type;
  P.N = t->N; P.E = t->E; P.K = t->K; P.H = t->H; P.M = t->M;
  P.S = S; P.C = C;
  b->wt = S <= 64 ? 1 : (S <= 128 ? 2 : (S <= 320 ? 5 : 8));
  P.W = b->wt;
  P.episode_length = c->episode_length;
  P.allow_rejection = c->allow_rejection ? 1 : 0;
  P.J = c->env_type == ORL_ENV_DEEPRMSA ? c->j : 1;
  P.bit_rate_mode = c->bit_rate_mode;
  P.br_lo = c->bit_rate_lo;
  P.n_br = c->n_bit_rates;
  P.rand_n = c->bit_rate_hi + 1 - c->bit_rate_lo;
  P.rand_bits = 0;
  for (int v = P.rand_n; v > 0; v >>= 1) P.rand_bits++;
  if (c->bit_rate_mode == 0 && c->env_type != ORL_ENV_RWA && (P.rand_n < 1 || P.rand_n != c->n_bit_rates))
    return fail(ORL_E_INVALID, "continuous mode needs n_bit_rates == hi - lo + 1");
  P.lambda_a = c->lambda_arrival;
  P.lambda_h = c->lambda_holding;
  if (!(P.lambda_a > 0) || !(P.lambda_h > 0)) return fail(ORL_E_INVALID, "rates must be positive");
  P.B = n_envs;
  int cap = c->event_capacity;
  if (cap <= 0) {
    double load = P.lambda_a / P.lambda_h;
    cap = (int)(load + 10.0 * sqrt(load) + 64.0);
  }
  P.ev_cap = (cap + 63) / 64 * 64;
  if ((b->step_impl == 1 || b->step_impl == 2) && P.ev_cap > 2048) b->step_impl = 64;  // the split pipeline indexes release slots with 8 + 3 bits
  int words = C * P.E * b->wt;
  P.bm_words = (words + 1) & ~1;
  int rej = P.allow_rejection;
  if (c->env_type == ORL_ENV_RWA) P.n_info = 2 + (P.K + rej) + (S + rej);
  else if (c->env_type == ORL_ENV_RMCSA) P.n_info = 4;
  else P.n_info = 8 + (c->bit_rate_mode == 1 ? c->n_bit_rates + 1 : 0);
  P.obs_dim = c->env_type == ORL_ENV_DEEPRMSA ? 1 + 2 * P.N + (2 * P.J + 3) * P.K : 0;
  P.cs_words = (4 * C + 15) & ~15;  // sums and their release part; whole 64-byte lines per env
  P.lds_bytes = ((P.bm_words + 4 * P.E + P.E + P.obs_dim) * 8 + P.cs_words * 4 + 15) & ~15;
  if (P.lds_bytes < 624 * 4) P.lds_bytes = 624 * 4;  // k_init_mt stages the MT state in the same window
  if (P.lds_bytes > 64 * 1024) { delete b; return fail(ORL_E_INVALID, "per-env LDS window too large (%d B)", P.lds_bytes); }

  P.n_paths = t->n_paths;
  P.path_length = t->path_length; P.edge_iter_order = t->edge_iter_order; P.link_pos = t->link_pos;
  int rc = 0;
  {
    double* p; int* q; unsigned char* u;
    rc |= upload_conv(&p, c->cum_src, (size_t)P.N, &b->allocs); P.cum_src = p;
    rc |= upload_conv(&p, c->cum_dst, (size_t)P.N * P.N, &b->allocs); P.cum_dst = p;
    if (c->bit_rates) { rc |= upload_conv(&q, c->bit_rates, (size_t)P.n_br, &b->allocs); P.bit_rates = q; }
    if (c->cum_bit_rate) { rc |= upload_conv(&p, c->cum_bit_rate, (size_t)P.n_br, &b->allocs); P.cum_br = p; }
    if (c->n_slots) { rc |= upload_conv(&u, c->n_slots, (size_t)P.n_br * P.M, &b->allocs); P.nslots = u; }
    if (c->lmax_snr) { rc |= upload_conv(&p, c->lmax_snr, (size_t)P.n_br * P.M, &b->allocs); P.lmax_snr = p; }
    if (c->lmax_xt) { rc |= upload_conv(&p, c->lmax_xt, (size_t)P.M, &b->allocs); P.lmax_xt = p; }
  }
  {
    // derived shared tables: 32-B path records and slots-per-path-per-bit-rate
    size_t npk = (size_t)P.N * P.N * P.K;
    std::vector<unsigned char> rec(npk * 32, 0), nsp(npk * (size_t)P.n_br, 1);
    for (size_t i = 0; i < npk; i++) {
      rec[i * 32 + 0] = (unsigned char)t->h_hops[i];
      rec[i * 32 + 1] = (unsigned char)t->h_mod[i];
      for (int h = 0; h < t->h_hops[i]; h++) rec[i * 32 + 2 + h] = (unsigned char)t->h_links[i * P.H + h];
      if (c->n_slots)
        for (int r = 0; r < P.n_br; r++) nsp[i * P.n_br + r] = c->n_slots[(size_t)r * P.M + t->h_mod[i]];
    }
    unsigned char* u;
    rc |= upload_conv(&u, rec.data(), rec.size(), &b->allocs); P.path_rec = u;
    rc |= upload_conv(&u, nsp.data(), nsp.size(), &b->allocs); P.nslots_path = u;
  }
  size_t B = (size_t)n_envs;
  rc |= dalloc(b, &P.svc_desc, B);
  {
    // one queue region per control wavefront (8 envs), sized for the most items its envs can produce in a step:
    // a provision touches <= H links, the releases of a step <= E links (one item per link)
    const size_t waves = ((B + 31) / 32 + 16) * 4;  // +16 workgroups: sub-batch views start at multiples of 32 envs
    P.q_wave = 8 * (P.H > P.E ? P.H : P.E);
    P.item_masks = ORL_IMASKS;
    // test knob: a smaller limit sends far more env-steps through the tally pass and the serial tail
    if (const char* mv = getenv("ORL_ITEM_MASKS")) { int v = atoi(mv); if (v >= 1 && v <= ORL_IMASKS) P.item_masks = v; }
    P.q_cap = (i64)waves * P.q_wave;
    rc |= dalloc(b, &P.q_a, (size_t)P.q_cap * 2);  // 32-byte items
    rc |= dalloc(b, &P.q_b, (size_t)P.q_cap * 2);
    rc |= dalloc(b, &P.q_cnt_a, waves);
    rc |= dalloc(b, &P.q_cnt_b, waves);
    rc |= dalloc(b, &P.q_stat, 16);
    // deferred-env lists: [B + 16] for the whole batch, then one region per sub-batch; two buffers (steps alternate)
    P.q_def_stride = (i64)(2 * B + 16 * 80);
    rc |= dalloc(b, &P.q_def, 2 * (size_t)P.q_def_stride);
    P.pipeline2 = (b->step_impl == 2) ? 1 : 0;
    if (b->step_impl == 2) {
      rc |= dalloc(b, &b->d_wg_step, (B + 7) / 8 + 16);
      rc |= dalloc(b, &b->d_unfinished, 16);
      const bool wide_policy = t->K > 8;
      b->persist = !wide_policy && !(c->env_type == ORL_ENV_DEEPRMSA && c->j > 8);
      if (const char* pv = getenv("ORL_PERSIST")) b->persist = b->persist && atoi(pv) != 0;
    }
    if (!rc) hipMemset(P.q_def, 0, 2 * (size_t)P.q_def_stride * sizeof(u32));
    rc |= dalloc(b, &P.soon_t, B * ORL_SOON);
    rc |= dalloc(b, &P.soon_i, B * ORL_SOON);
    if (!rc) hipMemset(P.q_stat, 0, 16 * sizeof(u32));

  }
  rc |= dalloc(b, &P.bitmap, B * P.bm_words);
  rc |= dalloc(b, &P.ev_time, B * P.ev_cap);
  rc |= dalloc(b, &P.ev_info, B * P.ev_cap);
  rc |= dalloc(b, &P.mt, B * 624);
  rc |= dalloc(b, &P.lstat, B * 4 * P.E);
  rc |= dalloc(b, &P.scal, B * ORL_SCAL_WORDS);
  rc |= dalloc(b, &P.core_sums, B * P.cs_words);
  if (c->bit_rate_mode == 1 && c->env_type != ORL_ENV_RWA) rc |= dalloc(b, &P.br_hist, B * 2 * P.n_br);
  if (c->env_type == ORL_ENV_RWA) rc |= dalloc(b, &P.act_hist, B * ((P.K + 1) + (S + 1)));
  rc |= dalloc(b, &P.actions, B * 4);
  rc |= dalloc(b, &P.reward, B);
  rc |= dalloc(b, &P.done, B);
  rc |= dalloc(b, &P.info, B * P.n_info);
  if (P.obs_dim) { rc |= dalloc(b, &P.obs, B * P.obs_dim); rc |= dalloc(b, &P.term_obs, B * P.obs_dim); }
  rc |= dalloc(b, &b->d_totals, 2);
  if (rc) { orl_batch_destroy(b); return ORL_E_HIP; }
  HIPCHK(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
  HIPCHK(hipEventCreate(&b->ev0));
  HIPCHK(hipEventCreate(&b->ev1));
  {
    // Env sub-ranges on separate streams: one range's kernel fills the other's tail.  Measured on MI355X, cfg2: pipeline at
    // 65 536 envs 1 stream 4.2e8, 2 streams 5.3e8; at 8 192 envs no gain (RWA: 1.78e8 -> 1.42e8), so from 16 384;
    // per-env kernel at 8 192 envs 1.12e8 -> 1.39e8 (10 240: 1.27e8 -> 1.62e8), so from 8 192.
    // Three sub-ranges (the batch's own stream + two more) beat two for the heavier env families: cfg2 65 536 envs
    // 5.3e8 -> 5.7e8, cfg5 32 768: 2.69e8 -> 2.80e8, RMCSA 16 384: 2.12e8 -> 2.20e8; RWA / DeepRMSA 32 768: 4.75e8 -> 4.57e8.
    const int many = (c->env_type == ORL_ENV_RWA || c->env_type == ORL_ENV_DEEPRMSA) ? 2 : 3;
    int n_streams = (n_envs >= (b->step_impl == 64 || b->step_impl == 8 ? 8192 : 16384)) ? many : 1;
    if (const char* sv = getenv("ORL_STREAMS")) { int v = atoi(sv); if (v >= 1 && v <= 16) n_streams = v; }
    int n_sub = n_streams;  // ORL_SUBS > ORL_STREAMS: sub-batch k runs on stream k % n_streams
    if (const char* sv = getenv("ORL_SUBS")) { int v = atoi(sv); if (v >= n_streams && v <= 64) n_sub = v; }
    // The batch's own stream carries the first sub-batch: the runtime maps streams onto 4 hardware queues by default, and
    // with one more stream than queues the sub-batches serialise (3 extra streams: 3.5e8 env-steps/s; 3 streams in all: 5.7e8).
    b->owned_streams.push_back(b->stream);
    for (int i = 1; i < n_streams; i++) {
      hipStream_t st;
      HIPCHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
      b->owned_streams.push_back(st);
    }
    i64 per = ((n_envs + n_sub - 1) / n_sub + 31) / 32 * 32;  // slot-scan workgroups cover 32 consecutive envs
    for (i64 lo = 0; lo < n_envs; lo += per) {
      DevParams q = P;
      i64 cnt = n_envs - lo < per ? n_envs - lo : per;
      q.B = cnt;
      q.bitmap += lo * P.bm_words; q.ev_time += lo * P.ev_cap; q.ev_info += lo * P.ev_cap; q.mt += lo * 624;
      q.lstat += lo * 4 * P.E; q.scal += lo * ORL_SCAL_WORDS; q.svc_desc += lo; q.core_sums += lo * P.cs_words;
      q.q_a += (lo / 8) * P.q_wave * 2; q.q_b += (lo / 8) * P.q_wave * 2; q.q_cnt_a += lo / 8; q.q_cnt_b += lo / 8; q.soon_t += lo * ORL_SOON; q.soon_i += lo * ORL_SOON;
      if (q.br_hist) q.br_hist += lo * 2 * P.n_br;
      if (q.act_hist) q.act_hist += lo * ((P.K + 1) + (S + 1));
      q.actions += lo * 4; q.reward += lo; q.done += lo; q.info += lo * P.n_info;
      q.q_def = P.q_def + (B + 16) + lo + 16 * b->subs.size();
      if (q.obs) { q.obs += lo * P.obs_dim; q.term_obs += lo * P.obs_dim; }
      b->sub_streams.push_back(b->owned_streams[b->subs.size() % (size_t)n_streams]);
      b->subs.push_back(q);
    }
  }
  // MT state upload + conversion, then the constructor's full reset
  u32* raw = nullptr;
  HIPCHK(hipMalloc((void**)&raw, B * 625 * sizeof(u32)));
  long long* dseeds = nullptr;
  if (mt_state) {
    HIPCHK(hipMemcpyAsync(raw, mt_state, B * 625 * sizeof(u32), hipMemcpyHostToDevice, b->stream));
  } else {
    HIPCHK(hipMalloc((void**)&dseeds, B * sizeof(long long)));
    HIPCHK(hipMemcpyAsync(dseeds, seeds, B * sizeof(long long), hipMemcpyHostToDevice, b->stream));
    hipLaunchKernelGGL(k_seed_mt, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, b->stream, dseeds, (i64)B, raw);
  }
  HIPCHK(hipMemsetAsync(P.actions, 0, B * 4 * sizeof(int), b->stream));
  hipLaunchKernelGGL(k_init_mt, dim3((unsigned)B), dim3(64), 624 * 4, b->stream, P, raw);
  launch_reset(b, 1, nullptr);
  if (P.obs_dim) launch_obs(b, 0);
  HIPCHK(hipStreamSynchronize(b->stream));
  HIPCHK(hipGetLastError());
  hipFree(raw);
  if (dseeds) hipFree(dseeds);
  *out = b;
  return ORL_OK;
}

extern "C" int orl_batch_create(const orl_env_config* c, const orl_topology* t, int64_t n_envs, const uint32_t* mt_state,
                                orl_batch** out) {
  if (!mt_state) return fail(ORL_E_INVALID, "mt_state is null");
  return batch_create_impl(c, t, n_envs, mt_state, nullptr, out);
}
extern "C" int orl_batch_create_seeded(const orl_env_config* c, const orl_topology* t, int64_t n_envs, const int64_t* seeds,
                                       orl_batch** out) {
  if (!seeds) return fail(ORL_E_INVALID, "seeds is null");
  return batch_create_impl(c, t, n_envs, nullptr, seeds, out);
}

extern "C" void orl_batch_destroy(orl_batch* b) {
  if (!b) return;
  hipSetDevice(b->device);
  if (b->stream) { hipStreamSynchronize(b->stream); hipStreamDestroy(b->stream); }
  for (hipStream_t st : b->owned_streams)
    if (st != b->stream) { hipStreamSynchronize(st); hipStreamDestroy(st); }
  if (b->ev0) hipEventDestroy(b->ev0);
  if (b->ev1) hipEventDestroy(b->ev1);
  for (void* p : b->allocs) hipFree(p);
  delete b;
}

extern "C" int orl_batch_info_dim(const orl_batch* b) { return b ? b->P.n_info : 0; }
extern "C" int orl_batch_obs_dim(const orl_batch* b) { return b ? b->P.obs_dim : 0; }

extern "C" int orl_batch_sync(orl_batch* b) {
  if (!b) return fail(ORL_E_INVALID, "null batch");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipStreamSynchronize(b->stream));
  HIPCHK(hipGetLastError());
  return ORL_OK;
}

extern "C" int orl_batch_reset(orl_batch* b, int full, const uint8_t* env_mask) {
  if (!b) return fail(ORL_E_INVALID, "null batch");
  HIPCHK(hipSetDevice(b->device));
  unsigned char* dmask = nullptr;
  if (env_mask) {
    HIPCHK(hipMalloc((void**)&dmask, (size_t)b->P.B));
    HIPCHK(hipMemcpyAsync(dmask, env_mask, (size_t)b->P.B, hipMemcpyHostToDevice, b->stream));
  }
  launch_reset(b, full ? 1 : 0, dmask);
  if (b->P.obs_dim) launch_obs(b, 0);
  HIPCHK(hipStreamSynchronize(b->stream));
  HIPCHK(hipGetLastError());
  if (dmask) hipFree(dmask);
  return ORL_OK;
}

extern "C" int orl_batch_policy(orl_batch* b, int policy_id, int32_t* actions_out) {
  if (!b) return fail(ORL_E_INVALID, "null batch");
  if (policy_id < 0 || policy_id > 3) return fail(ORL_E_INVALID, "unknown policy %d", policy_id);
  HIPCHK(hipSetDevice(b->device));
  launch_policy(b, policy_id);
  if (actions_out) {
    HIPCHK(hipMemcpyAsync(actions_out, b->P.actions, (size_t)b->P.B * 4 * sizeof(int), hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    HIPCHK(hipGetLastError());
  }
  return ORL_OK;
}

extern "C" int orl_batch_step(orl_batch* b, const int32_t* actions, int auto_reset, double* obs_out, double* reward_out,
                              uint8_t* done_out, double* info_out) {
  if (!b) return fail(ORL_E_INVALID, "null batch");
  HIPCHK(hipSetDevice(b->device));
  const size_t B = (size_t)b->P.B;
  if (actions) HIPCHK(hipMemcpyAsync(b->P.actions, actions, B * 4 * sizeof(int), hipMemcpyHostToDevice, b->stream));
  launch_step(b, auto_reset ? 1 : 0, 1);
  bool any = false;
  if (reward_out) { HIPCHK(hipMemcpyAsync(reward_out, b->P.reward, B * sizeof(double), hipMemcpyDeviceToHost, b->stream)); any = true; }
  if (done_out) { HIPCHK(hipMemcpyAsync(done_out, b->P.done, B, hipMemcpyDeviceToHost, b->stream)); any = true; }
  if (info_out) { HIPCHK(hipMemcpyAsync(info_out, b->P.info, B * b->P.n_info * sizeof(double), hipMemcpyDeviceToHost, b->stream)); any = true; }
  if (obs_out && b->P.obs_dim) { HIPCHK(hipMemcpyAsync(obs_out, b->P.obs, B * b->P.obs_dim * sizeof(double), hipMemcpyDeviceToHost, b->stream)); any = true; }
  if (any || actions) {
    HIPCHK(hipStreamSynchronize(b->stream));
    HIPCHK(hipGetLastError());
  }
  return ORL_OK;
}

extern "C" int orl_host_alloc(size_t bytes, void** out) {
  if (!out || bytes == 0) return fail(ORL_E_INVALID, "bad argument");
  HIPCHK(hipHostMalloc(out, bytes, hipHostMallocDefault));
  memset(*out, 0, bytes);
  return ORL_OK;
}
extern "C" int orl_host_free(void* p) {
  if (p) HIPCHK(hipHostFree(p));
  return ORL_OK;
}

extern "C" int orl_batch_device_buffer(orl_batch* b, int which, void** device_ptr, int64_t* n_elements) {
  if (!b || !device_ptr || !n_elements) return fail(ORL_E_INVALID, "null argument");
  const int64_t B = b->P.B;
  switch (which) {
    case ORL_BUF_ACTIONS: *device_ptr = b->P.actions; *n_elements = B * 4; break;
    case ORL_BUF_REWARD: *device_ptr = b->P.reward; *n_elements = B; break;
    case ORL_BUF_DONE: *device_ptr = b->P.done; *n_elements = B; break;
    case ORL_BUF_INFO: *device_ptr = b->P.info; *n_elements = B * b->P.n_info; break;
    case ORL_BUF_OBS: *device_ptr = b->P.obs; *n_elements = B * b->P.obs_dim; break;
    case ORL_BUF_TERM_OBS: *device_ptr = b->P.term_obs; *n_elements = B * b->P.obs_dim; break;
    default: return fail(ORL_E_INVALID, "unknown buffer %d", which);
  }
  return ORL_OK;
}

extern "C" int orl_batch_observation(orl_batch* b, double* obs_out) {
  if (!b || !obs_out) return fail(ORL_E_INVALID, "null argument");
  if (!b->P.obs_dim) return fail(ORL_E_INVALID, "this env family has no array observation");
  HIPCHK(hipSetDevice(b->device));
  launch_obs(b, 0);
  HIPCHK(hipMemcpyAsync(obs_out, b->P.obs, (size_t)b->P.B * b->P.obs_dim * sizeof(double), hipMemcpyDeviceToHost, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  HIPCHK(hipGetLastError());
  return ORL_OK;
}

extern "C" int orl_batch_run(orl_batch* b, int policy_id, int64_t n_steps, int time_kernels, orl_run_stats* stats) {
  if (!b || n_steps < 0) return fail(ORL_E_INVALID, "bad argument");
  if (policy_id < 0 || policy_id > 3) return fail(ORL_E_INVALID, "unknown policy %d", policy_id);
  HIPCHK(hipSetDevice(b->device));
  std::vector<hipEvent_t> evs;
  TkRec tk;
  if (time_kernels == 2) {
    evs.resize((size_t)n_steps * 3);
    for (auto& e : evs) HIPCHK(hipEventCreate(&e));
  }
  HIPCHK(hipStreamSynchronize(b->stream));
  const bool multi = !time_kernels && b->subs.size() > 1;
  HIPCHK(hipEventRecord(b->ev0, b->stream));
  if (!time_kernels && b->persist) {
    // one launch for the whole run (plus a relaunch whenever a workgroup had to leave its loop for the serial tail)
    const DevParams& VP = b->P;
    hipStream_t VS = b->stream;
    constexpr int EPW_ = 8 * ORL_PERSIST_WG;  // envs per workgroup
    dim3 gc((unsigned)((VP.B + EPW_ - 1) / EPW_)), blk(64 * ORL_PERSIST_WG), blk_tail(256);
    const size_t lds_a = (size_t)EPW_ * VP.E * sizeof(sp::SinkEntry);
    HIPCHK(hipMemsetAsync(b->d_wg_step, 0, gc.x * sizeof(int), VS));
    // The run is cut into chunks of steps: a workgroup that had to leave its loop for the serial tail resumes in the next
    // launch and is at most one chunk behind (left to one launch per run, it would finish its remaining steps alone on
    // the GPU: 3 000-step runs measured 5.6e8 env-steps/s against 6.3e8 for 100-step runs).  No host synchronisation
    // between chunks; only the last launch is checked for stragglers.
    const bool roomy = VP.env_type == ENV_RMCSA || VP.E >= 64;  // 3 waves/SIMD without spills (see k_persist3)
    int chunk = 64;
    if (const char* cv = getenv("ORL_PERSIST_CHUNK")) { int v = atoi(cv); if (v >= 1) chunk = v; }
    b->persist_launches = 0;
    for (int64_t tgt = 0; tgt < n_steps || tgt == 0;) {
      tgt = (tgt + chunk < n_steps) ? tgt + chunk : n_steps;
      for (;;) {
        HIPCHK(hipMemsetAsync(VP.q_def, 0, sizeof(u32), VS));
        HIPCHK(hipMemsetAsync(b->d_unfinished, 0, sizeof(unsigned int), VS));
#define CALLW(WW)                                                                                                     \
  do {                                                                                                                \
    if (roomy) {                                                                                                      \
      if (lds_a > 48 * 1024) hipFuncSetAttribute((const void*)k_persist3<EE, WW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a); \
      hipLaunchKernelGGL((k_persist3<EE, WW>), gc, blk, lds_a, VS, VP, policy_id, (int)tgt, b->d_wg_step, b->d_unfinished); \
    } else {                                                                                                          \
      if (lds_a > 48 * 1024) hipFuncSetAttribute((const void*)k_persist<EE, WW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a); \
      hipLaunchKernelGGL((k_persist<EE, WW>), gc, blk, lds_a, VS, VP, policy_id, (int)tgt, b->d_wg_step, b->d_unfinished); \
    }                                                                                                                 \
    hipLaunchKernelGGL((k_rel_tail<EE, WW>), dim3(1), blk_tail, 0, VS, VP, 0);                                        \
  } while (0)
#define PER_ENV(E_) { constexpr int EE = E_; ORL_FOR_W(CALLW) }
        b->persist_launches++;
        ORL_FOR_ENV(PER_ENV)
#undef PER_ENV
#undef CALLW
        if (tgt < n_steps) break;  // stragglers catch up in the next chunk's launch
        unsigned int left = 0;
        HIPCHK(hipMemcpyAsync(&left, b->d_unfinished, sizeof left, hipMemcpyDeviceToHost, VS));
        HIPCHK(hipStreamSynchronize(VS));
        if (!left) break;
      }
      if (tgt >= n_steps) break;
    }
    launch_finish2(b);
  } else if (multi) {
    // every sub-batch runs its own policy -> step -> policy -> ... chain on its own stream
    for (hipStream_t st : b->owned_streams)
      if (st != b->stream) HIPCHK(hipStreamWaitEvent(st, b->ev0, 0));
    int64_t s0 = 0;
    if (getenv("ORL_GRAPH") && b->subs.size() == b->owned_streams.size() && n_steps >= 64) {
      // experiment (off by default): 16 steps of every sub-batch captured into one hipGraph per stream
      constexpr int GS_ = 16;
      std::vector<hipGraphExec_t> ex(b->subs.size());
      bool ok = true;
      for (size_t k = 0; k < b->subs.size() && ok; k++) {
        b->view = &b->subs[k];
        b->view_stream = b->sub_streams[k];
        hipGraph_t g;
        ok = hipStreamBeginCapture(b->view_stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
        for (int i = 0; i < GS_ && ok; i++) launch_policy_step(b, policy_id);
        ok = ok && hipStreamEndCapture(b->view_stream, &g) == hipSuccess;
        ok = ok && hipGraphInstantiate(&ex[k], g, nullptr, nullptr, 0) == hipSuccess;
        if (ok) hipGraphDestroy(g);
      }
      if (!ok) return fail(ORL_E_HIP, "hipGraph capture failed");
      for (; s0 + GS_ <= n_steps; s0 += GS_)
        for (size_t k = 0; k < b->subs.size(); k++) HIPCHK(hipGraphLaunch(ex[k], b->sub_streams[k]));
      for (auto& e : ex) hipGraphExecDestroy(e);
    }
    for (int64_t s = s0; s < n_steps; s++) {
      for (size_t k = 0; k < b->subs.size(); k++) {
        b->view = &b->subs[k];
        b->view_stream = b->sub_streams[k];
        launch_policy_step(b, policy_id);
      }
    }
    if (b->step_impl == 2)
      for (size_t k = 0; k < b->subs.size(); k++) {
        b->view = &b->subs[k];
        b->view_stream = b->sub_streams[k];
        launch_finish2(b);
      }
    b->view = nullptr;
    b->view_stream = nullptr;
    for (hipStream_t st : b->owned_streams) {
      if (st == b->stream) continue;
      hipEvent_t e;
      HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      HIPCHK(hipEventRecord(e, st));
      HIPCHK(hipStreamWaitEvent(b->stream, e, 0));
      HIPCHK(hipEventDestroy(e));
    }
  } else if (time_kernels == 1) {
    // the production launches on the batch's stream, an event after every kernel (ORL_TK in the launch functions)
    for (int64_t s = 0; s < n_steps; s++) {
      hipStream_t VS = b->stream;
      b->tk = &tk;
      ORL_TK("");  // start of the step
      launch_policy_step(b, policy_id);
      b->tk = nullptr;
    }
    if (b->step_impl == 2) launch_finish2(b);
  } else {
    if (!time_kernels) {
      for (int64_t s = 0; s < n_steps; s++) launch_policy_step(b, policy_id);
      if (b->step_impl == 2) launch_finish2(b);
    }
    for (int64_t s = 0; s < n_steps; s++) {
      if (!time_kernels) break;
      HIPCHK(hipEventRecord(evs[3 * s], b->stream));
      launch_policy(b, policy_id);
      HIPCHK(hipEventRecord(evs[3 * s + 1], b->stream));
      launch_step(b, 1, 0);
      HIPCHK(hipEventRecord(evs[3 * s + 2], b->stream));
    }
  }
  HIPCHK(hipEventRecord(b->ev1, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  HIPCHK(hipGetLastError());
  if (stats) {
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, b->ev0, b->ev1));
    memset(stats, 0, sizeof *stats);
    stats->ms_total = ms;
    stats->launches = 2 * n_steps;
    if (!time_kernels && b->persist && n_steps > 0) {  // the whole run was (re)launches of one kernel
      stats->n_kernels = 1;
      stats->launches = b->persist_launches > 0 ? b->persist_launches : 1;  // chunks of ORL_PERSIST_CHUNK steps
      stats->ms_kernel[0] = ms / (double)stats->launches;                   // average duration of one launch
      snprintf(stats->kernel_name[0], sizeof stats->kernel_name[0], "k_persist");
    }
    if (time_kernels == 2 && n_steps > 0) {
      double sp = 0, ss = 0;
      for (int64_t s = 0; s < n_steps; s++) {
        float a = 0, c2 = 0;
        HIPCHK(hipEventElapsedTime(&a, evs[3 * s], evs[3 * s + 1]));
        HIPCHK(hipEventElapsedTime(&c2, evs[3 * s + 1], evs[3 * s + 2]));
        sp += a; ss += c2;
      }
      stats->ms_policy = sp / (double)n_steps;
      stats->ms_step = ss / (double)n_steps;
    }
    if (time_kernels == 1 && n_steps > 0) {
      const size_t per = tk.ev.size() / (size_t)n_steps;  // 1 start mark + one event per kernel
      const int nk = (int)per - 1 < ORL_MAX_STEP_KERNELS ? (int)per - 1 : ORL_MAX_STEP_KERNELS;
      stats->n_kernels = nk;
      stats->launches = (int64_t)(per - 1) * n_steps;
      for (int k = 0; k < nk; k++) {
        double sum = 0;
        for (int64_t s = 0; s < n_steps; s++) {
          float a = 0;
          HIPCHK(hipEventElapsedTime(&a, tk.ev[(size_t)s * per + k], tk.ev[(size_t)s * per + k + 1]));
          sum += a;
        }
        stats->ms_kernel[k] = sum / (double)n_steps;
        snprintf(stats->kernel_name[k], sizeof stats->kernel_name[k], "%s", tk.name[k + 1]);
      }
    }
  }
  for (auto& e : evs) hipEventDestroy(e);
  for (auto& e : tk.ev) hipEventDestroy(e);
  return ORL_OK;
}

// ---- read-back ----------------------------------------------------------------------------------
static int fetch_scal(orl_batch* b, std::vector<u64>& host) {
  host.resize((size_t)b->P.B * ORL_SCAL_WORDS);
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipStreamSynchronize(b->stream));
  HIPCHK(hipMemcpy(host.data(), b->P.scal, host.size() * sizeof(u64), hipMemcpyDeviceToHost));
  return 0;
}
static double as_f64(u64 v) { double d; memcpy(&d, &v, 8); return d; }

extern "C" int orl_batch_get_counters(orl_batch* b, int64_t* out) {
  if (!b || !out) return fail(ORL_E_INVALID, "null argument");
  std::vector<u64> h;
  if (fetch_scal(b, h)) return ORL_E_HIP;
  for (i64 i = 0; i < b->P.B; i++) {
    const u64* s = &h[(size_t)i * ORL_SCAL_WORDS];
    int64_t* o = out + i * ORL_N_COUNTERS;
    o[0] = (int64_t)s[SC_SP]; o[1] = (int64_t)s[SC_SA]; o[2] = (int64_t)s[SC_ESP]; o[3] = (int64_t)s[SC_ESA];
    o[4] = (int64_t)s[SC_BRQ]; o[5] = (int64_t)s[SC_BRP]; o[6] = (int64_t)s[SC_EBRQ]; o[7] = (int64_t)s[SC_EBRP];
  }
  return ORL_OK;
}
extern "C" int orl_batch_get_services(orl_batch* b, double* out) {
  if (!b || !out) return fail(ORL_E_INVALID, "null argument");
  std::vector<u64> h;
  if (fetch_scal(b, h)) return ORL_E_HIP;
  for (i64 i = 0; i < b->P.B; i++) {
    const u64* s = &h[(size_t)i * ORL_SCAL_WORDS];
    double* o = out + i * ORL_N_SERVICE;
    o[0] = as_f64(s[SC_AT]); o[1] = as_f64(s[SC_HT]);
    o[2] = (double)(int)(u32)s[SC_SRC_DST]; o[3] = (double)(int)(s[SC_SRC_DST] >> 32);
    o[4] = (double)(int)(u32)s[SC_BR_IDX]; o[5] = (double)(int)(u32)s[SC_ID_MTPOS];
  }
  return ORL_OK;
}
extern "C" int orl_batch_get_active(orl_batch* b, int32_t* out) {
  if (!b || !out) return fail(ORL_E_INVALID, "null argument");
  std::vector<u64> h;
  if (fetch_scal(b, h)) return ORL_E_HIP;
  for (i64 i = 0; i < b->P.B; i++) out[i] = (int32_t)(h[(size_t)i * ORL_SCAL_WORDS + SC_EV] >> 32);
  return ORL_OK;
}
extern "C" int orl_batch_get_flags(orl_batch* b, int32_t* out) {
  if (!b || !out) return fail(ORL_E_INVALID, "null argument");
  std::vector<u64> h;
  if (fetch_scal(b, h)) return ORL_E_HIP;
  for (i64 i = 0; i < b->P.B; i++) out[i] = (int32_t)(h[(size_t)i * ORL_SCAL_WORDS + SC_FLAGS] >> 32);
  return ORL_OK;
}
extern "C" int orl_batch_get_slots(orl_batch* b, int64_t env, uint8_t* out) {
  if (!b || !out || env < 0 || env >= b->P.B) return fail(ORL_E_INVALID, "bad argument");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipStreamSynchronize(b->stream));
  std::vector<u64> h((size_t)b->P.bm_words);
  HIPCHK(hipMemcpy(h.data(), b->P.bitmap + env * b->P.bm_words, h.size() * 8, hipMemcpyDeviceToHost));
  const int W = b->wt, S = b->P.S, rows = b->P.C * b->P.E;
  for (int r = 0; r < rows; r++)
    for (int s = 0; s < S; s++) out[(size_t)r * S + s] = (uint8_t)((h[(size_t)r * W + (s >> 6)] >> (s & 63)) & 1ull);
  return ORL_OK;
}
extern "C" int orl_batch_get_link_stats(orl_batch* b, int64_t env, double* out) {
  if (!b || !out || env < 0 || env >= b->P.B) return fail(ORL_E_INVALID, "bad argument");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipStreamSynchronize(b->stream));
  const int E = b->P.E;
  std::vector<double> h((size_t)4 * E);  // device layout [E][4] (one 32-byte record per link) -> ABI layout [4][E]
  HIPCHK(hipMemcpy(h.data(), b->P.lstat + env * 4 * E, h.size() * 8, hipMemcpyDeviceToHost));
  for (int l = 0; l < E; l++)
    for (int k = 0; k < 4; k++) out[(size_t)k * E + l] = h[(size_t)4 * l + k];
  return ORL_OK;
}
extern "C" int orl_batch_get_net_stats(orl_batch* b, int64_t env, double* out) {
  if (!b || !out || env < 0 || env >= b->P.B) return fail(ORL_E_INVALID, "bad argument");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipStreamSynchronize(b->stream));
  u64 s[ORL_SCAL_WORDS];
  HIPCHK(hipMemcpy(s, b->P.scal + env * ORL_SCAL_WORDS, sizeof s, hipMemcpyDeviceToHost));
  out[0] = as_f64(s[SC_GTHR]); out[1] = as_f64(s[SC_GCOMP]); out[2] = as_f64(s[SC_GLAST]); out[3] = as_f64(s[SC_NOW]);
  return ORL_OK;
}
extern "C" int orl_batch_totals(orl_batch* b, int64_t* processed, int64_t* accepted) {
  if (!b) return fail(ORL_E_INVALID, "null batch");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipMemsetAsync(b->d_totals, 0, 16, b->stream));
  unsigned blocks = (unsigned)((b->P.B + 255) / 256);
  hipLaunchKernelGGL(k_totals, dim3(blocks), dim3(256), 0, b->stream, b->P, b->d_totals);
  unsigned long long h[2];
  HIPCHK(hipMemcpyAsync(h, b->d_totals, 16, hipMemcpyDeviceToHost, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  if (processed) *processed = (int64_t)h[0];
  if (accepted) *accepted = (int64_t)h[1];
  return ORL_OK;
}

/* debug: stream the slot-map array (n_envs * bm_words * 8 bytes) once; returns that byte count */
extern "C" int64_t orl_batch_debug_stream_read(orl_batch* b, int width16) {
  if (!b) return fail(ORL_E_INVALID, "null batch");
  if (hipSetDevice(b->device) != hipSuccess) return ORL_E_HIP;
  i64 n_words = b->P.B * b->P.bm_words;
  hipLaunchKernelGGL(k_calib_read, dim3(2048), dim3(256), 0, b->stream, b->P.bitmap, n_words, width16, b->d_totals);
  if (hipStreamSynchronize(b->stream) != hipSuccess) return ORL_E_HIP;
  return n_words * 8;
}

extern "C" int orl_batch_matrix_obs_dim(const orl_batch* b) { return b ? 2 * b->P.N + b->P.C * b->P.E * b->P.S : 0; }

extern "C" int orl_batch_matrix_observation(orl_batch* b, uint8_t* out) {
  if (!b || !out) return fail(ORL_E_INVALID, "null argument");
  HIPCHK(hipSetDevice(b->device));
  const size_t dim = (size_t)orl_batch_matrix_obs_dim(b), B = (size_t)b->P.B;
  unsigned char* d = nullptr;
  HIPCHK(hipMalloc((void**)&d, B * dim));
  hipLaunchKernelGGL(k_matrix_obs, dim3((unsigned)B), dim3(256), 0, b->stream, b->P, d);
  HIPCHK(hipMemcpyAsync(out, d, B * dim, hipMemcpyDeviceToHost, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  HIPCHK(hipGetLastError());
  hipFree(d);
  return ORL_OK;
}

// ---- snapshot / restore: the per-env arrays, concatenated in a fixed order -------------------------
struct Section { void* ptr; size_t bytes; };
static std::vector<Section> state_sections(orl_batch* b) {
  const DevParams& P = b->P;
  const size_t B = (size_t)P.B;
  std::vector<Section> v;
  v.push_back({P.scal, B * ORL_SCAL_WORDS * 8});
  v.push_back({P.svc_desc, B * 8});
  v.push_back({P.bitmap, B * P.bm_words * 8});
  v.push_back({P.ev_time, B * P.ev_cap * 8});
  v.push_back({P.ev_info, B * P.ev_cap * 8});
  v.push_back({P.mt, B * 624 * 4});
  v.push_back({P.lstat, B * 4 * P.E * 8});
  v.push_back({P.core_sums, B * P.cs_words * 4});
  v.push_back({P.soon_t, B * ORL_SOON * 8});
  v.push_back({P.soon_i, B * ORL_SOON * 4});
  if (P.br_hist) v.push_back({P.br_hist, B * 2 * P.n_br * 8});
  if (P.act_hist) v.push_back({P.act_hist, B * ((P.K + 1) + (P.S + 1)) * 8});
  return v;
}
extern "C" int64_t orl_batch_state_bytes(orl_batch* b) {
  if (!b) return fail(ORL_E_INVALID, "null batch");
  int64_t n = 0;
  for (auto& s : state_sections(b)) n += (int64_t)s.bytes;
  return n;
}
extern "C" int orl_batch_get_state(orl_batch* b, void* out) {
  if (!b || !out) return fail(ORL_E_INVALID, "null argument");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipStreamSynchronize(b->stream));
  unsigned char* o = (unsigned char*)out;
  for (auto& s : state_sections(b)) { HIPCHK(hipMemcpy(o, s.ptr, s.bytes, hipMemcpyDeviceToHost)); o += s.bytes; }
  return ORL_OK;
}
extern "C" int orl_batch_set_state(orl_batch* b, const void* in) {
  if (!b || !in) return fail(ORL_E_INVALID, "null argument");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipStreamSynchronize(b->stream));
  const unsigned char* o = (const unsigned char*)in;
  for (auto& s : state_sections(b)) { HIPCHK(hipMemcpy(s.ptr, o, s.bytes, hipMemcpyHostToDevice)); o += s.bytes; }
  if (b->P.obs_dim) launch_obs(b, 0);
  HIPCHK(hipStreamSynchronize(b->stream));
  return ORL_OK;
}

/* debug: number of env-steps that fell back to the serial release path of the split pipeline since creation */
extern "C" int orl_batch_debug_prof(orl_batch* b, uint64_t* out32, int reset) {
  if (!b || !out32) return fail(ORL_E_INVALID, "bad argument");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipDeviceSynchronize());
  memset(out32, 0, 32 * 8);
#if defined(ORL_TIMING) && ORL_TIMING == 3
  {
    HIPCHK(hipMemcpyFromSymbol(out32, HIP_SYMBOL(g8::g_dbg), 64 * 8));  // debug builds: the caller passes 64 words
    if (reset) { unsigned long long z[64] = {0}; HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g8::g_dbg), z, 64 * 8)); }
  }
#elif defined(ORL_TIMING)
  {
    std::vector<unsigned long long> h((size_t)ORL_PROF_WAVES * ORL_PROF_SLOTS);
    HIPCHK(hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(sp::g_prof), h.size() * 8));
    for (size_t w = 0; w < ORL_PROF_WAVES; w++)
      for (int k = 0; k < ORL_PROF_SLOTS; k++) {
        if (k == 15) out32[k] = h[w * ORL_PROF_SLOTS + k] > out32[k] ? h[w * ORL_PROF_SLOTS + k] : out32[k];  // max
        else out32[k] += h[w * ORL_PROF_SLOTS + k];
      }
    if (reset) {
      std::fill(h.begin(), h.end(), 0ull);
      HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(sp::g_prof), h.data(), h.size() * 8));
    }
  }
#endif
  return ORL_OK;
}
extern "C" int64_t orl_batch_debug_serial_count(orl_batch* b) {
  if (!b) return -1;
  if (hipSetDevice(b->device) != hipSuccess) return -1;
  hipStreamSynchronize(b->stream);
  for (hipStream_t st : b->owned_streams) hipStreamSynchronize(st);
  u32 v = 0;
  if (hipMemcpy(&v, b->P.q_stat, 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return (int64_t)v;
}
