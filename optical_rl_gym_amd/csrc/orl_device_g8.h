// orl_device_g8.h — step() with EIGHT lanes per env (eight envs per wavefront).
//
// Why: with one wavefront per env (orl_device.h, k_step) almost every instruction is wave-uniform control flow
// that keeps 1-8 of 64 lanes busy; PMC showed ~2 100 VALU + ~1 000 SALU instructions per env-step and the kernel
// bound by instruction issue, not by HBM.  Here a group of 8 lanes owns an env and lane w of the group owns
//   * 64-bit word w of every link row of that env      (W <= 8 words = 512 slots)
//   * pending-release slots i with i % 8 == w
//   * the per-link statistics of the links whose position in topology.edges() is == w (mod 8)
// so every read-modify-write of env state is done by the SAME lane that wrote it last: no LDS staging, no
// cross-lane visibility hazards, and the state is touched in place in HBM (only the rows/slots a step needs).
// The instruction stream is about as long as before but serves 8 envs.  Control flow is group-uniform; groups of
// one wavefront diverge (different release counts, hop counts), which SIMT handles with exec masks.
//
// Semantics are identical to orl_device.h (same reference line ranges); shared primitives are reused from there.
#pragma once
#include "orl_device.h"

namespace orl {
namespace g8 {

__device__ __forceinline__ u32 gballot(bool p, int lane) { return (u32)((__ballot(p) >> (lane & 56)) & 0xffull); }
__device__ __forceinline__ int gget(int v, int src, int lane) { return __shfl(v, (lane & 56) | src, 64); }
__device__ __forceinline__ u32 gget(u32 v, int src, int lane) { return (u32)__shfl((int)v, (lane & 56) | src, 64); }
__device__ __forceinline__ double gget(double v, int src, int lane) { return __shfl(v, (lane & 56) | src, 64); }
__device__ __forceinline__ u64 gget(u64 v, int src, int lane) {
  u32 lo = gget((u32)v, src, lane), hi = gget((u32)(v >> 32), src, lane);
  return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ double g8_min(double v) {
  double t;
  t = dpp_d<ORL_DPP_XOR1>(v); v = t < v ? t : v;
  t = dpp_d<ORL_DPP_XOR2>(v); v = t < v ? t : v;
  t = dpp_d<ORL_DPP_HALF_MIRROR>(v); v = t < v ? t : v;
  return v;
}

struct EnvG {
  double now, at, ht, g_thr, g_comp, g_last, next_rel, t_soon;
  i64 sp, sa, esp, esa, brq, brp, ebrq, ebrp, s_br, s_nh;
  int src, dst, bit_rate, br_idx, id, mt_pos, ev_hwm, ev_cnt, new_service, flags;
  int nfree, pop_idx;  // free-slot stack: entries known, and the top entry (the next slot a push takes), -1 if none
  i64 env;
  u64* bm;
  double* ls;
  int* cs;
  int* rs;  // two-kernel pipeline: per-core sums of what this step's releases added (nullptr otherwise)
  double* ev_time;
  u64* ev_info;
  double* soon_t;
  u32* soon_i;
  // persistent kernel: this lane's soon-list entries (gl + 8k) live in registers for the whole launch
  bool sr_on;
  bool rank_pairs;  // release_soon: rank the due releases all-pairs (the forms with registers to spare)
  double sr_t[ORL_SOON_PER_LANE];
  int sr_i[ORL_SOON_PER_LANE];
  u32* mt;
  u64* scal;
};

// `rec`: where the env's record lives (the global array, or the persistent kernel's LDS copy)
__device__ __forceinline__ void env_load(const DevParams& P, EnvG& e, i64 env, u64* rec) {
  const u64* s = rec;
  e.scal = rec;
#define F64(slot) __longlong_as_double((i64)s[slot])
  e.now = F64(SC_NOW); e.at = F64(SC_AT); e.ht = F64(SC_HT);
  e.g_thr = F64(SC_GTHR); e.g_comp = F64(SC_GCOMP); e.g_last = F64(SC_GLAST); e.next_rel = F64(SC_NEXTREL);
  e.t_soon = F64(SC_TSOON);
#undef F64
  e.sp = (i64)s[SC_SP]; e.sa = (i64)s[SC_SA]; e.esp = (i64)s[SC_ESP]; e.esa = (i64)s[SC_ESA];
  e.brq = (i64)s[SC_BRQ]; e.brp = (i64)s[SC_BRP]; e.ebrq = (i64)s[SC_EBRQ]; e.ebrp = (i64)s[SC_EBRP];
  e.s_br = (i64)s[SC_SBR]; e.s_nh = (i64)s[SC_SNH];
  u64 t;
  t = s[SC_SRC_DST]; e.src = (int)(u32)t; e.dst = (int)(t >> 32);
  t = s[SC_BR_IDX]; e.bit_rate = (int)(u32)t; e.br_idx = (int)(t >> 32);
  t = s[SC_ID_MTPOS]; e.id = (int)(u32)t; e.mt_pos = (int)(t >> 32);
  t = s[SC_EV]; e.ev_hwm = (int)(u32)t; e.ev_cnt = (int)(t >> 32);
  t = s[SC_FLAGS]; e.new_service = (int)(u32)t; e.flags = (int)(t >> 32);
  {
    // the free-slot stack travels in the record; only its top entry is kept (one push per step at most)
    const u64 f0 = s[SC_FREE0], f1 = s[SC_FREE1], f2 = s[SC_FREE2], f3 = s[SC_FREE3];
    t = s[SC_HINT]; e.nfree = (int)(u32)t;
    const int top = e.nfree - 1;
    const u64 w = (top >> 2) == 0 ? f0 : (top >> 2) == 1 ? f1 : (top >> 2) == 2 ? f2 : f3;
    e.pop_idx = (top >= 0) ? (int)((w >> (16 * (top & 3))) & 0xffffu) : -1;
  }
  e.env = env;
  e.bm = P.bitmap + env * P.bm_words;
  e.ls = P.lstat + env * 4 * P.E;
  e.cs = P.core_sums + env * P.cs_words;
  e.rs = nullptr;
  e.ev_time = P.ev_time + env * P.ev_cap;
  e.ev_info = P.ev_info + env * P.ev_cap;
  e.soon_t = P.soon_t + env * ORL_SOON;
  e.soon_i = P.soon_i + env * ORL_SOON;
  e.sr_on = false;
  e.rank_pairs = false;
  e.mt = P.mt + env * 624;
}
__device__ __forceinline__ void env_load(const DevParams& P, EnvG& e, i64 env) { env_load(P, e, env, P.scal + env * ORL_SCAL_WORDS); }

// returns the service descriptor of the pending service (what the slot scan reads); stored only when write_desc
__device__ __forceinline__ u64 env_store(const DevParams& P, const EnvG& e, int gl, bool write_desc = true) {
  const u64 np_ = (u64)(u32)P.n_paths[e.src * P.N + e.dst];
  const u64 desc = (u64)(u32)((e.src * P.N + e.dst) * P.K) | ((u64)(u32)e.br_idx << 32) | (np_ << 48);
  if (gl != 0) return desc;
  u64* s = e.scal;
#define PF(slot, x) s[slot] = (u64)__double_as_longlong(x);
  PF(SC_NOW, e.now) PF(SC_AT, e.at) PF(SC_HT, e.ht) PF(SC_GTHR, e.g_thr) PF(SC_GCOMP, e.g_comp) PF(SC_GLAST, e.g_last)
  PF(SC_NEXTREL, e.next_rel) PF(SC_TSOON, e.t_soon)
#undef PF
  s[SC_SP] = (u64)e.sp; s[SC_SA] = (u64)e.sa; s[SC_ESP] = (u64)e.esp; s[SC_ESA] = (u64)e.esa;
  s[SC_BRQ] = (u64)e.brq; s[SC_BRP] = (u64)e.brp; s[SC_EBRQ] = (u64)e.ebrq; s[SC_EBRP] = (u64)e.ebrp;
  s[SC_SBR] = (u64)e.s_br; s[SC_SNH] = (u64)e.s_nh;
  s[SC_SRC_DST] = pack2(e.src, e.dst); s[SC_BR_IDX] = pack2(e.bit_rate, e.br_idx);
  s[SC_ID_MTPOS] = pack2(e.id, e.mt_pos); s[SC_EV] = pack2(e.ev_hwm, e.ev_cnt);
  s[SC_FLAGS] = pack2(e.new_service, e.flags); s[SC_HINT] = pack2(e.nfree, 0);
  if (write_desc) P.svc_desc[e.env] = desc;
  return desc;
}

// ---- MT19937, 16-word window per refill (lane w holds window words w and w+8) ----------------------
struct RngG { u32 out0, out1, nx0, nx1; int used; int pend_pos, pend_used; };  // pend_*: a commit whose stores are still to be done

__device__ __forceinline__ void mt_one(const u32* mt, int i, u32& out, u32& nx) {
  int i0 = i >= 624 ? i - 624 : i;
  int i1 = i0 + 1 >= 624 ? i0 + 1 - 624 : i0 + 1;
  int im = i0 + 397 >= 624 ? i0 + 397 - 624 : i0 + 397;
  u32 cur = mt[i0], nxt = mt[i1], far = mt[im];
  u32 y = (cur & 0x80000000u) | (nxt & 0x7fffffffu);
  nx = far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
  u32 t = cur;
  t ^= (t >> 11);
  t ^= (t << 7) & 0x9d2c5680u;
  t ^= (t << 15) & 0xefc60000u;
  t ^= (t >> 18);
  out = t;
}
__device__ __forceinline__ void rng_fill(const EnvG& e, RngG& r, int gl) {
  mt_one(e.mt, e.mt_pos + gl, r.out0, r.nx0);
  mt_one(e.mt, e.mt_pos + 8 + gl, r.out1, r.nx1);
  r.used = 0;
  r.pend_used = 0;
}
// the commit of next_service in two halves: the stream position now (the record needs it), the stores of the regenerated
// words later, behind the loads of the release detection (rng_commit_stores)
__device__ __forceinline__ void rng_commit_pos(EnvG& e, RngG& r) {
  r.pend_pos = e.mt_pos;
  r.pend_used = r.used;
  int p = e.mt_pos + r.used;
  e.mt_pos = p >= 624 ? p - 624 : p;
  r.used = 0;
}
__device__ __forceinline__ void rng_commit_stores(EnvG& e, RngG& r, int gl) {
  if (gl < r.pend_used) { int i = r.pend_pos + gl; e.mt[i >= 624 ? i - 624 : i] = r.nx0; }
  if (gl + 8 < r.pend_used) { int i = r.pend_pos + 8 + gl; e.mt[i >= 624 ? i - 624 : i] = r.nx1; }
  r.pend_used = 0;
}
__device__ __forceinline__ void rng_commit(EnvG& e, RngG& r, int gl) {
  if (gl < r.used) { int i = e.mt_pos + gl; e.mt[i >= 624 ? i - 624 : i] = r.nx0; }
  if (gl + 8 < r.used) { int i = e.mt_pos + 8 + gl; e.mt[i >= 624 ? i - 624 : i] = r.nx1; }
  int p = e.mt_pos + r.used;
  e.mt_pos = p >= 624 ? p - 624 : p;
  r.used = 0;
}
__device__ __forceinline__ u32 rng_u32(EnvG& e, RngG& r, int lane) {
  if (r.used == 16) {
    rng_commit(e, r, lane & 7);
    rng_fill(e, r, lane & 7);
  }
  u32 a = gget(r.out0, r.used & 7, lane), b = gget(r.out1, r.used & 7, lane);
  u32 v = r.used < 8 ? a : b;
  r.used++;
  return v;
}
__device__ __forceinline__ double rng_random(EnvG& e, RngG& r, int lane) {
  u32 a = rng_u32(e, r, lane) >> 5, b = rng_u32(e, r, lane) >> 6;
  return ((double)a * 67108864.0 + (double)b) * (1.0 / 9007199254740992.0);
}
__device__ __forceinline__ double rng_expovariate(EnvG& e, RngG& r, int lane, double lambd) {
  return -orl_log(1.0 - rng_random(e, r, lane)) / lambd;
}
__device__ __forceinline__ int rng_choice(EnvG& e, RngG& r, int lane, const double* cum, int n) {
  // (the table entries are requested four per lane at a time, before the first comparison: one memory round trip per 32
  // entries — Germany50's 49 took seven, one per 8 entries, twice per step: a fifth of cfg5's wavefront-step)
  const double tot = cum[n - 1];
  int cnt = 0;
  double x = 0.0;
  for (int base = 0; base < n - 1; base += 32) {
    double c[4];
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int i = base + 8 * k + (lane & 7);
      c[k] = cum[i < n - 1 ? i : n - 2];
    }
    if (base == 0) x = rng_random(e, r, lane) * (tot + 0.0);
#pragma unroll
    for (int k = 0; k < 4; k++) {
      const int i = base + 8 * k + (lane & 7);
      cnt += (int)__popc(gballot((i < n - 1) && (c[k] <= x), lane));
    }
  }
  if (n <= 1) x = rng_random(e, r, lane);  // (a single-entry table still draws)
  return cnt;
}

// ---- slot rows: lane w owns word w -------------------------------------------------------------------
__device__ __forceinline__ double net_compactness(const DevParams& P, const EnvG& e, int core, int lane) {
  // lane 0 of the group is the only writer of cs[]: take its copy
  int occ = gget(e.cs[2 * core], 0, lane), fb = gget(e.cs[2 * core + 1], 0, lane);
  if (fb > 0) return ((double)occ / (double)e.s_nh) * ((double)P.E / (double)fb);
  return 1.0;
}

// QoSConstrainedRA: one spectrum counter per link instead of a slot map, every service takes one unit
// (qos_constrained_ra.py:381-392 is_path_free, :287-300 / :318-333 provision and release with the link's utilization average);
// the hops of the path over the 8 lanes of the group — what qos_path_free / qos_path_apply (orl_device.h) do with 64 lanes
__device__ __forceinline__ bool qos_path_free(const DevParams& P, const EnvG& e, int lane, const PathRec& rec) {
  const int hops = path_rec_byte(rec, 0);
  bool busy = false;
  for (int h = lane & 7; h < hops; h += 8) busy = busy || (i64)e.bm[path_rec_byte(rec, 2 + h)] < 1;
  return P.S >= 1 && gballot(busy, lane) == 0u;
}
__device__ __forceinline__ int qos_path_apply(const DevParams& P, EnvG& e, int lane, const PathRec& rec, bool release) {
  const int hops = path_rec_byte(rec, 0);
  for (int h = lane & 7; h < hops; h += 8) {
    const int link = path_rec_byte(rec, 2 + h);
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

template <int ENV, int W>
__device__ __forceinline__ int path_apply(const DevParams& P, EnvG& e, int lane, int pidx, int core, int s0, int n, bool release) {
  if (ENV == ENV_QOS) return qos_path_apply(P, e, lane, path_rec_load(P, pidx), release);
  const PathRec rec = path_rec_load(P, pidx);
  const int hops = path_rec_byte(rec, 0), w = lane & 7;
  const int E = P.E, S = P.S;
  int d_occ = 0, d_fb = 0;
  const u64 m = word_range(s0 - 64 * w, s0 + n - 64 * w);
  for (int h = 0; h < hops; h++) {
    const int link = path_rec_byte(rec, 2 + h);
    u64* wp = e.bm + (core * E + link) * W + (w < W ? w : 0);
    u64 a = (w < W) ? *wp : 0ull;
    RowStat before, after;
    if (ENV != ENV_RWA) row_stat<W, false>(a, w, S, before);
    a = release ? (a | m) : (a & ~m);
    if (w < W) *wp = a;
    if (ENV != ENV_RWA) {
      row_stat<W, true>(a, w, S, after);
      d_occ += after.occ - before.occ;
      d_fb += after.fb - before.fb;
    } else {
      after.free_ = g8_sum(__popcll(a));
    }
    // _update_link_stats (rmsa_env.py:464-543).  The statistics of a link belong to lane link_pos[link] % 8.
    const bool own = (w == (P.link_pos[link] & 7));
    double last_update = e.ls[4 * link + 3];
    double time_diff = e.now - last_update;
    if (e.now > 0) {
      const int free_ = after.free_;
      double cur_util = (double)(S - free_) / (double)S;
      double util = ((e.ls[4 * link] * last_update) + (cur_util * time_diff)) / e.now;
      double frag = 0.0, comp = 0.0;
      if (ENV != ENV_RWA) {
        double cur_frag = 0.0, cur_comp = 0.0;
        const int top = (S - 1) - 64 * w;
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
      if (own) {
        e.ls[4 * link] = util;
        if (ENV != ENV_RWA) { e.ls[4 * link + 1] = frag; e.ls[4 * link + 2] = comp; }
      }
    }
    if (own) e.ls[4 * link + 3] = e.now;
  }
  if (ENV != ENV_RWA) {
    int c0 = gget(e.cs[2 * core], 0, lane) + d_occ, c1 = gget(e.cs[2 * core + 1], 0, lane) + d_fb;
    if (w == 0) { e.cs[2 * core] = c0; e.cs[2 * core + 1] = c1; }
    if (release && e.rs && w == 0) { e.rs[2 * core] += d_occ; e.rs[2 * core + 1] += d_fb; }
  }
  return hops;
}

// a released slot goes onto the env's free-slot stack (dropped when the stack is full: the next rebuild finds it)
__device__ __forceinline__ void free_push(EnvG& e, int gl, int slot) {
  if (e.nfree < ORL_FREE_SLOTS) {
    if (gl == 0) ((unsigned short*)(e.scal + SC_FREE0))[e.nfree] = (unsigned short)slot;
    e.nfree++;
  }
}

#define ORL_DBG(k, v) do { } while (0)
// entry `slot` of the env's soon list (owned by lane slot % 8): registers or memory
__device__ __forceinline__ void soon_set(EnvG& e, int gl, int slot, double t, int idx) {
  if (gl != (slot & 7)) return;
  if (e.sr_on) {
#pragma unroll
    for (int k = 0; k < ORL_SOON_PER_LANE; k++)
      if ((slot >> 3) == k) { e.sr_t[k] = t; e.sr_i[k] = idx; }
  } else {
    e.soon_t[slot] = t;
    e.soon_i[slot] = (u32)idx;
  }
}
// ---- pending releases: slot i belongs to lane i % 8 -------------------------------------------------
// store = false: the caller writes ev_time[idx] / ev_info[idx] itself (lane idx % 8), behind the loads that follow — vector
// stores and loads share one in-order counter, so a load issued after a store is not back before the store is acknowledged
__device__ __forceinline__ int ev_push(const DevParams& P, EnvG& e, int lane, double t, u64 info, bool store = true) {
  const int gl = lane & 7;
  // a slot for the entry: the top of the free-slot stack (fed by the releases and by the rebuild scan of control
  // kernel B2); a dense table appends; only a table with holes nobody recorded is searched
  int idx = -1;
  ORL_DBG(0, 1); ORL_DBG(1, (e.pop_idx < 0 && e.ev_cnt < e.ev_hwm) ? 1 : 0); ORL_DBG(2, e.ev_hwm); ORL_DBG(3, e.ev_cnt);
  if (e.pop_idx >= 0) {
    idx = e.pop_idx;
    e.pop_idx = -1;  // one push per launch; the remaining entries stay in the record
    e.nfree--;
  } else if (e.ev_cnt < e.ev_hwm) {
    for (int base = (e.ev_hwm - 1) & ~31; base >= 0 && idx < 0; base -= 32) {  // four 8-slot chunks requested together
      double t[4];
      ORL_DBG(4, 1);
#pragma unroll
      for (int c = 0; c < 4; c++) {
        int i = base + 8 * c + gl;
        t[c] = (i < e.ev_hwm) ? e.ev_time[i] : 0.0;
      }
#pragma unroll
      for (int c = 0; c < 4; c++) {
        u32 b = gballot(t[c] == __builtin_inf(), lane);
        if (b && idx < 0) idx = base + 8 * c + (int)__builtin_ctz(b);
      }
    }
  }
  if (idx < 0) {
    if (e.ev_hwm >= P.ev_cap) { e.flags |= ORL_FLAG_EV_OVERFLOW; return -1; }
    idx = e.ev_hwm++;
  }
  if (store && gl == (idx & 7)) { e.ev_time[idx] = t; e.ev_info[idx] = info; }
  e.ev_cnt++;
  e.next_rel = t < e.next_rel ? t : e.next_rel;  // -inf (unknown) stays -inf
  ORL_DBG(5, t < e.t_soon ? 1 : 0);
  if (t < e.t_soon) {
    // invariant of the soon list: it holds EVERY pending release earlier than t_soon
    double a[ORL_SOON_PER_LANE];
#pragma unroll
    for (int k = 0; k < ORL_SOON_PER_LANE; k++) a[k] = e.sr_on ? e.sr_t[k] : e.soon_t[gl + 8 * k];
    int slot = -1;
#pragma unroll
    for (int k = 0; k < ORL_SOON_PER_LANE; k++) {
      const u32 f = gballot(a[k] == __builtin_inf(), lane);
      if (f && slot < 0) slot = 8 * k + (int)__builtin_ctz(f);
    }
    if (slot >= 0) {
      soon_set(e, gl, slot, t, idx);
    } else {
      // full: keep the earliest ORL_SOON; the horizon moves down to the latest of what was there
      double m = a[0];
      int ms = gl;
#pragma unroll
      for (int k = 1; k < ORL_SOON_PER_LANE; k++)
        if (a[k] > m) { m = a[k]; ms = gl + 8 * k; }
#define ORL_MAX_STEP(CTRL) { double om = dpp_d<CTRL>(m); int os = dpp_i<CTRL>(ms); if (om > m || (om == m && os < ms)) { m = om; ms = os; } }
      ORL_MAX_STEP(ORL_DPP_XOR1) ORL_MAX_STEP(ORL_DPP_XOR2) ORL_MAX_STEP(ORL_DPP_HALF_MIRROR)
#undef ORL_MAX_STEP
      if (t < m) {
        soon_set(e, gl, ms, t, idx);
        e.t_soon = m;
      } else {
        e.t_soon = t;
      }
    }
  }
  return idx;
}

template <int ENV, int W>
__device__ __forceinline__ void release_one(const DevParams& P, EnvG& e, int lane, int bi) {
  const int gl = lane & 7, owner = bi & 7;
  u64 info = (gl == owner) ? e.ev_info[bi] : 0ull;
  info = gget(info, owner, lane);
  if (gl == owner) e.ev_time[bi] = __builtin_inf();
  const int pidx = (int)(info & 0xffffffu), s0 = (int)((info >> 24) & 0xfffu), n = (int)((info >> 36) & 0xffu);
  const int core = (int)((info >> 44) & 0x1fu), br = (int)((info >> 49) & 0x7fffu);
  e.ev_cnt--;
  free_push(e, gl, bi);
  const int hops_r = path_apply<ENV, W>(P, e, lane, pidx, core, s0, n, true);
  e.s_br -= br;
  e.s_nh -= (i64)n * hops_r;
}

// release every pending service with release_time <= now in increasing time order (rmsa_env.py:590-597), in place
template <int ENV, int W>
__device__ __forceinline__ void release_due(const DevParams& P, EnvG& e, int lane) {
  if (e.next_rel > e.now) return;  // nothing can be due (next_rel is a lower bound of every pending time)
  const int gl = lane & 7;
  for (;;) {
    // one pass over the lane's own slots: its two earliest due entries, whether it has more, and its earliest future one
    double d0t = __builtin_inf(), d1t = __builtin_inf(), rest = __builtin_inf();
    int d0i = 0x7fffffff, d1i = 0x7fffffff, ndue = 0;
    for (int base = gl; base < e.ev_hwm; base += 64) {
      double tt[8];
#pragma unroll
      for (int k = 0; k < 8; k++) {  // eight independent loads in flight per lane
        int i = base + 8 * k;
        tt[k] = (i < e.ev_hwm) ? e.ev_time[i] : __builtin_inf();
      }
#pragma unroll
      for (int k = 0; k < 8; k++) {
        const int i = base + 8 * k;
        const double t = tt[k];
        if (t <= e.now) {
          ndue++;
          if (t < d0t || (t == d0t && i < d0i)) { d1t = d0t; d1i = d0i; d0t = t; d0i = i; }
          else if (t < d1t || (t == d1t && i < d1i)) { d1t = t; d1i = i; }
        } else {
          rest = t < rest ? t : rest;
        }
      }
    }
    const bool overflow = gballot(ndue > 2, lane) != 0u;
#define ORL_MIN_STEP(CTRL) { double ot = dpp_d<CTRL>(bt); int oi = dpp_i<CTRL>(bi); if (ot < bt || (ot == bt && oi < bi)) { bt = ot; bi = oi; } }
    if (overflow) {
      // a lane holds 3+ due entries (rare): release only the globally earliest one, then rescan
      double bt = d0t; int bi = d0i;
      ORL_MIN_STEP(ORL_DPP_XOR1) ORL_MIN_STEP(ORL_DPP_XOR2) ORL_MIN_STEP(ORL_DPP_HALF_MIRROR)
      release_one<ENV, W>(P, e, lane, bi);
      continue;
    }
    for (;;) {
      double bt = d0t; int bi = d0i;
      ORL_MIN_STEP(ORL_DPP_XOR1) ORL_MIN_STEP(ORL_DPP_XOR2) ORL_MIN_STEP(ORL_DPP_HALF_MIRROR)
      if (!(bt <= e.now)) break;
      release_one<ENV, W>(P, e, lane, bi);
      if (gl == (bi & 7)) { d0t = d1t; d0i = d1i; d1t = __builtin_inf(); d1i = 0x7fffffff; }
    }
#undef ORL_MIN_STEP
    e.next_rel = g8_min(rest);
    break;
  }
  // shrink the scan window when its tail is empty (bounded work per step; a stale larger hwm is harmless)
  for (int it = 0; it < 4 && e.ev_hwm > 0; it++) {
    int i = e.ev_hwm - 1;
    double t = gget((gl == (i & 7)) ? e.ev_time[i] : 0.0, i & 7, lane);
    if (t == __builtin_inf()) { e.ev_hwm--; e.nfree = 0; } else break;  // the stack may name slots beyond the window: drop it
  }
}

// (the due releases are detected by sp::release_soon afterwards; RWA / RMCSA release before creating the service in the
// reference, which touches disjoint state, so the order within the kernel does not matter)
template <int ENV, int W>
__device__ __forceinline__ void next_service(const DevParams& P, EnvG& e, int lane, RngG& r) {
  if (e.new_service) return;
  const int gl = lane & 7;
  // at = now + expovariate(1/miat); ht = expovariate(1/mht) (rmsa_env.py:548-553): the two random() draws come first, in
  // the reference's order, then lanes 0-3 of the group evaluate -log(1 - u1) / lambda_a and lanes 4-7 -log(1 - u2) / lambda_h
  // in one pass (same operations on the same operands as two calls one after the other)
  const double u1 = rng_random(e, r, lane), u2 = rng_random(e, r, lane);
  const bool second = gl >= 4;
  const double q = -orl_log(1.0 - (second ? u2 : u1)) / (second ? P.lambda_h : P.lambda_a);
  double at = e.now + gget(q, 0, lane);
  e.now = at;
  double ht = gget(q, 4, lane);
  int src = rng_choice(e, r, lane, P.cum_src, P.N);
  int dst = rng_choice(e, r, lane, P.cum_dst + src * P.N, P.N);
  int bit_rate = 0, br_idx = 0;
  if (ENV == ENV_QOS) {  // the service class (qos_constrained_ra.py:262-265) rides in the bit-rate fields
    br_idx = rng_choice(e, r, lane, P.cum_class, P.n_classes);
    bit_rate = br_idx;
  }
  if (ENV != ENV_RWA && ENV != ENV_QOS) {
    if (P.bit_rate_mode == 0) {
      u32 v = rng_u32(e, r, lane) >> (32 - P.rand_bits);
      while ((int)v >= P.rand_n) v = rng_u32(e, r, lane) >> (32 - P.rand_bits);
      br_idx = (int)v;
      bit_rate = P.br_lo + br_idx;
    } else {
      br_idx = rng_choice(e, r, lane, P.cum_br, P.n_br);
      bit_rate = P.bit_rates[br_idx];
    }
  }
  rng_commit_pos(e, r);  // (the caller stores the regenerated words: rng_commit_stores)
  e.id = (int)e.esp;
  e.src = src; e.dst = dst; e.at = at; e.ht = ht; e.bit_rate = bit_rate; e.br_idx = br_idx;
  e.new_service = 1;
  if (ENV == ENV_RMSA || ENV == ENV_DEEPRMSA) { e.sp += 1; e.esp += 1; }
  if (ENV != ENV_RWA && ENV != ENV_QOS) {
    e.brq += bit_rate;
    e.ebrq += bit_rate;
    if (P.bit_rate_mode == 1 && gl == 0) P.br_hist[e.env * 2 * P.n_br + br_idx] += 1;
  }
}

// lane-private AND of a path's rows, only valid BEFORE this kernel modified the slot map (DeepRMSA action decode)
template <int W>
__device__ __forceinline__ Row<W> path_and_global(const DevParams& P, const EnvG& e, int pidx) {
  return path_and_rec<W>(path_rec_load(P, pidx), e.bm, P.E, P.S, 0);
}

}  // namespace g8
}  // namespace orl
