// orl_device_split.h — step() as two phases: per-env control -> work items -> one lane per touched link row.
//
// Persistent kernel k_persist (orl_kernels.hip; the device-resident loop): one wavefront owns 8 envs for a whole launch and
// alternates the two phases below; k_agent runs them once, for an agent-driven step, and adds validation, info and the
// observation; the two-kernel test form (-DORL_ALT_IMPLS) runs them as separate launches:
//   control     8 lanes per env (ctrl_a): the slot scan (policy) or the agent's action, decode + validate, counters, reward,
//               the release push, network throughput, the next service (RNG, node pair, bit rate), done / auto reset, and
//               the due releases of the step through the env's soon list (release_soon).  Output: work items in an LDS
//               sink, one per touched link — the provision's mask first, then the release masks (single-core families: one
//               bit word per link + a mask table per env; RMCSA / two-kernel form: 24-byte entries, 32-byte queue items)
//   rows        one lane per item (row_item_lane1 / row_item_lane): clear / set the slots, per-link statistics, compactness
//               sums (integer atomics)
//   rel_serial  the rare envs whose releases did not fit the item form release them in place (start of the owning
//               wavefront's next launch; k_agent: same launch; two-kernel form: k_rel_tail)
//   The network-compactness average needs the sums between the provision and the releases: the next control phase finishes
//   it from totals - rel_sums (k_finish2 at the end of a run, k_agent at the end of its launch).
//
// Why: in the monolithic kernels the row work (bit tricks + float64 running averages) ran under per-env control
// flow — one or two link rows per pass, multiplied by the worst hop count and release count among the envs sharing a
// wavefront.  Flattened into items, the row phase is a flat loop over independent items and the control phase shrinks to
// the genuinely serial part.  Design rules: request everything a phase needs in one batch, keep stores behind the last
// load (they share the in-order memory counter), no per-slot searches (free-slot stack, soon list), branch-free
// selection, rare paths out of line.  Round 3: the kernel is bound by instruction issue (VALU ~76 % busy at 3 waves per
// SIMD), so the rules that matter now are about instruction count — masks applied as the two words they lie in, row
// summaries cached per row, the releases of a wavefront-step applied by their holder lanes in parallel.
// Semantics, operation order of every float64 expression and the reference line ranges are those of orl_device.h /
// orl_device_g8.h; the parity suite runs every case against these paths and compares them with the monolithic one on
// every env of full-size batches.
#pragma once
#include "orl_device_g8.h"

namespace orl {
namespace sp {

// -DORL_TIMING (diagnostic builds only): shader-clock cycles per phase of the persistent kernel, accumulated per wavefront
// over a launch and added into g_prof at its end; orl_batch_debug_prof sums them over the wavefronts.  Slots: 0..15 control
// phase (ORL_PROFA), 16..31 release detection inside it (ORL_PROF), 32..47 row phase (ORL_PROFR; the clock of lane 0's item).
#define ORL_PROF_SLOTS 48
#define ORL_PROF_WAVES 16384
#ifdef ORL_TIMING
struct Prof { long long t; unsigned long long acc[ORL_PROF_SLOTS]; };
static __device__ unsigned long long g_prof[ORL_PROF_WAVES * ORL_PROF_SLOTS];
#define ORL_PROF_BEGIN_() do { for (int k_ = 0; k_ < ORL_PROF_SLOTS; k_++) prof.acc[k_] = 0; prof.t = clock64(); } while (0)
#define ORL_PROF_(k) do { long long n_ = clock64(); prof.acc[k] += (unsigned long long)(n_ - prof.t); prof.t = n_; } while (0)
#define ORL_PROF_END_() do { if ((threadIdx.x & 63) == 0) { const size_t w_ = ((size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) % ORL_PROF_WAVES; \
    unsigned long long* g_ = ::orl::sp::g_prof + w_ * ORL_PROF_SLOTS; \
    for (int k_ = 0; k_ < ORL_PROF_SLOTS; k_++) g_[k_] += prof.acc[k_]; } } while (0)
#define ORL_PROF_BEGIN() ORL_PROF_BEGIN_()
#define ORL_PROF_END() ORL_PROF_END_()
#define ORL_PROFA(k) ORL_PROF_(k)
#define ORL_PROF(k) ORL_PROF_(16 + (k))
#define ORL_PROFR(k) ORL_PROF_(32 + (k))
#else
struct Prof { long long t; };
#define ORL_PROF_BEGIN() do { } while (0)
#define ORL_PROF_END() do { } while (0)
#define ORL_PROFA(k) do { } while (0)
#define ORL_PROF(k) do { } while (0)
#define ORL_PROFR(k) do { } while (0)
#endif
#define ORL_PROFA_BEGIN() do { } while (0)
#define ORL_PROFA_END() do { } while (0)
#define ORL_PROFR_BEGIN() do { } while (0)
#define ORL_PROFR_END() do { } while (0)

// x / d for several x and one d.  The compiler's float64 division (v_div_scale x 2, v_rcp_f64, four refinement FMAs, quotient,
// remainder, v_div_fmas, v_div_fixup) spends half of its instructions on the reciprocal of the denominator; the running averages
// of a link divide three sums by the same clock.  Same operations on the same operands as that sequence for operands that need
// no rescaling (v_div_scale returns them unchanged unless an exponent is within 2^-/+768 of the range's ends; the clock is a
// positive simulation time, the sums are products of ratios in [0, 1] and such times), so the quotients are bit-identical —
// the parity suite compares every link's averages of every env with the oracle's.
struct Recip { double d, r; };
__device__ __forceinline__ Recip recip_of(double d) {
  Recip k;
  k.d = d;
  double r = __builtin_amdgcn_rcp(d);
  double e = __builtin_fma(-d, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-d, r, 1.0);
  k.r = __builtin_fma(r, e, r);
  return k;
}
__device__ __forceinline__ double div_by(double x, const Recip& k) {
  const double q = x * k.r;
  return __builtin_fma(__builtin_fma(-k.d, q, x), k.r, q);
}
// x / d for 0 <= x, 0 < d in that range (counts of slots, blocks and links; clocks; sums of such): the same sequence without
// the two v_div_scale, v_div_fmas' rescaling and v_div_fixup's special cases — 8 instructions instead of 14
__device__ __forceinline__ double div_pos(double x, double d) { return div_by(x, recip_of(d)); }

// Diagnostic builds (-DORL_DIAG plus -DORL_X_SKIP_<PHASE>, tools/valu_ab.sh): orl_diag.h holds what such a build puts in place
// of a phase it leaves out — their RESULTS ARE WRONG, their instruction counters are what is read.  Here and in orl_kernels.hip
// only the hook points (ORL_DIAG_*) remain; without -DORL_DIAG every one of them is the product code.
#include "orl_diag.h"
using g8::EnvG;
using g8::gballot;
using g8::gget;

// work item (32 bytes, two 16-byte halves):
//   a.x = env:32 | link:8 | nmask:4 | op:2 (bit 0: masks are releases (set), bit 1: mask 0 is this step's
//         provision (clear, at the provision clock) and the rest are releases — two-kernel pipeline)
//   a.y = masks 0..3, 16 bits each (s0:9 | n:7)      b.x = masks 4..7      b.y = core of mask k, 5 bits each
struct Item { ulonglong2 a, b; };
__device__ __forceinline__ Item make_item(i64 env, u32 link, int nmask, u64 m0, u64 m1, u64 cores, int op) {
  Item it;
  it.a.x = (u64)(u32)env | ((u64)(link & 0xffu) << 32) | ((u64)(u32)nmask << 40) | ((u64)(u32)op << 44);
  it.a.y = m0;
  it.b.x = m1;
  it.b.y = cores;
  return it;
}
__device__ __forceinline__ void item_store(ulonglong2* q, size_t idx, const Item& it) { q[2 * idx] = it.a; q[2 * idx + 1] = it.b; }

// Every control WAVEFRONT (8 envs) owns a fixed region of P.q_wave item slots in the queue (8 x the most items one env
// can produce: max(hops, links)) and publishes how many it filled; the row kernel maps its dense item index onto the
// four regions of a control workgroup.  No global atomic (a single queue-tail counter serialised the 2 048 workgroups
// of a 65 536-env launch, ~12 ns per atomic on one address) and no workgroup barrier (waiting for the slowest of four
// wavefronts was a third of control kernel B2's time).  `cnt` is uniform within each 8-lane group; all 64 lanes call.
__device__ __forceinline__ size_t wave_reserve(const DevParams& P, int cnt, u32* wave_counts, int lane) {
  const int grp = lane >> 3;
  int pre = 0, tot = 0;
#pragma unroll
  for (int g = 0; g < 8; g++) {
    const int c = __builtin_amdgcn_readlane(cnt, 8 * g);
    pre += (g < grp) ? c : 0;
    tot += c;
  }
  const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (lane == 0) wave_counts[wave] = (u32)tot;
  return wave * (size_t)P.q_wave + (size_t)pre;
}

// ---- release sink -----------------------------------------------------------------------------------
// Control kernel B2 does not touch link rows: the releases of a step are collected as work items for the row kernel,
// one item per touched LINK (per-link statistics are shared by all cores of the link) carrying up to ORL_IMASKS
// [s0, s0+n) masks in release order, each with its core.  While collecting, the items live in an LDS table indexed
// by the link (24 bytes per link and env): appending a mask is one LDS read-modify-write by the lane that owns the
// hop, with no search and no per-lane registers.
struct SinkEntry {
  u64 mk0;  // masks 0..3: (s0 | n << 9), 16 bits each, in release order
  u64 mk1;  // masks 4..7
  u64 crn;  // the core of each mask, 5 bits each (40 bits) | number of masks << 40 | mask 0 is a provision << 44
};
// compact form (single-core families in the persistent kernel's LDS window).  A release frees the SAME slots on every link of
// its path, so the masks need not be copied into per-link entries: per env one table of masks — entry 0 the step's provision,
// entry k (1..ORL_REL_MAX) its k-th release in heap-pop order, 16 bits each (first slot: 9 | slots: 6 — the split pipeline
// serves services of at most 63 slots, orl_api.hip) — and per (env, link) ONE 32-bit word of which of them touch the link.
// Appending is one LDS atomic OR (its return value says whether the entry was empty, i.e. whether a new work item opens);
// there is no per-link capacity any more (8 masks per 16-byte entry before: a tally pass guarded it), only the 31 releases
// per env-step of the table; and the table is 4 bytes per link and env instead of 16 (cfg2: 1 216 B with the masks instead
// of 2 816 B + 192 B of tallies), which is what lets the LDS window of the 4-wave form fit 16 times into a CU.
struct SinkEntryC {
  u32 bits;  // bit 0: the provision; bit k: the k-th release of this step
};
#define ORL_REL_MAX 31
#define ORL_MTAB 32  // masks per env in the table
struct Mask2 { u64 lo, hi; int w0; };
__device__ __forceinline__ Mask2 mask2(int s0, int n) {
  Mask2 m;
  const int b = s0 & 63;
  const u64 ones = (1ull << n) - 1ull;
  m.w0 = s0 >> 6;
  m.lo = ones << b;
  m.hi = (b + n > 64) ? (ones >> (64 - b)) : 0ull;  // (b + n > 64 implies b >= 2: the shift is in range)
  return m;
}
__device__ __forceinline__ u64 mask2_word(const Mask2& m, int w) { return (w == m.w0) ? m.lo : ((w == m.w0 + 1) ? m.hi : 0ull); }
// (the two-wavefront form: the control wavefront changes the slot maps itself, in the LDS window — no value comes back)
__device__ __forceinline__ void row_apply_mask(u64* row, int s0, int n, bool provision) {
  const Mask2 m = mask2(s0, n);
  if (provision) {
    __hip_atomic_fetch_and(row + m.w0, ~m.lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (m.hi) __hip_atomic_fetch_and(row + m.w0 + 1, ~m.hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  } else {
    __hip_atomic_fetch_or(row + m.w0, m.lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    if (m.hi) __hip_atomic_fetch_or(row + m.w0 + 1, m.hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
}
template <bool CP> struct SinkEntryOf { typedef SinkEntry type; };
template <> struct SinkEntryOf<true> { typedef SinkEntryC type; };
// RD (rows deferred, round 6): no table at all — the control phase changes the slot maps itself (as in the two-wavefront form)
// and appends one 16-byte EVENT per provision / release to the env's log of the launch (DevParams::elog): what k_rowstats replays
// the per-link statistics and the compactness sums from, one lane per link row
#define ORL_EV_META(S0, N, T, PROV) ((u64)(u32)(S0) | ((u64)(u32)(N) << 9) | ((u64)(u32)(T) << 15) | ((u64)((PROV) ? 1u : 0u) << 24))
template <bool CP, bool RD = false> struct SinkT {
  typedef typename SinkEntryOf<CP>::type Entry;
  ulonglong2* ev;  // RD: the env's event log of this launch, two 16-byte halves per event: {meta, link bits}, {clock, 0}
  double ev_clock; // RD: the clock the step was decided at (its provision's clock)
  int ev_at;       // RD: index of this step's first event (its provision, if it has one)
  int ev_rel0;     // RD: index of this step's first release event
  int ev_t;        // RD: the step's number within the launch
  unsigned short* list;  // LDS (persistent kernel, else nullptr): the wavefront's open table entries, (local env << 8) | link ...
  u32* list_n;           // ... and their number, zeroed at the start of the step
  Entry* tab;      // LDS, E entries of this env, mask count zeroed
  u32* tally;      // LDS, `tw` words per env, zeroed: per-link touch counters (4 x 8 bit per word) for the capacity check (!CP)
  int tw;
  unsigned short* mtab;  // CP: LDS, this env's mask table (ORL_MTAB entries)
  int nrel;              // CP: releases of this step appended so far (group-uniform)
  u64* rows;             // CP, two-wavefront form: the env's slot map (LDS) — masks are applied as they are appended; else nullptr
  int roww;              // ... and its words per row
  bool active;     // item mode decided: the releases of this step fit the item form
  bool deferred;   // they do not: nothing has been touched, k_rel_tail releases them in place
  int cnt;         // links this LANE has opened an item for
};
typedef SinkT<false> Sink;
__device__ __forceinline__ void sink_entry_clear(SinkEntry& t) { t.crn = 0ull; }
__device__ __forceinline__ void sink_entry_clear(SinkEntryC& t) { t.bits = 0u; }
// appends mask m to the entry; returns the number of masks it held, bit 8: the entry starts with the step's provision
__device__ __forceinline__ int sink_entry_add(SinkEntry* t, u64 m, int core, bool prov) {
  const u64 crn = t->crn | (prov ? (1ull << 44) : 0ull);  // (a provision is the first thing a step adds)
  const int j = (int)((crn >> 40) & 15);
  if (j < 4) t->mk0 = (j == 0 ? 0ull : t->mk0) | (m << (16 * j));
  else t->mk1 = (j == 4 ? 0ull : t->mk1) | (m << (16 * (j - 4)));
  t->crn = (crn & ((1ull << 44) | 0xffffffffffull)) | ((u64)(u32)core << (5 * j)) | ((u64)(u32)(j + 1) << 40);
  return j | (int)((crn >> 44) & 1) << 8;
}
// lane h of the group appends the mask to the item of hop h's link (the links of one path are distinct)
template <bool CP, bool RD>
__device__ __forceinline__ void sink_add(SinkT<CP, RD>& s, const PathRec& rec, int core, int s0, int n, int lane, bool prov = false) {
  const int hops = path_rec_byte(rec, 0);
  if constexpr (RD) {
    // (the step's provision: the hops over the group's lanes, their link bits ORed over the group; the releases of a step are
    // logged by their holder lanes, release_soon)
    u32 lm0 = 0u, lm1 = 0u;
    for (int h = lane & 7; h < hops; h += 8) {
      const int link = path_rec_byte(rec, 2 + h);
      row_apply_mask(s.rows + (size_t)link * s.roww, s0, n, prov);
      if (link < 32) lm0 |= 1u << link; else lm1 |= 1u << (link - 32);
    }
    lm0 |= (u32)dpp_i<ORL_DPP_XOR1>((int)lm0); lm1 |= (u32)dpp_i<ORL_DPP_XOR1>((int)lm1);
    lm0 |= (u32)dpp_i<ORL_DPP_XOR2>((int)lm0); lm1 |= (u32)dpp_i<ORL_DPP_XOR2>((int)lm1);
    lm0 |= (u32)dpp_i<ORL_DPP_HALF_MIRROR>((int)lm0); lm1 |= (u32)dpp_i<ORL_DPP_HALF_MIRROR>((int)lm1);
    if ((lane & 7) == 0) {
      s.ev[2 * s.ev_at] = make_ulonglong2(ORL_EV_META(s0, n, s.ev_t, prov), (u64)lm0 | ((u64)lm1 << 32));
      s.ev[2 * s.ev_at + 1] = make_ulonglong2((u64)__double_as_longlong(s.ev_clock), 0ull);
    }
  } else if constexpr (CP) {
    const int k = prov ? 0 : ++s.nrel;  // this mask's entry of the env's table (the caller keeps nrel <= ORL_REL_MAX)
    if ((lane & 7) == 0) s.mtab[k] = (unsigned short)((u32)s0 | ((u32)n << 9));
    // (no value comes back from the atomics: the item list is made from the table afterwards, sink_compact — until round 5 every
    // hop waited for its atomic OR to learn whether a list entry was due, and for an atomic add when one was: two dependent LDS
    // round trips per hop of every provision and release)
    for (int h = lane & 7; h < hops; h += 8) {
      const int link = path_rec_byte(rec, 2 + h);
      atomicOr(&s.tab[link].bits, 1u << k);
      if (s.rows) row_apply_mask(s.rows + (size_t)link * s.roww, s0, n, prov);
    }
  } else {
    const u64 m = (u64)(u32)s0 | ((u64)(u32)n << 9);
    for (int h = lane & 7; h < hops; h += 8) {
      const int r = sink_entry_add(s.tab + path_rec_byte(rec, 2 + h), m, core, prov);
      const int j = r & 15;
      s.cnt += (j == 0) ? 1 : 0;
      if (s.list && (j == 0 || (j == 1 && (r >> 8))))
        s.list[atomicAdd(s.list_n, 1u)] = (unsigned short)((j << 15) | (((lane >> 3) & 7) << 8) | path_rec_byte(rec, 2 + h));
    }
  }
}

// the env's items = the table entries that hold masks, in link order, into the wavefront's queue region
__device__ __forceinline__ void emit_items(const DevParams& P, i64 env, const Sink& sink, bool any, int lane, ulonglong2* q, u32* counts) {
  const int gl = lane & 7, E = P.E;
  const int cnt = g8_sum(any ? sink.cnt : 0);
  const size_t base = wave_reserve(P, cnt, counts, lane);
  if (cnt) {
    size_t at = base;
    for (int l0 = 0; l0 < E; l0 += 8) {
      const int l = l0 + gl;
      const u64 crn = (l < E) ? sink.tab[l].crn : 0ull;
      const int nm = (int)((crn >> 40) & 15);
      const u32 fb = gballot(nm > 0, lane);
      if (nm > 0)
        item_store(q, at + __popc(fb & ((1u << gl) - 1u)),
                   make_item(env, (u32)l, nm, sink.tab[l].mk0, nm > 4 ? sink.tab[l].mk1 : 0ull, crn & 0xffffffffffull,
                             1 | (int)((crn >> 44) & 1) << 1));
      at += __popc(fb);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------
// Service look-ahead of the persistent kernel.  _next_service (rmsa_env.py:545-597, rwa_env.py:258-288,
// rmcsa_env.py:690-739) draws the next arrival from the env's own random.Random — inter-arrival time, holding time, source,
// destination, bit rate — and nothing it draws depends on the state of the network: the traffic is open-loop.  Drawn inside
// the step it is the same arithmetic on all 8 lanes of an env's group (two glibc-exact logarithms, two float64 divisions, two
// table searches, a dozen cross-lane word fetches: every instruction of it serves 8 envs per wavefront) and a chain of three
// dependent memory round trips (logarithm table, source table, destination table).  Here every 8 steps the group draws its
// next 8 services AT ONCE, lane j the j-th: the lanes regenerate a window of 96 Mersenne-Twister words together (12 per lane),
// find where each service starts in the word stream (random.randint is a rejection loop, so a service takes 8 + r words: a
// walk over the window's accept bits), and each lane evaluates its own service — same operations on the same operands as
// the in-step draw, which the one-step kernels keep.  A step then takes (inter-arrival time, holding time, source,
// destination, bit-rate index) of its service from the lane that holds it.  Words are committed (regenerated words stored,
// stream position advanced) for exactly the services generated, and never more services than the launch has steps left: the
// Mersenne-Twister state a launch leaves is the state after the services its steps consumed.  A wavefront that leaves its
// loop early (releases to be done in place, orl_kernels.hip) parks what it holds in DevParams::svc_* and picks it up at the
// start of its next launch.
// ---------------------------------------------------------------------------------------------------------------
#define ORL_SVC_WIN 12  // Mersenne-Twister words per lane and batch: 8 services take 64 (+ 2 x 8 in the discrete bit-rate mode) or
                        // 64 + rejected bit-rate draws (8 x 0.68 expected for 76 rates out of 128; more than 32: once in 10^5
                        // batches — the group then gets the services that fit and draws again when they are used up)
struct SvcBuf {
  double q, ht;  // this lane's service of the batch: inter-arrival time, holding time
  u32 pk;        // source | destination << 10 | bit-rate index << 20 (<= 512 nodes, <= 4 096 bit rates: orl_api.hip)
  int cnt;       // group-uniform: services in the batch << 8 | next one to take
};
__device__ __forceinline__ bool svc_empty(const SvcBuf& b) { return (b.cnt & 0xff) >= (b.cnt >> 8); }
// i mod 624 for 0 <= i < 1248 (a subtraction and an unsigned minimum: below 624 the difference wraps to a huge value)
__device__ __forceinline__ int svc_wrap(int i) { const u32 u = (u32)i, d = u - 624u; return (int)(d < u ? d : u); }
// bisect(cum_weights, x, 0, n - 1) of random.choices: the number of entries cum[0 .. n-2] that are <= x (cum is non-decreasing).
// Two rounds of independent requests instead of a chain of log2(n) dependent ones: every eighth entry first (the block the
// answer lies in), then that block's entries — Germany50's 49 entries took six dependent memory round trips per table.
__device__ __forceinline__ int svc_choice(const double* cum, int n, double u) {
  const double x = u * (cum[n - 1] + 0.0);
  const int m = n - 1;  // entries searched
  if (m <= 0) return 0;
  int b = 0;
  if (m > 8 && m <= 72) {
    double p[8];
#pragma unroll
    for (int j = 0; j < 8; j++) p[j] = cum[8 * j + 7 < m ? 8 * j + 7 : m - 1];
#pragma unroll
    for (int j = 0; j < 8; j++) b += (8 * j + 7 < m && p[j] <= x) ? 1 : 0;
  } else if (m > 72) {
    int lo = 0, hi = m;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (cum[mid] <= x) lo = mid + 1; else hi = mid;
    }
    return lo;
  }
  double c[8];
#pragma unroll
  for (int k = 0; k < 8; k++) c[k] = cum[8 * b + k < m ? 8 * b + k : m - 1];
  int cnt = 8 * b;
#pragma unroll
  for (int k = 0; k < 8; k++) cnt += (8 * b + k < m && c[k] <= x) ? 1 : 0;
  return cnt;
}
// `rec`: the env's record (LDS window or global), `mt`: its Mersenne-Twister state, `active`: this group draws (its buffer
// is used up), `n_want`: services wanted (<= 8: the steps the launch has left).  All 64 lanes call.
template <int ENV>
__device__ __forceinline__ void svc_generate(const DevParams& P, u64* rec, u32* mt, int lane, int n_want, SvcBuf& sb, bool active) {
  const int gl = lane & 7;
  const bool seek = (ENV != ENV_RWA) && P.bit_rate_mode == 0;  // random.randint: words are drawn until one is below rand_n
  const int FIXED = (ENV == ENV_RWA) ? 8 : (P.bit_rate_mode == 0 ? 8 : 10);
  const u32 sh = 32u - (u32)P.rand_bits, rn = (u32)P.rand_n;
  const u64 idw = active ? rec[SC_ID_MTPOS] : 0ull;
  const int pos = (int)(idw >> 32);
  // the window: word t of the stream (t = 0 at the env's position) is held by lane t % 8 as its entry t / 8.  A position keeps
  // the current generation's word until that word is handed out and the next generation's afterwards ("update-behind",
  // orl_device_g8.h), so word t needs positions t, t + 1 (current) and t + 397 (mod 624: whichever generation is there) —
  // none of which another word of a 96-word window regenerates before it is read: all requested together
  u32 cur[ORL_SVC_WIN], nxt[ORL_SVC_WIN], far[ORL_SVC_WIN];
#pragma unroll
  for (int k = 0; k < ORL_SVC_WIN; k++) {
    const int i0 = svc_wrap(pos + gl + 8 * k);
    cur[k] = active ? mt[i0] : 0u;
    nxt[k] = active ? mt[svc_wrap(i0 + 1)] : 0u;
    far[k] = active ? mt[svc_wrap(i0 + 397)] : 0u;
  }
  u32 nx[ORL_SVC_WIN];
  u32 acc0 = 0u, acc1 = 0u, acc2 = 0u;  // the group's accept bits: bit t = word t is a valid bit-rate draw
#pragma unroll
  for (int k = 0; k < ORL_SVC_WIN; k++) {
    const u32 y = (cur[k] & 0x80000000u) | (nxt[k] & 0x7fffffffu);
    nx[k] = far[k] ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
    if (seek) {
      u32 t = cur[k];
      t ^= (t >> 11);
      t ^= (t << 7) & 0x9d2c5680u;
      t ^= (t << 15) & 0xefc60000u;
      t ^= (t >> 18);
      const u32 byte = gballot((t >> sh) < rn, lane);
      if (k < 4) acc0 |= byte << (8 * (k & 3));
      else if (k < 8) acc1 |= byte << (8 * (k & 3));
      else acc2 |= byte << (8 * (k & 3));
    }
  }
  // where the services start: a walk over the accept bits, the same on all lanes of the group; lane j keeps service j's
  int o = 0, my_o = 0, my_a = 0, got = 0;
  {
    const u64 m_lo = (u64)acc0 | ((u64)acc1 << 32);
    bool ok = active;
#pragma unroll
    for (int j = 0; j < 8; j++) {
      if (ok && j < n_want) {
        const int c = o + FIXED;
        int a = c - 1;
        if (seek) {
          u64 x = 0ull;
          if (c < 64) x = (m_lo >> c) | ((u64)acc2 << (64 - c));  // (c >= 8: the shift is in range)
          else if (c < 8 * ORL_SVC_WIN) x = (u64)(acc2 >> (c - 64));
          if (x == 0ull) ok = false;
          else a = c + (int)__builtin_ctzll(x);
        } else if (c > 8 * ORL_SVC_WIN) {
          ok = false;
        }
        if (ok) {
          if (gl == j) { my_o = o; my_a = a; }
          o = seek ? a + 1 : c;
          got = j + 1;
        }
      }
    }
  }
  // lane j evaluates service j.  Its words are spread over the group's lanes: read again from memory (just read: cache hits),
  // requested before the regenerated words are stored below (same wavefront: the loads are served first)
  const bool mine = active && gl < got;
  u32 w[10], wa = 0u;
#pragma unroll
  for (int t = 0; t < 10; t++) w[t] = (mine && t < FIXED) ? mt[svc_wrap(pos + my_o + t)] : 0u;
  if (seek && mine) wa = mt[svc_wrap(pos + my_a)];
#pragma unroll
  for (int k = 0; k < ORL_SVC_WIN; k++)
    if (active && gl + 8 * k < o) mt[svc_wrap(pos + gl + 8 * k)] = nx[k];
  if (active && gl == 0) rec[SC_ID_MTPOS] = (idw & 0xffffffffull) | ((u64)(u32)svc_wrap(pos + o) << 32);
  if (mine) {
#pragma unroll
    for (int t = 0; t < 10; t++) {
      u32 v = w[t];
      v ^= (v >> 11);
      v ^= (v << 7) & 0x9d2c5680u;
      v ^= (v << 15) & 0xefc60000u;
      v ^= (v >> 18);
      w[t] = v;
    }
#define ORL_SVC_RND(A, B) ((((double)((A) >> 5)) * 67108864.0 + (double)((B) >> 6)) * (1.0 / 9007199254740992.0))
    const double u1 = ORL_SVC_RND(w[0], w[1]), u2 = ORL_SVC_RND(w[2], w[3]);
    const double us = ORL_SVC_RND(w[4], w[5]), ud = ORL_SVC_RND(w[6], w[7]);
    // at = now + expovariate(1 / mean_iat), ht = expovariate(1 / mean_ht) (rmsa_env.py:548-553; random.expovariate)
    sb.q = -orl_log(1.0 - u1) / P.lambda_a;
    sb.ht = -orl_log(1.0 - u2) / P.lambda_h;
    const int src = svc_choice(P.cum_src, P.N, us);
    const int dst = svc_choice(P.cum_dst + src * P.N, P.N, ud);
    int br_idx = 0;
    if (ENV != ENV_RWA) {
      if (P.bit_rate_mode == 0) {
        u32 v = wa;
        v ^= (v >> 11);
        v ^= (v << 7) & 0x9d2c5680u;
        v ^= (v << 15) & 0xefc60000u;
        v ^= (v >> 18);
        br_idx = (int)(v >> sh);
      } else {
        br_idx = svc_choice(P.cum_br, P.n_br, ORL_SVC_RND(w[8], w[9]));
      }
    }
#undef ORL_SVC_RND
    sb.pk = (u32)src | ((u32)dst << 10) | ((u32)br_idx << 20);
  }
  if (active) {
    // not even the first service fits the window — 88 bit-rate draws rejected in a row: below 10^-26 per batch for any range of
    // rates — would draw the same nothing again for ever: the env is flagged like one that ran out of pending-release slots
    // (its state is no longer valid, the run reports it) and goes on with empty services
    if (got == 0 && n_want > 0) {
      if (gl == 0) rec[SC_FLAGS] |= ((u64)ORL_FLAG_EV_OVERFLOW << 32);
      sb.q = 0.0; sb.ht = 0.0; sb.pk = 0u;
      got = n_want;
    }
    sb.cnt = got << 8;
  }
}

// The work items of a step, from the compact sink table once every mask of the step is in it: one list entry (local env << 8 |
// link) per (env, link) word that holds a bit, and a second one (bit 15) where the word holds the step's provision AND a release
// — two lanes share that row's work (row_item_lane1).  Any order: items are independent.  All 64 lanes call: the 8 lanes of a
// group walk their env's links 8 at a time, ballots and prefix counts place the entries.  `tab`: the wavefront's table [8][E].
__device__ __forceinline__ void sink_compact(const SinkEntryC* tab, int E, int lane, unsigned short* list, u32* list_n) {
  const int gl = lane & 7, el = lane >> 3;
  u32 n = 0u;
  for (int l0 = 0; l0 < E; l0 += 8) {
    const int link = l0 + gl;
    const u32 bits = (link < E) ? tab[el * E + link].bits : 0u;
    const bool nz = bits != 0u, sh = (bits & 1u) != 0u && (bits >> 1) != 0u;
    const u64 bnz = __ballot(nz), bsh = __ballot(sh);
    const u32 pos = n + __builtin_amdgcn_mbcnt_hi((u32)(bnz >> 32), __builtin_amdgcn_mbcnt_lo((u32)bnz, 0u)) +
                    __builtin_amdgcn_mbcnt_hi((u32)(bsh >> 32), __builtin_amdgcn_mbcnt_lo((u32)bsh, 0u));
    const unsigned short code = (unsigned short)((el << 8) | link);
    if (nz) list[pos] = code;
    if (sh) list[pos + 1u] = (unsigned short)(code | 0x8000u);
    n += (u32)__popcll(bnz) + (u32)__popcll(bsh);
  }
  if (lane == 0) *list_n = n;
}

// ---------------------------------------------------------------------------------------------------------------
// control kernel A: everything of step() up to (and excluding) the effects of the provision on the link rows
// ---------------------------------------------------------------------------------------------------------------
template <int ENV, int W>
__device__ __forceinline__ bool service_part(const DevParams& P, EnvG& e, i64 env, int lane, int auto_reset, bool accepted, int core,
                                             bool write_io, g8::RngG& rng, Prof& prof, SvcBuf* svc);
#ifndef ORL_SCAN_BATCH
#define ORL_SCAN_BATCH 8  // release times a lane requests per round of the rebuild scan
#endif
struct SoonRegs { double t[ORL_SOON_PER_LANE]; int i[ORL_SOON_PER_LANE]; int dirty; };  // dirty: bit k = entry k of this lane changed

// Where the slot maps, link statistics and per-core sums of the envs a wavefront works on live: the global arrays
// (env0 = 0), or — persistent kernel, small topologies — the wavefront's own LDS window holding its 8 envs for the whole
// launch (env0 = its first env).  Indexed by (env - env0) either way; the compiler infers the address space per kernel.
struct Wmem {
  u64* bm0;      // [..][bm_words]
  double* ls0;   // [..][E][4]
  int* cs0;      // [..][cs_words]
  i64 env0;      // index base of bm0
  i64 senv0;     // index base of ls0
  i64 cenv0;     // index base of cs0
  u64* sc0;      // [..][sc_stride] env records
  int sc_stride; // words between two records: ORL_SCAL_WORDS in global memory, ORL_SCAL_LDS_WORDS in the LDS window
  i64 scenv0;    // index base of sc0
  int cs_stride; // ints per env in cs0
  u32* ic0;      // LDS [8][E]: per link row, the longest free run strictly inside each 64-slot word (6 bits per word, 63 =
                 // unknown), or nullptr; indexed with cenv0 (row_stat_lane)
  u32* oc0;      // LDS [8][E]: per link row, its contribution to the compactness sums, (occ << 16) | free blocks inside, or
                 // nullptr (then the row phase recomputes it from the row before it changes it); indexed with cenv0
  double* clk;   // LDS [8][2] {provision clock, step clock} of the wavefront's envs, or nullptr (row phase reads SC_NOWA / SC_NOW)
  i64 clk_env0;  // first env of the wavefront (index base of clk)
  bool cs_lds;   // the sums are in the wavefront's LDS window: plain reads; else they are read through L2 where the row phase's atomics land
  u32* ocg;      // GLOBAL [8][E]: the same words as oc0 where they do not fit the LDS window (the wavefront's level-2 area of
                 // DevParams::row_cache, live during the launch), or nullptr; indexed with cenv0
  double* evl0;  // LDS [8][ev_cap]: the pending release times of the wavefront's envs (small batches, two-wavefront form), or nullptr
  u64* mini;     // LDS [8][ORL_MINI_STRIDE]: the record words ctrl_d works on (deferred statistics, records in global memory), or nullptr
  i64 mini_env0; // index base of mini
};
__device__ __forceinline__ Wmem wmem_global(const DevParams& P) {
  Wmem m;
  m.bm0 = P.bitmap; m.ls0 = P.lstat; m.cs0 = P.core_sums; m.env0 = 0; m.senv0 = 0; m.cenv0 = 0; m.sc0 = P.scal; m.scenv0 = 0; m.sc_stride = ORL_SCAL_WORDS; m.cs_stride = P.cs_words; m.ic0 = nullptr; m.oc0 = nullptr; m.clk = nullptr; m.clk_env0 = 0; m.cs_lds = false; m.ocg = nullptr; m.mini = nullptr; m.mini_env0 = 0; m.evl0 = nullptr;
  return m;
}
__device__ __forceinline__ u64* wm_bm(const DevParams& P, const Wmem& m, i64 env) { return m.bm0 + (env - m.env0) * P.bm_words; }
__device__ __forceinline__ double* wm_ls(const DevParams& P, const Wmem& m, i64 env) { return m.ls0 + (env - m.senv0) * 4 * P.E; }
__device__ __forceinline__ u64* wm_scal(const DevParams& P, const Wmem& m, i64 env) { return m.sc0 + (env - m.scenv0) * m.sc_stride; }
__device__ __forceinline__ int* wm_cs(const DevParams& P, const Wmem& m, i64 env) { return m.cs0 + (env - m.cenv0) * m.cs_stride; }

struct CtrlOpts {
  bool persistent;  // inside k_persist: no kernel boundary between the row phase's L2 atomics and this phase's reads
  bool write_io;    // store the action / reward / done / service descriptor of this step (device-resident runs: last step only)
  bool trusted;     // the action comes from the in-kernel slot scan on the same slot map: is_path_free holds by construction
  bool emit_queue;  // two-kernel form: copy the items into the global queue for the row kernel
  bool prefetch;    // request the Mersenne-Twister window at the start of the phase (costs registers: 3-wave forms only)
  bool auto_reset;  // an env that reports done is soft-reset right away (the device-resident loop, SB3's VecEnv); k_agent: the caller's choice
  bool rank_pairs;  // (soon list in registers) rank the due releases all-pairs over DPP: needs the 168-VGPR budget
};
template <int ENV, int W, bool CP, bool RD = false>
__device__ __forceinline__ void release_soon(const DevParams& P, EnvG& e, int lane, SinkT<CP, RD>& sink, SoonRegs& out, Prof& prof,
                                             int extra = 0, int pushed_idx = -1, u64 pushed_info = 0ull, int pre_idx = -1,
                                             u64 pre_info = 0ull);

// The provision and the releases of the step go into ONE queue as mixed items (per link: the provision mask first, then the
// release masks) and one row phase applies them.  The network-compactness average, which needs the sums between the
// provision and the releases, is finished by the NEXT step from totals - (what the releases added): the row phase keeps
// the latter in rel_sums.
// Returns the service descriptor of the NEW pending service (what the next slot scan needs: pair base, bit-rate index,
// number of paths), and through *n_items_out the number of items this env's step left in its sink table.
// CP: compact sink entries (SinkEntryC); tw: tally words per env (>= ceil(E / 4))
// Host- or agent-driven single steps through these phases (k_agent): what info of step() needs beyond the device-resident
// loop's state (rmsa_env.py:228-264) — the network compactness before the provision, the occupied-slot sum right after it —
// carried from the control phase to the end of the step; the four blocking rates are stored by the control phase itself.
struct InfoCarry { double prev_comp; i64 s_nh_prov; };

template <int ENV, int W, bool CP = false>
__device__ __forceinline__ u64 ctrl_a(const DevParams& P, const Wmem& M, const CtrlOpts& O, i64 env, bool valid, int lane, Prof& prof,
                                      const int4* given, u32* s_tally, typename SinkEntryOf<CP>::type* s_tab, int parity,
                                      int* s_deferred, int* done_out, unsigned short* s_list = nullptr, u32* s_list_n = nullptr,
                                      int tw = 32, SoonRegs* carried = nullptr, unsigned short* s_mtab = nullptr,
                                      InfoCarry* ic = nullptr, SvcBuf* svc = nullptr) {
  // `svc` (persistent kernel): the group's batch of services drawn ahead (svc_generate); else the step draws its own
  const int K = P.K, S = P.S, rej = P.allow_rejection ? 1 : 0, gl = lane & 7;
  if (!O.persistent && blockIdx.x == 0 && threadIdx.x == 0) P.q_def[(size_t)(parity ^ 1) * P.q_def_stride] = 0u;  // the buffer the next step appends to
  u64 desc_out = 0ull;
  int cnt = 0, core = 0, slot = 0, n = 1;
  PathRec rec;
  rec.q[0] = rec.q[1] = rec.q[2] = rec.q[3] = 0;
  SinkT<CP> sink;
  sink.tab = nullptr; sink.tally = nullptr; sink.tw = tw; sink.active = false; sink.deferred = false; sink.cnt = 0;
  sink.list = s_list; sink.list_n = s_list_n; sink.mtab = nullptr; sink.nrel = 0; sink.rows = nullptr; sink.roww = 0;
  if (s_list_n && lane == 0) *s_list_n = 0u;
  {  // every wavefront clears the tables of its own 8 envs: no workgroup barrier
    typename SinkEntryOf<CP>::type* tb = s_tab + P.E * 8 * (int)(threadIdx.x >> 6);
    if constexpr (!CP) {
      u32* ty = s_tally + tw * 8 * (int)(threadIdx.x >> 6);
      for (int i = lane; i < 8 * tw; i += 64) ty[i] = 0u;
      sink.tally = s_tally + tw * (int)(threadIdx.x >> 3);
    } else {
      sink.mtab = s_mtab + ORL_MTAB * (int)(threadIdx.x >> 3);  // (entries are written before they are read: nothing to clear)
    }
    for (int i = lane; i < 8 * P.E; i += 64) sink_entry_clear(tb[i]);
    wave_fence();
    sink.tab = s_tab + P.E * (int)(threadIdx.x >> 3);
  }
  if (valid) {
    EnvG e;
    g8::env_load(P, e, env, wm_scal(P, M, env));
    // the Mersenne-Twister window the next service draws from: requested now, used after the provision
    if (carried) {  // persistent kernel: the soon list stays in registers from step to step
      e.sr_on = true;
      e.rank_pairs = O.rank_pairs;
#pragma unroll
      for (int k = 0; k < ORL_SOON_PER_LANE; k++) { e.sr_t[k] = carried->t[k]; e.sr_i[k] = carried->i[k]; }
    }
    // the info word of this lane's earliest list entry, if it may come due in this step: requested now, the release
    // detection at the end of the phase finds it in a register
    int pre_idx = -1;
    u64 pre_info = 0ull;
    if (carried && O.prefetch) {
      double pt = e.sr_t[0];
      int pi = e.sr_i[0];
#pragma unroll
      for (int k = 1; k < ORL_SOON_PER_LANE; k++)
        if (e.sr_t[k] < pt || (e.sr_t[k] == pt && e.sr_i[k] < pi)) { pt = e.sr_t[k]; pi = e.sr_i[k]; }
      // (t_soon == -inf: the list is stale — after a reset or the serial tail — and its entries mean nothing)
      if (pt <= e.now + P.pf_window && e.t_soon > -__builtin_inf() && (u32)pi < (u32)P.ev_cap) { pre_idx = pi; pre_info = e.ev_info[pi]; }
    }
    g8::RngG rng;
    rng.used = 0; rng.pend_used = 0;
    if (O.prefetch && !svc) g8::rng_fill(e, rng, gl);
    e.bm = wm_bm(P, M, env);
    e.ls = wm_ls(P, M, env);
    e.cs = wm_cs(P, M, env);
    int* rs = e.cs + 2 * P.C;
    {
      const u64 acc0 = e.scal[SC_ACC];
      if ((u32)acc0 & 2u) {
        // network compactness update the previous step left pending: the sums right after ITS provision are the totals
        // minus what its releases added (rmsa_env.py:439-462 with _get_network_compactness at provision time)
        const int c0 = (int)((acc0 >> 32) & 31);
        const i64 s_nh_prov = (i64)(acc0 >> 37);
        // (the sums are updated by L2 atomics: in the persistent kernel, where no kernel boundary invalidates the L1 in
        // between, they are read through L2 as well)
        int occ, fb;
        if (O.persistent && !M.cs_lds) {
          occ = atomicAdd(e.cs + 2 * c0, 0) - atomicAdd(rs + 2 * c0, 0);
          fb = atomicAdd(e.cs + 2 * c0 + 1, 0) - atomicAdd(rs + 2 * c0 + 1, 0);
        } else {
          occ = e.cs[2 * c0] - rs[2 * c0];
          fb = e.cs[2 * c0 + 1] - rs[2 * c0 + 1];
        }
        const double a0 = __longlong_as_double((i64)e.scal[SC_GC_A]), td = __longlong_as_double((i64)e.scal[SC_GC_TD]);
        const double now_a = __longlong_as_double((i64)e.scal[SC_NOWA]);
        const double cmp = (fb > 0) ? div_pos((double)occ, (double)s_nh_prov) * div_pos((double)P.E, (double)fb) : 1.0;
        e.g_comp = div_pos(a0 + (cmp * td), now_a);
      }
      for (int i = gl; i < 2 * P.C; i += 8) {  // this step's releases start from zero
        if (O.persistent && !M.cs_lds) atomicExch(rs + i, 0);
        else rs[i] = 0;
      }
    }
    if (ic && (ENV == ENV_RMSA || ENV == ENV_DEEPRMSA)) {  // _get_network_compactness before the provision (rmsa_env.py:189)
      int occ, fb;
      if (O.persistent && !M.cs_lds) { occ = atomicAdd(e.cs, 0); fb = atomicAdd(e.cs + 1, 0); }
      else { occ = e.cs[0]; fb = e.cs[1]; }
      ic->prev_comp = (fb > 0) ? ((double)occ / (double)e.s_nh) * ((double)P.E / (double)fb) : 1.0;
    }
    const int4 av = given ? *given : *(const int4*)(P.actions + env * 4);
    int path, mod = 0;
    bool bad = false;
    if (ENV == ENV_DEEPRMSA) {  // deeprmsa_env.py:48-58
      int aa = av.x;
      path = K; slot = S;
      if (O.trusted && given) {  // decoded by the in-kernel scan, on this slot map (policy_g)
        path = av.y; slot = av.z;
      } else if (aa >= 0 && aa < K * P.J) {
        int route = aa / P.J, block = aa - route * P.J;
        int start = 0;
        int pidx = pair_base(P, e.src, e.dst) + route;
        int nb = 0;
        if (route < P.n_paths[e.src * P.N + e.dst]) {
          Row<W> m = g8::path_and_global<W>(P, e, pidx);
          nb = nth_block<W>(m, S, P.nslots_path[(size_t)pidx * P.n_br + e.br_idx], block + 1, start);
        }
        if (block < nb) { path = route; slot = start; }
      }
    } else if (ENV == ENV_RMCSA) {
      path = av.x; mod = av.y; core = av.z; slot = av.w;
      bad = path < 0 || path > K || mod < 0 || mod > P.M || core < 0 || core > P.C || slot < 0 || slot > S;
    } else if (ENV == ENV_RWA) {
      path = av.x; slot = av.y;
      bad = path < 0 || path >= K + rej || slot < 0 || slot >= S + rej;
    } else {
      path = av.x; slot = av.y;
      bad = path < 0 || path > K || slot < 0 || slot > S;
    }
    if (bad) {
      e.flags |= ORL_FLAG_BAD_ACTION;
      path = K; slot = S; mod = P.M; core = P.C;
    }
    const int path0 = path, slot0 = slot, mod0 = mod, core0 = core;
    ORL_PROFA(2);
    bool accepted = false;
    int pushed_idx = -1;
    u64 pushed_info = 0ull;
    double pushed_t = 0.0;
    bool in_range = (ENV == ENV_RMCSA) ? (path < K && mod < P.M && core < P.C && slot < S) : (path < K && slot < S);
    if (in_range && path < P.n_paths[e.src * P.N + e.dst]) {
      int pidx = pair_base(P, e.src, e.dst) + path;
      if (ENV == ENV_RMCSA) n = P.nslots[e.br_idx * P.M + mod];
      else if (ENV != ENV_RWA) n = P.nslots_path[(size_t)pidx * P.n_br + e.br_idx];
      rec = path_rec_load(P, pidx);
      bool ok = O.trusted && (ENV != ENV_DEEPRMSA || given != nullptr);
      if (!ok && slot + n <= S) {  // is_path_free: lane w checks word w of every link row of the path
        const int hops = path_rec_byte(rec, 0);
        bool busy = false;
        if (gl < W) {
          const u64 m = word_range(slot - 64 * gl, slot + n - 64 * gl);
          const u64* rowbase = e.bm + (size_t)core * P.E * W + gl;
          u64 miss = 0;
          for (int h = 0; h < hops; h += 4) {  // four independent row-word loads in flight
            const u64 r0 = rowbase[path_rec_byte(rec, 2 + h) * W];
            const u64 r1 = (h + 1 < hops) ? rowbase[path_rec_byte(rec, 3 + h) * W] : ~0ull;
            const u64 r2 = (h + 2 < hops) ? rowbase[path_rec_byte(rec, 4 + h) * W] : ~0ull;
            const u64 r3 = (h + 3 < hops) ? rowbase[path_rec_byte(rec, 5 + h) * W] : ~0ull;
            miss |= m & ~(r0 & r1 & r2 & r3);
          }
          busy = miss != 0ull;
        }
        ok = gballot(busy, lane) == 0u;
      }
      if (ok && ENV == ENV_RMCSA) {
        double len = P.path_length[pidx];
        ok = (len < P.lmax_xt[mod]) && (len < P.lmax_snr[mod * P.n_br + e.br_idx]);
      }
      ORL_PROFA(3);
      if (ok) {
        const int hops = path_rec_byte(rec, 0);
        cnt = hops;
        e.s_br += e.bit_rate;
        e.s_nh += (i64)n * hops;
        if (ENV != ENV_RWA) {
          e.brp += e.bit_rate;
          e.ebrp += e.bit_rate;
          if (P.bit_rate_mode == 1 && gl == 0) P.br_hist[env * 2 * P.n_br + P.n_br + e.br_idx] += 1;
        }
        e.sa += 1;
        e.esa += 1;
        accepted = true;
        pushed_info = ev_pack(pidx, slot, n, core, e.bit_rate);
        pushed_t = e.at + e.ht;
        pushed_idx = g8::ev_push(P, e, lane, pushed_t, pushed_info, false);  // (its two stores: after the next service's loads)
        {  // the provision's rows: first mask of their items; they also count towards the per-link limit
          sink_add(sink, rec, core, slot, n, lane, true);
          if constexpr (!CP)
            for (int h = gl; h < hops; h += 8) {
              const int link = path_rec_byte(rec, 2 + h);
              atomicAdd(sink.tally + (link >> 2), 1u << (8 * (link & 3)));
            }
        }
        ORL_PROFA(4);
      }
    }
    if (ENV == ENV_RWA) { e.sp += 1; e.esp += 1; }
    if (ENV == ENV_RMCSA) { e.sp += 1; e.esp += 1; e.brq += e.bit_rate; e.ebrq += e.bit_rate; }
    if (ENV != ENV_RMCSA && P.act2d && !bad && gl == 0) act2d_count(P, env, path0, slot0, accepted);
    if (ENV == ENV_RMCSA && P.act2d && !bad && gl == 0) act4d_count(P, env, path0, mod0, core0, slot0, accepted);
    if (ENV == ENV_RWA) {  // actions_output marginals (rwa_env.py:103, 148-151)
      i64* h = P.act_hist + env * ((K + 1) + (S + 1));
      const int npa = K + rej, nsa = S + rej;
      for (int i = gl; i < npa + nsa; i += 8) {
        int hi = (i < npa) ? i : (K + 1) + (i - npa);
        bool hit = !bad && ((i < npa) ? (i == path0) : (i - npa == slot0));
        if (ic) {  // k_agent: path_action_probability / wavelength_action_probability of info (rwa_env.py:148-151): updated count / services_processed
          const i64 v = h[hi] + (hit ? 1 : 0);
          if (hit) h[hi] = v;
          P.info[env * P.n_info + 2 + i] = (double)v / (double)e.sp;
        } else if (hit) {
          h[hi] += 1;
        }
      }
    }
    if (gl == 0) {
      if (O.write_io) {
        P.reward[env] = accepted ? 1.0 : (ENV == ENV_DEEPRMSA ? -1.0 : 0.0);
        if (given) *(int4*)(P.actions + env * 4) = (ENV == ENV_DEEPRMSA) ? make_int4(av.x, 0, 0, 0) : av;
      }
      e.scal[SC_ACC] = pack2(accepted ? 1 : 0, core);
      e.scal[SC_NOWA] = (u64)__double_as_longlong(e.now);
      if (M.clk) M.clk[2 * (env - M.clk_env0)] = e.now;
    }
    ORL_PROFA(5);
    // the word service_part leaves in SC_ACC (recomputed here so that the deferral below need not read it back)
    const u64 acc_after = (accepted && ENV != ENV_RWA && e.now > 0)
                              ? (3ull | ((u64)(u32)core << 32) | ((u64)e.s_nh << 37))
                              : pack2(accepted ? 1 : 0, core);
    if (ic) {  // the counters as info sees them: after this step's decision, before the next service is counted (rmsa_env.py:234-249)
      ic->s_nh_prov = e.s_nh;
      if (gl == 0) {
        double* io = P.info + env * P.n_info;
        io[0] = (double)(e.sp - e.sa) / (double)e.sp;
        io[1] = (double)(e.esp - e.esa) / (double)e.esp;
        if (ENV != ENV_RWA) {  // (RWA: info goes on with the action probabilities, written beside the histogram update)
          io[2] = (double)(e.brq - e.brp) / (double)e.brq;
          io[3] = (double)(e.ebrq - e.ebrp) / (double)e.ebrq;
        }
        if ((ENV == ENV_RMSA || ENV == ENV_DEEPRMSA) && P.bit_rate_mode == 1) {  // discrete bit rates: blocking per rate + fairness (rmsa_env.py:217-227, 268-273)
          const i64* rq = P.br_hist + env * 2 * P.n_br;
          const i64* pv = rq + P.n_br;
          double mxv = -__builtin_inf(), mnv = __builtin_inf();
          for (int i = 0; i < P.n_br; i++) {
            double bl = 0.0;
            if (rq[i] > 0) bl = (double)(rq[i] - pv[i]) / (double)rq[i];
            io[8 + i] = bl;
            mxv = bl > mxv ? bl : mxv;
            mnv = bl < mnv ? bl : mnv;
          }
          io[8 + P.n_br] = mxv - mnv;
        }
      }
    }
    if (!O.prefetch && !svc) g8::rng_fill(e, rng, gl);
    const bool done = service_part<ENV, W>(P, e, env, lane, O.auto_reset ? 1 : 0, accepted, core, O.write_io, rng, prof, svc);
    if (done_out) *done_out = done ? 1 : 0;
    // the pending-release slot of this step's provision (the rebuild scan of the release detection must find it in memory)
    if (pushed_idx >= 0 && gl == (pushed_idx & 7)) { e.ev_time[pushed_idx] = pushed_t; e.ev_info[pushed_idx] = pushed_info; }
    desc_out = g8::env_store(P, e, gl, O.write_io);
    if (M.clk && gl == 0) M.clk[2 * (env - M.clk_env0) + 1] = e.now;
    ORL_PROFA(8);
    {
      // due releases of the step (rmsa_env.py:590-597) -> masks behind the provision's in the same table.  The env record
      // has gone back already (so that only the handful of release-related fields stays in registers through the
      // detection); those fields are written again below when the detection changed them.
      SoonRegs soon;
#ifdef ORL_DIAG_INSTEAD_OF_RELEASES
      ORL_DIAG_INSTEAD_OF_RELEASES
#else
      release_soon<ENV, W, CP>(P, e, lane, sink, soon, prof, accepted ? 1 : 0, pushed_idx, pushed_info, pre_idx, pre_info);
#endif
      if (!svc) g8::rng_commit_stores(e, rng, gl);  // the Mersenne-Twister words of next_service: behind the detection's loads
      ORL_PROFA(10);
      if (sink.deferred) {
        // more releases meet on one link than an item holds masks for: the release state stays as stored and the
        // serial path (rel_serial: k_rel_tail, or the start of the persistent kernel's next launch) releases them in
        // place after this step's items
        if (gl == 0) {
          if (!O.persistent) {  // two-kernel form: the list k_rel_tail works through
            u32* dq = P.q_def + (size_t)parity * P.q_def_stride;
            dq[16 + atomicAdd(dq, 1u)] = (u32)env;
          }
          if (s_deferred) *s_deferred = 1;  // persistent kernel: this workgroup stops after the row phase
          e.scal[SC_ACC] = acc_after | (1ull << 16);
          e.scal[SC_HINT] = pack2(e.nfree, 0);  // a rebuild may have rewritten the free-slot stack
        }
      } else {
#pragma unroll
        for (int k = 0; k < ORL_SOON_PER_LANE; k++) {
          if (e.sr_on) {
            if (soon.dirty) { e.sr_t[k] = soon.t[k]; e.sr_i[k] = soon.i[k]; }  // (dirty == 0: returned untouched)
          } else if ((soon.dirty >> k) & 1) {
            e.soon_t[gl + 8 * k] = soon.t[k];
            e.soon_i[gl + 8 * k] = (u32)soon.i[k];
          }
        }
        if (gl == 0) {
          e.scal[SC_NEXTREL] = (u64)__double_as_longlong(e.next_rel);
          e.scal[SC_TSOON] = (u64)__double_as_longlong(e.t_soon);
          e.scal[SC_SBR] = (u64)e.s_br;
          e.scal[SC_SNH] = (u64)e.s_nh;
          e.scal[SC_EV] = pack2(e.ev_hwm, e.ev_cnt);
          e.scal[SC_HINT] = pack2(e.nfree, 0);
        }
      }
    }
    if (carried) {
#pragma unroll
      for (int k = 0; k < ORL_SOON_PER_LANE; k++) { carried->t[k] = e.sr_t[k]; carried->i[k] = e.sr_i[k]; }
    }
  }
  ORL_PROFA(11);
  if constexpr (!CP) { if (O.emit_queue) emit_items(P, env, sink, true, lane, P.q_a, P.q_cnt_a); }
  if constexpr (CP) {
    if (s_list) {
      wave_fence();
      sink_compact(s_tab + P.E * 8 * (int)(threadIdx.x >> 6), P.E, lane, s_list, s_list_n);
    }
  }
  ORL_PROFA(9);
  return desc_out;
}

// ---------------------------------------------------------------------------------------------------------------
// Deferred statistics (round 5).  Most of what ctrl_a does per env is bookkeeping that nothing in the loop reads back: the
// eight service / bit-rate counters, the running averages of network throughput and compactness, the done / soft-reset
// logic on them, and the 32-word record that carries them in and out of registers every step — the same arithmetic on all
// 8 lanes of a group, ~350 VALU instructions and 31 loads + 13 stores per wavefront-step for 8 envs.  The dynamics of an env
// (slot maps, pending releases, clock, generator) never depend on it.  So the persistent kernel does
// not do it at all: ctrl_d keeps the clock, the pending service and the release queue's fields, and LOGS per env-step the
// three words the bookkeeping needs (DevParams::slog); k_stats (orl_kernels.hip) replays the log after the launch with one
// LANE per env — 64 envs per instruction instead of 8 — in the reference's operation order (rmsa_env.py:163-282, 439-462,
// 545-597), leaving the record exactly as ctrl_a would have.  Log words of a step:
//   w0  the clock after the step's next service was created (float64 bits)
//   w1  accepted:1 | n x hops of the provision:12 | bit-rate index of the NEW service:12 | the per-core sums at the START of the
//       step minus what the previous step's releases added — the network compactness right after the previous step's provision —
//       (lambda_max - lambda_min) sum:17 | free blocks inside:16   (RWA instead of the sums: the action's path:4 | wavelength:10,
//       for the actions_output marginals) | core of the provision:5 (RMCSA; the sums logged are those of the core the env's
//       previous accepted provision went to)
//   (rows-deferred forms: the sums fields of w1 stay zero — k_rowstats hands them to k_stats through DevParams::ssum)
//   w2  what the step's releases take off the sums: n x hops:20 | bit rate:24   (stored after the release detection; w0 and w1
//       before it, so that nothing of them is live across it)
// Slot n of a wavefront that logged n steps carries w1's sums only: those after its last row phase.
// ---------------------------------------------------------------------------------------------------------------
#define ORL_SLOG_WORDS ORL_SLOG_ROW_WORDS
__device__ __forceinline__ u64 slog_w1(bool accepted, int n_hops, int br_new, int occ, int fb, int core = 0) {
  return (u64)(accepted ? 1u : 0u) | ((u64)(u32)n_hops << 1) | ((u64)(u32)br_new << 13) | ((u64)(u32)occ << 25) | ((u64)(u32)fb << 42) |
         ((u64)(u32)core << 58);
}
__device__ __forceinline__ u64 slog_w1_rwa(bool accepted, int n_hops, int path, int slot) {
  return (u64)(accepted ? 1u : 0u) | ((u64)(u32)n_hops << 1) | ((u64)(u32)path << 25) | ((u64)(u32)slot << 29);
}
__device__ __forceinline__ u64 slog_w2(int d_nh, int d_br) { return (u64)(u32)d_nh | ((u64)(u32)d_br << 20); }

// The six record words ctrl_d reads and writes every step — clock, the pending service's holding time, the release queue's
// bound, horizon, window and free-slot count — live in the wavefront's LDS window for the launch where the records themselves
// stay in global memory (Wmem::mini: 7 words per env, the seventh for the banks): six loads and nine stores per wavefront-step
// that went through L2 in the step's dependent chain become LDS accesses or disappear (the pending service's source /
// destination / bit rate are the descriptor's: written to the record when the launch ends, svc_words).  The free-slot stack,
// the flags and the statistics' words stay where the record is.  448 bytes: with them cfg2's window keeps its eight LDS pieces.
#define ORL_MINI_WORDS 6
#define ORL_MINI_STRIDE 7
__device__ __forceinline__ int mini_slot(int k) {  // record slot of mini word k
  return k == 0 ? SC_NOW : k == 1 ? SC_HT : k == 2 ? SC_NEXTREL : k == 3 ? SC_TSOON : k == 4 ? SC_EV : SC_HINT;
}
__device__ __forceinline__ int mini_index(int slot) {
  return slot == SC_NOW ? 0 : slot == SC_HT ? 1 : slot == SC_NEXTREL ? 2 : slot == SC_TSOON ? 3 : slot == SC_EV ? 4 : 5;
}
// SC_SRC_DST and SC_BR_IDX of the pending service from its descriptor (pair base, bit-rate index): once per launch
template <int ENV> __device__ __forceinline__ void svc_words(const DevParams& P, u64 desc, u64& sd, u64& br) {
  const int pair = (int)(u32)desc / P.K, src = pair / P.N, dst = pair - src * P.N;
  const int br_idx = (ENV == ENV_RWA) ? 0 : (int)((desc >> 32) & 0xffffu);
  const int bit_rate = (ENV == ENV_RWA) ? 0 : ((P.bit_rate_mode == 0) ? P.br_lo + br_idx : P.bit_rates[br_idx]);
  sd = pack2(src, dst);
  br = pack2(bit_rate, br_idx);
}
// (MINI is a template parameter: a pointer chosen at run time between the LDS window and global memory would be a flat one)
template <bool MINI> __device__ __forceinline__ u64* mrec(const Wmem& M, u64* rec, i64 env, int slot) {
  if constexpr (MINI) return M.mini + (env - M.mini_env0) * ORL_MINI_STRIDE + mini_index(slot);
  else return rec + slot;
}
// the control phase of the persistent kernel without the bookkeeping (services drawn ahead; CP: the single-core families'
// compact sink, else RMCSA's entries with a core per mask).  `esp`: the env's episode step counter, kept by the caller for the
// whole launch (done / observation need it); `prev_core` (RMCSA): the core of the env's last accepted provision — the sums
// logged are that core's; `slog`: this step's log slot, at the env's column.  Returns the descriptor of the new pending service.
// RW (the two-wavefront form, k_persist): this wavefront applies the masks to the slot maps itself as it appends them (LDS
// atomics without a return value) and hands the row wavefront only the statistics; `rw_sync` are the pair's counters — [1] steps
// whose items the row wavefront has read (sink table, mask table, clocks and rows: they may be overwritten), [2] steps whose
// statistics are complete (the sums) — and `rw_k` the number of steps handed over so far.
// The hand-over between the two wavefronts of the pair form goes through LDS only: release / acquire at workgroup scope restricted to
// the LOCAL address space (clang's address-space MMRA on the fence builtin) — on gfx950 an s_waitcnt lgkmcnt(0) before the counter's
// store, and none of the vmcnt wait a full workgroup-scope release would add for global stores nobody in the pair reads.
#ifdef ORL_RW_FENCE_OLD  // (A/B: round 5's wavefront-scope fences)
__device__ __forceinline__ void rw_release_lds() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); }
__device__ __forceinline__ void rw_acquire_lds() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); }
#else
__device__ __forceinline__ void rw_release_lds() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); }
__device__ __forceinline__ void rw_acquire_lds() { __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); }
#endif
__device__ __forceinline__ void rw_wait_for(const u32* p, u32 want) {
  while ((u32)__builtin_amdgcn_readfirstlane((int)__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) < want) __builtin_amdgcn_s_sleep(1);
  rw_acquire_lds();
  ORL_DIAG_JITTER();
}
// RD (rows deferred, round 6): as RW, this phase changes the slot maps itself — but there is no row phase at all in the loop: every
// provision and release is logged as an event (SinkT<CP, true>, DevParams::elog; `ecur`: the env's event count so far in this
// launch, `t_log`: the step's number within the launch) and k_rowstats replays the link statistics and the compactness sums after
// the launch; the sums fields of log word w1 are left zero here and filled in by that kernel before k_stats reads them.
template <int ENV, int W, bool CP, bool MINI, bool RW = false, bool RD = false>
__device__ __forceinline__ u64 ctrl_d(const DevParams& P, const Wmem& M, const CtrlOpts& O, i64 env, bool valid, int lane, Prof& prof,
                                      const int4& av, u64 desc, typename SinkEntryOf<CP>::type* s_tab, u32* s_tally, int tw, int* s_deferred,
                                      int* done_out, unsigned short* s_list, u32* s_list_n, SoonRegs* carried, unsigned short* s_mtab,
                                      SvcBuf& svc, int& esp, int& prev_core, u64* slog, const ScanHand* hand = nullptr, int pop_pre = -2,
                                      const u32* rw_sync = nullptr, u32 rw_k = 0u, int* ecur = nullptr, int t_log = 0) {
  static_assert(!RD || (CP && !RW), "rows deferred: single-core families, one wavefront per 8 envs");
  // `hand`: the chosen path's slot count and record from the scan's winning lane (single-core families); `pop_pre`: the top entry
  // of the env's free-slot stack (-1: empty), requested by the caller before the scan (-2: not given) — with both, nothing the
  // decision needs is fetched from global memory behind the scan
  const int K = P.K, S = P.S, gl = lane & 7;
  u64 desc_out = 0ull;
  SinkT<CP, RD> sink;
  sink.tab = nullptr; sink.tally = nullptr; sink.tw = tw; sink.active = false; sink.deferred = false; sink.cnt = 0;
  sink.list = s_list; sink.list_n = s_list_n; sink.mtab = nullptr; sink.nrel = 0; sink.rows = nullptr; sink.roww = 0;
  sink.ev = nullptr; sink.ev_at = 0; sink.ev_rel0 = 0; sink.ev_t = t_log; sink.ev_clock = 0.0;
  if constexpr (RW) { rw_wait_for(rw_sync + 1, rw_k); ORL_PROFA(14); }  // (the row wavefront has read the previous step's tables and rows)
  if (!RW && !RD && lane == 0) *s_list_n = 0u;  // (RW: the list is the row wavefront's)
  if constexpr (!RD) {
    typename SinkEntryOf<CP>::type* tb = s_tab + P.E * 8 * (int)(threadIdx.x >> 6);
    if constexpr (!CP) {
      u32* ty = s_tally + tw * 8 * (int)(threadIdx.x >> 6);
      for (int i = lane; i < 8 * tw; i += 64) ty[i] = 0u;
      sink.tally = s_tally + tw * (int)(threadIdx.x >> 3);
    } else {
      sink.mtab = s_mtab + ORL_MTAB * (int)(threadIdx.x >> 3);
    }
    for (int i = lane; i < 8 * P.E; i += 64) sink_entry_clear(tb[i]);
    wave_fence();
    sink.tab = s_tab + P.E * (int)(threadIdx.x >> 3);
  }
  if (valid) {
    u64* rec = wm_scal(P, M, env);
    EnvG e;
    e.scal = rec;
    e.env = env;
    // what the loop itself needs of the record: clock, the pending service's holding time, the release queue
    {
      const u64 w_now = *mrec<MINI>(M, rec, env, SC_NOW), w_ht = *mrec<MINI>(M, rec, env, SC_HT), w_nr = *mrec<MINI>(M, rec, env, SC_NEXTREL);
      const u64 w_ts = *mrec<MINI>(M, rec, env, SC_TSOON), w_ev = *mrec<MINI>(M, rec, env, SC_EV), w_hint = *mrec<MINI>(M, rec, env, SC_HINT);
      e.now = __longlong_as_double((i64)w_now); e.ht = __longlong_as_double((i64)w_ht);
      e.next_rel = __longlong_as_double((i64)w_nr); e.t_soon = __longlong_as_double((i64)w_ts);
      // (the pending service's bit rate from its descriptor: no load)
      e.br_idx = (int)((desc >> 32) & 0xffffu);
      e.bit_rate = (ENV == ENV_RWA) ? 0 : ((P.bit_rate_mode == 0) ? P.br_lo + e.br_idx : P.bit_rates[e.br_idx]);
      e.ev_hwm = (int)(u32)w_ev; e.ev_cnt = (int)(w_ev >> 32);
      e.nfree = (int)(u32)w_hint;
      if (pop_pre != -2) {
        e.pop_idx = pop_pre;
      } else {
        const u64 f0 = rec[SC_FREE0], f1 = rec[SC_FREE1], f2 = rec[SC_FREE2], f3 = rec[SC_FREE3];
        const int top = e.nfree - 1;
        const u64 w = (top >> 2) == 0 ? f0 : (top >> 2) == 1 ? f1 : (top >> 2) == 2 ? f2 : f3;
        e.pop_idx = (top >= 0) ? (int)((w >> (16 * (top & 3))) & 0xffffu) : -1;
      }
    }
    e.flags = 0;
    e.s_br = 0; e.s_nh = 0;  // (here: minus what this step's releases take off the sums)
    e.ev_time = M.evl0 ? M.evl0 + (env - M.env0) * P.ev_cap : P.ev_time + env * P.ev_cap;
    e.ev_info = P.ev_info + env * P.ev_cap;
    e.soon_t = P.soon_t + env * ORL_SOON;
    e.soon_i = P.soon_i + env * ORL_SOON;
    e.sr_on = false;
    e.rank_pairs = false;
    if (carried) {
      e.sr_on = true;
      e.rank_pairs = O.rank_pairs;
#pragma unroll
      for (int k = 0; k < ORL_SOON_PER_LANE; k++) { e.sr_t[k] = carried->t[k]; e.sr_i[k] = carried->i[k]; }
    }
    int pre_idx = -1;
    u64 pre_info = 0ull;
    if (carried && O.prefetch) {  // (as ctrl_a: the info word of this lane's earliest list entry, if it may come due in this step)
      double pt = e.sr_t[0];
      int pi = e.sr_i[0];
#pragma unroll
      for (int k = 1; k < ORL_SOON_PER_LANE; k++)
        if (e.sr_t[k] < pt || (e.sr_t[k] == pt && e.sr_i[k] < pi)) { pt = e.sr_t[k]; pi = e.sr_i[k]; }
      if (pt <= e.now + P.pf_window && e.t_soon > -__builtin_inf() && (u32)pi < (u32)P.ev_cap) { pre_idx = pi; pre_info = e.ev_info[pi]; }
    }
    e.bm = wm_bm(P, M, env);
    e.ls = wm_ls(P, M, env);
    e.cs = wm_cs(P, M, env);
    if constexpr ((RW || RD) && CP) { sink.rows = e.bm; sink.roww = W; }
    if constexpr (RD) { sink.ev = P.elog + env * (i64)(2 * P.elog_cap); sink.ev_at = *ecur; sink.ev_rel0 = *ecur; sink.ev_clock = e.now; }
    int occ_s = 0, fb_s = 0;
    const int pc_s = (ENV == ENV_RMCSA) ? prev_core : 0;
    auto read_sums = [&]() {
      if (ENV != ENV_RWA && !RD) {
        // the sums right after the previous step's provision — of the core it went to — (its pending network-compactness update,
        // rmsa_env.py:439-462, is finished by the replay from them); this step's releases start from zero
        int* rs = e.cs + 2 * P.C;
        const int pc = pc_s;
        if (!M.cs_lds) {
          occ_s = atomicAdd(e.cs + 2 * pc, 0) - atomicAdd(rs + 2 * pc, 0);
          fb_s = atomicAdd(e.cs + 2 * pc + 1, 0) - atomicAdd(rs + 2 * pc + 1, 0);
          for (int i = gl; i < 2 * P.C; i += 8) atomicExch(rs + i, 0);
        } else {
          occ_s = e.cs[2 * pc] - rs[2 * pc];
          fb_s = e.cs[2 * pc + 1] - rs[2 * pc + 1];
          for (int i = gl; i < 2 * P.C; i += 8) rs[i] = 0;
        }
      }
    };
    if constexpr (!RW) read_sums();  // (RW: at the end of the phase, when the row wavefront has long finished the previous step)
    int path, slot, mod = 0, core = 0;
    if (ENV == ENV_DEEPRMSA) { path = av.y; slot = av.z; }  // (decoded by the in-kernel scan on this slot map, policy_g)
    else if (ENV == ENV_RMCSA) { path = av.x; mod = av.y; core = av.z; slot = av.w; }
    else { path = av.x; slot = av.y; }
    const int path0 = path, slot0 = slot;
    ORL_PROFA(2);
    const int pb = (int)(u32)desc, np_ = (int)((desc >> 48) & 0xffu);
    bool accepted = false;
    int n = 1, n_hops = 0;
    int pushed_idx = -1;
    u64 pushed_info = 0ull;
    double pushed_t = 0.0;
    const bool in_range = (ENV == ENV_RMCSA) ? (path < K && mod < P.M && core < P.C && slot < S) : (path < K && slot < S);
    if (in_range && path < np_) {
      const int pidx = pb + path;
      PathRec prec;
      if (ENV != ENV_RMCSA && hand) {  // (the scan's winning lane had both)
        n = hand->n;
        prec.q[0] = hand->q[0]; prec.q[1] = hand->q[1]; prec.q[2] = hand->q[2]; prec.q[3] = hand->q[3];
      } else {
        if (ENV == ENV_RMCSA) n = P.nslots[e.br_idx * P.M + mod];
        else if (ENV != ENV_RWA) n = P.nslots_path[(size_t)pidx * P.n_br + e.br_idx];
        prec = path_rec_load(P, pidx);
      }
      bool ok = true;
      if (ENV == ENV_RMCSA) {  // _crosstalk_is_acceptable: the two reach limits (rmcsa_env.py:341-384), as ctrl_a
        const double len = P.path_length[pidx];
        ok = (len < P.lmax_xt[mod]) && (len < P.lmax_snr[mod * P.n_br + e.br_idx]);
      }
      ORL_PROFA(3);
      if (ok) {
        const int hops = path_rec_byte(prec, 0);
        n_hops = n * hops;
        accepted = true;
        if constexpr (RD) sink.ev_rel0 = sink.ev_at + 1;
        pushed_info = ev_pack(pidx, slot, n, core, e.bit_rate);
        pushed_t = e.now + e.ht;  // (arrival time + holding time: the clock stands at the pending service's arrival)
#ifndef ORL_DIAG_NO_PUSH
        pushed_idx = g8::ev_push(P, e, lane, pushed_t, pushed_info, false);
#endif
        sink_add(sink, prec, core, slot, n, lane, true);
        if constexpr (!CP)
          for (int h = gl; h < hops; h += 8) {
            const int link = path_rec_byte(prec, 2 + h);
            atomicAdd(sink.tally + (link >> 2), 1u << (8 * (link & 3)));
          }
        ORL_PROFA(4);
      }
    }
    if (ENV != ENV_RMCSA && P.act2d && gl == 0) act2d_count(P, env, path0, slot0, accepted);
    if (ENV == ENV_RMCSA && P.act2d && gl == 0) act4d_count(P, env, path0, mod, core, slot0, accepted);
    if (gl == 0) {
      if (O.write_io) {
        P.reward[env] = accepted ? 1.0 : (ENV == ENV_DEEPRMSA ? -1.0 : 0.0);
        *(int4*)(P.actions + env * 4) = (ENV == ENV_DEEPRMSA) ? make_int4(av.x, 0, 0, 0) : av;
      }
      if constexpr (!RD) M.clk[2 * (env - M.clk_env0)] = e.now;  // (SC_NOWA is the replay's: every form with this control phase has the clock pair)
    }
    ORL_PROFA(5);
    // the next service, drawn ahead by svc_generate: from the lane of the group that holds it
    int br_new;
    {
      const int k = svc.cnt & 0xff;
      const double q = gget(svc.q, k, lane), ht = gget(svc.ht, k, lane);
      const u32 pk = gget(svc.pk, k, lane);
      svc.cnt += 1;
      e.now = e.now + q;
      br_new = (int)(pk >> 20);
      int bit_rate = 0;
      if (ENV != ENV_RWA) bit_rate = (P.bit_rate_mode == 0) ? P.br_lo + br_new : P.bit_rates[br_new];
      const int src = (int)(pk & 0x3ffu), dst = (int)((pk >> 10) & 0x3ffu);
      const u64 npn = (u64)(u32)P.n_paths[src * P.N + dst];
      desc_out = (u64)(u32)((src * P.N + dst) * K) | ((u64)(u32)(ENV != ENV_RWA ? br_new : 0) << 32) | (npn << 48);
      if (gl == 0) {
        *mrec<MINI>(M, rec, env, SC_NOW) = (u64)__double_as_longlong(e.now);
        if (!MINI) rec[SC_AT] = (u64)__double_as_longlong(e.now);  // (with the words in LDS: written back with them at the end of the launch)
        *mrec<MINI>(M, rec, env, SC_HT) = (u64)__double_as_longlong(ht);
        if (!MINI) {  // (MINI: the caller writes them from the descriptor when the launch ends)
          rec[SC_SRC_DST] = pack2(src, dst);
          rec[SC_BR_IDX] = pack2(bit_rate, ENV != ENV_RWA ? br_new : 0);
        }
        if (O.write_io) P.svc_desc[env] = desc_out;
      }
    }
    if constexpr (!RW) {
      if (gl < 2) {
        const u64 w1 = (ENV == ENV_RWA) ? slog_w1_rwa(accepted, n_hops, path0, slot0) : slog_w1(accepted, n_hops, br_new, occ_s, fb_s, core);
        slog[(size_t)gl * (size_t)P.log_stride] = gl == 0 ? (u64)__double_as_longlong(e.now) : w1;
      }
    } else if (gl == 0) {
      slog[0] = (u64)__double_as_longlong(e.now);
    }
    if (ENV == ENV_RMCSA && accepted) prev_core = core;
    ORL_PROFA(7);
    // episode end (rmsa_env.py:263, 310-315: the soft reset re-counts the pending service; rwa_env.py:141, rmcsa_env.py:294: RWA
    // and RMCSA count at the decision)
    esp += 1;
    const bool done = (esp == P.episode_length);
    if (done) esp = (ENV == ENV_RWA) ? 0 : 1;
    if (gl == 0 && O.write_io) P.done[env] = done ? 1 : 0;
    if (done_out) *done_out = done ? 1 : 0;
    if (pushed_idx >= 0 && gl == (pushed_idx & 7)) { e.ev_time[pushed_idx] = pushed_t; e.ev_info[pushed_idx] = pushed_info; }
    if constexpr (!RD) { if (gl == 0) M.clk[2 * (env - M.clk_env0) + 1] = e.now; }
    ORL_PROFA(8);
    {
      SoonRegs soon;
#ifdef ORL_DIAG_INSTEAD_OF_RELEASES
      ORL_DIAG_INSTEAD_OF_RELEASES
#else
      release_soon<ENV, W, CP, RD>(P, e, lane, sink, soon, prof, accepted ? 1 : 0, pushed_idx, pushed_info, pre_idx, pre_info);
#endif
      if constexpr (RD) *ecur = sink.ev_rel0 + sink.nrel;
      ORL_PROFA(10);
      if (sink.deferred) {
        // (as ctrl_a) the releases stay pending; rel_serial does them in place at the start of this wavefront's next launch
        if (gl == 0) {
          *s_deferred = 1;
          rec[SC_ACC] = rec[SC_ACC] | (1ull << 16);
        }
      } else {
#pragma unroll
        for (int k = 0; k < ORL_SOON_PER_LANE; k++) {
          if (e.sr_on) {
            if (soon.dirty) { e.sr_t[k] = soon.t[k]; e.sr_i[k] = soon.i[k]; }
          } else if ((soon.dirty >> k) & 1) {
            e.soon_t[gl + 8 * k] = soon.t[k];
            e.soon_i[gl + 8 * k] = (u32)soon.i[k];
          }
        }
      }
      if (gl == 0) {
        *mrec<MINI>(M, rec, env, SC_NEXTREL) = (u64)__double_as_longlong(e.next_rel);
        *mrec<MINI>(M, rec, env, SC_TSOON) = (u64)__double_as_longlong(e.t_soon);
        *mrec<MINI>(M, rec, env, SC_EV) = pack2(e.ev_hwm, e.ev_cnt);
        *mrec<MINI>(M, rec, env, SC_HINT) = pack2(e.nfree, 0);
        if (e.flags) rec[SC_FLAGS] = rec[SC_FLAGS] | ((u64)(u32)e.flags << 32);
      }
    }
    if (gl == 2) slog[2 * (size_t)P.log_stride] = slog_w2((int)(-e.s_nh), (int)(-e.s_br));
    if constexpr (RW) {
      rw_wait_for(rw_sync + 2, rw_k);
      ORL_PROFA(15);
      read_sums();
      if (gl == 1)
        slog[(size_t)P.log_stride] = (ENV == ENV_RWA) ? slog_w1_rwa(accepted, n_hops, path0, slot0) : slog_w1(accepted, n_hops, br_new, occ_s, fb_s, core);
    }
    if (carried) {
#pragma unroll
      for (int k = 0; k < ORL_SOON_PER_LANE; k++) { carried->t[k] = e.sr_t[k]; carried->i[k] = e.sr_i[k]; }
    }
  }
  if constexpr (CP && !RW && !RD) {  // (RW: the row wavefront makes its list from the table itself)
    wave_fence();
    sink_compact(s_tab + P.E * 8 * (int)(threadIdx.x >> 6), P.E, lane, s_list, s_list_n);
  }
  ORL_PROFA(11);
  return desc_out;
}

// persistent kernel: an item of the row phase read from the sink table in place (RMCSA; the single-core families hand the
// entry's bit word and the env's mask table to row_item_lane1)
__device__ __forceinline__ Item item_from_sink(i64 env, int link, const SinkEntry& t) {
  const int nm = (int)((t.crn >> 40) & 15);
  return make_item(env, (u32)link, nm, t.mk0, nm > 4 ? t.mk1 : 0ull, t.crn & 0xffffffffffull, 1 | (int)((t.crn >> 44) & 1) << 1);
}

// ---------------------------------------------------------------------------------------------------------------
// what step() does after the provision, except the link rows: network statistics, next service, done / auto reset
// ---------------------------------------------------------------------------------------------------------------
template <int ENV, int W>
__device__ __forceinline__ bool service_part(const DevParams& P, EnvG& e, i64 env, int lane, int auto_reset, bool accepted, int core,
                                             bool write_io, g8::RngG& rng, Prof& prof, SvcBuf* svc) {
  const int gl = lane & 7;
  if (accepted && ENV != ENV_RWA) {  // _update_network_stats (rmsa_env.py:439-462)
    double last_update = e.g_last, time_diff = e.now - last_update;
    if (e.now > 0) {
      double cur_thr = (double)e.s_br;
      e.g_thr = div_pos((e.g_thr * last_update) + (cur_thr * time_diff), e.now);
      // the compactness term needs the sums after the provision's row updates: the next step finishes
      // g_comp = (g_comp * last_update + compactness * time_diff) / now from these two stashed factors
      if (gl == 0) {
        e.scal[SC_GC_A] = (u64)__double_as_longlong(e.g_comp * last_update);
        e.scal[SC_GC_TD] = (u64)__double_as_longlong(time_diff);
        e.scal[SC_ACC] = 3ull | ((u64)(u32)core << 32) | ((u64)e.s_nh << 37);  // s_nh at provision time (< 2^27)
      }
    }
    e.g_last = e.now;
  }
  e.new_service = 0;
  ORL_PROFA(6);
  if (svc) {
    // _next_service with the draws done ahead (svc_generate): the service comes from the lane of the group that holds it
    if (!e.new_service) {
      const int k = svc->cnt & 0xff;
      const double q = gget(svc->q, k, lane), ht = gget(svc->ht, k, lane);
      const u32 pk = gget(svc->pk, k, lane);
      svc->cnt += 1;
      const double at = e.now + q;
      e.now = at;
      const int br_idx = (int)(pk >> 20);
      int bit_rate = 0;
      if (ENV != ENV_RWA) bit_rate = (P.bit_rate_mode == 0) ? P.br_lo + br_idx : P.bit_rates[br_idx];
      e.id = (int)e.esp;
      e.src = (int)(pk & 0x3ffu); e.dst = (int)((pk >> 10) & 0x3ffu); e.at = at; e.ht = ht;
      e.bit_rate = bit_rate; e.br_idx = (ENV != ENV_RWA) ? br_idx : 0;
      e.new_service = 1;
      if (ENV == ENV_RMSA || ENV == ENV_DEEPRMSA) { e.sp += 1; e.esp += 1; }
      if (ENV != ENV_RWA) {
        e.brq += bit_rate;
        e.ebrq += bit_rate;
        if (P.bit_rate_mode == 1 && gl == 0) P.br_hist[e.env * 2 * P.n_br + br_idx] += 1;
      }
    }
  } else {
    g8::next_service<ENV, W>(P, e, lane, rng);  // the due releases are release_soon's job
  }
  ORL_PROFA(7);
  bool done = (e.esp == (i64)P.episode_length);
  if (done && P.ep_log && gl == 0) episode_log(P, env, e.esa);
  if (done && auto_reset) {
    e.ebrq = 0; e.ebrp = 0; e.esp = 0; e.esa = 0;
    if (ENV != ENV_RWA && e.new_service) { e.esp += 1; e.ebrq += e.bit_rate; }
  }
  if (gl == 0 && write_io) P.done[env] = done ? 1 : 0;
  return done;
}

// ---------------------------------------------------------------------------------------------------------------
// Due releases through the SOON LIST.  Scanning all ~330 pending release times of an env every step was the largest
// cost of control kernel B2 (41 dependent-latency loads per lane).  Each env keeps its earliest pending releases in
// a small list (ORL_SOON_PER_LANE per lane of its group) with the invariant "every pending release earlier than
// t_soon is in the list"; while now < t_soon the due releases are found by looking at the lane's own list slots, and
// only when the clock passes t_soon the list is rebuilt from a full scan.  Pushes keep the invariant (g8::ev_push).
// ---------------------------------------------------------------------------------------------------------------
// The list is returned in registers: the caller writes it back (when dirty) after everything that still loads.
// an opaque use of every value of a batch of requests at one point: the requests are all issued before it, one wait covers them
template <int N> __device__ __forceinline__ void scan_batch_fence(double (&t)[N]) {
  if constexpr (N == 4) asm volatile("" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]));
  else if constexpr (N == 6) asm volatile("" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]));
  else if constexpr (N == 8) asm volatile("" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]));
  else if constexpr (N == 12) {
    asm volatile("" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]), "+v"(t[8]), "+v"(t[9]), "+v"(t[10]), "+v"(t[11]));
  } else if constexpr (N == 16) {
    asm volatile("" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]), "+v"(t[8]), "+v"(t[9]), "+v"(t[10]), "+v"(t[11]),
                 "+v"(t[12]), "+v"(t[13]), "+v"(t[14]), "+v"(t[15]));
  }
}
template <int ENV, int W, bool CP, bool RD>
// `pre_idx / pre_info`: a release slot of this LANE's list whose info word the caller requested at the start of the step.
// `extra`: masks this step already put on links (its provision, two-kernel pipeline); `pushed_idx / pushed_info`: the
// release slot the same kernel has just written — its info word is taken from registers, not re-read through memory.
__device__ __forceinline__ void release_soon(const DevParams& P, EnvG& e, int lane, SinkT<CP, RD>& sink, SoonRegs& out, Prof& prof,
                                             int extra, int pushed_idx, u64 pushed_info, int pre_idx, u64 pre_info) {
  constexpr int NS = ORL_SOON_PER_LANE;
  out.dirty = 0;
  // The rebuild scan costs the wavefront the same whether one of its 8 envs runs it or all of them (the other lanes
  // idle meanwhile), so when any env's horizon has passed, every env of the wavefront rebuilds: their horizons
  // stay in phase and the wavefront pays for a scan far less often.
  const bool sync_rebuild = __ballot(!(e.now < e.t_soon) && !(e.next_rel > e.now)) != 0ull;
  if (e.next_rel > e.now && !sync_rebuild) return;
  const int gl = lane & 7;
  const double INF = __builtin_inf();
  double st[NS];
  int si[NS];
#pragma unroll
  for (int k = 0; k < NS; k++) {
    st[k] = e.sr_on ? e.sr_t[k] : e.soon_t[gl + 8 * k];
    si[k] = e.sr_on ? e.sr_i[k] : (int)e.soon_i[gl + 8 * k];
  }
  int dirty = 0;  // which of this lane's entries changed (only those go back to memory: a release touches one)
  ORL_PROF(4);
  for (int round = 0; round < 64; round++) {
    int due_all = -1;  // number of due entries overall, known when this round scanned everything
    if (!(e.now < e.t_soon) || (round == 0 && sync_rebuild)) {
      // rebuild: the NS earliest pending releases among this lane's slots (i % 8 == lane).  The selection runs on
      // 32-bit keys — quantised time-to-release (1/4096 time unit, saturating) above the slot's ordinal in the lane —
      // so that keeping the NS + 1 smallest is a branch-free min/max chain per slot; 64-bit compare-and-select
      // insertion made the rebuilding wavefronts, which set this kernel's duration, compute-bound.  The exact times of
      // the chosen slots are re-read afterwards, and the horizon is a lower bound taken from the (NS+1)-th key (when
      // it does not lie beyond the clock, the loop below releases what the list holds and rebuilds again).
      u32 kb[NS + 1];
#pragma unroll
      for (int k = 0; k <= NS; k++) kb[k] = 0xffffffffu;
      int nd = 0, top = -1, h0 = 0x7fffffff, h1 = 0x7fffffff;  // h0 < h1: the first two empty slots this lane sees
      const int hwm = e.ev_hwm;
      for (int base = gl, ord = 0; base < hwm; base += 8 * ORL_SCAN_BATCH, ord += ORL_SCAN_BATCH) {
        double tt[ORL_SCAN_BATCH];
#pragma unroll
        for (int k = 0; k < ORL_SCAN_BATCH; k++) {
          const int i = base + 8 * k;
          tt[k] = e.ev_time[i < hwm ? i : hwm - 1];
        }
        // (all of the batch's requests in flight before the first result is looked at: left to the register allocator, the
        // 128-VGPR forms came out as load - wait - load - wait, one dependent round trip per release time instead of one per batch —
        // 44 instead of 6 for an env's ~350 pending releases, the 26 us a rebuilding wavefront of k_agent spent in this scan)
        scan_batch_fence(tt);
#pragma unroll
        for (int k = 0; k < ORL_SCAN_BATCH; k++) {
          const int i = base + 8 * k;
          tt[k] = (i < hwm) ? tt[k] : INF;
        }
#pragma unroll
        for (int k = 0; k < ORL_SCAN_BATCH; k++) {
          const double t = tt[k];
          const int i = base + 8 * k;
          nd += (t <= e.now) ? 1 : 0;
          const bool empty = (t == INF);
          top = empty ? top : i;  // i grows along the scan; slots beyond ev_hwm read as INF
          h1 = (empty && h0 != 0x7fffffff && h1 == 0x7fffffff) ? i : h1;
          h0 = (empty && h0 == 0x7fffffff) ? i : h0;
          // time to release in 1/4096 units, offset by 2^19 so that overdue entries keep their order (a long
          // inter-arrival gap leaves a dozen releases due at once), saturating at 2^23 - 1 (and for empty slots)
          u32 q = (u32)__builtin_fmax(__builtin_fmin((t - e.now) * 4096.0 + 524288.0, 8388607.0), 0.0);
          q = empty ? 8388607u : q;
          u32 x = (q << 8) | (u32)(ord + k);  // 8 ordinal bits: ev_cap <= 2048 (checked on the host)
#pragma unroll
          for (int j = 0; j <= NS; j++) {  // sorted insertion: kb[] ascending
            const u32 lo = kb[j] < x ? kb[j] : x;
            x = kb[j] < x ? x : kb[j];
            kb[j] = lo;
          }
        }
      }
      // exact times of the NS chosen slots (just read: cache hits), in key order; equal keys keep slot order
      double bt[NS];
      int bi[NS];
#pragma unroll
      for (int k = 0; k < NS; k++) {
        const bool any = (kb[k] >> 8) < 8388607u;
        bi[k] = gl + 8 * (int)(kb[k] & 255u);
        bt[k] = any ? e.ev_time[bi[k]] : INF;
      }
      // every slot this lane did not choose is no earlier than its (NS+1)-th key; two quanta lower covers the
      // rounding of (t - now) * 4096 and of the sum below
      const u32 qn = kb[NS] >> 8;
      const double T_lane = (qn >= 8388607u) ? INF : (qn < 2u ? -INF : e.now + ((double)qn - 524290.0) * (1.0 / 4096.0));
      const double T = g8::g8_min(T_lane);
#pragma unroll
      for (int k = 0; k < NS; k++) { st[k] = bt[k] < T ? bt[k] : INF; si[k] = bi[k]; }
      e.t_soon = T;
      dirty = (1 << NS) - 1;
      due_all = g8_sum(nd);
      // the scan saw every slot: shrink the window to the highest occupied one (a stale larger window is harmless, so
      // the steps between two rebuilds do not bother)
      e.ev_hwm = g8_max(top) + 1;
      {
        // ... and refill the free-slot stack from what the scan saw: up to two empty slots per lane below the new window
        const bool v0 = h0 < e.ev_hwm, v1 = h1 < e.ev_hwm;
        const u32 b0 = gballot(v0, lane), b1 = gballot(v1, lane);
        unsigned short* fs = (unsigned short*)(e.scal + SC_FREE0);
        if (v0) fs[__popc(b0 & ((1u << gl) - 1u))] = (unsigned short)h0;
        if (v1) fs[__popc(b0) + __popc(b1 & ((1u << gl) - 1u))] = (unsigned short)h1;
        e.nfree = __popc(b0) + __popc(b1);
      }
      ORL_PROF(5);
    }
    // this lane's candidate: the earliest due entry of its list slots (equal times: lower event slot first)
    int ndl = 0;
    double ct = INF;
    int ci = 0x7fffffff, ck = 0;
#pragma unroll
    for (int k = 0; k < NS; k++) {
      const bool due = st[k] <= e.now;
      ndl += due ? 1 : 0;
      if (due && (st[k] < ct || (st[k] == ct && si[k] < ci))) { ct = st[k]; ci = si[k]; ck = k; }
    }
    const int tot = g8_sum(ndl);
    if (tot == 0) {
      if (e.now < e.t_soon) break;
      if (round > 0 || due_all >= 0) {  // nothing below the horizon is due although the clock passed it: equal times
        if (!sink.active) { ORL_DBG(8, 1); ORL_DBG(9, e.now < e.t_soon ? 0 : 1); ORL_DBG(10, due_all > 0 ? 1 : 0); sink.deferred = true; return; }
        break;
      }
      continue;
    }
    if (!sink.active) {
      // The number of due releases is known exactly: the whole list (now < t_soon: nothing outside is due) or the full scan
      // just done.
      const int total = due_all >= 0 ? due_all : tot;
      if constexpr (CP) {
        // compact sink: a mask table of ORL_REL_MAX releases per env-step and no limit per link (the number of releases in
        // a step is geometric with mean ~1: 31 are exceeded once in 10^10 env-steps; the test knob lowers the limit)
        sink.active = total <= P.rel_limit;
      } else {
      // Item mode needs <= ORL_IMASKS releases meeting on one link.
      sink.active = total + extra <= P.item_masks;
      if (!sink.active && total < 200) {
        // More releases than one item holds masks for (the release count per step is geometric: ~0.2 % of env-steps
        // exceed 8).  What matters is the count PER LINK: tally the touches of every due release first — every lane
        // walks its own slots, no side effects — and take item mode when no link exceeds the item form.
        // (release times requested eight at a time: one dependent load per slot made these rare wavefronts the
        // stragglers that set the kernel's duration)
        for (int base = gl; base < e.ev_hwm; base += 64) {
          double tt[8];
#pragma unroll
          for (int k = 0; k < 8; k++) {
            const int i = base + 8 * k;
            const double v = e.ev_time[i < e.ev_hwm ? i : e.ev_hwm - 1];
            tt[k] = (i < e.ev_hwm) ? v : INF;
          }
#pragma unroll
          for (int k = 0; k < 8; k++) {
            if (tt[k] <= e.now) {
              const u64 info = e.ev_info[base + 8 * k];
              const PathRec rec = path_rec_load(P, (int)(info & 0xffffffu));
              const int hops = path_rec_byte(rec, 0);
              for (int h = 0; h < hops; h++) {
                const int link = path_rec_byte(rec, 2 + h);
                atomicAdd(sink.tally + (link >> 2), 1u << (8 * (link & 3)));
              }
            }
          }
        }
        wave_fence();
        u32 mx = 0;
        for (int wd = gl; wd < sink.tw; wd += 8) {
          const u32 v = sink.tally[wd];
          const u32 a0 = v & 0xff, a1 = (v >> 8) & 0xff, a2 = (v >> 16) & 0xff, a3 = v >> 24;
          u32 m01 = a0 > a1 ? a0 : a1, m23 = a2 > a3 ? a2 : a3;
          m01 = m01 > m23 ? m01 : m23;
          mx = mx > m01 ? mx : m01;
        }
        sink.active = g8_max((int)mx) <= P.item_masks;
      }
      }
      if (!sink.active) {
        ORL_DBG(11, 1);
        sink.deferred = true;  // nothing has been touched: k_rel_tail takes this env
        return;
      }
    }
    if constexpr (CP) {
      // Compact sink: nothing in it depends on the order in which masks ARRIVE (a release is a bit in the link words and an
      // entry of the env's mask table, both addressed by its rank), so only the RANKS are found one after the other — a
      // light loop: the 8-lane minimum and the holder's bookkeeping — and then every holder lane works on its own release
      // at the same time: info word and path record (one memory round trip for the wavefront instead of one per release of
      // its busiest env), the mask, one LDS atomic per hop, its slot of the free-slot stack.  Before, the whole body ran
      // once per release with 8 lanes cooperating on the hops, and a wavefront took as many rounds as the env with the most
      // releases (3-4 on average for 8 envs at one release per env-step): the largest item of the phase profile.
      u32 rk = 0u;  // rank (1-based within this round) of this lane's list entry k, 6 bits each
      int n_round = 0;
      if (e.rank_pairs && __ballot(ndl > 1) == 0ull) {
        // (the forms with registers to spare — soon list in registers, 3 waves per SIMD: in the 128-VGPR forms this path
        // spills, cfg3 -9 %)  No lane of the wavefront holds more than one due entry (the usual case): every lane ranks its entry against the
        // seven other lanes of its group directly — the partners i^1, i^2, i^3 (quad permutations), i^7 (half mirror) and
        // the quad permutations of the mirrored value (i^6, i^5, i^4) — instead of one 8-lane minimum per release: ~100
        // instructions whatever the number of releases, where the loop below costs ~70 per release of the busiest env.
        // Each due entry's release time is cleared by the lane that scans its slot, which sees it either as its own or as
        // exactly one partner's.
        int r = 1;
#define ORL_RANK_STEP(OT, OI) { const double ot_ = (OT); const int oi_ = (OI); \
                                r += (ot_ < ct || (ot_ == ct && oi_ < ci)) ? 1 : 0; \
                                if (ot_ < INF && gl == (oi_ & 7)) e.ev_time[oi_] = INF; }
        ORL_RANK_STEP(dpp_d<ORL_DPP_XOR1>(ct), dpp_i<ORL_DPP_XOR1>(ci))
        ORL_RANK_STEP(dpp_d<ORL_DPP_XOR2>(ct), dpp_i<ORL_DPP_XOR2>(ci))
        ORL_RANK_STEP(dpp_d<ORL_DPP_XOR3>(ct), dpp_i<ORL_DPP_XOR3>(ci))
        {
          const double mt = dpp_d<ORL_DPP_HALF_MIRROR>(ct);  // lane i^7; its quad permutations are i^6, i^5, i^4
          const int mi = dpp_i<ORL_DPP_HALF_MIRROR>(ci);
          ORL_RANK_STEP(mt, mi)
          ORL_RANK_STEP(dpp_d<ORL_DPP_XOR1>(mt), dpp_i<ORL_DPP_XOR1>(mi))
          ORL_RANK_STEP(dpp_d<ORL_DPP_XOR2>(mt), dpp_i<ORL_DPP_XOR2>(mi))
          ORL_RANK_STEP(dpp_d<ORL_DPP_XOR3>(mt), dpp_i<ORL_DPP_XOR3>(mi))
        }
#undef ORL_RANK_STEP
        n_round = tot;
        if (ndl == 1) {
          if (gl == (ci & 7)) e.ev_time[ci] = INF;
          rk = (u32)r << (6 * ck);
          dirty |= 1 << ck;
#pragma unroll
          for (int k = 0; k < NS; k++)
            if (k == ck) st[k] = INF;
        }
      } else {
        for (;;) {
          double bt = ct;
          int bi = ci, bl = gl;
#define ORL_MIN_STEP(CTRL) { double ot = dpp_d<CTRL>(bt); int oi = dpp_i<CTRL>(bi); int ol = dpp_i<CTRL>(bl); \
                             if (ot < bt || (ot == bt && oi < bi)) { bt = ot; bi = oi; bl = ol; } }
          ORL_MIN_STEP(ORL_DPP_XOR1) ORL_MIN_STEP(ORL_DPP_XOR2) ORL_MIN_STEP(ORL_DPP_HALF_MIRROR)
#undef ORL_MIN_STEP
          if (!(bt <= e.now)) break;
          n_round++;
          if (gl == (bi & 7)) e.ev_time[bi] = INF;  // written by the lane that scans this slot
          if (gl == bl) {  // the holder notes the rank, drops the entry from its list and moves to its next due entry, if any (rare)
            rk |= (u32)n_round << (6 * ck);
            dirty |= 1 << ck;
            ct = INF; ci = 0x7fffffff;
            int nk = 0;
#pragma unroll
            for (int k = 0; k < NS; k++) {
              if (k == ck) st[k] = INF;
              const bool due = st[k] <= e.now;
              if (due && (st[k] < ct || (st[k] == ct && si[k] < ci))) { ct = st[k]; ci = si[k]; nk = k; }
            }
            ck = nk;
          }
        }
      }
      ORL_PROF(6);
      int d_br = 0, d_nh = 0;
      const int nfree0 = e.nfree, nrel0 = sink.nrel;
      unsigned short* fs = (unsigned short*)(e.scal + SC_FREE0);
      while (rk) {  // (per lane: one entry, rarely two)
        const int k = (int)__builtin_ctz(rk) / 6;
        const int r = (int)((rk >> (6 * k)) & 63u);
        rk &= ~(63u << (6 * k));
        int idx = si[0];
#pragma unroll
        for (int j = 1; j < NS; j++) idx = (k == j) ? si[j] : idx;
        const u64 info = (idx == pushed_idx) ? pushed_info : ((idx == pre_idx) ? pre_info : e.ev_info[idx]);
        const PathRec rec = path_rec_load(P, (int)(info & 0xffffffu));
        const int s0 = (int)((info >> 24) & 0xfffu), n = (int)((info >> 36) & 0xffu), br = (int)((info >> 49) & 0x7fffu);
        const int hops = path_rec_byte(rec, 0), kk = nrel0 + r;
        if constexpr (RD) {
          // rows deferred: the holder lane frees the slots itself and logs the release as the step's kk-th event behind the provision
          u64 lm = 0ull;
          for (int h = 0; h < hops; h++) {
            const int link = path_rec_byte(rec, 2 + h);
            row_apply_mask(sink.rows + (size_t)link * sink.roww, s0, n, false);
            lm |= 1ull << link;
          }
          sink.ev[2 * (sink.ev_rel0 + kk - 1)] = make_ulonglong2(ORL_EV_META(s0, n, sink.ev_t, false), lm);
          sink.ev[2 * (sink.ev_rel0 + kk - 1) + 1] = make_ulonglong2((u64)__double_as_longlong(e.now), 0ull);  // (a release happens at the step's new clock)
        } else {
        sink.mtab[kk] = (unsigned short)((u32)s0 | ((u32)n << 9));
        for (int h = 0; h < hops; h++) {
          const int link = path_rec_byte(rec, 2 + h);
          atomicOr(&sink.tab[link].bits, 1u << kk);  // (items: sink_compact)
          if (sink.rows) row_apply_mask(sink.rows + (size_t)link * sink.roww, s0, n, false);
        }
        }
        // the freed slot goes onto the env's free-slot stack at the place its rank gives it (g8::free_push, one by one before)
        const int fp = nfree0 + r - 1;
        if (fp < ORL_FREE_SLOTS) fs[fp] = (unsigned short)idx;
        d_br += br;
        d_nh += n * hops;
      }
      e.ev_cnt -= n_round;
      e.nfree = (nfree0 + n_round < ORL_FREE_SLOTS) ? nfree0 + n_round : ORL_FREE_SLOTS;
      sink.nrel = nrel0 + n_round;
      e.s_br -= (i64)g8_sum(d_br);
      e.s_nh -= (i64)g8_sum(d_nh);
    } else {
    // info word + path record of this lane's candidate, requested by all lanes together
      u64 inf0 = 0;
      PathRec rc0 = PathRec();
      if (ci != 0x7fffffff) { inf0 = (ci == pushed_idx) ? pushed_info : ((ci == pre_idx) ? pre_info : e.ev_info[ci]); rc0 = path_rec_load(P, (int)(inf0 & 0xffffffu)); }
      ORL_PROF(6);
      for (;;) {
        double bt = ct;
        int bi = ci, bl = gl;
  #define ORL_MIN_STEP(CTRL) { double ot = dpp_d<CTRL>(bt); int oi = dpp_i<CTRL>(bi); int ol = dpp_i<CTRL>(bl); \
                               if (ot < bt || (ot == bt && oi < bi)) { bt = ot; bi = oi; bl = ol; } }
        ORL_MIN_STEP(ORL_DPP_XOR1) ORL_MIN_STEP(ORL_DPP_XOR2) ORL_MIN_STEP(ORL_DPP_HALF_MIRROR)
  #undef ORL_MIN_STEP
        if (!(bt <= e.now)) break;
        const u64 info = gget(inf0, bl, lane);
        PathRec rec;
        rec.q[0] = gget(rc0.q[0], bl, lane); rec.q[1] = gget(rc0.q[1], bl, lane);
        rec.q[2] = gget(rc0.q[2], bl, lane); rec.q[3] = gget(rc0.q[3], bl, lane);
        if (gl == (bi & 7)) e.ev_time[bi] = INF;  // written by the lane that scans this slot
        if (gl == bl) {  // the holder drops the entry from its list and moves to its next due entry, if any (rare)
          dirty |= 1 << ck;
          ct = INF; ci = 0x7fffffff;
          int nk = 0;
  #pragma unroll
          for (int k = 0; k < NS; k++) {
            if (k == ck) st[k] = INF;
            const bool due = st[k] <= e.now;
            if (due && (st[k] < ct || (st[k] == ct && si[k] < ci))) { ct = st[k]; ci = si[k]; nk = k; }
          }
          ck = nk;
          if (ci != 0x7fffffff) { inf0 = (ci == pushed_idx) ? pushed_info : ((ci == pre_idx) ? pre_info : e.ev_info[ci]); rc0 = path_rec_load(P, (int)(inf0 & 0xffffffu)); }
        }
        const int s0 = (int)((info >> 24) & 0xfffu), n = (int)((info >> 36) & 0xffu);
        const int core = (int)((info >> 44) & 0x1fu), br = (int)((info >> 49) & 0x7fffu);
        e.ev_cnt--;
        g8::free_push(e, gl, bi);
        sink_add(sink, rec, core, s0, n, lane);
        e.s_br -= br;
        e.s_nh -= (i64)n * path_rec_byte(rec, 0);
      }
    }
    ORL_PROF(7);
    if (e.now < e.t_soon) break;
  }
  {
    double m = st[0];
#pragma unroll
    for (int k = 1; k < NS; k++) m = st[k] < m ? st[k] : m;
    const double lm = g8::g8_min(m);
    e.next_rel = lm < e.t_soon ? lm : e.t_soon;
  }
  out.dirty = dirty;
#pragma unroll
  for (int k = 0; k < NS; k++) { out.t[k] = st[k]; out.i[k] = si[k]; }
  ORL_PROF(8);
}

// rare path: envs whose due releases did not fit the item form (flag bit 16 of SC_ACC) release them in place
template <int ENV, int W>
__device__ __forceinline__ void rel_serial(const DevParams& P, i64 env, int lane) {
  u64* s = P.scal + env * ORL_SCAL_WORDS;
  if (!((s[SC_ACC] >> 16) & 1ull)) return;
  const int gl = lane & 7;
  if (gl == 0) atomicAdd(P.q_stat, 1u);  // statistics: env-steps that took the serial path
  EnvG e;
  g8::env_load(P, e, env);
  if (P.pipeline2) e.rs = e.cs + 2 * P.C;  // two-kernel pipeline: the next step needs what releases added
  g8::release_due<ENV, W>(P, e, lane);
  e.t_soon = -__builtin_inf();  // released in place: the soon list is stale
  if (gl == 0) {
    // only what releases change goes back (the whole record, as env_store writes it, kept every word of it live through
    // release_due: the spills of the 128-VGPR forms sat here)
    s[SC_ACC] = s[SC_ACC] & ~(1ull << 16);
    s[SC_NEXTREL] = (u64)__double_as_longlong(e.next_rel);
    s[SC_TSOON] = (u64)__double_as_longlong(e.t_soon);
    s[SC_SBR] = (u64)e.s_br;
    s[SC_SNH] = (u64)e.s_nh;
    s[SC_EV] = pack2(e.ev_hwm, e.ev_cnt);
    s[SC_HINT] = pack2(e.nfree, 0);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// row kernel, one LANE per work item: the whole link row (W words) sits in the lane's registers, so the row
// statistics are plain per-lane bit arithmetic without cross-lane steps, a wavefront retires 64 items, and a
// 65 536-env launch is a single round of ~2 700 wavefronts.  Same arithmetic, expression for expression, as
// path_apply of the monolithic kernels (orl_device.h / orl_device_g8.h): _update_link_stats (rmsa_env.py:464-543) on
// the first touch of a link in a step, the time_diff == 0 form on later touches, and the integer sums behind
// _get_network_compactness kept per core with each row's cached contribution.
// ---------------------------------------------------------------------------------------------------------------
// CACHED: `cache` is this row's word of Wmem::ic0; `touched`: the 64-slot words that differ from the row the cache
// describes; `update`: the row summarised is the one that stays (write the refreshed cache word back)
template <int W, bool CACHED = false>
__device__ __forceinline__ void row_stat_lane(const u64 (&a)[W], int S, RowStat& st, int& max_empty, int& edge, u32* cache = nullptr,
                                              u32 touched = 0u, bool update = false) {
  int nu = 0, lo = 1 << 20, hi = 0, free_ = 0, nf = 0, best = 0, c = 0;
  u64 inter[CACHED ? W : 1];
#pragma unroll
  for (int w = 0; w < W; w++) {
    const u64 maskw = word_mask_lo(S - 64 * w);
    const u64 used = ~a[w] & maskw;
    const u64 prev_top = (w == 0) ? 0ull : (a[w - 1] >> 63);
    const u64 carry_u = (w == 0) ? 0ull : (prev_top ^ 1ull);  // slot 64w-1 < S for every w < W
    nu += __popcll(used & ~((used << 1) | carry_u));
    if (used) {
      const int l = 64 * w + (int)__builtin_ctzll(used);
      lo = l < lo ? l : lo;
      hi = 64 * w + 64 - (int)__builtin_clzll(used);  // w ascending: the last word with a used slot wins
    }
    free_ += __popcll(a[w]);
    // longest run of free slots (runs continue across word boundaries)
    if (CACHED) {
      // the runs that touch a word boundary here, branch-free (a lane whose word is entirely free and one whose word is not
      // took both sides of an if / else before): lead / trail read 64 for a free word, which then only lengthens the run
      // that crosses it; inter[w]: the word without its leading and trailing runs
      const u64 na = ~a[w];
      const bool full = na == 0ull;
      const int lead = full ? 64 : (int)__builtin_ctzll(na), trail = full ? 64 : (int)__builtin_clzll(na);
      best = (c + lead) > best ? (c + lead) : best;
      c = full ? c + 64 : trail;
      // (a & (a + 1) drops the run of free slots at the word's low end — the carry runs through it —, the shifted all-ones
      // mask the run at its high end: a 64-bit add, a shift and two ANDs where two word_range() masks took a dozen instructions)
      inter[CACHED ? w : 0] = full ? 0ull : ((a[w] & (a[w] + 1ull)) & (~0ull >> trail));
    } else if (a[w] == ~0ull) {
      c += 64;
      best = c > best ? c : best;
    } else {
      const int lead = (int)__builtin_ctzll(~a[w]);
      const int inner = word_longest_run_flat(a[w]);  // (a[w] has a used slot here: runs of at most 63)
      const int cand = (c + lead) > inner ? (c + lead) : inner;
      best = cand > best ? cand : best;
      c = (int)__builtin_clzll(~a[w]);
    }
  }
  if (CACHED) {
    best = c > best ? c : best;
    // ... then the runs inside a word: a dozen dependent 64-bit shift-and-test steps per word (word_longest_run), 10 % of the
    // persistent kernel's time when done for every word of every row (S = 320).  Only the words this step changed are
    // searched; the result for the others comes from the row's cache word.
    u32 cw = *cache;
    u32 need = touched;
#pragma unroll
    for (int w = 0; w < W; w++) {
      const int f = (int)((cw >> (6 * w)) & 63u);
      if (!((touched >> w) & 1u)) {
        if (f == 63) need |= 1u << w;
        else best = f > best ? f : best;
      }
    }
    while (need) {  // (per lane: one word, two when a mask straddles a word boundary)
      const int w = (int)__builtin_ctz(need);
      need &= need - 1u;
      u64 x = inter[0];
#pragma unroll
      for (int k = 1; k < (CACHED ? W : 1); k++) x = (w == k) ? inter[k] : x;
      // (searched whatever `best` is: an unknown entry would have to be searched in a later step, when the word has not
      // changed, and the wavefront runs as many rounds as its lane with the most words to search)
      const int f = word_longest_run_flat(x);  // <= 62: x has neither its lowest nor its highest bit set
      best = f > best ? f : best;
      cw = (cw & ~(63u << (6 * w))) | ((u32)f << (6 * w));
    }
    if (update) *cache = cw;
  }
  // free blocks strictly inside [lambda_min, lambda_max): that range starts and ends with a used block and blocks alternate,
  // so there is exactly one free block between consecutive used blocks (what rmsa_env.py:733-741 counts by run-length
  // encoding the slice)
  const bool two = nu > 1;
  const int fb = two ? nu - 1 : 0;
  max_empty = best;
  const int tw = (S - 1) >> 6, tb = (S - 1) & 63;
  int top_bit = 0;
#pragma unroll
  for (int w = 0; w < W; w++) top_bit = (w == tw) ? (int)((a[w] >> tb) & 1ull) : top_bit;
  edge = (int)(a[0] & 1ull) + top_bit;
  // free blocks: used and free blocks alternate along the row, so there is one more, as many, or one fewer of them than used
  // blocks according to how many ends of the row are free (a row without a used slot: 0 - 1 + 2 = 1) — no second pass of
  // block-start counts over the words
  nf = nu - 1 + edge;
  st.nu = nu; st.lo = lo; st.hi = hi; st.occ = two ? hi - lo : 0; st.fb = fb; st.free_ = free_; st.nf = nf;
}

// ---- incremental row summary (round 6: the replay kernel k_rowstats) ----------------------------------------------------------
// A lane that owns a link row for a whole launch need not summarise its W words again at every touch: a provision takes n slots out
// of ONE free run, a release merges the freed slots with the free runs on either side, and everything _update_link_stats reads of
// the row (rmsa_env.py:464-543: free slots, used blocks, first / last used slot, longest free run) follows from the two free-run
// lengths next to the mask — found with a count-leading / count-trailing on a 64-slot window on either side — except when a
// provision splits the row's longest free run (then that one figure is searched again).  ~120 instructions per touch instead of
// ~330 for the five-word summary with its inner-run cache; same integers, checked on every env against the in-loop row phase.
struct RowInc { int free_, nu, lo, hi, me; };  // free slots; used blocks; first used slot (1 << 20: none); last used slot + 1 (0: none); longest free run
template <int W> __device__ __forceinline__ u64 row_word(const u64 (&a)[W], int w) {  // word w of the row, 0 outside it
  u64 x = 0ull;
#pragma unroll
  for (int k = 0; k < W; k++) x = (w == k) ? a[k] : x;
  return x;
}
// free slots directly below slot p (p - 1, p - 2, ...) / from slot p upwards; slots outside the row count as taken (the bits above
// the last slot of a row are zero)
template <int W> __device__ __forceinline__ int row_run_below(const u64 (&a)[W], int p) {
  int len = 0;
  for (int q = p; q > 0; q -= 64) {
    const int wq = q >> 6, b = q & 63;
    const u64 hi = row_word<W>(a, wq), lo = row_word<W>(a, wq - 1);
    const u64 x = (b == 0) ? lo : ((hi << (64 - b)) | (lo >> b));  // slots [q - 64, q): bit 63 = slot q - 1
    const int c = (x == ~0ull) ? 64 : (int)__builtin_clzll(~x);
    len += c;
    if (c < 64) break;
  }
  return len;
}
template <int W> __device__ __forceinline__ int row_run_from(const u64 (&a)[W], int p) {
  int len = 0;
  for (int q = p; q < 64 * W; q += 64) {
    const int wq = q >> 6, b = q & 63;
    const u64 lo = row_word<W>(a, wq), hi = row_word<W>(a, wq + 1);
    const u64 x = (b == 0) ? lo : ((lo >> b) | (hi << (64 - b)));  // slots [q, q + 64): bit 0 = slot q
    const int c = (x == ~0ull) ? 64 : (int)__builtin_ctzll(~x);
    len += c;
    if (c < 64) break;
  }
  return len;
}
// the longest run of free slots of the row (runs continue across word boundaries; the non-cached branch of row_stat_lane)
template <int W> __device__ __forceinline__ int row_longest_free(const u64 (&a)[W]) {
  int best = 0, c = 0;
#pragma unroll
  for (int w = 0; w < W; w++) {
    if (a[w] == ~0ull) {
      c += 64;
    } else {
      const int lead = (int)__builtin_ctzll(~a[w]), trail = (int)__builtin_clzll(~a[w]);
      const int inner = word_longest_run_flat((a[w] & (a[w] + 1ull)) & (~0ull >> trail));  // the word without its boundary runs
      const int cand = (c + lead) > inner ? (c + lead) : inner;
      best = cand > best ? cand : best;
      c = trail;
    }
  }
  return c > best ? c : best;
}
// the mask [s0, s0 + n) applied to the row (a provision clears free slots, a release sets taken ones) and the summary brought up to date
template <int W>
__device__ __forceinline__ void row_inc_apply(u64 (&a)[W], int S, int s0, int n, bool prov, RowInc& s) {
  const int l = row_run_below<W>(a, s0), r = row_run_from<W>(a, s0 + n);  // (the slots next to the mask do not change)
  const int joins = ((s0 > 0 && l == 0) ? 1 : 0) + ((s0 + n < S && r == 0) ? 1 : 0);  // used slots directly next to the mask
  const Mask2 mm = mask2(s0, n);
#pragma unroll
  for (int w = 0; w < W; w++) a[w] ^= mask2_word(mm, w);
  if (prov) {
    s.free_ -= n;
    s.nu += 1 - joins;
    s.lo = s0 < s.lo ? s0 : s.lo;
    s.hi = s0 + n > s.hi ? s0 + n : s.hi;
    if (l + n + r == s.me) s.me = row_longest_free<W>(a);  // (the run it split was the longest, or as long)
  } else {
    s.free_ += n;
    s.nu -= 1 - joins;
    const int m = l + n + r;
    s.me = m > s.me ? m : s.me;
    if (s.nu == 0) { s.lo = 1 << 20; s.hi = 0; }
    else {
      if (s0 == s.lo) s.lo = s0 + n + r;       // the first used block began with these slots: the next used slot lies behind the merged run
      if (s0 + n == s.hi) s.hi = s0 - l;       // the last used block ended with them
    }
  }
}

// the cache word of a row, from scratch
template <int W>
__device__ __forceinline__ u32 row_inner_cache(const u64* row) {
  u32 cw = 0u;
#pragma unroll
  for (int w = 0; w < (W <= 5 ? W : 0); w++) {
    const u64 a = row[w];
    int f = 0;
    if (a != ~0ull) {
      const int lead = (int)__builtin_ctzll(~a), trail = (int)__builtin_clzll(~a);
      f = word_longest_run(a & ~word_range(0, lead) & ~word_range(64 - trail, 64));
    }
    cw |= (u32)f << (6 * w);
  }
  return cw;
}
// the 64-slot words the slots [s0, s0 + n) lie in
__device__ __forceinline__ u32 mask_words(int s0, int n) { return (1u << (s0 >> 6)) | (1u << ((s0 + n - 1) >> 6)); }
// The slots [s0, s0 + n), 1 <= n <= 63, as bits of the (at most two) 64-slot words they lie in: word s0 >> 6 gets `lo`, the
// next one `hi`.  word_range() per word of the row — two clamps, two 64-bit shifts and half a dozen selects, times W words,
// times two or three masks per item — was a tenth of the persistent kernel's instruction stream.
// the part of the row summary the compactness sums need: used blocks, lambda_min, lambda_max
template <int W>
__device__ __forceinline__ void row_occ_fb(const u64 (&a)[W], int S, int& occ, int& fb) {
  int nu = 0, lo = 1 << 20, hi = 0;
#pragma unroll
  for (int w = 0; w < W; w++) {
    const u64 used = ~a[w] & word_mask_lo(S - 64 * w);
    const u64 carry_u = (w == 0) ? 0ull : ((a[w > 0 ? w - 1 : 0] >> 63) ^ 1ull);
    nu += __popcll(used & ~((used << 1) | carry_u));
    if (used) {
      const int l = 64 * w + (int)__builtin_ctzll(used);
      lo = l < lo ? l : lo;
      hi = 64 * w + 64 - (int)__builtin_clzll(used);
    }
  }
  const bool two = nu > 1;
  occ = two ? hi - lo : 0;
  fb = two ? nu - 1 : 0;
}

// Single-core families (RMSA, DeepRMSA, RWA): every mask of an item works on the same row, and only the first touch of
// the link at a clock value needs the row's statistics — the provision at the provision clock and the first release at the
// step clock (_update_link_stats, rmsa_env.py:464-543); further releases of the step see time_diff == 0, i.e.
// new = ((old * now) + (cur * 0.0)) / now with a finite cur, whatever the row looks like.  So an item costs at most two
// full row summaries and two float64 updates, however many releases meet on the link.  Few links carry both in one step
// (two per wavefront-step), and a lane that looped twice would hold the other 63 back: such an item is worked on by TWO
// lanes at once (role): lane A summarises the row after the provision and updates the statistics at the provision clock;
// lane B applies the provision and the releases, summarises the row after the first release, and — once A's statistics are
// stored — does the update at the step clock, the further releases and the row store.  Everything else (`whole`) is one lane.
// The compactness sums change by (summary after the provision - before) and (final - after the provision), the latter
// also into rel_sums.
// EARLY (the two-wavefront form of the persistent kernel, persist_row_wave): the control wavefront has already applied every
// mask to the row; this function only derives the statistics, from the row states it reconstructs, and stores no slot map.
// `sig` (LDS, when given) is set to `sig_val` once the row, the mask table and the clocks have been read: the control wavefront
// may overwrite them from there on.
template <int ENV, int W, bool EARLY = false>
__device__ __forceinline__ void row_item_lane1(const DevParams& P, const Wmem& M, const i64 env, const int link, const u32 bits,
                                               const unsigned short* mtab, int second, Prof& prof, bool early_ls = false,
                                               double* stash_env = nullptr, u32* sig = nullptr, u32 sig_val = 0u) {
  // `stash_env` (k_agent; LDS, [E][2] of this env): where a lane whose evaluated mask is a release leaves the link's
  // utilization and compactness as they were BEFORE its update, i.e. after the step's provision — what info's link averages use
  // `bits`: the (env, link) word of the compact sink — bit 0 the step's provision, bit k its k-th release; `mtab`: the env's
  // masks, entry k = (first slot: 9 | slots: 6)
  const int E = P.E, S = P.S;
  const int nmask = __popc(bits);
  const bool prov_first = (bits & 1u) != 0;
  const bool shared = prov_first && nmask >= 2;   // two lanes work on this item
  const bool role_b = shared && second;           // ... this one on the releases
  const bool role_a = shared && !second;          // ... this one on the provision
  u64* row = wm_bm(P, M, env) + (size_t)link * W;
  int* cs = wm_cs(P, M, env);
  int* rs = cs + 2;
  double* ls = wm_ls(P, M, env) + 4 * link;
  const double now = M.clk ? M.clk[2 * (env - M.clk_env0) + 1] : __longlong_as_double((i64)wm_scal(P, M, env)[SC_NOW]);
  const double now_prov = M.clk ? M.clk[2 * (env - M.clk_env0)] : __longlong_as_double((i64)wm_scal(P, M, env)[SC_NOWA]);
  // the link's 32-byte statistics record is requested before the row summary (a global round trip hidden behind it); a B
  // lane reads it in round 1 below, after the A lane's store
  double2 ls01 = make_double2(0.0, 0.0), ls23 = make_double2(0.0, 0.0);
  if (early_ls && !role_b) { ls01 = *(const double2*)ls; ls23 = *(const double2*)(ls + 2); }
  // the row's contribution to the compactness sums as the previous step left it, where the launch keeps those words in GLOBAL
  // memory (Wmem::ocg: no room in the LDS window): requested with the link's record; a B lane needs none (see the sums below)
  u32* ocg = (ENV != ENV_RWA && !M.oc0 && M.ocg) ? M.ocg + (env - M.cenv0) * E + link : nullptr;
  u32 ocg_v = 0u;
  if (ocg && !role_b) ocg_v = *ocg;
  u64 a[W];
#pragma unroll
  for (int w = 0; w < W; w++) a[w] = row[w];
  ORL_PROFR(3);
  // EARLY: the row read above is the FINAL one.  A provision takes slots that were free and a release frees slots that were
  // taken, so taking the masks of the word back in reverse order gives the row as the step found it; from there on the function
  // works as it does on a row it changes itself.  Everything it needs of the tables is read here.
  u64 fin[EARLY ? W : 1];
  u32 later_e = 0u, mw0_e = 0u, mwf_e = 0u;
  if constexpr (EARLY) {
    u32 rest_e = bits & (bits - 1u);
    int first_e = (int)__builtin_ctz(bits);
    if (role_b) { first_e = (int)__builtin_ctz(rest_e); rest_e &= rest_e - 1u; }
    if (role_a) rest_e = 0u;
#pragma unroll
    for (int w = 0; w < W; w++) fin[w] = a[w];
    // (in reverse: the releases first, the provision last — a service released in the step it was provisioned in has the same
    // mask twice, and its slots were free before the step)
    for (u32 r = bits & ~1u; r; r &= r - 1u) {
      const int k = (int)__builtin_ctz(r);
      const u32 mw = mtab[k];
      const int s0 = (int)(mw & 0x1ff), n = (int)(mw >> 9);
      if (k == first_e) mwf_e = mw;
      if ((rest_e >> k) & 1u) later_e |= mask_words(s0, n);
      const Mask2 mm = mask2(s0, n);
#pragma unroll
      for (int w = 0; w < W; w++) a[w] &= ~mask2_word(mm, w);
    }
    if (bits & 1u) {
      mw0_e = mtab[0];
      if (first_e == 0) mwf_e = mw0_e;
      const Mask2 mm = mask2((int)(mw0_e & 0x1ff), (int)(mw0_e >> 9));
#pragma unroll
      for (int w = 0; w < W; w++) a[w] |= mask2_word(mm, w);
    }
    rw_release_lds();  // (this lane's reads of the row, the mask table and the clocks are done: workgroup-scope release on LDS only)
    if (sig) __hip_atomic_store(sig, sig_val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    ORL_PROFR(9);
  }
  // the masks of this lane, in table (= bit) order: `first` is the one whose statistics it evaluates — B: the first release
  // (after applying the provision, entry 0), else the lowest entry — and `rest` the further releases of the step (none for A)
  u32 rest = bits & (bits - 1u);
  int first = (int)__builtin_ctz(bits);
  u32 touched = 0u;  // words of the row this lane changes before it summarises it (row_stat_lane's cache)
  if (role_b) {
    const u32 mw = EARLY ? mw0_e : (u32)mtab[0];
    const int s0 = (int)(mw & 0x1ff), n = (int)(mw >> 9);
    touched = mask_words(s0, n);
    const Mask2 mm = mask2(s0, n);
#pragma unroll
    for (int w = 0; w < W; w++) a[w] &= ~mask2_word(mm, w);
    first = (int)__builtin_ctz(rest);
    rest &= rest - 1u;
  }
  if (role_a) rest = 0u;
  // what the row contributes to the compactness sums before this lane's masks (B: after the provision): from the row's
  // cache word where the launch keeps one — B's is what the A lane of the same item leaves there, read after A's round —
  // else from the row itself (a hundred instructions for five words)
  u32* ocw = (ENV != ENV_RWA && M.oc0) ? M.oc0 + (env - M.cenv0) * E + link : nullptr;
  int occ0 = 0, fb0 = 0;
  if (ENV != ENV_RWA) {
    if (ocg) { }  // (unpacked where it is used, behind the summary: touching it here would wait for the load in front of it)
    else if (!ocw) row_occ_fb<W>(a, S, occ0, fb0);
    else if (!role_b) { const u32 c = *ocw; occ0 = (int)(c >> 16); fb0 = (int)(c & 0xffffu); }
  }
  const bool rel_f = first != 0;  // the evaluated mask is a release
  {
    const u32 mw = EARLY ? mwf_e : (u32)mtab[first];
    const int s0 = (int)(mw & 0x1ff), n = (int)(mw >> 9);
    touched |= mask_words(s0, n);
    const Mask2 mm = mask2(s0, n);
#pragma unroll
    for (int w = 0; w < W; w++) {
      const u64 m = mask2_word(mm, w);
      a[w] = rel_f ? (a[w] | m) : (a[w] & ~m);
    }
  }
  ORL_PROFR(4);
  RowStat after;
  int max_empty = 0, edge = 0;
  u32* icw = (ENV != ENV_RWA && M.ic0) ? M.ic0 + (env - M.cenv0) * E + link : nullptr;
  if (ENV != ENV_RWA) {
#ifdef ORL_DIAG_INSTEAD_OF_ROW_SUMMARY
    ORL_DIAG_INSTEAD_OF_ROW_SUMMARY
#else
    if (W >= 3 && W <= 5 && icw) row_stat_lane<W, (W >= 3 && W <= 5)>(a, S, after, max_empty, edge, icw, touched, !role_a);
    else row_stat_lane<W>(a, S, after, max_empty, edge);
#endif
  } else {
    int f = 0;
#pragma unroll
    for (int w = 0; w < W; w++) f += __popcll(a[w]);
    after.free_ = f;
  }
  ORL_PROFR(5);
  // the values _update_link_stats derives from the row (rmsa_env.py:464-543)
  const int free_ = after.free_;
  const double cur_util = div_pos((double)(S - free_), (double)S);
  double cur_frag = 0.0, cur_comp = 0.0;
  if (ENV != ENV_RWA && free_ > 0) {
    int me = (after.nf > 1 && !(after.nf == 2 && edge == 2)) ? max_empty : 0;
    cur_frag = 1.0 - div_pos((double)me, (double)free_);
    if (after.nu > 1) cur_comp = div_pos((double)(after.hi - after.lo), (double)(S - free_)) * div_pos(1.0, (double)after.nu);
    else cur_comp = 1.0;
  }
  if (ocw && role_a) *ocw = ((u32)after.occ << 16) | (u32)after.fb;  // (B reads it after the fence that ends round 0)
  const double clock = rel_f ? now : now_prov;
  const int n_rest = __popc(rest);
  // the running averages: round 0 every lane but the B lanes, round 1 the B lanes (their link's record has been updated and
  // stored by the A lane of the same wavefront in round 0)
#ifdef ORL_DIAG_NO_SECOND_ROUND
  const u64 any_b = 0ull;
#else
  const u64 any_b = __ballot(role_b);
#endif
  for (int round = 0; round < 2; round++) {
    if (round == 1 && !any_b) break;
    if ((round == 1) == role_b) {
      if (role_b || !early_ls) { ls01 = *(const double2*)ls; ls23 = *(const double2*)(ls + 2); }
      double last_update = ls23.y;
      double util = ls01.x, frag = ls01.y, comp = ls23.x;
      if (stash_env && rel_f) { stash_env[2 * link] = util; stash_env[2 * link + 1] = comp; }
#ifdef ORL_DIAG_NO_F64
      if (false) {
#else
      if (clock > 0) {  // the first touch of the link at this clock value
#endif
        const double time_diff = clock - last_update;
        const Recip rc = recip_of(clock);
        util = div_by((util * last_update) + (cur_util * time_diff), rc);
        if (ENV != ENV_RWA) {
          frag = div_by((frag * last_update) + (cur_frag * time_diff), rc);
          comp = div_by((comp * last_update) + (cur_comp * time_diff), rc);
        }
      }
      last_update = clock;
      // further releases of the step on this link: the time_diff == 0 form of the update (never for an A lane)
      if (now > 0 && n_rest > 0) {
        const Recip rn = recip_of(now);
        for (int k = 0; k < n_rest; k++) {
          util = div_by((util * now) + 0.0, rn);
          if (ENV != ENV_RWA) { frag = div_by((frag * now) + 0.0, rn); comp = div_by((comp * now) + 0.0, rn); }
        }
      }
      *(double2*)ls = make_double2(util, frag);
      *(double2*)(ls + 2) = make_double2(comp, last_update);
    }
    wave_fence();
  }
  ORL_PROFR(6);
  if (ocw && role_b) { const u32 c = *ocw; occ0 = (int)(c >> 16); fb0 = (int)(c & 0xffffu); }
  int occL = after.occ, fbL = after.fb;
  if (rest) {  // the masks of the further releases, then what the row contributes in the end
    u32 later = 0u;
    if constexpr (EARLY) {
      later = later_e;
#pragma unroll
      for (int w = 0; w < W; w++) a[w] = fin[w];
    } else {
      for (u32 r = rest; r; r &= r - 1u) {
        const u32 mw = mtab[__builtin_ctz(r)];
        const int s0 = (int)(mw & 0x1ff), n = (int)(mw >> 9);
        later |= mask_words(s0, n);
        const Mask2 mm = mask2(s0, n);
#pragma unroll
        for (int w = 0; w < W; w++) a[w] |= mask2_word(mm, w);
      }
    }
    if (ENV != ENV_RWA) row_occ_fb<W>(a, S, occL, fbL);
    if (icw) {  // the cache word describes the row before these masks: their words become unknown
      u32 cw = *icw;
#pragma unroll
      for (int w = 0; w < (W <= 5 ? W : 0); w++) cw |= ((later >> w) & 1u) ? (63u << (6 * w)) : 0u;
      *icw = cw;
    }
  }
  if (ENV != ENV_RWA) {
    if (ocg) { occ0 = (int)(ocg_v >> 16); fb0 = (int)(ocg_v & 0xffffu); }
    if (ocg && shared) {
      // the two lanes of a link that carries provision and release, without handing the after-provision value from A to B: the
      // sums are integer atomics, so A takes the row's old contribution out of the totals and its after-provision one out of
      // the release part, and B puts the final one into both — totals += final - before, release part += final - after provision
      if (role_a) { atomicAdd(cs, -occ0); atomicAdd(cs + 1, -fb0); atomicAdd(rs, -after.occ); atomicAdd(rs + 1, -after.fb); }
      else { atomicAdd(cs, occL); atomicAdd(cs + 1, fbL); atomicAdd(rs, occL); atomicAdd(rs + 1, fbL); }
    } else {
      // this lane's part of the row's contribution to the compactness sums; releases also go into rel_sums
      const int d_occ = occL - occ0, d_fb = fbL - fb0;
      if (d_occ) atomicAdd(cs, d_occ);
      if (d_fb) atomicAdd(cs + 1, d_fb);
      if (rel_f) {
        if (d_occ) atomicAdd(rs, d_occ);
        if (d_fb) atomicAdd(rs + 1, d_fb);
      }
    }
  }
  if (!role_a) {
    if constexpr (!EARLY) {
#pragma unroll
      for (int w = 0; w < W; w++) row[w] = a[w];
    }
    if (ocw) *ocw = ((u32)occL << 16) | (u32)fbL;
    if (ocg) *ocg = ((u32)occL << 16) | (u32)fbL;
  }
  ORL_PROFR(7);
}

// MIXED (two-kernel pipeline): mask 0 of an item may be this step's provision — slots cleared, statistics at the
// provision clock (SC_NOWA) — followed by the step's releases at the new clock (SC_NOW); what the releases add to
// the per-core sums is also accumulated in rel_sums (the next step needs the sums as they were in between).
template <int ENV, int W>
__device__ __forceinline__ void row_item_lane(const DevParams& P, const Wmem& M, const Item it, Prof& prof) {
  constexpr bool MIXED = true;
  const int E = P.E, S = P.S;
  const i64 env = (i64)(u32)it.a.x;
  const int link = (int)((it.a.x >> 32) & 0xff), nmask = (int)((it.a.x >> 40) & 15);
  const bool release = ((it.a.x >> 44) & 1) != 0;
  const bool prov_first = MIXED && ((it.a.x >> 45) & 1) != 0;
  const u64 cores = it.b.y;
  u64* bm = wm_bm(P, M, env);
  int* cs = wm_cs(P, M, env);
  int* rs = MIXED ? cs + 2 * P.C : nullptr;
  double* ls = wm_ls(P, M, env);
  const double now = M.clk ? M.clk[2 * (env - M.clk_env0) + 1] : __longlong_as_double((i64)wm_scal(P, M, env)[SC_NOW]);
  const double now_prov = M.clk ? M.clk[2 * (env - M.clk_env0)] : __longlong_as_double((i64)wm_scal(P, M, env)[SC_NOWA]);
  const double2 ls01 = *(const double2*)(ls + 4 * link), ls23 = *(const double2*)(ls + 4 * link + 2);  // one 32-byte record
  double last_update = ls23.y;
  double util = ls01.x, frag = ls01.y, comp = ls23.x;
  ORL_PROFR(3);
  u64 a[W];
  int pk = 0, prev_core = -1;
  for (int k = 0; k < nmask; k++) {
    const int core = (int)((cores >> (5 * k)) & 0x1f);
    const u64 mw = k < 4 ? (it.a.y >> (16 * k)) : (it.b.x >> (16 * (k - 4)));
    const int s0 = (int)(mw & 0x1ff), n = (int)((mw >> 9) & 0x7f);
    const bool rel_k = MIXED ? !(k == 0 && prov_first) : release;
    const double clock = (MIXED && !rel_k) ? now_prov : now;
    // the first operation at a clock value is the reference's generic update; further touches of the link at the same
    // clock have time_diff == 0
    const bool first_at_clock = (k == 0) || (MIXED && k == 1 && prov_first);
    u64* row = bm + (size_t)(core * E + link) * W;
    if (core != prev_core) {  // several releases on the same core row keep working on the registers
#pragma unroll
      for (int w = 0; w < W; w++) a[w] = row[w];
      if (ENV != ENV_RWA) {
        // the row's contribution to the compactness sums BEFORE the change, recomputed from the row itself: the
        // kernel is bound by scattered memory requests, not ALU — the cached copy the per-env kernels keep
        // (core_sums[2C + ...]) would cost a read and a write of one more line per item
        int occ_b, fb_b;
        row_occ_fb<W>(a, S, occ_b, fb_b);
        pk = (occ_b << 16) | fb_b;
      }
      prev_core = core;
    }
    {
      const Mask2 mm = mask2(s0, n);
#pragma unroll
      for (int w = 0; w < W; w++) {
        const u64 m = mask2_word(mm, w);
        a[w] = rel_k ? (a[w] | m) : (a[w] & ~m);
        if (m) row[w] = a[w];
      }
    }
    RowStat after;
    int max_empty = 0, edge = 0;
    ORL_PROFR(4);
    if (ENV != ENV_RWA) {
      row_stat_lane<W>(a, S, after, max_empty, edge);
      ORL_PROFR(5);
      // this row's contribution to the compactness sums of its core
      const int d_occ = after.occ - (pk >> 16), d_fb = after.fb - (pk & 0xffff);
      pk = (after.occ << 16) | after.fb;
      if (d_occ) atomicAdd(cs + 2 * core, d_occ);
      if (d_fb) atomicAdd(cs + 2 * core + 1, d_fb);
      if (MIXED && rel_k) {
        if (d_occ) atomicAdd(rs + 2 * core, d_occ);
        if (d_fb) atomicAdd(rs + 2 * core + 1, d_fb);
      }
    } else {
      int f = 0;
#pragma unroll
      for (int w = 0; w < W; w++) f += __popcll(a[w]);
      after.free_ = f;
    }
    if (clock > 0) {
      if (first_at_clock) {  // _update_link_stats (rmsa_env.py:464-543)
        const double time_diff = clock - last_update;
        const int free_ = after.free_;
        const Recip rc = recip_of(clock);  // (one reciprocal for the three quotients: see div_by)
        double cur_util = div_pos((double)(S - free_), (double)S);
        util = div_by((util * last_update) + (cur_util * time_diff), rc);
        if (ENV != ENV_RWA) {
          double cur_frag = 0.0, cur_comp = 0.0;
          if (free_ > 0) {
            int me = (after.nf > 1 && !(after.nf == 2 && edge == 2)) ? max_empty : 0;
            cur_frag = 1.0 - div_pos((double)me, (double)free_);
            if (after.nu > 1) cur_comp = div_pos((double)(after.hi - after.lo), (double)(S - free_)) * div_pos(1.0, (double)after.nu);
            else cur_comp = 1.0;
          }
          frag = div_by((frag * last_update) + (cur_frag * time_diff), rc);
          comp = div_by((comp * last_update) + (cur_comp * time_diff), rc);
        }
      } else {
        // the same link touched again at the same clock: the reference's update has last_update == now and
        // time_diff == 0, i.e. new = ((old * now) + (cur * 0.0)) / now with a finite cur
        const Recip rc = recip_of(clock);
        util = div_by((util * clock) + 0.0, rc);
        if (ENV != ENV_RWA) { frag = div_by((frag * clock) + 0.0, rc); comp = div_by((comp * clock) + 0.0, rc); }
      }
    }
    last_update = clock;
  }
  ORL_PROFR(6);
  *(double2*)(ls + 4 * link) = make_double2(util, frag);
  *(double2*)(ls + 4 * link + 2) = make_double2(comp, last_update);
  ORL_PROFR(7);
}

}  // namespace sp
}  // namespace orl
