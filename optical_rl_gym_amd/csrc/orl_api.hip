// orl_api.hip — the C ABI (include/orl.h) of liborlgpu.so: topology / batch construction, dispatch to the per-W kernel units
// (orl_kernels.hip through orl_host.h), the small W-independent kernels, state read-back.  gfx950 only.
// Host side: plain HIP runtime, one stream per batch, no torch types.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <dlfcn.h>
#include <limits.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <exception>
#include <memory>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "orl_host.h"

using namespace orl;

// =============================================================================================
// W-independent kernels
// =============================================================================================
extern __shared__ __attribute__((aligned(16))) unsigned char orl_lds_api[];

// CPython MT19937 state -> in-place "update-behind" form (orl_device.h, Rng).  keep_record: reseeding a live env
// (optical_network_env.py:205-210 replaces only self.rng) — the scalar record keeps everything but the stream position.
__global__ void __launch_bounds__(64) k_init_mt(DevParams P, const u32* raw, const unsigned char* mask, int keep_record) {
  const i64 env = blockIdx.x;
  if (mask && !mask[env]) return;
  const int lane = lane_id();
  u32* m = (u32*)orl_lds_api;
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
  const int pos = p0 >= 624 ? 0 : p0;
  if (keep_record) {
    u64* rec = P.scal + env * ORL_SCAL_WORDS;
    const u64 fl = rec[SC_FLAGS], idp = rec[SC_ID_MTPOS];
    const bool first = P.mt2 && !((fl >> 32) & ORL_FLAG_MT2) && P.env_type != ENV_RWA;
    if (first) {  // the stream the env was constructed with lives on as its bit-rate stream (see DevParams::mt2)
      u32* d2 = P.mt2 + env * 624;
      for (int i = lane; i < 624; i += 64) d2[i] = dst[i];
    }
    wave_fence();
    for (int i = lane; i < 624; i += 64) dst[i] = m[i];
    if (lane == 0) {
      rec[SC_ID_MTPOS] = (idp & 0xffffffffull) | ((u64)(u32)pos << 32);
      if (first) {
        rec[SC_FLAGS] = fl | ((u64)ORL_FLAG_MT2 << 32);
        rec[SC_HINT] = (rec[SC_HINT] & 0xffffffffull) | ((idp >> 32) << 32);
      }
    }
    return;
  }
  for (int i = lane; i < 624; i += 64) dst[i] = m[i];
  u64 v = 0;
  if (lane == SC_ID_MTPOS) v = pack2(0, pos);
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
// available_slots.flatten()] as uint8, one workgroup per env, bits unpacked one per thread-iteration.
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

// end of a device-resident run: the network-compactness update the last step left pending (one thread per env), so
// that every host-visible state is final; also the OR of every env's flag word (as k_flags_or) — one launch, and
// orl_batch_run fetches its result together with the straggler count in a single copy
__global__ void k_finish2(DevParams P, int finish, unsigned int* flag_or) {
  const i64 env = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  u32 f = 0;
  if (env < P.B) {
    u64* s = P.scal + env * ORL_SCAL_WORDS;
    if (finish) {
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
    const u64 v = s[SC_FLAGS];
    f = (u32)(v >> 32);
    if (f & ORL_FLAG_BAD_ACTION) s[SC_FLAGS] = v & ~((u64)ORL_FLAG_BAD_ACTION << 32);
  }
  for (int o = 32; o > 0; o >>= 1) f |= (u32)__shfl_xor((int)f, o, 64);
  if ((threadIdx.x & 63) == 0 && f) atomicOr(flag_or, f);
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

// 16-byte-per-lane copy (the read + write companion of k_calib_read for orl_debug_stream_peak)
__global__ void k_calib_copy(const ulonglong2* __restrict__ src, ulonglong2* __restrict__ dst, i64 n16) {
  for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < n16; j += (i64)gridDim.x * blockDim.x) dst[j] = src[j];
}

// observation array -> float32 (orl_batch_get_obs_f32)
__global__ void k_cast_f32(const double* __restrict__ src, float* __restrict__ dst, i64 n) {
  for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (i64)gridDim.x * blockDim.x) dst[i] = (float)src[i];
}

// rows idx[0..n) of a [B][w] array -> out[n][w] (orl_batch_get_info_rows)
__global__ void k_gather_rows(const double* __restrict__ src, const long long* __restrict__ idx, i64 n, int w, double* __restrict__ out) {
  const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n * w) out[i] = src[idx[i / w] * w + (i % w)];
}

// sums of services_processed / services_accepted over the batch (two atomics per wave)
// orl_batch_set_info_mode(1): the info columns the 8-lanes-per-env step kernel stops writing (network compactness, its difference,
// the two link means: entries 4..7 of RMSA / DeepRMSA) read NaN from here on instead of the last values written in mode 0
__global__ void k_info_nan(DevParams P) {
  const i64 i = (i64)blockIdx.x * 256 + threadIdx.x;
  if (i < P.B * 4) P.info[(i >> 2) * P.n_info + 4 + (i & 3)] = __longlong_as_double(0x7ff8000000000000ll);
}
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
// (a fixed buffer: reporting an error must not itself allocate — std::bad_alloc is one of the errors reported)
static thread_local char g_err[512];
static int fail(int code, const char* fmt, ...) noexcept {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof g_err, fmt, ap);
  va_end(ap);
  return code;
}
// Exception barrier: nothing throws across the C ABI.  Every entry point with a body that can allocate (new orl_batch,
// std::vector growth, std::string) is a function-try-block ending in one of these.
#define ORL_ABI_CATCH_INT                                                                                          \
  catch (const std::bad_alloc&) { return fail(ORL_E_INTERNAL, "out of host memory"); }                             \
  catch (const std::exception& e_) { return fail(ORL_E_INTERNAL, "C++ exception at the ABI: %s", e_.what()); }     \
  catch (...) { return fail(ORL_E_INTERNAL, "unknown C++ exception at the ABI"); }
#define ORL_ABI_CATCH_VOID catch (...) { fail(ORL_E_INTERNAL, "C++ exception in a destroy call"); }
#define HIPCHK(x)                                                                                     \
  do {                                                                                                \
    hipError_t _e = (x);                                                                              \
    if (_e != hipSuccess) return fail(ORL_E_HIP, "%s failed: %s", #x, hipGetErrorString(_e));         \
  } while (0)

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
extern "C" const char* orl_last_error(void) { return g_err; }
extern "C" int orl_device_count(void) try {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
ORL_ABI_CATCH_INT
extern "C" int orl_build_has_alt(void) try {
#ifdef ORL_ALT_IMPLS
  return 1;
#else
  return 0;
#endif
}
ORL_ABI_CATCH_INT

extern "C" int orl_topology_create(const orl_topology_desc* d, int device_id, orl_topology** out) try {
  if (!d || !out) return fail(ORL_E_INVALID, "null argument");
  if (d->n_nodes < 2 || d->n_nodes > 512 || d->n_links < 1 || d->n_links > 128 || d->k_paths < 1 || d->k_paths > 64 ||
      d->max_hops < 1 || d->max_hops > 30 || d->n_modulations < 1 || d->n_modulations > 255)
    return fail(ORL_E_INVALID, "topology out of supported range (N<=512, E<=128, k<=64, hops<=30)");
  if (!d->n_paths || !d->path_hops || !d->path_links || !d->path_length || !d->path_modulation || !d->edge_iter_order)
    return fail(ORL_E_INVALID, "null topology table");
  const int N = d->n_nodes, E = d->n_links, K = d->k_paths, H = d->max_hops, M = d->n_modulations;
  const size_t nn = (size_t)N * N, npk = nn * K;
  for (size_t i = 0; i < nn; i++)
    if (d->n_paths[i] < 0 || d->n_paths[i] > K) return fail(ORL_E_INVALID, "n_paths[%zu] out of range", i);
  for (size_t i = 0; i < npk; i++) {
    const int hops = d->path_hops[i];
    if (hops < 0 || hops > H || d->path_modulation[i] >= M) return fail(ORL_E_INVALID, "bad path table entry %zu", i);
    for (int h = 0; h < hops; h++) {
      const int l = d->path_links[i * H + h];
      if (l < 0 || l >= E) return fail(ORL_E_INVALID, "bad link index %d at hop %d of path %zu", l, h, i);
    }
  }
  {
    std::vector<char> seen((size_t)E, 0);
    for (int i = 0; i < E; i++) {
      const int l = d->edge_iter_order[i];
      if (l < 0 || l >= E || seen[(size_t)l]) return fail(ORL_E_INVALID, "edge_iter_order is not a permutation of the links");
      seen[(size_t)l] = 1;
    }
  }
  HIPCHK(hipSetDevice(device_id));
  struct Destroy { void operator()(orl_topology* p) const { orl_topology_destroy(p); } };
  std::unique_ptr<orl_topology, Destroy> hold(new orl_topology());
  orl_topology* t = hold.get();
  t->n_paths = nullptr;
  t->path_length = nullptr; t->edge_iter_order = nullptr; t->link_pos = nullptr;
  t->device = device_id;
  t->N = N; t->E = E; t->K = K; t->H = H; t->M = M;
  int rc = 0;
  std::vector<int32_t> mod(npk);
  for (size_t i = 0; i < npk; i++) mod[i] = d->path_modulation[i] < 0 ? 0 : d->path_modulation[i];
  t->h_hops.assign(d->path_hops, d->path_hops + npk);
  t->h_links.assign(d->path_links, d->path_links + npk * H);
  t->h_mod = mod;
  rc |= upload_conv(&t->n_paths, d->n_paths, nn, nullptr);
  rc |= upload_conv(&t->path_length, d->path_length, npk, nullptr);
  rc |= upload_conv(&t->edge_iter_order, d->edge_iter_order, (size_t)E, nullptr);
  {
    std::vector<int32_t> pos((size_t)E, 0);
    for (int i = 0; i < E; i++) pos[(size_t)d->edge_iter_order[i]] = i;
    rc |= upload_conv(&t->link_pos, pos.data(), (size_t)E, nullptr);
  }
  if (rc) return ORL_E_HIP;
  *out = hold.release();
  return ORL_OK;
}
ORL_ABI_CATCH_INT

extern "C" void orl_topology_destroy(orl_topology* t) try {
  if (!t) return;
  hipSetDevice(t->device);
  if (t->n_paths) hipFree(t->n_paths);
  if (t->path_length) hipFree(t->path_length);
  if (t->edge_iter_order) hipFree(t->edge_iter_order);
  if (t->link_pos) hipFree(t->link_pos);
  delete t;
}
ORL_ABI_CATCH_VOID

// ---- launch dispatch over the row width ---------------------------------------------------------------
// Anything but the persistent kernel is about to write slot maps: the row caches the persistent kernel left with the state
// (DevParams::row_cache) no longer describe them.  The key of the next persistent launch differs from every stored stamp.
static void slot_maps_change(orl_batch* b) {
  if (++b->cache_epoch >= (1 << 22)) {  // (the key keeps 22 bits of it: start over with no stamp left standing)
    if (b->P.row_cache_stamp) hipMemsetAsync(b->P.row_cache_stamp, 0, (size_t)((b->P.B + 7) / 8) * sizeof(int), b->stream);
    b->cache_epoch = 1;
  }
}
static void launch_reset(orl_batch* b, int full, const unsigned char* dmask) {
  slot_maps_change(b);
#define CALL(WW) orl_launch::reset<WW>(b, full, dmask)
  ORL_DISPATCH_W(b, CALL)
#undef CALL
}
static void launch_policy(orl_batch* b, int pol) {
#define CALL(WW) orl_launch::policy<WW>(b, pol)
  ORL_DISPATCH_W(b, CALL)
#undef CALL
}
static void launch_step64(orl_batch* b, int auto_reset, int want_info, int fused_policy) {
  slot_maps_change(b);
#define CALL(WW) orl_launch::step64<WW>(b, auto_reset, want_info, fused_policy)
  ORL_DISPATCH_W(b, CALL)
#undef CALL
}
static void launch_agent_step(orl_batch* b, int auto_reset, int pol = -1) {
  slot_maps_change(b);
#define CALL(WW) orl_launch::agent_step<WW>(b, auto_reset, pol)
  ORL_DISPATCH_W(b, CALL)
#undef CALL
}
static void launch_obs(orl_batch* b, int with_terminal) {
#define CALL(WW) orl_launch::obs<WW>(b, with_terminal)
  ORL_DISPATCH_W(b, CALL)
#undef CALL
}
static void launch_persist(orl_batch* b, const DevParams& VP, hipStream_t st, int pol, int target, int* wg_step, unsigned int* unfinished,
                           unsigned int* clear_next, int finish) {
#define CALL(WW) orl_launch::persist<WW>(b, VP, st, pol, target, wg_step, unfinished, clear_next, finish)
  ORL_DISPATCH_W(b, CALL)
#undef CALL
}
static int persist_resident(orl_batch* b) {
  int r = 0;
#define CALL(WW) r = orl_launch::persist_resident<WW>(b, b->n_cu)
  ORL_DISPATCH_W(b, CALL)
#undef CALL
  return r;
}
// the per-env arrays of envs [lo, lo + cnt) as a batch of their own (lo a multiple of 8: wavefronts own 8 consecutive envs)
static DevParams env_view(const DevParams& P, i64 lo, i64 cnt, int part) {
  DevParams q = P;
  q.B = cnt;
  q.bitmap += lo * P.bm_words; q.ev_time += lo * P.ev_cap; q.ev_info += lo * P.ev_cap; q.mt += lo * 624;
  if (q.mt2) q.mt2 += lo * 624;
  q.lstat += lo * 4 * P.E; q.scal += lo * ORL_SCAL_WORDS; q.svc_desc += lo; q.core_sums += lo * P.cs_words;
  q.soon_t += lo * ORL_SOON; q.soon_i += lo * ORL_SOON;
  q.svc_q += lo * 8; q.svc_ht += lo * 8; q.svc_pk += lo * 8; q.svc_cnt += lo * 8;  // (64 lanes per 8 envs; lo is a multiple of 8)
  q.row_cache += (lo / 8) * 2 * (i64)P.row_cache_words; q.row_cache_stamp += lo / 8;
  if (q.slog) { q.slog += lo; q.log_n += lo / 8; }  // (rows of the log span the whole batch: log_stride stays)
  if (q.elog) { q.elog += lo * (i64)(2 * P.elog_cap); q.elog_n += lo; q.bitmap0 += lo * P.bm_words; }
  if (q.ssum) q.ssum += lo;
  if (q.br_hist) q.br_hist += lo * 2 * P.n_br;
  if (q.act_hist) q.act_hist += lo * ((P.K + 1) + (P.S + 1));
  if (q.act2d) q.act2d += lo * P.act2d_words;
  if (q.ep_log) { q.ep_log += lo * P.ep_cap; q.ep_count += lo; }
  if (q.ep_rew) { q.ep_rew += lo * P.ep_cap; q.ep_rew_acc += lo; }
  q.path_col += lo;
  q.actions += lo * 4; q.reward += lo; q.done += lo; q.info += lo * P.n_info;
  if (q.obs) { q.obs += lo * P.obs_dim; q.term_obs += lo * P.obs_dim; }
  q.q_a += (lo / 8) * P.q_wave * 2; q.q_cnt_a += lo / 8;
  q.q_def = P.q_def + (size_t)part * P.q_def_stride;  // each part has its own list of deferred envs (indices relative to lo)
  return q;
}
static void launch_step2(orl_batch* b, int pol) {
  slot_maps_change(b);
#define CALL(WW) orl_launch::step2<WW>(b, pol)
  ORL_DISPATCH_W(b, CALL)
#undef CALL
}
// OR of the env flag words into d_unfinished[17] (the pair report_flags owns; the persistent launches have their own slots)
static void launch_finish2(orl_batch* b, int finish) {
  hipLaunchKernelGGL(k_finish2, dim3((unsigned)((b->P.B + 255) / 256)), dim3(256), 0, b->stream, b->P, finish, b->d_unfinished + 17);
}
static int flags_to_rc(const orl_batch* b, unsigned int f);

template <typename T> static int dalloc(orl_batch* b, T** p, size_t n) {
  HIPCHK(hipMalloc((void**)p, n * sizeof(T) + 64));
  b->allocs.push_back(*p);
  return 0;
}

static int policy_ok(const orl_batch* b, int policy_id) {
  if (policy_id < 0 || policy_id > ORL_POLICY_PATH_FF) return 0;
  if (policy_id == ORL_POLICY_PATH_FF && b->P.env_type != ENV_RMSA && b->P.env_type != ENV_RWA) return 0;
  if (b->P.env_type == ENV_QOS && policy_id > ORL_POLICY_LLP_FF) return 0;
  return 1;
}

// the scalar sizes of DevParams that follow from the configuration alone (no device needed: also what keys a specialisation)
static void derive_sizes(const orl_env_config* c, int N, int E, int K, int H, int M, int64_t n_envs, DevParams& P, int* wt_out) {
  const bool qos = c->env_type == ORL_ENV_QOS;
  const int S = c->num_spectrum_resources, C = c->num_spatial_resources;
  P.env_type = c->env_type;
  P.N = N; P.E = E; P.K = K; P.H = H; P.M = M;
  P.S = S; P.C = C;
  int wt = S <= 64 ? 1 : (S <= 128 ? 2 : (S <= 320 ? 5 : 8));
  if (qos) wt = 1;  // one counter per link (available_spectrum) instead of a slot row
  *wt_out = wt;
  P.W = wt;
  P.n_classes = qos ? c->n_service_classes : 0;
  P.episode_length = c->episode_length;
  P.allow_rejection = c->allow_rejection ? 1 : 0;
  P.J = c->env_type == ORL_ENV_DEEPRMSA ? c->j : 1;
  P.bit_rate_mode = c->bit_rate_mode;
  P.br_lo = c->bit_rate_lo;
  P.n_br = c->n_bit_rates;
  P.rand_n = c->bit_rate_hi + 1 - c->bit_rate_lo;
  P.rand_bits = 0;
  for (int v = P.rand_n; v > 0; v >>= 1) P.rand_bits++;
  P.lambda_a = c->lambda_arrival;
  P.pf_window = 4.0 / c->lambda_arrival;
  P.lambda_h = c->lambda_holding;
  P.B = n_envs;
  int cap = c->event_capacity;
  if (cap <= 0) {
    double load = P.lambda_a / P.lambda_h;
    cap = (int)(load + 10.0 * sqrt(load) + 64.0);
  }
  P.ev_cap = (cap + 63) / 64 * 64;
  int words = C * P.E * wt;
  P.bm_words = (words + 1) & ~1;
  int rej = P.allow_rejection;
  if (qos) P.n_info = 2;
  else if (c->env_type == ORL_ENV_RWA) P.n_info = 2 + (P.K + rej) + (S + rej);
  else if (c->env_type == ORL_ENV_RMCSA) P.n_info = 4;
  else P.n_info = 8 + (c->bit_rate_mode == 1 ? c->n_bit_rates + 1 : 0);
  P.obs_dim = c->env_type == ORL_ENV_DEEPRMSA ? 1 + 2 * P.N + (2 * P.J + 3) * P.K : 0;
  P.cs_words = (4 * C + 15) & ~15;  // sums and their release part; whole 64-byte lines per env
  P.lds_bytes = ((P.bm_words + 4 * P.E + P.E + P.obs_dim) * 8 + P.cs_words * 4 + 15) & ~15;
  if (P.lds_bytes < 624 * 4) P.lds_bytes = 624 * 4;
}
// does the device-resident loop of this configuration go through the persistent kernel? (k <= 8 paths, release slots indexed
// with 8 + 3 bits, services of at most 63 slots: the row items carry first slot: 9 bits | slots: 6 bits)
static bool pipeline_applies(const orl_env_config* c, const DevParams& P) {
  int max_n = 1;
  if (c->n_slots)
    for (size_t i = 0; i < (size_t)P.n_br * P.M; i++) max_n = c->n_slots[i] > max_n ? c->n_slots[i] : max_n;
  return P.K <= 8 && P.ev_cap <= 2048 && c->env_type != ORL_ENV_QOS && max_n <= 63 && P.S <= 512;
}

static int batch_create_impl(const orl_env_config* c, const orl_topology* t, int64_t n_envs, const uint32_t* mt_state,
                             const int64_t* seeds, orl_batch** out) {
  if (!c || !t || !out || (!mt_state && !seeds) || n_envs < 1) return fail(ORL_E_INVALID, "null/invalid argument");
  if (c->struct_size != sizeof(orl_env_config))
    return fail(ORL_E_INVALID, "orl_env_config.struct_size is %u, this library (ABI %d) expects %zu: client built against another header",
                c->struct_size, ORL_ABI_VERSION, sizeof(orl_env_config));
  if (n_envs > (int64_t)1 << 30) return fail(ORL_E_INVALID, "n_envs must be <= 2^30");
  if (c->env_type < 0 || c->env_type > ORL_ENV_QOS) return fail(ORL_E_INVALID, "unknown env_type %d", c->env_type);
  const bool qos = c->env_type == ORL_ENV_QOS;
  if (qos && (c->n_service_classes < 1 || c->n_service_classes > 64 || !c->cum_class || !c->class_reward))
    return fail(ORL_E_INVALID, "QoSConstrainedRA needs 1..64 service classes with their tables");
  const int S = c->num_spectrum_resources, C = c->num_spatial_resources;
  if (S < 2 || S > 512) return fail(ORL_E_INVALID, "num_spectrum_resources must be in [2, 512]");
  if (C < 1 || C > 31 || (c->env_type != ORL_ENV_RMCSA && C != 1)) return fail(ORL_E_INVALID, "bad num_spatial_resources");
  if (c->env_type == ORL_ENV_DEEPRMSA && (c->j < 1 || c->j > 8)) return fail(ORL_E_INVALID, "j must be in [1, 8]");
  if (c->n_bit_rates < 1 || c->n_bit_rates > 4096) return fail(ORL_E_INVALID, "bad n_bit_rates");
  if (c->episode_length < 1) return fail(ORL_E_INVALID, "episode_length must be positive");
  if (!c->cum_src || !c->cum_dst) return fail(ORL_E_INVALID, "node probability tables missing");
  const bool no_rates = c->env_type == ORL_ENV_RWA || qos;  // one unit per service, no bit rate
  if (!no_rates && (!c->n_slots || !c->bit_rates)) return fail(ORL_E_INVALID, "bit-rate tables missing");
  if (c->env_type == ORL_ENV_RMCSA && (!c->lmax_snr || !c->lmax_xt)) return fail(ORL_E_INVALID, "RMCSA reach tables missing");
  if (c->bit_rate_mode == 1 && !c->cum_bit_rate) return fail(ORL_E_INVALID, "cum_bit_rate missing");
  if (!no_rates) {
    for (int i = 0; i < c->n_bit_rates * t->M; i++)
      if (c->n_slots[i] < 1 || c->n_slots[i] > 64) return fail(ORL_E_INVALID, "n_slots entries must be in [1, 64]");
    for (int i = 0; i < c->n_bit_rates; i++)
      if (c->bit_rates[i] < 0 || c->bit_rates[i] > 32767) return fail(ORL_E_INVALID, "bit rates must be < 32768");
  }
  const int rand_n = c->bit_rate_hi + 1 - c->bit_rate_lo;
  if (c->bit_rate_mode == 0 && !no_rates && (rand_n < 1 || rand_n != c->n_bit_rates))
    return fail(ORL_E_INVALID, "continuous mode needs n_bit_rates == hi - lo + 1");
  if (!(c->lambda_arrival > 0) || !(c->lambda_holding > 0)) return fail(ORL_E_INVALID, "rates must be positive");
  if (c->action_histograms && qos) return fail(ORL_E_INVALID, "action histograms are not kept for this env family");
  HIPCHK(hipSetDevice(t->device));
  // (owned until the last statement: an early return or an exception on the way frees everything allocated so far)
  struct Destroy { void operator()(orl_batch* p) const { orl_batch_destroy(p); } };
  std::unique_ptr<orl_batch, Destroy> hold(new orl_batch());
  orl_batch* b = hold.get();
  memset(&b->P, 0, sizeof b->P);
  b->device = t->device;
#define FAIL_B(...) do { return fail(__VA_ARGS__); } while (0)
#define HIPCHK_B(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) FAIL_B(ORL_E_HIP, "%s failed: %s", #x, hipGetErrorString(e_)); } while (0)
  DevParams& P = b->P;
  derive_sizes(c, t->N, t->E, t->K, t->H, t->M, n_envs, P, &b->wt);
  if (P.lds_bytes > 64 * 1024) FAIL_B(ORL_E_INVALID, "per-env LDS window too large (%d B)", P.lds_bytes);
  {
    // The persistent kernel (k_persist) serves the device-resident loop wherever its 8-lanes-per-env slot scan applies
    // (k <= 8 paths, release slots indexed with 8 + 3 bits): cfg2 64 envs 2.6e6 vs 2.1e6 env-steps/s for the per-env kernel;
    // 4 096: 1.6e8 vs 7.6e7; 32 768: 6.3e8 vs 4.0e8; RWA 4 096: 2.4e8 vs 8.3e7.  ORL_STEP_IMPL=64 forces the per-env kernel
    // (cross-checks); ORL_STEP_IMPL=2 with ORL_PERSIST=0 selects the two-kernel form in ORL_ALT_IMPLS builds.
    const char* impl = getenv("ORL_STEP_IMPL");
    const bool pipeline_ok = pipeline_applies(c, P);
    b->persist = pipeline_ok && !(impl && atoi(impl) == 64);
    if (const char* pv = getenv("ORL_PERSIST")) {
      if (atoi(pv) == 0 && b->persist) {
        b->persist = 0;
#ifdef ORL_ALT_IMPLS
        b->two_kernel = (impl && atoi(impl) == 2) ? 1 : 0;
#endif
      }
    }
  }
  P.pipeline2 = (b->persist || b->two_kernel) ? 1 : 0;
  {
    // Host- / agent-driven steps with auto reset (what SB3's VecEnv issues) through the phases of the persistent kernel
    // (k_agent) wherever they apply and the batch is large enough to fill the GPU with 8 envs per wavefront: cfg2 65 536 envs
    // 305 us per step in k_step (one wavefront per env), ~90 us in k_agent.  ORL_AGENT_STEP=1 forces it for any batch size
    // (parity tests), 0 disables it.
    // (QoSConstrainedRA, which no persistent kernel serves, has its own 8-lanes-per-env step kernel, k_agent_qos: its releases
    // are found by 8 lanes scanning the env's release times where k_step has 64, a longer chain per step that pays once the
    // batch fills the GPU — 65 536 envs 120 against 225 us per step, 32 768: 77 / 109, 16 384: 65 / 66, 4 096: 47 / 33)
    const char* impl64 = getenv("ORL_STEP_IMPL");
    const bool fits = (b->persist && P.E <= 128) || (qos && P.K <= 8 && !(impl64 && atoi(impl64) == 64));
    b->agent_step = fits && n_envs >= (qos ? 20480 : 2048);
    if (const char* av = getenv("ORL_AGENT_STEP")) b->agent_step = fits && atoi(av) != 0;
  }

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
    if (qos) {
      rc |= upload_conv(&p, c->cum_class, (size_t)P.n_classes, &b->allocs); P.cum_class = p;
      rc |= upload_conv(&p, c->class_reward, (size_t)P.n_classes, &b->allocs); P.class_reward = p;
    }
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
    const size_t waves = ((B + 31) / 32 + 16) * 4;
    P.q_wave = 8 * (P.H > P.E ? P.H : P.E);
    P.item_masks = ORL_IMASKS;
    // test knob: a smaller limit sends far more env-steps through the serial tail (and, RMCSA, the tally pass)
    P.rel_limit = 31;
    if (const char* mv = getenv("ORL_ITEM_MASKS")) { int v = atoi(mv); if (v >= 1 && v <= ORL_IMASKS) { P.item_masks = v; P.rel_limit = v; } }
    P.q_cap = (i64)waves * P.q_wave;
    rc |= dalloc(b, &P.q_a, (size_t)P.q_cap * 2);  // 32-byte items
    rc |= dalloc(b, &P.q_cnt_a, waves);
    rc |= dalloc(b, &P.q_stat, 16);
    // deferred-env lists: count at [0], env indices from [16]; two buffers (the steps of the two-kernel form alternate)
    P.q_def_stride = (i64)(B + 16);
    rc |= dalloc(b, &P.q_def, 2 * (size_t)P.q_def_stride);
    rc |= dalloc(b, &b->d_wg_step, (B + 7) / 8 + 16);
    rc |= dalloc(b, &b->d_unfinished, 32);
    if (!rc) HIPCHK_B(hipMemset(b->d_unfinished, 0, 32 * sizeof(unsigned int)));
    P.row_cache_words = ((8 * P.E * 4 + 15) & ~15) / 4;
    rc |= dalloc(b, &P.row_cache, ((B + 7) / 8) * 2 * (size_t)P.row_cache_words);
    rc |= dalloc(b, &P.row_cache_stamp, (B + 7) / 8 + 16);
    if (!rc) HIPCHK_B(hipMemset(P.row_cache_stamp, 0, ((B + 7) / 8 + 16) * sizeof(int)));
    rc |= dalloc(b, &P.soon_t, B * ORL_SOON);
    rc |= dalloc(b, &P.soon_i, B * ORL_SOON);
    {
      const size_t n_lanes = ((B + 7) / 8) * 64;
      rc |= dalloc(b, &P.svc_q, n_lanes);
      rc |= dalloc(b, &P.svc_ht, n_lanes);
      rc |= dalloc(b, &P.svc_pk, n_lanes);
      rc |= dalloc(b, &P.svc_cnt, n_lanes);
      if (!rc) HIPCHK_B(hipMemset(P.svc_cnt, 0, n_lanes * sizeof(int)));  // nothing drawn ahead
    }
    // (the statistics log and the event log of the persistent kernel's launches are allocated by the first device-resident run,
    // for the launch length it uses: ensure_logs)
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
  if (c->action_histograms) {
    // RMSA / DeepRMSA / RWA: [2][k+1][S+1]; RMCSA: [2][k+1][M+1][C+1][S+1] (rmcsa_env.py:145-180; 863 KB per env at 7 x 320)
    const long long cells = (long long)(P.K + 1) * (S + 1) * (c->env_type == ORL_ENV_RMCSA ? (long long)(P.M + 1) * (C + 1) : 1);
    if (2 * cells > INT_MAX) FAIL_B(ORL_E_INVALID, "action histograms too large");
    P.act2d_words = (int)(2 * cells);
    rc |= dalloc(b, &P.act2d, B * (size_t)P.act2d_words);
  }
  rc |= dalloc(b, &P.path_col, B);
  rc |= dalloc(b, &P.actions, B * 4);
  rc |= dalloc(b, &P.reward, B);
  rc |= dalloc(b, &P.done, B);
  rc |= dalloc(b, &P.info, B * P.n_info);
  if (P.obs_dim) { rc |= dalloc(b, &P.obs, B * P.obs_dim); rc |= dalloc(b, &P.term_obs, B * P.obs_dim); }
  rc |= dalloc(b, &b->d_totals, 2);
  if (rc) return ORL_E_HIP;
  HIPCHK_B(hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking));
  HIPCHK_B(hipStreamCreateWithFlags(&b->stream2, hipStreamNonBlocking));
  HIPCHK_B(hipEventCreateWithFlags(&b->ev_half, hipEventDisableTiming));
  {
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, t->device) == hipSuccess && prop.multiProcessorCount > 0) b->n_cu = prop.multiProcessorCount;
  }
  HIPCHK_B(hipHostMalloc((void**)&b->h_tail, 32 * sizeof(unsigned int), hipHostMallocPortable));
  HIPCHK_B(hipEventCreate(&b->ev0));
  HIPCHK_B(hipEventCreate(&b->ev1));
  HIPCHK_B(hipDeviceSynchronize());  // (the null-stream memsets above are not ordered with the batch's non-blocking stream)
  // everything below is ordered on the batch's own stream
  HIPCHK_B(hipMemsetAsync(P.q_def, 0, 2 * (size_t)P.q_def_stride * sizeof(u32), b->stream));
  HIPCHK_B(hipMemsetAsync(P.q_stat, 0, 16 * sizeof(u32), b->stream));
  HIPCHK_B(hipMemsetAsync(P.path_col, 0, B * sizeof(int), b->stream));
  if (P.act2d) HIPCHK_B(hipMemsetAsync(P.act2d, 0, B * (size_t)P.act2d_words * sizeof(int), b->stream));  // (RMSAEnv.reset never clears them)
  HIPCHK_B(hipMemsetAsync(P.actions, 0, B * 4 * sizeof(int), b->stream));
  // MT state upload + conversion, then the constructor's full reset
  u32* raw = nullptr;
  HIPCHK_B(hipMalloc((void**)&raw, B * 625 * sizeof(u32)));
  b->allocs.push_back(raw);
  long long* dseeds = nullptr;
  if (mt_state) {
    HIPCHK_B(hipMemcpyAsync(raw, mt_state, B * 625 * sizeof(u32), hipMemcpyHostToDevice, b->stream));
  } else {
    HIPCHK_B(hipMalloc((void**)&dseeds, B * sizeof(long long)));
    b->allocs.push_back(dseeds);
    HIPCHK_B(hipMemcpyAsync(dseeds, seeds, B * sizeof(long long), hipMemcpyHostToDevice, b->stream));
    hipLaunchKernelGGL(k_seed_mt, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, b->stream, dseeds, (i64)B, raw);
  }
  hipLaunchKernelGGL(k_init_mt, dim3((unsigned)B), dim3(64), 624 * 4, b->stream, P, raw, (const unsigned char*)nullptr, 0);
  launch_reset(b, 1, nullptr);
  if (P.obs_dim) launch_obs(b, 0);
  HIPCHK_B(hipStreamSynchronize(b->stream));
  HIPCHK_B(hipGetLastError());
  hipFree(raw);
  b->allocs.erase(std::find(b->allocs.begin(), b->allocs.end(), (void*)raw));
  if (dseeds) { hipFree(dseeds); b->allocs.erase(std::find(b->allocs.begin(), b->allocs.end(), (void*)dseeds)); }
#undef FAIL_B
#undef HIPCHK_B
  *out = hold.release();
  return ORL_OK;
}

extern "C" int orl_batch_create(const orl_env_config* c, const orl_topology* t, int64_t n_envs, const uint32_t* mt_state,
                                orl_batch** out) try {
  if (!mt_state) return fail(ORL_E_INVALID, "mt_state is null");
  return batch_create_impl(c, t, n_envs, mt_state, nullptr, out);
}
ORL_ABI_CATCH_INT
extern "C" int orl_batch_create_seeded(const orl_env_config* c, const orl_topology* t, int64_t n_envs, const int64_t* seeds,
                                       orl_batch** out) try {
  if (!seeds) return fail(ORL_E_INVALID, "seeds is null");
  return batch_create_impl(c, t, n_envs, nullptr, seeds, out);
}
ORL_ABI_CATCH_INT

extern "C" void orl_batch_destroy(orl_batch* b) try {
  if (!b) return;
  hipSetDevice(b->device);
  if (b->stream) { hipStreamSynchronize(b->stream); hipStreamDestroy(b->stream); }
  if (b->stream2) { hipStreamSynchronize(b->stream2); hipStreamDestroy(b->stream2); }
  if (b->ev_half) hipEventDestroy(b->ev_half);
  if (b->ev0) hipEventDestroy(b->ev0);
  if (b->ev1) hipEventDestroy(b->ev1);
  if (b->h_tail) hipHostFree(b->h_tail);
  if (b->h_actions) hipHostFree(b->h_actions);
  if (b->spec_handle) dlclose(b->spec_handle);
  for (void* p : b->allocs) hipFree(p);
  delete b;
}
ORL_ABI_CATCH_VOID

extern "C" int orl_batch_info_dim(const orl_batch* b) { return b ? b->P.n_info : 0; }
extern "C" int orl_batch_obs_dim(const orl_batch* b) { return b ? b->P.obs_dim : 0; }

extern "C" int orl_batch_sync(orl_batch* b) try {
  if (!b) return fail(ORL_E_INVALID, "null batch");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipStreamSynchronize(b->stream));
  HIPCHK(hipGetLastError());
  return ORL_OK;
}
ORL_ABI_CATCH_INT

// device copy of a host env mask for the duration of one call
struct DevMask {
  unsigned char* d = nullptr;
  ~DevMask() { if (d) hipFree(d); }
  int upload(const orl_batch* b, const uint8_t* host) {
    if (!host) return 0;
    HIPCHK(hipMalloc((void**)&d, (size_t)b->P.B));
    HIPCHK(hipMemcpyAsync(d, host, (size_t)b->P.B, hipMemcpyHostToDevice, b->stream));
    return 0;
  }
};

extern "C" int orl_batch_reset(orl_batch* b, int full, const uint8_t* env_mask) try {
  if (!b) return fail(ORL_E_INVALID, "null batch");
  HIPCHK(hipSetDevice(b->device));
  DevMask m;
  if (m.upload(b, env_mask)) return ORL_E_HIP;
  launch_reset(b, full ? 1 : 0, m.d);
  if (b->P.obs_dim) launch_obs(b, 0);
  HIPCHK(hipStreamSynchronize(b->stream));
  HIPCHK(hipGetLastError());
  // (a full reset of every env leaves nothing of an abandoned run behind: k_reset dropped the parked services of the envs it
  // reset, the step counters start over with the next run)
  if (full && !env_mask) { b->run_abandoned = false; b->wg_dirty = true; }
  return ORL_OK;
}
ORL_ABI_CATCH_INT

extern "C" int orl_batch_reseed(orl_batch* b, const int64_t* seeds, const uint8_t* env_mask) try {
  if (!b || !seeds) return fail(ORL_E_INVALID, "null argument");
  HIPCHK(hipSetDevice(b->device));
  const size_t B = (size_t)b->P.B;
  DevMask m;
  if (m.upload(b, env_mask)) return ORL_E_HIP;
  if (!b->P.mt2 && b->P.env_type != ENV_RWA) {
    // from now on a reseeded env draws its bit rates from the stream it was constructed with (rmsa_env.py:85-87); the
    // persistent kernel does not carry that second stream: device-resident runs of this batch use the per-env kernel
    u32* p = nullptr;
    HIPCHK(hipMalloc((void**)&p, B * 624 * sizeof(u32) + 64));
    b->allocs.push_back(p);
    b->P.mt2 = p;
    b->persist = 0;
    b->agent_step = 0;
    b->two_kernel = 0;
    b->P.pipeline2 = 0;
  }
  u32* raw = nullptr;
  long long* dseeds = nullptr;
  HIPCHK(hipMalloc((void**)&raw, B * 625 * sizeof(u32)));
  if (hipMalloc((void**)&dseeds, B * sizeof(long long)) != hipSuccess) { hipFree(raw); return fail(ORL_E_HIP, "hipMalloc failed"); }
  hipError_t e = hipMemcpyAsync(dseeds, seeds, B * sizeof(long long), hipMemcpyHostToDevice, b->stream);
  // (services parked by an abandoned run were drawn from the stream that is being replaced; between completed runs none are)
  if (e == hipSuccess && b->P.svc_cnt) e = hipMemsetAsync(b->P.svc_cnt, 0, ((B + 7) / 8) * 64 * sizeof(int), b->stream);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_seed_mt, dim3((unsigned)((B + 63) / 64)), dim3(64), 0, b->stream, dseeds, (i64)B, raw);
    hipLaunchKernelGGL(k_init_mt, dim3((unsigned)B), dim3(64), 624 * 4, b->stream, b->P, raw, (const unsigned char*)m.d, 1);
    e = hipStreamSynchronize(b->stream);
  }
  hipFree(raw);
  hipFree(dseeds);
  if (e != hipSuccess) return fail(ORL_E_HIP, "reseed failed: %s", hipGetErrorString(e));
  HIPCHK(hipGetLastError());
  return ORL_OK;
}
ORL_ABI_CATCH_INT

extern "C" int orl_batch_episode_log(orl_batch* b, int32_t capacity) try {
  if (!b || capacity < 0) return fail(ORL_E_INVALID, "bad argument");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipStreamSynchronize(b->stream));
  const size_t B = (size_t)b->P.B;
  if (capacity == 0) { b->P.ep_log = nullptr; b->P.ep_rew = nullptr; b->P.ep_cap = 0; return ORL_OK; }  // (the buffers stay for the next arming)
  if (capacity > b->ep_alloc) {  // grow: the old buffer is released, the log is [n_envs][capacity] from here on
    int* lg = nullptr;
    HIPCHK(hipMalloc((void**)&lg, B * (size_t)capacity * sizeof(int) + 64));
    if (b->ep_buf) {
      hipFree(b->ep_buf);
      b->allocs.erase(std::find(b->allocs.begin(), b->allocs.end(), (void*)b->ep_buf));
    }
    b->allocs.push_back(lg);
    b->ep_buf = lg;
    if (b->P.env_type == ENV_QOS) {  // + the float64 reward sums (class rewards)
      double* rw = nullptr;
      HIPCHK(hipMalloc((void**)&rw, B * (size_t)capacity * sizeof(double) + 64));
      if (b->ep_rew_buf) {
        hipFree(b->ep_rew_buf);
        b->allocs.erase(std::find(b->allocs.begin(), b->allocs.end(), (void*)b->ep_rew_buf));
      }
      b->allocs.push_back(rw);
      b->ep_rew_buf = rw;
    }
    b->ep_alloc = capacity;
  }
  if (b->P.env_type == ENV_QOS && !b->P.ep_rew_acc) {
    double* ac = nullptr;
    HIPCHK(hipMalloc((void**)&ac, B * sizeof(double) + 64));
    b->allocs.push_back(ac);
    b->P.ep_rew_acc = ac;
  }
  if (!b->P.ep_count) {
    int* ct = nullptr;
    HIPCHK(hipMalloc((void**)&ct, B * sizeof(int) + 64));
    b->allocs.push_back(ct);
    b->P.ep_count = ct;
  }
  // the row stride of the log is the ARMED capacity (what orl_batch_get_episode_log copies), whatever the buffer could hold
  b->P.ep_log = b->ep_buf;
  b->P.ep_cap = capacity;
  HIPCHK(hipMemsetAsync(b->P.ep_count, 0, B * sizeof(int), b->stream));
  HIPCHK(hipMemsetAsync(b->P.ep_log, 0, B * (size_t)b->P.ep_cap * sizeof(int), b->stream));
  if (b->P.env_type == ENV_QOS) {
    b->P.ep_rew = b->ep_rew_buf;
    HIPCHK(hipMemsetAsync(b->P.ep_rew, 0, B * (size_t)b->P.ep_cap * sizeof(double), b->stream));
    HIPCHK(hipMemsetAsync(b->P.ep_rew_acc, 0, B * sizeof(double), b->stream));
  }
  HIPCHK(hipStreamSynchronize(b->stream));
  return ORL_OK;
}
ORL_ABI_CATCH_INT
extern "C" int orl_batch_get_episode_rewards(orl_batch* b, double* rewards) try {
  if (!b || !rewards) return fail(ORL_E_INVALID, "null argument");
  if (!b->P.ep_log || !b->P.ep_rew) return fail(ORL_E_INVALID, "no reward log: QoSConstrainedRA batches with the episode log armed keep one");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipStreamSynchronize(b->stream));
  HIPCHK(hipMemcpy(rewards, b->P.ep_rew, (size_t)b->P.B * b->P.ep_cap * sizeof(double), hipMemcpyDeviceToHost));
  return ORL_OK;
}
ORL_ABI_CATCH_INT
extern "C" int orl_batch_get_episode_log(orl_batch* b, int32_t* counts, int32_t* accepted) try {
  if (!b || !counts || !accepted) return fail(ORL_E_INVALID, "null argument");
  if (!b->P.ep_log) return fail(ORL_E_INVALID, "the episode log is not armed (orl_batch_episode_log)");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipStreamSynchronize(b->stream));
  HIPCHK(hipMemcpy(counts, b->P.ep_count, (size_t)b->P.B * sizeof(int), hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(accepted, b->P.ep_log, (size_t)b->P.B * b->P.ep_cap * sizeof(int), hipMemcpyDeviceToHost));
  return ORL_OK;
}
ORL_ABI_CATCH_INT

extern "C" int orl_batch_set_paths(orl_batch* b, const int32_t* paths) try {
  if (!b || !paths) return fail(ORL_E_INVALID, "null argument");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipMemcpyAsync(b->P.path_col, paths, (size_t)b->P.B * sizeof(int), hipMemcpyHostToDevice, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  return ORL_OK;
}
ORL_ABI_CATCH_INT

extern "C" int orl_batch_policy(orl_batch* b, int policy_id, int32_t* actions_out) try {
  if (!b) return fail(ORL_E_INVALID, "null batch");
  if (!policy_ok(b, policy_id)) return fail(ORL_E_INVALID, "policy %d is not defined for this env family", policy_id);
  HIPCHK(hipSetDevice(b->device));
  launch_policy(b, policy_id);
  if (actions_out) {
    HIPCHK(hipMemcpyAsync(actions_out, b->P.actions, (size_t)b->P.B * 4 * sizeof(int), hipMemcpyDeviceToHost, b->stream));
    HIPCHK(hipStreamSynchronize(b->stream));
    HIPCHK(hipGetLastError());
  }
  return ORL_OK;
}
ORL_ABI_CATCH_INT

// the index ranges of the reference's actions_output arrays (rmsa_env.py:126-137, 167; rwa_env.py:52-58, 103;
// rmcsa_env.py:145-153, 219); DeepRMSA takes any integer (deeprmsa_env.py:48-58)
static int64_t first_bad_action(const orl_batch* b, const int32_t* a) {
  const DevParams& P = b->P;
  const int rej = P.allow_rejection ? 1 : 0;
  for (i64 i = 0; i < P.B; i++) {
    const int32_t* r = a + 4 * i;
    bool bad = false;
    if (P.env_type == ENV_RMSA) bad = r[0] < 0 || r[0] > P.K || r[1] < 0 || r[1] > P.S;
    else if (P.env_type == ENV_RWA) bad = r[0] < 0 || r[0] >= P.K + rej || r[1] < 0 || r[1] >= P.S + rej;
    else if (P.env_type == ENV_RMCSA) bad = r[0] < 0 || r[0] > P.K || r[1] < 0 || r[1] > P.M || r[2] < 0 || r[2] > P.C || r[3] < 0 || r[3] > P.S;
    else if (P.env_type == ENV_QOS) bad = r[0] < 0 || r[0] >= P.K + rej;  // qos_constrained_ra.py:101
    if (bad) return i;
  }
  return -1;
}

static int flags_to_rc(const orl_batch* b, unsigned int f) {
  if (f & ORL_FLAG_EV_OVERFLOW)
    return fail(ORL_E_OVERFLOW, "an env ran out of pending-release slots (event_capacity %d): its state is no longer valid", b->P.ev_cap);
  if (f & ORL_FLAG_BAD_ACTION)
    return fail(ORL_E_ACTION, "a device-resident action was outside the action space (it was treated as a rejection)");
  return ORL_OK;
}
// after a synchronous call: report what the kernels flagged (device-resident actions cannot be checked beforehand)
static int report_flags(orl_batch* b) {
  unsigned int* f = b->h_tail + 16;
  HIPCHK(hipMemsetAsync(b->d_unfinished + 16, 0, 2 * sizeof(unsigned int), b->stream));
  launch_finish2(b, 0);
  HIPCHK(hipMemcpyAsync(f, b->d_unfinished + 16, 2 * sizeof(unsigned int), hipMemcpyDeviceToHost, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  HIPCHK(hipGetLastError());
  return flags_to_rc(b, f[1]);
}

extern "C" int orl_batch_step(orl_batch* b, const int32_t* actions, int auto_reset, double* obs_out, double* reward_out,
                              uint8_t* done_out, double* info_out) try {
  if (!b) return fail(ORL_E_INVALID, "null batch");
  HIPCHK(hipSetDevice(b->device));
  const size_t B = (size_t)b->P.B;
  if (actions) {
    const int64_t bad = first_bad_action(b, actions);
    if (bad >= 0)  // like the reference's IndexError at its first statement: nothing has been modified
      return fail(ORL_E_ACTION, "action (%d, %d, %d, %d) of env %lld is outside the action space", actions[4 * bad],
                  actions[4 * bad + 1], actions[4 * bad + 2], actions[4 * bad + 3], (long long)bad);
    HIPCHK(hipMemcpyAsync(b->P.actions, actions, B * 4 * sizeof(int), hipMemcpyHostToDevice, b->stream));
  }
  if (b->agent_step) launch_agent_step(b, auto_reset ? 1 : 0);
  else launch_step64(b, auto_reset ? 1 : 0, 1, -1);
  bool any = false;
  if (reward_out) { HIPCHK(hipMemcpyAsync(reward_out, b->P.reward, B * sizeof(double), hipMemcpyDeviceToHost, b->stream)); any = true; }
  if (done_out) { HIPCHK(hipMemcpyAsync(done_out, b->P.done, B, hipMemcpyDeviceToHost, b->stream)); any = true; }
  if (info_out) { HIPCHK(hipMemcpyAsync(info_out, b->P.info, B * b->P.n_info * sizeof(double), hipMemcpyDeviceToHost, b->stream)); any = true; }
  if (obs_out && b->P.obs_dim) { HIPCHK(hipMemcpyAsync(obs_out, b->P.obs, B * b->P.obs_dim * sizeof(double), hipMemcpyDeviceToHost, b->stream)); any = true; }
  if (any) return report_flags(b);  // synchronises
  if (actions) {
    HIPCHK(hipStreamSynchronize(b->stream));
    HIPCHK(hipGetLastError());
  }
  return ORL_OK;
}
ORL_ABI_CATCH_INT

// step() in two halves (SB3's VecEnv.step_async / step_wait): the first validates the host actions and QUEUES everything — the
// copy of the actions, the step kernel, the copies of whatever results are asked for (into the caller's buffers, which must
// stay valid until the second half; page-locked ones from orl_host_alloc make the copies asynchronous), the flag word — on the
// batch's stream and returns; the second waits for the stream and reports like orl_batch_step.  Host work done in between
// (the agent's bookkeeping of the previous step) overlaps the device's.
// one pass over the caller's compact action rows: range check (the reference's IndexError) and expansion into the [n][4] int32
// rows the kernels read; returns the first bad env or -1
template <typename T> static int64_t stage_actions(const orl_batch* b, const T* src, int width, int32_t* dst) {
  const DevParams& P = b->P;
  const int rej = P.allow_rejection ? 1 : 0;
  int hi[4] = {0, 0, 0, 0};  // exclusive upper bounds per column (0: any value, DeepRMSA)
  if (P.env_type == ENV_RMSA) { hi[0] = P.K + 1; hi[1] = P.S + 1; }
  else if (P.env_type == ENV_RWA) { hi[0] = P.K + rej; hi[1] = P.S + rej; }
  else if (P.env_type == ENV_RMCSA) { hi[0] = P.K + 1; hi[1] = P.M + 1; hi[2] = P.C + 1; hi[3] = P.S + 1; }
  else if (P.env_type == ENV_QOS) { hi[0] = P.K + rej; }
  int64_t bad = -1;
  for (i64 i = 0; i < P.B; i++) {
    const T* r = src + (size_t)i * width;
    int32_t* d = dst + 4 * i;
    for (int c = 0; c < 4; c++) {
      const long long v = c < width ? (long long)r[c] : 0;
      if (hi[c] > 0 && c < width && (v < 0 || v >= hi[c]) && bad < 0) bad = i;
      d[c] = (int32_t)v;
    }
  }
  return bad;
}

extern "C" int orl_batch_step_async(orl_batch* b, const void* actions, int action_width, int action_elem_bytes, int auto_reset,
                                    double* obs_out, float* obs_f32_out, double* reward_out, uint8_t* done_out, double* info_out) try {
  if (!b) return fail(ORL_E_INVALID, "null batch");
  if (b->step_pending) return fail(ORL_E_INVALID, "orl_batch_step_async: the previous step has not been waited for");
  HIPCHK(hipSetDevice(b->device));
  const size_t B = (size_t)b->P.B;
  if (actions) {
    if (action_width < 1 || action_width > 4 || (action_elem_bytes != 4 && action_elem_bytes != 8))
      return fail(ORL_E_INVALID, "actions: 1..4 columns of int32 or int64");
    if (!b->h_actions) HIPCHK(hipHostMalloc((void**)&b->h_actions, B * 4 * sizeof(int32_t), hipHostMallocPortable));
    const int64_t bad = action_elem_bytes == 4 ? stage_actions(b, (const int32_t*)actions, action_width, b->h_actions)
                                               : stage_actions(b, (const int64_t*)actions, action_width, b->h_actions);
    if (bad >= 0) {
      const int32_t* r = b->h_actions + 4 * bad;
      return fail(ORL_E_ACTION, "action (%d, %d, %d, %d) of env %lld is outside the action space", r[0], r[1], r[2], r[3], (long long)bad);
    }
    HIPCHK(hipMemcpyAsync(b->P.actions, b->h_actions, B * 4 * sizeof(int), hipMemcpyHostToDevice, b->stream));
  }
  if (b->agent_step) launch_agent_step(b, auto_reset ? 1 : 0);
  else launch_step64(b, auto_reset ? 1 : 0, 1, -1);
  if (reward_out) HIPCHK(hipMemcpyAsync(reward_out, b->P.reward, B * sizeof(double), hipMemcpyDeviceToHost, b->stream));
  if (done_out) HIPCHK(hipMemcpyAsync(done_out, b->P.done, B, hipMemcpyDeviceToHost, b->stream));
  if (info_out) HIPCHK(hipMemcpyAsync(info_out, b->P.info, B * b->P.n_info * sizeof(double), hipMemcpyDeviceToHost, b->stream));
  if (b->P.obs_dim && obs_out)
    HIPCHK(hipMemcpyAsync(obs_out, b->P.obs, B * b->P.obs_dim * sizeof(double), hipMemcpyDeviceToHost, b->stream));
  if (b->P.obs_dim && obs_f32_out) {
    const i64 n = b->P.B * b->P.obs_dim;
    if (!b->obs_f32) {
      HIPCHK(hipMalloc((void**)&b->obs_f32, (size_t)n * sizeof(float) + 64));
      b->allocs.push_back(b->obs_f32);
    }
    hipLaunchKernelGGL(k_cast_f32, dim3(2048), dim3(256), 0, b->stream, b->P.obs, b->obs_f32, n);
    HIPCHK(hipMemcpyAsync(obs_f32_out, b->obs_f32, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, b->stream));
  }
  // what the kernels flagged (device-resident actions cannot be checked beforehand), as report_flags gathers it
  HIPCHK(hipMemsetAsync(b->d_unfinished + 16, 0, 2 * sizeof(unsigned int), b->stream));
  launch_finish2(b, 0);
  HIPCHK(hipMemcpyAsync(b->h_tail + 16, b->d_unfinished + 16, 2 * sizeof(unsigned int), hipMemcpyDeviceToHost, b->stream));
  b->step_pending = 1;
  return ORL_OK;
}
ORL_ABI_CATCH_INT
extern "C" int orl_batch_step_wait(orl_batch* b) try {
  if (!b) return fail(ORL_E_INVALID, "null batch");
  if (!b->step_pending) return fail(ORL_E_INVALID, "orl_batch_step_wait without orl_batch_step_async");
  HIPCHK(hipSetDevice(b->device));
  b->step_pending = 0;
  HIPCHK(hipStreamSynchronize(b->stream));
  HIPCHK(hipGetLastError());
  return flags_to_rc(b, b->h_tail[17]);
}
ORL_ABI_CATCH_INT

// policy + step in one call, for an agent whose action source is one of the library's heuristics (or which wants a heuristic's
// transitions: imitation targets, baselines): where k_agent serves the batch (and k <= 8) ONE launch — the slot scan is the
// step kernel's first phase — else k_policy followed by the step kernel.  Everything is queued on the batch's stream; with no
// output buffer the call does not synchronise (like orl_batch_step(b, NULL, ...)).
extern "C" int orl_batch_policy_step(orl_batch* b, int policy_id, int auto_reset, int32_t* actions_out, double* obs_out, double* reward_out,
                                     uint8_t* done_out, double* info_out) try {
  if (!b) return fail(ORL_E_INVALID, "null batch");
  if (!policy_ok(b, policy_id)) return fail(ORL_E_INVALID, "policy %d is not defined for this env family", policy_id);
  if (b->step_pending) return fail(ORL_E_INVALID, "orl_batch_policy_step: a step queued by orl_batch_step_async has not been waited for");
  HIPCHK(hipSetDevice(b->device));
  const size_t B = (size_t)b->P.B;
  if (b->agent_step && b->P.K <= 8) {
    launch_agent_step(b, auto_reset ? 1 : 0, policy_id);
  } else {
    launch_policy(b, policy_id);
    if (b->agent_step) launch_agent_step(b, auto_reset ? 1 : 0);
    else launch_step64(b, auto_reset ? 1 : 0, 1, -1);
  }
  bool any = false;
  if (actions_out) { HIPCHK(hipMemcpyAsync(actions_out, b->P.actions, B * 4 * sizeof(int), hipMemcpyDeviceToHost, b->stream)); any = true; }
  if (reward_out) { HIPCHK(hipMemcpyAsync(reward_out, b->P.reward, B * sizeof(double), hipMemcpyDeviceToHost, b->stream)); any = true; }
  if (done_out) { HIPCHK(hipMemcpyAsync(done_out, b->P.done, B, hipMemcpyDeviceToHost, b->stream)); any = true; }
  if (info_out) { HIPCHK(hipMemcpyAsync(info_out, b->P.info, B * b->P.n_info * sizeof(double), hipMemcpyDeviceToHost, b->stream)); any = true; }
  if (obs_out && b->P.obs_dim) { HIPCHK(hipMemcpyAsync(obs_out, b->P.obs, B * b->P.obs_dim * sizeof(double), hipMemcpyDeviceToHost, b->stream)); any = true; }
  if (any) return report_flags(b);  // synchronises
  return ORL_OK;
}
ORL_ABI_CATCH_INT

extern "C" int orl_batch_set_info_mode(orl_batch* b, int mode) try {
  if (!b || mode < 0 || mode > 1) return fail(ORL_E_INVALID, "info mode 0 (all entries) or 1 (blocking rates only)");
  if (mode == 1 && b->P.info_mode == 0 && (b->P.env_type == ENV_RMSA || b->P.env_type == ENV_DEEPRMSA) && b->P.n_info >= 8) {
    HIPCHK(hipSetDevice(b->device));
    hipLaunchKernelGGL(k_info_nan, dim3((unsigned)((b->P.B * 4 + 255) / 256)), dim3(256), 0, b->stream, b->P);
  }
  b->P.info_mode = mode;
  return ORL_OK;
}
ORL_ABI_CATCH_INT

extern "C" int orl_batch_check(orl_batch* b) try {
  if (!b) return fail(ORL_E_INVALID, "null batch");
  HIPCHK(hipSetDevice(b->device));
  return report_flags(b);
}
ORL_ABI_CATCH_INT

// (hipHostMallocPortable: page-locked for EVERY device of the process, whichever is current — a MultiDeviceBatch hands slices of
// one such array to the shards of several GPUs, each of which copies into its slice on its own stream)
extern "C" int orl_host_alloc(size_t bytes, void** out) try {
  if (!out || bytes == 0) return fail(ORL_E_INVALID, "bad argument");
  HIPCHK(hipHostMalloc(out, bytes, hipHostMallocPortable));
  memset(*out, 0, bytes);
  return ORL_OK;
}
ORL_ABI_CATCH_INT
extern "C" int orl_host_free(void* p) try {
  if (p) HIPCHK(hipHostFree(p));
  return ORL_OK;
}
ORL_ABI_CATCH_INT

extern "C" int orl_batch_device_buffer(orl_batch* b, int which, void** device_ptr, int64_t* n_elements) try {
  if (!b || !device_ptr || !n_elements) return fail(ORL_E_INVALID, "null argument");
  const int64_t B = b->P.B;
  switch (which) {
    case ORL_BUF_ACTIONS: *device_ptr = b->P.actions; *n_elements = B * 4; break;
    case ORL_BUF_REWARD: *device_ptr = b->P.reward; *n_elements = B; break;
    case ORL_BUF_DONE: *device_ptr = b->P.done; *n_elements = B; break;
    case ORL_BUF_INFO: *device_ptr = b->P.info; *n_elements = B * b->P.n_info; break;
    case ORL_BUF_OBS: *device_ptr = b->P.obs; *n_elements = B * b->P.obs_dim; break;
    case ORL_BUF_TERM_OBS: *device_ptr = b->P.term_obs; *n_elements = B * b->P.obs_dim; break;
    case ORL_BUF_PATHS: *device_ptr = b->P.path_col; *n_elements = B; break;
    default: return fail(ORL_E_INVALID, "unknown buffer %d", which);
  }
  return ORL_OK;
}
ORL_ABI_CATCH_INT

extern "C" int orl_batch_observation(orl_batch* b, double* obs_out) try {
  if (!b || !obs_out) return fail(ORL_E_INVALID, "null argument");
  if (!b->P.obs_dim) return fail(ORL_E_INVALID, "this env family has no array observation");
  HIPCHK(hipSetDevice(b->device));
  launch_obs(b, 0);
  HIPCHK(hipMemcpyAsync(obs_out, b->P.obs, (size_t)b->P.B * b->P.obs_dim * sizeof(double), hipMemcpyDeviceToHost, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  HIPCHK(hipGetLastError());
  return ORL_OK;
}
ORL_ABI_CATCH_INT

extern "C" int orl_batch_get_obs_f32(orl_batch* b, float* obs_out) try {
  if (!b || !obs_out) return fail(ORL_E_INVALID, "null argument");
  if (!b->P.obs_dim) return fail(ORL_E_INVALID, "this env family has no array observation");
  HIPCHK(hipSetDevice(b->device));
  const i64 n = b->P.B * b->P.obs_dim;
  if (!b->obs_f32) {
    HIPCHK(hipMalloc((void**)&b->obs_f32, (size_t)n * sizeof(float) + 64));
    b->allocs.push_back(b->obs_f32);
  }
  hipLaunchKernelGGL(k_cast_f32, dim3(2048), dim3(256), 0, b->stream, b->P.obs, b->obs_f32, n);
  HIPCHK(hipMemcpyAsync(obs_out, b->obs_f32, (size_t)n * sizeof(float), hipMemcpyDeviceToHost, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  HIPCHK(hipGetLastError());
  return ORL_OK;
}
ORL_ABI_CATCH_INT

extern "C" int orl_batch_stream(orl_batch* b, void** hip_stream_out) {
  if (!b || !hip_stream_out) return fail(ORL_E_INVALID, "null argument");
  *hip_stream_out = (void*)b->stream;
  return ORL_OK;
}

extern "C" int orl_batch_get_info_rows(orl_batch* b, const int64_t* env_index, int64_t n, double* info_out) try {
  if (!b || n < 0 || (n > 0 && (!env_index || !info_out))) return fail(ORL_E_INVALID, "bad argument");
  if (n == 0) return ORL_OK;
  if (n > b->P.B) return fail(ORL_E_INVALID, "more rows than envs");
  for (int64_t i = 0; i < n; i++)
    if (env_index[i] < 0 || env_index[i] >= b->P.B) return fail(ORL_E_INVALID, "env index %lld out of range", (long long)env_index[i]);
  HIPCHK(hipSetDevice(b->device));
  const int w = b->P.n_info;
  if (n > b->gather_cap) {
    int64_t cap = b->gather_cap > 0 ? b->gather_cap : 1024;
    while (cap < n) cap *= 2;
    if (cap > b->P.B) cap = b->P.B;
    HIPCHK(hipMalloc((void**)&b->gather_idx, (size_t)cap * sizeof(long long) + 64));
    b->allocs.push_back(b->gather_idx);
    HIPCHK(hipMalloc((void**)&b->gather_out, (size_t)cap * w * sizeof(double) + 64));
    b->allocs.push_back(b->gather_out);
    b->gather_cap = cap;
  }
  HIPCHK(hipMemcpyAsync(b->gather_idx, env_index, (size_t)n * sizeof(long long), hipMemcpyHostToDevice, b->stream));
  hipLaunchKernelGGL(k_gather_rows, dim3((unsigned)((n * w + 255) / 256)), dim3(256), 0, b->stream, (const double*)b->P.info,
                     (const long long*)b->gather_idx, (i64)n, w, b->gather_out);
  HIPCHK(hipMemcpyAsync(info_out, b->gather_out, (size_t)n * w * sizeof(double), hipMemcpyDeviceToHost, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  HIPCHK(hipGetLastError());
  return ORL_OK;
}
ORL_ABI_CATCH_INT

// The logs of a launch of the persistent kernel (deferred statistics: 24 bytes per env-step, DevParams::slog; rows-deferred forms:
// 16 bytes per provision / release, DevParams::elog), allocated by the first device-resident run and sized for the launches that
// run makes — `chunk` steps each, a wavefront that left a launch early catching up over at most two chunks — instead of the 256
// steps' worth every batch used to get at creation (404 MB per 65 536-env batch whether it ever ran a device loop or not).  Each
// log stays below 1 GiB: larger batches run shorter launches.  A later run with longer launches replaces them.
template <typename T> static void dfree(orl_batch* b, T** p) {
  if (!*p) return;
  for (size_t i = 0; i < b->allocs.size(); i++)
    if (b->allocs[i] == (void*)*p) { b->allocs.erase(b->allocs.begin() + (long)i); break; }
  hipFree(*p);
  *p = nullptr;
}
static int ensure_logs(orl_batch* b, int64_t n_steps, int* chunk_io) {
  DevParams& P = b->P;
  if (!b->persist || !orl_persist_deferred(P.env_type)) return ORL_OK;
  const size_t B = (size_t)P.B;
  int chunk = *chunk_io;
  size_t want = (n_steps <= chunk) ? (size_t)(n_steps > 0 ? n_steps : 1) : (size_t)2 * chunk;
  const size_t per_step = (size_t)ORL_SLOG_ROW_WORDS * 8 * B;
  size_t most = ((size_t)1 << 30) / per_step;
  most = most > 256 ? 256 : (most < 2 ? 2 : most);
  if (want > most) want = most;
  if (want < 2) want = 2;
  if (const char* lv = getenv("ORL_LOG_CAP")) { const int v = atoi(lv); if (v >= 2 && v <= 256) want = (size_t)v; }  // tests
  if ((size_t)b->log_cap < want) {
    HIPCHK(hipStreamSynchronize(b->stream));
    if (b->stream2) HIPCHK(hipStreamSynchronize(b->stream2));
    dfree(b, &P.slog);
    dfree(b, &P.elog);
    dfree(b, &P.ssum);
    int rc = dalloc(b, &P.slog, (want + 1) * (size_t)ORL_SLOG_ROW_WORDS * B);
    if (rc) return rc;
    if (!P.log_n) {
      rc = dalloc(b, &P.log_n, (B + 7) / 8 + 16);
      if (rc) return rc;
      // (on the batch's stream: it is non-blocking, so a memset on the null stream is NOT ordered in front of the launch that writes
      // log_n — the first run of a batch could have its first launch's step counts zeroed behind the kernel and its replay skipped:
      // seen once in ~5 runs of the three-thread shard test)
      HIPCHK(hipMemsetAsync(P.log_n, 0, ((B + 7) / 8 + 16) * sizeof(int), b->stream));
    }
    P.log_cap = (int)want;
    P.log_stride = (i64)B;
    b->log_cap = (int)want;
    // events: a step logs its provision and its releases, two per step on average; a wavefront whose envs' logs cannot take
    // another step stops early like one that used up the statistics log
    if (P.env_type != ENV_RMCSA && P.E <= 64) {
      size_t ecap = 3 * want + 40, emost = ((size_t)1 << 30) / (32 * B);
      if (emost < 80) emost = 80;
      if (ecap > emost) ecap = emost;
      if (const char* ev = getenv("ORL_ELOG_CAP")) { const int v = atoi(ev); if (v >= 34 && v <= 4096) ecap = (size_t)v; }  // tests: wavefronts stop for a full event log
      rc = dalloc(b, &P.elog, 2 * ecap * B);
      if (rc) return rc;
      if (!P.elog_n) {
        rc = dalloc(b, &P.elog_n, B + 16);
        if (rc) return rc;
        rc = dalloc(b, &P.bitmap0, B * P.bm_words);
        if (rc) return rc;
      }
      P.elog_cap = (int)ecap;
    }
    // (allocated for every family: k_stats names it in an expression the compiler may evaluate on both sides of a select)
    rc = dalloc(b, &P.ssum, (want + 1) * B);
    if (rc) return rc;
  }
  // (a launch logs at most log_cap steps per wavefront, a straggler up to two chunks)
  if (n_steps > b->log_cap && chunk > b->log_cap / 2) chunk = b->log_cap / 2 > 0 ? b->log_cap / 2 : 1;
  *chunk_io = chunk;
  return ORL_OK;
}

// events created for one call, destroyed on every exit path
struct EventPool {
  std::vector<hipEvent_t> ev;
  ~EventPool() { for (auto& e : ev) hipEventDestroy(e); }
};

extern "C" int orl_batch_run(orl_batch* b, int policy_id, int64_t n_steps, int time_kernels, orl_run_stats* stats) try {
  if (!b || n_steps < 0) return fail(ORL_E_INVALID, "bad argument");
  if (n_steps > INT_MAX) return fail(ORL_E_INVALID, "n_steps must be <= %d per call", INT_MAX);
  if (!policy_ok(b, policy_id)) return fail(ORL_E_INVALID, "policy %d is not defined for this env family", policy_id);
  HIPCHK(hipSetDevice(b->device));
  EventPool pool;
  TkRec tk;
  struct TkGuard { TkRec& t; orl_batch* b; ~TkGuard() { b->tk = nullptr; for (auto& e : t.ev) hipEventDestroy(e); } } tkg{tk, b};
  if (time_kernels == 2) {
    pool.ev.resize((size_t)n_steps * 3);
    for (auto& e : pool.ev) HIPCHK(hipEventCreate(&e));
  }
  HIPCHK(hipStreamSynchronize(b->stream));
  HIPCHK(hipEventRecord(b->ev0, b->stream));
  unsigned int tail[2] = {0, 0};  // straggler workgroups, OR of the env flags
  bool have_flags = false;
  if (!time_kernels && b->persist) {
    // The run is cut into chunks of steps: a wavefront that had to leave its loop for the serial tail resumes in the next
    // launch and is at most one chunk behind (left to one launch per run, it would finish its remaining steps alone on
    // the GPU: 3 000-step runs measured 5.6e8 env-steps/s against 6.3e8 for 100-step runs).  No host synchronisation
    // between chunks; after the last one k_finish2 finalises the state and reduces the flags, and ONE 8-byte copy tells
    // the host whether stragglers are left (then: relaunch) and what the envs flagged.
    const unsigned n_wg = (unsigned)((b->P.B + 7) / 8);
    // d_wg_step counts steps since run_base was 0: between runs every workgroup stands at run_base, so a run needs no clearing
    // (a fill kernel in front of every run: ~1 % of a 20-step run)
    int64_t base_limit = (int64_t)1 << 30;  // the counters are ints
    if (const char* lv = getenv("ORL_RUN_BASE_LIMIT")) { const long long v = atoll(lv); if (v >= 1) base_limit = v; }  // tests
    if (b->wg_dirty || b->run_base + n_steps > base_limit) {
      HIPCHK(hipMemsetAsync(b->d_wg_step, 0, n_wg * sizeof(int), b->stream));
      b->run_base = 0;
    }
    b->wg_dirty = true;  // until this run has completed
    b->run_abandoned = true;
    const int64_t base = b->run_base;
    // (launches of 128 steps: every launch boundary costs a wavefront its window fill / write-back and a cold first step —
    // cfg2 1.265e9 with 64-step launches, 1.295e9 with 128; with the bit-word sink a wavefront practically never has to
    // leave its loop early, so longer launches leave no stragglers behind)
    int chunk = 128;
    if (const char* cv = getenv("ORL_PERSIST_CHUNK")) { int v = atoi(cv); if (v >= 1) chunk = v; }
    // (deferred statistics: the logs a launch writes, allocated on first use; a launch logs at most log_cap steps per wavefront)
    { const int lrc = ensure_logs(b, n_steps, &chunk); if (lrc) return lrc; }
    // A launch occupies the GPU in rounds of `resident` wavefronts, and a last round that is not full leaves CUs idle until
    // the launch ends (cfg2: 8 192 wavefronts over 3 072 resident = 2.67 rounds, 11 % of the machine-time lost).  When the
    // rounds do not come out even, the batch runs as two halves on two streams: the tail of one half's launch overlaps the
    // other half's next one.  (ORL_PERSIST_PARTS=1|2 forces either.)
    int parts = 1;
    {
      const int resident = persist_resident(b);
      const double rounds = (double)n_wg / (double)(resident > 0 ? resident : 1);
      // (a run of a single chunk has no next launch to overlap with: one part)
      if (n_steps > chunk && n_wg >= 2048 && rounds > 1.0) parts = 2;
      if (const char* pv = getenv("ORL_PERSIST_PARTS")) { int v = atoi(pv); if (v == 1 || (v == 2 && n_wg >= 2)) parts = v; }
    }
    const i64 half = parts == 2 ? (i64)((n_wg + 1) / 2) * 8 : b->P.B;
    DevParams view[2] = {env_view(b->P, 0, parts == 2 ? half : b->P.B, 0), env_view(b->P, parts == 2 ? half : 0, parts == 2 ? b->P.B - half : 0, 1)};
    hipStream_t strm[2] = {b->stream, b->stream2};
    int* wg_step[2] = {b->d_wg_step, b->d_wg_step + half / 8};
    if (parts == 2) {
      HIPCHK(hipEventRecord(b->ev_half, b->stream));
      HIPCHK(hipStreamWaitEvent(b->stream2, b->ev_half, 0));
    }
    b->persist_launches = 0;
    for (int64_t tgt = 0; tgt < n_steps;) {
      tgt = (tgt + chunk < n_steps) ? tgt + chunk : n_steps;
      for (int attempt = 0;; attempt++) {
        b->persist_launches++;
        unsigned int* cnt[2] = {nullptr, nullptr};  // the slot each half's launch counts into; it clears the other one
        // the launch that reaches the end of the run finishes the state itself (pending network-compactness update, flags:
        // DevParams::persist_finish); only a relaunch for stragglers — wavefronts that left that launch with releases still
        // to do in place — is followed by k_finish2, which does the same for every env in a launch of its own
        const int finish = (tgt >= n_steps && attempt == 0) ? 1 : 0;
        for (int p = 0; p < parts; p++) {
          cnt[p] = b->d_unfinished + 8 * p + 4 * b->un_slot[p];
          launch_persist(b, view[p], strm[p], policy_id, (int)(base + tgt), wg_step[p], cnt[p], b->d_unfinished + 8 * p + 4 * (b->un_slot[p] ^ 1), finish);
          b->un_slot[p] ^= 1;
        }
        if (tgt < n_steps) break;  // stragglers catch up in the next chunk's launch
        if (attempt > 0)
          for (int p = 0; p < parts; p++)  // (harmless for a straggler: it does what that env's next control phase would do first)
            hipLaunchKernelGGL(k_finish2, dim3((unsigned)((view[p].B + 255) / 256)), dim3(256), 0, strm[p], view[p], 1, cnt[p] + 1);
        if (parts == 2) {
          HIPCHK(hipEventRecord(b->ev_half, b->stream2));
          HIPCHK(hipStreamWaitEvent(b->stream, b->ev_half, 0));
        }
        HIPCHK(hipEventRecord(b->ev1, b->stream));
        unsigned int* both = b->h_tail;
        HIPCHK(hipMemcpyAsync(both, b->d_unfinished, 16 * sizeof(unsigned int), hipMemcpyDeviceToHost, b->stream));
        HIPCHK(hipStreamSynchronize(b->stream));
        const unsigned int* c0 = both + (cnt[0] - b->d_unfinished);
        const unsigned int* c1 = parts == 2 ? both + (cnt[1] - b->d_unfinished) : nullptr;
        tail[0] = c0[0] + (c1 ? c1[0] : 0);
        tail[1] |= c0[1] | (c1 ? c1[1] : 0);  // (accumulated: k_finish2 clears a reported flag in the records, a relaunch would lose it)
        if (!tail[0]) break;
      }
    }
    if (n_steps == 0) HIPCHK(hipEventRecord(b->ev1, b->stream));
    have_flags = n_steps > 0;
    b->run_base = base + n_steps;  // every workgroup stands here now
    b->wg_dirty = false;
    b->run_abandoned = false;
  } else if (time_kernels == 2) {
    for (int64_t s = 0; s < n_steps; s++) {
      HIPCHK(hipEventRecord(pool.ev[3 * s], b->stream));
      launch_policy(b, policy_id);
      HIPCHK(hipEventRecord(pool.ev[3 * s + 1], b->stream));
      launch_step64(b, 1, 0, -1);
      HIPCHK(hipEventRecord(pool.ev[3 * s + 2], b->stream));
    }
  } else {
    // the per-env kernel with the slot scan inside it (one launch per policy + step), or — ORL_ALT_IMPLS builds — the
    // two-kernel form of the persistent kernel's phases; time_kernels == 1: an event after every kernel
    for (int64_t s = 0; s < n_steps; s++) {
      if (time_kernels == 1) { b->tk = &tk; ORL_TK(b, ""); }
      if (b->two_kernel) launch_step2(b, policy_id);
      // (QoSConstrainedRA batches that step through k_agent_qos: the stand-alone scan + that kernel, two launches that together
      // take half the time of k_step with the scan inside it — 65 536 envs 129 against 233 us)
      else if (b->agent_step && b->P.env_type == ENV_QOS) launch_agent_step(b, 1, policy_id);
      else launch_step64(b, 1, 0, policy_id);
      b->tk = nullptr;
    }
    if (b->two_kernel) launch_finish2(b, 1);
  }
  if (!(!time_kernels && b->persist)) HIPCHK(hipEventRecord(b->ev1, b->stream));
  HIPCHK(hipStreamSynchronize(b->stream));
  HIPCHK(hipGetLastError());
  if (stats) {
    float ms = 0;
    HIPCHK(hipEventElapsedTime(&ms, b->ev0, b->ev1));
    memset(stats, 0, sizeof *stats);
    stats->ms_total = ms;
    stats->launches = n_steps;
    if (!time_kernels && b->persist && n_steps > 0) {  // the whole run was (re)launches of one kernel
      stats->n_kernels = 1;
      stats->launches = b->persist_launches > 0 ? b->persist_launches : 1;  // chunks of ORL_PERSIST_CHUNK steps
      stats->ms_kernel[0] = ms / (double)stats->launches;                   // average duration of one launch (+ its k_rel_tail)
      snprintf(stats->kernel_name[0], sizeof stats->kernel_name[0], "k_persist");
    }
    if (time_kernels == 2 && n_steps > 0) {
      double sp = 0, ss = 0;
      for (int64_t s = 0; s < n_steps; s++) {
        float a = 0, c2 = 0;
        HIPCHK(hipEventElapsedTime(&a, pool.ev[3 * s], pool.ev[3 * s + 1]));
        HIPCHK(hipEventElapsedTime(&c2, pool.ev[3 * s + 1], pool.ev[3 * s + 2]));
        sp += a; ss += c2;
      }
      stats->ms_policy = sp / (double)n_steps;
      stats->ms_step = ss / (double)n_steps;
      stats->launches = 2 * n_steps;
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
  return have_flags ? flags_to_rc(b, tail[1]) : report_flags(b);
}
ORL_ABI_CATCH_INT

// ---- read-back ----------------------------------------------------------------------------------
static int fetch_scal(orl_batch* b, std::vector<u64>& host) {
  host.resize((size_t)b->P.B * ORL_SCAL_WORDS);
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipStreamSynchronize(b->stream));
  HIPCHK(hipMemcpy(host.data(), b->P.scal, host.size() * sizeof(u64), hipMemcpyDeviceToHost));
  return 0;
}
static double as_f64(u64 v) { double d; memcpy(&d, &v, 8); return d; }

extern "C" int orl_batch_get_counters(orl_batch* b, int64_t* out) try {
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
ORL_ABI_CATCH_INT
extern "C" int orl_batch_get_services(orl_batch* b, double* out) try {
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
ORL_ABI_CATCH_INT
extern "C" int orl_batch_get_active(orl_batch* b, int32_t* out) try {
  if (!b || !out) return fail(ORL_E_INVALID, "null argument");
  std::vector<u64> h;
  if (fetch_scal(b, h)) return ORL_E_HIP;
  for (i64 i = 0; i < b->P.B; i++) out[i] = (int32_t)(h[(size_t)i * ORL_SCAL_WORDS + SC_EV] >> 32);
  return ORL_OK;
}
ORL_ABI_CATCH_INT
extern "C" int orl_batch_get_flags(orl_batch* b, int32_t* out) try {
  if (!b || !out) return fail(ORL_E_INVALID, "null argument");
  std::vector<u64> h;
  if (fetch_scal(b, h)) return ORL_E_HIP;
  for (i64 i = 0; i < b->P.B; i++) out[i] = (int32_t)(h[(size_t)i * ORL_SCAL_WORDS + SC_FLAGS] >> 32);
  return ORL_OK;
}
ORL_ABI_CATCH_INT
extern "C" int orl_batch_get_slots(orl_batch* b, int64_t env, uint8_t* out) try {
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
ORL_ABI_CATCH_INT
extern "C" int orl_batch_get_spectrum(orl_batch* b, int64_t env, int32_t* out) try {
  if (!b || !out || env < 0 || env >= b->P.B) return fail(ORL_E_INVALID, "bad argument");
  if (b->P.env_type != ENV_QOS) return fail(ORL_E_INVALID, "available_spectrum counters exist for QoSConstrainedRA only");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipStreamSynchronize(b->stream));
  std::vector<u64> h((size_t)b->P.E);
  HIPCHK(hipMemcpy(h.data(), b->P.bitmap + env * b->P.bm_words, h.size() * 8, hipMemcpyDeviceToHost));
  for (int l = 0; l < b->P.E; l++) out[l] = (int32_t)h[(size_t)l];
  return ORL_OK;
}
ORL_ABI_CATCH_INT
extern "C" int orl_batch_get_link_stats(orl_batch* b, int64_t env, double* out) try {
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
ORL_ABI_CATCH_INT
extern "C" int orl_batch_get_net_stats(orl_batch* b, int64_t env, double* out) try {
  if (!b || !out || env < 0 || env >= b->P.B) return fail(ORL_E_INVALID, "bad argument");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipStreamSynchronize(b->stream));
  u64 s[ORL_SCAL_WORDS];
  HIPCHK(hipMemcpy(s, b->P.scal + env * ORL_SCAL_WORDS, sizeof s, hipMemcpyDeviceToHost));
  out[0] = as_f64(s[SC_GTHR]); out[1] = as_f64(s[SC_GCOMP]); out[2] = as_f64(s[SC_GLAST]); out[3] = as_f64(s[SC_NOW]);
  return ORL_OK;
}
ORL_ABI_CATCH_INT
extern "C" int orl_batch_row_words(const orl_batch* b) { return b ? b->wt : 0; }
extern "C" int orl_batch_map_words(const orl_batch* b) { return b ? b->P.bm_words : 0; }
extern "C" int orl_batch_get_slots_packed(orl_batch* b, uint64_t* out) try {
  if (!b || !out) return fail(ORL_E_INVALID, "null argument");
  if (b->P.env_type == ENV_QOS) return fail(ORL_E_INVALID, "QoSConstrainedRA keeps counters, not slot maps");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipStreamSynchronize(b->stream));
  HIPCHK(hipMemcpy(out, b->P.bitmap, (size_t)b->P.B * b->P.bm_words * 8, hipMemcpyDeviceToHost));
  return ORL_OK;
}
ORL_ABI_CATCH_INT
extern "C" int orl_batch_get_link_stats_all(orl_batch* b, double* out) try {
  if (!b || !out) return fail(ORL_E_INVALID, "null argument");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipStreamSynchronize(b->stream));
  const size_t E = (size_t)b->P.E, B = (size_t)b->P.B;
  std::vector<double> h(B * 4 * E);  // device layout [B][E][4] -> ABI layout [B][4][E]
  HIPCHK(hipMemcpy(h.data(), b->P.lstat, h.size() * 8, hipMemcpyDeviceToHost));
  for (size_t i = 0; i < B; i++)
    for (size_t l = 0; l < E; l++)
      for (size_t k = 0; k < 4; k++) out[(i * 4 + k) * E + l] = h[(i * E + l) * 4 + k];
  return ORL_OK;
}
ORL_ABI_CATCH_INT
extern "C" int orl_batch_get_net_stats_all(orl_batch* b, double* out) try {
  if (!b || !out) return fail(ORL_E_INVALID, "null argument");
  std::vector<u64> h;
  if (fetch_scal(b, h)) return ORL_E_HIP;
  for (i64 i = 0; i < b->P.B; i++) {
    const u64* s = &h[(size_t)i * ORL_SCAL_WORDS];
    out[4 * i] = as_f64(s[SC_GTHR]); out[4 * i + 1] = as_f64(s[SC_GCOMP]); out[4 * i + 2] = as_f64(s[SC_GLAST]); out[4 * i + 3] = as_f64(s[SC_NOW]);
  }
  return ORL_OK;
}
ORL_ABI_CATCH_INT
extern "C" int orl_batch_get_action_histograms(orl_batch* b, int64_t env, int32_t* out) try {
  if (!b || !out || env < 0 || env >= b->P.B) return fail(ORL_E_INVALID, "bad argument");
  if (!b->P.act2d) return fail(ORL_E_INVALID, "the batch was created without action_histograms");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipStreamSynchronize(b->stream));
  HIPCHK(hipMemcpy(out, b->P.act2d + env * b->P.act2d_words, (size_t)b->P.act2d_words * sizeof(int), hipMemcpyDeviceToHost));
  return ORL_OK;
}
ORL_ABI_CATCH_INT
extern "C" int orl_batch_get_pending(orl_batch* b, int64_t env, int32_t capacity, double* time_out, int32_t* rec_out) try {
  if (!b || env < 0 || env >= b->P.B || capacity < 0) return fail(ORL_E_INVALID, "bad argument");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipStreamSynchronize(b->stream));
  const int cap = b->P.ev_cap;
  std::vector<double> t((size_t)cap);
  std::vector<u64> inf((size_t)cap);
  HIPCHK(hipMemcpy(t.data(), b->P.ev_time + env * cap, (size_t)cap * 8, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(inf.data(), b->P.ev_info + env * cap, (size_t)cap * 8, hipMemcpyDeviceToHost));
  u64 ev = 0;
  HIPCHK(hipMemcpy(&ev, b->P.scal + env * ORL_SCAL_WORDS + SC_EV, 8, hipMemcpyDeviceToHost));
  const int hwm = (int)(u32)ev;
  int n = 0;
  for (int i = 0; i < hwm && i < cap; i++) {
    if (!(t[(size_t)i] < INFINITY)) continue;
    if (n < capacity && time_out && rec_out) {
      const u64 v = inf[(size_t)i];
      const int pidx = (int)(v & 0xffffffu);
      time_out[n] = t[(size_t)i];
      int32_t* r = rec_out + 6 * n;
      r[0] = pidx / b->P.K;                  // src * N + dst
      r[1] = pidx % b->P.K;                  // path index within the pair
      r[2] = (int)((v >> 24) & 0xfffu);      // initial slot
      r[3] = (int)((v >> 36) & 0xffu);       // number of slots
      r[4] = (int)((v >> 44) & 0x1fu);       // core
      r[5] = (int)((v >> 49) & 0x7fffu);     // bit rate
    }
    n++;
  }
  return n;
}
ORL_ABI_CATCH_INT
extern "C" int orl_batch_totals(orl_batch* b, int64_t* processed, int64_t* accepted) try {
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
ORL_ABI_CATCH_INT

/* debug: stream the slot-map array (n_envs * bm_words * 8 bytes) once; returns that byte count */
extern "C" int64_t orl_batch_debug_stream_read(orl_batch* b, int width16) try {
  if (!b) return fail(ORL_E_INVALID, "null batch");
  if (hipSetDevice(b->device) != hipSuccess) return ORL_E_HIP;
  i64 n_words = b->P.B * b->P.bm_words;
  hipLaunchKernelGGL(k_calib_read, dim3(2048), dim3(256), 0, b->stream, b->P.bitmap, n_words, width16, (u64*)b->d_totals);
  if (hipStreamSynchronize(b->stream) != hipSuccess) return ORL_E_HIP;
  return n_words * 8;
}
ORL_ABI_CATCH_INT

/* debug: what this GPU's HBM delivers to the library's own plain streaming kernels, measured now (HIP events, best of `reps`):
 * a 16-byte-per-lane read of `bytes` bytes (k_calib_read, the kernel the counter calibration uses) and a 16-byte-per-lane copy of
 * them (read + write bytes counted).  bench.py reports both beside the 8 TB/s spec peak (roofline.peak_measured). */
extern "C" int orl_debug_stream_peak(int device, int64_t bytes, int reps, double* read_gbs, double* copy_gbs) try {
  if (bytes < (1 << 20) || reps < 1 || !read_gbs || !copy_gbs) return fail(ORL_E_INVALID, "bad argument");
  HIPCHK(hipSetDevice(device));
  const i64 n16 = bytes / 16;
  void *a = nullptr, *c = nullptr;
  u64* sink = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipStream_t st = nullptr;
  int rc = ORL_OK;
  double best_r = 0.0, best_c = 0.0;
  if (hipMalloc(&a, (size_t)n16 * 16) != hipSuccess || hipMalloc(&c, (size_t)n16 * 16) != hipSuccess || hipMalloc((void**)&sink, 64) != hipSuccess ||
      hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess || hipStreamCreate(&st) != hipSuccess) {
    rc = fail(ORL_E_HIP, "orl_debug_stream_peak: allocation failed");
  } else {
    hipMemsetAsync(a, 1, (size_t)n16 * 16, st);
    hipMemsetAsync(c, 2, (size_t)n16 * 16, st);
    for (int r = 0; r < reps + 2 && rc == ORL_OK; r++) {  // (two untimed launches of each first)
      float ms_r = 0.f, ms_c = 0.f;
      hipEventRecord(e0, st);
      hipLaunchKernelGGL(k_calib_read, dim3(4096), dim3(256), 0, st, (const u64*)a, n16 * 2, 1, sink);
      hipEventRecord(e1, st);
      if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms_r, e0, e1) != hipSuccess) { rc = fail(ORL_E_HIP, "stream read failed"); break; }
      hipEventRecord(e0, st);
      hipLaunchKernelGGL(k_calib_copy, dim3(4096), dim3(256), 0, st, (const ulonglong2*)a, (ulonglong2*)c, n16);
      hipEventRecord(e1, st);
      if (hipEventSynchronize(e1) != hipSuccess || hipEventElapsedTime(&ms_c, e0, e1) != hipSuccess) { rc = fail(ORL_E_HIP, "stream copy failed"); break; }
      if (r >= 2) {
        const double gr = (double)n16 * 16.0 / (ms_r * 1e-3) / 1e9, gc = 2.0 * (double)n16 * 16.0 / (ms_c * 1e-3) / 1e9;
        best_r = gr > best_r ? gr : best_r;
        best_c = gc > best_c ? gc : best_c;
      }
    }
  }
  if (st) hipStreamDestroy(st);
  if (e0) hipEventDestroy(e0);
  if (e1) hipEventDestroy(e1);
  if (a) hipFree(a);
  if (c) hipFree(c);
  if (sink) hipFree(sink);
  *read_gbs = best_r;
  *copy_gbs = best_c;
  return rc;
}
ORL_ABI_CATCH_INT

extern "C" int orl_batch_matrix_obs_dim(const orl_batch* b) { return b ? 2 * b->P.N + b->P.C * b->P.E * b->P.S : 0; }

extern "C" int orl_batch_matrix_observation(orl_batch* b, uint8_t* out) try {
  if (!b || !out) return fail(ORL_E_INVALID, "null argument");
  HIPCHK(hipSetDevice(b->device));
  const size_t dim = (size_t)orl_batch_matrix_obs_dim(b), B = (size_t)b->P.B;
  unsigned char* d = nullptr;
  HIPCHK(hipMalloc((void**)&d, B * dim));
  hipLaunchKernelGGL(k_matrix_obs, dim3((unsigned)B), dim3(256), 0, b->stream, b->P, d);
  hipError_t e = hipMemcpyAsync(out, d, B * dim, hipMemcpyDeviceToHost, b->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
  hipFree(d);
  if (e != hipSuccess) return fail(ORL_E_HIP, "matrix observation failed: %s", hipGetErrorString(e));
  HIPCHK(hipGetLastError());
  return ORL_OK;
}
ORL_ABI_CATCH_INT

// ---- specialisation libraries: k_persist with one configuration's sizes as compile-time constants --------------------------
static void persist_form_of(const DevParams& P, int wt, int* lds, int* waves) {
  switch (wt) {
    case 1: orl_launch::persist_form<1>(P, lds, waves); break;
    case 2: orl_launch::persist_form<2>(P, lds, waves); break;
    case 5: orl_launch::persist_form<5>(P, lds, waves); break;
    default: orl_launch::persist_form<8>(P, lds, waves); break;
  }
}
static int spec_flags(const DevParams& P, int wt, char* buf, int capacity) {
  int lds = 0, waves = 0;
  persist_form_of(P, wt, &lds, &waves);
  const int n = snprintf(buf, (size_t)capacity,
                         "-DORL_SPEC_ONLY -DORL_W=%d -DORL_SPEC_ENV=%d -DORL_SPEC_LDS=%d -DORL_SPEC_WAVES=%d -DORL_SPEC_RW=%d -DORL_SPEC_N=%d -DORL_SPEC_E=%d "
                         "-DORL_SPEC_K=%d -DORL_SPEC_H=%d -DORL_SPEC_M=%d -DORL_SPEC_S=%d -DORL_SPEC_C=%d -DORL_SPEC_J=%d -DORL_SPEC_BRMODE=%d "
                         "-DORL_SPEC_BRLO=%d -DORL_SPEC_NBR=%d -DORL_SPEC_RANDN=%d -DORL_SPEC_RANDBITS=%d -DORL_SPEC_EVCAP=%d "
                         "-DORL_SPEC_BMWORDS=%d -DORL_SPEC_CSWORDS=%d -DORL_SPEC_OBSDIM=%d -DORL_SPEC_NINFO=%d",
                         wt, P.env_type, lds, waves & 15, waves >> 4, P.N, P.E, P.K, P.H, P.M, P.S, P.C, P.J, P.bit_rate_mode, P.br_lo, P.n_br, P.rand_n,
                         P.rand_bits, P.ev_cap, P.bm_words, P.cs_words, P.obs_dim, P.n_info);
  return (n > 0 && n < capacity) ? n : 0;
}
extern "C" int orl_batch_spec_flags(orl_batch* b, char* buf, int capacity) try {
  if (!b || !buf || capacity < 1) return 0;
  buf[0] = 0;
  if (!b->persist) return 0;
  return spec_flags(b->P, b->wt, buf, capacity);
}
catch (...) { return 0; }
extern "C" int orl_spec_flags_for(const orl_env_config* cfg, const orl_topology_desc* topo, char* buf, int capacity) {
  return orl_spec_flags_for_batch(cfg, topo, (int64_t)1 << 20, buf, capacity);
}
extern "C" int orl_spec_flags_for_batch(const orl_env_config* cfg, const orl_topology_desc* topo, int64_t n_envs, char* buf, int capacity) try {
  if (!cfg || !topo || !buf || capacity < 1 || n_envs < 1) return 0;
  buf[0] = 0;
  if (cfg->struct_size != sizeof(orl_env_config) || cfg->env_type < 0 || cfg->env_type > ORL_ENV_QOS || !(cfg->lambda_arrival > 0) || !(cfg->lambda_holding > 0)) return 0;
  DevParams P;
  memset(&P, 0, sizeof P);
  int wt = 1;
  derive_sizes(cfg, topo->n_nodes, topo->n_links, topo->k_paths, topo->max_hops, topo->n_modulations, n_envs, P, &wt);
  if (!pipeline_applies(cfg, P)) return 0;
  return spec_flags(P, wt, buf, capacity);
}
catch (...) { return 0; }
extern "C" int orl_batch_load_spec(orl_batch* b, const char* so_path) try {
  if (!b || !so_path) return fail(ORL_E_INVALID, "null argument");
  if (!b->persist) return fail(ORL_E_INVALID, "this batch does not run the persistent kernel");
  void* h = dlopen(so_path, RTLD_NOW | RTLD_LOCAL);
  if (!h) return fail(ORL_E_INVALID, "dlopen(%s): %s", so_path, dlerror());
  typedef int (*bytes_fn)(void);
  typedef void (*describe_fn)(int*);
  bytes_fn bytes = (bytes_fn)dlsym(h, "orl_spec_struct_bytes");
  describe_fn describe = (describe_fn)dlsym(h, "orl_spec_describe");
  void* launch = dlsym(h, "orl_spec_launch");
  if (!bytes || !describe || !launch) { dlclose(h); return fail(ORL_E_INVALID, "%s is not a specialisation library", so_path); }
  if (bytes() != (int)sizeof(DevParams)) { dlclose(h); return fail(ORL_E_INVALID, "%s was built from other sources (parameter block %d B, here %zu B)", so_path, bytes(), sizeof(DevParams)); }
  int d[22];
  describe(d);
  const DevParams& P = b->P;
  const int want[22] = {P.env_type, b->wt, d[2], d[3], P.N, P.E, P.K, P.H, P.M, P.S, P.C, P.J, P.bit_rate_mode, P.br_lo, P.n_br, P.rand_n, P.rand_bits,
                        P.ev_cap, P.bm_words, P.cs_words, P.obs_dim, P.n_info};
  for (int i = 0; i < 22; i++)
    if (d[i] != want[i]) { dlclose(h); return fail(ORL_E_INVALID, "%s was built for another configuration (field %d: %d, this batch %d)", so_path, i, d[i], want[i]); }
  if (b->spec_handle) {  // kernels of the library being replaced may still be queued
    HIPCHK(hipSetDevice(b->device));
    HIPCHK(hipStreamSynchronize(b->stream));
    if (b->stream2) HIPCHK(hipStreamSynchronize(b->stream2));
    dlclose(b->spec_handle);
  }
  b->spec_handle = h;
  b->spec_launch = (decltype(b->spec_launch))launch;
  b->spec_agent_launch = (decltype(b->spec_agent_launch))dlsym(h, "orl_spec_agent_launch");  // (k_agent with the same constants)
  b->spec_lds = d[2];
  b->spec_waves = d[3];
  return ORL_OK;
}
ORL_ABI_CATCH_INT

// ---- several devices, one process: contiguous shards of the env index range, one orl_batch per shard ----------------------
struct orl_multi {
  std::vector<orl_topology*> topo;
  std::vector<orl_batch*> shard;
  std::vector<int64_t> first;  // [n_shards + 1]
};
extern "C" void orl_multi_destroy(orl_multi* m) try {
  if (!m) return;
  for (orl_batch* b : m->shard) orl_batch_destroy(b);
  for (orl_topology* t : m->topo) orl_topology_destroy(t);
  delete m;
}
ORL_ABI_CATCH_VOID
extern "C" int orl_multi_create(const orl_env_config* cfg, const orl_topology_desc* topo, int64_t n_envs, const int64_t* seeds,
                                int n_devices, const int* device_ids, orl_multi** out) try {
  if (!cfg || !topo || !seeds || !device_ids || !out) return fail(ORL_E_INVALID, "null argument");
  if (n_devices < 1 || n_devices > 64 || n_envs < n_devices) return fail(ORL_E_INVALID, "need 1..64 devices and at least one env per device");
  const int have = orl_device_count();
  for (int r = 0; r < n_devices; r++)
    if (device_ids[r] < 0 || device_ids[r] >= have) return fail(ORL_E_INVALID, "device %d is not visible (%d GPU(s))", device_ids[r], have);
  struct Destroy { void operator()(orl_multi* p) const { orl_multi_destroy(p); } };
  std::unique_ptr<orl_multi, Destroy> m(new orl_multi());
  m->first.push_back(0);
  const int64_t per = n_envs / n_devices, rem = n_envs % n_devices;  // balanced, contiguous (sharding.shard_range)
  for (int r = 0; r < n_devices; r++) {
    const int64_t n = per + (r < rem ? 1 : 0);
    orl_topology* t = nullptr;
    int rc = orl_topology_create(topo, device_ids[r], &t);
    if (rc) return rc;
    m->topo.push_back(t);
    orl_batch* b = nullptr;
    rc = orl_batch_create_seeded(cfg, t, n, seeds + m->first.back(), &b);
    if (rc) return rc;
    m->shard.push_back(b);
    m->first.push_back(m->first.back() + n);
  }
  *out = m.release();
  return ORL_OK;
}
ORL_ABI_CATCH_INT
extern "C" int orl_multi_n_shards(const orl_multi* m) { return m ? (int)m->shard.size() : 0; }
extern "C" orl_batch* orl_multi_shard(orl_multi* m, int r, int64_t* first_env, int64_t* n_envs) {
  if (!m || r < 0 || r >= (int)m->shard.size()) { fail(ORL_E_INVALID, "no such shard"); return nullptr; }
  if (first_env) *first_env = m->first[(size_t)r];
  if (n_envs) *n_envs = m->first[(size_t)r + 1] - m->first[(size_t)r];
  return m->shard[(size_t)r];
}
extern "C" int orl_multi_run(orl_multi* m, int policy_id, int64_t n_steps, orl_run_stats* stats) try {
  if (!m) return fail(ORL_E_INVALID, "null group");
  const size_t n = m->shard.size();
  std::vector<int> rc(n, 0);
  std::vector<std::string> msg(n);
  std::vector<std::thread> th;
  for (size_t r = 0; r < n; r++)
    th.emplace_back([&, r] {  // (the error text is thread-local: carried back by value)
      rc[r] = orl_batch_run(m->shard[r], policy_id, n_steps, 0, stats ? stats + r : nullptr);
      if (rc[r]) msg[r] = orl_last_error();
    });
  for (auto& t : th) t.join();
  for (size_t r = 0; r < n; r++)
    if (rc[r]) return fail(rc[r], "shard %zu: %s", r, msg[r].c_str());
  return ORL_OK;
}
ORL_ABI_CATCH_INT

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
  if (P.act2d) v.push_back({P.act2d, B * (size_t)P.act2d_words * 4});
  if (P.mt2) v.push_back({P.mt2, B * 624 * 4});
  return v;
}
extern "C" int64_t orl_batch_state_bytes(orl_batch* b) try {
  if (!b) return fail(ORL_E_INVALID, "null batch");
  int64_t n = 0;
  for (auto& s : state_sections(b)) n += (int64_t)s.bytes;
  return n;
}
ORL_ABI_CATCH_INT
extern "C" int orl_batch_get_state(orl_batch* b, void* out) try {
  if (!b || !out) return fail(ORL_E_INVALID, "null argument");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipStreamSynchronize(b->stream));
  // (a device-resident run that did not complete — a HIP error mid-launch — may have left services drawn ahead and envs at
  // different step counts: not a state a snapshot can resume from exactly)
  if (b->run_abandoned)
    return fail(ORL_E_INVALID, "the last device-resident run did not complete: reset or restore the batch before taking a snapshot");
  unsigned char* o = (unsigned char*)out;
  for (auto& s : state_sections(b)) { HIPCHK(hipMemcpy(o, s.ptr, s.bytes, hipMemcpyDeviceToHost)); o += s.bytes; }
  return ORL_OK;
}
ORL_ABI_CATCH_INT
extern "C" int orl_batch_set_state(orl_batch* b, const void* in) try {
  if (!b || !in) return fail(ORL_E_INVALID, "null argument");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipStreamSynchronize(b->stream));
  const unsigned char* o = (const unsigned char*)in;
  for (auto& s : state_sections(b)) { HIPCHK(hipMemcpy(s.ptr, o, s.bytes, hipMemcpyHostToDevice)); o += s.bytes; }
  // services the persistent kernel drew ahead and parked when a run was abandoned mid-launch (a HIP error) belong to the
  // generator state that has just been replaced: a snapshot is always taken between runs, where nothing is parked
  if (b->P.svc_cnt) HIPCHK(hipMemsetAsync(b->P.svc_cnt, 0, (size_t)((b->P.B + 7) / 8) * 64 * sizeof(int), b->stream));
  b->run_abandoned = false;  // (the step counters of an abandoned run are cleared by the next run: wg_dirty stays set)
  slot_maps_change(b);
  if (b->P.obs_dim) launch_obs(b, 0);
  HIPCHK(hipStreamSynchronize(b->stream));
  return ORL_OK;
}
ORL_ABI_CATCH_INT

/* diagnostic builds (-DORL_TIMING): shader-clock cycles per phase of the persistent kernel, summed over wavefronts */
extern "C" int orl_batch_debug_prof(orl_batch* b, uint64_t* out48, int reset) try {
  if (!b || !out48) return fail(ORL_E_INVALID, "bad argument");
  HIPCHK(hipSetDevice(b->device));
  HIPCHK(hipDeviceSynchronize());
  int rc = 0;
#define CALL(WW) rc = orl_launch::prof_read<WW>((unsigned long long*)out48, reset)
  ORL_DISPATCH_W(b, CALL)
#undef CALL
  return rc ? fail(ORL_E_HIP, "reading the profile failed") : ORL_OK;
}
ORL_ABI_CATCH_INT
extern "C" int orl_batch_debug_persist_spec(orl_batch* b) try {
  if (!b) return -1;
  return b->persist ? b->persist_spec : -1;
}
ORL_ABI_CATCH_INT
extern "C" int orl_batch_debug_persist_form(orl_batch* b) try {
  if (!b) return -1;
  return b->persist ? b->persist_form_last : -1;
}
ORL_ABI_CATCH_INT
extern "C" int orl_batch_debug_step_kernel(orl_batch* b) try {
  if (!b) return -1;
  return b->agent_step ? 2 : 0;
}
ORL_ABI_CATCH_INT
extern "C" int64_t orl_batch_debug_serial_count(orl_batch* b) try {
  if (!b) return -1;
  if (hipSetDevice(b->device) != hipSuccess) return -1;
  hipStreamSynchronize(b->stream);
  u32 v = 0;
  if (hipMemcpy(&v, b->P.q_stat, 4, hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return (int64_t)v;
}
ORL_ABI_CATCH_INT
