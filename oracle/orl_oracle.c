/*
 * orl_oracle.c — CPU restatement of the reference's per-env step() path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library.  The shipped path is the HIP
 * library (optical_rl_gym_amd/csrc); it never calls into this file.
 *
 * What it restates (file:line relative to /root/reference):
 *   optical_rl_gym/envs/optical_network_env.py   base env: clock, counters, heap, node-pair draw
 *   optical_rl_gym/envs/rmsa_env.py              RMSAEnv + heuristics
 *   optical_rl_gym/envs/deeprmsa_env.py          DeepRMSAEnv (action decode, observation, +-1 reward)
 *   optical_rl_gym/envs/rwa_env.py               RWAEnv + heuristics
 *   optical_rl_gym/envs/rmcsa_env.py             RMCSAEnv + heuristic
 * and the third-party arithmetic those call (not under /root/reference):
 *   CPython 3.10 Modules/_randommodule.c  (MT19937 genrand_uint32, random(), getrandbits)
 *   CPython 3.10 Lib/random.py            (expovariate, choices, _randbelow_with_getrandbits, randrange)
 *   CPython 3.10 Lib/heapq.py             (heappush/_siftdown, heappop/_siftup)
 *   numpy pairwise summation (np.sum / np.mean on float64), glibc libm log()/pow()
 *
 * Representation is deliberately the reference's own (dense 0/1 slot arrays, run-length
 * encoding, a binary heap of (release_time, service)), i.e. NOT the bit-packed formulation
 * of the HIP kernels, so agreement between the two is meaningful.
 *
 * Parity is pinned by tests/test_oracle_golden.py against the .npz traces under tests/golden, which were
 * produced by importing the reference itself in the build container (oracle/gen_golden.py).
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off; no fast-math).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ENV_RMSA 0
#define ENV_DEEPRMSA 1
#define ENV_RWA 2
#define ENV_RMCSA 3
#define ENV_QOS 4 /* QoSConstrainedRA (qos_constrained_ra.py): per-link spectrum counters, service classes */

#define POLICY_SP_FF 0
#define POLICY_SAP_FF 1
#define POLICY_LLP_FF 2
#define POLICY_SAP_LF 3
#define POLICY_PATH_FF 4

typedef struct {
  int32_t env_type, n_nodes, n_links, k_paths, max_hops, n_mods;
  int32_t num_slots, num_cores, episode_length, allow_rejection, j;
  int32_t bit_rate_mode; /* 0 continuous (randint lo..hi), 1 discrete (choices) */
  int32_t br_lo, br_hi, n_bit_rates;
  int32_t n_classes; /* QoSConstrainedRA: num_service_classes */
  double mean_iat, mean_ht; /* mean_service_inter_arrival_time, mean_service_holding_time */
  double channel_width;
  double worst_xt; /* RMCSA: value after the +4 dB of rmcsa_env.py:129 */
} orc_config;

/* cells of one action histogram: actions_output / actions_taken (rmsa_env.py:126-137, rwa_env.py:52-58, rmcsa_env.py:145-180) */
static size_t hist_cells(const orc_config* c) {
  size_t n = (size_t)(c->k_paths + 1) * (size_t)(c->num_slots + 1);
  if (c->env_type == 3 /* RMCSA */) n *= (size_t)(c->n_mods + 1) * (size_t)(c->num_cores + 1);
  return n;
}
static size_t hist4(const orc_config* c, int path, int mod, int core, int slot) { /* numpy C order of [k+1][M+1][C+1][S+1] */
  return (((size_t)path * (c->n_mods + 1) + mod) * (c->num_cores + 1) + core) * (size_t)(c->num_slots + 1) + slot;
}

typedef struct {
  const int32_t* n_paths;       /* [N*N] */
  const int32_t* path_hops;     /* [N*N*k] */
  const int32_t* path_links;    /* [N*N*k*H] */
  const double* path_length;    /* [N*N*k] */
  const int32_t* path_best_mod; /* [N*N*k] */
  const int32_t* mod_se;        /* [M] */
  const double* mod_max_length; /* [M] */
  const double* mod_min_osnr;   /* [M] */
  const double* mod_inband_xt;  /* [M] (RMCSA: after the +4 dB of rmcsa_env.py:128) */
  const int32_t* edge_iter_order; /* [E] link index of the i-th edge of topology.edges() */
  const double* node_probs;     /* [N] node_request_probabilities */
  const int32_t* bit_rates;     /* [n_bit_rates] */
  const double* bit_rate_probs; /* [n_bit_rates] */
  const double* class_probs;    /* [n_classes] QoSConstrainedRA classes_arrival_probabilities */
  const double* class_reward;   /* [n_classes] classes_reward */
} orc_tables;

typedef struct {
  int32_t id, src, dst, bit_rate;
  int32_t path_k, initial_slot, number_slots, core, hops, accepted;
  double at, ht;
} service;

typedef struct {
  double time;
  int32_t sid; /* index into pool */
} heap_item;

typedef struct {
  uint32_t mt[624];
  int32_t mti;
  /* seed() replaces self.rng, but bit_rate_function stays bound to the Random object of __init__ (functools.partial,
     rmsa_env.py:85-87, 97-99): after a reseed the bit rates keep coming from the construction-time stream */
  uint32_t mt_init[624];
  int32_t mti_init, reseeded;
  double current_time;
  int64_t services_processed, services_accepted, episode_services_processed, episode_services_accepted;
  int64_t bit_rate_requested, bit_rate_provisioned, episode_bit_rate_requested, episode_bit_rate_provisioned;
  int32_t new_service;
  service cur;
  heap_item* heap;
  int32_t heap_n, heap_cap;
  service* pool;
  int32_t pool_cap;
  int32_t* free_list;
  int32_t free_n;
  int32_t* running; /* pool indices of topology.graph["running_services"] */
  int32_t running_n;
  uint8_t* avail; /* [C][E][S] 1 = free */
  int32_t* spectrum; /* QoSConstrainedRA: topology.graph["available_spectrum"], free units per link */
  double *l_util, *l_frag, *l_comp, *l_last; /* per link */
  double g_throughput, g_compactness, g_last_update;
  int64_t *br_req_hist, *br_prov_hist; /* discrete mode, per bit-rate index */
  int64_t *act_path, *act_slot;        /* RWA: marginals of actions_output */
  int64_t act_total;
  int64_t *actions_output, *actions_taken; /* [(k+1)][(S+1)] (rmsa_env.py:126-137; RWA uses the top-left corner, rwa_env.py:52-58);
                                            * RMCSA: [(k+1)][(M+1)][(C+1)][(S+1)] (rmcsa_env.py:145-180) */
  int32_t path_choice;                     /* PathOnlyFirstFitAction: the agent's Discrete(k + reject) action */
  int32_t error;
} orc_env;

typedef struct {
  orc_config cfg;
  /* owned copies of the tables */
  int32_t *n_paths, *path_hops, *path_links, *path_best_mod, *mod_se, *edge_iter_order, *bit_rates;
  double *path_length, *mod_max_length, *mod_min_osnr, *mod_inband_xt, *node_probs, *bit_rate_probs;
  double *class_probs, *class_reward;
  int64_t n_envs;
  orc_env* envs;
  int32_t n_info, obs_dim;
} orc_batch;

/* ------------------------------------------------------------------------------------------
 * CPython _randommodule.c: MT19937
 * ---------------------------------------------------------------------------------------- */
static uint32_t genrand_uint32(orc_env* e) {
  static const uint32_t mag01[2] = {0x0U, 0x9908b0dfU};
  uint32_t y;
  uint32_t* mt = e->mt;
  if (e->mti >= 624) {
    int kk;
    for (kk = 0; kk < 624 - 397; kk++) {
      y = (mt[kk] & 0x80000000U) | (mt[kk + 1] & 0x7fffffffU);
      mt[kk] = mt[kk + 397] ^ (y >> 1) ^ mag01[y & 0x1U];
    }
    for (; kk < 623; kk++) {
      y = (mt[kk] & 0x80000000U) | (mt[kk + 1] & 0x7fffffffU);
      mt[kk] = mt[kk + (397 - 624)] ^ (y >> 1) ^ mag01[y & 0x1U];
    }
    y = (mt[623] & 0x80000000U) | (mt[0] & 0x7fffffffU);
    mt[623] = mt[396] ^ (y >> 1) ^ mag01[y & 0x1U];
    e->mti = 0;
  }
  y = mt[e->mti++];
  y ^= (y >> 11);
  y ^= (y << 7) & 0x9d2c5680U;
  y ^= (y << 15) & 0xefc60000U;
  y ^= (y >> 18);
  return y;
}

/* random_random(): 53-bit float in [0,1) */
static double py_random(orc_env* e) {
  uint32_t a = genrand_uint32(e) >> 5, b = genrand_uint32(e) >> 6;
  return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0);
}

/* Lib/random.py expovariate: -log(1.0 - random()) / lambd */
static double py_expovariate(orc_env* e, double lambd) { return -log(1.0 - py_random(e)) / lambd; }

/* Lib/random.py _randbelow_with_getrandbits (n < 2^32) */
static int64_t py_randbelow(orc_env* e, int64_t n) {
  int k = 0;
  int64_t t = n, r;
  if (n == 0) return 0;
  while (t) { k++; t >>= 1; }
  r = (int64_t)(genrand_uint32(e) >> (32 - k));
  while (r >= n) r = (int64_t)(genrand_uint32(e) >> (32 - k));
  return r;
}

/* Lib/random.py choices(population, weights)[0]: cum = accumulate(weights); bisect_right(cum, random()*total, 0, n-1) */
static int py_choices(orc_env* e, const double* weights, int n) {
  double cum[512];
  double acc = 0.0, x;
  int i, lo = 0, hi = n - 1;
  for (i = 0; i < n; i++) { acc = (i == 0) ? weights[0] : acc + weights[i]; cum[i] = acc; }
  x = py_random(e) * (cum[n - 1] + 0.0);
  while (lo < hi) { /* bisect_right */
    int mid = (lo + hi) / 2;
    if (x < cum[mid]) hi = mid; else lo = mid + 1;
  }
  return lo;
}

/* numpy pairwise summation of a contiguous float64 vector (np.sum / np.mean), n <= 128 blocks recursed */
static double np_pairwise_sum(const double* a, int n) {
  if (n < 8) {
    double res = 0.;
    int i;
    for (i = 0; i < n; i++) res += a[i];
    return res;
  } else if (n <= 128) {
    double r[8], res;
    int i, jj;
    for (jj = 0; jj < 8; jj++) r[jj] = a[jj];
    for (i = 8; i < n - (n % 8); i += 8)
      for (jj = 0; jj < 8; jj++) r[jj] += a[i + jj];
    res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; i++) res += a[i];
    return res;
  } else {
    int n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
  }
}

/* ------------------------------------------------------------------------------------------
 * Lib/heapq.py
 * ---------------------------------------------------------------------------------------- */
static void heap_siftdown(heap_item* h, int startpos, int pos) {
  heap_item newitem = h[pos];
  while (pos > startpos) {
    int parentpos = (pos - 1) >> 1;
    if (newitem.time < h[parentpos].time) { h[pos] = h[parentpos]; pos = parentpos; continue; }
    break;
  }
  h[pos] = newitem;
}
static void heap_siftup(heap_item* h, int n, int pos) {
  int endpos = n, startpos = pos, childpos = 2 * pos + 1;
  heap_item newitem = h[pos];
  while (childpos < endpos) {
    int rightpos = childpos + 1;
    if (rightpos < endpos && !(h[childpos].time < h[rightpos].time)) childpos = rightpos;
    h[pos] = h[childpos];
    pos = childpos;
    childpos = 2 * pos + 1;
  }
  h[pos] = newitem;
  heap_siftdown(h, startpos, pos);
}
static void heap_push(orc_env* e, heap_item it) {
  if (e->heap_n == e->heap_cap) {
    e->heap_cap *= 2;
    e->heap = (heap_item*)realloc(e->heap, sizeof(heap_item) * e->heap_cap);
  }
  e->heap[e->heap_n++] = it;
  heap_siftdown(e->heap, 0, e->heap_n - 1);
}
static heap_item heap_pop(orc_env* e) {
  heap_item last = e->heap[--e->heap_n];
  if (e->heap_n > 0) {
    heap_item ret = e->heap[0];
    e->heap[0] = last;
    heap_siftup(e->heap, e->heap_n, 0);
    return ret;
  }
  return last;
}

/* ------------------------------------------------------------------------------------------
 * helpers
 * ---------------------------------------------------------------------------------------- */
static int pool_alloc(orc_env* e) {
  if (e->free_n == 0) {
    int old = e->pool_cap, i;
    e->pool_cap *= 2;
    e->pool = (service*)realloc(e->pool, sizeof(service) * e->pool_cap);
    e->free_list = (int32_t*)realloc(e->free_list, sizeof(int32_t) * e->pool_cap);
    e->running = (int32_t*)realloc(e->running, sizeof(int32_t) * e->pool_cap);
    for (i = e->pool_cap - 1; i >= old; i--) e->free_list[e->free_n++] = i;
  }
  return e->free_list[--e->free_n];
}

#define PIDX(b, s, d, p) ((((s) * (b)->cfg.n_nodes + (d)) * (b)->cfg.k_paths) + (p))
static const int32_t* path_links(const orc_batch* b, int s, int d, int p) {
  return b->path_links + (size_t)PIDX(b, s, d, p) * b->cfg.max_hops;
}
static uint8_t* row(const orc_batch* b, orc_env* e, int core, int link) {
  return e->avail + ((size_t)core * b->cfg.n_links + link) * b->cfg.num_slots;
}

/* RMSAEnv.rle (rmsa_env.py:651-665): starts p[], values v[], lengths z[]; returns number of runs */
static int rle(const uint8_t* ia, int n, int* p, int* v, int* z) {
  int runs = 0, i, start = 0;
  if (n == 0) return 0;
  for (i = 1; i <= n; i++) {
    if (i == n || ia[i] != ia[i - 1]) {
      p[runs] = start; v[runs] = ia[i - 1]; z[runs] = i - start;
      runs++; start = i;
    }
  }
  return runs;
}

/* get_number_slots (rmsa_env.py:610-621, rmcsa_env.py:753-765) */
static int number_slots(const orc_batch* b, int bit_rate, int mod) {
  return (int)ceil((double)bit_rate / ((double)b->mod_se[mod] * b->cfg.channel_width)) + 1;
}

/* is_path_free (rmsa_env.py:623-636, rmcsa_env.py:767-794) */
static int is_path_free(const orc_batch* b, orc_env* e, int s, int d, int p, int core, int initial_slot, int n) {
  int h, i, hops = b->path_hops[PIDX(b, s, d, p)];
  const int32_t* links = path_links(b, s, d, p);
  if (initial_slot + n > b->cfg.num_slots) return 0;
  for (h = 0; h < hops; h++) {
    const uint8_t* r = row(b, e, core, links[h]);
    for (i = initial_slot; i < initial_slot + n; i++)
      if (r[i] == 0) return 0;
  }
  return 1;
}

/* get_available_slots (rmsa_env.py:638-649): elementwise product of the path's link rows (core 0) */
static void available_slots(const orc_batch* b, orc_env* e, int s, int d, int p, uint8_t* out) {
  int h, i, hops = b->path_hops[PIDX(b, s, d, p)], S = b->cfg.num_slots;
  const int32_t* links = path_links(b, s, d, p);
  for (i = 0; i < S; i++) out[i] = 1;
  for (h = 0; h < hops; h++) {
    const uint8_t* r = row(b, e, 0, links[h]);
    for (i = 0; i < S; i++) out[i] = (uint8_t)(out[i] * r[i]);
  }
}

/* get_available_blocks (rmsa_env.py:667-697): first j free runs with length >= slots */
static int available_blocks(const orc_batch* b, orc_env* e, int p, int jmax, int* starts, int* lens) {
  int S = b->cfg.num_slots, runs, i, found = 0;
  uint8_t av[4096];
  int rp[4096], rv[4096], rz[4096];
  int slots = number_slots(b, e->cur.bit_rate, b->path_best_mod[PIDX(b, e->cur.src, e->cur.dst, p)]);
  available_slots(b, e, e->cur.src, e->cur.dst, p, av);
  runs = rle(av, S, rp, rv, rz);
  for (i = 0; i < runs && found < jmax; i++)
    if (rv[i] == 1 && rz[i] >= slots) { starts[found] = rp[i]; lens[found] = rz[i]; found++; }
  return found;
}

/* _get_network_compactness (rmsa_env.py:699-744, rmcsa_env.py:825-871) */
static double network_compactness(const orc_batch* b, orc_env* e, int core) {
  int S = b->cfg.num_slots, E = b->cfg.n_links, it, i;
  int64_t sum_slots_paths = 0, sum_occupied = 0, sum_unused_spectrum_blocks = 0;
  int rp[4096], rv[4096], rz[4096];
  for (i = 0; i < e->running_n; i++) {
    const service* s = &e->pool[e->running[i]];
    sum_slots_paths += (int64_t)s->number_slots * s->hops;
  }
  for (it = 0; it < E; it++) {
    const uint8_t* r = row(b, e, core, b->edge_iter_order[it]);
    int runs = rle(r, S, rp, rv, rz), first = -1, last = -1, nused = 0;
    for (i = 0; i < runs; i++)
      if (rv[i] == 0) { if (first < 0) first = i; last = i; nused++; }
    if (nused > 1) {
      int lambda_min = rp[first], lambda_max = rp[last] + rz[last];
      int irp[4096], irv[4096], irz[4096], iruns, k2;
      sum_occupied += lambda_max - lambda_min;
      iruns = rle(r + lambda_min, lambda_max - lambda_min, irp, irv, irz);
      for (k2 = 0; k2 < iruns; k2++) sum_unused_spectrum_blocks += irv[k2];
    }
  }
  if (sum_unused_spectrum_blocks > 0)
    return ((double)sum_occupied / (double)sum_slots_paths) * ((double)E / (double)sum_unused_spectrum_blocks);
  return 1.0;
}

/* _update_link_stats (rmsa_env.py:464-543, rmcsa_env.py:591-688; RWA: rwa_env.py:365-383) */
static void update_link_stats(const orc_batch* b, orc_env* e, int core, int link) {
  int S = b->cfg.num_slots, i;
  double last_update = e->l_last[link];
  double time_diff = e->current_time - e->l_last[link];
  if (e->current_time > 0) {
    const uint8_t* r = row(b, e, core, link);
    double last_util = e->l_util[link];
    int64_t free_sum = 0;
    double cur_util;
    for (i = 0; i < S; i++) free_sum += r[i];
    cur_util = (double)(S - free_sum) / (double)S;
    e->l_util[link] = ((last_util * last_update) + (cur_util * time_diff)) / e->current_time;
    if (b->cfg.env_type != ENV_RWA) {
      double last_frag = e->l_frag[link], last_comp = e->l_comp[link];
      double cur_frag = 0.0, cur_comp = 0.0;
      if (free_sum > 0) {
        int rp[4096], rv[4096], rz[4096];
        int runs = rle(r, S, rp, rv, rz), n_unused = 0, n_used = 0, first_used = -1, last_used = -1;
        int first_unused = -1, last_unused = -1, max_empty = 0, mx = 0;
        for (i = 0; i < runs; i++) {
          if (rv[i] == 1) { if (first_unused < 0) first_unused = i; last_unused = i; n_unused++; if (rz[i] > mx) mx = rz[i]; }
          else { if (first_used < 0) first_used = i; last_used = i; n_used++; }
        }
        /* len(unused_blocks) > 1 and unused_blocks != [0, len(values) - 1] */
        if (n_unused > 1 && !(n_unused == 2 && first_unused == 0 && last_unused == runs - 1)) max_empty = mx;
        cur_frag = 1.0 - ((double)max_empty / (double)free_sum);
        if (n_used > 1) {
          int lambda_min = rp[first_used], lambda_max = rp[last_used] + rz[last_used];
          int irp[4096], irv[4096], irz[4096], iruns, k2;
          int64_t unused_spectrum_slots = 0; /* np.sum(1 - internal_values) = number of used runs inside */
          iruns = rle(r + lambda_min, lambda_max - lambda_min, irp, irv, irz);
          for (k2 = 0; k2 < iruns; k2++) unused_spectrum_slots += 1 - irv[k2];
          if (unused_spectrum_slots > 0)
            cur_comp = ((double)(lambda_max - lambda_min) / (double)(S - free_sum)) * (1.0 / (double)unused_spectrum_slots);
          else
            cur_comp = 1.0;
        } else {
          cur_comp = 1.0;
        }
      }
      e->l_frag[link] = ((last_frag * last_update) + (cur_frag * time_diff)) / e->current_time;
      e->l_comp[link] = ((last_comp * last_update) + (cur_comp * time_diff)) / e->current_time;
    }
  }
  e->l_last[link] = e->current_time;
}

/* _update_network_stats (rmsa_env.py:439-462, rmcsa_env.py:560-589); RWA's is a no-op (rwa_env.py:351-356) */
static void update_network_stats(const orc_batch* b, orc_env* e, int core) {
  double last_update = e->g_last_update;
  double time_diff = e->current_time - last_update;
  if (e->current_time > 0) {
    double cur_throughput = 0.0;
    int i;
    for (i = 0; i < e->running_n; i++) cur_throughput += (double)e->pool[e->running[i]].bit_rate;
    e->g_throughput = ((e->g_throughput * last_update) + (cur_throughput * time_diff)) / e->current_time;
    e->g_compactness = ((e->g_compactness * last_update) + (network_compactness(b, e, core) * time_diff)) / e->current_time;
  }
  e->g_last_update = e->current_time;
}

/* _provision_path (rmsa_env.py:364-415, rmcsa_env.py:488-534, rwa_env.py:293-321) */
static void provision_path(const orc_batch* b, orc_env* e, int p, int core, int initial_slot, int n) {
  int h, i, hops = b->path_hops[PIDX(b, e->cur.src, e->cur.dst, p)], sid;
  const int32_t* links = path_links(b, e->cur.src, e->cur.dst, p);
  for (h = 0; h < hops; h++) {
    uint8_t* r = row(b, e, core, links[h]);
    for (i = initial_slot; i < initial_slot + n; i++) r[i] = 0;
    update_link_stats(b, e, core, links[h]);
  }
  e->cur.path_k = p; e->cur.initial_slot = initial_slot; e->cur.number_slots = n; e->cur.core = core; e->cur.hops = hops;
  sid = pool_alloc(e);
  e->pool[sid] = e->cur;
  e->running[e->running_n++] = sid;
  if (b->cfg.env_type != ENV_RWA) {
    update_network_stats(b, e, core);
    e->services_accepted += 1;
    e->episode_services_accepted += 1;
    e->bit_rate_provisioned += e->cur.bit_rate;
    e->episode_bit_rate_provisioned += e->cur.bit_rate;
  }
  /* _add_release (optical_network_env.py:143-154) is called by step() right after; the pool index rides along */
  e->cur.accepted = sid + 1; /* temporarily carries sid+1; normalised to 1 by the caller */
}

/* _release_path (rmsa_env.py:417-437, rmcsa_env.py:536-558, rwa_env.py:323-349) */
static void qos_update_link_stats(const orc_batch* b, orc_env* e, int link);
static void release_path(const orc_batch* b, orc_env* e, int sid) {
  service* s = &e->pool[sid];
  int h, i;
  const int32_t* links = path_links(b, s->src, s->dst, s->path_k);
  if (b->cfg.env_type == ENV_QOS) { /* qos_constrained_ra.py:313-338 */
    for (h = 0; h < s->hops; h++) {
      e->spectrum[links[h]] += s->number_slots;
      qos_update_link_stats(b, e, links[h]);
    }
    e->free_list[e->free_n++] = sid;
    return;
  }
  for (h = 0; h < s->hops; h++) {
    uint8_t* r = row(b, e, s->core, links[h]);
    for (i = s->initial_slot; i < s->initial_slot + s->number_slots; i++) r[i] = 1;
    update_link_stats(b, e, s->core, links[h]);
  }
  for (i = 0; i < e->running_n; i++)
    if (e->running[i] == sid) { memmove(e->running + i, e->running + i + 1, sizeof(int32_t) * (e->running_n - i - 1)); e->running_n--; break; }
  e->free_list[e->free_n++] = sid;
}

/* _get_node_pair (optical_network_env.py:156-173) */
static void get_node_pair(const orc_batch* b, orc_env* e, int* src, int* dst) {
  int N = b->cfg.n_nodes, i;
  double w[512], tot;
  *src = py_choices(e, b->node_probs, N);
  for (i = 0; i < N; i++) w[i] = b->node_probs[i];
  w[*src] = 0.0;
  tot = np_pairwise_sum(w, N);
  for (i = 0; i < N; i++) w[i] = w[i] / tot;
  *dst = py_choices(e, w, N);
}

static int bit_rate_index(const orc_batch* b, int bit_rate) {
  int i;
  for (i = 0; i < b->cfg.n_bit_rates; i++) if (b->bit_rates[i] == bit_rate) return i;
  return -1;
}

static void release_due(const orc_batch* b, orc_env* e) {
  while (e->heap_n > 0) {
    heap_item it = heap_pop(e);
    if (it.time <= e->current_time) release_path(b, e, it.sid);
    else { heap_push(e, it); break; }
  }
}

/* _next_service (rmsa_env.py:545-597, rwa_env.py:258-288, rmcsa_env.py:690-739) */
static void next_service(const orc_batch* b, orc_env* e) {
  double at, ht;
  int src, dst, bit_rate = 0, t = b->cfg.env_type;
  if (e->new_service) return;
  at = e->current_time + py_expovariate(e, 1 / b->cfg.mean_iat);
  e->current_time = at;
  ht = py_expovariate(e, 1 / b->cfg.mean_ht);
  get_node_pair(b, e, &src, &dst);
  if (t == ENV_QOS) bit_rate = py_choices(e, b->class_probs, b->cfg.n_classes); /* the service class rides in `bit_rate` */
  if (t != ENV_RWA && t != ENV_QOS) {
    uint32_t keep[624];
    int32_t keep_i = 0;
    if (e->reseeded) { /* draw from the construction-time stream */
      memcpy(keep, e->mt, sizeof keep); keep_i = e->mti;
      memcpy(e->mt, e->mt_init, sizeof keep); e->mti = e->mti_init;
    }
    if (b->cfg.bit_rate_mode == 0) bit_rate = b->cfg.br_lo + (int)py_randbelow(e, b->cfg.br_hi + 1 - b->cfg.br_lo);
    else bit_rate = b->bit_rates[py_choices(e, b->bit_rate_probs, b->cfg.n_bit_rates)];
    if (e->reseeded) {
      memcpy(e->mt_init, e->mt, sizeof keep); e->mti_init = e->mti;
      memcpy(e->mt, keep, sizeof keep); e->mti = keep_i;
    }
  }
  if (t == ENV_RWA || t == ENV_RMCSA || t == ENV_QOS) release_due(b, e); /* these release BEFORE creating the service */
  memset(&e->cur, 0, sizeof(e->cur));
  e->cur.id = (int32_t)e->episode_services_processed;
  e->cur.src = src; e->cur.dst = dst; e->cur.at = at; e->cur.ht = ht; e->cur.bit_rate = bit_rate;
  e->cur.number_slots = (t == ENV_RWA || t == ENV_QOS) ? 1 : 0;
  e->new_service = 1;
  if (t == ENV_RMSA || t == ENV_DEEPRMSA) {
    e->services_processed += 1;
    e->episode_services_processed += 1;
  }
  if (t != ENV_RWA && t != ENV_QOS) {
    e->bit_rate_requested += bit_rate;
    e->episode_bit_rate_requested += bit_rate;
    if (b->cfg.bit_rate_mode == 1) e->br_req_hist[bit_rate_index(b, bit_rate)] += 1;
  }
  if (t == ENV_RMSA || t == ENV_DEEPRMSA) release_due(b, e);
}

static double blocking(int64_t req, int64_t prov);
/* ------------------------------------------------------------------------------------------
 * QoSConstrainedRA (qos_constrained_ra.py).  Upstream its constructor raises (an unexpected k_paths keyword for the base
 * class, :32-41, and a service_class field utils.Service does not have, :281-291); the fixtures come from the reference
 * with exactly those two things patched at import time (oracle/gen_golden_qos.py).
 * ---------------------------------------------------------------------------------------- */
static void qos_update_link_stats(const orc_batch* b, orc_env* e, int link) { /* :355-372 */
  double last_update = e->l_last[link];
  double time_diff = e->current_time - e->l_last[link];
  if (e->current_time > 0) {
    double last_util = e->l_util[link];
    double cur_util = (double)(b->cfg.num_slots - e->spectrum[link]) / (double)b->cfg.num_slots;
    e->l_util[link] = ((last_util * last_update) + (cur_util * time_diff)) / e->current_time;
  }
  e->l_last[link] = e->current_time;
}
static int qos_is_path_free(const orc_batch* b, orc_env* e, int p, int number_slots) { /* :381-392 */
  int h, hops = b->path_hops[PIDX(b, e->cur.src, e->cur.dst, p)];
  const int32_t* links = path_links(b, e->cur.src, e->cur.dst, p);
  if (number_slots > b->cfg.num_slots) return 0;
  for (h = 0; h < hops; h++)
    if (e->spectrum[links[h]] < number_slots) return 0;
  return 1;
}
static int qos_step(const orc_batch* b, orc_env* e, const int32_t* action, double* reward, uint8_t* done, double* info) { /* :100-157 */
  const orc_config* c = &b->cfg;
  int k = c->k_paths, rej = c->allow_rejection ? 1 : 0, a = action[0], clazz = e->cur.bit_rate;
  int np_ = b->n_paths[e->cur.src * c->n_nodes + e->cur.dst];
  if (a < 0 || a >= k + rej) return -2; /* actions_output[action] += 1 */
  e->cur.accepted = 0;
  if ((clazz == 0 && a == 0) || (clazz != 0 && a < np_)) {
    if (qos_is_path_free(b, e, a, e->cur.number_slots)) {
      int h, hops = b->path_hops[PIDX(b, e->cur.src, e->cur.dst, a)], sid;
      const int32_t* links = path_links(b, e->cur.src, e->cur.dst, a);
      heap_item it;
      for (h = 0; h < hops; h++) { /* _provision_path :296-311 */
        e->spectrum[links[h]] -= e->cur.number_slots;
        qos_update_link_stats(b, e, links[h]);
      }
      e->cur.path_k = a; e->cur.hops = hops;
      sid = pool_alloc(e);
      e->pool[sid] = e->cur;
      e->cur.accepted = 1;
      e->services_accepted += 1;
      e->episode_services_accepted += 1;
      it.sid = sid; it.time = e->cur.at + e->cur.ht;
      heap_push(e, it);
    }
  }
  e->services_processed += 1;
  e->episode_services_processed += 1;
  *reward = e->cur.accepted ? b->class_reward[clazz] : 0.0;
  info[0] = blocking(e->services_processed, e->services_accepted);
  info[1] = blocking(e->episode_services_processed, e->episode_services_accepted);
  e->new_service = 0;
  next_service(b, e);
  *done = (uint8_t)(e->episode_services_processed == c->episode_length);
  return 0;
}
/* the module-level heuristics :408-450 (policy ids: SP_FF = shortest_path, SAP_FF = shortest_available_path,
   LLP_FF = least_loaded_path) */
static void qos_policy(const orc_batch* b, orc_env* e, int policy, int32_t* action) {
  const orc_config* c = &b->cfg;
  int k = c->k_paths, np_ = b->n_paths[e->cur.src * c->n_nodes + e->cur.dst], idp, h;
  action[0] = action[1] = action[2] = action[3] = 0;
  if (policy == POLICY_SP_FF) {
    action[0] = qos_is_path_free(b, e, 0, e->cur.number_slots) ? 0 : k;
  } else if (policy == POLICY_SAP_FF) {
    int best_hops = 1 << 30;
    if (e->cur.bit_rate == 0) return; /* high-priority services only accept the shortest path */
    action[0] = k;
    for (idp = 0; idp < np_; idp++) {
      int hops = b->path_hops[PIDX(b, e->cur.src, e->cur.dst, idp)];
      if (hops < best_hops && qos_is_path_free(b, e, idp, e->cur.number_slots)) { best_hops = hops; action[0] = idp; }
    }
  } else {
    double best_load = -1.7976931348623157e308; /* np.finfo(0.0).min */
    if (e->cur.bit_rate == 0) return;
    action[0] = k;
    for (idp = 0; idp < np_; idp++) {
      const int32_t* links = path_links(b, e->cur.src, e->cur.dst, idp);
      int hops = b->path_hops[PIDX(b, e->cur.src, e->cur.dst, idp)];
      double cap = 1.7976931348623157e308; /* np.finfo(0.0).max */
      for (h = 0; h < hops; h++) if ((double)e->spectrum[links[h]] < cap) cap = (double)e->spectrum[links[h]];
      if (cap > best_load) { best_load = cap; action[0] = idp; }
    }
  }
}

/* reset (rmsa_env.py:284-359, rwa_env.py:164-208, rmcsa_env.py:386-483, optical_network_env.py:181-203) */
static void env_reset(const orc_batch* b, orc_env* e, int full) {
  int t = b->cfg.env_type, E = b->cfg.n_links, i;
  e->episode_bit_rate_requested = 0;
  e->episode_bit_rate_provisioned = 0;
  e->episode_services_processed = 0;
  e->episode_services_accepted = 0;
  if (!full) {
    if (t != ENV_RWA && t != ENV_QOS && e->new_service) {
      e->episode_services_processed += 1;
      e->episode_bit_rate_requested += e->cur.bit_rate;
    }
    return;
  }
  e->heap_n = 0;
  e->current_time = 0;
  e->services_processed = 0;
  e->services_accepted = 0;
  e->running_n = 0;
  e->free_n = 0;
  for (i = e->pool_cap - 1; i >= 0; i--) e->free_list[e->free_n++] = i;
  e->g_last_update = 0.0; e->g_compactness = 0.0; e->g_throughput = 0.0;
  for (i = 0; i < E; i++) { e->l_util[i] = 0.0; e->l_last[i] = 0.0; e->l_frag[i] = 0.0; e->l_comp[i] = 0.0; }
  e->bit_rate_requested = 0;
  e->bit_rate_provisioned = 0;
  if (t == ENV_QOS) for (i = 0; i < E; i++) e->spectrum[i] = b->cfg.num_slots; /* optical_network_env.py:189-193 */
  else memset(e->avail, 1, (size_t)b->cfg.num_cores * E * b->cfg.num_slots);
  if (b->cfg.bit_rate_mode == 1)
    for (i = 0; i < b->cfg.n_bit_rates; i++) { e->br_req_hist[i] = 0; e->br_prov_hist[i] = 0; }
  if (t == ENV_RWA) {
    int rej = b->cfg.allow_rejection ? 1 : 0;
    for (i = 0; i < b->cfg.k_paths + rej; i++) e->act_path[i] = 0;
    for (i = 0; i < b->cfg.num_slots + rej; i++) e->act_slot[i] = 0;
    e->act_total = 0;
    /* rwa_env.py:194-203; RMSAEnv.reset never clears actions_output / actions_taken (rmsa_env.py:284-359) */
    memset(e->actions_output, 0, sizeof(int64_t) * hist_cells(&b->cfg));
    memset(e->actions_taken, 0, sizeof(int64_t) * hist_cells(&b->cfg));
  }
  if (t == ENV_RMCSA) { /* rmcsa_env.py:437-454 */
    memset(e->actions_output, 0, sizeof(int64_t) * hist_cells(&b->cfg));
    memset(e->actions_taken, 0, sizeof(int64_t) * hist_cells(&b->cfg));
  }
  e->new_service = 0;
  next_service(b, e);
}

/* _crosstalk_is_acceptable (rmcsa_env.py:341-384) */
static int crosstalk_is_acceptable(const orc_batch* b, orc_env* e, int mod, double path_length) {
  double average_power = 1;
  double nf_db = 5.5;
  double nf = pow(10.0, nf_db / 10.0);
  double amp_spam = 100;
  double amp_gain_db = 20;
  double amp_gain = pow(10.0, amp_gain_db / 10.0);
  double lambda_ = 1550;
  double h = 6.626068e-34;
  double f_hz = 2.99e8 / (lambda_ * 1e-9);
  double snr_min_calc = pow(10.0, (b->mod_min_osnr[mod] + 2) / 10);
  double lmax_snr = (average_power * amp_spam) /
                    (snr_min_calc * h * f_hz * amp_gain * nf * ((double)e->cur.bit_rate / (double)b->mod_se[mod]) * 1e9);
  double lmax_xt;
  lmax_snr = lmax_snr / 1000;
  lmax_xt = pow(10.0, (b->mod_inband_xt[mod] - b->cfg.worst_xt - 4) / 10);
  return (path_length < lmax_xt && path_length < lmax_snr) ? 1 : 0;
}

static double blocking(int64_t req, int64_t prov) { return (double)(req - prov) / (double)req; }

/* step(): action[4]; RMSA/RWA use [0..1], DeepRMSA [0], RMCSA [0..3].  Returns 0 or a negative error. */
static int env_step(const orc_batch* b, orc_env* e, const int32_t* action, double* reward, uint8_t* done, double* info) {
  if (b->cfg.env_type == ENV_QOS) return qos_step(b, e, action, reward, done, info);
  const orc_config* c = &b->cfg;
  int t = c->env_type, k = c->k_paths, S = c->num_slots;
  int path, slot, mod = 0, core = 0, rej = c->allow_rejection ? 1 : 0, i;
  double prev_compactness = 0.0, cur_compactness = 0.0;
  if (t == ENV_DEEPRMSA) { /* deeprmsa_env.py:48-58 */
    int a = action[0];
    path = k; slot = S;
    if (a < k * c->j) {
      int route = a / c->j, block = a % c->j, starts[64], lens[64];
      int nb = available_blocks(b, e, route, c->j, starts, lens);
      if (block < nb) { path = route; slot = starts[block]; }
    }
  } else if (t == ENV_RMCSA) {
    path = action[0]; mod = action[1]; core = action[2]; slot = action[3];
  } else {
    path = action[0]; slot = action[1];
  }
  /* actions_output[...] += 1 raises IndexError when out of the histogram's shape */
  if (t == ENV_RWA) {
    if (path < 0 || path >= k + rej || slot < 0 || slot >= S + rej) return -2;
    e->act_path[path] += 1; e->act_slot[slot] += 1; e->act_total += 1;
  } else if (t == ENV_RMCSA) {
    if (path < 0 || path > k || mod < 0 || mod > c->n_mods || core < 0 || core > c->num_cores || slot < 0 || slot > S) return -2;
  } else {
    if (path < 0 || path > k || slot < 0 || slot > S) return -2;
  }
  if (t != ENV_RMCSA) e->actions_output[(size_t)path * (S + 1) + slot] += 1; /* rmsa_env.py:167, rwa_env.py:103 */
  else e->actions_output[hist4(c, path, mod, core, slot)] += 1;                 /* rmcsa_env.py:219 */
  if (t == ENV_RMSA || t == ENV_DEEPRMSA) prev_compactness = network_compactness(b, e, 0);
  e->cur.accepted = 0;
  if (t == ENV_RMCSA) {
    if (path < k && mod < c->n_mods && core < c->num_cores && slot < S) {
      int n = number_slots(b, e->cur.bit_rate, mod);
      if (path >= b->n_paths[e->cur.src * c->n_nodes + e->cur.dst]) return -3;
      if (is_path_free(b, e, e->cur.src, e->cur.dst, path, core, slot, n)) {
        double path_length = b->path_length[PIDX(b, e->cur.src, e->cur.dst, path)];
        if (crosstalk_is_acceptable(b, e, mod, path_length)) {
          heap_item it;
          provision_path(b, e, path, core, slot, n);
          it.sid = e->cur.accepted - 1; it.time = e->cur.at + e->cur.ht;
          e->cur.accepted = 1;
          heap_push(e, it);
        }
      }
    }
    e->services_processed += 1;
    e->episode_services_processed += 1;
    e->bit_rate_requested += e->cur.bit_rate;
    e->episode_bit_rate_requested += e->cur.bit_rate;
  } else if (path < k && slot < S) {
    int n = 1;
    if (path >= b->n_paths[e->cur.src * c->n_nodes + e->cur.dst]) return -3;
    if (t != ENV_RWA) n = number_slots(b, e->cur.bit_rate, b->path_best_mod[PIDX(b, e->cur.src, e->cur.dst, path)]);
    if (is_path_free(b, e, e->cur.src, e->cur.dst, path, 0, slot, n)) {
      heap_item it;
      provision_path(b, e, path, 0, slot, n);
      it.sid = e->cur.accepted - 1; it.time = e->cur.at + e->cur.ht;
      e->cur.accepted = 1;
      if (t == ENV_RWA) { e->services_accepted += 1; e->episode_services_accepted += 1; }
      else if (c->bit_rate_mode == 1) e->br_prov_hist[bit_rate_index(b, e->cur.bit_rate)] += 1;
      heap_push(e, it);
    }
  }
  if (t == ENV_RWA) { e->services_processed += 1; e->episode_services_processed += 1; }
  if (t != ENV_RMCSA) { /* actions_taken: rmsa_env.py:201, 211-212; rwa_env.py:125, 132-133 */
    if (e->cur.accepted) e->actions_taken[(size_t)path * (S + 1) + slot] += 1;
    else e->actions_taken[(size_t)k * (S + 1) + S] += 1;
  } else { /* rmcsa_env.py:273, 284-289 */
    if (e->cur.accepted) e->actions_taken[hist4(c, path, mod, core, slot)] += 1;
    else e->actions_taken[hist4(c, k, c->n_mods, c->num_cores, S)] += 1;
  }
  if (t == ENV_RMSA || t == ENV_DEEPRMSA) cur_compactness = network_compactness(b, e, 0);

  *reward = e->cur.accepted ? 1.0 : ((t == ENV_DEEPRMSA) ? -1.0 : 0.0);
  info[0] = blocking(e->services_processed, e->services_accepted);
  info[1] = blocking(e->episode_services_processed, e->episode_services_accepted);
  if (t == ENV_RWA) {
    int np_ = k + rej, ns = S + rej;
    for (i = 0; i < np_; i++) info[2 + i] = (double)e->act_path[i] / (double)e->act_total;
    for (i = 0; i < ns; i++) info[2 + np_ + i] = (double)e->act_slot[i] / (double)e->act_total;
  } else {
    info[2] = blocking(e->bit_rate_requested, e->bit_rate_provisioned);
    info[3] = blocking(e->episode_bit_rate_requested, e->episode_bit_rate_provisioned);
  }
  if (t == ENV_RMSA || t == ENV_DEEPRMSA) {
    double v[512];
    int E = c->n_links;
    info[4] = cur_compactness;
    info[5] = prev_compactness - cur_compactness;
    for (i = 0; i < E; i++) v[i] = e->l_comp[b->edge_iter_order[i]];
    info[6] = np_pairwise_sum(v, E) / (double)E;
    for (i = 0; i < E; i++) v[i] = e->l_util[b->edge_iter_order[i]];
    info[7] = np_pairwise_sum(v, E) / (double)E;
    if (c->bit_rate_mode == 1) { /* rmsa_env.py:217-227, 268-273 */
      double mx = -INFINITY, mn = INFINITY;
      for (i = 0; i < c->n_bit_rates; i++) {
        double bl = 0.0;
        if (e->br_req_hist[i] > 0) bl = (double)(e->br_req_hist[i] - e->br_prov_hist[i]) / (double)e->br_req_hist[i];
        info[8 + i] = bl;
        if (bl > mx) mx = bl;
        if (bl < mn) mn = bl;
      }
      info[8 + c->n_bit_rates] = mx - mn;
    }
  }
  e->new_service = 0;
  next_service(b, e);
  *done = (uint8_t)(e->episode_services_processed == c->episode_length);
  return 0;
}

/* get_best_modulation_format (utils.py:84-96): stable sort by spectral efficiency, descending */
static int best_modulation(const orc_batch* b, double length) {
  int M = b->cfg.n_mods, order[64], i, jx;
  for (i = 0; i < M; i++) order[i] = i;
  for (i = 1; i < M; i++) { /* stable insertion sort, descending SE */
    int v = order[i];
    jx = i - 1;
    while (jx >= 0 && b->mod_se[order[jx]] < b->mod_se[v]) { order[jx + 1] = order[jx]; jx--; }
    order[jx + 1] = v;
  }
  for (i = 0; i < M; i++) if (length <= b->mod_max_length[order[i]]) return order[i];
  return -1;
}

/* heuristics; writes action[4] */
static void env_policy(const orc_batch* b, orc_env* e, int policy, int32_t* action) {
  const orc_config* c = &b->cfg;
  int t = c->env_type, k = c->k_paths, S = c->num_slots, src = e->cur.src, dst = e->cur.dst;
  int np_ = b->n_paths[src * c->n_nodes + dst], idp, s0;
  action[0] = action[1] = action[2] = action[3] = 0;
  if (t == ENV_QOS) { qos_policy(b, e, policy, action); return; }
  if (policy == POLICY_PATH_FF) { /* PathOnlyFirstFitAction.action: rmsa_env.py:848-871, rwa_env.py:513-533 */
    int a = e->path_choice;
    action[0] = k; action[1] = S;
    if (a >= 0 && a < k && a < np_) {
      if (t == ENV_RWA) {
        for (s0 = 0; s0 < S; s0++)
          if (is_path_free(b, e, src, dst, a, 0, s0, 1)) { action[0] = a; action[1] = s0; return; }
      } else {
        int n = number_slots(b, e->cur.bit_rate, b->path_best_mod[PIDX(b, src, dst, a)]);
        for (s0 = 0; s0 < S - n; s0++)
          if (is_path_free(b, e, src, dst, a, 0, s0, n)) { action[0] = a; action[1] = s0; return; }
      }
    }
    return;
  }
  if (t == ENV_RMSA) {
    action[0] = k; action[1] = S;
    if (policy == POLICY_SP_FF) { /* rmsa_env.py:747-764 */
      int n = number_slots(b, e->cur.bit_rate, b->path_best_mod[PIDX(b, src, dst, 0)]);
      for (s0 = 0; s0 < S - n; s0++)
        if (is_path_free(b, e, src, dst, 0, 0, s0, n)) { action[0] = 0; action[1] = s0; return; }
    } else if (policy == POLICY_SAP_FF) { /* rmsa_env.py:767-779 */
      for (idp = 0; idp < np_; idp++) {
        int n = number_slots(b, e->cur.bit_rate, b->path_best_mod[PIDX(b, src, dst, idp)]);
        for (s0 = 0; s0 < S - n; s0++)
          if (is_path_free(b, e, src, dst, idp, 0, s0, n)) { action[0] = idp; action[1] = s0; return; }
      }
    } else if (policy == POLICY_LLP_FF) { /* rmsa_env.py:782-803 */
      int64_t max_free = 0;
      uint8_t av[4096];
      for (idp = 0; idp < np_; idp++) {
        int n = number_slots(b, e->cur.bit_rate, b->path_best_mod[PIDX(b, src, dst, idp)]);
        for (s0 = 0; s0 < S - n; s0++)
          if (is_path_free(b, e, src, dst, idp, 0, s0, n)) {
            int64_t free_slots = 0;
            int i;
            available_slots(b, e, src, dst, idp, av);
            for (i = 0; i < S; i++) free_slots += av[i];
            if (free_slots > max_free) { action[0] = idp; action[1] = s0; max_free = free_slots; }
            break;
          }
      }
    }
  } else if (t == ENV_DEEPRMSA) {
    int starts[64], lens[64];
    if (policy == POLICY_SP_FF) { /* deeprmsa_env.py:135-143 */
      if (!c->allow_rejection) action[0] = 0;
      else action[0] = (available_blocks(b, e, 0, c->j, starts, lens) > 0) ? 0 : k * c->j;
    } else { /* deeprmsa_env.py:146-155 */
      action[0] = k * c->j;
      for (idp = 0; idp < np_; idp++)
        if (available_blocks(b, e, idp, c->j, starts, lens) > 0) { action[0] = idp * c->j; break; }
    }
  } else if (t == ENV_RWA) {
    action[0] = k; action[1] = S;
    if (policy == POLICY_SP_FF) { /* rwa_env.py:425-435 */
      for (s0 = 0; s0 < S; s0++)
        if (is_path_free(b, e, src, dst, 0, 0, s0, 1)) { action[0] = 0; action[1] = s0; return; }
    } else if (policy == POLICY_SAP_FF || policy == POLICY_SAP_LF) { /* rwa_env.py:438-479 */
      double best_hops = 1.7976931348623157e308;
      for (idp = 0; idp < np_; idp++) {
        int hops = b->path_hops[PIDX(b, src, dst, idp)];
        if ((double)hops < best_hops) {
          if (policy == POLICY_SAP_FF) {
            for (s0 = 0; s0 < S; s0++)
              if (is_path_free(b, e, src, dst, idp, 0, s0, 1)) { best_hops = hops; action[0] = idp; action[1] = s0; break; }
          } else {
            for (s0 = S - 1; s0 > 0; s0--) /* range(S-1, 0, -1): wavelength 0 is never tried */
              if (is_path_free(b, e, src, dst, idp, 0, s0, 1)) { best_hops = hops; action[0] = idp; action[1] = s0; break; }
          }
        }
      }
    } else if (policy == POLICY_LLP_FF) { /* rwa_env.py:403-422, 482-502 */
      double best_load = -1.7976931348623157e308;
      for (idp = 0; idp < np_; idp++) {
        int cap = 0;
        for (s0 = 0; s0 < S; s0++) cap += is_path_free(b, e, src, dst, idp, 0, s0, 1);
        if ((double)cap > best_load)
          for (s0 = 0; s0 < S; s0++)
            if (is_path_free(b, e, src, dst, idp, 0, s0, 1)) { best_load = cap; action[0] = idp; action[1] = s0; break; }
      }
    }
  } else if (t == ENV_RMCSA) { /* rmcsa_env.py:882-911 */
    int core;
    /* reject: the reference returns a 3-tuple here that step() cannot index (IndexError);
       the restatement uses the all-out-of-range 4-tuple, which step() treats as a rejection */
    action[0] = k; action[1] = c->n_mods; action[2] = c->num_cores; action[3] = S;
    for (idp = 0; idp < np_; idp++) {
      int mod = best_modulation(b, b->path_length[PIDX(b, src, dst, idp)]);
      int n = number_slots(b, e->cur.bit_rate, mod);
      for (core = 0; core < c->num_cores; core++)
        for (s0 = 0; s0 < S - n; s0++)
          if (is_path_free(b, e, src, dst, idp, core, s0, n)) {
            action[0] = idp; action[1] = mod; action[2] = core; action[3] = s0;
            return;
          }
    }
  }
}

/* DeepRMSAEnv.observation (deeprmsa_env.py:60-121) */
static void env_observation(const orc_batch* b, orc_env* e, double* obs) {
  const orc_config* c = &b->cfg;
  int N = c->n_nodes, k = c->k_paths, jj = c->j, S = c->num_slots, W = 2 * jj + 3, i, idp;
  int src = e->cur.src, dst = e->cur.dst, np_ = b->n_paths[src * N + dst];
  int mn = src < dst ? src : dst, mx = src < dst ? dst : src;
  double* tau = obs + 1;
  double* sp = obs + 1 + 2 * N;
  for (i = 0; i < 2 * N; i++) tau[i] = 0.0;
  tau[mn] = 1; tau[N + mx] = 1;
  for (i = 0; i < k * W; i++) sp[i] = -1.0;
  for (idp = 0; idp < np_; idp++) {
    uint8_t av[4096];
    int rp[4096], rv[4096], rz[4096], starts[64], lens[64];
    int num_slots = number_slots(b, e->cur.bit_rate, b->path_best_mod[PIDX(b, src, dst, idp)]);
    int nb = available_blocks(b, e, idp, jj, starts, lens), runs, idb, nfree_runs = 0;
    int64_t tot = 0, len_sum = 0;
    available_slots(b, e, src, dst, idp, av);
    for (idb = 0; idb < nb; idb++) {
      sp[idp * W + idb * 2 + 0] = 2 * (starts[idb] - 0.5 * S) / S;
      sp[idp * W + idb * 2 + 1] = (double)(lens[idb] - 8) / 8;
    }
    sp[idp * W + jj * 2] = (num_slots - 5.5) / 3.5;
    runs = rle(av, S, rp, rv, rz);
    for (i = 0; i < S; i++) tot += av[i];
    sp[idp * W + jj * 2 + 1] = 2 * (tot - 0.5 * S) / S;
    for (i = 0; i < runs; i++) if (rv[i] == 1) { nfree_runs++; len_sum += rz[i]; }
    if (nfree_runs > 0) sp[idp * W + jj * 2 + 2] = ((double)len_sum / (double)nfree_runs - 4) / 4;
  }
  obs[0] = (double)e->cur.bit_rate / 100;
}

/* ------------------------------------------------------------------------------------------
 * exported C interface (ctypes)
 * ---------------------------------------------------------------------------------------- */
static void* dup_mem(const void* p, size_t n) {
  void* q;
  if (!p || !n) return NULL;
  q = malloc(n);
  memcpy(q, p, n);
  return q;
}

int orc_info_dim(const orc_config* c) {
  int rej = c->allow_rejection ? 1 : 0;
  if (c->env_type == ENV_QOS) return 2;
  if (c->env_type == ENV_RWA) return 2 + (c->k_paths + rej) + (c->num_slots + rej);
  if (c->env_type == ENV_RMCSA) return 4;
  return 8 + (c->bit_rate_mode == 1 ? c->n_bit_rates + 1 : 0);
}
int orc_obs_dim(const orc_config* c) {
  if (c->env_type == ENV_DEEPRMSA) return 1 + 2 * c->n_nodes + (2 * c->j + 3) * c->k_paths;
  return 0;
}

orc_batch* orc_create(const orc_config* cfg, const orc_tables* tb, int64_t n_envs, const uint32_t* mt_state /*[n][625]*/) {
  orc_batch* b = (orc_batch*)calloc(1, sizeof(orc_batch));
  size_t nn = (size_t)cfg->n_nodes * cfg->n_nodes, npk = nn * cfg->k_paths;
  int64_t i;
  b->cfg = *cfg;
  b->n_paths = (int32_t*)dup_mem(tb->n_paths, nn * 4);
  b->path_hops = (int32_t*)dup_mem(tb->path_hops, npk * 4);
  b->path_links = (int32_t*)dup_mem(tb->path_links, npk * cfg->max_hops * 4);
  b->path_length = (double*)dup_mem(tb->path_length, npk * 8);
  b->path_best_mod = (int32_t*)dup_mem(tb->path_best_mod, npk * 4);
  b->mod_se = (int32_t*)dup_mem(tb->mod_se, (size_t)cfg->n_mods * 4);
  b->mod_max_length = (double*)dup_mem(tb->mod_max_length, (size_t)cfg->n_mods * 8);
  b->mod_min_osnr = (double*)dup_mem(tb->mod_min_osnr, (size_t)cfg->n_mods * 8);
  b->mod_inband_xt = (double*)dup_mem(tb->mod_inband_xt, (size_t)cfg->n_mods * 8);
  b->edge_iter_order = (int32_t*)dup_mem(tb->edge_iter_order, (size_t)cfg->n_links * 4);
  b->node_probs = (double*)dup_mem(tb->node_probs, (size_t)cfg->n_nodes * 8);
  b->bit_rates = (int32_t*)dup_mem(tb->bit_rates, (size_t)cfg->n_bit_rates * 4);
  b->bit_rate_probs = (double*)dup_mem(tb->bit_rate_probs, (size_t)cfg->n_bit_rates * 8);
  if (cfg->env_type == ENV_QOS) {
    b->class_probs = (double*)dup_mem(tb->class_probs, (size_t)cfg->n_classes * 8);
    b->class_reward = (double*)dup_mem(tb->class_reward, (size_t)cfg->n_classes * 8);
  }
  b->n_envs = n_envs;
  b->n_info = orc_info_dim(cfg);
  b->obs_dim = orc_obs_dim(cfg);
  b->envs = (orc_env*)calloc((size_t)n_envs, sizeof(orc_env));
  for (i = 0; i < n_envs; i++) {
    orc_env* e = &b->envs[i];
    int E = cfg->n_links, rej = cfg->allow_rejection ? 1 : 0;
    memcpy(e->mt, mt_state + i * 625, 624 * 4);
    e->mti = (int32_t)mt_state[i * 625 + 624];
    e->heap_cap = 64; e->heap = (heap_item*)malloc(sizeof(heap_item) * e->heap_cap);
    e->pool_cap = 64; e->pool = (service*)malloc(sizeof(service) * e->pool_cap);
    e->free_list = (int32_t*)malloc(sizeof(int32_t) * e->pool_cap);
    e->running = (int32_t*)malloc(sizeof(int32_t) * e->pool_cap);
    e->avail = (uint8_t*)malloc((size_t)cfg->num_cores * E * cfg->num_slots);
    e->spectrum = (int32_t*)calloc(E, 4);
    e->l_util = (double*)calloc(E, 8); e->l_frag = (double*)calloc(E, 8);
    e->l_comp = (double*)calloc(E, 8); e->l_last = (double*)calloc(E, 8);
    if (cfg->bit_rate_mode == 1) {
      e->br_req_hist = (int64_t*)calloc(cfg->n_bit_rates, 8);
      e->br_prov_hist = (int64_t*)calloc(cfg->n_bit_rates, 8);
    }
    e->actions_output = (int64_t*)calloc(hist_cells(cfg), 8);
    e->actions_taken = (int64_t*)calloc(hist_cells(cfg), 8);
    if (cfg->env_type == ENV_RWA) {
      e->act_path = (int64_t*)calloc(cfg->k_paths + rej, 8);
      e->act_slot = (int64_t*)calloc(cfg->num_slots + rej, 8);
    }
    env_reset(b, e, 1);
  }
  return b;
}

void orc_destroy(orc_batch* b) {
  int64_t i;
  if (!b) return;
  for (i = 0; i < b->n_envs; i++) {
    orc_env* e = &b->envs[i];
    free(e->heap); free(e->pool); free(e->free_list); free(e->running); free(e->avail); free(e->spectrum);
    free(e->l_util); free(e->l_frag); free(e->l_comp); free(e->l_last);
    free(e->br_req_hist); free(e->br_prov_hist); free(e->act_path); free(e->act_slot);
    free(e->actions_output); free(e->actions_taken);
  }
  free(b->envs);
  free(b->n_paths); free(b->path_hops); free(b->path_links); free(b->path_length); free(b->path_best_mod);
  free(b->mod_se); free(b->mod_max_length); free(b->mod_min_osnr); free(b->mod_inband_xt);
  free(b->edge_iter_order); free(b->node_probs); free(b->bit_rates); free(b->bit_rate_probs);
  free(b->class_probs); free(b->class_reward);
  free(b);
}

/* mask may be NULL (= all envs) */
void orc_reset(orc_batch* b, int full, const uint8_t* mask) {
  int64_t i;
  for (i = 0; i < b->n_envs; i++)
    if (!mask || mask[i]) env_reset(b, &b->envs[i], full);
}

void orc_policy(orc_batch* b, int policy, int32_t* actions /*[n][4]*/) {
  int64_t i;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
  for (i = 0; i < b->n_envs; i++) env_policy(b, &b->envs[i], policy, actions + i * 4);
}

/* auto_reset: soft reset() right after a step that returned done (SB3 VecEnv behaviour).
   obs: [n][obs_dim] observation AFTER the step (and after the auto reset), or NULL. */
int orc_step(orc_batch* b, const int32_t* actions /*[n][4]*/, int auto_reset, double* reward, uint8_t* done,
             double* info /*[n][n_info]*/, double* obs) {
  int64_t i;
  int err = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
  for (i = 0; i < b->n_envs; i++) {
    orc_env* e = &b->envs[i];
    int rc = env_step(b, e, actions + i * 4, reward + i, done + i, info + i * b->n_info);
    if (rc) { e->error = rc; err = rc; continue; }
    if (done[i] && auto_reset) env_reset(b, e, 0);
    if (obs && b->obs_dim) env_observation(b, e, obs + i * b->obs_dim);
  }
  return err;
}

void orc_observation(orc_batch* b, double* obs) {
  int64_t i;
  if (!b->obs_dim) return;
  for (i = 0; i < b->n_envs; i++) env_observation(b, &b->envs[i], obs + i * b->obs_dim);
}

/* fused policy+step loop with auto reset; returns the number of accepted services over the run */
int64_t orc_run(orc_batch* b, int policy, int64_t n_steps) {
  int64_t i, acc = 0;
#ifdef _OPENMP
#pragma omp parallel for schedule(static) reduction(+ : acc)
#endif
  for (i = 0; i < b->n_envs; i++) {
    orc_env* e = &b->envs[i];
    double* info = (double*)malloc(sizeof(double) * b->n_info);
    int64_t s;
    for (s = 0; s < n_steps; s++) {
      int32_t a[4];
      double r;
      uint8_t d;
      env_policy(b, e, policy, a);
      if (env_step(b, e, a, &r, &d, info)) break;
      if (r > 0) acc++;
      if (d) env_reset(b, e, 0);
    }
    free(info);
  }
  return acc;
}

/* PathOnlyFirstFitAction: the agents' path choices for POLICY_PATH_FF */
void orc_set_paths(orc_batch* b, const int32_t* paths /*[n]*/) {
  int64_t i;
  for (i = 0; i < b->n_envs; i++) b->envs[i].path_choice = paths[i];
}
/* seed(seed) (optical_network_env.py:205-210): self.rng = random.Random(seed); nothing else changes */
void orc_reseed(orc_batch* b, const uint32_t* mt_state /*[n][625]*/, const uint8_t* mask) {
  int64_t i;
  for (i = 0; i < b->n_envs; i++)
    if (!mask || mask[i]) {
      orc_env* e = &b->envs[i];
      if (!e->reseeded && b->cfg.env_type != ENV_RWA) {
        memcpy(e->mt_init, e->mt, 624 * 4);
        e->mti_init = e->mti;
        e->reseeded = 1;
      }
      memcpy(e->mt, mt_state + i * 625, 624 * 4);
      e->mti = (int32_t)mt_state[i * 625 + 624];
    }
}
void orc_get_action_histograms(orc_batch* b, int64_t env, int64_t* out /*[2][(k+1)][(S+1)], RMCSA [2][(k+1)][(M+1)][(C+1)][(S+1)]*/) {
  size_t n = hist_cells(&b->cfg);
  memcpy(out, b->envs[env].actions_output, n * 8);
  memcpy(out + n, b->envs[env].actions_taken, n * 8);
}

/* state read-back for the parity tests */
void orc_get_service(orc_batch* b, double* out /*[n][6]: at, ht, src, dst, bit_rate, id*/) {
  int64_t i;
  for (i = 0; i < b->n_envs; i++) {
    const service* s = &b->envs[i].cur;
    double* o = out + i * 6;
    o[0] = s->at; o[1] = s->ht; o[2] = s->src; o[3] = s->dst; o[4] = s->bit_rate; o[5] = s->id;
  }
}
void orc_get_counters(orc_batch* b, int64_t* out /*[n][8]*/) {
  int64_t i;
  for (i = 0; i < b->n_envs; i++) {
    const orc_env* e = &b->envs[i];
    int64_t* o = out + i * 8;
    o[0] = e->services_processed; o[1] = e->services_accepted;
    o[2] = e->episode_services_processed; o[3] = e->episode_services_accepted;
    o[4] = e->bit_rate_requested; o[5] = e->bit_rate_provisioned;
    o[6] = e->episode_bit_rate_requested; o[7] = e->episode_bit_rate_provisioned;
  }
}
void orc_get_slots(orc_batch* b, int64_t env, uint8_t* out /*[C][E][S]*/) {
  memcpy(out, b->envs[env].avail, (size_t)b->cfg.num_cores * b->cfg.n_links * b->cfg.num_slots);
}
/* bulk read-backs for the every-env comparisons: the slot arrays bit-packed (bit s of word s/64 of a row = slot s free; rows
 * of `words` 64-bit words, `stride` words per env), the link statistics of all envs [n][4][E], the network statistics [n][4] */
void orc_get_slots_packed_all(orc_batch* b, uint64_t* out, int words, int64_t stride) {
  int64_t i;
  const int rows = b->cfg.num_cores * b->cfg.n_links, S = b->cfg.num_slots;
  for (i = 0; i < b->n_envs; i++) {
    const uint8_t* a = b->envs[i].avail;
    uint64_t* o = out + i * stride;
    int r, s;
    memset(o, 0, (size_t)stride * 8);
    for (r = 0; r < rows; r++)
      for (s = 0; s < S; s++)
        if (a[(size_t)r * S + s]) o[(size_t)r * words + (s >> 6)] |= 1ull << (s & 63);
  }
}
void orc_get_link_stats_all(orc_batch* b, double* out /*[n][4][E]*/) {
  int64_t i;
  const int E = b->cfg.n_links;
  for (i = 0; i < b->n_envs; i++) {
    const orc_env* e = &b->envs[i];
    double* o = out + (size_t)i * 4 * E;
    memcpy(o, e->l_util, 8 * E); memcpy(o + E, e->l_frag, 8 * E);
    memcpy(o + 2 * E, e->l_comp, 8 * E); memcpy(o + 3 * E, e->l_last, 8 * E);
  }
}
void orc_get_net_stats_all(orc_batch* b, double* out /*[n][4]*/) {
  int64_t i;
  for (i = 0; i < b->n_envs; i++) {
    const orc_env* e = &b->envs[i];
    out[4 * i] = e->g_throughput; out[4 * i + 1] = e->g_compactness; out[4 * i + 2] = e->g_last_update; out[4 * i + 3] = e->current_time;
  }
}
void orc_get_active_all(orc_batch* b, int32_t* out /*[n]*/) {
  int64_t i;
  for (i = 0; i < b->n_envs; i++) out[i] = b->envs[i].heap_n;
}
void orc_get_spectrum(orc_batch* b, int64_t env, int32_t* out /*[E]*/) { /* QoSConstrainedRA: available_spectrum */
  memcpy(out, b->envs[env].spectrum, (size_t)b->cfg.n_links * 4);
}
void orc_get_link_stats(orc_batch* b, int64_t env, double* out /*[4][E]*/) {
  const orc_env* e = &b->envs[env];
  int E = b->cfg.n_links;
  memcpy(out, e->l_util, 8 * E); memcpy(out + E, e->l_frag, 8 * E);
  memcpy(out + 2 * E, e->l_comp, 8 * E); memcpy(out + 3 * E, e->l_last, 8 * E);
}
void orc_get_net_stats(orc_batch* b, int64_t env, double* out /*[4]*/) {
  const orc_env* e = &b->envs[env];
  out[0] = e->g_throughput; out[1] = e->g_compactness; out[2] = e->g_last_update; out[3] = e->current_time;
}
int32_t orc_get_active(orc_batch* b, int64_t env) { return b->envs[env].heap_n; }
