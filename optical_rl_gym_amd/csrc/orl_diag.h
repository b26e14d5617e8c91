// orl_diag.h — DIAGNOSTIC builds only (-DORL_DIAG): what a build that leaves a phase of the persistent kernel out puts in its
// place, so that the rocprofv3 instruction counters of the rest can be read (tools/valu_ab.sh: the difference to the full kernel is
// that phase's share), and the A/B switches of single design choices.  THE RESULTS OF SUCH BUILDS ARE WRONG (the switches
// excepted).  The product library is never built with -DORL_DIAG: every hook point (ORL_DIAG_*) in orl_device_split.h and
// orl_kernels.hip then compiles to the product code, and nothing of this file is seen.
//
//   -DORL_DIAG -DORL_X_SKIP_SVC     constant services instead of svc_generate
//   -DORL_DIAG -DORL_X_SKIP_SCAN    a constant action instead of the slot scan
//   -DORL_DIAG -DORL_X_SKIP_PUSH    no pending-release entry for an accepted service
//   -DORL_DIAG -DORL_X_SKIP_REL     no release detection
//   -DORL_DIAG -DORL_X_SKIP_ROWS    no row phase
//   -DORL_DIAG -DORL_X_SKIP_STAT    constants instead of the row summary
//   -DORL_DIAG -DORL_X_SKIP_ROUND2  no second round for the B lanes of the row phase
//   -DORL_DIAG -DORL_X_SKIP_F64     no float64 running averages
//   -DORL_DIAG -DORL_X_NOMINI       (correct results) the control phase's record words stay in the global records
//   -DORL_DIAG -DORL_X_NOOCG        (correct results) cache level 1 recomputes a row's contribution to the compactness sums
//   -DORL_DIAG -DORL_X_JITTER       (correct results) the two wavefronts of the pair form sleep a pseudo-random time before every
//                                   signal and after every wait of their hand-over: a stress of its ordering (tests)
#pragma once
#ifdef ORL_DIAG
#ifdef ORL_X_SKIP_SVC
#define ORL_DIAG_INSTEAD_OF_SERVICES if (need) { svb.q = 0.08; svb.ht = 20.0; svb.pk = 3u | (7u << 10) | (30u << 20); svb.cnt = 8 << 8; }
#endif
#ifdef ORL_X_SKIP_SCAN
#define ORL_DIAG_INSTEAD_OF_SCAN a[0] = 0; a[1] = (int)(desc & 63u); a[2] = 0; a[3] = 0;
#endif
#ifdef ORL_X_SKIP_PUSH
#define ORL_DIAG_NO_PUSH 1
#endif
#ifdef ORL_X_SKIP_REL
#define ORL_DIAG_INSTEAD_OF_RELEASES soon.dirty = 0;
#endif
#ifdef ORL_X_SKIP_ROWS
#define ORL_DIAG_NO_ROWS 1
#endif
#ifdef ORL_X_SKIP_STAT
#define ORL_DIAG_INSTEAD_OF_ROW_SUMMARY                                                                                       \
  after.free_ = (int)(a[0] & 255ull) + 1; after.nu = 3; after.nf = 3; after.lo = 2; after.hi = 200; after.occ = 198; after.fb = 2; \
  max_empty = 7; edge = 1;
#endif
#ifdef ORL_X_SKIP_ROUND2
#define ORL_DIAG_NO_SECOND_ROUND 1
#endif
#ifdef ORL_X_SKIP_F64
#define ORL_DIAG_NO_F64 1
#endif
#ifdef ORL_X_NOMINI
#define ORL_DIAG_NO_MINI 1
#endif
#ifdef ORL_X_NOOCG
#define ORL_DIAG_NO_OCG 1
#endif
#ifdef ORL_X_JITTER
#define ORL_DIAG_JITTER() do { const int n_ = (int)((clock64() >> 5) & 31); for (int i_ = 0; i_ < n_; i_++) __builtin_amdgcn_s_sleep(3); } while (0)
#endif
#ifdef ORL_X_WAVESYNC
#define ORL_DIAG_WAVE_SYNC 1
#endif
#endif  // ORL_DIAG
#ifndef ORL_DIAG_JITTER
#define ORL_DIAG_JITTER() do { } while (0)
#endif
