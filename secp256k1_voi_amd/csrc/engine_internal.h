// engine_internal.h — pieces shared by the translation units of the engine (engine.hip, msm.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <new>

#include "../../include/secp256k1_voi_amd.h"
#include "fe.h"
#include "point.h"

using namespace s2k;

// Most per-key tables a verification call builds (9 KiB each: 36 GiB at the cap; the buffer is sized by the batch,
// n / min_group tables, so a 2^20 batch holds 2.4 GB whatever this says).  Was 2^18 until the end of round 3: a batch of
// 2^24 signatures of 2^20 keys then built no tables at all (threshold raised to 64 per key) and took 129 ms instead of 80.
constexpr uint32_t KG_MAX_TABLES_DEFAULT = 1u << 22;

// ---- generator tables (layout: engine.hip) ----
// The window width is a RUNTIME property of a table since round 5: a context starts on a narrow table
// that is built while it is created (S2K_GT_BITS_FIRST: 20 bits, 0.8 GiB, < 0.1 s) and moves to the wide one when a background
// thread has built it (S2K_GT_BITS: the width an automatic context aims for when the device has the memory; else 24, else 22).
// Same-box A/B on MI355X, ms per 2^20 verifications / table size / time to build: round 3, ungrouped flow (tools/ab_gtbits.sh)
// 16 bits 9.21 / 64 MiB / 0.05 s, 20 bits 9.10 / 0.8 GiB, 22 bits 9.04 / 3 GiB / 0.3 s, 24 bits 9.03 / 11 GiB / 0.9 s; end of
// round 4, grouped flow (tools/gpu_gtbits_keyset.sh): 22 / 24 / 26 bits 4.95-4.97 / 4.97 / 4.85-4.87 ms, key set with 5-bit
// joint tables 2.46-2.47 / 2.42 / 2.41-2.42 ms.  26 bits are TEN windows (nine would take 29 bits and 309 GB): 40 GiB per device,
// shared by the contexts of a process, 3.1 s to build - which no caller waits for any more.
#ifndef S2K_GT_BITS
#define S2K_GT_BITS 26
#endif
#ifndef S2K_GT_BITS_FIRST
#define S2K_GT_BITS_FIRST 20
#endif
constexpr int GT_BITS_TARGET = S2K_GT_BITS, GT_BITS_FIRST = S2K_GT_BITS_FIRST, GT_BITS_MIN = 8, GT_BITS_MAX = 26;
static_assert(GT_BITS_TARGET >= GT_BITS_MIN && GT_BITS_TARGET <= GT_BITS_MAX, "generator window width out of range");   // (26 bits: 10 windows, 43 GB)
static_assert(GT_BITS_FIRST >= GT_BITS_MIN && GT_BITS_FIRST <= GT_BITS_MAX, "first generator window width out of range");
inline constexpr uint32_t gt_windows_of(int bits) { return (uint32_t)((256 + bits - 1) / bits); }
inline constexpr size_t gt_entries_of(int bits) { return (size_t)gt_windows_of(bits) << bits; }
inline constexpr size_t gt_bytes_of(int bits) { return gt_entries_of(bits) * 64; }

// what a kernel gets: the table and its geometry, by value.  A CALL uses one view for all its launches (s2k_ctx::gt_call,
// loaded by ctx_enter): the ladder kernels leave lanes for the worklist kernel whose tags name an entry of the table they ran
// on (WL_DOUBLE_LAST, engine.hip), so the background build may publish the wide table between two calls, never inside one.
struct gt_view {
  const uint32_t* p;
  uint32_t bits, windows;
};

S2K_DEV apt gt_load(const gt_view& gt, uint32_t window, uint32_t digit) {
  const uint4* p = reinterpret_cast<const uint4*>(gt.p + ((((size_t)window << gt.bits) | digit) << 4));
  uint4 a = p[0], b = p[1], c = p[2], d = p[3];
  apt r;
  r.x.v[0] = a.x; r.x.v[1] = a.y; r.x.v[2] = a.z; r.x.v[3] = a.w;
  r.x.v[4] = b.x; r.x.v[5] = b.y; r.x.v[6] = b.z; r.x.v[7] = b.w;
  r.y.v[0] = c.x; r.y.v[1] = c.y; r.y.v[2] = c.z; r.y.v[3] = c.w;
  r.y.v[4] = d.x; r.y.v[5] = d.y; r.y.v[6] = d.z; r.y.v[7] = d.w;
  return r;
}
// next digit (`bits` wide, 8 .. 26) of a 256-bit scalar held in u[0..7] (consumed from the bottom)
S2K_DEV uint32_t gt_next_digit(uint32_t u[8], uint32_t bits) {
  uint32_t d = u[0] & ((1u << bits) - 1u);
#pragma unroll
  for (int i = 0; i < 7; ++i) u[i] = (u[i] >> bits) | (u[i + 1] << (32u - bits));
  u[7] >>= bits;
  return d;
}

// big-endian 32-byte strings at arbitrary alignment (65-byte point records)
S2K_DEV void load_be32_unaligned(uint32_t out[8], const uint8_t* p) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const uint8_t* q = p + (7 - i) * 4;
    out[i] = ((uint32_t)q[0] << 24) | ((uint32_t)q[1] << 16) | ((uint32_t)q[2] << 8) | q[3];
  }
}
S2K_DEV void store_be32_unaligned(uint8_t* p, const uint32_t in[8]) {
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    uint8_t* q = p + (7 - i) * 4;
    q[0] = (uint8_t)(in[i] >> 24); q[1] = (uint8_t)(in[i] >> 16); q[2] = (uint8_t)(in[i] >> 8); q[3] = (uint8_t)in[i];
  }
}

// Largest batch the wave-per-signature ladders (k_verify_row / k_schnorr_row / k_recover_row, engine.hip) take: above it the
// lane-per-signature kernels win (profiles/r05_small_batch_ab.txt, ECDSA: 0.19 against 0.68 ms up to 1024 signatures - one wave
// per SIMD -, 0.41 against 0.7-1.1 at 4096, 0.71 against 1.08 at 8192, 1.31 against 1.10 at 16384 device-resident; from host
// memory 0.28 against 0.74 ms at 1024, 0.50 against 0.75 at 4096, 0.77 against 0.78 at 8192).
// Largest batch the four-lanes-per-signature ladder (k_verify_quad) takes, above the row kernels' threshold
// (profiles/r05_mid_batch_ab.txt: 0.35 ms for anything from 2^11 to 2^14 signatures - one wave per SIMD at 2^14 -, 0.61 at 2^15,
// 1.10 at 2^16; the lane kernels 0.59-0.72 over that range, the row kernels 0.26 at 2^11 and 0.42 at 2^12).
#ifndef S2K_QUAD_MAX_DEFAULT
#define S2K_QUAD_MAX_DEFAULT 32768
#endif
#ifndef S2K_ROW_MAX_DEFAULT
#define S2K_ROW_MAX_DEFAULT 3072
#endif

// ---- host side ----
struct s2k_ctx {
  int device = -1;
  bool gt_held = false;         // this context holds a reference on its device's generator tables (gtable_acquire)
  int gt_fixed = 0;             // s2k_ctx_create_ex with an explicit width: only the table of that width is used (0: the widest ready)
  gt_view gt_call = {nullptr, 0, 0};   // the generator table of the call being enqueued: loaded ONCE per call (ctx_enter), passed to every launch of it
  int dbg_gt_swap = 0;          // test hook (s2k_debug_gt_swap_in_call): publish this width between the ladder and the worklist launch of the next call
  char gt_note[200] = {0};      // s2k_ctx_gt_note's answer (copied under the registry's lock)
  void* ws = nullptr;           // workspace of the verification path
  size_t ws_bytes = 0;
  void* msm_ws = nullptr;       // workspace of the multi-scalar multiplication
  size_t msm_ws_bytes = 0;
  void* rlc_save = nullptr;     // kept terms of a rejected BIP-340 batch while its failing signatures are located
  size_t rlc_save_bytes = 0;
  // host-buffer entry point: device staging for inputs / verdicts, a copy stream and a compute
  // stream, events that chain them (created on first use)
  void* io = nullptr;
  size_t io_bytes = 0;
  // small synchronous host calls (ctx_small_block, engine.hip): a page-locked block the kernels read and write in place
  uint8_t* h_small = nullptr;
  uint8_t* d_small = nullptr;   // the same block as the device addresses it
  size_t h_small_bytes = 0;
  hipStream_t s_copy = nullptr, s_comp = nullptr;
  // A child context of submit / wait owns buffers, not streams: its copy stream is its parent's, its two compute streams
  // those of its LANE (even tickets: the parent's own; odd tickets: a second pair).  Within a lane tickets run their kernels
  // strictly one after the other, each with the two-stream overlap of a resident call; the two lanes run beside each other
  // like two contexts (worth 3 %), and the transfers run ahead of both (also for what was tried
  // before: a stream set per ticket - the front ends crawl beside three ladders, empty tail launches of 161 VGPRs wait
  // milliseconds for room, and streams beyond the hardware queues serialise at random).
  bool streams_shared = false;
  hipEvent_t ev_copied[2] = {nullptr, nullptr};
  hipEvent_t ev_arrival[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};   // pieces of a pinned batch on their way in
  int cu_count = 0;
  // Calls on one context share its workspaces.  Every enqueue ends by recording ev_done on its
  // stream, and an enqueue on a different stream than the previous one first waits for it, so
  // consecutive calls never overlap on the device whatever streams they use.
  hipEvent_t ev_done = nullptr;
  hipStream_t last_stream = nullptr;
  bool have_last = false;
  // optional per-kernel timing of the verification path (s2k_ctx_profile): event quadruples
  // around k_scalar_prep / k_verify_fast / k_verify_fallback, read back by s2k_ctx_profile_read
  bool prof_on = false;
  hipEvent_t* prof_ev = nullptr;
  size_t prof_cap = 0, prof_used = 0;
  uint64_t* clk = nullptr;      // device: s_memtime / s_memrealtime stamps of one wave of k_verify_fast
  // the same for the multi-scalar path (s2k_ctx_profile_msm): MSM_PROF_EV events per call, see msm_prof_mark
  hipEvent_t* msm_prof_ev = nullptr;
  size_t msm_prof_cap = 0, msm_prof_used = 0;
  bool msm_prof_on = false;
  // repeated-key path (keyed.hip): grouping arrays and per-key tables, grown on demand
  uint32_t quad_max = S2K_QUAD_MAX_DEFAULT;   // ... and up to this many the four-lanes-per-signature ladder (s2k_ctx_set_mid_batch_max)
  uint32_t row_max = S2K_ROW_MAX_DEFAULT;   // batches of up to this many signatures take the wave-per-signature ladder (s2k_ctx_set_small_batch_max)
  int kg_mode = S2K_KEYS_ADAPTIVE;
  uint32_t kg_min_group = 0, kg_hash_bits = 0, kg_max_tables = KG_MAX_TABLES_DEFAULT;
  uint32_t kg_table_cap = 0;         // 0: none; else the table count the device had memory for (s2k_internal_key_reserve)
  uint64_t kg_seed = 0;              // hash seed of the key grouping (operating-system randomness, per context)
  hipStream_t s_aux = nullptr;       // scalar preparation runs here, beside the grouping and the table kernels
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_mid = nullptr, ev_part0 = nullptr, ev_part1 = nullptr;
  hipStream_t s_aux2 = nullptr;      // two-part flow: the second half of the tables is built here, beside the first half's ladder
  hipStream_t s_msm_tail = nullptr;  // multi-scalar multiplication in two parts (msm.hip): the upper windows' tail, beside the lower windows' bucket pass
  uint32_t kg_parts = 1;             // 1: all tables, then all ladders; 2: the two-part flow (S2K_KEYED_PARTS; measured slower)
  uint32_t gp_first_percent = 60;    // share of k_generator_part launched beside k_key_chain (the rest: after k_key_odd)
  void* kg = nullptr;
  size_t kg_bytes = 0;
  void* ktab = nullptr;
  size_t ktab_bytes = 0;
  void* xkeys = nullptr;             // BIP-340 over a key set: the x-only keys of a call's signatures, expanded from the set
  size_t xkeys_bytes = 0;
  uint32_t* kg_counters = nullptr;   // device, KG_COUNTERS words (of the last call; null: it did not group)
  uint32_t kg_last_max_tables = 0;   // table cap of that call (the device counter of tables is not clamped)
  uint32_t* last_wl_count = nullptr; // device: the last verification call's complete-formula worklist length
  // S2K_KEYS_ADAPTIVE (the default): what the grouping of recent calls found, learned WITHOUT synchronising.  k_key_counts of
  // a grouped call stores (sequence number << 32 | signatures that got a table) into a slot of `kga_note` (page-locked host
  // memory, eight slots); the next calls read the slots that have arrived.  After KG_ADAPT_MISSES consecutive observed
  // batches (of at least KG_ADAPT_MIN_BATCH signatures) in which no key reached the threshold, the next KG_ADAPT_SKIP such
  // batches are verified without looking (the general ladder, exactly as S2K_KEYS_OFF), then one batch looks again; the first
  // observed batch that does find a group ends the skipping.  Children of submit / wait use their parent's state (parent).
  s2k_ctx* parent = nullptr;                // of a child context of submit / wait (else null)
  unsigned long long* kga_note = nullptr;   // host, page-locked, KG_ADAPT_SLOTS entries
  uint32_t kga_seq = 0;                     // sequence number of the last grouped call that was asked to leave a note
  uint32_t kga_seen = 0;                    // ... of the last note taken into account
  uint32_t kga_miss_streak = 0, kga_skip_left = 0;
  bool kga_skipping = false;
  uint32_t kga_skipped = 0, kga_probes = 0, kga_observed = 0;   // totals (s2k_ctx_key_grouping_adaptive)
  unsigned long long* kg_note_dst = nullptr;   // where THIS call's k_key_counts leaves its note (null: nowhere), and its number
  uint32_t kg_note_seq = 0;
  // submit / wait (s2k_ecdsa_verify_batch_submit, s2k_wait): child contexts on the same device take the submitted
  // batches in turn, so that one batch's transfer, grouping and tables run beside another's ladder.  A child owns
  // its own workspaces, staging and streams (the generator tables are shared per device); `pipe` is empty in a child.
  struct pipe_slot {
    s2k_ctx* ctx = nullptr;        // the child context (created on first use)
    uint64_t ticket = 0;           // ticket in flight on it (0: none)
    uint8_t* dst = nullptr;        // where its verdicts go (the caller's array)
    size_t n = 0;
    hipEvent_t done = nullptr;     // recorded behind the ticket's last operation (the verdicts' copy out)
    uint8_t* h_valid = nullptr;    // page-locked landing buffer of the verdicts (when dst is pageable)
    size_t h_valid_bytes = 0;
    bool direct = false;           // dst is page-locked itself: the device-to-host copy lands there
    hipEvent_t t_begin = nullptr, t_copied = nullptr, t_end = nullptr;   // s2k_ctx_ticket_timing: before / behind the ticket's host-to-device copies, behind its verdicts
    bool timed = false;            // the three events were recorded for the ticket in flight
  };
  static constexpr unsigned PIPE_SLOTS = 4;   // batches in flight: two LANES (even and odd tickets), each with one batch
                                              // computing and one arriving
  pipe_slot pipe[PIPE_SLOTS];
  hipStream_t lane1_comp = nullptr, lane1_aux = nullptr;   // the compute streams of the odd tickets (the even ones: s_comp, s_aux)
  uint64_t pipe_next = 1;          // next ticket (ticket t runs on slot t mod PIPE_SLOTS)
  uint64_t pipe_failed[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // tickets retired with an error (by a later submit), and their codes
  int pipe_failed_rc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned pipe_failed_n = 0;
  bool pipe_timing = false;        // s2k_ctx_ticket_timing
  uint64_t pipe_times_ticket[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // the last retired tickets and their times (s2k_ticket_times)
  float pipe_times_ms[8][2] = {};
  unsigned pipe_times_n = 0;
  uint64_t generation = 0;         // distinguishes contexts that reuse an address (key sets compare it)
  char err[512] = {0};
};

// Most items one call takes: the worklist entries of the ladder kernels carry a 2-bit tag above a 30-bit index (WL_TAG_LIMIT,
// engine.hip), so an index with bit 30 set must never reach them.
constexpr size_t S2K_MAX_BATCH = (size_t)1 << 30;

#include <atomic>
inline std::atomic<size_t>& s2k_internal_key_table_limit() {   // s2k_set_table_memory_budgets (engine.hip), read by keyed.hip
  static std::atomic<size_t> v{0};
  return v;
}
gt_view s2k_internal_gt_load(const s2k_ctx* ctx);   // engine.hip: the generator table a call that starts now uses (one load per call)
size_t s2k_internal_gt_pending_bytes(int device);   // engine.hip: device memory the background table build is about to allocate

inline thread_local char g_err[512];

inline int fail(s2k_ctx* ctx, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (ctx) snprintf(ctx->err, sizeof ctx->err, "%s", buf);
  snprintf(g_err, sizeof g_err, "%s", buf);
  return code;
}
#define HIP_TRY(ctx, expr)                                                                       \
  do {                                                                                           \
    hipError_t e_ = (expr);                                                                      \
    if (e_ != hipSuccess) return fail(ctx, S2K_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

// grow-only device buffer owned by the context
inline int ctx_reserve(s2k_ctx* ctx, void** p, size_t* have, size_t need) {
  if (need <= *have) return S2K_OK;
  if (*p) {
    HIP_TRY(ctx, hipFree(*p));
    *p = nullptr;
    *have = 0;
  }
  HIP_TRY(ctx, hipMalloc(p, need));
  *have = need;
  return S2K_OK;
}
static inline unsigned blocks_for(size_t n) { return (unsigned)((n + 255) / 256); }
// lane stride of the [word][lane] planes: n rounded up to a wave, plus S2K_STRIDE_PAD lanes so
// that consecutive planes need not start at the same power-of-two offset
#ifndef S2K_STRIDE_PAD
#define S2K_STRIDE_PAD 0
#endif
static inline size_t lane_stride(size_t n) { return ((n + 63) & ~(size_t)63) + S2K_STRIDE_PAD; }


// Serialisation of the calls of one context across streams (see s2k_ctx::ev_done).
inline int ctx_enter(s2k_ctx* ctx, hipStream_t st) {
  ctx->gt_call = s2k_internal_gt_load(ctx);   // every launch of this call reads this view and no other
  if (ctx->have_last && ctx->last_stream != st) HIP_TRY(ctx, hipStreamWaitEvent(st, ctx->ev_done, 0));
  return S2K_OK;
}
void s2k_internal_gt_kick(int device);      // engine.hip: a call has been enqueued on the device (the background table build may start)
inline int ctx_leave(s2k_ctx* ctx, hipStream_t st) {
  s2k_internal_gt_kick(ctx->device);
  HIP_TRY(ctx, hipEventRecord(ctx->ev_done, st));
  ctx->last_stream = st;
  ctx->have_last = true;
  return S2K_OK;
}
// grid of the complete-formula worklist kernels: enough workgroups for the whole batch when
// every lane is undecided (adversarial input), at most 8 per CU (they loop over the list)
static inline unsigned fallback_blocks(const s2k_ctx* ctx, size_t n) {
  size_t want = (n + 255) / 256, cap = (size_t)(ctx->cu_count > 0 ? ctx->cu_count : 256) * 8;
  return (unsigned)(want < cap ? want : cap);
}

// second / third stream of the grouped flows and of the BIP-340 whole-batch check, with the events that fork and join them
inline int ctx_aux_streams(s2k_ctx* ctx) {
  if (ctx->ev_fork) return S2K_OK;
  if (!ctx->s_aux) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->s_aux, hipStreamNonBlocking));   // (a child context is given its parent's)
  // The third stream only where it is used (the two-part flow, off by default): the runtime multiplexes a process's
  // streams onto four hardware queues, and with the caller's stream, the copy stream and the compute stream a fifth one
  // made two of them share a queue - when those were the copy and a compute stream, the host-buffer path lost its
  // overlap (2^20 verifications from pinned memory: 9.3 instead of 7.3 ms, in some processes and not in others).
  if (ctx->kg_parts > 1 && !ctx->s_aux2) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->s_aux2, hipStreamNonBlocking));
  HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_fork, hipEventDisableTiming));
  HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
  HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_mid, hipEventDisableTiming));
  HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_part0, hipEventDisableTiming));
  HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_part1, hipEventDisableTiming));
  return S2K_OK;
}
// after a fork: whatever happened, the caller's stream waits for the second one again (an error return must not leave
// work of this call in flight on a stream the next call does not wait for)
inline void ctx_aux_join(s2k_ctx* ctx, hipStream_t st) {
  if (!ctx->s_aux) return;
  (void)hipEventRecord(ctx->ev_join, ctx->s_aux);
  (void)hipStreamWaitEvent(st, ctx->ev_join, 0);
}

// stage timing of the multi-scalar path (s2k_ctx_profile_msm): events before the front end (parsing, or the BIP-340
// preparation with grouping and key terms) | the sort | the bucket pass | stitching, reduction and tree | the Horner
// tail | the end
constexpr int MSM_PROF_EV = 6;
inline void msm_prof_mark(s2k_ctx* ctx, hipStream_t st, int slot) {
  if (!ctx->msm_prof_on || ctx->msm_prof_used + MSM_PROF_EV > ctx->msm_prof_cap) return;
  (void)hipEventRecord(ctx->msm_prof_ev[ctx->msm_prof_used + slot], st);
  if (slot == MSM_PROF_EV - 1) ctx->msm_prof_used += MSM_PROF_EV;
}

// submit / wait plumbing (engine.hip), shared with the encoded entry point (ingest.hip)
int s2k_internal_pipe_slot(s2k_ctx* ctx, size_t n, uint8_t* valid, s2k_ctx::pipe_slot** out);   // free slot for the next ticket
int s2k_internal_pipe_retire(s2k_ctx* ctx, s2k_ctx::pipe_slot& sl);                             // wait + deliver verdicts
void s2k_internal_pipe_issue(s2k_ctx* ctx, s2k_ctx::pipe_slot* sl, uint64_t* ticket);
void s2k_internal_drain(s2k_ctx* ctx);                                                          // after an error: nothing left in flight
bool s2k_internal_host_pinned(const void* p, size_t bytes);
// small synchronous host calls without DMA transfers (engine.hip: ctx_small_block)
bool s2k_internal_small_call(const s2k_ctx* ctx, size_t n, uint32_t flags);
int s2k_internal_small_block(s2k_ctx* ctx, const size_t* sizes, int count, uint8_t** host, uint8_t** dev);

// Synchronous host-pointer entry points are transfer, kernels, answer in series.  Two verifiers (two contexts on two host
// threads) that take whole batches alternately could hide one's transfer behind the other's kernels, but left alone they fall
// into lock step - both copy, then both compute.  So the PHASES of such calls take turns per device: one call's transfer in,
// one call's kernels (and answer out) at a time.  s2k_phase_transfer locks the transfer phase of a call that moves at least
// PHASE_MIN_BYTES (smaller calls: no-op guards); `landed()` waits for the copies on the stream and hands over to the kernel
// phase.  A lone verifier meets no contention; what it pays is that its kernels are enqueued once its copies have landed
// instead of behind them on the stream (tens of microseconds).  Measured: DESIGN.md section 5.
struct s2k_phase_locks {
  std::mutex xfer, comp;
};
s2k_phase_locks& s2k_internal_phase(int device);          // engine.hip
constexpr size_t S2K_PHASE_MIN_BYTES = (size_t)4 << 20;
struct s2k_phase_guard {
  s2k_phase_locks& ph;
  bool on;
  std::unique_lock<std::mutex> x, c;
  s2k_phase_guard(int device, size_t bytes) : ph(s2k_internal_phase(device)), on(bytes >= S2K_PHASE_MIN_BYTES), x(ph.xfer, std::defer_lock), c(ph.comp, std::defer_lock) {
    if (on) x.lock();
  }
  // the copies are enqueued on `st`: wait for them, let the next call transfer, take the kernel phase
  hipError_t landed(hipStream_t st) {
    if (!on) return hipSuccess;
    const hipError_t e = hipStreamSynchronize(st);
    x.unlock();
    c.lock();
    return e;
  }
};

// copy / compute streams and the events chaining them, for the host-buffer entry points
inline int ctx_streams(s2k_ctx* ctx) {
  if (!ctx->s_copy) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->s_copy, hipStreamNonBlocking));   // (a child context is given its parent's streams)
  if (!ctx->s_comp) HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->s_comp, hipStreamNonBlocking));
  for (hipEvent_t& e : ctx->ev_copied)
    if (!e) HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  return S2K_OK;
}

inline int ctx_arrival_events(s2k_ctx* ctx) {
  for (hipEvent_t& e : ctx->ev_arrival)
    if (!e) HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
  return S2K_OK;
}

// carve `count` buffers (256-byte aligned) out of the context's staging allocation, grown on demand:
// the host-buffer entry points stage their inputs and outputs here instead of allocating per call
inline int ctx_stage(s2k_ctx* ctx, const size_t* sizes, int count, uint8_t** ptrs) {
  size_t total = 0;
  for (int i = 0; i < count; ++i) total += (sizes[i] + 255) & ~(size_t)255;
  int rc = ctx_reserve(ctx, &ctx->io, &ctx->io_bytes, total + 256);
  if (rc) return rc;
  size_t off = 0;
  for (int i = 0; i < count; ++i) {
    ptrs[i] = (uint8_t*)ctx->io + off;
    off += (sizes[i] + 255) & ~(size_t)255;
  }
  return S2K_OK;
}

// ---- repeated-key path (keyed.hip; the ladder itself is k_verify_fast<MODE_ECDSA_KEYED>, engine.hip) ----
// A key that signs several signatures of a batch gets ONE table, built once: for each of the eight
// 16-bit chunks c of a 128-bit half scalar the odd multiples {1,3,..,15} * 2^(16c) Q, and 2^116 Q for
// the recoding's leading digit, all affine (one shared inversion per key) with their beta*x column.
// A signature's ladder is then 12 doublings instead of 128 around the same 66 additions.
// Two geometries.  CHUNKS = 8: what a verification call builds for itself (16-bit chunks, 12 doublings left per signature,
// 9 KiB per key).  CHUNKS = 32: key sets (s2k_keyset_*), built once, so their size is free: one chunk per 4-bit digit,
// {1,3,..,15} * 2^(4c) Q for c < 32 and the lead pair from L = 2^128 Q - a signature's ladder is 64 additions and NO
// doubling (36 KiB per key).
template <int CHUNKS>
struct kt_geom {
  static constexpr int DBL = 128 / CHUNKS;          // doublings from one chunk's base to the next (then 4 more to the lead point)
  static constexpr int LEAD = CHUNKS * 8;           // entries of L + phi(L) and (LEAD + 1) L - phi(L): the two starting points of a ladder
  static constexpr int ENTRIES = LEAD + 2;
  static constexpr int SCR = ENTRIES;               // first scratch entry; three field elements per entry (keyed.hip: kt_scratch)
  static constexpr int W_SLOT = CHUNKS + 1;         // scratch element that stays part of the table: W, the Z all entries of the key share
  static constexpr int PRE_SLOT = CHUNKS + 2;       // (32 chunks) prefix products of the nine / 33 Z, CHUNKS + 1 of them
  static constexpr int SCR_ELEMS = CHUNKS == 8 ? 18 : 2 * CHUNKS + 3;
  static constexpr int SLOTS = CHUNKS == 8 ? 72 : ((ENTRIES + (SCR_ELEMS + 2) / 3 + 7) / 8 * 8);   // entries of 128 bytes reserved per key
};
constexpr int KT_CHUNKS = 8;
constexpr int KT_LEAD = kt_geom<8>::LEAD;
constexpr int KT_ENTRIES = kt_geom<8>::ENTRIES;
constexpr int KT_SLOTS = kt_geom<8>::SLOTS;     // 72: 66 + 6 of build scratch
constexpr int KT_SCR = kt_geom<8>::SCR;
constexpr int KT_W_SLOT = kt_geom<8>::W_SLOT;   // 9
constexpr int KS_CHUNKS = 32;                   // key sets
constexpr int KS_SLOTS = kt_geom<32>::SLOTS;    // 288
// Key sets, JOINT tables: for every digit position c the sums E_a + s phi(E_b) of the chunk's odd multiples (a, b < 8, s = +-),
// so that the two half scalars' digits at a position cost ONE table addition: 32 additions per signature instead of 64.
// 128 entries per position, 80 bytes each (x and y of an affine point of the key's isomorphic curve: five quads
// [x 0-3][x 4-7][y 0-3][y 4-7][x8, y8, -, -]), 4096 entries = 320 KiB per key, built once from the 32-chunk table.
constexpr int KJ_ENTRY_QUADS = 5;
constexpr int KJ_PER_CHUNK = 128;
constexpr size_t KJ_KEY_QUADS = (size_t)KS_CHUNKS * KJ_PER_CHUNK * KJ_ENTRY_QUADS;   // 20480 quads = 320 KiB
// Wider joint tables (S2K_KEYSET_JOINT5 / S2K_KEYSET_JOINT6): the same idea on W-bit digits.  A half scalar k (odd, < 2^129) is
// k = 2^(W POS) + sum_i d_i 2^(W i), d_i = 2 w_i - (2^W - 1), w_i the W-bit windows of (k - 1) / 2, POS = ceil(128 / W) of them;
// per position the sums (2a + 1) B_i + s phi((2b + 1) B_i), a, b < 2^(W-1), s = +-, B_i = 2^(W i) Q: POS additions per signature
// instead of 32 (26 at W = 5, 22 at W = 6) for 2^(2W - 1) entries per position (0.81 MiB / 2.75 MiB per key).  The points are
// derived from the key's 32-chunk table by affine doublings and additions, so they live on the same isomorphic curve (same W
// factor at the end of the ladder).  The ladder's two starting points, 2^(W POS) Q +- phi(2^(W POS) Q), follow the positions.
template <int W>
struct kjw_geom {
  static constexpr int NE = 1 << (W - 1);                        // odd multiples per position
  static constexpr int POS = (128 + W - 1) / W;                  // digit positions
  static constexpr int PER_POS = NE * NE * 2;                    // joint entries per position
  static constexpr size_t LEAD = (size_t)POS * PER_POS;          // entries LEAD, LEAD + 1: the lead pair
  // entries of 64 bytes: x and y as eight 32-bit words each (canonical values; converted to limbs on load: 34 instructions
  // per lookup), so that a lookup is ONE aligned 64-byte fetch where an 80-byte entry of limbs straddles two or three
  static constexpr int EQ = 4;
  static constexpr size_t KEY_QUADS = (LEAD + 2) * EQ;
  static constexpr int LEAD_SHIFT = W * POS - 128;               // doublings from the chunk table's L = 2^128 Q to 2^(W POS) Q
};
static_assert(KT_SLOTS == 72 && KT_W_SLOT == 9 && kt_geom<32>::SLOTS == 288, "table geometry");
static_assert(kjw_geom<5>::POS == 26 && kjw_geom<5>::LEAD_SHIFT == 2 && kjw_geom<6>::POS == 22 && kjw_geom<6>::LEAD_SHIFT == 4, "joint geometry");
enum { KG_NKEYED = 0, KG_NTAB = 1, KG_NLEFT = 2, KG_SPLIT_T = 3, KG_SPLIT_LANE = 4, KG_ALLOC64 = 6 /* and 7 */, KG_COUNTERS = 16 };
constexpr uint32_t KG_NONE = 0xffffffffu;
constexpr size_t KG_MIN_BATCH = 256;            // smaller batches skip the grouping
// S2K_KEYS_ADAPTIVE (s2k_ctx above): batches it learns from and applies to, misses before it stops looking, batches it then skips
constexpr size_t KG_ADAPT_MIN_BATCH = (size_t)1 << 16;
constexpr uint32_t KG_ADAPT_MISSES = 2, KG_ADAPT_SKIP = 15, KG_ADAPT_SLOTS = 8;
// Signatures per key from which a table pays, measured (MI355X, 2^20 signatures, ms with tables / without):
// 4 per key 7.94 / 8.02, 5 per key 8.08 / 8.01 (the table kernels do not scale linearly and the clock sags),
// 6 per key 6.65 / 7.93, 8 per key 6.02 / 7.96, 16 per key 5.1 / 8.0
constexpr uint32_t KG_MIN_GROUP = 4;   // measured break-even: 4 per key 7.78 against 8.14 ms without tables, 3 per key no gain
struct key_groups {        // device pointers of one call
  uint32_t* counters;      // [KG_NKEYED] signatures on the keyed path, [KG_NTAB] tables, [KG_NLEFT] the rest
  uint32_t* perm;          // keyed lane -> signature
  uint32_t* ptab;          // keyed lane -> table
  uint32_t* left;          // dense lane of the general kernel -> signature
  const uint4* ktab;       // tables, KT_SLOTS * 8 quads each
  const uint4* jtab;       // key sets with joint tables: KJ_KEY_QUADS quads per key (else null)
  const uint8_t* tinfo;    // per table: 1 if the key is a valid public key
  const uint32_t* trep;    // per table: a signature that carries the key
  const uint32_t* gp;      // per signature: u1*G (Jacobian, three fin-format elements; k_generator_part)
  uint32_t max_tables;
  int chunks;              // table geometry: 8 (built per call) or 32 (key sets); 0 means 8
  int key_bytes;           // 64: X || Y (ECDSA), 32: x-only (BIP-340)
  // two-part flow: part 0 = tables [0, counters[KG_SPLIT_T]) and lanes [0, counters[KG_SPLIT_LANE]), part 1 the
  // rest; nparts == 1: everything (kernels take these from the copy of the struct they are launched with)
  uint32_t part, nparts;
};
// groups the batch's signatures by public key, then builds the tables (enqueue only, no host sync)
int s2k_internal_key_group(s2k_ctx* ctx, size_t n, const uint8_t* d_pub, int key_bytes, hipStream_t st, key_groups* out);
int s2k_internal_key_reserve(s2k_ctx* ctx, size_t n, int key_bytes);   // grow the grouping arrays / table buffer (before any fork)
size_t s2k_internal_key_bytes(const s2k_ctx* ctx, size_t n);            // what those hold for a batch of n
int s2k_internal_key_chains(s2k_ctx* ctx, const uint8_t* d_pub, hipStream_t st, const key_groups* g);
// (ev_after_odd, if any, is recorded on st between k_key_odd and k_key_cofactors)
int s2k_internal_key_tables(s2k_ctx* ctx, hipStream_t st, const key_groups* g, uint32_t part, uint32_t nparts,
                            hipEvent_t ev_after_odd);

// grouping of x-only keys for the BIP-340 whole-batch check (msm.hip): every key a group, long groups cut
// into virtual groups of KG_VGROUP signatures
constexpr uint32_t KG_VGROUP = 1024;
struct key_groups32 {
  const uint32_t *rep, *cnt, *base, *tix;   // per hash slot: first signature, size, first sorted position, first virtual group
  const uint32_t* perm;                     // sorted position -> signature
  const uint32_t* left;                     // signatures whose key found no slot (each is its own group)
  const uint32_t* vslot;                    // virtual group -> slot
  uint32_t ngroups, nleft;                  // (host values)
};
int s2k_internal_key_group32(s2k_ctx* ctx, size_t n, const uint8_t* d_pk32, hipStream_t st, key_groups32* out);
int s2k_internal_key_reserve32(s2k_ctx* ctx, size_t n);
// key sets (s2k_keyset_*): buffer layout, table build, scratch reservation, sort of a batch by key index
size_t s2k_internal_keyset_bytes(size_t n, size_t off[5]);
int s2k_internal_keyset_build(s2k_ctx* ctx, uint8_t* base, size_t n, hipStream_t st);
int s2k_internal_keyset_build_joint(s2k_ctx* ctx, const uint8_t* base, size_t n, uint4* joint, hipStream_t st);   // from the 32-chunk tables
size_t s2k_internal_keyset_joint_bytes(size_t n, int w);                  // joint tables of n keys at digit width w (4, 5, 6)
size_t s2k_internal_keyset_joint_scratch_bytes(size_t n, int w);          // build scratch of the wide layouts (0 at w = 4)
int s2k_internal_keyset_build_joint_wide(s2k_ctx* ctx, const uint8_t* base, size_t n, int w, uint4* joint, uint4* scratch, hipStream_t st);
int s2k_internal_keyset_reserve(s2k_ctx* ctx, size_t nkeys, size_t n);
int s2k_internal_keyset_sort(s2k_ctx* ctx, const uint8_t* set_base, size_t nkeys, size_t n, const uint32_t* d_kidx, hipStream_t st,
                             key_groups* out);

struct dev_buf {
  void* p = nullptr;
  ~dev_buf() {
    if (p) (void)hipFree(p);
  }
  hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 16); }
  hipError_t upload(const void* src, size_t bytes) {
    hipError_t e = alloc(bytes);
    if (e != hipSuccess || !bytes) return e;
    return hipMemcpy(p, src, bytes, hipMemcpyHostToDevice);
  }
};

