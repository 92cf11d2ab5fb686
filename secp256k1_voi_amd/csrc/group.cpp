// group.cpp — several devices behind ONE host process (s2k_group_*, include/secp256k1_voi_amd.h).
//
// The reference verifies on the caller's goroutine (secec.PublicKey.Verify, secec/ecdsa.go:171-228) and scales by running
// more goroutines; north_star's "batches shard trivially across the 8 GPUs" is the same independence, per signature.  The
// multi-process form of that (one rank per GPU, torch.distributed, one bitmap all-gather: secp256k1_voi_amd/sharding.py)
// is what bench.py runs; a cgo shim lives in ONE process and cannot be a rank.  In one process nothing has to be
// exchanged at all: a group is one context + one host thread per device, a batch is cut into contiguous index shards
// (SURVEY.md section 8e), every member pushes its shard through its own context's submit / wait pair, and the verdicts
// land in the caller's array at the shard's offset.  No RCCL, no device-to-device traffic; the only shared state is the
// job queue.  Host code only: everything that touches a GPU goes through the C-ABI of the context.
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <deque>
#include <map>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "../../include/secp256k1_voi_amd.h"

namespace {

struct shard_job {
  uint64_t ticket = 0;
  size_t lo = 0, n = 0;            // the member's shard of the batch
  const uint8_t *pub = nullptr, *dig = nullptr, *r = nullptr, *s = nullptr;   // packed arrays; encoded: pub / dig / s = the three blobs
  const s2k_group_keyset* gks = nullptr;       // key-set form: the keys named by index (kidx) into the group's key set
  const uint32_t* kidx = nullptr;
  // whole-batch forms (synchronous on the member: one verdict / one partial sum per shard, written to `result`)
  int kind = 0;                                // 0: per-signature verification (above); 1: BIP-340 whole-batch check; 2: multiscalar multiplication
  const uint8_t* seed32 = nullptr;             // kind 1
  size_t msg_len = 0;                          // kind 1: fixed message length (dig_off == nullptr), messages in `dig`
  uint8_t* result = nullptr;                   // kind 1: one byte per member (1 accept, 0 reject); kind 2: 65 bytes per member
  const uint64_t *pub_off = nullptr, *dig_off = nullptr, *sig_off = nullptr;    // encoded form only (pub_off != nullptr)
  int encoding = 0;
  size_t digest_len = 0;
  uint32_t flags = 0;
  uint8_t* valid = nullptr;
  std::chrono::steady_clock::time_point t0;
};

struct member {
  s2k_group* group = nullptr;
  size_t index = 0;                // position in the group (its key set in a s2k_group_keyset)
  int device = -1;
  int gt_bits = 0;                 // s2k_group_create_ex: the members' generator table width (0: automatic)
  s2k_ctx* ctx = nullptr;
  int create_rc = S2K_OK;
  char create_err[256] = {0};
  bool created = false;
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  std::deque<shard_job> q;
  bool stop = false;
  // statistics of the last finished shard
  double st_n = 0, st_lo = 0, st_ms = 0, st_h2d_ms = 0, st_dev_ms = 0;
  // placement (topology.cpp): the NUMA node the device hangs off (-1 unknown) and the CPUs the member's thread was bound to (0: left alone)
  int numa_node = -1, bound_cpus = 0;
};

struct pending {
  int remaining = 0;
  int rc = S2K_OK;
};

struct host_block {     // s2k_group_host_alloc
  void* p = nullptr;
  size_t bytes = 0;
};

}  // namespace

// a key set on every member's context (the tables are replicated per device, like the generator tables)
struct s2k_group_keyset {
  s2k_group* group = nullptr;
  std::vector<s2k_keyset*> sets;
  size_t n_keys = 0;
};

struct s2k_group {
  std::vector<member*> members;
  std::mutex m;                    // pending, err, settings
  std::condition_variable cv;
  std::map<uint64_t, pending> pend;
  uint64_t next_ticket = 1;
  // tickets whose result was dropped from `pend` after they FAILED: s2k_group_wait still reports them (as s2k_ctx does
  // with pipe_failed)
  uint64_t failed_ticket[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int failed_rc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned failed_n = 0;
  std::vector<host_block> blocks;
  bool timing = false;             // s2k_group_member_stats_ex was asked for: the members' contexts time their tickets
  char err[512] = {0};
};

namespace {

// the 100 microsecond nap between two polls of the oldest shard.  On the steady clock - except under ThreadSanitizer: gcc 11's
// runtime does not intercept pthread_cond_clockwait (what wait_for on the steady clock compiles to), loses track of the mutex
// across the wait and reports races that are not there; the system clock takes the intercepted pthread_cond_timedwait.
template <class Pred>
void poll_wait(std::condition_variable& cv, std::unique_lock<std::mutex>& lock, Pred pred) {
#if defined(__SANITIZE_THREAD__)
  cv.wait_until(lock, std::chrono::system_clock::now() + std::chrono::microseconds(100), pred);
#else
  cv.wait_for(lock, std::chrono::microseconds(100), pred);
#endif
}

int gfail(s2k_group* g, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (g) {
    std::lock_guard<std::mutex> lock(g->m);
    snprintf(g->err, sizeof g->err, "%s", buf);
  }
  return code;
}

void complete(member* me, const shard_job& job, int rc, s2k_ticket ctx_ticket = 0) {
  s2k_group* g = me->group;
  const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - job.t0).count();
  double dev_ms[2] = {0, 0};
  if (ctx_ticket) (void)s2k_ticket_times(me->ctx, ctx_ticket, dev_ms);   // zeros unless s2k_ctx_ticket_timing is on
  std::lock_guard<std::mutex> lock(g->m);
  me->st_n = (double)job.n;
  me->st_lo = (double)job.lo;
  me->st_ms = ms;
  me->st_h2d_ms = dev_ms[0];
  me->st_dev_ms = dev_ms[1];
  auto it = g->pend.find(job.ticket);
  if (it == g->pend.end()) return;
  if (rc && !it->second.rc) {
    it->second.rc = rc;
    snprintf(g->err, sizeof g->err, "member on device %d, signatures [%zu, %zu): %s", me->device, job.lo, job.lo + job.n,
             s2k_last_error(me->ctx));
  }
  --it->second.remaining;
  g->cv.notify_all();
}

// One member: its context is created here (the generator tables of the devices are built side by side), then shards are
// taken from the queue.  A shard is submitted as soon as it arrives, so consecutive group batches overlap on the
// member's device exactly as consecutive s2k_ecdsa_verify_batch_submit calls do.
void member_main(member* me) {
  {
    // the thread that feeds a device runs on the CPUs next to it (no-op on one node, in a cpuset without them, ...)
    me->numa_node = s2k_device_numa_node(me->device);
    me->bound_cpus = s2k_bind_thread_to_node(me->numa_node);
    s2k_ctx* ctx = nullptr;
    const int rc = s2k_ctx_create_ex(me->device, me->gt_bits, 0, &ctx);
    std::lock_guard<std::mutex> lock(me->m);
    me->ctx = ctx;
    me->create_rc = rc;
    if (rc) snprintf(me->create_err, sizeof me->create_err, "%s", s2k_last_error(nullptr));
    me->created = true;
    me->cv.notify_all();
    if (rc) return;
  }
  // Shards in flight on this member's context, oldest first (at most as many as the context has slots).  New shards are
  // preferred over waiting: the oldest shard in flight is POLLED (s2k_poll) between looks at the queue, so a shard that
  // arrives while one is being waited for is submitted within ~0.1 ms - its transfer has to fit under the ladder of the
  // shard before it.
  constexpr size_t MAX_IN_FLIGHT = 4;
  struct flying {
    shard_job job;
    s2k_ticket ticket;
  };
  std::deque<flying> inflight;
  auto finish_oldest = [&](bool block) -> bool {     // true: the oldest shard is done (and reported)
    flying& f = inflight.front();
    const int rc = block ? s2k_wait(me->ctx, f.ticket) : s2k_poll(me->ctx, f.ticket);
    if (rc == S2K_PENDING) return false;
    complete(me, f.job, rc, f.ticket);
    inflight.pop_front();
    return true;
  };
  for (;;) {
    shard_job job;
    bool have_job = false;
    bool block_on_oldest = false;
    {
      std::unique_lock<std::mutex> lock(me->m);
      if (inflight.empty())
        me->cv.wait(lock, [&] { return me->stop || !me->q.empty(); });
      else if (me->q.empty()) {
        // nothing new to submit.  With three or more shards in flight both lanes of the context have work beyond the oldest one: wait for that
        // one in the runtime (a new shard that arrives meanwhile is submitted when it is done - the device is not idle);
        // with one or two in flight a new one must not wait for it: poll (polling at 10 kHz all the time cost 2-7 % of the
        // rate: every query goes through the runtime)
        if (inflight.size() >= 3) block_on_oldest = true;
        else poll_wait(me->cv, lock, [&] { return !me->q.empty(); });
      }
      if (!block_on_oldest && !me->q.empty()) {
        job = me->q.front();
        me->q.pop_front();
        have_job = true;
      } else if (inflight.empty() && me->stop) {
        break;
      }
    }
    if (block_on_oldest) {
      (void)finish_oldest(true);
      continue;
    }
    if (have_job) {
      while (inflight.size() >= MAX_IN_FLIGHT) (void)finish_oldest(true);
      s2k_ticket t = 0;
      int rc = S2K_OK;
      if (job.kind) {
        // whole-batch forms run to their end on this thread (after the tickets in flight: completions are reported in order)
        while (!inflight.empty()) (void)finish_oldest(true);
        if (job.kind == 1) {
          int ok = 1;
          if (job.n) {
            if (job.dig_off) {       // variable-length messages: the shard's offsets, rebased to its first message
              std::vector<uint64_t> off(job.n + 1);
              for (size_t i = 0; i <= job.n; ++i) off[i] = job.dig_off[job.lo + i] - job.dig_off[job.lo];
              rc = s2k_schnorr_batch_verify_rlc(me->ctx, job.n, job.pub + job.lo * 32, job.dig + job.dig_off[job.lo], off.data(), 0,
                                                job.s + job.lo * 64, job.seed32, &ok);
            } else {
              rc = s2k_schnorr_batch_verify_rlc(me->ctx, job.n, job.pub + job.lo * 32, job.dig ? job.dig + job.lo * job.msg_len : nullptr, nullptr,
                                                job.msg_len, job.s + job.lo * 64, job.seed32, &ok);
            }
          }
          job.result[me->index] = (uint8_t)(rc == S2K_OK && ok ? 1 : 0);
        } else {
          uint8_t* out = job.result + me->index * 65;
          memset(out, 0, 65);                    // (an empty shard sums to the identity: the all-zero record)
          if (job.n) rc = s2k_multi_scalar_mult(me->ctx, job.n, job.r + job.lo * 32, job.pub + job.lo * 65, out);
        }
        complete(me, job, rc);
        continue;
      }
      if (job.n && job.pub_off)   // encoded items: the offsets are absolute, so a shard is the same blobs with shifted offset arrays
        rc = s2k_ecdsa_verify_encoded_batch_submit(me->ctx, job.n, job.pub, job.pub_off + job.lo, job.dig, job.dig_off + job.lo, job.s,
                                                   job.sig_off + job.lo, job.encoding, job.digest_len, job.flags, job.valid + job.lo, &t);
      else if (job.n && job.gks)
        rc = s2k_ecdsa_verify_batch_keyset_submit(me->ctx, job.gks->sets[me->index], job.n, job.kidx + job.lo, job.dig + job.lo * 32,
                                                  job.r + job.lo * 32, job.s + job.lo * 32, job.flags, job.valid + job.lo, &t);
      else if (job.n)
        rc = s2k_ecdsa_verify_batch_submit(me->ctx, job.n, job.pub + job.lo * 64, job.dig + job.lo * 32, job.r + job.lo * 32,
                                           job.s + job.lo * 32, job.flags, job.valid + job.lo, &t);
      if (rc || !job.n) {
        while (!inflight.empty()) (void)finish_oldest(true);   // completions are reported in submit order
        complete(me, job, rc);
      } else {
        inflight.push_back(flying{job, t});
      }
    }
    while (!inflight.empty() && finish_oldest(false)) {
    }
  }
}

}  // namespace

extern "C" {

int s2k_group_create(const int* devices, size_t n_devices, s2k_group** out) { return s2k_group_create_ex(devices, n_devices, 0, 0, out); }

// gt_bits / flags: what s2k_ctx_create_ex takes, applied to every member (0, 0: automatic tables).  The members' child contexts
// (submit / wait) inherit the width, so a group asked for a narrow table never allocates a wide one behind the caller's back.
int s2k_group_create_ex(const int* devices, size_t n_devices, int gt_bits, uint32_t flags, s2k_group** out) {
  if (!out) return S2K_ERR_ARG;
  *out = nullptr;
  if (!devices || n_devices == 0 || n_devices > 1024) return S2K_ERR_ARG;
  if (gt_bits != 0 && (gt_bits < 16 || gt_bits > 26)) return S2K_ERR_ARG;
  if (flags & ~(uint32_t)S2K_CTX_WAIT_TABLES) return S2K_ERR_ARG;
  const int count = s2k_device_count();
  if (count <= 0) return S2K_ERR_NO_DEVICE;
  for (size_t i = 0; i < n_devices; ++i)
    if (devices[i] < 0 || devices[i] >= count) return S2K_ERR_ARG;
  s2k_group* g = new (std::nothrow) s2k_group();
  if (!g) return S2K_ERR_NOMEM;
  for (size_t i = 0; i < n_devices; ++i) {
    member* me = new (std::nothrow) member();
    if (!me) {
      s2k_group_destroy(g);
      return S2K_ERR_NOMEM;
    }
    me->group = g;
    me->index = i;
    me->device = devices[i];
    me->gt_bits = gt_bits;
    g->members.push_back(me);
    me->th = std::thread(member_main, me);
  }
  int rc = S2K_OK;
  for (member* me : g->members) {
    std::unique_lock<std::mutex> lock(me->m);
    me->cv.wait(lock, [&] { return me->created; });
    if (me->create_rc && !rc) {
      rc = me->create_rc;
      snprintf(g->err, sizeof g->err, "device %d: %s", me->device, me->create_err);
    }
  }
  if (rc) {
    fprintf(stderr, "s2k_group_create: %s\n", g->err);
    s2k_group_destroy(g);
    return rc;
  }
  if (flags & S2K_CTX_WAIT_TABLES) (void)s2k_group_gt_wait(g);
  *out = g;
  return S2K_OK;
}

void s2k_group_destroy(s2k_group* g) {
  if (!g) return;
  for (member* me : g->members) {
    {
      std::lock_guard<std::mutex> lock(me->m);
      me->stop = true;
    }
    me->cv.notify_all();
    if (me->th.joinable()) me->th.join();          // (shards still queued are verified first)
    if (me->ctx) s2k_ctx_destroy(me->ctx);
    delete me;
  }
  for (host_block& b : g->blocks) {                // buffers the caller did not return
    (void)s2k_host_unregister(b.p);
    (void)munmap(b.p, b.bytes);
  }
  delete g;
}

size_t s2k_group_size(const s2k_group* g) { return g ? g->members.size() : 0; }
const char* s2k_group_last_error(const s2k_group* g) { return g ? g->err : "group is NULL"; }

static void group_wait_idle(s2k_group* g) {
  std::unique_lock<std::mutex> lock(g->m);
  g->cv.wait(lock, [&] {
    for (auto& kv : g->pend)
      if (kv.second.remaining) return false;
    return true;
  });
}

int s2k_group_set_key_grouping(s2k_group* g, int mode, uint32_t min_group, uint32_t hash_bits, uint32_t max_tables) {
  if (!g) return S2K_ERR_ARG;
  group_wait_idle(g);                              // the members' threads are parked: their contexts may be touched from here
  for (member* me : g->members) {
    const int rc = s2k_ctx_set_key_grouping(me->ctx, mode, min_group, hash_bits, max_tables);
    if (rc) return gfail(g, rc, "%s", s2k_last_error(me->ctx));
  }
  return S2K_OK;
}

int s2k_group_set_small_batch_max(s2k_group* g, uint32_t max_n) {
  if (!g) return S2K_ERR_ARG;
  group_wait_idle(g);
  for (member* me : g->members) {
    const int rc = s2k_ctx_set_small_batch_max(me->ctx, max_n);
    if (rc) return gfail(g, rc, "%s", s2k_last_error(me->ctx));
  }
  return S2K_OK;
}

int s2k_group_set_mid_batch_max(s2k_group* g, uint32_t max_n) {
  if (!g) return S2K_ERR_ARG;
  group_wait_idle(g);
  for (member* me : g->members) {
    const int rc = s2k_ctx_set_mid_batch_max(me->ctx, max_n);
    if (rc) return gfail(g, rc, "%s", s2k_last_error(me->ctx));
  }
  return S2K_OK;
}

static int group_submit(s2k_group* g, size_t n, const shard_job& proto, s2k_ticket* ticket) {
  const size_t D = g->members.size();
  // contiguous index shards, rounded up to whole workgroups so that no member gets a ragged tail but the last
  size_t per = (n + D - 1) / D;
  per = (per + 255) & ~(size_t)255;
  if (per > 0x7fffffffu) return gfail(g, S2K_ERR_ARG, "batch too large");
  uint64_t t;
  {
    std::unique_lock<std::mutex> lock(g->m);
    // at most four group batches in flight: every member keeps up to four shards on its device
    g->cv.wait(lock, [&] {
      int busy = 0;
      for (auto& kv : g->pend) busy += kv.second.remaining ? 1 : 0;
      return busy < 4;
    });
    t = g->next_ticket++;
    while (g->pend.size() > 16 && g->pend.begin()->second.remaining == 0) {   // old results: only failures are remembered
      if (g->pend.begin()->second.rc) {
        const unsigned i = g->failed_n++ % 8u;
        g->failed_ticket[i] = g->pend.begin()->first;
        g->failed_rc[i] = g->pend.begin()->second.rc;
      }
      g->pend.erase(g->pend.begin());
    }
    pending p;
    p.remaining = (int)D;
    g->pend[t] = p;
  }
  const auto now = std::chrono::steady_clock::now();
  for (size_t i = 0; i < D; ++i) {
    member* me = g->members[i];
    shard_job job = proto;
    job.ticket = t;
    job.lo = i * per < n ? i * per : n;
    job.n = job.lo + per <= n ? per : n - job.lo;
    job.t0 = now;
    {
      std::lock_guard<std::mutex> lock(me->m);
      me->q.push_back(job);
    }
    me->cv.notify_all();
  }
  *ticket = t;
  return S2K_OK;
}

int s2k_group_ecdsa_verify_batch_submit(s2k_group* g, size_t n, const uint8_t* pub, const uint8_t* dig, const uint8_t* r,
                                        const uint8_t* s, uint32_t flags, uint8_t* valid, s2k_ticket* ticket) {
  if (!g || !ticket) return S2K_ERR_ARG;
  *ticket = 0;
  if (n && (!pub || !dig || !r || !s || !valid)) return gfail(g, S2K_ERR_ARG, "null buffer");
  shard_job proto;
  proto.pub = pub;
  proto.dig = dig;
  proto.r = r;
  proto.s = s;
  proto.flags = flags;
  proto.valid = valid;
  return group_submit(g, n, proto, ticket);
}

// PublicKey.Verify(digest, sig, opts) on encoded items (s2k_ecdsa_verify_encoded_batch) across the group: the blobs are
// shared, every member takes a contiguous range of ITEMS (its slice of the three offset arrays)
int s2k_group_ecdsa_verify_encoded_batch_submit(s2k_group* g, size_t n, const uint8_t* pubs, const uint64_t* pub_off,
                                                const uint8_t* digests, const uint64_t* dig_off, const uint8_t* sigs,
                                                const uint64_t* sig_off, int encoding, size_t digest_len, uint32_t flags,
                                                uint8_t* valid, s2k_ticket* ticket) {
  if (!g || !ticket) return S2K_ERR_ARG;
  *ticket = 0;
  if (n && (!pubs || !pub_off || !digests || !dig_off || !sigs || !sig_off || !valid)) return gfail(g, S2K_ERR_ARG, "null buffer");
  shard_job proto;
  proto.pub = pubs;
  proto.dig = digests;
  proto.s = sigs;
  proto.pub_off = pub_off;
  proto.dig_off = dig_off;
  proto.sig_off = sig_off;
  proto.encoding = encoding;
  proto.digest_len = digest_len;
  proto.flags = flags;
  proto.valid = valid;
  if (n == 0) proto.pub_off = nullptr;
  return group_submit(g, n, proto, ticket);
}

int s2k_group_ecdsa_verify_encoded_batch(s2k_group* g, size_t n, const uint8_t* pubs, const uint64_t* pub_off, const uint8_t* digests,
                                         const uint64_t* dig_off, const uint8_t* sigs, const uint64_t* sig_off, int encoding,
                                         size_t digest_len, uint32_t flags, uint8_t* valid) {
  s2k_ticket t = 0;
  const int rc = s2k_group_ecdsa_verify_encoded_batch_submit(g, n, pubs, pub_off, digests, dig_off, sigs, sig_off, encoding, digest_len,
                                                             flags, valid, &t);
  if (rc) return rc;
  return s2k_group_wait(g, t);
}

// Key sets across the group (s2k_keyset_*, engine.hip): every member builds the tables of ALL the keys on its own device -
// a shard may name any key - side by side, one host thread each; a verification call is sharded like any other, the
// members' shards go through s2k_ecdsa_verify_batch_keyset_submit over their own copy.
int s2k_group_keyset_create(s2k_group* g, size_t n_keys, const uint8_t* pub_xy, int layout, s2k_group_keyset** out) {
  if (!g || !out) return S2K_ERR_ARG;
  *out = nullptr;
  if (!pub_xy || n_keys == 0) return gfail(g, S2K_ERR_ARG, "empty key set");
  group_wait_idle(g);                              // the members' threads are parked: their contexts may be used from other threads
  s2k_group_keyset* gks = new (std::nothrow) s2k_group_keyset();
  if (!gks) return gfail(g, S2K_ERR_NOMEM, "out of host memory");
  gks->group = g;
  gks->n_keys = n_keys;
  const size_t D = g->members.size();
  gks->sets.assign(D, nullptr);
  std::vector<int> rcs(D, S2K_OK);
  std::vector<std::thread> th;
  for (size_t i = 0; i < D; ++i)
    th.emplace_back([&, i] { rcs[i] = s2k_keyset_create_ex(g->members[i]->ctx, n_keys, pub_xy, layout, &gks->sets[i]); });
  for (std::thread& t : th) t.join();
  for (size_t i = 0; i < D; ++i)
    if (rcs[i]) {
      const int rc = gfail(g, rcs[i], "key set on device %d: %s", g->members[i]->device, s2k_last_error(g->members[i]->ctx));
      s2k_group_keyset_destroy(gks);
      return rc;
    }
  *out = gks;
  return S2K_OK;
}

void s2k_group_keyset_destroy(s2k_group_keyset* gks) {
  if (!gks) return;
  group_wait_idle(gks->group);
  for (s2k_keyset* ks : gks->sets) s2k_keyset_destroy(ks);
  delete gks;
}
size_t s2k_group_keyset_size(const s2k_group_keyset* gks) { return gks ? gks->n_keys : 0; }
// layout and device memory of the set on the group's first member (the members hold the same set; with S2K_KEYSET_AUTO a
// member short of memory may have fallen back to chunk tables on its own)
int s2k_group_keyset_layout(const s2k_group_keyset* gks) { return gks && !gks->sets.empty() ? s2k_keyset_layout(gks->sets[0]) : 0; }
size_t s2k_group_keyset_device_bytes(const s2k_group_keyset* gks) { return gks && !gks->sets.empty() ? s2k_keyset_device_bytes(gks->sets[0]) : 0; }

int s2k_group_ecdsa_verify_batch_keyset_submit(s2k_group* g, const s2k_group_keyset* gks, size_t n, const uint32_t* key_index,
                                               const uint8_t* dig, const uint8_t* r, const uint8_t* s, uint32_t flags, uint8_t* valid,
                                               s2k_ticket* ticket) {
  if (!g || !ticket) return S2K_ERR_ARG;
  *ticket = 0;
  if (!gks || gks->group != g) return gfail(g, S2K_ERR_ARG, "key set of another group");
  if (n && (!key_index || !dig || !r || !s || !valid)) return gfail(g, S2K_ERR_ARG, "null buffer");
  shard_job proto;
  proto.gks = gks;
  proto.kidx = key_index;
  proto.dig = dig;
  proto.r = r;
  proto.s = s;
  proto.flags = flags;
  proto.valid = valid;
  return group_submit(g, n, proto, ticket);
}

int s2k_group_ecdsa_verify_batch_keyset(s2k_group* g, const s2k_group_keyset* gks, size_t n, const uint32_t* key_index,
                                        const uint8_t* dig, const uint8_t* r, const uint8_t* s, uint32_t flags, uint8_t* valid) {
  s2k_ticket t = 0;
  const int rc = s2k_group_ecdsa_verify_batch_keyset_submit(g, gks, n, key_index, dig, r, s, flags, valid, &t);
  if (rc) return rc;
  return s2k_group_wait(g, t);
}

// BASELINE configs 3 and 4 across the group (the multi-process forms are sharding.py's
// schnorr_batch_verify_sharded / msm_sharded).  Synchronous: a shard's answer is one verdict / one point.
// BIP-340 whole-batch check: every member checks its contiguous shard as one random linear combination
// (s2k_schnorr_batch_verify_rlc: the coefficients of a call are keyed by the seed AND 32 bytes of the operating system's
// randomness, so the members' combinations are independent); the batch is accepted iff every shard is.
int s2k_group_schnorr_batch_verify_rlc(s2k_group* g, size_t n, const uint8_t* pk, const uint8_t* msgs, const uint64_t* msg_offsets,
                                       size_t msg_len, const uint8_t* sig, const uint8_t* seed32, int* all_valid) {
  if (!g || !all_valid) return S2K_ERR_ARG;
  *all_valid = 0;
  if (!seed32 || (n && (!pk || !sig))) return gfail(g, S2K_ERR_ARG, "null buffer");
  if (n && !msgs && (msg_offsets ? msg_offsets[n] != 0 : msg_len != 0)) return gfail(g, S2K_ERR_ARG, "null message buffer");
  std::vector<uint8_t> res(g->members.size(), 0);
  shard_job proto;
  proto.kind = 1;
  proto.pub = pk;
  proto.dig = msgs;
  proto.dig_off = msg_offsets;
  proto.msg_len = msg_len;
  proto.s = sig;
  proto.seed32 = seed32;
  proto.result = res.data();
  s2k_ticket t = 0;
  int rc = group_submit(g, n, proto, &t);
  if (rc) return rc;
  rc = s2k_group_wait(g, t);
  if (rc) return rc;
  int ok = 1;
  for (uint8_t v : res) ok &= v;
  *all_valid = ok;
  return S2K_OK;
}

// Multiscalar multiplication: every member sums its contiguous shard of the terms (s2k_multi_scalar_mult), the first member
// adds up the partial sums (one more multiscalar multiplication with unit scalars: point addition with every special case)
int s2k_group_multi_scalar_mult(s2k_group* g, size_t n, const uint8_t* k, const uint8_t* points, uint8_t* out65) {
  if (!g || !out65) return S2K_ERR_ARG;
  if (n && (!k || !points)) return gfail(g, S2K_ERR_ARG, "null buffer");
  const size_t D = g->members.size();
  std::vector<uint8_t> parts(D * 65, 0);
  shard_job proto;
  proto.kind = 2;
  proto.r = k;
  proto.pub = points;
  proto.result = parts.data();
  s2k_ticket t = 0;
  int rc = group_submit(g, n, proto, &t);
  if (rc) return rc;
  rc = s2k_group_wait(g, t);
  if (rc) return rc;
  // the partial sums as points: the all-zero record of an identity becomes the one-byte identity's 65-byte form the entry point takes
  std::vector<uint8_t> ones(D * 32, 0);
  for (size_t i = 0; i < D; ++i) ones[i * 32 + 31] = 1;
  group_wait_idle(g);                              // the members' threads are parked: member 0's context may be used from here
  rc = s2k_multi_scalar_mult(g->members[0]->ctx, D, ones.data(), parts.data(), out65);
  if (rc) return gfail(g, rc, "%s", s2k_last_error(g->members[0]->ctx));
  return S2K_OK;
}

int s2k_group_wait(s2k_group* g, s2k_ticket ticket) {
  if (!g) return S2K_ERR_ARG;
  std::unique_lock<std::mutex> lock(g->m);
  if (ticket == 0 || ticket >= g->next_ticket) {
    snprintf(g->err, sizeof g->err, "s2k_group_wait: ticket %llu was never issued", (unsigned long long)ticket);
    return S2K_ERR_ARG;
  }
  // the entry is looked up again after every wake-up: a submit on another thread may erase finished entries (and with
  // them any iterator) while this thread waits without the lock
  int rc = S2K_OK;
  bool dropped = false;
  g->cv.wait(lock, [&] {
    auto it = g->pend.find(ticket);
    if (it == g->pend.end()) {
      dropped = true;
      return true;
    }
    rc = it->second.rc;
    return it->second.remaining == 0;
  });
  if (dropped) {                                   // long done, its result dropped from the table: a failure is still known
    for (unsigned i = 0; i < 8; ++i)
      if (g->failed_ticket[i] == ticket) {
        snprintf(g->err, sizeof g->err, "ticket %llu failed (code %d)", (unsigned long long)ticket, g->failed_rc[i]);
        return g->failed_rc[i];
      }
    return S2K_OK;
  }
  return rc;
}

int s2k_group_ecdsa_verify_batch(s2k_group* g, size_t n, const uint8_t* pub, const uint8_t* dig, const uint8_t* r, const uint8_t* s,
                                 uint32_t flags, uint8_t* valid) {
  s2k_ticket t = 0;
  const int rc = s2k_group_ecdsa_verify_batch_submit(g, n, pub, dig, r, s, flags, valid, &t);
  if (rc) return rc;
  return s2k_group_wait(g, t);
}

int s2k_group_member_stats(s2k_group* g, double* stats) {
  if (!g || !stats) return S2K_ERR_ARG;
  std::lock_guard<std::mutex> lock(g->m);
  for (size_t i = 0; i < g->members.size(); ++i) {
    const member* me = g->members[i];
    stats[4 * i + 0] = me->st_n;
    stats[4 * i + 1] = me->st_lo;
    stats[4 * i + 2] = me->st_ms;
    stats[4 * i + 3] = (double)me->device;
  }
  return S2K_OK;
}

// stats[m * 8 + 0..7] for member m: the four values of s2k_group_member_stats, then the NUMA node of its device (-1 unknown),
// the CPUs its thread is bound to (0: not bound), and of its last finished shard the milliseconds its host-to-device copies took and
// the milliseconds from the first copy to the verdicts (both on the device's clock; the first call switches the timing on,
// so they are zero until a shard has been submitted after it).
int s2k_group_member_stats_ex(s2k_group* g, double* stats) {
  if (!g || !stats) return S2K_ERR_ARG;
  bool enable = false;
  {
    std::lock_guard<std::mutex> lock(g->m);
    enable = !g->timing;
    g->timing = true;
  }
  if (enable) {
    group_wait_idle(g);                              // the members' threads are parked: their contexts may be touched from here
    for (member* me : g->members) (void)s2k_ctx_ticket_timing(me->ctx, 1);
  }
  std::lock_guard<std::mutex> lock(g->m);
  for (size_t i = 0; i < g->members.size(); ++i) {
    const member* me = g->members[i];
    double* o = stats + 8 * i;
    o[0] = me->st_n;
    o[1] = me->st_lo;
    o[2] = me->st_ms;
    o[3] = (double)me->device;
    o[4] = (double)me->numa_node;
    o[5] = (double)me->bound_cpus;
    o[6] = me->st_h2d_ms;
    o[7] = me->st_dev_ms;
  }
  return S2K_OK;
}

// Blocks until the background build of the wide generator tables has ended on every member's device (s2k_ctx_gt_wait): a
// benchmark, or a service that wants its full rate from the first batch, calls this once after s2k_group_create.  Returns the
// smallest window width in use among the members.
int s2k_group_gt_wait(s2k_group* g) {
  if (!g) return S2K_ERR_ARG;
  int bits = 0;
  for (member* me : g->members) {                    // (touches the per-device table registry only, not the contexts' streams)
    const int b = s2k_ctx_gt_wait(me->ctx);
    if (b > 0 && (bits == 0 || b < bits)) bits = b;
  }
  return bits;
}

// The partition group_submit uses, for callers that lay out their own buffers: member i takes the items
// [i * per, min(n, (i + 1) * per)), per = ceil(n / members) rounded up to 256.
size_t s2k_group_shard_size(const s2k_group* g, size_t n) {
  if (!g || g->members.empty()) return 0;
  const size_t D = g->members.size();
  return ((n + D - 1) / D + 255) & ~(size_t)255;
}

// A page-locked array of n items of bytes_per_item bytes whose pages lie next to the device that will read them: the range of
// every member's shard (s2k_group_shard_size) is placed on the NUMA node of that member's device (mbind where the kernel allows,
// and first touched by a thread bound to the node's CPUs either way), then the whole block is pinned (s2k_host_register).  One
// node, unknown nodes: an ordinary pinned block.  Pages that two shards share go to the lower shard's node.  NULL on failure.
void* s2k_group_host_alloc(s2k_group* g, size_t bytes_per_item, size_t n) {
  if (!g || bytes_per_item == 0 || n == 0 || n > ((size_t)1 << 40) / bytes_per_item) return nullptr;
  const size_t page = (size_t)sysconf(_SC_PAGESIZE);
  const size_t bytes = (n * bytes_per_item + page - 1) / page * page;
  void* p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
  if (p == MAP_FAILED) {
    (void)gfail(g, S2K_ERR_NOMEM, "s2k_group_host_alloc: mmap of %zu bytes failed", bytes);
    return nullptr;
  }
  const size_t D = g->members.size(), per = s2k_group_shard_size(g, n);
  std::vector<std::thread> th;
  for (size_t i = 0; i < D; ++i) {
    const size_t lo = i * per < n ? i * per : n, hi = lo + per < n ? lo + per : n;
    if (hi <= lo) continue;
    // whole pages of the shard's byte range: from the first page that STARTS in it (the page a shard shares with the one below
    // belongs to that one) to the page that holds its last byte
    const size_t b0 = (lo * bytes_per_item + page - 1) / page * page, b1 = i + 1 == D || hi == n ? bytes : (hi * bytes_per_item + page - 1) / page * page;
    const size_t first = i == 0 ? 0 : b0;
    if (b1 <= first) continue;
    const int node = g->members[i]->numa_node;
    uint8_t* base = (uint8_t*)p + first;
    const size_t len = b1 - first;
    th.emplace_back([base, len, node, page] {
      (void)s2k_topology_prefer_node(base, len, node);
      (void)s2k_bind_thread_to_node(node);             // first touch from the node's own CPUs
      for (size_t o = 0; o < len; o += page) base[o] = 0;
    });
  }
  for (std::thread& t : th) t.join();
  if (s2k_host_register(p, bytes) != S2K_OK) {       // (no GPU, or the runtime refused: the caller gets nothing rather than pageable memory)
    (void)gfail(g, S2K_ERR_HIP, "s2k_group_host_alloc: pinning %zu bytes failed", bytes);
    (void)munmap(p, bytes);
    return nullptr;
  }
  std::lock_guard<std::mutex> lock(g->m);
  g->blocks.push_back(host_block{p, bytes});
  return p;
}

void s2k_group_host_free(s2k_group* g, void* p) {
  if (!g || !p) return;
  host_block b;
  {
    std::lock_guard<std::mutex> lock(g->m);
    for (size_t i = 0; i < g->blocks.size(); ++i)
      if (g->blocks[i].p == p) {
        b = g->blocks[i];
        g->blocks.erase(g->blocks.begin() + (long)i);
        break;
      }
  }
  if (!b.p) return;                                  // not one of this group's blocks
  (void)s2k_host_unregister(b.p);
  (void)munmap(b.p, b.bytes);
}

}  // extern "C"
