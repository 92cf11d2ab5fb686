// A context WITHOUT a GPU, for the sanitizer runs of the threaded host code: csrc/group.cpp and
// csrc/topology.cpp touch the device only through the C-ABI, so they link against these stand-ins and run under
// ThreadSanitizer / AddressSanitizer on the CPU (tests/test_sanitizers_cpu.py, driver: tests/c/group_sanitize_main.cpp).
// A stub context keeps a worker thread - the "device": a submitted ticket completes 0-2 ms later ON THAT THREAD, which is
// also the thread that writes the verdicts (as the DMA engine does), so a caller that reads them before s2k_wait is a
// reported race.  The verdict of an item is a fixed function of its FIRST input byte, so the driver can check every shard
// of every batch.  Four tickets in flight per context, the fifth submit retires the oldest, as engine.hip does.
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <deque>
#include <mutex>
#include <random>
#include <thread>

#include "secp256k1_voi_amd.h"

struct s2k_keyset {
  s2k_ctx* ctx;
  size_t n;
};

struct s2k_ctx {
  int device = 0;
  struct work {
    uint64_t ticket = 0;
    size_t n = 0;
    uint8_t* valid = nullptr;
    const uint8_t* src = nullptr;      // verdict i = src[i * stride] & 1 ... (offsets: src[off[i]])
    size_t stride = 0;
    const uint64_t* off = nullptr;
    const uint32_t* kidx = nullptr;
    std::chrono::system_clock::time_point due;   // (system clock: gcc 11 ThreadSanitizer does not know pthread_cond_clockwait)
    bool done = false;
  };
  std::mutex m;
  std::condition_variable cv;
  std::deque<work> q;                  // in flight, oldest first
  uint64_t next = 1;
  bool stop = false, timing = false;
  std::thread worker;
  std::mt19937 rng{12345};
  char err[128] = {0};
};

static std::atomic<int> g_live_contexts{0};
static thread_local char g_err[128] = "stub";

static void worker_main(s2k_ctx* c) {
  std::unique_lock<std::mutex> lock(c->m);
  for (;;) {
    s2k_ctx::work* w = nullptr;
    for (auto& x : c->q)
      if (!x.done) {
        w = &x;
        break;
      }
    if (!w) {
      if (c->stop) return;
      c->cv.wait(lock);
      continue;
    }
    const auto due = w->due;
    const uint64_t t = w->ticket;
    if (std::chrono::system_clock::now() < due) {
      c->cv.wait_until(lock, due);
      continue;                        // (the deque may have changed: look again)
    }
    // "the device writes the verdicts"
    s2k_ctx::work copy = *w;
    lock.unlock();
    for (size_t i = 0; i < copy.n; ++i) {
      uint8_t b;
      if (copy.kidx) b = (uint8_t)copy.kidx[i];
      else if (copy.off) b = copy.src[copy.off[i]];
      else b = copy.src[i * copy.stride];
      copy.valid[i] = b & 1u;
    }
    lock.lock();
    for (auto& x : c->q)
      if (x.ticket == t) x.done = true;
    c->cv.notify_all();
  }
}

static int submit(s2k_ctx* c, s2k_ctx::work w, s2k_ticket* ticket) {
  std::unique_lock<std::mutex> lock(c->m);
  while (c->q.size() >= 4) {           // the fifth submit retires the oldest
    c->cv.wait(lock, [&] { return c->q.front().done; });
    c->q.pop_front();
  }
  w.ticket = c->next++;
  w.due = std::chrono::system_clock::now() + std::chrono::microseconds(c->rng() % 2000);
  c->q.push_back(w);
  *ticket = w.ticket;
  c->cv.notify_all();
  return S2K_OK;
}

extern "C" {

int s2k_device_count(void) { return 4; }
int s2k_device_pci_bus_id(int device, char* out, size_t len) {
  snprintf(out, len, "0000:%02x:00.0", 0x10 + device);
  return S2K_OK;
}
const char* s2k_last_error(const s2k_ctx* ctx) { return ctx ? ctx->err : g_err; }
int s2k_ctx_create_ex(int device, int gt_bits, uint32_t flags, s2k_ctx** out);
int s2k_ctx_create(int device, s2k_ctx** out) { return s2k_ctx_create_ex(device, 0, 0, out); }
int s2k_ctx_create_ex(int device, int /*gt_bits*/, uint32_t /*flags*/, s2k_ctx** out) {
  if (device == 3) {                   // device 3 of the stub machine always fails: the group's error path
    snprintf(g_err, sizeof g_err, "stub: device 3 is broken");
    return S2K_ERR_HIP;
  }
  s2k_ctx* c = new s2k_ctx();
  c->device = device;
  c->rng.seed(1000u + (unsigned)device);
  c->worker = std::thread(worker_main, c);
  ++g_live_contexts;
  *out = c;
  return S2K_OK;
}
void s2k_ctx_destroy(s2k_ctx* c) {
  if (!c) return;
  {
    std::unique_lock<std::mutex> lock(c->m);
    c->cv.wait(lock, [&] {             // batches still in flight are finished first
      for (auto& x : c->q)
        if (!x.done) return false;
      return true;
    });
    c->stop = true;
  }
  c->cv.notify_all();
  c->worker.join();
  --g_live_contexts;
  delete c;
}
int s2k_stub_live_contexts(void) { return g_live_contexts.load(); }
int s2k_ctx_set_key_grouping(s2k_ctx*, int, uint32_t, uint32_t, uint32_t) { return S2K_OK; }
int s2k_ctx_set_small_batch_max(s2k_ctx*, uint32_t) { return S2K_OK; }
int s2k_ctx_set_mid_batch_max(s2k_ctx*, uint32_t) { return S2K_OK; }
int s2k_ctx_ticket_timing(s2k_ctx* c, int e) {
  c->timing = e != 0;
  return S2K_OK;
}
int s2k_ticket_times(s2k_ctx* c, s2k_ticket, double ms[2]) {
  ms[0] = c->timing ? 0.5 : 0.0;
  ms[1] = c->timing ? 1.5 : 0.0;
  return c->timing ? S2K_OK : S2K_PENDING;
}
int s2k_ctx_gt_wait(s2k_ctx*) { return 26; }
int s2k_host_register(void*, size_t) { return S2K_OK; }
int s2k_host_unregister(void*) { return S2K_OK; }

int s2k_ecdsa_verify_batch_submit(s2k_ctx* c, size_t n, const uint8_t* pub, const uint8_t*, const uint8_t*, const uint8_t*, uint32_t flags,
                                  uint8_t* valid, s2k_ticket* ticket) {
  if (flags & 0x80000000u) {           // the driver's way to make a shard fail
    snprintf(c->err, sizeof c->err, "stub: asked to fail");
    return S2K_ERR_ARG;
  }
  s2k_ctx::work w;
  w.n = n;
  w.valid = valid;
  w.src = pub;
  w.stride = 64;
  return submit(c, w, ticket);
}
int s2k_ecdsa_verify_encoded_batch_submit(s2k_ctx* c, size_t n, const uint8_t* pubs, const uint64_t* pub_off, const uint8_t*, const uint64_t*,
                                          const uint8_t*, const uint64_t*, int, size_t, uint32_t, uint8_t* valid, s2k_ticket* ticket) {
  s2k_ctx::work w;
  w.n = n;
  w.valid = valid;
  w.src = pubs;
  w.off = pub_off;
  return submit(c, w, ticket);
}
int s2k_keyset_create_ex(s2k_ctx* c, size_t n_keys, const uint8_t*, int, s2k_keyset** out) {
  *out = new s2k_keyset{c, n_keys};
  return S2K_OK;
}
void s2k_keyset_destroy(s2k_keyset* ks) { delete ks; }
int s2k_keyset_layout(const s2k_keyset*) { return 1; }
size_t s2k_keyset_device_bytes(const s2k_keyset* ks) { return ks->n * 1024; }
int s2k_ecdsa_verify_batch_keyset_submit(s2k_ctx* c, const s2k_keyset* ks, size_t n, const uint32_t* kidx, const uint8_t*, const uint8_t*,
                                         const uint8_t*, uint32_t, uint8_t* valid, s2k_ticket* ticket) {
  if (ks->ctx != c) return S2K_ERR_ARG;
  s2k_ctx::work w;
  w.n = n;
  w.valid = valid;
  w.kidx = kidx;
  return submit(c, w, ticket);
}
static int wait_ticket(s2k_ctx* c, s2k_ticket t, bool block) {
  std::unique_lock<std::mutex> lock(c->m);
  if (t == 0 || t >= c->next) return S2K_ERR_ARG;
  for (;;) {
    bool found = false, done = false;
    for (auto& x : c->q)
      if (x.ticket == t) {
        found = true;
        done = x.done;
      }
    if (!found) return S2K_OK;         // retired earlier
    if (done) {
      // retire it and everything older that is done (slots free up in order)
      while (!c->q.empty() && c->q.front().done && c->q.front().ticket <= t) c->q.pop_front();
      c->cv.notify_all();
      return S2K_OK;
    }
    if (!block) return S2K_PENDING;
    c->cv.wait(lock);
  }
}
int s2k_wait(s2k_ctx* c, s2k_ticket t) { return wait_ticket(c, t, true); }
int s2k_poll(s2k_ctx* c, s2k_ticket t) { return wait_ticket(c, t, false); }

// whole-batch forms: accept iff no signature's first byte is 0xFF; the multiscalar "sum" is the XOR of the first bytes
int s2k_schnorr_batch_verify_rlc(s2k_ctx*, size_t n, const uint8_t*, const uint8_t*, const uint64_t*, size_t, const uint8_t* sig,
                                 const uint8_t*, int* all_valid) {
  std::this_thread::sleep_for(std::chrono::microseconds(300));
  int ok = 1;
  for (size_t i = 0; i < n; ++i) ok &= sig[64 * i] != 0xFF;
  *all_valid = ok;
  return S2K_OK;
}
int s2k_multi_scalar_mult(s2k_ctx*, size_t n, const uint8_t* k, const uint8_t* points, uint8_t* out) {
  std::this_thread::sleep_for(std::chrono::microseconds(300));
  memset(out, 0, 65);
  uint8_t x = 0;
  for (size_t i = 0; i < n; ++i) x ^= (uint8_t)(k[32 * i + 31] * points[65 * i + 1]);
  out[0] = 0x04;
  out[1] = x;
  return S2K_OK;
}

}  // extern "C"
