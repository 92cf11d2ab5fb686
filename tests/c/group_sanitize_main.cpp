// Drives csrc/group.cpp (+ topology.cpp) over tests/c/stub_ctx.cpp under ThreadSanitizer / AddressSanitizer: random
// sequences of submit / wait / wait out of order / a fifth and sixth submit / destroy with work in flight, for every group
// entry point, from one caller thread and from a producer / consumer pair; every shard of every batch is checked.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <deque>
#include <condition_variable>
#include <thread>
#include <vector>

#include "secp256k1_voi_amd.h"

extern "C" int s2k_stub_live_contexts(void);

static uint64_t st = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {
  st ^= st << 13;
  st ^= st >> 7;
  st ^= st << 17;
  return st;
}
static int bad = 0;
#define CHECK(c)                                                 \
  do {                                                           \
    if (!(c)) {                                                  \
      ++bad;                                                     \
      fprintf(stderr, "line %d: %s\n", __LINE__, #c);            \
    }                                                            \
  } while (0)

struct batch {
  size_t n = 0;
  std::vector<uint8_t> pub, dig, r, s, valid;
  std::vector<uint32_t> kidx;
  std::vector<uint64_t> off;
  s2k_ticket t = 0;
  int kind = 0;    // 0 packed, 1 encoded, 2 key set
};

static batch* make_batch(size_t n, int kind) {
  batch* b = new batch();
  b->n = n;
  b->kind = kind;
  b->pub.resize(n * 64 + 64);
  b->dig.resize(n * 32 + 32);
  b->r.resize(n * 32 + 32);
  b->s.resize(n * 32 + 32);
  b->valid.assign(n + 1, 7);
  b->kidx.resize(n + 1);
  b->off.resize(n + 2);
  for (size_t i = 0; i < n; ++i) {
    b->pub[64 * i] = (uint8_t)rnd();
    b->kidx[i] = (uint32_t)rnd();
    b->off[i] = 64 * i;
  }
  b->off[n] = 64 * n;
  return b;
}
static void check_batch(const batch* b) {
  for (size_t i = 0; i < b->n; ++i) {
    const uint8_t e = b->kind == 2 ? (uint8_t)(b->kidx[i] & 1u) : (uint8_t)(b->pub[64 * i] & 1u);
    if (b->valid[i] != e) {
      ++bad;
      fprintf(stderr, "batch of %zu (kind %d): verdict %zu is %u, expected %u\n", b->n, b->kind, i, b->valid[i], e);
      return;
    }
  }
  CHECK(b->valid[b->n] == 7);          // nothing written past the end
}
static int submit(s2k_group* g, s2k_group_keyset* gks, batch* b) {
  if (b->kind == 0)
    return s2k_group_ecdsa_verify_batch_submit(g, b->n, b->pub.data(), b->dig.data(), b->r.data(), b->s.data(), 0, b->valid.data(), &b->t);
  if (b->kind == 1)
    return s2k_group_ecdsa_verify_encoded_batch_submit(g, b->n, b->pub.data(), b->off.data(), b->dig.data(), b->off.data(), b->s.data(),
                                                       b->off.data(), 0, 32, 0, b->valid.data(), &b->t);
  return s2k_group_ecdsa_verify_batch_keyset_submit(g, gks, b->n, b->kidx.data(), b->dig.data(), b->r.data(), b->s.data(), 0, b->valid.data(), &b->t);
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 40;
  // a broken device in the list: creation fails as a whole and leaves nothing behind
  {
    const int devs[3] = {0, 3, 1};
    s2k_group* g = nullptr;
    CHECK(s2k_group_create(devs, 3, &g) != S2K_OK && g == nullptr);
    CHECK(s2k_stub_live_contexts() == 0);
  }
  for (int round = 0; round < rounds; ++round) {
    const size_t D = 1 + rnd() % 4;
    int devs[4];
    for (size_t i = 0; i < D; ++i) devs[i] = (int)(rnd() % 3);
    s2k_group* g = nullptr;
    CHECK(s2k_group_create(devs, D, &g) == S2K_OK && g && s2k_group_size(g) == D);
    if (!g) return 1;
    s2k_group_keyset* gks = nullptr;
    std::vector<uint8_t> keys(64 * 5, 1);
    CHECK(s2k_group_keyset_create(g, 5, keys.data(), 0, &gks) == S2K_OK && s2k_group_keyset_size(gks) == 5);
    std::vector<double> stx(8 * D);
    if (round & 1) CHECK(s2k_group_member_stats_ex(g, stx.data()) == S2K_OK && s2k_group_gt_wait(g) == 26);
    std::vector<batch*> flying, all;
    const int ops = 20 + (int)(rnd() % 30);
    for (int op = 0; op < ops; ++op) {
      const unsigned what = (unsigned)(rnd() % 10);
      if (what < 6) {                                   // submit (sizes around the rounding to 256 and to the member count)
        const size_t sizes[8] = {0, 1, 255, 256, 257, 1000, 4096 + D, (size_t)(rnd() % 3000)};
        batch* b = make_batch(sizes[rnd() % 8], (int)(rnd() % 3));
        CHECK(submit(g, gks, b) == S2K_OK && b->t != 0);
        flying.push_back(b);
        all.push_back(b);
      } else if (what < 9 && !flying.empty()) {         // wait for a random ticket in flight (often not the oldest)
        const size_t i = rnd() % flying.size();
        CHECK(s2k_group_wait(g, flying[i]->t) == S2K_OK);
        check_batch(flying[i]);
        CHECK(s2k_group_wait(g, flying[i]->t) == S2K_OK);   // twice is fine
        flying.erase(flying.begin() + (long)i);
      } else if (what == 9) {                           // the synchronous and the whole-batch forms in between
        batch* b = make_batch(300 + rnd() % 500, 0);
        CHECK(s2k_group_ecdsa_verify_batch(g, b->n, b->pub.data(), b->dig.data(), b->r.data(), b->s.data(), 0, b->valid.data()) == S2K_OK);
        check_batch(b);
        int ok = 0;
        std::vector<uint8_t> sig(64 * b->n, 1), seed(32, 9);
        if (rnd() & 1) sig[64 * (rnd() % b->n)] = 0xFF;
        bool expect = true;
        for (size_t i = 0; i < b->n; ++i) expect = expect && sig[64 * i] != 0xFF;
        CHECK(s2k_group_schnorr_batch_verify_rlc(g, b->n, b->pub.data(), b->dig.data(), nullptr, 32, sig.data(), seed.data(), &ok) == S2K_OK && (ok != 0) == expect);
        delete b;
        CHECK(s2k_group_member_stats(g, stx.data()) == S2K_OK);
      }
    }
    CHECK(s2k_group_wait(g, 0) != S2K_OK);              // never issued
    CHECK(s2k_group_wait(g, 1u << 30) != S2K_OK);
    // a failing shard is reported by the wait on its ticket, and later tickets still work
    {
      batch* b = make_batch(777, 0);
      CHECK(s2k_group_ecdsa_verify_batch_submit(g, b->n, b->pub.data(), b->dig.data(), b->r.data(), b->s.data(), 0x80000000u, b->valid.data(), &b->t) == S2K_OK);
      CHECK(s2k_group_wait(g, b->t) != S2K_OK);
      // push it out of the table of recent tickets: the failure must still be known
      for (int i = 0; i < 20; ++i) {
        batch* c = make_batch(10, 0);
        CHECK(submit(g, gks, c) == S2K_OK && s2k_group_wait(g, c->t) == S2K_OK);
        check_batch(c);
        delete c;
      }
      CHECK(s2k_group_wait(g, b->t) != S2K_OK);
      delete b;
    }
    // producer / consumer: one thread submits, another waits (a cgo shim's Stream does this)
    {
      std::mutex m;
      std::condition_variable cv;
      std::deque<batch*> handoff;
      bool closed = false;
      std::thread consumer([&] {
        for (;;) {
          batch* b = nullptr;
          {
            std::unique_lock<std::mutex> lock(m);
            cv.wait(lock, [&] { return closed || !handoff.empty(); });
            if (handoff.empty()) return;
            b = handoff.front();
            handoff.pop_front();
          }
          if (s2k_group_wait(g, b->t) != S2K_OK) ++bad;
          check_batch(b);
          delete b;
        }
      });
      for (int i = 0; i < 30; ++i) {
        batch* b = make_batch(100 + rnd() % 900, (int)(rnd() % 3));
        CHECK(submit(g, gks, b) == S2K_OK);
        {
          std::lock_guard<std::mutex> lock(m);
          handoff.push_back(b);
        }
        cv.notify_all();
      }
      {
        std::lock_guard<std::mutex> lock(m);
        closed = true;
      }
      cv.notify_all();
      consumer.join();
    }
    if (round % 3 == 0) {                               // orderly end: wait for everything
      for (batch* b : flying) {
        CHECK(s2k_group_wait(g, b->t) == S2K_OK);
        check_batch(b);
      }
      flying.clear();
    }
    if (round % 5 == 0) {                               // host blocks: one freed, one left to the group
      void* p = s2k_group_host_alloc(g, 64, 100000);
      void* q = s2k_group_host_alloc(g, 32, 777);
      CHECK(p && q);
      if (p) memset(p, 1, 64 * 100000);
      s2k_group_host_free(g, p);
      CHECK(s2k_group_shard_size(g, 1000) % 256 == 0 && s2k_group_shard_size(g, 1000) * D >= 1000);
    }
    s2k_group_keyset_destroy(gks);
    s2k_group_destroy(g);                               // with work in flight in two rounds of three: the queued shards are verified first
    for (batch* b : flying) check_batch(b);
    for (batch* b : all) delete b;
    CHECK(s2k_stub_live_contexts() == 0);
  }
  printf("%s\n", bad ? "FAILED" : "ok");
  return bad != 0;
}
