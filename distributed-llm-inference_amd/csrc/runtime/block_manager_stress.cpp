// Sanitizer stress driver for the block manager (ASan/UBSan builds of the C++ allocator).
// Built and run by tests/test_sanitizers.py with -fsanitize=address,undefined (host code only).
// Random mix of admit / grow / fork / copy-on-write / commit / free with invariants checked each round:
//   * every page is either free, evictable-cached, or referenced; ref counts equal the table multiplicity;
//   * build_batch slots always land inside pages owned by the sequence.
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>

#include "block_manager.h"

using dli_rt::BlockManager;

#define CHECK(c)                                                             \
  do {                                                                       \
    if (!(c)) {                                                              \
      std::fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #c); \
      std::abort();                                                          \
    }                                                                        \
  } while (0)

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? std::atoi(argv[1]) : 20000;
  const int NB = 96, BS = 32;
  BlockManager bm(NB, BS, true);
  std::mt19937 rng(1234);
  struct S { std::vector<int32_t> toks; int computed; };
  std::map<int64_t, S> live;
  int64_t next_id = 1;
  std::vector<std::vector<int32_t>> corpus;     // shared prefixes → prefix-cache hits
  for (int i = 0; i < 8; ++i) {
    std::vector<int32_t> p(64 + 32 * (i % 4));
    for (auto& t : p) t = static_cast<int32_t>(rng() % 1000);
    corpus.push_back(p);
  }
  for (int r = 0; r < rounds; ++r) {
    const int op = rng() % 10;
    if (op < 3 || live.empty()) {                                   // admit
      S s;
      s.toks = corpus[rng() % corpus.size()];
      const int extra = rng() % 70;
      for (int i = 0; i < extra; ++i) s.toks.push_back(static_cast<int32_t>(rng() % 1000));
      const int64_t id = next_id++;
      const int cached = bm.match_prefix(id, s.toks.data(), static_cast<int>(s.toks.size()));
      CHECK(cached % BS == 0 && cached < static_cast<int>(s.toks.size()));
      if (!bm.allocate(id, static_cast<int>(s.toks.size()))) { bm.free_seq(id); continue; }
      s.computed = static_cast<int>(s.toks.size());
      bm.commit(id, s.toks.data(), static_cast<int>(s.toks.size()), s.computed);
      live[id] = s;
    } else if (op < 6) {                                            // decode growth of a random sequence
      auto it = live.begin();
      std::advance(it, rng() % live.size());
      S& s = it->second;
      if (bm.num_free() < 2) { bm.free_seq(it->first); live.erase(it); continue; }   // a preemption, in engine terms
      auto cow = bm.cow_tail(it->first, static_cast<int>(s.toks.size()));
      CHECK((cow.first < 0) == (cow.second < 0));
      if (!bm.allocate(it->first, static_cast<int>(s.toks.size()) + 1)) { bm.free_seq(it->first); live.erase(it); continue; }
      s.toks.push_back(static_cast<int32_t>(rng() % 1000));
      int64_t id = it->first;
      int32_t nc = static_cast<int32_t>(s.toks.size()) - 1, nn = 1;
      int32_t pos[4], slot[4], bt[64], sl[1], qs[2];
      const int T = bm.build_batch(1, &id, &nc, &nn, pos, slot, 4, bt, 64, sl, qs);
      CHECK(T == 1 && pos[0] == nc && sl[0] == nc + 1);
      const auto table = bm.block_table(id);
      CHECK(slot[0] / BS == table[nc / BS] && slot[0] % BS == nc % BS);
      bm.commit(id, s.toks.data(), static_cast<int>(s.toks.size()), static_cast<int>(s.toks.size()));
    } else if (op < 7 && bm.num_free() > 8) {                        // fork (shares every page)
      auto it = live.begin();
      std::advance(it, rng() % live.size());
      const int64_t child = next_id++;
      bm.fork(it->first, child);
      live[child] = it->second;
    } else {                                                        // finish
      auto it = live.begin();
      std::advance(it, rng() % live.size());
      bm.free_seq(it->first);
      live.erase(it);
    }
    if (r % 64 == 0) {                                              // global invariants
      std::vector<int> refs(NB, 0);
      for (auto& kv : live)
        for (int b : bm.block_table(kv.first)) { CHECK(b >= 0 && b < NB); ++refs[b]; }
      int used = 0;
      for (int b = 0; b < NB; ++b) { CHECK(bm.ref_count(b) == refs[b]); used += refs[b] > 0; }
      CHECK(used + bm.num_free() == NB);
    }
  }
  for (auto& kv : live) bm.free_seq(kv.first);
  CHECK(bm.num_free() == NB);
  std::printf("stress ok: %d rounds, %lld prefix-cache hits of %lld queries\n", rounds, (long long)bm.cache_hits(),
              (long long)bm.cache_queries());
  return 0;
}
