// Paged-KV block manager core (no Python dependency): see block_manager.cpp for the binding and
// block_manager_stress.cpp for the sanitizer stress driver.
#pragma once
#include <algorithm>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

namespace dli_rt {


inline uint64_t mix64(uint64_t x) {  // splitmix64 finaliser
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

inline uint64_t hash_block(uint64_t parent, const int32_t* tok, int n) {
  uint64_t h = mix64(parent ^ 0xA0761D6478BD642Full);
  for (int i = 0; i < n; ++i) h = mix64(h ^ (static_cast<uint64_t>(static_cast<uint32_t>(tok[i])) + (uint64_t(i) << 32)));
  return h ? h : 1;  // 0 is reserved for "no hash"
}

struct Block {
  int ref = 0;
  uint64_t hash = 0;  // 0 → not in the prefix cache
  int lru_prev = -1, lru_next = -1;
  bool in_lru = false;
};

struct Seq {
  std::vector<int> blocks;
  int hashed_blocks = 0;    // leading blocks whose chain hash has been computed
  uint64_t chain = 0;       // hash of block hashed_blocks-1 (0 at the root)
  bool no_publish = false;  // chain lost (copy-on-write below the hashed prefix): commit() leaves this sequence alone
};

class BlockManager {
 public:
  BlockManager(int num_blocks, int block_size, bool prefix_caching)
      : bs_(block_size), caching_(prefix_caching), blocks_(num_blocks) {
    if (num_blocks <= 0 || block_size <= 0) throw std::invalid_argument("num_blocks and block_size must be > 0");
    free_.reserve(num_blocks);
    for (int i = num_blocks - 1; i >= 0; --i) free_.push_back(i);  // pop_back hands out 0,1,2,…
  }

  int num_blocks() const { return static_cast<int>(blocks_.size()); }
  int block_size() const { return bs_; }
  int num_free() const { return static_cast<int>(free_.size()) + lru_size_; }
  int num_cached() const { return static_cast<int>(cache_.size()); }
  bool has_seq(int64_t id) const { return seqs_.count(id) != 0; }
  int64_t cache_hits() const { return hits_; }
  int64_t cache_queries() const { return queries_; }

  // Attach the longest cached prefix of `tokens` (whole blocks, and never the entire prompt: the last
  // token must be recomputed to produce logits).  Returns the number of tokens covered.
  int match_prefix(int64_t id, const int32_t* tok, int n) {
    Seq& s = seqs_[id];
    if (!s.blocks.empty()) throw std::runtime_error("match_prefix on a sequence that already owns blocks");
    if (!caching_) return 0;
    const int max_blocks = (n - 1) / bs_;
    uint64_t chain = 0;
    for (int b = 0; b < max_blocks; ++b) {
      uint64_t h = hash_block(chain, tok + size_t(b) * bs_, bs_);
      ++queries_;
      auto it = cache_.find(h);
      if (it == cache_.end()) break;
      ++hits_;
      take(it->second);
      s.blocks.push_back(it->second);
      chain = h;
    }
    s.hashed_blocks = static_cast<int>(s.blocks.size());
    s.chain = chain;
    return s.hashed_blocks * bs_;
  }

  // Grow the block table so that `total_tokens` fit.  All-or-nothing.
  bool allocate(int64_t id, int total_tokens) {
    Seq& s = seqs_[id];
    const int need = (total_tokens + bs_ - 1) / bs_ - static_cast<int>(s.blocks.size());
    if (need <= 0) return true;
    if (need > num_free()) return false;
    for (int i = 0; i < need; ++i) s.blocks.push_back(pop_free());
    return true;
  }

  bool can_allocate(int64_t id, int total_tokens) const {
    auto it = seqs_.find(id);
    const int have = it == seqs_.end() ? 0 : static_cast<int>(it->second.blocks.size());
    return (total_tokens + bs_ - 1) / bs_ - have <= num_free();
  }

  // Publish the hashes of blocks that became full (tokens[0:num_computed] are in the cache).
  void commit(int64_t id, const int32_t* tok, int n, int num_computed) {
    if (!caching_) return;
    Seq& s = at(id);
    if (s.no_publish) return;
    if (num_computed > n) num_computed = n;
    const int full = std::min(num_computed / bs_, static_cast<int>(s.blocks.size()));
    for (int b = s.hashed_blocks; b < full; ++b) {
      uint64_t h = hash_block(s.chain, tok + size_t(b) * bs_, bs_);
      Block& blk = blocks_[s.blocks[b]];
      if (blk.hash == 0 && cache_.find(h) == cache_.end()) {
        blk.hash = h;
        cache_.emplace(h, s.blocks[b]);
      }
      s.chain = h;
      s.hashed_blocks = b + 1;
    }
  }

  void free_seq(int64_t id) {
    auto it = seqs_.find(id);
    if (it == seqs_.end()) return;
    auto& v = it->second.blocks;
    for (auto r = v.rbegin(); r != v.rend(); ++r) release(*r);  // tail first → tail is evicted first
    seqs_.erase(it);
  }

  // Child shares every block of the parent (n>1 sampling / beam style forks).
  void fork(int64_t parent, int64_t child) {
    const Seq& p = at(parent);
    if (seqs_.count(child)) throw std::runtime_error("fork: child exists");
    Seq c = p;
    for (int b : c.blocks) ++blocks_[b].ref;
    seqs_.emplace(child, std::move(c));
  }

  // Before writing token number `num_tokens_in_seq` (0-based position) of this sequence: if the block it falls into
  // is shared, give the sequence a private one.  Returns (src, dst) for the device-side block copy; src = -1 when
  // the block is replaced but nothing has to be copied (the position is the first of its block: no kept token
  // lives there); (-1, -1) when the block is already private or not allocated yet.
  std::pair<int, int> cow_tail(int64_t id, int num_tokens_in_seq) {
    Seq& s = at(id);
    if (s.blocks.empty()) return {-1, -1};
    const int idx = num_tokens_in_seq / bs_;
    if (idx >= static_cast<int>(s.blocks.size())) return {-1, -1};
    const int src = s.blocks[idx];
    if (blocks_[src].ref <= 1) return {-1, -1};
    if (num_free() < 1) throw std::runtime_error("cow_tail: out of blocks");
    const int dst = pop_free();
    --blocks_[src].ref;
    s.blocks[idx] = dst;
    if (s.hashed_blocks > idx) {                             // the private block is not the cached one any more
      s.hashed_blocks = idx;
      s.chain = 0;
      // chain of block idx-1 is not stored per block: stop publishing for this sequence past the fork point
      s.no_publish = true;
    }
    return {num_tokens_in_seq % bs_ == 0 ? -1 : src, dst};
  }

  std::vector<int> block_table(int64_t id) const { return at(id).blocks; }
  int ref_count(int block) const { return blocks_.at(block).ref; }

  // Drop every unreferenced cached block (used by tests and on model reload).
  void reset_prefix_cache() {
    while (lru_head_ != -1) {
      int b = lru_head_;
      lru_remove(b);
      cache_.erase(blocks_[b].hash);
      blocks_[b].hash = 0;
      free_.push_back(b);
    }
  }

  // Fill one step's flattened metadata.  Sequence i contributes num_new[i] tokens at positions
  // num_computed[i] … num_computed[i]+num_new[i]-1.  Returns the total token count.
  int build_batch(int B, const int64_t* seq_ids, const int32_t* num_computed, const int32_t* num_new, int32_t* pos,
                  int32_t* slot, int64_t token_cap, int32_t* bt, int max_blocks, int32_t* sl, int32_t* qs) {
    int t = 0;
    qs[0] = 0;
    for (int i = 0; i < B; ++i) {
      const Seq& s = at(seq_ids[i]);
      const int c = num_computed[i], n = num_new[i];
      const int total = c + n;
      const int nb = (total + bs_ - 1) / bs_;
      if (nb > static_cast<int>(s.blocks.size())) throw std::runtime_error("build_batch: sequence lacks blocks for its new tokens");
      if (nb > max_blocks) throw std::runtime_error("build_batch: block table row too short");
      if (t + n > token_cap) throw std::runtime_error("build_batch: token buffers too small");
      int32_t* row = bt + size_t(i) * max_blocks;
      for (int b = 0; b < nb; ++b) row[b] = s.blocks[b];
      for (int p = c; p < total; ++p, ++t) {
        pos[t] = p;
        slot[t] = s.blocks[p / bs_] * bs_ + p % bs_;
      }
      sl[i] = total;
      qs[i + 1] = t;
    }
    return t;
  }

 private:
  Seq& at(int64_t id) {
    auto it = seqs_.find(id);
    if (it == seqs_.end()) throw std::out_of_range("unknown sequence id " + std::to_string(id));
    return it->second;
  }
  const Seq& at(int64_t id) const {
    auto it = seqs_.find(id);
    if (it == seqs_.end()) throw std::out_of_range("unknown sequence id " + std::to_string(id));
    return it->second;
  }

  void lru_push_back(int b) {
    Block& k = blocks_[b];
    k.lru_prev = lru_tail_;
    k.lru_next = -1;
    if (lru_tail_ != -1) blocks_[lru_tail_].lru_next = b; else lru_head_ = b;
    lru_tail_ = b;
    k.in_lru = true;
    ++lru_size_;
  }
  void lru_remove(int b) {
    Block& k = blocks_[b];
    if (k.lru_prev != -1) blocks_[k.lru_prev].lru_next = k.lru_next; else lru_head_ = k.lru_next;
    if (k.lru_next != -1) blocks_[k.lru_next].lru_prev = k.lru_prev; else lru_tail_ = k.lru_prev;
    k.lru_prev = k.lru_next = -1;
    k.in_lru = false;
    --lru_size_;
  }
  void take(int b) {  // reference a cached block
    Block& k = blocks_[b];
    if (k.in_lru) lru_remove(b);
    ++k.ref;
  }
  int pop_free() {
    int b;
    if (!free_.empty()) {
      b = free_.back();
      free_.pop_back();
    } else {  // evict the least recently released cached block
      b = lru_head_;
      if (b == -1) throw std::runtime_error("pop_free: no block available");
      lru_remove(b);
      cache_.erase(blocks_[b].hash);
      blocks_[b].hash = 0;
    }
    blocks_[b].ref = 1;
    return b;
  }
  void release(int b) {
    Block& k = blocks_[b];
    if (k.ref <= 0) throw std::runtime_error("release: block is not referenced");
    if (--k.ref > 0) return;
    if (k.hash != 0) lru_push_back(b); else free_.push_back(b);
  }

  int bs_;
  bool caching_;
  std::vector<Block> blocks_;
  std::vector<int> free_;
  std::unordered_map<uint64_t, int> cache_;
  std::unordered_map<int64_t, Seq> seqs_;
  int lru_head_ = -1, lru_tail_ = -1, lru_size_ = 0;
  int64_t hits_ = 0, queries_ = 0;
};

}  // namespace dli_rt
