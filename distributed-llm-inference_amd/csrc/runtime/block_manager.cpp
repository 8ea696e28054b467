// Paged-KV block manager + per-step batch-metadata builder (host side, C++).
//
// free-list of fixed-size blocks, ref-counts for prefix sharing, copy-on-write.
// The reference has no counterpart (its server was a black-box Ollama box); this is the native
// runtime piece that the scheduler calls on every engine step, so it is kept out of Python:
//   * allocate / free / fork with ref counts,
//   * hash-chained prefix cache (full blocks only) with LRU eviction of unreferenced cached blocks,
//   * copy-on-write of a shared, partially filled tail block,
//   * build_batch(): fills positions / slot_mapping / block_tables / seq_lens / query_start_loc for a
//     whole step straight into (pinned) numpy buffers.
//
// Built with g++ + pybind11 into dli_amd/runtime/_runtime*.so (no GPU dependency, so the same
// object serves the CPU tests).
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include "block_manager.h"

namespace py = pybind11;

namespace {

using dli_rt::BlockManager;
using i32arr = py::array_t<int32_t, py::array::c_style | py::array::forcecast>;
using i64arr = py::array_t<int64_t, py::array::c_style | py::array::forcecast>;
using i32out = py::array_t<int32_t, py::array::c_style>;

int match_prefix(BlockManager& bm, int64_t id, i32arr tokens) { return bm.match_prefix(id, tokens.data(), static_cast<int>(tokens.size())); }
void commit(BlockManager& bm, int64_t id, i32arr tokens, int num_computed) {
  bm.commit(id, tokens.data(), static_cast<int>(tokens.size()), num_computed);
}
int build_batch(BlockManager& bm, i64arr seq_ids, i32arr num_computed, i32arr num_new, i32out positions, i32out slot_mapping,
                i32out block_tables, i32out seq_lens, i32out query_start_loc) {
  const int B = static_cast<int>(seq_ids.size());
  if (num_computed.size() != B || num_new.size() != B) throw std::invalid_argument("build_batch: length mismatch");
  if (block_tables.ndim() != 2 || block_tables.shape(0) < B) throw std::invalid_argument("build_batch: block_tables too small");
  if (seq_lens.size() < B || query_start_loc.size() < B + 1) throw std::invalid_argument("build_batch: seq_lens/query_start_loc too small");
  return bm.build_batch(B, seq_ids.data(), num_computed.data(), num_new.data(), positions.mutable_data(),
                        slot_mapping.mutable_data(), std::min<int64_t>(positions.size(), slot_mapping.size()),
                        block_tables.mutable_data(), static_cast<int>(block_tables.shape(1)), seq_lens.mutable_data(),
                        query_start_loc.mutable_data());
}

}  // namespace

PYBIND11_MODULE(_runtime, m) {
  m.doc() = "dli_amd host runtime: paged-KV block manager and batch builder";
  py::class_<BlockManager>(m, "BlockManager")
      .def(py::init<int, int, bool>(), py::arg("num_blocks"), py::arg("block_size"), py::arg("prefix_caching") = true)
      .def_property_readonly("num_blocks", &BlockManager::num_blocks)
      .def_property_readonly("block_size", &BlockManager::block_size)
      .def_property_readonly("num_free", &BlockManager::num_free)
      .def_property_readonly("num_cached", &BlockManager::num_cached)
      .def_property_readonly("cache_hits", &BlockManager::cache_hits)
      .def_property_readonly("cache_queries", &BlockManager::cache_queries)
      .def("has_seq", &BlockManager::has_seq)
      .def("match_prefix", &match_prefix)
      .def("allocate", &BlockManager::allocate)
      .def("can_allocate", &BlockManager::can_allocate)
      .def("commit", &commit)
      .def("free_seq", &BlockManager::free_seq)
      .def("fork", &BlockManager::fork)
      .def("cow_tail", &BlockManager::cow_tail)
      .def("block_table", &BlockManager::block_table)
      .def("ref_count", &BlockManager::ref_count)
      .def("reset_prefix_cache", &BlockManager::reset_prefix_cache)
      .def("build_batch", &build_batch);
}
