// Host-side launchers of the gfx950 kernels (plain HIP, no torch types: the kernel translation
// units never include torch headers, so each compiles in seconds; torch_bindings.cpp adapts them).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dli {

void launch_rms_norm(void* out, const void* x, const void* w, float eps, int rows, int H, int64_t x_stride,
                     int64_t out_stride, hipStream_t stream);
void launch_fused_add_rms_norm(void* x, void* residual, const void* w, float eps, int rows, int H, int64_t x_stride,
                               hipStream_t stream);
void launch_silu_and_mul(void* out, const void* x, int rows, int inter, int64_t x_stride, hipStream_t stream);
void launch_rms_norm_quant(const void* x, void* residual, const void* w, float eps, int rows, int H, int64_t x_stride, void* q,
                           float* scales, hipStream_t stream);
void launch_quant_rows_fp8(bool silu, const void* x, int64_t x_stride, void* q, int64_t q_stride, float* scales, int rows,
                           int K, hipStream_t stream);

void launch_rope_and_cache(void* qkv, int64_t row_stride, const int* positions, const float* cos_sin,
                           const int* slot_mapping, void* k_cache, void* v_cache, int T, int Hq, int Hkv,
                           bool kv_fp8, float k_scale, float v_scale, const float* kv_scales, hipStream_t stream);
void launch_copy_pages(void* cache, const int* pairs, int n_pairs, int64_t page_bytes, hipStream_t stream);

void launch_paged_decode(const void* q, int64_t q_stride, const void* k_cache, const void* v_cache,
                         const int* block_tables, int max_blocks, const int* seq_lens, void* out, int64_t out_stride,
                         float* part_o, float* part_ml, int num_seqs, int Hq, int Hkv, int num_splits, float scale,
                         bool kv_fp8, float k_scale, float v_scale, const float* kv_scales, hipStream_t stream);
void launch_paged_decode_worklist(const void* q, int64_t q_stride, const void* k_cache, const void* v_cache,
                                  const int* block_tables, int max_blocks, const int* seq_lens, void* out,
                                  int64_t out_stride, float* part_o, float* part_ml, const int* work, const int* red,
                                  const int* counts, int items, int reds, int Hq, int Hkv, float scale, bool kv_fp8,
                                  float k_scale, float v_scale, const float* kv_scales, hipStream_t stream);
void launch_paged_prefill(const void* q, int64_t q_stride, const void* k_cache, const void* v_cache,
                          const int* block_tables, int max_blocks, const int* seq_lens, const int* query_start_loc,
                          const int* tile_seq, const int* tile_q0, int num_tiles, void* out, int64_t out_stride, int Hq,
                          int Hkv, float scale, bool kv_fp8, float k_scale, float v_scale, const float* kv_scales,
                          const int* counts, hipStream_t stream);

void launch_skinny_linear(void* out, int64_t out_stride, const void* x, int64_t x_stride, const void* w, const void* bias,
                          int M, int N, int K, bool swiglu, const float* w_scale, hipStream_t stream);
void launch_moe_skinny(bool down, const void* a, int64_t a_stride, const void* w, int64_t w_expert_stride,
                       const int* sorted_ids, const int* tile_expert, const int* meta, const float* topk_w, int top_k,
                       void* out, int64_t out_stride, int K, int N, int max_tiles, int max_rows, hipStream_t stream);
void launch_gather_sampled(int* input, const int* src, const int64_t* sampled, int n, hipStream_t stream);
void launch_sample(const void* logits, bool logits_fp32, int64_t row_stride, int B, int V, const float* temperature,
                   const float* top_p, const int* top_k, const int64_t* seeds, const int64_t* steps, const float* noise,
                   int64_t* out, hipStream_t stream);

// mixture of experts (moe.hip)
void launch_moe_route(const void* logits, int T, int E, int K, int* ids, float* weights, hipStream_t stream);
void launch_moe_align(const int* ids, int n, int expert_offset, int E_local, int* sorted_ids, int* tile_expert, int* meta,
                      int capacity_rows, int tile_rows, hipStream_t stream);
void launch_grouped_gemm(int epilogue, const void* A, int64_t a_stride, const void* W, int64_t w_expert_stride,
                         const int* sorted_ids, const int* tile_expert, const int* meta, const float* topk_w, int top_k,
                         void* out, int64_t out_stride, int K, int N_out, int max_tiles, int k_splits, int64_t slab_elems,
                         hipStream_t stream);
// 256×256×64 tile (gemm256.hip): large-M grouped GEMM and dense GEMM with optional SwiGLU epilogue
void launch_grouped_gemm256(int epilogue, const void* A, int64_t a_stride, const void* W, int64_t w_expert_stride,
                            const int* sorted_ids, const int* tile_expert, const int* meta, const float* topk_w, int top_k,
                            void* out, int64_t out_stride, int K, int N_out, int max_tiles, int k_splits, int64_t slab_elems,
                            hipStream_t stream);
void launch_dense_gemm256(bool swiglu, const void* x, int64_t x_stride, const void* W, void* out, int64_t out_stride, int M,
                          int K, int N_out, int k_splits, float* workspace, bool split_tail, hipStream_t stream);
void dense_gemm256_tail_plan(int M, int K, int N_out, bool swiglu, int* tail_tiles, int* tail_slices);
void launch_moe_sum(const void* y, bool y_fp32, int n_slabs, int64_t slab_elems, const int* ids, int expert_offset, int E_local,
                    int top_k, int T, int H, void* out, hipStream_t stream);

// custom all-reduce over IPC-mapped peer buffers (custom_all_reduce.hip)
void* ar_alloc(size_t bytes, bool uncached);
void ar_free(void* p);
int ar_ipc_handle(void* p, void* handle_out64);
void* ar_ipc_open(const void* handle64);
size_t ar_signal_bytes();
void* ar_create(int rank, int world, const int64_t* staging, const int64_t* result, const int64_t* signal, size_t max_bytes);
void ar_destroy(void* st);
int ar_ipc_close(void* p);
int ar_max_blocks();
int ar_set_tuning(void* st, int n, const int64_t* thr, const int* algo, const int* blocks, int default_blocks,
                  int64_t one_shot_max_bytes);
int ar_error(void* st);
int ar_error_async(void* st, void* pinned_dst4, hipStream_t stream);
int launch_all_reduce(void* st, const void* in, void* out, int64_t bytes, int algo, int blocks, hipStream_t stream);
int launch_all_reduce_add_norm(void* st, void* residual, const void* weight, float eps, void* out, int T, int H, int blocks,
                               hipStream_t stream);
int launch_all_gather(void* st, const void* in, void* out, int64_t bytes, hipStream_t stream);
int launch_broadcast(void* st, void* buf, int64_t bytes, int root, hipStream_t stream);
// expert-parallel dispatch / combine over the same IPC buffers (custom_all_reduce.hip)
int64_t ep_layout(void* st, int64_t n_cap, int H, int64_t* ids_off, int64_t* wts_off, int64_t* meta_off);
void* ep_result_ptr(void* st);
void* ar_staging_ptr(void* st);
int64_t ep_max_bytes(void* st);
int launch_ep_dispatch(void* st, const void* x, int64_t x_stride, const int* ids, const float* wts, int n, int n_cap, int top_k,
                       int H, int local_experts, int* slot, void* recv_x, int* ids_local, float* w_local, int r_cap,
                       int* recv_total, int blocks, hipStream_t stream);
int launch_ep_combine(void* st, const int* ids, const int* slot, int n, int n_cap, int H, int local_experts, void* y_back, int blocks,
                      hipStream_t stream);

}  // namespace dli
