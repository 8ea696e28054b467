// torch.ops.dli.* — argument checking + stream plumbing around the HIP launchers in kernels/.
// Every op launches on the *current* HIP stream, so the ops are capturable into hipGraphs
// (torch.cuda.graph) without any host synchronisation (cdna guide §6 G9).
#include <algorithm>
#include <cstdlib>
#include <ATen/ATen.h>
#include <c10/hip/HIPStream.h>
#include <torch/library.h>

#include "kernels/kernels.h"

namespace {

using at::Tensor;

hipStream_t cur_stream(const Tensor& t) { return c10::hip::getCurrentHIPStream(t.device().index()).stream(); }

void check_bf16_rows(const Tensor& t, const char* name) {
  TORCH_CHECK(t.is_cuda(), name, " must be a GPU tensor");
  TORCH_CHECK(t.scalar_type() == at::kBFloat16, name, " must be bfloat16");
  TORCH_CHECK(t.dim() == 2 && t.stride(1) == 1, name, " must be 2-D with contiguous rows");
  TORCH_CHECK(reinterpret_cast<uintptr_t>(t.data_ptr()) % 16 == 0 && t.stride(0) % 8 == 0, name, " rows must be 16-byte aligned");
}
void check_i32(const Tensor& t, const char* name) {
  TORCH_CHECK(t.is_cuda() && t.scalar_type() == at::kInt && t.is_contiguous(), name, " must be a contiguous int32 GPU tensor");
}
bool is_fp8(const Tensor& t) { return t.scalar_type() == at::kFloat8_e4m3fn; }

Tensor rms_norm(const Tensor& x, const Tensor& weight, double eps) {
  check_bf16_rows(x, "x");
  const int64_t H = x.size(1);
  TORCH_CHECK(H % 8 == 0 && weight.numel() == H && weight.scalar_type() == at::kBFloat16 && weight.is_contiguous());
  Tensor out = at::empty({x.size(0), H}, x.options());
  dli::launch_rms_norm(out.data_ptr(), x.data_ptr(), weight.data_ptr(), (float)eps, (int)x.size(0), (int)H, x.stride(0), H,
                       cur_stream(x));
  return out;
}

void fused_add_rms_norm(Tensor x, Tensor residual, const Tensor& weight, double eps) {
  check_bf16_rows(x, "x");
  check_bf16_rows(residual, "residual");
  const int64_t H = x.size(1);
  TORCH_CHECK(residual.is_contiguous() && residual.sizes() == x.sizes(), "residual must be contiguous and shaped like x");
  TORCH_CHECK(H % 8 == 0 && weight.numel() == H && weight.scalar_type() == at::kBFloat16 && weight.is_contiguous());
  dli::launch_fused_add_rms_norm(x.data_ptr(), residual.data_ptr(), weight.data_ptr(), (float)eps, (int)x.size(0), (int)H,
                                 x.stride(0), cur_stream(x));
}

Tensor silu_and_mul(const Tensor& x) {
  check_bf16_rows(x, "x");
  const int64_t I = x.size(1) / 2;
  TORCH_CHECK(x.size(1) % 16 == 0, "silu_and_mul: row width must be a multiple of 16");
  Tensor out = at::empty({x.size(0), I}, x.options());
  dli::launch_silu_and_mul(out.data_ptr(), x.data_ptr(), (int)x.size(0), (int)I, x.stride(0), cur_stream(x));
  return out;
}

// Per-row dynamic e4m3 quantisation of GEMM inputs (kernels/quant.hip): (q [rows, K] float8_e4m3fn, scales [rows, 1] fp32).
// silu: x is [rows, 2·K] = [gate | up]; the quantised row is silu(gate) ⊙ up.
std::tuple<Tensor, Tensor> quant_rows_fp8(const Tensor& x, bool silu) {
  check_bf16_rows(x, "x");
  const int64_t rows = x.size(0), K = silu ? x.size(1) / 2 : x.size(1);
  TORCH_CHECK(K % 16 == 0 && (!silu || x.size(1) == 2 * K), "quant_rows_fp8: row width must be a multiple of 16");
  Tensor q = at::empty({rows, K}, x.options().dtype(at::kFloat8_e4m3fn));
  Tensor scales = at::empty({rows, 1}, x.options().dtype(at::kFloat));
  dli::launch_quant_rows_fp8(silu, x.data_ptr(), x.stride(0), q.data_ptr(), K, scales.data_ptr<float>(), (int)rows, (int)K,
                             cur_stream(x));
  return {q, scales};
}

// (residual += x;) rmsnorm(·)·w quantised per row to e4m3 in one kernel (norm.hip, QUANT): (q [rows, H], scales [rows, 1]).
std::tuple<Tensor, Tensor> rms_norm_quant(const Tensor& x, const c10::optional<Tensor>& residual, const Tensor& weight, double eps) {
  check_bf16_rows(x, "x");
  const int64_t rows = x.size(0), H = x.size(1);
  TORCH_CHECK(H % 16 == 0 && H <= 8192 && weight.numel() == H && weight.scalar_type() == at::kBFloat16 && weight.is_contiguous(),
              "rms_norm_quant: H must be a multiple of 16, at most 8192");
  void* res = nullptr;
  if (residual.has_value()) {
    check_bf16_rows(*residual, "residual");
    TORCH_CHECK(residual->is_contiguous() && residual->sizes() == x.sizes(), "residual must be contiguous and shaped like x");
    res = residual->data_ptr();
  }
  Tensor q = at::empty({rows, H}, x.options().dtype(at::kFloat8_e4m3fn));
  Tensor scales = at::empty({rows, 1}, x.options().dtype(at::kFloat));
  dli::launch_rms_norm_quant(x.data_ptr(), res, weight.data_ptr(), (float)eps, (int)rows, (int)H, x.stride(0), q.data_ptr(),
                             scales.data_ptr<float>(), cur_stream(x));
  return {q, scales};
}

// y = x · Wᵀ (+ bias) for decode-sized x (≤ 16 rows): the weight-streaming kernel in kernels/skinny_gemm.hip
// `out` (optional): write the result there instead of a fresh tensor — the row-parallel layers hand in a view of the custom
// all-reduce's IPC staging buffer, so the reduction reads the GEMM's output in place.
Tensor take_out(const c10::optional<Tensor>& out, int64_t M, int64_t N, const Tensor& like) {
  if (!out.has_value()) return at::empty({M, N}, like.options());
  TORCH_CHECK(out->is_cuda() && out->scalar_type() == at::kBFloat16 && out->dim() == 2 && out->size(0) == M && out->size(1) == N &&
              out->is_contiguous() && reinterpret_cast<uintptr_t>(out->data_ptr()) % 16 == 0, "out must be a contiguous bf16 [", M, ", ", N, "] GPU tensor");
  return *out;
}

// w_scale given: `w` is float8_e4m3fn with one fp32 scale per output channel ([1, N] or [N]) — W8A16 decode of the fp8 option.
Tensor skinny_linear(const Tensor& x, const Tensor& w, const c10::optional<Tensor>& bias, bool swiglu,
                     const c10::optional<Tensor>& out_, const c10::optional<Tensor>& w_scale) {
  check_bf16_rows(x, "x");
  const int64_t M = x.size(0), K = x.size(1), N = w.size(0);
  TORCH_CHECK(M >= 1 && M <= 64, "skinny_linear: 1..64 rows, got ", M);
  const float* ws = nullptr;
  if (w_scale.has_value()) {
    TORCH_CHECK(w.scalar_type() == at::kFloat8_e4m3fn, "skinny_linear: a weight scale goes with float8_e4m3fn weights");
    TORCH_CHECK(w_scale->is_cuda() && w_scale->scalar_type() == at::kFloat && w_scale->is_contiguous() && w_scale->numel() == N,
                "skinny_linear: w_scale must be a contiguous fp32 GPU tensor with one entry per output channel");
    ws = w_scale->data_ptr<float>();
  }
  TORCH_CHECK(w.is_cuda() && (ws != nullptr || w.scalar_type() == at::kBFloat16) && w.dim() == 2 && w.is_contiguous() && w.size(1) == K,
              "skinny_linear: weight must be a contiguous bf16 (or, with w_scale, float8_e4m3fn) [N, K] GPU tensor with K = ", K);
  TORCH_CHECK(K % 128 == 0 && N % 16 == 0 && reinterpret_cast<uintptr_t>(w.data_ptr()) % 16 == 0,
              "skinny_linear: K % 128 == 0 and N % 16 == 0 required, got N=", N, " K=", K);
  const void* b = nullptr;
  if (bias.has_value()) {
    TORCH_CHECK(bias->is_cuda() && bias->scalar_type() == at::kBFloat16 && bias->is_contiguous() && bias->numel() == N);
    b = bias->data_ptr();
  }
  // swiglu: w = [gate; up] (N = 2·I rows), result silu(x·gateᵀ) ⊙ (x·upᵀ) of width I
  TORCH_CHECK(!swiglu || (N % 32 == 0 && b == nullptr), "skinny_linear: swiglu needs N % 32 == 0 and no bias");
  const int64_t n_out = swiglu ? N / 2 : N;
  Tensor out = take_out(out_, M, n_out, x);
  dli::launch_skinny_linear(out.data_ptr(), n_out, x.data_ptr(), x.stride(0), w.data_ptr(), b, (int)M, (int)N, (int)K, swiglu,
                            ws, cur_stream(x));
  return out;
}

void check_cache(const Tensor& k_cache, const Tensor& v_cache, int64_t Hkv) {
  TORCH_CHECK(k_cache.is_cuda() && v_cache.is_cuda() && k_cache.is_contiguous() && v_cache.is_contiguous());
  TORCH_CHECK(k_cache.scalar_type() == v_cache.scalar_type() && (k_cache.scalar_type() == at::kBFloat16 || is_fp8(k_cache)),
              "KV cache must be bfloat16 or float8_e4m3fn");
  TORCH_CHECK(v_cache.dim() == 4 && v_cache.size(1) == Hkv && v_cache.size(2) == 128 && v_cache.size(3) == 32,
              "V cache must be [blocks, Hkv, 128, 32]");
  TORCH_CHECK(k_cache.dim() == 5 && k_cache.size(0) == v_cache.size(0) && k_cache.size(1) == Hkv && k_cache.size(3) == 32 &&
                  k_cache.size(2) * k_cache.size(4) == 128 && k_cache.size(4) * k_cache.element_size() == 64,
              "K cache must be [blocks, Hkv, 128/CH, 32, CH] with 64-byte chunks");
}

// per-(layer, kv head) fp8 scales: float32 [2, Hkv] on the GPU (row 0: K, row 1: V), or absent → the scalar arguments
const float* scales_ptr(const c10::optional<Tensor>& kv_scales, int64_t Hkv) {
  if (!kv_scales.has_value()) return nullptr;
  TORCH_CHECK(kv_scales->is_cuda() && kv_scales->scalar_type() == at::kFloat && kv_scales->is_contiguous() &&
                  kv_scales->numel() == 2 * Hkv, "kv_scales must be a contiguous float32 GPU tensor of 2 × Hkv values");
  return kv_scales->data_ptr<float>();
}

void rope_and_cache(Tensor qkv, const Tensor& positions, const c10::optional<Tensor>& cos_sin, const Tensor& slot_mapping,
                    Tensor k_cache, Tensor v_cache, int64_t Hq, int64_t Hkv, double k_scale, double v_scale,
                    const c10::optional<Tensor>& kv_scales) {
  check_bf16_rows(qkv, "qkv");
  check_i32(positions, "positions");
  check_i32(slot_mapping, "slot_mapping");
  check_cache(k_cache, v_cache, Hkv);
  const int64_t T = qkv.size(0);
  TORCH_CHECK(qkv.size(1) == (Hq + 2 * Hkv) * 128, "qkv row must be (Hq + 2 Hkv) * 128 wide");
  TORCH_CHECK(positions.numel() >= T && slot_mapping.numel() >= T);
  const float* cs = nullptr;
  if (cos_sin.has_value()) {
    TORCH_CHECK(cos_sin->is_cuda() && cos_sin->scalar_type() == at::kFloat && cos_sin->is_contiguous() && cos_sin->size(1) == 128);
    cs = cos_sin->data_ptr<float>();
  }
  dli::launch_rope_and_cache(qkv.data_ptr(), qkv.stride(0), positions.data_ptr<int>(), cs, slot_mapping.data_ptr<int>(),
                             k_cache.data_ptr(), v_cache.data_ptr(), (int)T, (int)Hq, (int)Hkv, is_fp8(k_cache), (float)k_scale,
                             (float)v_scale, scales_ptr(kv_scales, Hkv), cur_stream(qkv));
}

void copy_pages(Tensor cache, const Tensor& pairs) {
  check_i32(pairs, "pairs");
  TORCH_CHECK(cache.is_cuda() && cache.is_contiguous() && pairs.dim() == 2 && pairs.size(1) == 2);
  const int64_t page_bytes = cache.numel() / cache.size(0) * cache.element_size();
  TORCH_CHECK(page_bytes % 16 == 0);
  dli::launch_copy_pages(cache.data_ptr(), pairs.data_ptr<int>(), (int)pairs.size(0), page_bytes, cur_stream(cache));
}

// q: [T, Hq*128] view (row stride arbitrary); the first num_seqs rows are the decode queries.
void paged_attention_decode(Tensor out, const Tensor& q, const Tensor& k_cache, const Tensor& v_cache, const Tensor& block_tables,
                            const Tensor& seq_lens, int64_t num_seqs, int64_t Hq, int64_t Hkv, double scale, int64_t num_splits,
                            const c10::optional<Tensor>& workspace, double k_scale, double v_scale,
                            const c10::optional<Tensor>& kv_scales) {
  check_bf16_rows(q, "q");
  check_bf16_rows(out, "out");
  check_i32(block_tables, "block_tables");
  check_i32(seq_lens, "seq_lens");
  check_cache(k_cache, v_cache, Hkv);
  TORCH_CHECK(Hq % Hkv == 0 && Hq / Hkv <= 16, "decode kernel supports up to 16 query heads per KV head");
  TORCH_CHECK(q.size(0) >= num_seqs && out.size(0) >= num_seqs && q.size(1) == Hq * 128 && out.size(1) == Hq * 128);
  TORCH_CHECK(block_tables.dim() == 2 && block_tables.size(0) >= num_seqs && seq_lens.numel() >= num_seqs);
  TORCH_CHECK(num_splits >= 1 && num_splits <= 64);
  float* part_o = nullptr;
  float* part_ml = nullptr;
  if (num_splits > 1) {
    TORCH_CHECK(workspace.has_value() && workspace->is_cuda() && workspace->scalar_type() == at::kFloat && workspace->is_contiguous());
    const int64_t need = num_seqs * Hq * num_splits * (128 + 2);
    TORCH_CHECK(workspace->numel() >= need, "decode workspace too small: need ", need, " floats");
    part_o = workspace->data_ptr<float>();
    part_ml = part_o + num_seqs * Hq * num_splits * 128;
  }
  dli::launch_paged_decode(q.data_ptr(), q.stride(0), k_cache.data_ptr(), v_cache.data_ptr(), block_tables.data_ptr<int>(),
                           (int)block_tables.size(1), seq_lens.data_ptr<int>(), out.data_ptr(), out.stride(0), part_o, part_ml,
                           (int)num_seqs, (int)Hq, (int)Hkv, (int)num_splits, (float)scale, is_fp8(k_cache), (float)k_scale,
                           (float)v_scale, scales_ptr(kv_scales, Hkv), cur_stream(q));
}

// Work-list decode (ops.build_decode_work): `work` int32 [≥ items, 4] = {sequence, first page, end page, partial slot},
// `red` int32 [≥ reds, 4] = {sequence, first partial, partials, 0}, `counts` int32 [≥ 4] ON THE DEVICE = {valid items,
// valid reduce entries, prefill tiles, -}.  `items` / `reds` are the launched grid extents (≥ the valid counts).
void paged_attention_decode_worklist(Tensor out, const Tensor& q, const Tensor& k_cache, const Tensor& v_cache,
                                     const Tensor& block_tables, const Tensor& seq_lens, const Tensor& work, const Tensor& red,
                                     const Tensor& counts, int64_t items, int64_t reds, int64_t max_parts, int64_t Hq, int64_t Hkv,
                                     double scale, const Tensor& workspace, double k_scale, double v_scale,
                                     const c10::optional<Tensor>& kv_scales) {
  check_bf16_rows(q, "q");
  check_bf16_rows(out, "out");
  check_i32(block_tables, "block_tables");
  check_i32(seq_lens, "seq_lens");
  check_i32(work, "work");
  check_i32(red, "red");
  check_i32(counts, "counts");
  check_cache(k_cache, v_cache, Hkv);
  TORCH_CHECK(Hq % Hkv == 0 && Hq / Hkv <= 16, "decode kernel supports up to 16 query heads per KV head");
  TORCH_CHECK(q.size(1) == Hq * 128 && out.size(1) == Hq * 128 && block_tables.dim() == 2);
  TORCH_CHECK(items >= 0 && items <= 65535 && reds >= 0 && reds <= 65535 && work.numel() >= 4 * items && red.numel() >= 4 * reds &&
              counts.numel() >= 4, "work list / reduce list shorter than the launched grid");
  TORCH_CHECK(reinterpret_cast<uintptr_t>(work.data_ptr()) % 16 == 0 && reinterpret_cast<uintptr_t>(red.data_ptr()) % 16 == 0);
  TORCH_CHECK(workspace.is_cuda() && workspace.scalar_type() == at::kFloat && workspace.is_contiguous());
  TORCH_CHECK(workspace.numel() >= max_parts * Hq * (128 + 2), "decode workspace too small for ", max_parts, " partials");
  float* part_o = workspace.data_ptr<float>();
  float* part_ml = part_o + max_parts * Hq * 128;
  dli::launch_paged_decode_worklist(q.data_ptr(), q.stride(0), k_cache.data_ptr(), v_cache.data_ptr(), block_tables.data_ptr<int>(),
                                    (int)block_tables.size(1), seq_lens.data_ptr<int>(), out.data_ptr(), out.stride(0), part_o,
                                    part_ml, work.data_ptr<int>(), red.data_ptr<int>(), counts.data_ptr<int>(), (int)items,
                                    (int)reds, (int)Hq, (int)Hkv, (float)scale, is_fp8(k_cache), (float)k_scale, (float)v_scale,
                                    scales_ptr(kv_scales, Hkv), cur_stream(q));
}

void paged_attention_prefill(Tensor out, const Tensor& q, const Tensor& k_cache, const Tensor& v_cache, const Tensor& block_tables,
                             const Tensor& seq_lens, const Tensor& query_start_loc, const Tensor& tile_seq, const Tensor& tile_q0,
                             int64_t num_tiles, int64_t Hq, int64_t Hkv, double scale, double k_scale, double v_scale,
                             const c10::optional<Tensor>& kv_scales, const c10::optional<Tensor>& counts) {
  check_bf16_rows(q, "q");
  check_bf16_rows(out, "out");
  check_i32(block_tables, "block_tables");
  check_i32(seq_lens, "seq_lens");
  check_i32(query_start_loc, "query_start_loc");
  check_i32(tile_seq, "tile_seq");
  check_i32(tile_q0, "tile_q0");
  check_cache(k_cache, v_cache, Hkv);
  const int64_t G = Hq / Hkv;
  TORCH_CHECK(Hq % Hkv == 0 && (G == 1 || G == 2 || G == 4 || G == 8), "prefill kernel supports 1, 2, 4 or 8 query heads per KV head");
  TORCH_CHECK(q.size(1) == Hq * 128 && out.size(1) == Hq * 128 && out.size(0) == q.size(0));
  TORCH_CHECK(tile_seq.numel() >= num_tiles && tile_q0.numel() >= num_tiles && block_tables.dim() == 2);
  const int* cnt = nullptr;                 // device-side tile count (counts[2]); num_tiles is then the grid's upper bound
  if (counts.has_value()) {
    check_i32(*counts, "counts");
    TORCH_CHECK(counts->numel() >= 4);
    cnt = counts->data_ptr<int>();
  }
  dli::launch_paged_prefill(q.data_ptr(), q.stride(0), k_cache.data_ptr(), v_cache.data_ptr(), block_tables.data_ptr<int>(),
                            (int)block_tables.size(1), seq_lens.data_ptr<int>(), query_start_loc.data_ptr<int>(),
                            tile_seq.data_ptr<int>(), tile_q0.data_ptr<int>(), (int)num_tiles, out.data_ptr(), out.stride(0),
                            (int)Hq, (int)Hkv, (float)scale, is_fp8(k_cache), (float)k_scale, (float)v_scale,
                            scales_ptr(kv_scales, Hkv), cnt, cur_stream(q));
}

void sample(Tensor out, const Tensor& logits, const Tensor& temperature, const Tensor& top_p, const Tensor& top_k,
            const Tensor& seeds, const Tensor& steps, const c10::optional<Tensor>& noise) {
  TORCH_CHECK(logits.is_cuda() && logits.dim() == 2 && logits.stride(1) == 1);
  TORCH_CHECK(logits.scalar_type() == at::kBFloat16 || logits.scalar_type() == at::kFloat, "logits must be bf16 or fp32");
  const int64_t B = logits.size(0), V = logits.size(1);
  TORCH_CHECK(out.is_cuda() && out.scalar_type() == at::kLong && out.is_contiguous() && out.numel() >= B);
  TORCH_CHECK(temperature.scalar_type() == at::kFloat && top_p.scalar_type() == at::kFloat && top_k.scalar_type() == at::kInt &&
              seeds.scalar_type() == at::kLong && steps.scalar_type() == at::kLong);
  TORCH_CHECK(temperature.numel() >= B && top_p.numel() >= B && top_k.numel() >= B && seeds.numel() >= B && steps.numel() >= B);
  TORCH_CHECK(temperature.is_cuda() && top_p.is_cuda() && top_k.is_cuda() && seeds.is_cuda() && steps.is_cuda());
  const float* nz = nullptr;
  if (noise.has_value()) {
    TORCH_CHECK(noise->is_cuda() && noise->scalar_type() == at::kFloat && noise->is_contiguous() && noise->numel() == B * V);
    nz = noise->data_ptr<float>();
  }
  dli::launch_sample(logits.data_ptr(), logits.scalar_type() == at::kFloat, logits.stride(0), (int)B, (int)V,
                     temperature.data_ptr<float>(), top_p.data_ptr<float>(), top_k.data_ptr<int>(), seeds.data_ptr<int64_t>(),
                     steps.data_ptr<int64_t>(), nz, out.data_ptr<int64_t>(), cur_stream(logits));
}

void gather_sampled(Tensor input, const Tensor& src, const Tensor& sampled) {
  TORCH_CHECK(input.is_cuda() && src.is_cuda() && sampled.is_cuda());
  TORCH_CHECK(input.scalar_type() == at::kInt && src.scalar_type() == at::kInt && sampled.scalar_type() == at::kLong);
  TORCH_CHECK(input.is_contiguous() && src.is_contiguous() && sampled.is_contiguous() && src.numel() >= input.numel());
  // every src entry is a row of the previous step's sampler output: < sampled.numel() by construction of the
  // scheduler (checked on the host in ModelRunner._stage)
  dli::launch_gather_sampled(input.data_ptr<int>(), src.data_ptr<int>(), sampled.data_ptr<int64_t>(), (int)input.numel(),
                             cur_stream(input));
}

// y = x · Wᵀ on the 256×256×64 tile kernel (gemm256.hip); swiglu: W = [gate; up] rows, y = silu(x·gateᵀ) ⊙ (x·upᵀ)
// split_tail (k_splits == 1): the tiles beyond the last whole round of one tile per CU are K-split over the idle CUs
Tensor dense_gemm256(const Tensor& x, const Tensor& w, bool swiglu, int64_t k_splits, const c10::optional<Tensor>& out_,
                     bool split_tail) {
  check_bf16_rows(x, "x");
  const int64_t M = x.size(0), K = x.size(1), N = w.size(0);
  TORCH_CHECK(w.is_cuda() && w.scalar_type() == at::kBFloat16 && w.dim() == 2 && w.is_contiguous() && w.size(1) == K,
              "dense_gemm256: weight must be a contiguous bf16 [N, K] GPU tensor with K = ", K);
  const int64_t n_out = swiglu ? N / 2 : N;
  TORCH_CHECK(K % 64 == 0 && n_out % (swiglu ? 128 : 256) == 0 && N % 2 == 0, "dense_gemm256: unsupported shape N=", N, " K=", K);
  TORCH_CHECK(M >= 1 && M * x.stride(0) * 2 < (int64_t(1) << 32) && N * K * 2 < (int64_t(1) << 32), "dense_gemm256: operand too large");
  TORCH_CHECK(k_splits >= 1 && k_splits <= 32 && k_splits <= K / 64, "dense_gemm256: k_splits must be in [1, min(32, K / 64)]");
  Tensor out = take_out(out_, M, n_out, x);
  Tensor ws;                                                  // fp32 partial slabs of the split-K path
  if (k_splits > 1) ws = at::empty({k_splits, M, N}, x.options().dtype(at::kFloat));
  split_tail = split_tail && k_splits == 1;
  if (split_tail) {
    int r = 0, s = 0;
    dli::dense_gemm256_tail_plan((int)M, (int)K, (int)n_out, swiglu, &r, &s);
    if (s >= 2) ws = at::empty({(int64_t)s * r, 256 * 256}, x.options().dtype(at::kFloat));
    else split_tail = false;
  }
  dli::launch_dense_gemm256(swiglu, x.data_ptr(), x.stride(0), w.data_ptr(), out.data_ptr(), out.stride(0), (int)M, (int)K, (int)n_out,
                            (int)k_splits, ws.defined() ? ws.data_ptr<float>() : nullptr, split_tail, cur_stream(x));
  return out;
}

// row tile of the grouped GEMM: 256 from this many routed rows per expert on (DLI_MOE_TILE=128|256 forces one).
// Measured crossover at Mixtral shapes (profiles/kernels_moe_tile_crossover_r2z.txt): 96 rows per expert → 128-row tiles
// win by 8 %, from 128 rows per expert on the 256-row tile wins by 1.2–1.37 ×.
const int64_t g_moe_tile_force = std::getenv("DLI_MOE_TILE") ? std::atoll(std::getenv("DLI_MOE_TILE")) : 0;
const int64_t g_moe_tile256_min_rows = std::getenv("DLI_MOE_TILE256_MIN_ROWS") ? std::atoll(std::getenv("DLI_MOE_TILE256_MIN_ROWS")) : 128;

const int64_t g_moe_splitk_min_blocks256 = std::getenv("DLI_MOE_SPLITK_MIN_BLOCKS256") ? std::atoll(std::getenv("DLI_MOE_SPLITK_MIN_BLOCKS256")) : 512;

// largest batch routed to the weight-streaming MoE path; DLI_NO_SKINNY_MOE / DLI_SKINNY_MOE_MAX_T: A-B measurements
const int64_t g_moe_skinny_max_t = std::getenv("DLI_NO_SKINNY_MOE") ? 0
                                   : std::getenv("DLI_SKINNY_MOE_MAX_T") ? std::atoll(std::getenv("DLI_SKINNY_MOE_MAX_T")) : 32;

// align → grouped GEMM (SwiGLU epilogue) → grouped GEMM (weighted scatter) → sum over the slots of a token, for (token,
// slot) pairs whose expert ids / gate weights are already on the device (``ids`` < 0 or outside the local range: skipped).
// ``out``: [T, H] bf16 (any device memory: the expert-parallel path hands in its IPC-visible result buffer).
void moe_experts_forward(const Tensor& x, const Tensor& ids, const Tensor& wts, const Tensor& w13, const Tensor& w2, int64_t top_k,
                         int64_t expert_offset, bool big, void* out, hipStream_t stream) {
  const int64_t T = x.size(0), H = x.size(1), El = w13.size(0), I = w13.size(1) / 2;
  const int64_t n = T * top_k;
  auto i32 = x.options().dtype(at::kInt);
  const int64_t TR = big ? 256 : 128;
  const int64_t capacity = (n + TR - 1) / TR * TR + El * TR;
  const int64_t max_tiles = capacity / TR;
  Tensor sorted = at::empty({capacity}, i32), tile_expert = at::empty({max_tiles}, i32), meta = at::empty({2}, i32);
  Tensor act = at::empty({capacity, I}, x.options());
  // Down projection: with few row tiles (decode-sized batches) a (row tile × 128 columns) grid cannot fill 256 CUs and
  // every block would stream a 3.7 MB weight panel through a 16 KB-deep pipeline, so K is cut into slices whose fp32
  // partial slabs are summed (in fixed order) by moe_sum.  The slice count depends on T only: static under hipGraph.
  const int64_t col_blocks2 = H / TR, nk2 = I / 64;
  int64_t k_splits = 1;
  // (512-thread 256² tiles run one per CU: two full rounds of the chip are enough; the 128² tiles run 2–3 per CU)
  const int64_t want_blocks = big ? g_moe_splitk_min_blocks256 : 1024;
  if (max_tiles * col_blocks2 < want_blocks)
    k_splits = std::min<int64_t>({(want_blocks + max_tiles * col_blocks2 - 1) / (max_tiles * col_blocks2), 16, nk2});
  const bool split = k_splits > 1;
  Tensor y = split ? at::empty({k_splits, n, H}, x.options().dtype(at::kFloat)) : at::empty({n, H}, x.options());
  dli::launch_moe_align(ids.data_ptr<int>(), (int)n, (int)expert_offset, (int)El, sorted.data_ptr<int>(),
                        tile_expert.data_ptr<int>(), meta.data_ptr<int>(), (int)capacity, (int)TR, stream);
  auto gg = big ? dli::launch_grouped_gemm256 : dli::launch_grouped_gemm;
  gg(0, x.data_ptr(), H, w13.data_ptr(), 2 * I * H, sorted.data_ptr<int>(), tile_expert.data_ptr<int>(), meta.data_ptr<int>(),
     wts.data_ptr<float>(), (int)top_k, act.data_ptr(), I, (int)H, (int)I, (int)max_tiles, 1, 0, stream);
  gg(split ? 2 : 1, act.data_ptr(), I, w2.data_ptr(), H * I, sorted.data_ptr<int>(), tile_expert.data_ptr<int>(),
     meta.data_ptr<int>(), wts.data_ptr<float>(), (int)top_k, y.data_ptr(), H, (int)I, (int)H, (int)max_tiles, (int)k_splits,
     n * H, stream);
  dli::launch_moe_sum(y.data_ptr(), split, (int)k_splits, n * H, ids.data_ptr<int>(), (int)expert_offset, (int)El, (int)top_k,
                      (int)T, (int)H, out, stream);
}

// ---- MoE: route → align → grouped GEMM (SwiGLU epilogue) → grouped GEMM (weighted scatter) → sum ----
Tensor fused_moe(const Tensor& x, const Tensor& router_logits, const Tensor& w13, const Tensor& w2, int64_t top_k,
                 int64_t expert_offset, int64_t tile_rows, const c10::optional<Tensor>& out_) {
  check_bf16_rows(x, "x");
  TORCH_CHECK(router_logits.is_cuda() && router_logits.scalar_type() == at::kBFloat16 && router_logits.dim() == 2,
              "router_logits must be a 2-D bfloat16 GPU tensor");
  TORCH_CHECK(x.is_contiguous() && router_logits.is_contiguous());
  TORCH_CHECK(w13.is_cuda() && w2.is_cuda() && w13.is_contiguous() && w2.is_contiguous() && w13.scalar_type() == at::kBFloat16 &&
              w2.scalar_type() == at::kBFloat16 && w13.dim() == 3 && w2.dim() == 3);
  const int64_t T = x.size(0), H = x.size(1), E = router_logits.size(1), El = w13.size(0), I = w13.size(1) / 2;
  TORCH_CHECK(router_logits.size(0) == T && w13.size(2) == H && w2.size(0) == El && w2.size(1) == H && w2.size(2) == I);
  TORCH_CHECK(E <= 64 && top_k >= 1 && top_k <= E && H % 128 == 0 && I % 64 == 0, "unsupported MoE shape");
  auto stream = cur_stream(x);
  auto i32 = x.options().dtype(at::kInt);
  const int64_t n = T * top_k;
  // 256-row tiles when the experts see enough rows to fill them (and the shapes divide): gemm256.hip
  bool big = n / E >= g_moe_tile256_min_rows;               // expected rows per expert (local or not) under uniform routing
  if (g_moe_tile_force) big = g_moe_tile_force == 256;
  if (tile_rows) big = tile_rows == 256;                     // A/B measurements in one process
  big = big && I % 128 == 0 && H % 256 == 0 && n * std::max(H, I) * 2 < (int64_t(1) << 32) && 2 * I * H * 2 < (int64_t(1) << 32);
  const int64_t TR = big ? 256 : 128;
  const int64_t capacity = (n + TR - 1) / TR * TR + El * TR;
  const int64_t max_tiles = capacity / TR;
  Tensor ids = at::empty({n}, i32), wts = at::empty({n}, x.options().dtype(at::kFloat));
  Tensor sorted = at::empty({capacity}, i32), tile_expert = at::empty({max_tiles}, i32), meta = at::empty({2}, i32);
  Tensor out = take_out(out_, T, H, x);
  if (T <= std::min<int64_t>(g_moe_skinny_max_t, 128) && I % 128 == 0 && tile_rows == 0) {
    // Decode-sized batch: an expert sees ≤ 16 rows (at T ≤ 32 and 2-of-8 routing: nearly always; more rows cost an
    // extra pass over that expert's weights from cache), so both projections are pure weight streams — the
    // weight-streaming kernel of skinny_gemm.hip per (expert, 16 weight rows), SwiGLU in between.
    const int64_t tiles = std::min<int64_t>(El, n), rows_pad = (T + 15) / 16 * 16;
    Tensor gu = at::empty({tiles * rows_pad, 2 * I}, x.options()), act16 = at::empty({tiles * rows_pad, I}, x.options());
    Tensor y16 = at::empty({n, H}, x.options());
    dli::launch_moe_route(router_logits.data_ptr(), (int)T, (int)E, (int)top_k, ids.data_ptr<int>(), wts.data_ptr<float>(), stream);
    dli::launch_moe_align(ids.data_ptr<int>(), (int)n, (int)expert_offset, (int)El, sorted.data_ptr<int>(),
                          tile_expert.data_ptr<int>(), meta.data_ptr<int>(), (int)capacity, 128, stream);
    dli::launch_moe_skinny(false, x.data_ptr(), H, w13.data_ptr(), 2 * I * H, sorted.data_ptr<int>(), tile_expert.data_ptr<int>(),
                           meta.data_ptr<int>(), wts.data_ptr<float>(), (int)top_k, gu.data_ptr(), 2 * I, (int)H, (int)(2 * I),
                           (int)tiles, (int)T, stream);
    dli::launch_silu_and_mul(act16.data_ptr(), gu.data_ptr(), (int)(tiles * rows_pad), (int)I, 2 * I, stream);
    dli::launch_moe_skinny(true, act16.data_ptr(), I, w2.data_ptr(), H * I, sorted.data_ptr<int>(), tile_expert.data_ptr<int>(),
                           meta.data_ptr<int>(), wts.data_ptr<float>(), (int)top_k, y16.data_ptr(), H, (int)I, (int)H,
                           (int)tiles, (int)T, stream);
    dli::launch_moe_sum(y16.data_ptr(), false, 1, n * H, ids.data_ptr<int>(), (int)expert_offset, (int)El, (int)top_k, (int)T,
                        (int)H, out.data_ptr(), stream);
    return out;
  }
  dli::launch_moe_route(router_logits.data_ptr(), (int)T, (int)E, (int)top_k, ids.data_ptr<int>(), wts.data_ptr<float>(), stream);
  moe_experts_forward(x, ids, wts, w13, w2, top_k, expert_offset, big, out.data_ptr(), stream);
  return out;
}

// ---- custom all-reduce: raw allocations + IPC handles travel as int64 / uint8 tensors -------------
int64_t ar_alloc(int64_t bytes, bool uncached) {
  void* p = dli::ar_alloc((size_t)bytes, uncached);
  TORCH_CHECK(p != nullptr, "ar_alloc: hipMalloc of ", bytes, " bytes failed");
  return reinterpret_cast<int64_t>(p);
}
void ar_free(int64_t ptr) { dli::ar_free(reinterpret_cast<void*>(ptr)); }
Tensor ar_ipc_handle(int64_t ptr) {
  Tensor h = at::zeros({64}, at::TensorOptions().dtype(at::kByte));
  const int rc = dli::ar_ipc_handle(reinterpret_cast<void*>(ptr), h.data_ptr());
  TORCH_CHECK(rc == 0, "hipIpcGetMemHandle failed with code ", rc, " (is HSA_ENABLE_IPC_MODE_LEGACY=0 set?)");
  return h;
}
int64_t ar_ipc_open(const Tensor& handle) {
  TORCH_CHECK(!handle.is_cuda() && handle.scalar_type() == at::kByte && handle.numel() == 64 && handle.is_contiguous());
  void* p = dli::ar_ipc_open(handle.data_ptr());
  TORCH_CHECK(p != nullptr, "hipIpcOpenMemHandle failed");
  return reinterpret_cast<int64_t>(p);
}
int64_t ar_signal_bytes() { return (int64_t)dli::ar_signal_bytes(); }
int64_t ar_create(int64_t rank, int64_t world, const Tensor& staging, const Tensor& result, const Tensor& signal, int64_t max_bytes) {
  for (const Tensor* t : {&staging, &result, &signal})
    TORCH_CHECK(!t->is_cuda() && t->scalar_type() == at::kLong && t->numel() == world && t->is_contiguous());
  void* st = dli::ar_create((int)rank, (int)world, staging.data_ptr<int64_t>(), result.data_ptr<int64_t>(),
                            signal.data_ptr<int64_t>(), (size_t)max_bytes);
  TORCH_CHECK(st != nullptr, "ar_create: world size not supported");
  return reinterpret_cast<int64_t>(st);
}
void ar_destroy(int64_t st) { dli::ar_destroy(reinterpret_cast<void*>(st)); }
int64_t ar_error(int64_t st) { return dli::ar_error(reinterpret_cast<void*>(st)); }
void ar_all_reduce(int64_t st, const Tensor& inp, Tensor out, int64_t algo, int64_t blocks) {
  TORCH_CHECK(inp.is_cuda() && out.is_cuda() && inp.scalar_type() == at::kBFloat16 && out.scalar_type() == at::kBFloat16);
  TORCH_CHECK(inp.is_contiguous() && out.is_contiguous() && inp.numel() == out.numel());
  TORCH_CHECK(reinterpret_cast<uintptr_t>(inp.data_ptr()) % 16 == 0 && reinterpret_cast<uintptr_t>(out.data_ptr()) % 16 == 0);
  const int rc = dli::launch_all_reduce(reinterpret_cast<void*>(st), inp.data_ptr(), out.data_ptr(), inp.numel() * 2, (int)algo,
                                        (int)blocks, cur_stream(inp));
  TORCH_CHECK(rc == 0, "custom all-reduce: message of ", inp.numel() * 2, " bytes not supported");
}

// ---- expert parallelism without the host: route my token slice, dispatch the (token, expert) rows to the
// ranks that own the experts over the IPC buffers, run the local experts on what arrived, pull the outputs back, sum the
// k slots of each token.  Every rank of the group calls this at the same point of its stream with the same n_cap.
//   x_l [n_l, H] (n_l may be 0), router_logits_l [n_l, E]; n_cap: largest n_l · top_k over the ranks of this call.
// Returns [n_l, H].
Tensor ep_moe(int64_t st, const Tensor& x_l, const Tensor& router_logits_l, const Tensor& w13, const Tensor& w2, int64_t top_k,
              int64_t n_cap, int64_t world, int64_t rank) {
  TORCH_CHECK(x_l.is_cuda() && x_l.scalar_type() == at::kBFloat16 && x_l.dim() == 2 && x_l.is_contiguous());
  TORCH_CHECK(router_logits_l.is_cuda() && router_logits_l.scalar_type() == at::kBFloat16 && router_logits_l.dim() == 2 &&
              router_logits_l.is_contiguous() && router_logits_l.size(0) == x_l.size(0));
  TORCH_CHECK(w13.is_cuda() && w2.is_cuda() && w13.is_contiguous() && w2.is_contiguous() && w13.scalar_type() == at::kBFloat16 &&
              w2.scalar_type() == at::kBFloat16 && w13.dim() == 3 && w2.dim() == 3);
  const int64_t n_l = x_l.size(0), H = x_l.size(1), E = router_logits_l.size(1), El = w13.size(0), I = w13.size(1) / 2;
  TORCH_CHECK(El * world == E && E <= 64 && top_k >= 1 && top_k <= E && H % 128 == 0 && I % 64 == 0, "unsupported EP MoE shape");
  TORCH_CHECK(w13.size(2) == H && w2.size(0) == El && w2.size(1) == H && w2.size(2) == I);
  const int64_t n = n_l * top_k, r_cap = world * n_cap;
  TORCH_CHECK(n <= n_cap, "ep_moe: n_cap smaller than this rank's pair count");
  void* state = reinterpret_cast<void*>(st);
  int64_t o1, o2, o3;
  TORCH_CHECK(dli::ep_layout(state, n_cap, (int)H, &o1, &o2, &o3) == 0 && r_cap * H * 2 <= dli::ep_max_bytes(state),
              "ep_moe: ", n_cap, " pairs of ", H, "-wide rows per rank do not fit the IPC buffers (process fewer tokens per call)");
  auto stream = cur_stream(w13);
  auto i32 = w13.options().dtype(at::kInt);
  auto f32 = w13.options().dtype(at::kFloat);
  Tensor ids = at::empty({std::max<int64_t>(n, 1)}, i32), wts = at::empty({std::max<int64_t>(n, 1)}, f32);
  Tensor slot = at::empty({std::max<int64_t>(n, 1)}, i32), recv_total = at::empty({1}, i32);
  Tensor recv_x = at::empty({r_cap, H}, w13.options()), ids_local = at::empty({r_cap}, i32), w_local = at::empty({r_cap}, f32);
  if (n_l > 0)
    dli::launch_moe_route(router_logits_l.data_ptr(), (int)n_l, (int)E, (int)top_k, ids.data_ptr<int>(), wts.data_ptr<float>(), stream);
  const int blocks = 64;
  int rc = dli::launch_ep_dispatch(state, x_l.data_ptr(), H, ids.data_ptr<int>(), wts.data_ptr<float>(), (int)n, (int)n_cap,
                                   (int)top_k, (int)H, (int)El, slot.data_ptr<int>(), recv_x.data_ptr(), ids_local.data_ptr<int>(),
                                   w_local.data_ptr<float>(), (int)r_cap, recv_total.data_ptr<int>(), blocks, stream);
  TORCH_CHECK(rc == 0, "ep_moe: dispatch launch rejected");
  // local experts on the received rows: one "slot" per row, gate weight applied here, output straight into the IPC result buffer.
  // Row tile: the expected rows per local expert under uniform routing (what arrives is ≈ r_cap / world of the capacity)
  bool big = (n_cap * world / std::max<int64_t>(world, 1)) / std::max<int64_t>(El, 1) >= g_moe_tile256_min_rows;
  if (g_moe_tile_force) big = g_moe_tile_force == 256;
  big = big && I % 128 == 0 && H % 256 == 0 && r_cap * std::max(H, I) * 2 < (int64_t(1) << 32) && 2 * I * H * 2 < (int64_t(1) << 32);
  moe_experts_forward(recv_x, ids_local, w_local, w13, w2, 1, rank * El, big, dli::ep_result_ptr(state), stream);
  Tensor y_back = at::empty({std::max<int64_t>(n, 1), H}, w13.options());
  rc = dli::launch_ep_combine(state, ids.data_ptr<int>(), slot.data_ptr<int>(), (int)n, (int)n_cap, (int)H, (int)El, y_back.data_ptr(),
                              blocks, stream);
  TORCH_CHECK(rc == 0, "ep_moe: combine launch rejected");
  Tensor out = at::empty({n_l, H}, w13.options());
  if (n_l > 0)      // every slot counts (weights were applied by the owners): "all experts local" for the sum
    dli::launch_moe_sum(y_back.data_ptr(), false, 1, n * H, ids.data_ptr<int>(), 0, (int)E, (int)top_k, (int)n_l, (int)H,
                        out.data_ptr(), stream);
  return out;
}

// all-reduce(staged partial sums) → residual add (this rank's rows, in place) → RMSNorm → all rows gathered into `out`
void ar_all_reduce_add_norm(int64_t st, const Tensor& staged, Tensor residual, const Tensor& weight, double eps, Tensor out,
                            int64_t blocks) {
  void* state = reinterpret_cast<void*>(st);
  TORCH_CHECK(staged.data_ptr() == dli::ar_staging_ptr(state), "all_reduce_add_norm: the input must be the staging view (ar_staging_tensor)");
  TORCH_CHECK(staged.dim() == 2 && residual.is_cuda() && out.is_cuda() && weight.is_cuda() && residual.scalar_type() == at::kBFloat16 &&
              out.scalar_type() == at::kBFloat16 && weight.scalar_type() == at::kBFloat16 && residual.is_contiguous() &&
              out.is_contiguous() && weight.is_contiguous());
  const int64_t T = staged.size(0), H = staged.size(1);
  TORCH_CHECK(residual.size(0) == T && residual.size(1) == H && out.size(0) == T && out.size(1) == H && weight.numel() == H);
  TORCH_CHECK(reinterpret_cast<uintptr_t>(residual.data_ptr()) % 16 == 0 && reinterpret_cast<uintptr_t>(out.data_ptr()) % 16 == 0);
  const int rc = dli::launch_all_reduce_add_norm(state, residual.data_ptr(), weight.data_ptr(), (float)eps, out.data_ptr(), (int)T, (int)H,
                                                 (int)blocks, cur_stream(residual));
  TORCH_CHECK(rc == 0, "all_reduce_add_norm: [", T, ", ", H, "] not supported");
}

// [rows, cols] bf16 view of THIS rank's IPC staging buffer: a GEMM that writes its output here has "copied its input
// into the staging buffer" already, and the all-reduce kernels skip that pass when they are handed this very pointer.
Tensor ar_staging_tensor(int64_t st, int64_t rows, int64_t cols, const Tensor& like) {
  void* state = reinterpret_cast<void*>(st);
  TORCH_CHECK(rows >= 1 && cols >= 1 && rows * cols * 2 <= dli::ep_max_bytes(state), "ar_staging_tensor: larger than the staging buffer");
  return at::from_blob(dli::ar_staging_ptr(state), {rows, cols}, like.options().dtype(at::kBFloat16));
}

// launch-shape table chosen by the start-up sweep (identical on every rank of the group)
void ar_set_tuning(int64_t st, const Tensor& thr, const Tensor& algo, const Tensor& blocks, int64_t default_blocks,
                   int64_t one_shot_max_bytes) {
  TORCH_CHECK(!thr.is_cuda() && thr.scalar_type() == at::kLong && thr.is_contiguous());
  TORCH_CHECK(!algo.is_cuda() && algo.scalar_type() == at::kInt && algo.is_contiguous() && algo.numel() == thr.numel());
  TORCH_CHECK(!blocks.is_cuda() && blocks.scalar_type() == at::kInt && blocks.is_contiguous() && blocks.numel() == thr.numel());
  const int rc = dli::ar_set_tuning(reinterpret_cast<void*>(st), (int)thr.numel(), thr.data_ptr<int64_t>(), algo.data_ptr<int>(),
                                    blocks.data_ptr<int>(), (int)default_blocks, one_shot_max_bytes);
  TORCH_CHECK(rc == 0, "ar_set_tuning: invalid table");
}

// queue a copy of the kernel's timeout word into `pinned` (int32[1], pinned host memory) on the current stream
void ar_error_async(int64_t st, Tensor pinned) {
  TORCH_CHECK(!pinned.is_cuda() && pinned.is_pinned() && pinned.scalar_type() == at::kInt && pinned.numel() >= 1);
  const auto stream = c10::hip::getCurrentHIPStream().stream();
  const int rc = dli::ar_error_async(reinterpret_cast<void*>(st), pinned.data_ptr(), stream);
  TORCH_CHECK(rc == 0, "ar_error_async: hipMemcpyAsync failed with code ", rc);
}

// in place broadcast of `buf` (any dtype, byte size a multiple of 16) from TP rank `root`
void ar_broadcast(int64_t st, Tensor buf, int64_t root) {
  TORCH_CHECK(buf.is_cuda() && buf.is_contiguous() && reinterpret_cast<uintptr_t>(buf.data_ptr()) % 16 == 0);
  const int64_t bytes = buf.numel() * buf.element_size();
  const int rc = dli::launch_broadcast(reinterpret_cast<void*>(st), buf.data_ptr(), bytes, (int)root, cur_stream(buf));
  TORCH_CHECK(rc == 0, "custom broadcast: message of ", bytes, " bytes not supported");
}
void ar_ipc_close(int64_t ptr) { (void)dli::ar_ipc_close(reinterpret_cast<void*>(ptr)); }
int64_t ar_max_blocks() { return dli::ar_max_blocks(); }

void ar_all_gather(int64_t st, const Tensor& inp, Tensor out) {
  TORCH_CHECK(inp.is_cuda() && out.is_cuda() && inp.scalar_type() == at::kBFloat16 && out.scalar_type() == at::kBFloat16);
  TORCH_CHECK(inp.is_contiguous() && out.is_contiguous() && out.numel() % inp.numel() == 0);
  TORCH_CHECK(reinterpret_cast<uintptr_t>(inp.data_ptr()) % 16 == 0 && reinterpret_cast<uintptr_t>(out.data_ptr()) % 16 == 0);
  const int rc = dli::launch_all_gather(reinterpret_cast<void*>(st), inp.data_ptr(), out.data_ptr(), inp.numel() * 2, cur_stream(inp));
  TORCH_CHECK(rc == 0, "custom all-gather: message of ", inp.numel() * 2, " bytes not supported");
}

}  // namespace

TORCH_LIBRARY(dli, m) {
  m.def("fused_moe(Tensor x, Tensor router_logits, Tensor w13, Tensor w2, int top_k, int expert_offset, int tile_rows=0, Tensor? out=None) -> Tensor",
        &fused_moe);
  m.def("ar_alloc(int bytes, bool uncached) -> int", &ar_alloc);
  m.def("ar_free(int ptr) -> ()", &ar_free);
  m.def("ar_ipc_handle(int ptr) -> Tensor", &ar_ipc_handle);
  m.def("ar_ipc_open(Tensor handle) -> int", &ar_ipc_open);
  m.def("ar_signal_bytes() -> int", &ar_signal_bytes);
  m.def("ar_create(int rank, int world, Tensor staging, Tensor result, Tensor signal, int max_bytes) -> int", &ar_create);
  m.def("ar_destroy(int state) -> ()", &ar_destroy);
  m.def("ar_error(int state) -> int", &ar_error);
  m.def("ar_all_reduce(int state, Tensor inp, Tensor(a!) out, int algo, int blocks=0) -> ()", &ar_all_reduce);
  m.def("ar_set_tuning(int state, Tensor thr, Tensor algo, Tensor blocks, int default_blocks, int one_shot_max_bytes) -> ()",
        &ar_set_tuning);
  m.def("ar_error_async(int state, Tensor(a!) pinned) -> ()", &ar_error_async);
  m.def("ar_broadcast(int state, Tensor(a!) buf, int root) -> ()", &ar_broadcast);
  m.def("ar_ipc_close(int ptr) -> ()", &ar_ipc_close);
  m.def("ar_max_blocks() -> int", &ar_max_blocks);
  m.def("ar_all_gather(int state, Tensor inp, Tensor(a!) out) -> ()", &ar_all_gather);
  m.def("ep_moe(int state, Tensor x_l, Tensor router_logits_l, Tensor w13, Tensor w2, int top_k, int n_cap, int world, int rank) -> Tensor",
        &ep_moe);
  m.def("rms_norm(Tensor x, Tensor weight, float eps) -> Tensor", &rms_norm);
  m.def("fused_add_rms_norm(Tensor(a!) x, Tensor(b!) residual, Tensor weight, float eps) -> ()", &fused_add_rms_norm);
  m.def("silu_and_mul(Tensor x) -> Tensor", &silu_and_mul);
  m.def("quant_rows_fp8(Tensor x, bool silu=False) -> (Tensor, Tensor)", &quant_rows_fp8);
  m.def("rms_norm_quant(Tensor x, Tensor(a!)? residual, Tensor weight, float eps) -> (Tensor, Tensor)", &rms_norm_quant);
  m.def("skinny_linear(Tensor x, Tensor w, Tensor? bias, bool swiglu=False, Tensor? out=None, Tensor? w_scale=None) -> Tensor", &skinny_linear);
  m.def("dense_gemm256(Tensor x, Tensor w, bool swiglu=False, int k_splits=1, Tensor? out=None, bool split_tail=False) -> Tensor", &dense_gemm256);
  m.def("ar_staging_tensor(int state, int rows, int cols, Tensor like) -> Tensor", &ar_staging_tensor);
  m.def("ar_all_reduce_add_norm(int state, Tensor staged, Tensor(a!) residual, Tensor weight, float eps, Tensor(b!) out, int blocks=0) -> ()",
        &ar_all_reduce_add_norm);
  m.def("rope_and_cache(Tensor(a!) qkv, Tensor positions, Tensor? cos_sin, Tensor slot_mapping, Tensor(b!) k_cache, "
        "Tensor(c!) v_cache, int num_heads, int num_kv_heads, float k_scale, float v_scale, Tensor? kv_scales=None) -> ()",
        &rope_and_cache);
  m.def("copy_pages(Tensor(a!) cache, Tensor pairs) -> ()", &copy_pages);
  m.def("paged_attention_decode(Tensor(a!) out, Tensor q, Tensor k_cache, Tensor v_cache, Tensor block_tables, Tensor seq_lens, "
        "int num_seqs, int num_heads, int num_kv_heads, float scale, int num_splits, Tensor? workspace, float k_scale, "
        "float v_scale, Tensor? kv_scales=None) -> ()", &paged_attention_decode);
  m.def("paged_attention_prefill(Tensor(a!) out, Tensor q, Tensor k_cache, Tensor v_cache, Tensor block_tables, Tensor seq_lens, "
        "Tensor query_start_loc, Tensor tile_seq, Tensor tile_q0, int num_tiles, int num_heads, int num_kv_heads, float scale, "
        "float k_scale, float v_scale, Tensor? kv_scales=None, Tensor? counts=None) -> ()", &paged_attention_prefill);
  m.def("paged_attention_decode_worklist(Tensor(a!) out, Tensor q, Tensor k_cache, Tensor v_cache, Tensor block_tables, "
        "Tensor seq_lens, Tensor work, Tensor red, Tensor counts, int items, int reds, int max_parts, int Hq, int Hkv, float scale, "
        "Tensor workspace, float k_scale, float v_scale, Tensor? kv_scales=None) -> ()", &paged_attention_decode_worklist);
  m.def("gather_sampled(Tensor(a!) input, Tensor src, Tensor sampled) -> ()", &gather_sampled);
  m.def("sample(Tensor(a!) out, Tensor logits, Tensor temperature, Tensor top_p, Tensor top_k, Tensor seeds, Tensor steps, "
        "Tensor? noise) -> ()", &sample);
}
