"""MI355X-native distributed LLM inference stack (imported as ``dli_amd``).

Layer map:
  csrc/      HIP kernels for gfx950 + C++ host runtime (block manager, batch builder)
  ops/       python entry points of the kernels (+ fp32 torch references used on CPU and in tests)
  parallel/  process groups (RCCL / gloo), TP/EP layers, custom xGMI all-reduce
  runtime/   paged KV cache, block manager binding, hipGraph decode runner, model runner, TP workers
  models/    OPT, Llama-3, Mixtral + weight loading / random init
  engine/    requests, continuous-batching + chunked-prefill scheduler, sampler glue, tokenizer, metrics
  server/    aiohttp front end: /api/generate (Ollama NDJSON), /v1/completions (OpenAI SSE), /metrics
  router/    data-parallel replica router
"""
__version__ = "0.1.0"
