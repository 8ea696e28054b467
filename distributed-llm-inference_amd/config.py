"""Configuration dataclasses for the serving stack.

The reference has no server-side configuration at all (its only config is the harness dict at
`/root/reference/traffic_generator/main.py:302-313`); SURVEY.md §5.6 asks for a dataclass config
for the new server.  Model shapes are the public ones listed in SURVEY.md §2.5.
"""
from __future__ import annotations

import dataclasses
from dataclasses import dataclass, field
from typing import Optional

KV_BLOCK_SIZE = 32          # tokens per KV page: one MFMA k-step (16x16x32) of the P·V product


@dataclass
class ModelConfig:
    name: str = "llama3-8b"
    arch: str = "llama"                 # "llama" | "mixtral" | "opt"
    vocab_size: int = 128256
    hidden_size: int = 4096
    intermediate_size: int = 14336
    num_layers: int = 32
    num_heads: int = 32
    num_kv_heads: int = 8
    head_dim: int = 128
    max_position: int = 8192
    rope_theta: float = 500000.0
    rms_eps: float = 1e-5
    # mixture of experts
    num_experts: int = 0
    top_k_experts: int = 0
    # OPT specifics
    pos_offset: int = 0                 # learned positions start at this offset (OPT: 2)
    tie_embeddings: bool = False
    dtype: str = "bfloat16"
    bos_token_id: int = 1
    eos_token_id: int = 2

    @property
    def q_per_kv(self) -> int:
        return self.num_heads // self.num_kv_heads

    def param_count(self) -> int:
        h, i, v, l = self.hidden_size, self.intermediate_size, self.vocab_size, self.num_layers
        attn = h * (self.num_heads + 2 * self.num_kv_heads) * self.head_dim + self.num_heads * self.head_dim * h
        if self.arch == "opt":
            mlp = 2 * h * i
        else:
            mlp = 3 * h * i * max(self.num_experts, 1)
        emb = v * h * (1 if self.tie_embeddings else 2)
        return l * (attn + mlp) + emb


MODEL_PRESETS: dict[str, ModelConfig] = {
    "llama3-8b": ModelConfig(),
    "llama3-70b": ModelConfig(name="llama3-70b", hidden_size=8192, intermediate_size=28672, num_layers=80,
                              num_heads=64, num_kv_heads=8),
    "mixtral-8x7b": ModelConfig(name="mixtral-8x7b", arch="mixtral", vocab_size=32000, hidden_size=4096,
                                intermediate_size=14336, num_layers=32, num_heads=32, num_kv_heads=8,
                                rope_theta=1e6, max_position=32768, num_experts=8, top_k_experts=2),
    "opt-125m": ModelConfig(name="opt-125m", arch="opt", vocab_size=50272, hidden_size=768,
                            intermediate_size=3072, num_layers=12, num_heads=12, num_kv_heads=12,
                            head_dim=64, max_position=2048, pos_offset=2, tie_embeddings=True,
                            dtype="float32", bos_token_id=2, eos_token_id=2),
    # ONE tensor-parallel rank's share of the 8-GPU configs as a plain model (kernel benchmarks / single-GPU studies of
    # the shard shapes; `bench.py --emulate-tp 8` gets the same shapes from the real presets by sharding):
    # Llama-3-70B / 8: 8 Q + 1 KV heads, FFN 3584, vocabulary 16032; Mixtral-8x7B / 8 with EP = 8: 4 Q + 1 KV heads and
    # ONE whole expert per rank (the router is approximated by top-1 of 1: every token takes the expert)
    "llama3-70b-tp8shard": ModelConfig(name="llama3-70b-tp8shard", vocab_size=16032, hidden_size=8192,
                                       intermediate_size=3584, num_layers=80, num_heads=8, num_kv_heads=1),
    "mixtral-8x7b-ep8shard": ModelConfig(name="mixtral-8x7b-ep8shard", arch="mixtral", vocab_size=4000, hidden_size=4096,
                                         intermediate_size=14336, num_layers=32, num_heads=4, num_kv_heads=1,
                                         rope_theta=1e6, max_position=32768, num_experts=1, top_k_experts=1),
    # tiny shapes for CPU tests / smoke (same code paths, seconds to run)
    "tiny-llama": ModelConfig(name="tiny-llama", vocab_size=512, hidden_size=128, intermediate_size=256,
                              num_layers=2, num_heads=8, num_kv_heads=2, head_dim=16, max_position=512,
                              dtype="float32"),
    "tiny-llama-d128": ModelConfig(name="tiny-llama-d128", vocab_size=1024, hidden_size=512,
                                   intermediate_size=1024, num_layers=2, num_heads=4, num_kv_heads=1,
                                   head_dim=128, max_position=4096),
    # TP-shardable D=128 shapes for the GPU TP tests (2 KV heads, 4 experts)
    "tiny-llama-tp": ModelConfig(name="tiny-llama-tp", vocab_size=1024, hidden_size=512, intermediate_size=1024,
                                 num_layers=2, num_heads=8, num_kv_heads=2, head_dim=128, max_position=4096),
    # 8 KV heads / 8 experts at D = 128: TP = 8 / EP = 8 rehearsals with 8 ranks sharing one GPU
    "tiny-llama-tp8": ModelConfig(name="tiny-llama-tp8", vocab_size=1024, hidden_size=1024, intermediate_size=2048,
                                  num_layers=2, num_heads=8, num_kv_heads=8, head_dim=128, max_position=4096),
    "tiny-mixtral-tp8": ModelConfig(name="tiny-mixtral-tp8", arch="mixtral", vocab_size=1024, hidden_size=1024,
                                    intermediate_size=2048, num_layers=2, num_heads=8, num_kv_heads=8, head_dim=128,
                                    max_position=4096, num_experts=8, top_k_experts=2),
    "tiny-mixtral-tp": ModelConfig(name="tiny-mixtral-tp", arch="mixtral", vocab_size=1024, hidden_size=512,
                                   intermediate_size=1024, num_layers=2, num_heads=8, num_kv_heads=2, head_dim=128,
                                   max_position=4096, num_experts=4, top_k_experts=2),
    "tiny-mixtral-d128": ModelConfig(name="tiny-mixtral-d128", arch="mixtral", vocab_size=1024, hidden_size=512,
                                     intermediate_size=1024, num_layers=2, num_heads=4, num_kv_heads=1,
                                     head_dim=128, max_position=4096, num_experts=4, top_k_experts=2),
    "tiny-mixtral": ModelConfig(name="tiny-mixtral", arch="mixtral", vocab_size=512, hidden_size=128,
                                intermediate_size=256, num_layers=2, num_heads=8, num_kv_heads=2,
                                head_dim=16, max_position=512, num_experts=4, top_k_experts=2,
                                dtype="float32"),
    # 8 KV heads / 8 experts: the world-size-8 layouts of BASELINE configs 4 and 5 (TP = 8, EP = 8) on CPU
    "tiny-llama-8kv": ModelConfig(name="tiny-llama-8kv", vocab_size=512, hidden_size=128, intermediate_size=256,
                                  num_layers=2, num_heads=16, num_kv_heads=8, head_dim=8, max_position=512,
                                  dtype="float32"),
    "tiny-mixtral-8e": ModelConfig(name="tiny-mixtral-8e", arch="mixtral", vocab_size=512, hidden_size=128,
                                   intermediate_size=256, num_layers=2, num_heads=16, num_kv_heads=8, head_dim=8,
                                   max_position=512, num_experts=8, top_k_experts=2, dtype="float32"),
    "tiny-opt": ModelConfig(name="tiny-opt", arch="opt", vocab_size=512, hidden_size=64,
                            intermediate_size=128, num_layers=2, num_heads=4, num_kv_heads=4,
                            head_dim=16, max_position=512, pos_offset=2, tie_embeddings=True,
                            dtype="float32", bos_token_id=2, eos_token_id=2),
}


def get_model_config(name: str, **overrides) -> ModelConfig:
    if name not in MODEL_PRESETS:
        raise KeyError(f"unknown model preset {name!r}; have {sorted(MODEL_PRESETS)}")
    return dataclasses.replace(MODEL_PRESETS[name], **overrides)


@dataclass
class CacheConfig:
    block_size: int = KV_BLOCK_SIZE
    kv_dtype: str = "auto"              # "auto" (= model dtype) | "fp8" (OCP e4m3, per-layer per-kv-head scales)
    calibrate_fp8_scales: bool = True   # fp8: set the scales from a warm-up prefill (amax / 448); False keeps 1.0
    # fp8: text file whose content (tokenised with the served tokenizer) is the calibration stream instead of random
    # token ids — real prompts put attention sinks / first-token outliers where they occur in service
    calibration_file: Optional[str] = None
    num_blocks: Optional[int] = None    # None → size from free HBM after weights + activation peak
    gpu_memory_utilization: float = 0.92
    enable_prefix_caching: bool = True


@dataclass
class SchedulerConfig:
    max_num_seqs: int = 512                 # concurrent sequences per step
    max_num_batched_tokens: int = 8192      # token budget per step (decode tokens + prefill chunk)
    max_model_len: int = 4096
    long_prefill_chunk: int = 2048          # per-request cap on one step's prefill chunk
    enable_chunked_prefill: bool = True
    # Large steps are trimmed (by shortening one prefill chunk) to a multiple of this many tokens: the GEMMs
    # run on 256-row macro-tiles, and a ragged last tile costs as much as a full one.  0 disables.
    token_alignment: int = 256
    # SLO-aware admission: a waiting request whose time-to-first-token deadline (arrival + ttft_slo seconds) has
    # already passed yields to requests that can still make theirs; it is admitted when none of those is waiting.
    # Under overload FCFS makes EVERY request late (the queue only grows); this keeps goodput at capacity.
    ttft_slo: Optional[float] = None
    ttft_slo_margin: float = 0.3            # fraction of ttft_slo kept for the prefill itself (queueing budget = the rest)
    # Bounded deferral / overload shedding: a request that has waited this long without being admitted is finished
    # with reason "shed" (its client gets an error line instead of hanging for ever); None keeps every request.
    max_queue_delay: Optional[float] = None
    # TPOT-targeting token budget: when set, the share of a step given to prefill shrinks (and admission pauses)
    # while the measured step time would push running decodes past this many seconds per output token.
    tpot_slo: Optional[float] = None
    tpot_slo_margin: float = 0.85           # aim the step time at this fraction of tpot_slo
    min_prefill_tokens: int = 128           # the prefill share never shrinks below this while decodes meet the target
    # TTFT ⇄ TPOT trade (both SLOs set): TPOT is a per-request MEAN, so step time saved under ``tpot_slo`` by earlier steps
    # is credit; it is spent on a longer step when waiting prompts are about to miss their TTFT deadline (a burst drains
    # in fewer steps instead of queueing behind a fixed per-step cap), and repaid by the margin of the steps that follow.
    ttft_tpot_trade: bool = True
    burst_credit_cap: float = 6.0           # most credit held / spent on one step, in units of tpot_slo
    burst_urgent_frac: float = 0.3          # a waiting prompt is urgent once this fraction of ttft_slo has passed


@dataclass
class ParallelConfig:
    tp: int = 1
    dp: int = 1
    ep: int = 1                             # expert-parallel degree (MoE); ep ≤ tp, experts sharded over tp group
    custom_all_reduce: bool = True
    backend: str = "auto"                   # "auto" → nccl(=RCCL) on GPU, gloo on CPU
    # expert parallelism: steps of at least this many tokens dispatch (token, expert) rows by all-to-all instead of
    # evaluating the local experts on replicated activations + all-reduce (0: always replicated)
    ep_a2a_min_tokens: int = 2048


@dataclass
class EngineConfig:
    model: ModelConfig = field(default_factory=ModelConfig)
    cache: CacheConfig = field(default_factory=CacheConfig)
    sched: SchedulerConfig = field(default_factory=SchedulerConfig)
    parallel: ParallelConfig = field(default_factory=ParallelConfig)
    device: str = "auto"                    # "auto" | "cuda" | "cpu"
    seed: int = 0
    use_hip_graphs: bool = True
    # pipelined stepping: queue step N+1 while step N runs (None → on for GPU engines, off on CPU)
    async_scheduling: Optional[bool] = None
    graph_batch_sizes: tuple = (1, 2, 4, 8, 16, 24, 32, 48, 64, 96, 128, 192, 256, 384, 512, 768, 1024, 1280, 1536,
                                2048, 3072, 4096)
    # steps that hold prefill chunks replay captured graphs too (body keyed by the padded token count + LM-head/sampler
    # graph keyed by the padded number of sampled rows); off: only pure-decode steps are captured
    mixed_step_graphs: bool = True
    # time library GEMM vs the own 256² tile (split-K variants) per projection shape and captured row count at start-up
    # and dispatch on the result (ops/gemm_plan.py); off: library above 64 rows, except the gate/up heuristics
    tune_gemms: bool = True
    weights_path: Optional[str] = None      # safetensors dir/file; None → random init
    # "fp8": W8A8 dense projections — weights quantised once to OCP e4m3 with one scale per output channel, activations per
    # token on the fly (csrc/kernels/quant.hip), GEMMs on the fp8 matrix cores (hipBLASLt through torch._scaled_mm) for
    # steps of more than 64 rows; decode-sized steps keep the bf16 weight-streaming kernels (both copies stay resident:
    # 288 GB).  None: bf16 throughout (the default, and what every headline number is measured with).
    quantization: Optional[str] = None
    # Path of a PyTorch TunableOp results file: the vendor-library GEMMs (hipBLASLt / rocBLAS behind torch) are tuned per
    # shape during warm-up (seconds per shape on the first start; the file is written and simply loaded by later starts),
    # and the measured dispatch (ops/gemm_plan.py) then times the TUNED library against the own tile.  None: the library's
    # own heuristics.  Probe on this GPU: −4…6 % on the long-K shapes at M ≈ 2300, −35 % at 256 × 6144 × 4096, else flat.
    library_gemm_tuning: Optional[str] = None
    tokenizer_path: Optional[str] = None    # HF tokenizer.json; None → byte-level tokenizer


@dataclass
class SamplingParams:
    temperature: float = 1.0
    top_p: float = 1.0
    top_k: int = 0                          # 0 → disabled
    max_tokens: int = 128
    ignore_eos: bool = False
    seed: Optional[int] = None
    stop_token_ids: tuple = ()

    def clamp(self, max_model_len: int, prompt_len: int) -> "SamplingParams":
        room = max(max_model_len - prompt_len, 1)
        return dataclasses.replace(self, max_tokens=max(1, min(self.max_tokens, room)))
