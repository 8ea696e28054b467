"""Request state machine and the per-step batch description handed to the model runner."""
from __future__ import annotations

import enum
import time
from dataclasses import dataclass, field
from typing import Optional

import numpy as np

from ..config import SamplingParams


class RequestStatus(enum.Enum):
    WAITING = "waiting"
    RUNNING = "running"
    PREEMPTED = "preempted"
    FINISHED_STOPPED = "stop"          # EOS / stop token
    FINISHED_LENGTH = "length"         # max_tokens or max_model_len
    FINISHED_ABORTED = "abort"
    FINISHED_SHED = "shed"             # overload: waited longer than max_queue_delay without being admitted

    @property
    def finished(self) -> bool:
        return self.name.startswith("FINISHED")


class Request:
    """One generation request.  ``tokens[:num_tokens]`` = prompt + generated ids; ``num_computed`` of
    them have their K/V in the paged cache (or written by a step already queued on the stream)."""

    __slots__ = ("req_id", "seq_id", "params", "tokens", "num_prompt", "num_tokens", "num_computed", "status",
                 "arrival", "first_scheduled", "first_token_time", "finish_time", "num_cached_prompt",
                 "preemptions", "seed", "client_tag", "last_token_time", "token_times", "pending", "sample_row",
                 "finishing", "n_samples", "parent_id", "sample_index", "forked")

    def __init__(self, req_id: str, seq_id: int, prompt_ids, params: SamplingParams, arrival: Optional[float] = None,
                 client_tag=None):
        n = len(prompt_ids)
        assert n > 0, "empty prompt"
        self.req_id = req_id
        self.seq_id = seq_id
        self.params = params
        self.tokens = np.empty(n + params.max_tokens + 1, dtype=np.int32)
        self.tokens[:n] = prompt_ids
        self.num_prompt = n
        self.num_tokens = n
        self.num_computed = 0
        self.status = RequestStatus.WAITING
        self.arrival = time.perf_counter() if arrival is None else arrival
        self.first_scheduled = None
        self.first_token_time = None
        self.last_token_time = None
        self.finish_time = None
        self.num_cached_prompt = 0
        self.preemptions = 0
        self.seed = params.seed if params.seed is not None else (hash(req_id) & 0x7FFFFFFF)
        self.client_tag = client_tag
        self.token_times = None
        # pipelined stepping: sampled tokens still in flight on the GPU.  They are already counted in
        # ``num_tokens`` (placeholder id -1) so that the next step can be scheduled before they are known.
        self.pending = 0
        self.sample_row = -1            # row of this request in the in-flight step's sampler output
        self.finishing = False          # length limit reached at launch time; finishes when resolved
        # n > 1 sampling (OpenAI `n`): the parent prefills the prompt once; when it completes, n-1 children are forked
        # off its KV pages (shared, copy-on-write tail) and sample their own continuations
        self.n_samples = 1
        self.parent_id = None           # request id of the parent (children only)
        self.sample_index = 0           # choice index: 0 = the parent itself
        self.forked = False

    @property
    def num_output(self) -> int:
        return self.num_tokens - self.num_prompt

    @property
    def output_ids(self) -> np.ndarray:
        return self.tokens[self.num_prompt:self.num_tokens]

    @property
    def all_ids(self) -> np.ndarray:
        return self.tokens[:self.num_tokens]

    def append(self, tok: int) -> None:
        self.tokens[self.num_tokens] = tok
        self.num_tokens += 1

    @property
    def num_known(self) -> int:
        """Tokens whose ids the host knows (everything except in-flight placeholders)."""
        return self.num_tokens - self.pending


@dataclass
class ScheduledBatch:
    """What one engine step computes.  Sequences are ordered: single-token (decode) entries first,
    then multi-token prefill chunks."""
    requests: list                         # Request objects in batch order
    num_new: np.ndarray                    # int32 [B] tokens computed this step per sequence
    num_decode: int                        # leading entries with num_new == 1
    total_tokens: int
    sample_mask: np.ndarray                # bool [B]: sequence produces a token this step
    preempted: list = field(default_factory=list)
    burst_grant: float = 0.0               # seconds of step time above the TPOT target granted to urgent prompts


@dataclass
class RequestOutput:
    req_id: str
    new_token_ids: list
    finished: bool
    finish_reason: Optional[str] = None
    num_prompt_tokens: int = 0
    num_output_tokens: int = 0
    num_cached_tokens: int = 0
    client_tag: object = None
    parent_id: Optional[str] = None        # n > 1: the request the client knows; sample_index = its choice number
    sample_index: int = 0
