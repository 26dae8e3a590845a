"""Host-side engine mirror: Sequence / SamplingParams / BlockManager / Scheduler / LLMEngine and the
Qwen3ModelRunner that plugs the HIP step in (src/engine/*.rs, src/sampling_params.rs).

Only `Qwen3ModelRunner.run` touches the GPU.  The rest is the reference's single-threaded control loop,
reproduced so the bench and the tests drive the hot path exactly the way the reference engine would
(prefill-first scheduling, post_process, max_num_seqs), including its quirks.
"""
import itertools
from collections import deque
from dataclasses import dataclass

import numpy as np


@dataclass
class SamplingParams:
    """src/sampling_params.rs:1-46"""
    temperature: float = 1.0
    max_tokens: int = 64
    ignore_eos: bool = False

    def __post_init__(self):
        if not self.temperature > 1e-10:
            raise AssertionError("temperature must be > 0 for sampling")  # sampling_params.rs:20,30


_SEQ_COUNTER = itertools.count()
WAITING, RUNNING, FINISHED = "Waiting", "Running", "Finished"


class Sequence:
    """src/engine/sequence.rs:14-98"""

    def __init__(self, token_ids, sampling_params):
        self.seq_id = next(_SEQ_COUNTER)
        self.block_size = 256  # sequence.rs:35 (ignores SchedulerConfig.kvcache_block_size)
        self.block_table = []
        self.status = WAITING
        self.token_ids = list(token_ids)
        self.last_token = self.token_ids[-1] if self.token_ids else 0
        self.num_tokens = len(self.token_ids)
        self.num_prompt_tokens = len(self.token_ids)
        self.num_cached_tokens = 0
        self.temperature = sampling_params.temperature
        self.max_tokens = sampling_params.max_tokens
        self.ignore_eos = sampling_params.ignore_eos

    def __len__(self):
        return self.num_tokens

    def is_finished(self):
        return self.status == FINISHED

    def num_completion_tokens(self):
        return self.num_tokens - self.num_prompt_tokens

    def prompt_token_ids(self):
        return self.token_ids[:self.num_prompt_tokens]

    def completion_token_ids(self):
        return self.token_ids[self.num_prompt_tokens:]

    def num_blocks(self):
        return -(-self.num_tokens // self.block_size)

    def append_token(self, token_id):
        self.token_ids.append(int(token_id))
        self.last_token = int(token_id)
        self.num_tokens += 1


class BlockManager:
    """src/engine/block_manager.rs:24-99 -- the reference's stub, kept as-is on purpose: the real block
    pool lives behind the C ABI (nvllm_kv_alloc / nvllm_seq_free); this table is never consumed."""

    def __init__(self, num_blocks=0, block_size=256):
        self.num_blocks, self.block_size = num_blocks, block_size

    def can_allocate(self, seq):
        return True

    def allocate(self, seq):
        seq.block_table = list(range(seq.num_blocks()))
        seq.num_cached_tokens = len(seq)

    def deallocate(self, seq):
        seq.block_table = []
        seq.num_cached_tokens = 0

    def can_append(self, seq):
        return True

    def may_append(self, seq):
        pass


@dataclass
class SchedulerConfig:
    """src/engine/scheduler.rs:10-56"""
    max_num_seqs: int = 1
    max_num_batched_tokens: int = 4096
    eos: int = 0
    num_kvcache_blocks: int = 0
    kvcache_block_size: int = 256


class Scheduler:
    """src/engine/scheduler.rs:58-250"""

    def __init__(self, config, on_release=None):
        self.max_num_seqs = config.max_num_seqs
        self.max_num_batched_tokens = config.max_num_batched_tokens
        self.block_manager = BlockManager(config.num_kvcache_blocks, config.kvcache_block_size)
        self.waiting, self.running = deque(), deque()
        self.eos = config.eos
        self.on_release = on_release  # hook where BlockManager::deallocate is called -> nvllm_seq_free

    def is_finished(self):
        return not self.waiting and not self.running

    def add(self, seq):
        self.waiting.append(seq)

    def schedule(self):
        r = self._try_schedule_prefill()
        return r if r is not None else self._schedule_decode()

    def _try_schedule_prefill(self):
        scheduled, num_batched = [], 0
        while self.waiting and len(scheduled) < self.max_num_seqs:
            seq = self.waiting[0]
            if not (num_batched + len(seq) <= self.max_num_batched_tokens and self.block_manager.can_allocate(seq)):
                break
            self.block_manager.allocate(seq)
            num_batched += len(seq) - seq.num_cached_tokens  # always 0: allocate sets cached = len
            seq.status = RUNNING
            self.waiting.popleft()
            self.running.append(seq)
            scheduled.append(seq)
        return (scheduled, True) if scheduled else None

    def _schedule_decode(self):
        scheduled = []
        while self.running and len(scheduled) < self.max_num_seqs:
            seq = self.running.popleft()
            if self._ensure_can_append(seq):
                self.block_manager.may_append(seq)
                scheduled.append(seq)
        assert scheduled, "decode stage should schedule at least one sequence"
        for seq in reversed(scheduled):
            self.running.appendleft(seq)
        return scheduled, False

    def _ensure_can_append(self, seq):
        while not self.block_manager.can_append(seq):
            if self.running:
                self.preempt(self.running.pop())
            else:
                self.preempt(seq)
                return False
        return True

    def preempt(self, seq):
        seq.status = WAITING
        self.block_manager.deallocate(seq)
        if self.on_release:
            self.on_release(seq.seq_id)
        self.waiting.appendleft(seq)

    def post_process(self, seqs, token_ids):
        for seq, tok in zip(seqs, token_ids):
            seq.append_token(tok)
            finished = (not seq.ignore_eos and tok == self.eos) or seq.num_completion_tokens() >= seq.max_tokens
            if finished:
                seq.status = FINISHED
                self.block_manager.deallocate(seq)
                if self.on_release:
                    self.on_release(seq.seq_id)
                self.running = deque(s for s in self.running if s.seq_id != seq.seq_id)


class ModelRunner:
    """trait ModelRunner (src/engine/llm_engine.rs:16-18)"""

    def run(self, seqs, is_prefill):
        raise NotImplementedError


_M64 = (1 << 64) - 1


def _finalize64(z):
    """splitmix64 finalizer (csrc/synth_device.h synth_finalize), on Python ints"""
    z &= _M64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def sample_key(seed, seq_id, length):
    """per-row key of the device sampler's counter RNG (csrc/api.cpp sample_key)"""
    return _finalize64(_finalize64((seed * 0x9E3779B97F4A7C15 + (seq_id & _M64)) & _M64) + length * 0xD1B54A32D192ED03)


def sample_token_host(logits_last, temperature, key):
    """Host mirror of the device sampler (csrc/kernels.hip sample_rows_kernel), itself the reference's sample_token
    (llm_engine.rs:97-133): t = max(T, 1e-6); weights exp((l - max) / t); a row whose weights do not form a distribution
    falls back to the last-max arg-max; otherwise the categorical draw in Gumbel-max form over the counter RNG `key`."""
    l = np.asarray(logits_last, np.float32)
    t = np.float32(max(float(temperature), 1e-6))
    with np.errstate(all="ignore"):
        w = np.exp((l - l.max()) / t, dtype=np.float32)
        s = w.sum(dtype=np.float32)
    if not np.isfinite(s) or s <= 0 or np.isnan(l).any():
        return Qwen3ModelRunner.argmax(np.where(np.isnan(l), -np.inf, l))  # a NaN logit ranks below every number
    i = np.arange(1, l.size + 1, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = np.uint64(key) + i * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    u = ((z >> np.uint64(40)).astype(np.float32) + np.float32(0.5)) * np.float32(5.9604644775390625e-08)
    score = l / t - np.log(-np.log(u, dtype=np.float32), dtype=np.float32)
    return int(np.flatnonzero(score == score.max())[-1])


class Qwen3ModelRunner(ModelRunner):
    """src/engine/llm_engine.rs:35-189 with the forward replaced by the HIP step.

    greedy=False (default, like the reference): every id goes through sample_token -- softmax((l-max)/max(T,1e-6)) and a
                    categorical draw (llm_engine.rs:97-133) -- on the DEVICE (nvllm_step_sample), seeded per runner
                    (`seed`; the reference's RNG is unseeded, so a run of the reference is one such seed).
    greedy=True  -> ids come from the device arg-max (the reference's `argmax` fallback rule: last max).
    An empty sequence is fed as [eos] (the reference pads rows with eos and reads row 0, llm_engine.rs:80-90,181).
    Errors never propagate: any failure -> eos for every sequence (llm_engine.rs:153-175)."""

    def __init__(self, model, greedy=False, seed=None, raise_errors=False):
        self.model = model
        self.cfg = model.cfg
        self.eos_id = model.cfg.eos_token_id
        self.greedy = greedy
        self.seed = int.from_bytes(__import__("os").urandom(8), "little") if seed is None else int(seed)
        self.raise_errors = raise_errors
        self.last_error = None

    def run(self, seqs, is_prefill):
        if not seqs:
            return []
        toks = [s.token_ids if s.token_ids else [self.eos_id] for s in seqs]
        try:
            if self.greedy:
                ids, _ = self.model.step([s.seq_id for s in seqs], toks, is_prefill)
            else:
                ids, _ = self.model.step_sample([s.seq_id for s in seqs], toks, is_prefill, [s.temperature for s in seqs], self.seed)
        except Exception as e:  # noqa: BLE001 -- the reference logs and returns eos for all
            if self.raise_errors:
                raise
            self.last_error = e
            return [self.eos_id] * len(seqs)
        return [int(t) for t in ids]

    @staticmethod
    def argmax(logits):
        m = logits.max()
        return int(np.flatnonzero(logits == m)[-1])  # last max (Iterator::max_by)


class LLMEngine:
    """src/engine/llm_engine.rs:214-325"""

    def __init__(self, scheduler, model_runner):
        self.scheduler, self.model_runner = scheduler, model_runner
        if isinstance(model_runner, Qwen3ModelRunner) and scheduler.on_release is None:
            scheduler.on_release = model_runner.model.seq_free

    def add_request(self, token_ids, sampling_params):
        self.scheduler.add(Sequence(token_ids, sampling_params))

    def step(self):
        seqs, is_prefill = self.scheduler.schedule()
        token_ids = self.model_runner.run(seqs, is_prefill)
        self.scheduler.post_process(seqs, token_ids)
        outputs = [(s.seq_id, s.completion_token_ids()) for s in seqs if s.is_finished()]
        num_tokens = sum(len(s) for s in seqs) if is_prefill else -len(seqs)
        return outputs, is_prefill, num_tokens

    def is_finished(self):
        return self.scheduler.is_finished()

    def generate(self, prompts, sampling_params):
        for p in prompts:
            self.add_request(list(p), sampling_params)
        outputs = {}
        self.prefill_tokens = self.decode_tokens = 0
        while not self.is_finished():
            outs, is_prefill, n = self.step()
            if is_prefill:
                self.prefill_tokens += n
            else:
                self.decode_tokens += -n
            for sid, toks in outs:
                outputs[sid] = toks
        return [(sid, outputs[sid]) for sid in sorted(outputs)]
