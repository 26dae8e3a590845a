"""CPU: host-side mirror of the reference's non-GPU logic (tp.rs, sampling_params.rs, scheduler.rs,
llm_engine.rs, qwen3.rs config parsing).  No GPU, no oracle arithmetic."""
import json
import struct

import numpy as np
import pytest

import nano_vllm_candle_amd as pkg
from nano_vllm_candle_amd.engine import LLMEngine, ModelRunner, SamplingParams, Scheduler, SchedulerConfig, Sequence
from nano_vllm_candle_amd.qwen3 import Qwen3Config, checkpoint_files, read_safetensors
from nano_vllm_candle_amd.tp import TPConfig, get_tp, shard_region


# ---- src/tp.rs:76-98 -------------------------------------------------------------------------------
def test_tp_single_config():
    cfg = TPConfig.single()
    assert (cfg.size, cfg.rank) == (1, 0) and not cfg.is_distributed()


def test_tp_builder_pattern():
    cfg = TPConfig.single().with_size(4).with_rank(2).with_dim(1)
    assert (cfg.size, cfg.rank, cfg.dim) == (4, 2, 1) and cfg.is_distributed()


def test_tp_shard_calculation():
    cfg = TPConfig(4, 2, 0)
    assert cfg.shard_size(100) == 25 and cfg.shard_offset(100) == 50


def test_tp_from_env(monkeypatch):
    monkeypatch.setenv("TP_SIZE", "4")
    monkeypatch.setenv("TP_RANK", "3")
    assert get_tp() == TPConfig(4, 3, 0)
    monkeypatch.setenv("TP_RANK", "7")  # rank >= size folds to 0 (tp.rs:24-29)
    assert get_tp().rank == 0
    monkeypatch.delenv("TP_SIZE")
    assert get_tp() == TPConfig(1, 0, 0)


def test_tp_shard_regions_partition_every_tensor():
    cfg = Qwen3Config.qwen3_0_6b()
    for name, shape in cfg.hf_tensor_shapes().items():
        if ".layers." in name and not name.startswith("model.layers.0."):
            continue
        rows, cols = (shape[0], shape[1]) if len(shape) == 2 else (1, shape[0])
        for tp in (1, 2, 8):
            regs = [shard_region(cfg, tp, r, name) for r in range(tp)]
            replicated = "norm" in name or "embed" in name
            if replicated:
                assert all(r == (0, 0, rows, cols) for r in regs)
            else:
                area = sum(r[2] * r[3] for r in regs)
                assert area == rows * cols
                assert len({(r[0], r[1]) for r in regs}) == tp


# ---- src/sampling_params.rs:52-76 ------------------------------------------------------------------
def test_sampling_params_default_and_builder():
    p = SamplingParams()
    assert abs(p.temperature - 1.0) < 1e-12 and p.max_tokens == 64 and not p.ignore_eos
    p = SamplingParams(temperature=0.7, max_tokens=128, ignore_eos=True)
    assert (p.temperature, p.max_tokens, p.ignore_eos) == (0.7, 128, True)


def test_zero_temperature_panics():
    with pytest.raises(AssertionError, match="temperature must be > 0"):
        SamplingParams(temperature=0.0)


# ---- engine (llm_engine.rs / scheduler.rs) ---------------------------------------------------------
class DummyModelRunner(ModelRunner):
    """llm_engine.rs:20-33: returns last_token + 1"""

    def __init__(self):
        self.calls = []

    def run(self, seqs, is_prefill):
        self.calls.append((len(seqs), is_prefill, [len(s) for s in seqs]))
        return [s.last_token + 1 for s in seqs]


def test_engine_prefill_first_then_decode_until_max_tokens():
    runner = DummyModelRunner()
    eng = LLMEngine(Scheduler(SchedulerConfig(max_num_seqs=2, eos=9999)), runner)
    out = eng.generate([[10, 11], [20], [30, 31, 32]], SamplingParams(max_tokens=3, ignore_eos=True))
    assert [toks for _, toks in out] == [[12, 13, 14], [21, 22, 23], [33, 34, 35]]
    # first call: prefill of 2 (max_num_seqs), then prefill of the third while 2 are running, then decodes
    assert runner.calls[0] == (2, True, [2, 1])
    assert runner.calls[1][1] is True and runner.calls[1][0] == 1
    assert all(not c[1] for c in runner.calls[2:])
    # llm_engine.rs:253-257 counts len() AFTER post_process appended the new token: (2+1)+(1+1)+(3+1)
    assert eng.decode_tokens == 6 and eng.prefill_tokens == 9


def test_engine_eos_stops_sequence_unless_ignored():
    class EosRunner(ModelRunner):
        def run(self, seqs, is_prefill):
            return [7 for _ in seqs]

    eng = LLMEngine(Scheduler(SchedulerConfig(max_num_seqs=4, eos=7)), EosRunner())
    out = eng.generate([[1, 2, 3]], SamplingParams(max_tokens=5))
    assert out[0][1] == [7]
    eng = LLMEngine(Scheduler(SchedulerConfig(max_num_seqs=4, eos=7)), EosRunner())
    out = eng.generate([[1, 2, 3]], SamplingParams(max_tokens=5, ignore_eos=True))
    assert out[0][1] == [7] * 5


def test_scheduler_release_hook_called_where_reference_deallocates():
    freed = []
    sch = Scheduler(SchedulerConfig(max_num_seqs=2, eos=0), on_release=freed.append)
    eng = LLMEngine(sch, DummyModelRunner())
    eng.generate([[5], [6]], SamplingParams(max_tokens=2, ignore_eos=True))
    assert len(freed) == 2


def test_sequence_block_math():
    s = Sequence(list(range(300)), SamplingParams())
    assert s.block_size == 256 and s.num_blocks() == 2 and len(s) == 300
    s.append_token(9)
    assert s.last_token == 9 and s.num_completion_tokens() == 1 and s.completion_token_ids() == [9]


# ---- config / checkpoint files ----------------------------------------------------------------------
def test_config_from_hf_dir(tmp_path):
    d = dict(vocab_size=151936, hidden_size=1024, num_hidden_layers=28, num_attention_heads=16, num_key_value_heads=8,
             intermediate_size=3072, max_position_embeddings=40960, rms_norm_eps=1e-6, hidden_act="silu",
             bos_token_id=151643, eos_token_id=151645)
    (tmp_path / "config.json").write_text(json.dumps(d))
    c = Qwen3Config.from_hf_dir(str(tmp_path))
    assert c.head_dim == 64 and c.rope_theta == 1e6  # head_dim absent -> hidden/heads; rope_theta default (qwen3.rs:90-97)
    d.update(head_dim=128, rope_theta=5e5)
    (tmp_path / "config.json").write_text(json.dumps(d))
    c = Qwen3Config.from_hf_dir(str(tmp_path))
    assert c.head_dim == 128 and c.rope_theta == 5e5
    with pytest.raises(RuntimeError, match="Failed to read HF config"):
        Qwen3Config.from_hf_dir(str(tmp_path / "missing"))
    (tmp_path / "config.json").write_text("{not json")
    with pytest.raises(RuntimeError, match="Failed to parse HF config"):
        Qwen3Config.from_hf_dir(str(tmp_path))


def test_safetensors_reader_roundtrip(tmp_path):
    a = np.arange(12, dtype=np.float32).reshape(3, 4)
    b = (np.arange(8, dtype=np.uint16) + 0x3F80).reshape(2, 4)  # bf16 bit patterns
    blobs = a.tobytes() + b.tobytes()
    header = {"x.weight": {"dtype": "F32", "shape": [3, 4], "data_offsets": [0, a.nbytes]},
              "y.weight": {"dtype": "BF16", "shape": [2, 4], "data_offsets": [a.nbytes, a.nbytes + b.nbytes]},
              "__metadata__": {"format": "pt"}}
    hj = json.dumps(header).encode()
    p = tmp_path / "model.safetensors"
    p.write_bytes(struct.pack("<Q", len(hj)) + hj + blobs)
    got = {n: (arr.copy(), code) for n, arr, code in read_safetensors(str(p))}
    assert np.array_equal(got["x.weight"][0], a) and got["x.weight"][1] == pkg._lib.DTYPE_F32
    assert np.array_equal(got["y.weight"][0], b) and got["y.weight"][1] == pkg._lib.DTYPE_BF16


def test_checkpoint_files_single_sharded_and_missing(tmp_path):
    with pytest.raises(RuntimeError, match="no such file"):
        checkpoint_files(str(tmp_path))
    (tmp_path / "model.safetensors").write_bytes(b"")
    assert checkpoint_files(str(tmp_path)) == [str(tmp_path / "model.safetensors")]
    wm = {"a": "model-00002-of-00002.safetensors", "b": "model-00001-of-00002.safetensors", "c": "model-00002-of-00002.safetensors"}
    (tmp_path / "model.safetensors.index.json").write_text(json.dumps({"weight_map": wm}))
    assert checkpoint_files(str(tmp_path)) == [str(tmp_path / "model-00001-of-00002.safetensors"),
                                               str(tmp_path / "model-00002-of-00002.safetensors")]  # the index wins; each shard once


def test_hf_tensor_names_match_reference_loader():
    names = Qwen3Config.tiny().hf_tensor_shapes()
    for n in ("model.embed_tokens.weight", "lm_head.weight", "model.norm.weight", "model.layers.1.self_attn.q_norm.weight",
              "model.layers.0.mlp.down_proj.weight", "model.layers.1.post_attention_layernorm.weight"):
        assert n in names
    assert len(names) == 3 + 11 * 2


# ---- sample_token (src/engine/llm_engine.rs:97-133): host mirror of the device sampler --------------------------
def test_sampler_mirror_degenerate_rows_fall_back_to_last_max():
    from nano_vllm_candle_amd.engine import sample_key, sample_token_host

    k = sample_key(1, 2, 3)
    assert sample_token_host(np.array([0.0, 3.0, 2.9, 1.0], np.float32), 1e-12, k) == 1      # T clamps to 1e-6: the arg-max
    assert sample_token_host(np.array([1.0, np.nan, 5.0, 5.0], np.float32), 1.0, k) == 3     # NaN weights: no distribution
    assert sample_token_host(np.array([-np.inf] * 4, np.float32), 1.0, k) == 3               # exp(nan): no distribution
    assert sample_key(1, 2, 3) == sample_key(1, 2, 3) != sample_key(1, 2, 4)


def test_sampler_mirror_draws_from_the_softmax():
    from nano_vllm_candle_amd.engine import sample_key, sample_token_host

    logits = np.array([2.0, 0.5, -1.0, 1.5, 0.0, -3.0, 1.0, 0.25], np.float32)
    for temp in (0.7, 1.0, 2.5):
        p = np.exp((logits - logits.max()) / temp)
        p /= p.sum()
        n = 20000
        counts = np.bincount([sample_token_host(logits, temp, sample_key(7, 11, i)) for i in range(n)], minlength=8)
        # every class within 4 standard deviations of its binomial expectation
        assert (np.abs(counts - n * p) <= 4 * np.sqrt(n * p * (1 - p)) + 1).all(), (temp, counts, n * p)


def test_every_tool_script_compiles():
    # tools/ run on the GPU box only; a syntax error there would surface in the middle of a measurement session
    import ast
    import glob
    import os

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    files = sorted(glob.glob(os.path.join(root, "tools", "*.py")) + [os.path.join(root, "bench.py"), os.path.join(root, "__graft_entry__.py")])
    assert len(files) > 15
    for f in files:
        ast.parse(open(f).read(), filename=f)
