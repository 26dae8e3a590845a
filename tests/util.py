import numpy as np


def oracle_config(O, cfg):
    return O.make_config(vocab_size=cfg.vocab_size, hidden_size=cfg.hidden_size, head_dim=cfg.head_dim,
                         num_hidden_layers=cfg.num_hidden_layers, num_attention_heads=cfg.num_attention_heads,
                         num_key_value_heads=cfg.num_key_value_heads, intermediate_size=cfg.intermediate_size,
                         max_position_embeddings=cfg.max_position_embeddings, rms_norm_eps=cfg.rms_norm_eps,
                         rope_theta=cfg.rope_theta, bos_token_id=cfg.bos_token_id, eos_token_id=cfg.eos_token_id)


def rel_err(got, ref):
    """the parity metric, fixed once: max|got-ref| / max|ref| (per tensor / per logits row)"""
    return float(np.abs(np.asarray(got, np.float64) - np.asarray(ref, np.float64)).max() / max(np.abs(ref).max(), 1e-30))


def row_rel_err(got, ref):
    return max(rel_err(g, r) for g, r in zip(got, ref))


def bf16_round(a):
    u = np.ascontiguousarray(a, np.float32).view(np.uint32)
    u = (u + 0x7FFF + ((u >> 16) & 1)) & 0xFFFF0000
    return u.view(np.float32)


LOGITS_TOL = 1e-3  # BASELINE.json north_star: logits within 1e-3 relative of the reference CPU path
