"""Secondary cross-check of the CPU oracle: compare oracle/qwen3_oracle.c with
transformers' Qwen3 (models/qwen3/modeling_qwen3.py) built from a LOCAL random config on CPU.

Run in the build container only (needs `transformers`; no hub access, no from_pretrained of a model
name).  It is NOT the reference (the reference is Rust/Candle and cannot be built here); it is an
independent implementation of the same architecture whose state-dict keys are the HF names the
reference loads (src/models/qwen3.rs:150...526).  Output of the last run is kept in
oracle/validate_vs_hf.log.

    python oracle/validate_vs_hf.py
"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402


def main():
    from transformers import Qwen3Config, Qwen3ForCausalLM

    torch.manual_seed(0)
    worst = 0.0
    for (H, L, nh, kv, hd, I, V) in [(64, 2, 4, 2, 32, 192, 512), (128, 3, 8, 2, 16, 256, 300), (96, 2, 6, 6, 32, 160, 257)]:
        hf_cfg = Qwen3Config(vocab_size=V, hidden_size=H, num_hidden_layers=L, num_attention_heads=nh,
                             num_key_value_heads=kv, head_dim=hd, intermediate_size=I, max_position_embeddings=512,
                             rms_norm_eps=1e-6, rope_theta=1e6, tie_word_embeddings=False, attention_bias=False,
                             hidden_act="silu", use_sliding_window=False)
        hf_cfg._attn_implementation = "eager"
        model = Qwen3ForCausalLM(hf_cfg).float().eval()
        # make the norm weights non-trivial
        with torch.no_grad():
            for n, p in model.named_parameters():
                if "norm" in n:
                    p.copy_(1.0 + 0.1 * torch.randn_like(p))
        cfg = O.make_config(vocab_size=V, hidden_size=H, head_dim=hd, num_hidden_layers=L, num_attention_heads=nh,
                            num_key_value_heads=kv, intermediate_size=I, max_position_embeddings=512)
        om = O.Model(cfg)
        for name, p in model.state_dict().items():
            om.set_tensor(name, p.detach().numpy())
        ids = torch.randint(0, V, (3, 17))
        with torch.no_grad():
            ref = model(input_ids=ids).logits.numpy()
        hidden = om.forward(ids.numpy().astype(np.uint32))
        got = om.compute_logits(hidden)
        err = float(np.abs(got - ref).max() / np.abs(ref).max())
        worst = max(worst, err)
        print(f"H={H} L={L} nh={nh} kv={kv} hd={hd} I={I} V={V}: max|d|/max|ref| = {err:.3e}")
    print("worst", worst)
    assert worst < 2e-5, "oracle disagrees with transformers Qwen3"
    print("OK")


if __name__ == "__main__":
    main()
