#!/usr/bin/env python3
"""Decode step of the TUPLE cache path (enable_duo_attention_eval) at long context, one Llama-3-8B-shaped
layer: growing arena (this repo) vs the reference's re-concatenation of the whole retrieval cache per token."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "duo-attention_amd"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from transformers import LlamaConfig, LlamaForCausalLM  # noqa: E402

from duo_attn.patch import _duo, enable_duo_attention_eval  # noqa: E402


def reference_style_append(module, past_full, new_k, new_v):
    bsz = new_k.shape[0]
    if past_full is not None:
        p = past_full.transpose(1, 2)
        new_k = torch.cat([p[:bsz], new_k], dim=1)
        new_v = torch.cat([p[bsz:], new_v], dim=1)
    return torch.cat([new_k, new_v], dim=0).transpose(1, 2)


def main():
    N, chunk = int(sys.argv[1]) if len(sys.argv) > 1 else 131072, 16384
    cfg = LlamaConfig(hidden_size=4096, intermediate_size=1024, num_hidden_layers=1, num_attention_heads=32,
                      num_key_value_heads=8, vocab_size=256, max_position_embeddings=1048576,
                      rope_theta=3580165449.0, attn_implementation="eager", tie_word_embeddings=False)
    torch.set_default_dtype(torch.bfloat16)
    with torch.device("cuda"):
        model = LlamaForCausalLM(cfg).eval()
    torch.set_default_dtype(torch.float32)
    enable_duo_attention_eval(model, np.array([[1.0, 1.0, 1.0, 1.0, 0.0, 0.0, 0.0, 0.0]]), 128, 256)
    ids = torch.randint(0, 256, (1, N), device="cuda")
    for name, fn in (("arena", _duo._tuple_full_kv_append), ("reference-style cat", reference_style_append)):
        _duo._tuple_full_kv_append = fn
        past = None
        with torch.no_grad():
            for s in range(0, N, chunk):
                past = model(input_ids=ids[:, s:s + chunk], past_key_values=past, use_cache=True).past_key_values
            tok = ids[:, :1]
            for _ in range(5):
                past = model(input_ids=tok, past_key_values=past, use_cache=True).past_key_values
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            steps = 30
            for _ in range(steps):
                past = model(input_ids=tok, past_key_values=past, use_cache=True).past_key_values
            torch.cuda.synchronize()
        print(f"{name}: {1e3 * (time.perf_counter() - t0) / steps:.3f} ms per decode step (1 layer, {N} tokens, 4 retrieval kv heads)")
        del past
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
