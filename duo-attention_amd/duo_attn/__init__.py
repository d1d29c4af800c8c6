"""duo_attn — MI355X-native DuoAttention hot path behind the reference's Python API.

    from duo_attn.patch import enable_duo_attention_eval
    from duo_attn.patch.llama import enable_llama_duo_attention_static_kv_cache_eval, DuoAttentionStaticKVCache
    from duo_attn.utils import load_attn_pattern, sparsify_attention_heads, seed_everything

Attention, RoPE, RMSNorm and the KV-pool updates run in hand-written HIP kernels
for gfx950 (``duo-attention_amd/csrc`` -> ``lib/libduoattn_hip.so``, C ABI in
``include/duo_attn_hip.h``).  There is no CPU path.
"""
__version__ = "0.1.0"
