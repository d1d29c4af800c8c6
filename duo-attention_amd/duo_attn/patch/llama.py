"""Llama family entry points — same names as reference duo_attn/patch/llama.py.

The reference's llama.py and mistral.py are identical up to the family name; both
re-export the single implementation in _duo.py.
"""
from ._duo import (
    duo_attention_forward_one_way_reordered as llama_duo_attention_forward_one_way_reordered,
    duo_attention_forward_one_way_reordered_static as llama_duo_attention_forward_one_way_reordered_static,
    enable_duo_attention_eval as enable_llama_duo_attention_eval,
    enable_duo_attention_static_kv_cache_eval as enable_llama_duo_attention_static_kv_cache_eval,
    enable_duo_attention_training as enable_llama_duo_attention_training,
    get_full_attention_heads as get_llama_full_attention_heads,
    map_full_attention_heads as map_llama_full_attention_heads,
    set_full_attention_heads as set_llama_full_attention_heads,
)
from .static_kv_cache import (  # noqa: F401
    DuoAttentionStaticKVCache,
    enable_duo_attention_static_kv_cache_for_llama,
)
from .tuple_kv_cache import enable_tuple_kv_cache_for_llama  # noqa: F401

__all__ = [
    "llama_duo_attention_forward_one_way_reordered",
    "llama_duo_attention_forward_one_way_reordered_static",
    "enable_llama_duo_attention_eval",
    "enable_llama_duo_attention_static_kv_cache_eval",
    "enable_llama_duo_attention_training",
    "get_llama_full_attention_heads",
    "set_llama_full_attention_heads",
    "map_llama_full_attention_heads",
    "DuoAttentionStaticKVCache",
]
