"""Mistral family entry points — same names as reference duo_attn/patch/mistral.py.

The reference's llama.py and mistral.py are identical up to the family name; both
re-export the single implementation in _duo.py.
"""
from ._duo import (
    duo_attention_forward_one_way_reordered as mistral_duo_attention_forward_one_way_reordered,
    duo_attention_forward_one_way_reordered_static as mistral_duo_attention_forward_one_way_reordered_static,
    enable_duo_attention_eval as enable_mistral_duo_attention_eval,
    enable_duo_attention_static_kv_cache_eval as enable_mistral_duo_attention_static_kv_cache_eval,
    enable_duo_attention_training as enable_mistral_duo_attention_training,
    get_full_attention_heads as get_mistral_full_attention_heads,
    map_full_attention_heads as map_mistral_full_attention_heads,
    set_full_attention_heads as set_mistral_full_attention_heads,
)
from .static_kv_cache import (  # noqa: F401
    DuoAttentionStaticKVCache,
    enable_duo_attention_static_kv_cache_for_mistral,
)
from .tuple_kv_cache import enable_tuple_kv_cache_for_mistral  # noqa: F401

__all__ = [
    "mistral_duo_attention_forward_one_way_reordered",
    "mistral_duo_attention_forward_one_way_reordered_static",
    "enable_mistral_duo_attention_eval",
    "enable_mistral_duo_attention_static_kv_cache_eval",
    "enable_mistral_duo_attention_training",
    "get_mistral_full_attention_heads",
    "set_mistral_full_attention_heads",
    "map_mistral_full_attention_heads",
    "DuoAttentionStaticKVCache",
]
