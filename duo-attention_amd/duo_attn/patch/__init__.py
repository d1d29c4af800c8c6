"""``duo_attn.patch`` — the public patch API (reference ``duo_attn/patch/__init__.py:22-121``)."""
import os

import numpy as np
import torch

from .llama import (
    enable_llama_duo_attention_eval,
    enable_llama_duo_attention_training,
    get_llama_full_attention_heads,
    map_llama_full_attention_heads,
    set_llama_full_attention_heads,
)
from .mistral import (
    enable_mistral_duo_attention_eval,
    enable_mistral_duo_attention_training,
    get_mistral_full_attention_heads,
    map_mistral_full_attention_heads,
    set_mistral_full_attention_heads,
)


def _family(model):
    mt = model.config.model_type
    if "llama" in mt:
        return "llama"
    if "mistral" in mt or "mixtral" in mt:
        return "mistral"
    raise ValueError(f"Model type {mt} not supported")


def enable_duo_attention_training(model, sink_size, recent_size, max_length, initial_value=1.0,
                                  enable_ulysses_attention=False, streaming_attn_implementation="blocksparse"):
    fam = _family(model)
    fn = enable_llama_duo_attention_training if fam == "llama" else enable_mistral_duo_attention_training
    fn(model, sink_size, recent_size, max_length, initial_value=initial_value,
       enable_ulysses_attention=enable_ulysses_attention,
       streaming_attn_implementation=streaming_attn_implementation)


def enable_duo_attention_eval(model, full_attention_heads, sink_size, recent_size):
    print(f"Enabling DuoAttention evaluation using sink size {sink_size} and recent size {recent_size}")
    fam = _family(model)
    fn = enable_llama_duo_attention_eval if fam == "llama" else enable_mistral_duo_attention_eval
    fn(model, full_attention_heads, sink_size, recent_size)


def get_full_attention_heads(model):
    return (get_llama_full_attention_heads if _family(model) == "llama" else get_mistral_full_attention_heads)(model)


def set_full_attention_heads(model, full_attention_heads):
    fn = set_llama_full_attention_heads if _family(model) == "llama" else set_mistral_full_attention_heads
    fn(model, full_attention_heads)
    return model


def map_full_attention_heads(model, func):
    fn = map_llama_full_attention_heads if _family(model) == "llama" else map_mistral_full_attention_heads
    return fn(model, func)


def load_full_attention_heads(load_dir, filename="full_attention_heads.tsv"):
    heads = np.loadtxt(os.path.join(load_dir, filename), dtype=float, delimiter="\t")
    heads = np.clip(heads, 0, 1)
    return torch.tensor(heads, dtype=torch.float32)
