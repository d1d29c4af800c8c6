"""Config / pattern / device utilities (reference ``duo_attn/utils.py``).

In scope: ``parse_args`` (same flag names so scripts/efficiency.sh
style invocations keep working), ``get_model``, ``get_tokenizer``, ``to_device``
(single device and the layer-pipeline branch), ``load_attn_pattern``,
``sparsify_attention_heads``, ``seed_everything``, ``save_full_attention_heads``.
The third-party ``tensor_parallel`` branch of ``to_device`` is replaced by ``duo_attn.tp`` (one process per GPU).
"""
import argparse
import json
import os

import numpy as np
import torch


def parse_args(argv=None):
    """Same flags as reference utils.py:12-83."""
    p = argparse.ArgumentParser(description="kv_reduction")
    p.add_argument("--model_name", type=str, default="models/Llama-3-8B-Instruct-Gradient-1048k")
    p.add_argument("--config_name", type=str, default=None)
    p.add_argument("--dataset_name", type=str, default=None)
    p.add_argument("--dataset_format", type=str, default="multiple_passkey")
    p.add_argument("--split", type=str, default="train")
    p.add_argument("--lr", type=float, default=1e-1)
    p.add_argument("--num_steps", type=int, default=1000)
    p.add_argument("--batch_size", type=int, default=1)
    p.add_argument("--max_length", type=int, default=4096)
    p.add_argument("--context_length_min", type=int, default=1024)
    p.add_argument("--context_length_max", type=int, default=4096)
    p.add_argument("--context_lengths_num_intervals", type=int, default=20)
    p.add_argument("--depth_ratio_num_intervals", type=int, default=10)
    p.add_argument("--num_passkeys", type=int, default=10)
    p.add_argument("--output_dir", type=str, default="outputs")
    p.add_argument("--sink_size", type=int, default=64)
    p.add_argument("--recent_size", type=int, default=256)
    p.add_argument("--deploy_sink_size", type=int, default=None)
    p.add_argument("--deploy_recent_size", type=int, default=None)
    p.add_argument("--reg_weight", type=float, default=0.05)
    p.add_argument("--initial_value", type=float, default=1.0)
    p.add_argument("--exp_name", type=str, default=None)
    p.add_argument("--enable_pp", action="store_true")
    p.add_argument("--enable_tp", action="store_true")
    p.add_argument("--disable_wandb", action="store_true")
    p.add_argument("--min_needle_depth_ratio", type=float, default=0)
    p.add_argument("--max_needle_depth_ratio", type=float, default=1.0)
    p.add_argument("--save_steps", type=int, default=50)
    p.add_argument("--gradient_accumulation_steps", type=int, default=1)
    p.add_argument("--resume", action="store_true")
    p.add_argument("--rope_theta", type=float, default=None)
    p.add_argument("--device", type=str, default="0")
    p.add_argument("--streaming_attn_implementation", type=str, default="blocksparse")
    p.add_argument("--supervision", type=str, default="distill", choices=["classify", "distill"])
    p.add_argument("--n_samples", type=int, default=None)
    p.add_argument("--task", type=str, default="default")
    p.add_argument("--attn_load_dir", type=str, default=None)
    p.add_argument("--threshold", type=float, default=0.5)
    p.add_argument("--sparsity", type=float, default=None)
    p.add_argument("--passkey_length", type=int, default=32)
    p.add_argument("--context_length", type=int, default=16384)
    p.add_argument("--generation_length", type=int, default=256)
    p.add_argument("--stride_length", type=int, default=256)
    p.add_argument("--prefilling_chunk_size", type=int, default=4096)
    p.add_argument("--seed", type=int, default=42)
    args = p.parse_args(argv)
    args.device = parse_device(args.device)
    return args


def parse_device(device: str):
    if "," in device:
        return [int(d) for d in device.split(",")]
    if device in ["auto", "cpu"]:
        return device
    return f"cuda:{device}"


def get_model(model_name):
    """reference utils.py:94-105 — ``attn_implementation="eager"`` only selects the HF class whose
    attention forward is then replaced by the patch API."""
    import transformers

    model = transformers.AutoModelForCausalLM.from_pretrained(
        model_name, dtype=torch.bfloat16, low_cpu_mem_usage=True, attn_implementation="eager"
    )
    if hasattr(model.config, "sliding_window") and model.config.sliding_window is None:
        model.config.sliding_window = model.config.max_position_embeddings
    return model


def get_tokenizer(tokenizer_name):
    import transformers

    tok = transformers.AutoTokenizer.from_pretrained(tokenizer_name, use_fast=False, trust_remote_code=True)
    if tok.pad_token_id is None:
        tok.pad_token_id = tok.eos_token_id if tok.eos_token_id is not None else 0
    return tok


def even_layer_split(num_layers: int, num_stages: int):
    """Contiguous even split of decoder layers over pipeline stages (reference utils.py:251-271):
    returns ``[(first_layer, last_layer_exclusive), ...]`` per stage."""
    per = num_layers / num_stages
    bounds, start = [], 0
    for s in range(num_stages):
        end = num_layers if s == num_stages - 1 else int(round(per * (s + 1)))
        bounds.append((start, end))
        start = end
    return bounds


def balanced_layer_split(costs, num_stages: int):
    """Contiguous split of the layers that minimises the most expensive stage.

    With DuoAttention the per-layer cost is ragged: a layer's attention work and KV bytes grow with
    its number of retrieval heads, so the reference's even split (``even_layer_split``) leaves the
    pipeline waiting on whichever stage drew the retrieval-heavy layers.  ``costs[l]`` is any positive
    per-layer cost (e.g. ``base + n_full_kv_heads[l]``).  Exact DP over cut points; every stage gets
    at least one layer.  Returns ``[(first, last_exclusive), ...]``."""
    n = len(costs)
    if not 1 <= num_stages <= n:
        raise ValueError(f"{num_stages} stages for {n} layers")
    pre = [0.0]
    for c in costs:
        pre.append(pre[-1] + float(c))
    INF = float("inf")
    # best[s][i]: minimal bottleneck splitting the first i layers into s stages
    best = [[INF] * (n + 1) for _ in range(num_stages + 1)]
    cut = [[0] * (n + 1) for _ in range(num_stages + 1)]
    best[0][0] = 0.0
    for s in range(1, num_stages + 1):
        for i in range(s, n - (num_stages - s) + 1):
            for j in range(s - 1, i):
                b = max(best[s - 1][j], pre[i] - pre[j])
                if b < best[s][i]:
                    best[s][i], cut[s][i] = b, j
    bounds, i = [], n
    for s in range(num_stages, 0, -1):
        j = cut[s][i]
        bounds.append((j, i))
        i = j
    return bounds[::-1]


def to_device(model, device, enable_tp=False, enable_pp=False, reverse_device_map=True, even_split_layers=True,
              pp_handoff=None, full_attention_heads=None):
    """Single device: ``model.to(device)``.  ``enable_pp`` with a device list is the layer pipeline of reference
    utils.py:228-283 — as ONE PROCESS PER GPU (launch with torch.distributed.run): this rank keeps its contiguous
    block of decoder layers (embedding on the first stage, norm + lm_head on the last) on ``device[rank]`` and
    hands the hidden state to the next rank over RCCL point-to-point (``duo_attn.pipeline.shard_model_for_pp``).
    The calls that follow in the reference's harnesses work unchanged on every rank:
    ``enable_*_duo_attention_static_kv_cache_eval(model, heads)``, ``DuoAttentionStaticKVCache(model, heads, ...)``
    (this rank's pools only) and ``model(input_ids=chunk, past_key_values=kv)`` (logits on the last rank; for a
    decode step on every rank).  ``reverse_device_map`` (an accelerate hook-ordering detail) has no meaning here.
    ``pp_handoff="cpu"``: the hidden state crosses through host memory (a gloo group: one-GPU rehearsal of the path)."""
    if isinstance(device, list):
        if len(device) == 1:
            return model.to(f"cuda:{device[0]}")
        if enable_tp:
            # Reference utils.py:206-227 shards with the third-party `tensor_parallel` package inside one process; here it
            # is one process per GPU (launch with torch.distributed.run): this rank keeps Hkv / tp kv heads of every layer
            # (duo_attn.tp.shard_model_for_tp).  Either order works: BEFORE enable_*_duo_attention*_eval — the enablers and
            # the KV cache then take the WHOLE-model pattern and slice it to this rank's heads themselves — or AFTER it, as
            # the reference's NIAH / LongBench harnesses do (the reordered weights and the registered pattern are sharded).
            # `full_attention_heads` (optional, the pattern in the ORIGINAL head order) lets the split deal the retrieval
            # heads evenly over the ranks when the model does not carry the pattern itself.
            import torch.distributed as dist

            from .launch import SHARED_GPU_ENV, ensure_ranks

            # no process group yet: a rank of torch.distributed.run initialises it here; a plain `python harness.py` (the
            # reference's own launch shape, scripts/niah.sh:17) is started again as one rank per device and never returns
            ensure_ranks(len(device), "Tensor parallelism")
            if dist.get_world_size() != len(device):
                raise ValueError(f"{len(device)} devices for {dist.get_world_size()} ranks")
            from .tp import shard_model_for_tp

            # (shared-GPU rehearsal: every rank on the first device, gloo between them — not a measurement mode)
            dev = device[0 if os.environ.get(SHARED_GPU_ENV) == "1" else dist.get_rank()]
            dev = dev if isinstance(dev, str) else f"cuda:{dev}"
            if dev.startswith("cuda"):
                torch.cuda.set_device(dev)       # launches go to the CURRENT device's stream (duo_attn/_hip.py)
            model.to(dev)
            cfg = model.config
            heads = full_attention_heads
            if heads is None:
                # the reference's harness order (eval/needle/needle_in_haystack.py:195-214, eval/LongBench/pred.py:243):
                # enable_duo_attention_eval FIRST, then to_device(enable_tp=True) — the pattern is on the modules then
                # (reordered: retrieval heads first), and the split deals THOSE heads evenly.  Unpatched model, no
                # pattern: rank d takes the d-th contiguous block of kv heads, as the reference's split does.
                layers = model.model.layers
                if all("full_attention_heads" in l.self_attn._buffers for l in layers):
                    Hkv = cfg.num_key_value_heads
                    heads = np.zeros((len(layers), Hkv))
                    for li, l in enumerate(layers):
                        order = l.self_attn.__dict__.get("_duo_head_order") or list(range(Hkv))
                        heads[li, order] = l.self_attn._buffers["full_attention_heads"].detach().float().cpu().numpy()
                else:
                    heads = np.zeros((cfg.num_hidden_layers, cfg.num_key_value_heads))
            shard_model_for_tp(model, heads)
            return model
        if enable_pp:
            import torch.distributed as dist

            from .launch import SHARED_GPU_ENV, ensure_ranks

            ensure_ranks(len(device), "Layer pipeline")
            if dist.get_world_size() != len(device):
                raise ValueError(f"{len(device)} devices for {dist.get_world_size()} ranks")
            from .pipeline import shard_model_for_pp

            shared = os.environ.get(SHARED_GPU_ENV) == "1"
            dev = device[0 if shared else dist.get_rank()]
            shard_model_for_pp(model, dev if isinstance(dev, str) else f"cuda:{dev}",
                               handoff="cpu" if shared and pp_handoff is None else pp_handoff)
            return model
        raise ValueError("a device list needs enable_pp (or enable_tp)")
    return model.to(device)


def load_attn_pattern(attn_load_dir):
    """reference utils.py:326-336"""
    heads = np.loadtxt(os.path.join(attn_load_dir, "full_attention_heads.tsv"), dtype=float, delimiter="\t")
    heads = np.clip(heads, 0, 1)
    with open(os.path.join(attn_load_dir, "config.json")) as f:
        config = json.load(f)
    return heads, config["sink_size"], config["recent_size"]


def seed_everything(seed):
    """reference utils.py:339-350"""
    import random

    random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
    torch.backends.cudnn.deterministic = True
    torch.backends.cudnn.benchmark = True


def sparsify_attention_heads(full_attention_heads, threshold=None, sparsity=None):
    """reference utils.py:353-373.  Adds U(0, 1e-6) tie-break noise IN PLACE (np.random state matters:
    call seed_everything first), then thresholds at the global ``sparsity`` quantile; ``sparsity >= 1``
    prunes every head, ``<= 0`` none.  Returns ({0,1} array, true sparsity)."""
    full_attention_heads += np.random.uniform(0, 1e-6, full_attention_heads.shape)
    if sparsity is not None:
        threshold = np.quantile(full_attention_heads, sparsity)
    else:
        assert threshold is not None, "Either threshold or sparsity must be provided"
    # (the reference then compares sparsity to 1 and 0 unconditionally, :364-369, which raises
    #  TypeError for sparsity=None; threshold-only mode is kept usable here)
    if sparsity is not None and sparsity >= 1:
        threshold = 2
    if sparsity is not None and sparsity <= 0:
        threshold = -1
    full_attention_heads = (full_attention_heads >= threshold).astype(float)
    return full_attention_heads, 1 - np.mean(full_attention_heads)


def save_full_attention_heads(full_attention_heads, output_filename):
    heads = full_attention_heads
    if hasattr(heads, "detach"):           # torch tensor (or a list of them, below) -> numpy, no copy keyword games
        heads = heads.detach().float().cpu().numpy()
    elif isinstance(heads, (list, tuple)):
        heads = [h.detach().float().cpu().numpy() if hasattr(h, "detach") else h for h in heads]
    np.savetxt(output_filename, np.asarray(heads), delimiter="\t")
