"""Build container only: execute the reference's OWN efficiency harnesses — ``eval/efficiency/benchmark_static.py`` and
``benchmark_dynamic.py``, VERBATIM from /root/reference through ``runpy`` (nothing of them is copied here) — against THIS
package's ``duo_attn``: every name they import (``duo_attn.utils.{get_model, get_tokenizer, parse_args, to_device,
load_attn_pattern, seed_everything, sparsify_attention_heads}``, ``duo_attn.patch.enable_duo_attention_eval``,
``duo_attn.patch.llama.{enable_llama_duo_attention_static_kv_cache_eval, DuoAttentionStaticKVCache}``), every call signature
and every attribute they touch (``kv_cache.clear() / evict_last(1) / memory_usage``, ``outputs.logits / past_key_values``) has to
be there for the script to reach its last line and write ``benchmark_result.txt``.

There is no GPU in the build container and the product has no CPU path, so the run uses the two seams the test-suite uses
anyway: the CPU oracle plugged in as the device backend (``backend._set_backend_for_testing``), and — because the harness
hard-codes ``.to("cuda")`` and times with ``torch.cuda.Event`` — a shim that maps "cuda" to "cpu" and stubs the
``torch.cuda`` timing / memory calls.  What is exercised is the drop-in SURFACE (imports, signatures, control flow, cache
protocol, file output) with a tiny random Llama, a five-word tokenizer and a drawn head pattern written to a temp directory.

    python tests/golden/run_reference_harness.py benchmark_static.py /tmp/work
"""
import json
import os
import runpy
import sys
import time

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"


def build_inputs(work):
    import numpy as np
    import torch
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import LlamaConfig, LlamaForCausalLM, PreTrainedTokenizerFast

    mdir, pdir = os.path.join(work, "model"), os.path.join(work, "pattern")
    os.makedirs(mdir, exist_ok=True)
    os.makedirs(pdir, exist_ok=True)
    torch.manual_seed(3)
    cfg = LlamaConfig(hidden_size=512, intermediate_size=256, num_hidden_layers=3, num_attention_heads=4, num_key_value_heads=2,
                      head_dim=128, vocab_size=8, max_position_embeddings=4096, rope_theta=10000.0, tie_word_embeddings=False)
    LlamaForCausalLM(cfg).to(torch.bfloat16).save_pretrained(mdir)
    tok = Tokenizer(models.WordLevel({"<unk>": 0, "<s>": 1, "</s>": 2, "a": 3, "\n": 4}, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.Split("", behavior="isolated")
    PreTrainedTokenizerFast(tokenizer_object=tok, unk_token="<unk>", bos_token="<s>", eos_token="</s>").save_pretrained(mdir)
    np.savetxt(os.path.join(pdir, "full_attention_heads.tsv"), np.array([[0.9, 0.1], [0.2, 0.7], [0.95, 0.6]]), delimiter="\t")
    with open(os.path.join(pdir, "config.json"), "w") as f:
        json.dump({"sink_size": 4, "recent_size": 12}, f)
    return mdir, pdir


def cuda_shim():
    """the harness's hard-coded device string and torch.cuda timing calls, on a machine without a GPU"""
    import torch

    orig_to = torch.Tensor.to

    def to(self, *a, **kw):
        a = tuple("cpu" if isinstance(x, str) and x.startswith("cuda") else x for x in a)
        if isinstance(kw.get("device"), str) and kw["device"].startswith("cuda"):
            kw["device"] = "cpu"
        return orig_to(self, *a, **kw)

    torch.Tensor.to = to

    class Event:
        def __init__(self, enable_timing=False):
            self.t = None

        def record(self):
            self.t = time.perf_counter()

        def elapsed_time(self, other):
            return (other.t - self.t) * 1e3

    torch.cuda.Event = Event
    torch.cuda.synchronize = lambda *a, **k: None
    torch.cuda.reset_peak_memory_stats = lambda *a, **k: None
    torch.cuda.max_memory_allocated = lambda *a, **k: 0


def main():
    script, work = sys.argv[1], sys.argv[2]
    if not os.path.isdir(os.path.join(REF, "eval", "efficiency")):
        raise SystemExit("/root/reference is not here: this script runs in the build container only")
    # this package's duo_attn first; the reference's eval/efficiency for its own `from utils import bench_func`;
    # the reference's root is NOT on the path, so `import duo_attn` cannot reach the reference's package
    for p in (os.path.join(REF, "eval", "efficiency"), ROOT, os.path.join(ROOT, "duo-attention_amd")):
        sys.path.insert(0, p)
    mdir, pdir = build_inputs(work)
    import duo_attn
    from duo_attn import backend
    from oracle.duo_oracle import OracleBackend

    assert os.path.realpath(duo_attn.__file__).startswith(os.path.realpath(ROOT)), duo_attn.__file__
    backend._set_backend_for_testing(OracleBackend())
    cuda_shim()
    out = os.path.join(work, "out")
    sys.argv = [script, "--model_name", mdir, "--attn_load_dir", pdir, "--sparsity", "0.5", "--max_length", "50",
                "--prefilling_chunk_size", "20", "--device", "cpu", "--output_dir", out, "--seed", "42"]
    runpy.run_path(os.path.join(REF, "eval", "efficiency", script), run_name="__main__")
    print("RESULT " + json.dumps(open(os.path.join(out, "benchmark_result.txt")).read().splitlines()))


if __name__ == "__main__":
    main()
