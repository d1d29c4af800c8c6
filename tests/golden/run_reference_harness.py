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


def readme_quick_start(work):
    """the ```python block of the reference's README "Quick Start for DuoAttention" (README.md:119-153), executed as it
    is printed there — relative model / pattern paths, keyword ``sparsity=0.5``, ``enable_duo_attention_eval(model, heads,
    sink_size=64, recent_size=256)``, ``model.cuda()`` — then the prefill + greedy decode loop the README describes,
    through the tuple caches"""
    import re
    import shutil

    import numpy as np
    import torch
    from transformers import LlamaConfig, LlamaForCausalLM

    text = open(os.path.join(REF, "README.md")).read()
    block = re.search(r"## Quick Start for DuoAttention.*?```python\n(.*?)```", text, re.S).group(1)
    model_rel = re.search(r'from_pretrained\(\s*"([^"]+)"', block).group(1)
    pattern_rel = re.search(r'load_attn_pattern\(\s*"([^"]+)"', block).group(1)
    # the paths of the snippet, relative to the working directory: a random-init model of the pattern's geometry (32 layers
    # x 8 kv heads, one q head per kv head) and the reference's own shipped pattern files, copied at run time
    os.makedirs(os.path.join(work, os.path.dirname(pattern_rel)), exist_ok=True)
    shutil.copytree(os.path.join(REF, pattern_rel), os.path.join(work, pattern_rel))
    heads = np.loadtxt(os.path.join(work, pattern_rel, "full_attention_heads.tsv"), delimiter="\t")
    torch.manual_seed(4)
    cfg = LlamaConfig(hidden_size=heads.shape[1] * 128, intermediate_size=64, num_hidden_layers=heads.shape[0],
                      num_attention_heads=heads.shape[1], num_key_value_heads=heads.shape[1], head_dim=128, vocab_size=32,
                      max_position_embeddings=4096, rope_theta=10000.0, tie_word_embeddings=False)
    LlamaForCausalLM(cfg).to(torch.bfloat16).save_pretrained(os.path.join(work, model_rel))
    torch.nn.Module.cuda = lambda self, *a, **k: self
    os.chdir(work)
    ns = {}
    exec(compile(block, "reference README.md quick start", "exec"), ns)
    model, sparsity = ns["model"], ns["sparsity"]
    ids = torch.randint(0, 32, (1, 333), generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        out = model(input_ids=ids[:, :330], past_key_values=None, use_cache=True)
        past = out.past_key_values
        toks = []
        for t in range(330, 333):
            out = model(input_ids=ids[:, t:t + 1], past_key_values=past, use_cache=True)
            past = out.past_key_values
            toks.append(int(out.logits[0, -1].argmax()))
    nf = [int((np.asarray(h) > 0.5).sum()) for h in ns["attn_heads"]]
    shapes_ok = all(past[l][0].shape == (2, nf[l], 333, 128) and past[l][1].shape == (2, heads.shape[1] - nf[l], 64 + 256, 128)
                    for l in range(heads.shape[0]))
    print("RESULT " + json.dumps([f"sparsity: {sparsity}", f"retrieval heads: {sum(nf)}", f"cache shapes ok: {shapes_ok}",
                                  f"finite: {bool(torch.isfinite(out.logits).all())}", f"tokens: {len(toks)}"]))


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
    if script == "README":
        return readme_quick_start(work)
    out = os.path.join(work, "out")
    sys.argv = [script, "--model_name", mdir, "--attn_load_dir", pdir, "--sparsity", "0.5", "--max_length", "50",
                "--prefilling_chunk_size", "20", "--device", "cpu", "--output_dir", out, "--seed", "42"]
    runpy.run_path(os.path.join(REF, "eval", "efficiency", script), run_name="__main__")
    print("RESULT " + json.dumps(open(os.path.join(out, "benchmark_result.txt")).read().splitlines()))


if __name__ == "__main__":
    main()
