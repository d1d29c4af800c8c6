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


def cuda_shim(ndev=1):
    """the harness's hard-coded device string and torch.cuda timing calls, on a machine without a GPU"""
    import torch

    orig_to = torch.Tensor.to

    def on_gpu(x):
        return (isinstance(x, str) and x.startswith("cuda")) or (isinstance(x, torch.device) and x.type == "cuda")

    def to(self, *a, **kw):
        a = tuple("cpu" if on_gpu(x) else x for x in a)
        if on_gpu(kw.get("device")):
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
    torch.cuda.device_count = lambda: ndev      # (the needle harness hands every visible GPU to to_device(..., enable_tp=True))
    torch.cuda.set_device = lambda *a, **k: None


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
    ids = torch.randint(0, 32, (1, 43), generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        out = model(input_ids=ids[:, :40], past_key_values=None, use_cache=True)
        past = out.past_key_values
        toks = []
        for t in range(40, 43):
            out = model(input_ids=ids[:, t:t + 1], past_key_values=past, use_cache=True)
            past = out.past_key_values
            toks.append(int(out.logits[0, -1].argmax()))
    nf = [int((np.asarray(h) > 0.5).sum()) for h in ns["attn_heads"]]
    shapes_ok = all(past[l][0].shape == (2, nf[l], 43, 128) and past[l][1].shape == (2, heads.shape[1] - nf[l], min(43, 64 + 256), 128)
                    for l in range(heads.shape[0]))
    print("RESULT " + json.dumps([f"sparsity: {sparsity}", f"retrieval heads: {sum(nf)}", f"cache shapes ok: {shapes_ok}",
                                  f"finite: {bool(torch.isfinite(out.logits).all())}", f"tokens: {len(toks)}"]))


def needle_in_a_haystack(work):
    """``eval/needle/needle_in_haystack.py`` VERBATIM (the accuracy harness of the tuple-cache API: enable_duo_attention_eval with
    --sink_size / --recent_size overrides, to_device(model, [gpus], enable_tp=True) on one device, chunked prefill handing
    ``output.past_key_values`` back, the question fed one token at a time, greedy generation until EOS) on a random-init model:
    two context lengths x two needle depths.  Its one missing third-party import, ``rouge_score``, is stood in for by a
    ten-line module written into the temp directory (the score of a random model is meaningless either way)."""
    import numpy as np

    # (a run with two "GPUs" is started again by to_device() as two ranks of torch.distributed.run, which execute this very
    #  function: the inputs exist by then — the first process wrote them before it reached to_device — and are not rewritten)
    ranks_of_a_relaunch = os.environ.get("DUO_ATTN_SELF_LAUNCHED") == "1"
    mdir, pdir = os.path.join(work, "tiny-llama") if ranks_of_a_relaunch else _char_model(work, "tiny-llama"), os.path.join(work, "pattern")
    for d in (pdir, os.path.join(work, "PaulGrahamEssays"), os.path.join(work, "rouge_score")):
        os.makedirs(d, exist_ok=True)
    if ranks_of_a_relaunch:
        return _needle_run(work, mdir, pdir)
    np.savetxt(os.path.join(pdir, "full_attention_heads.tsv"), np.array([[0.9, 0.1], [0.2, 0.7], [0.95, 0.6]]), delimiter="\t")
    with open(os.path.join(pdir, "config.json"), "w") as f:
        json.dump({"sink_size": 64, "recent_size": 256}, f)
    with open(os.path.join(work, "PaulGrahamEssays", "essay.txt"), "w") as f:
        f.write("the quick brown fox jumps over the lazy dog and keeps running through the long grass. " * 40)
    with open(os.path.join(work, "rouge_score", "__init__.py"), "w") as f:
        f.write("from . import rouge_scorer\n")
    with open(os.path.join(work, "rouge_score", "rouge_scorer.py"), "w") as f:
        f.write("import collections\nScore = collections.namedtuple('Score', 'precision recall fmeasure')\n"
                "class RougeScorer:\n    def __init__(self, kinds, use_stemmer=False):\n        self.kinds = kinds\n"
                "    def score(self, target, prediction):\n        a, b = set(target.split()), set(prediction.split())\n"
                "        f = 2 * len(a & b) / max(1, len(a) + len(b))\n        return {k: Score(f, f, f) for k in self.kinds}\n")
    return _needle_run(work, mdir, pdir)


def _needle_run(work, mdir, pdir):
    sys.path.insert(0, work)
    os.chdir(work)
    sys.argv = ["needle_in_haystack.py", "-s", "300", "-e", "700", "--model_path", mdir, "--attn_load_dir", pdir, "--sink_size", "8",
                "--recent_size", "24", "--simulation_length", "6", "--context_lengths_num_intervals", "2",
                "--document_depth_percent_intervals", "2", "--context_lengths_min", "450", "--context_lengths_max", "600",
                "--prefilling_chunk_size", "96", "--sparsity", "0.5"]
    runpy.run_path(os.path.join(REF, "eval", "needle", "needle_in_haystack.py"), run_name="__main__")
    import glob

    res = [json.load(open(p)) for p in sorted(glob.glob(os.path.join(work, "results", "tiny-llama", "*_results.json")))]
    print("RESULT " + json.dumps([f"results: {len(res)}", f"lengths: {sorted({r['context_length'] for r in res})}",
                                  f"depths: {sorted({r['depth_percent'] for r in res})}",
                                  f"fields ok: {all('model_response' in r and 'score' in r for r in res)}"]))


def _char_model(work, name):
    """a random-init three-layer Llama with a character-level tokenizer and a generation config, saved under work/name"""
    import string

    import torch
    from tokenizers import Tokenizer, models, pre_tokenizers
    from transformers import LlamaConfig, LlamaForCausalLM, PreTrainedTokenizerFast

    mdir = os.path.join(work, name)
    os.makedirs(mdir, exist_ok=True)
    chars = ["<unk>", "<s>", "</s>"] + sorted(set(string.ascii_letters + string.digits + string.punctuation + " \n"))
    tok = Tokenizer(models.WordLevel({c: i for i, c in enumerate(chars)}, unk_token="<unk>"))
    tok.pre_tokenizer = pre_tokenizers.Split("", behavior="isolated")
    PreTrainedTokenizerFast(tokenizer_object=tok, unk_token="<unk>", bos_token="<s>", eos_token="</s>").save_pretrained(mdir)
    torch.manual_seed(6)
    cfg = LlamaConfig(hidden_size=512, intermediate_size=256, num_hidden_layers=3, num_attention_heads=4, num_key_value_heads=2,
                      head_dim=128, vocab_size=128, max_position_embeddings=4096, rope_theta=10000.0, tie_word_embeddings=False,
                      bos_token_id=1, eos_token_id=2)
    LlamaForCausalLM(cfg).to(torch.bfloat16).save_pretrained(mdir)
    return mdir


def longbench_pred(work, method):
    """``eval/LongBench/pred.py`` VERBATIM, task ``trec``, with ``--method duo_attn`` (enable_duo_attention_eval, sink / recent
    overrides, sparsify with the keyword ``sparsity=``) or ``--method full`` (``enable_tuple_kv_cache``: the full-attention tuple
    baseline); ``to_device(model, [gpu ids], enable_tp=True)`` on one device; single-shot prefill, the last 50
    prompt tokens fed one at a time, greedy generation.  The harness's relative config paths are created in the temp directory
    (prompt / length tables copied from the reference at run time, the model table pointing at the random-init model);
    ``datasets.load_dataset`` — the hub is unreachable — is stood in for by a module returning three drawn TREC-shaped records."""
    import shutil

    import numpy as np

    mdir = _char_model(work, "tiny-llama")
    pdir = os.path.join(work, "pattern")
    cfg_dir = os.path.join(work, "eval", "LongBench", "config")
    for d in (pdir, cfg_dir, os.path.join(work, "datasets")):
        os.makedirs(d, exist_ok=True)
    np.savetxt(os.path.join(pdir, "full_attention_heads.tsv"), np.array([[0.9, 0.1], [0.2, 0.7], [0.95, 0.6]]), delimiter="\t")
    json.dump({"sink_size": 64, "recent_size": 256}, open(os.path.join(pdir, "config.json"), "w"))
    for n in ("dataset2prompt.json", "dataset2maxlen.json"):
        shutil.copy(os.path.join(REF, "eval", "LongBench", "config", n), cfg_dir)
    json.dump({"tiny-llama": mdir}, open(os.path.join(cfg_dir, "model2path.json"), "w"))
    json.dump({"tiny-llama": 400}, open(os.path.join(cfg_dir, "model2maxlen.json"), "w"))
    with open(os.path.join(work, "datasets", "__init__.py"), "w") as f:
        f.write("def load_dataset(name, subset, split='test'):\n"
                "    assert (name, subset, split) == ('THUDM/LongBench', 'trec', 'test')\n"
                "    mk = lambda i, n: {'context': 'Question: what is item %d ?\\nType: number\\n' % i * n, 'input': 'Question: how many ?\\nType:',\n"
                "                       'answers': ['number'], 'all_classes': ['number', 'person'], 'length': 40 * n}\n"
                "    return [mk(1, 6), mk(2, 14), mk(3, 3)]\n")       # (the second record is longer than max_length: truncated in the middle)
    sys.path.insert(0, work)
    os.chdir(work)
    sys.argv = ["pred.py", "--model", "tiny-llama", "--task", "trec", "--method", method, "--decoding_simulation_length", "7"]
    if method == "duo_attn":
        sys.argv += ["--attn_load_dir", pdir, "--sink_size", "8", "--recent_size", "24", "--sparsity", "0.5"]
    runpy.run_path(os.path.join(REF, "eval", "LongBench", "pred.py"), run_name="__main__")
    import glob

    out = glob.glob(os.path.join(work, "eval", "LongBench", "pred", "tiny-llama", "trec-*.jsonl"))
    rows = [json.loads(ln) for ln in open(out[0])]
    print("RESULT " + json.dumps([f"file: {os.path.basename(out[0])}", f"records: {len(rows)}",
                                  f"fields ok: {all(set(r) == {'pred', 'answers', 'all_classes', 'length'} for r in rows)}"]))


def main():
    script, work = sys.argv[1], sys.argv[2]
    if not os.path.isdir(os.path.join(REF, "eval", "efficiency")):
        raise SystemExit("/root/reference is not here: this script runs in the build container only")
    # this package's duo_attn first; the reference's eval/efficiency for its own `from utils import bench_func`;
    # the reference's root is NOT on the path, so `import duo_attn` cannot reach the reference's package
    for p in (os.path.join(REF, "eval", "efficiency"), ROOT, os.path.join(ROOT, "duo-attention_amd")):
        sys.path.insert(0, p)
    relaunched = os.environ.get("DUO_ATTN_SELF_LAUNCHED") == "1"
    mdir, pdir = (os.path.join(work, "model"), os.path.join(work, "pattern")) if relaunched else build_inputs(work)
    import duo_attn
    from duo_attn import backend
    from oracle.duo_oracle import OracleBackend

    assert os.path.realpath(duo_attn.__file__).startswith(os.path.realpath(ROOT)), duo_attn.__file__
    backend._set_backend_for_testing(OracleBackend())
    cuda_shim(2 if script == "needle-tp2" else 1)
    if script == "README":
        return readme_quick_start(work)
    if script in ("needle", "needle-tp2"):
        # "needle-tp2": the harness sees TWO devices and calls to_device(model, [0, 1], enable_tp=True) inside this one
        # process, as scripts/niah.sh starts it (python eval/needle/needle_in_haystack.py ...)
        return needle_in_a_haystack(work)
    if script.startswith("longbench:"):
        return longbench_pred(work, script.split(":", 1)[1])
    out = os.path.join(work, "out")
    sys.argv = [script, "--model_name", mdir, "--attn_load_dir", pdir, "--sparsity", "0.5", "--max_length", "50",
                "--prefilling_chunk_size", "20", "--device", "cpu", "--output_dir", out, "--seed", "42"]
    runpy.run_path(os.path.join(REF, "eval", "efficiency", script), run_name="__main__")
    print("RESULT " + json.dumps(open(os.path.join(out, "benchmark_result.txt")).read().splitlines()))


if __name__ == "__main__":
    main()
