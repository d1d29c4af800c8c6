#!/usr/bin/env python3
"""The INT4 legs of bench.py on their own (BASELINE configs[4]): one-layer kernel roofline at 1 M context, the 32-layer
decode step at 3.3 M context, one chunk of the INT4 prefill pipeline.  Prints one JSON object.
    python tools/debug/int4_legs.py [kernel|step|prefill|all]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "duo-attention_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

import bench  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "all"
dev = torch.device("cuda", 0)
counts = bench.LLAMA3_8B_FULL_KV_HEADS
res = {}
if which in ("kernel", "all"):
    res["kernel_1M"] = bench.int4_leg(dev, prefill=False, parity=False)["decode"]
if which in ("step", "all"):
    res["whole_step_3p3M"] = bench.int4_whole_step(dev, counts)
if which in ("prefill", "all"):
    res["prefill_chunk_pipeline"] = bench.int4_prefill_chunk(dev, counts)
print(json.dumps(res, indent=1))
