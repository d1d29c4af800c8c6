"""32-layer fused decode steps of the bench pattern at 131072 context (for rocprofv3 --kernel-trace)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "duo-attention_amd"))
import torch
import bench

dev = torch.device("cuda", 0)
counts = bench.LLAMA3_8B_FULL_KV_HEADS
hp = bench.HotPath(counts, (0, len(counts)), 131072, 16384, dev)
for l in range(len(counts)):          # pretend the context is cached (contents irrelevant for timing)
    hp.cache.kv_seq_len_list[l] = 131072
    hp.cache.streaming_kv_seq_len_list[l] = bench.SINK + bench.RECENT
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 20):
    hp.decode_stage(i, None)
torch.cuda.synchronize()
