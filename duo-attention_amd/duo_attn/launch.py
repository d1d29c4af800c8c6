"""One process per GPU without asking the caller to type the launcher.

The multi-GPU paths of this package (layer pipeline ``duo_attn.pipeline``, head-parallel TP ``duo_attn.tp``) are one
rank per GPU over RCCL (reference: one process with accelerate hooks / ``tensor_parallel``, ``duo_attn/utils.py:206-283``).
``bench.py --gpus N`` and ``tools/benchmark_static.py --pp/--tp --gpus N`` are started like any single-GPU script;
when they find no rank environment they re-execute themselves under ``python -m torch.distributed.run`` on this node
(rendezvous on 127.0.0.1 — container hostnames do not resolve) and hand the children's exit code back.
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
from typing import List, Optional, Sequence

SHARED_GPU_ENV = "DUO_BENCH_DEBUG_SHARED_GPU"      # =1: every rank on cuda:0, gloo hand-off (one-GPU rehearsal, not a measurement)


def launched_by_torchrun(env=None) -> bool:
    env = os.environ if env is None else env
    return "WORLD_SIZE" in env and "RANK" in env


def free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def check_visible_gpus(nproc: int, visible: Optional[int] = None, env=None) -> None:
    """fewer GPUs than ranks is an error — unless the shared-GPU rehearsal mode is on"""
    env = os.environ if env is None else env
    if env.get(SHARED_GPU_ENV) == "1":
        return
    if visible is None:
        import torch

        visible = torch.cuda.device_count()
    if visible < nproc:
        raise SystemExit(f"--gpus {nproc} but {visible} GPU(s) visible: one rank per GPU (set {SHARED_GPU_ENV}=1 to rehearse "
                         f"the {nproc}-rank code path on one GPU over gloo — not a measurement mode)")


def torchrun_command(script: str, argv: Sequence[str], nproc: int, port: Optional[int] = None) -> List[str]:
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
            "--master-addr", "127.0.0.1", "--master-port", str(port if port is not None else free_port()),
            os.path.abspath(script), *argv]


def self_launch(script: str, argv: Sequence[str], nproc: int, visible: Optional[int] = None) -> int:
    """Re-execute ``script argv`` as ``nproc`` ranks under torch.distributed.run; stdout / stderr of the ranks pass
    through (rank 0 prints the result line), returns the launcher's exit code."""
    check_visible_gpus(nproc, visible)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL needs it on this driver stack
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("OMP_NUM_THREADS", "4")                # (torchrun would set 1 and say so on stderr)
    cmd = torchrun_command(script, argv, nproc)
    print(f"[duo_attn.launch] no rank environment: starting {nproc} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.run(cmd, env=env).returncode
