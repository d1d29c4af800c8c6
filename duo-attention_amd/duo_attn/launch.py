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


# ----------------------------------------------------------------------------------------------------------------------
# to_device(model, [gpu ids], enable_tp / enable_pp) from ONE python process, as the reference's harnesses call it
# ----------------------------------------------------------------------------------------------------------------------
SELF_LAUNCHED_ENV = "DUO_ATTN_SELF_LAUNCHED"      # set for ranks started by relaunch_under_torchrun()


def original_command(orig_argv: Optional[Sequence[str]] = None) -> List[str]:
    """What follows the interpreter (and its own options) on the command line this process was started with, in the form
    ``torch.distributed.run`` takes: ``[script, args...]`` or ``["-m", module, args...]``.  ``python -c`` / an interactive
    session cannot be started again: ValueError."""
    a = list(sys.orig_argv if orig_argv is None else orig_argv)[1:]
    i = 0
    while i < len(a):
        t = a[i]
        if t == "-m":
            if i + 1 >= len(a):
                break
            return ["-m", *a[i + 1:]]
        if t == "-c" or t == "-":
            break
        if t in ("-X", "-W"):           # interpreter options with a value
            i += 2
            continue
        if t.startswith("-"):
            i += 1
            continue
        return [os.path.abspath(t), *a[i + 1:]]
    raise ValueError("this process was not started from a script or a module (python -c / interactive): it cannot be started "
                     "again as one rank per GPU — launch it with torch.distributed.run")


def relaunch_under_torchrun(nproc: int, visible: Optional[int] = None) -> int:
    """Start the command line of THIS process again as ``nproc`` ranks under torch.distributed.run (rendezvous on 127.0.0.1)
    and return the launcher's exit code.  The ranks see ``DUO_ATTN_SELF_LAUNCHED=1``."""
    check_visible_gpus(nproc, visible)
    target = original_command()
    env = dict(os.environ)
    env[SELF_LAUNCHED_ENV] = "1"
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), *target]
    print(f"[duo_attn.launch] to_device() got {nproc} devices in a single process: one rank per GPU — starting this command "
          f"again as {nproc} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    sys.stdout.flush()
    return subprocess.run(cmd, env=env).returncode


def quiet_this_rank() -> None:
    """Ranks other than 0 of a self-launched harness compute the same tokens as rank 0 (tensor parallelism: identical logits
    after the all-reduce; layer pipeline: the decode logits are broadcast) and would write the same result files and print
    the same lines: their stdout goes to os.devnull, every file they open FOR WRITING is os.devnull and they create no
    directories, so rank 0 alone writes results — without an edit to the harness."""
    import builtins

    sys.stdout = open(os.devnull, "w")
    real_open = builtins.open

    def rank_open(file, mode="r", *a, **kw):
        if isinstance(mode, str) and any(c in mode for c in "wax+") and not isinstance(file, int):
            return real_open(os.devnull, mode.replace("x", "w"), *a, **kw)
        return real_open(file, mode, *a, **kw)

    builtins.open = rank_open
    # ... and they create no directories (`if not os.path.exists(d): os.makedirs(d)` on two ranks at once is a race)
    os.makedirs = lambda *a, **kw: None
    os.mkdir = lambda *a, **kw: None


class _AtomicWrite:
    """file object of ``open(path, "w")`` on rank 0 of a self-launched harness: the bytes go to a temporary file next to
    ``path`` that replaces it on close — another rank that lists the directory and reads every result file it finds (the
    needle harness's ``result_exists``, eval/needle/needle_in_haystack.py:380-397) never sees a half-written one."""

    def __init__(self, real_open, path, mode, a, kw):
        self._path = os.fspath(path)
        self._tmp = f"{self._path}.tmp-rank0-{os.getpid()}"
        self._f = real_open(self._tmp, mode, *a, **kw)

    def __getattr__(self, name):
        return getattr(self._f, name)

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __iter__(self):
        return iter(self._f)

    def close(self):
        if not self._f.closed:
            self._f.close()
            os.replace(self._tmp, self._path)


def atomic_writes_on_this_rank() -> None:
    import builtins

    real_open = builtins.open

    def rank0_open(file, mode="r", *a, **kw):
        if mode in ("w", "wt", "wb") and isinstance(file, (str, os.PathLike)) and os.fspath(file) != os.devnull:
            return _AtomicWrite(real_open, file, mode, a, kw)
        return real_open(file, mode, *a, **kw)

    builtins.open = rank0_open


def ensure_ranks(n_devices: int, what: str) -> None:
    """Called by ``duo_attn.utils.to_device`` with a device list longer than one and no process group.

    * started by torch.distributed.run (rank environment present): initialise the group — RCCL ("nccl") with one GPU per rank,
      gloo in the shared-GPU rehearsal mode (``DUO_BENCH_DEBUG_SHARED_GPU=1``) or without a GPU;
    * a plain ``python harness.py`` — how the reference's scripts/niah.sh and scripts/longbench.sh start their harnesses, which
      then hand every visible GPU to ``to_device(..., enable_tp=True)`` inside that one process (eval/needle/
      needle_in_haystack.py:213-214, eval/LongBench/pred.py:237-243): the command line is started again as one rank per
      GPU and THIS process exits with the ranks' exit code — everything the harness did before the call (argument parsing,
      loading the checkpoint) is repeated by every rank, everything after it is done by the ranks only."""
    import torch
    import torch.distributed as dist

    if dist.is_initialized():
        return
    if not launched_by_torchrun():
        if os.environ.get(SELF_LAUNCHED_ENV) == "1":
            raise RuntimeError("a self-launched rank without a rank environment")
        try:
            rc = relaunch_under_torchrun(n_devices)
        except ValueError as e:
            raise RuntimeError(f"{what} = one process per GPU: {e}") from e
        except SystemExit as e:        # (check_visible_gpus: fewer GPUs than devices asked for)
            raise RuntimeError(f"{what} = one process per GPU: {e}") from None
        sys.exit(rc)
    world = int(os.environ["WORLD_SIZE"])
    if world != n_devices:
        raise ValueError(f"{n_devices} devices for {world} ranks")
    shared = os.environ.get(SHARED_GPU_ENV) == "1" or not torch.cuda.is_available()
    if shared:
        dist.init_process_group("gloo")
    else:
        local = int(os.environ.get("LOCAL_RANK", os.environ["RANK"]))
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    if os.environ.get(SELF_LAUNCHED_ENV) == "1":
        if dist.get_rank() != 0:
            quiet_this_rank()
        else:
            atomic_writes_on_this_rank()
