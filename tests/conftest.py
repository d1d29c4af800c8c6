import os
import sys

import pytest

# The suite runs with the product's defaults: since round 6 the automatic HIP-graph capture of the reference's decode loop
# is OPT-IN (duo_attn/graph.py; DUO_AUTO_DECODE_GRAPH=1), so model-level, fuzz, sharded and full-size runs step eagerly — the
# path users get.  The capture keeps its own coverage: tests/test_auto_graph_gpu.py and tests/fuzz_model_decode.py switch it
# on explicitly and compare it with the eager loop bit for bit.  An inherited environment setting must not change that.
os.environ.pop("DUO_AUTO_DECODE_GRAPH", None)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "duo-attention_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _has_gpu():
    try:
        import torch

        return torch.cuda.is_available()
    except Exception:
        return False


def pytest_collection_modifyitems(config, items):
    if _has_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU visible")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)


@pytest.fixture
def eager_decode_steps(monkeypatch):
    """every decode step re-enters Python (no automatic graph capture): for tests that count launches per step"""
    from duo_attn import graph

    monkeypatch.setattr(graph, "AUTO_DECODE_GRAPH", False)
    yield


@pytest.fixture
def oracle_backend():
    """Plug the CPU oracle in as the device backend for host-plumbing tests."""
    from duo_attn import backend
    from oracle.duo_oracle import OracleBackend

    backend._set_backend_for_testing(OracleBackend())
    yield
    backend._set_backend_for_testing(None)


def pytest_sessionfinish(session, exitstatus):
    """GPU runs leave the MEASURED parity errors (not just pass/fail) in gpurun_out/parity_report.json."""
    try:
        import json

        from helpers import PARITY_LOG
    except Exception:
        return
    if not PARITY_LOG or not _has_gpu():
        return
    out_dir = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out_dir, exist_ok=True)
    with open(os.path.join(out_dir, "parity_report.json"), "w") as f:
        json.dump({"bar": "helpers.attn_close: elementwise tol + rms(err) <= 2.5e-3 rms(ref)", "cases": PARITY_LOG},
                  f, indent=1, sort_keys=True)
