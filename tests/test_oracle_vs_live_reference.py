"""Build container only: a 400-case slice (about 20 s on an idle machine) of tests/golden/fuzz_against_reference.py — the oracle and the product's host
path (oracle backend) against the reference's OWN static / tuple forwards, decoder-layer forward, whole models through its enablers, the INT4
demo cache class, cache and utilities on freshly drawn cases.
Skipped where /root/reference does not exist (the GPU box); nothing under -m gpu, smoke() or bench.py reads the reference."""
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir("/root/reference/duo_attn"), reason="the reference is only present in the build container")
def test_oracle_and_host_path_equal_the_live_reference_on_drawn_cases():
    env = {k: v for k, v in os.environ.items() if k != "PYTHONPATH"}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "fuzz_against_reference.py"), "--cases", "400",
                        "--seed", "12"], capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-3000:], r.stderr[-2000:])
    m = re.search(r"(\d+) cases \(\{'static': (\d+), 'tuple': (\d+), 'utils': (\d+), 'layer': (\d+), 'model': (\d+), 'int4': (\d+)\}\).* (\d+) failed", r.stdout)
    assert m, r.stdout[-2000:]
    n, n_static, n_tuple, n_utils, n_layer, n_model, n_int4, bad = (int(x) for x in m.groups())
    assert bad == 0 and min(n_static, n_tuple) >= 20 and n_utils >= 5 and min(n_layer, n_model, n_int4) >= 10, r.stdout[-500:]
