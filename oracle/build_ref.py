"""Build the reference's OWN INT4 kernels (demo/quantize_int4.cu) for gfx950 -> oracle/_ref/*.so.

TEST INFRASTRUCTURE ONLY.  Nothing in the product imports this or anything it produces.

The reference JIT-loads that file with ``torch.utils.cpp_extension.load(..., extra_cuda_cflags=
["--use_fast_math"])`` (demo/int4_kv.py:46-56).  PyTorch-ROCm hipifies `.cu` sources on load, so the
same call works here with two adjustments that do not touch the source text:

* ``--use_fast_math`` is an nvcc spelling clang rejects; its hipcc translation is ``-ffast-math``.
* ``-D__restrict__=``: the kernel declares ``const int64_t __restrict__ stride_batch`` (quantize_int4.cu:
  84-86); nvcc ignores ``restrict`` on a non-pointer, clang makes it an error.  Defining the (purely
  advisory) qualifier away is the smallest change that compiles the file unmodified.

The source is read where it lies under /root/reference; hipify wants to write its translation next to
the source, so the build works on a scratch COPY under a temporary directory (never in this repo, never in
/root/reference) and only the resulting shared objects land in ``oracle/_ref/`` (git-ignored; they
travel to the GPU box with the snapshot, like the product's own .so).

Three builds, because the arithmetic that decides bit-exactness is chosen by compiler flags, not by the
source (looked at in the gfx950 ISA of each build):

  name        flags                 division (x - zero) / scale      dequant  half(q)*s + z
  ----------  --------------------  -------------------------------  -----------------------------
  nocontract  -ffp-contract=off     IEEE  (v_div_scale/fmas/fixup)   v_pk_mul_f16 ; v_pk_add_f16   <- the source as written
  default     (hipcc defaults)      IEEE                             v_pk_fma_f16 (one rounding)
  fast        -ffast-math           x * v_rcp_f32(scale) (approx)    v_pk_fma_f16 (one rounding)   <- "--use_fast_math"

tests/golden/make_int4_golden.py runs all three on a GPU and records their outputs.
"""
import os
import shutil
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SRC = "/root/reference/demo/quantize_int4.cu"
OUT_DIR = os.path.join(ROOT, "oracle", "_ref")

VARIANTS = {
    "nocontract": ["-ffp-contract=off"],
    "default": [],
    "fast": ["-ffast-math"],
}


def module_name(variant: str) -> str:
    return f"quantize_int4_ref_{variant}"


def build(force: bool = False, verbose: bool = False) -> bool:
    """Returns True when oracle/_ref holds all three modules afterwards; False (and does nothing) when the
    reference tree is not present (the GPU box: it only uses the prebuilt files)."""
    want = [os.path.join(OUT_DIR, module_name(v) + ".so") for v in VARIANTS]
    if not force and all(os.path.exists(p) for p in want):
        return True
    if not os.path.exists(REF_SRC):
        return all(os.path.exists(p) for p in want)
    os.environ.setdefault("PYTORCH_ROCM_ARCH", "gfx950")
    from torch.utils.cpp_extension import load

    os.makedirs(OUT_DIR, exist_ok=True)
    with tempfile.TemporaryDirectory(prefix="duo_ref_build_") as tmp:
        src = os.path.join(tmp, "quantize_int4.cu")
        shutil.copy(REF_SRC, src)
        for variant, flags in VARIANTS.items():
            bd = os.path.join(tmp, variant)
            os.makedirs(bd)
            load(name=module_name(variant), sources=[src], extra_cuda_cflags=flags + ["-D__restrict__="],
                 build_directory=bd, is_python_module=False, verbose=verbose)
            shutil.copy(os.path.join(bd, module_name(variant) + ".so"), OUT_DIR)
    return True


def load_ref(variant: str):
    """Import a prebuilt module (needs ``import torch`` first: it links libtorch / libc10_hip)."""
    import importlib.machinery
    import importlib.util

    import torch  # noqa: F401

    name = module_name(variant)
    path = os.path.join(OUT_DIR, name + ".so")
    if not os.path.exists(path):
        raise FileNotFoundError(f"{path}: run `python oracle/build_ref.py` in the build container first")
    loader = importlib.machinery.ExtensionFileLoader(name, path)
    spec = importlib.util.spec_from_loader(name, loader)
    mod = importlib.util.module_from_spec(spec)
    loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    ok = build(force="--force" in sys.argv, verbose="-v" in sys.argv)
    print("oracle/_ref:", sorted(os.listdir(OUT_DIR)) if ok else "reference tree absent, nothing built")
