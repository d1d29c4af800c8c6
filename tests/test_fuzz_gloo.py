"""Drawn cases of the multi-process fuzzers (tests/fuzz_pipeline_gloo.py, tests/fuzz_tp_gloo.py) kept as CPU tests: geometries
the fixed gloo tests do not have — three pipeline stages with two batch rows and a prompt the chunk does not divide; four
tensor-parallel ranks on a model that went through the enabler first, two batch rows, a layer without retrieval heads."""
import fuzz_pipeline_gloo as P
import fuzz_tp_gloo as T


def test_pipeline_three_stages_two_batch_rows_ragged_chunks():
    P.run_case(dict(world=3, heads=[[0.0, 1.0], [1.0, 1.0], [0.0, 0.0], [1.0, 0.0], [0.0, 1.0]], Hkv=2, group=2, B=2, sink=2, recent=12,
                    prompt=45, chunk=31, n_new=3, mode="drop_in", row_block=8, seed=20260926))


def test_pipeline_row_blocks_that_do_not_divide_the_chunk():
    P.run_case(dict(world=2, heads=[[1.0], [0.0], [1.0]], Hkv=1, group=2, B=2, sink=4, recent=6, prompt=59, chunk=31, n_new=2,
                    mode="row_blocks", row_block=13, seed=7))


def test_tp4_after_the_enabler_two_batch_rows():
    T.run_case(dict(world=4, Hkv=4, group=2, heads=[[1.0, 0.0, 1.0, 1.0], [0.0, 0.0, 0.0, 0.0], [0.0, 1.0, 0.0, 0.0]],
                    chunks=[19, 7, 1, 1], B=2, sink=2, recent=8, mode="patched_first", seed=11))
