"""Re-run named cases of tests/fuzz_static_path.py (paste the dict printed after FAIL):  python tests/fuzz_repro.py "<dict>" ..."""
import ast
import sys

import fuzz_static_path as F

for s in sys.argv[1:]:
    c = ast.literal_eval(s)
    try:
        F.run_case(c)
        print("ok  ", c)
    except AssertionError as e:
        print("FAIL", c, "\n    ", str(e)[:500])
