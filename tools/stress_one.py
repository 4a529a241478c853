"""Runs one case of tools/stress_parity.py (seed, index) and prints where the streams differ."""
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import stress_parity as sp  # noqa: E402

seed, idx = int(sys.argv[1]), int(sys.argv[2])
try:
    print("ok", sp.one_case(np.random.default_rng([seed, idx]), idx))
except AssertionError as e:
    print("FAIL", e)
