#!/usr/bin/env python3
"""Randomised configurations of the own solver against the oracle on the GPU (a one-off stress run; a fixed-seed share of it is in the GPU suite,
tests/test_random_configs_gpu.py): dimension, size, depth, sweep counts, damping, mesh, pair threshold, fuse bits, recording on / off, precision,
Chebyshev, and what is done with the handle (solve | the bench's fixed-count loop in one or two calls | that, a reset and a solve) drawn at
random; iteration count and residual history must agree and u must be bit-identical.  The draw lives in tools/stress_solver_mock.py, which runs
it on the CPU over the host mock with the small sizes.  usage: stress_solver.py [count] [seed]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from stress_solver_mock import main

if __name__ == "__main__":
    main(mock=False)
