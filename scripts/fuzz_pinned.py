#!/usr/bin/env python3
"""Randomised differential test of the PINNED path (solver::step semantics): GPU through the C ABI against the CPU
oracle on random scenes, body counts, time steps, substep counts, schedules (fused / per substep) and block sizes;
poses and the contact masks of every substep must match bit for bit.  (A fixed-seed selection of the same cases runs
inside `pytest -m gpu`: tests/test_gpu_fuzz.py.)
Usage (on a GPU box): python3 scripts/fuzz_pinned.py [--cases 60] [--seed 0]"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
from fuzz_cases import pinned_case  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", type=int, default=60)
    ap.add_argument("--seed", type=int, default=0)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    bad = 0
    for case in range(args.cases):
        ok, what = pinned_case(rng)
        print("case %3d %s: %s" % (case, what, "ok" if ok else "MISMATCH"), flush=True)
        bad += 0 if ok else 1
    print("mismatches:", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
