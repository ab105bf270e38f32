#!/usr/bin/env python3
"""Adversarial parity run (longer version of tests/test_adversarial_gpu.py; cases in tests/adversarial.py).
`stress_adversarial.py [rounds]`, SEED and DP_FORM (KVM_OPT_DP_KERNEL) in the environment."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "kv-match_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import kvmatch
from kvmatch import datagen
from oracle import oracle
import adversarial
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
ctx = kvmatch.Phase2Context(0)
ctx.set_option(kvmatch.OPT_DP_KERNEL, int(os.environ.get("DP_FORM", "0")))   # (3: every DTW list through the packed DP)
total, bad, quirk = adversarial.run(kvmatch, datagen, oracle, ctx, rounds, int(os.environ.get("SEED", "11")),
                                    log=lambda s: print(s, flush=True))
print("verifications", total, "mismatches", bad, "DTW verifications through the sequential tie path", quirk)
sys.exit(1 if bad else 0)
