"""Development aid: tests/test_gpu_fuzz.py::test_random_pairs_match_oracle for a list of seeds in ONE process, with per-seed environment overrides.
usage: python tools/fuzz_repro.py 184 185:PSK_GSI_STAGE=1 ..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa
import pyskani_amd as psk
from oracle import oracle
oracle.build()
import test_gpu_fuzz as T


class MP:
    def __init__(self): self.set = []
    def setenv(self, k, v): os.environ[k] = v; self.set.append(k)
    def undo(self):
        for k in self.set: os.environ.pop(k, None)


for arg in sys.argv[1:]:
    seed, *ov = arg.split(":")
    mp = MP()
    try:
        fn = T.test_random_pairs_match_oracle
        orig = mp.setenv
        if ov:      # overrides win over what the test sets
            over = dict(x.split("=") for x in ov)
            def setenv(k, v, _o=orig, _over=over): _o(k, _over.get(k, v))
            mp.setenv = setenv
            for k, v in over.items(): orig(k, v)
        fn.__wrapped__(psk, oracle, int(seed), mp) if hasattr(fn, "__wrapped__") else fn(psk, oracle, int(seed), mp)
        print("seed", seed, "ok", flush=True)
    finally:
        mp.undo()
