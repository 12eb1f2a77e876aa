import gzip
import os
import sys

import numpy as np
import pytest

# torch ships its own copy of the HIP runtime; it must initialise BEFORE libpyskani_amd.so brings in /opt/rocm's
# (the other order leaves torch with "No HIP GPUs are available"). Tests that use both in one process rely on this.
import torch  # noqa: F401,E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def load_fasta_first_record(name):
    """Same behaviour as the reference's tests/fasta.py parser for the first record (test_ani.py:16-19)."""
    seq, started = [], False
    with gzip.open(os.path.join(GOLDEN, name), "rt") as f:
        for line in f:
            if line.startswith(">"):
                if started:
                    break
                started = True
                continue
            if line.strip():
                seq.append(line.strip())
    return "".join(seq).encode("ascii")


@pytest.fixture(scope="session")
def ecoli():
    return load_fasta_first_record("e.coli-EC590.fasta.gz"), load_fasta_first_record("e.coli-K12.fasta.gz")


def random_genome(rng, length):
    return np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, length)].tobytes()


def mutate(rng, seq, rate, indel_rate=0.0):
    a = np.frombuffer(seq, dtype=np.uint8).copy()
    m = rng.random(len(a)) < rate
    alphabet = np.frombuffer(b"ACGT", dtype=np.uint8)
    a[m] = alphabet[rng.integers(0, 4, int(m.sum()))]
    if indel_rate > 0:
        keep = rng.random(len(a)) >= indel_rate
        a = a[keep]
    return a.tobytes()


@pytest.fixture(scope="session")
def oracle():
    from oracle import oracle as O
    O.build()
    return O
