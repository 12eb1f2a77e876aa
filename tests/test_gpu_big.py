"""GPU: BASELINE configs[4] at reduced count — mammalian-scale genomes (24 contigs; 2 genomes instead of 50) sketched and
chained pairwise, every integer of the chain against the oracle. Default size 1 Gb per genome
(> 2^18 seeds, > 4 096 chunks, the group selection: ~1-2 min, mostly the CPU oracle; PSK_BIG_MB=300 for a quick pass);
PSK_BIG_MB=3000 runs the real 3 Gb shape (24 x 125 Mb contigs, ~24 M seeds per genome; minutes, mostly the CPU oracle)."""
import os
import time

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
INT_FIELDS = ("n_anchors", "n_chunks", "n_intervals", "covered_query", "covered_ref", "sum_chain_anchors", "sum_chunk_seeds")


def big_pair(total_mb, n_contigs=24, divergence=0.01, seed=5):
    """Two genomes of n_contigs equal contigs: an iid ancestor and a copy with substitutions, two contigs reversed-
    complemented and the contig order rotated (strands and cross-contig chains are exercised). Built on the GPU."""
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev); g.manual_seed(seed)
    L = total_mb * 1_000_000 // n_contigs
    lut = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=dev)
    a_contigs, b_contigs = [], []
    for i in range(n_contigs):
        a = torch.randint(0, 4, (L,), generator=g, device=dev, dtype=torch.uint8)
        mut = torch.rand((L,), generator=g, device=dev) < divergence
        b = torch.where(mut, (a + torch.randint(1, 4, (L,), generator=g, device=dev, dtype=torch.uint8)) & 3, a)
        if i in (3, 17):
            b = (3 - b).flip(0)
        a_contigs.append(lut[a.long()].cpu().numpy().tobytes())
        b_contigs.append(lut[b.long()].cpu().numpy().tobytes())
        del a, b, mut
    b_contigs = b_contigs[5:] + b_contigs[:5]
    return a_contigs, b_contigs


def test_mammalian_scale_pair_matches_oracle(oracle):
    import pyskani_amd as psk
    mb = int(os.environ.get("PSK_BIG_MB", "1000"))
    ref, qry = big_pair(mb)
    t0 = time.time()
    db = psk.Database()
    db.sketch("ref", *ref)
    t1 = time.time()
    hits = db.query("qry", *qry, learned_ani=False)
    t2 = time.time()
    hits_m = db.query("qry", *qry, learned_ani=False, median=True)
    hits_r = db.query("qry", *qry, learned_ani=False, robust=True)
    t3 = time.time()
    print(f"\n[{mb} Mb x 2] sketch ref {t1 - t0:.2f} s, sketch query + chain {t2 - t1:.2f} s, median + robust {t3 - t2:.2f} s")
    osr, osq = oracle.Sketch(ref), oracle.Sketch(qry)
    gs, gm = db._sketch("q", qry, True).export()
    assert len(gs) == len(osq.seeds)
    for f in ("kmer", "pos", "contig", "canon"):
        assert np.array_equal(gs[f], osq.seeds[f]), f
    assert np.array_equal(gm, osq.markers)
    for got, kw in ((hits, {}), (hits_m, {"median": True}), (hits_r, {"robust": True})):
        want = oracle.chain(osr, osq, **kw)
        assert len(got) == 1
        for f in INT_FIELDS:
            assert got[0]._raw[f] == getattr(want, f), (kw, f, got[0]._raw[f], getattr(want, f))
        assert abs(got[0].identity - want.ani) < 1e-6 and abs(got[0].query_fraction - want.af_query) < 1e-6
        assert abs(got[0]._raw["ani_std"] - want.ani_std) < 1e-6
    assert abs(hits[0].identity - 0.99) < 0.002 and hits[0].query_fraction > 0.9
    if mb >= 1000:
        assert hits[0]._raw["n_chunks"] > 4096 and len(gs) > (1 << 18)      # beyond the LDS paths of select / pair_reduce / index_block


def test_group_selection_matches_solo_and_serial():
    """Pairs with more than 1 024 candidate chains: the one-workgroup selection, the cooperative several-workgroups-per-pair
    selection (forced down to small pairs by PSK_BIG_SOLO; by default it takes pairs with more than 32 768 candidates) and the
    lane-serial transliteration of the oracle (PSK_CHAIN_SERIAL) must give the same integers. 60 Mb genomes with an inserted
    repeat family so that a few hundred chains conflict on the reference."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys; sys.path.insert(0, %r)\n"
        "import numpy as np, torch\n"
        "import pyskani_amd as psk\n"
        "rng = np.random.default_rng(11)\n"
        "L = 60_000_000\n"
        "a = rng.integers(0, 4, L, dtype=np.uint8)\n"
        "rep = rng.integers(0, 4, 30_000, dtype=np.uint8)\n"
        "for s in rng.integers(0, L - 30_000, 60): a[s:s + 30_000] = rep\n"
        "b = a.copy(); m = rng.random(L) < 0.01; b[m] = (b[m] + rng.integers(1, 4, int(m.sum()), dtype=np.uint8)) & 3\n"
        "lut = np.frombuffer(b'ACGT', np.uint8)\n"
        "cut = [0, 9_000_000, 25_000_000, 41_000_000, L]\n"
        "ref = [lut[a[cut[i]:cut[i + 1]]].tobytes() for i in range(4)]\n"
        "qry = [lut[b[cut[i]:cut[i + 1]]].tobytes() for i in (2, 0, 3, 1)]\n"
        "db = psk.Database(); db.sketch('ref', *ref)\n"
        "for kw in ({}, {'median': True}):\n"
        "    h = db.query('qry', *qry, learned_ani=False, **kw)[0]\n"
        "    print(h._raw['n_chunks'], h._raw['n_intervals'], h._raw['covered_query'], h._raw['covered_ref'], h._raw['sum_chain_anchors'], h._raw['sum_chunk_seeds'], repr(h.identity))\n"
    ) % (root,)
    outs = {}
    # "few_slots": the group selection on a device that reports 6 co-resident workgroups (a small partition, a CU mask): the occupancy
    # query must size the launch down to one workgroup per pair instead of spinning at a barrier nobody else reaches
    for name, extra in (("solo", {}), ("group", {"PSK_BIG_SOLO": "1024"}), ("few_slots", {"PSK_BIG_SOLO": "1024", "PSK_HUGE_SLOTS": "6"}), ("xtrees", {"PSK_LANE_XTREES": "1"}), ("serial", {"PSK_CHAIN_SERIAL": "1"})):
        env = dict(os.environ)
        for k in ("PSK_BIG_SOLO", "PSK_CHAIN_SERIAL", "PSK_HUGE_SLOTS", "PSK_LANE_XTREES"):
            env.pop(k, None)
        env.update(extra)
        outs[name] = subprocess.check_output([sys.executable, "-c", code], env=env, timeout=900).decode().strip()
    assert len(set(outs.values())) == 1, outs
    assert int(outs["solo"].split()[1]) > 1024
