"""GPU: the paths that only large batches take - one-workgroup-per-pair emit with the chunk table (>= 1 024 mid-sized pairs),
XCD turns in the join, two-tier selection, 2^29-seed batches, the seed prefilter of rescued contigs (>= 2^20 pairs) - against
the same batch with each of them switched off. Every hit (reference, all chaining integers, ANI, AF) must be identical.
(The small-batch paths are held to the oracle in test_gpu_parity / test_gpu_fuzz; these runs are held to those paths, and a random
sample of the large all-vs-all batch's hits is held to the oracle directly: test_large_batch_sample_matches_oracle.)"""
import hashlib
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

COMMON = r"""
import sys, hashlib
sys.path.insert(0, %r)
import numpy as np
import pyskani_amd as psk
lut = np.frombuffer(b"ACGT", np.uint8)
rng = np.random.default_rng(77)
def mutate(a, d):
    b = a.copy(); m = rng.random(len(a)) < d; b[m] = (b[m] + rng.integers(1, 4, int(m.sum()), dtype=np.uint8)) & 3; return b
def digest(hit_lists):
    h = hashlib.sha256(); n = 0
    for hs in hit_lists:
        for x in hs:
            r = x._raw
            h.update(repr((x.reference_name, int(r["n_anchors"]), int(r["n_chunks"]), int(r["n_intervals"]), int(r["covered_query"]), int(r["covered_ref"]),
                           int(r["sum_chain_anchors"]), int(r["sum_chunk_seeds"]), float(r["ani"]), float(r["af_query"]), float(r["af_ref"]), float(r["ani_std"]))).encode())
            n += 1
    return n, h.hexdigest()
""" % (ROOT,)

ALL_VS_ALL = COMMON + r"""
anc = [rng.integers(0, 4, 600_000, dtype=np.uint8) for _ in range(8)]
genomes = [(f"g{f}_{j}", lut[mutate(anc[f], 0.002 * j)].tobytes()) for f in range(8) for j in range(40)]      # 320 genomes, families of 40
db = psk.Database()
db.sketch_many(genomes)
print(*digest(db.query_many(genomes, learned_ani=False)))
"""

RESCUE = COMMON + r"""
anc = [rng.integers(0, 4, 400_000, dtype=np.uint8) for _ in range(10)]
refs = [(f"r{f}_{j}", lut[mutate(anc[f], 0.003 * j)].tobytes()) for f in range(10) for j in range(30)]        # 300 references
contigs = []
for i in range(4800):
    a = anc[i % 10]; L = int(rng.integers(1200, 3500)) if i % 4 else int(rng.integers(6000, 12000)); st = int(rng.integers(0, len(a) - L))
    contigs.append((f"c{i}", lut[mutate(a[st:st + L], rng.uniform(0, 0.04))].tobytes()))
db = psk.Database(compression=30, marker_compression=200)
db.sketch_many(refs)
print(*digest(db.query_many(contigs, learned_ani=False)))
"""


def _run(code, extra):
    env = dict(os.environ)
    for k in ("PSK_EMIT_PAIRS", "PSK_EMIT_HEADS", "PSK_XCD_GROUP", "PSK_BATCH_ITEMS_LOG2", "PSK_PREFILTER", "PSK_JOIN_PAIRS", "PSK_CHUNK_HOPS", "PSK_PROBE", "PSK_CHAIN_QUAD_DEEP", "PSK_SELECT_TINY", "PSK_CHAIN_WAVE_REG", "PSK_ROW_SORT", "PSK_ROUND_QUERIES", "PSK_REDUCE_TINY", "PSK_PROBE_LOCAL", "PSK_REDUCE_SMALL", "PSK_GSI_JOIN", "PSK_GSI_ONEPASS", "PSK_DP_PRUNE", "PSK_GSI_SLICE", "PSK_GSL_STAGE", "PSK_GSL_MAX_BLOCKS", "PSK_BSI_SMALL", "PSK_GSI_STAGE", "PSK_PIPELINE", "PSK_BIG_SOLO", "PSK_HUGE_MIN_SEEDS", "PSK_HUGE_SLOTS"):
        env.pop(k, None)
    env.update(extra)
    out = subprocess.check_output([sys.executable, "-c", code], env=env, timeout=900).decode().split()
    return int(out[0]), out[1]


def test_all_vs_all_batch_paths_agree():
    base = _run(ALL_VS_ALL, {})
    assert base[0] >= 320 * 40                      # every genome finds its family: >= 1 024 chained pairs in one batch
    # (switches that act on different stages share a run: every run is a process that loads the library and sketches the genomes again)
    # (the default joins such a batch through the database-wide seed index, one wave per (query, slice of 512 seeds): slice_join.hip; PSK_GSI_SLICE=0 = the per-pair
    # merge join + per-pair emit it replaced, which stays the route of databases that cannot have the index - its switches only act there)
    # (PSK_PIPELINE=1: alternate batches on a second lane, two in flight - the default of rounds of >= 2^31 (pair, seed) items - here over nine batches, and over one)
    for extra in ({"PSK_GSI_SLICE": "0"}, {"PSK_GSL_STAGE": "0", "PSK_BATCH_ITEMS_LOG2": "22"}, {"PSK_PIPELINE": "1", "PSK_BATCH_ITEMS_LOG2": "22"}, {"PSK_PIPELINE": "1"}, {"PSK_PIPELINE": "0", "PSK_BATCH_ITEMS_LOG2": "23"}, {"PSK_GSI_SLICE": "0", "PSK_EMIT_PAIRS": "0"}, {"PSK_GSI_SLICE": "0", "PSK_EMIT_HEADS": "0", "PSK_XCD_GROUP": "0"},
                  {"PSK_GSI_SLICE": "0", "PSK_BATCH_ITEMS_LOG2": "22", "PSK_ROW_SORT": "0"}, {"PSK_GSI_SLICE": "0", "PSK_CHUNK_HOPS": "1", "PSK_REDUCE_SMALL": "0"}):
        assert _run(ALL_VS_ALL, extra) == base, extra


ALL_VS_ALL_CALLERS = COMMON + r"""
import threading
anc = [rng.integers(0, 4, 600_000, dtype=np.uint8) for _ in range(8)]
genomes = [(f"g{f}_{j}", lut[mutate(anc[f], 0.002 * j)].tobytes()) for f in range(8) for j in range(40)]
db = psk.Database()
db.sketch_many(genomes)
out = [None] * 5
def run(i): out[i] = digest(db.query_many(genomes, learned_ani=False))
th = [threading.Thread(target=run, args=(i,)) for i in range(5)]
for t in th: t.start()
for t in th: t.join()
assert all(o == out[0] for o in out), out
print(*out[0])
"""


def test_switches_are_read_once_per_call_not_once_per_process():
    """profiles/r6/paths.md: the query path's PSK_* switches are read into a struct when a call begins (csrc/query_parts.h Switches) - a test, or bench.py, may flip them between
    two calls of ONE process. The same all-vs-all three times in one process: index joins off (no index lookups counted), on by default (lookups counted), off again - the same hits."""
    code = COMMON + r"""
import ctypes as C, os
anc = [rng.integers(0, 4, 300_000, dtype=np.uint8) for _ in range(4)]
genomes = [(f"g{f}_{j}", lut[mutate(anc[f], 0.002 * j)].tobytes()) for f in range(4) for j in range(40)]      # 160 genomes x 40 relatives: 6 400 pairs of ~2 400 seeds
db = psk.Database()
db.sketch_many(genomes)
def lookups():
    lk = C.c_uint64(); db._lib.psk_ctx_join_work(db._ctx._h, C.byref(lk), None, None, None, 1); return lk.value
out = []
for setting in ("0", None, "0"):
    if setting is None: os.environ.pop("PSK_GSI_SLICE", None); os.environ.pop("PSK_GSI_JOIN", None)
    else: os.environ["PSK_GSI_SLICE"] = setting; os.environ["PSK_GSI_JOIN"] = setting
    lookups()
    d = digest(db.query_many(genomes, learned_ani=False))
    out.append((d, lookups()))
assert out[0][0] == out[1][0] == out[2][0], out
assert out[0][1] == 0 and out[1][1] > 0 and out[2][1] == 0, out
print(*out[0][0])
"""
    env = {k: v for k, v in os.environ.items() if not k.startswith("PSK_")}
    n, _ = subprocess.check_output([sys.executable, "-c", code], env=env, timeout=900).decode().split()
    assert int(n) >= 160 * 40


def test_callers_that_each_want_a_second_lane():
    """Five concurrent query_many callers whose rounds keep two batches in flight (PSK_PIPELINE=1, nine batches each): eight lanes for ten wishes - a caller that
    finds no free lane runs one chain and never waits for one - and every caller gets the single caller's hits."""
    base = _run(ALL_VS_ALL, {})
    assert _run(ALL_VS_ALL_CALLERS, {"PSK_PIPELINE": "1", "PSK_BATCH_ITEMS_LOG2": "22"}) == base


def test_two_lanes_whose_batches_both_take_the_full_size_selection_launch():
    """a batch that holds a Gb-scale pair takes the device's one full-size group-selection launch under a host mutex (huge_mu) inside chain_run; the helper
    threads of the two-batches-in-flight mode never gave it back, so the second such batch - on the other lane - waited forever. Forced here on small pairs
    (PSK_HUGE_MIN_SEEDS=1: every batch counts as Gb-scale; PSK_BIG_SOLO at its floor: the cooperative launch is always made), nine batches over two lanes: the same hits,
    and the call returns (the subprocess has a timeout; the deadlock was on the host, not on the GPU)."""
    base = _run(ALL_VS_ALL, {})
    assert _run(ALL_VS_ALL, {"PSK_PIPELINE": "1", "PSK_BATCH_ITEMS_LOG2": "22", "PSK_BIG_SOLO": "1024", "PSK_HUGE_MIN_SEEDS": "1"}) == base
    assert _run(ALL_VS_ALL_CALLERS, {"PSK_PIPELINE": "1", "PSK_BATCH_ITEMS_LOG2": "22", "PSK_BIG_SOLO": "1024", "PSK_HUGE_MIN_SEEDS": "1"}) == base


ONE_FAMILY = COMMON + r"""
anc = rng.integers(0, 4, 90_000, dtype=np.uint8)
rep = rng.integers(0, 4, 2_500, dtype=np.uint8)
def member(j):
    a = mutate(anc, 0.0004 * j)
    if j % 3 == 0: a = np.concatenate([a[:30_000], rep, a[30_000:60_000], mutate(rep, 0.01), a[60_000:]])      # a repeat: seeds with two anchors in one pair
    parts = [a] if j % 4 else [a[:41_000], a[41_000:41_600], a[41_600:]]                                       # every fourth genome in three contigs (one of them a single chunk)
    return [lut[x].tobytes() for x in parts]
genomes = [(f"m{j}", member(j)) for j in range(300)]      # ONE family of 300: every query passes against all 300 references - two entries (256 + 44 pairs) per query
db = psk.Database(compression=30, marker_compression=200)
db.sketch_many([(n, *c) for n, c in genomes])
res = db.query_many([(n, *c) for n, c in genomes], learned_ani=False)
print(*digest(res))
"""


INTERLEAVED = COMMON + r"""
anc = [rng.integers(0, 4, 50_000, dtype=np.uint8) for _ in range(3)]
genomes = [(f"i{j}", lut[mutate(anc[j % 3], 0.0002 * (j // 3))].tobytes()) for j in range(700)]      # three families dealt out in turn: a query's ~233 relatives sit in all three index blocks
db = psk.Database(compression=30, marker_compression=200)
db.sketch_many(genomes)
print(*digest(db.query_many(genomes[::7], learned_ani=False)))
"""


def test_slice_join_walks_every_index_block_that_holds_a_passing_reference():
    """The slice join's seed index comes in blocks of 256 references and a query walks the blocks that hold one of its passing references. Here every query's relatives
    are spread over all three blocks (the plan still takes the slice join: three blocks per query), each pair's anchors come from the walk of its reference's block;
    with PSK_GSL_MAX_BLOCKS=1 the plan refuses (too many blocks per query) and the per-pair join runs: the same hits."""
    base = _run(INTERLEAVED, {})
    assert base[0] > 100 * 150
    assert _run(INTERLEAVED, {"PSK_GSI_SLICE": "0"}) == base
    assert _run(INTERLEAVED, {"PSK_GSL_MAX_BLOCKS": "1"}) == base


SCATTERED = COMMON + r"""
import os, ctypes as C
# 47 index blocks of 256 references; family f = the references f, 256 + f, 512 + f, ...: ONE member per block, so a query's 47 passing references sit in 47 different blocks
NB, FAMS, L = 47, 256, 21000
def near(a, n_mut):
    b = a.copy(); p = rng.integers(0, len(a), n_mut); b[p] = (b[p] + rng.integers(1, 4, n_mut, dtype=np.uint8)) & 3; return b
anc = [rng.integers(0, 4, L, dtype=np.uint8) for _ in range(FAMS)]
refs = [(f"r{b * FAMS + f}", lut[near(anc[f], 8 * b)].tobytes()) for b in range(NB) for f in range(FAMS)]
db = psk.Database(compression=10, marker_compression=100)
db.sketch_many(refs)
genomes = [refs[i] for i in range(0, NB * FAMS, 50)]      # 241 genome queries x 47 passing references
contigs = []
for j in range(3000):
    i = int(rng.integers(0, NB * FAMS)); g = np.frombuffer(refs[i][1], np.uint8); ln = int(rng.integers(1500, 4000)); st = int(rng.integers(0, len(g) - ln))
    contigs.append((f"c{j}", g[st:st + ln].tobytes()))
ng, dg = digest(db.query_many(genomes, learned_ani=False))
nc, dc = digest(db.query_many(contigs, learned_ani=False))
lk = C.c_uint64(); db._lib.psk_ctx_join_work(db._ctx._h, C.byref(lk), None, None, None, 0)
print(ng, nc, dg + dc, lk.value)
"""


def test_block_tables_of_entries_whose_references_sit_in_dozens_of_blocks():
    """Round 6: the index walks take their blocks from per-entry block tables (gsl_blocks_kernel; up to one row per pair of the entry). Here a query's 47 passing references sit in 47
    DIFFERENT index blocks - the plan refuses the index joins for such a round (more than four blocks per query: the per-pair join for genomes, the database-wide index for
    contigs) unless told otherwise: with PSK_GSI_SLICE=1 / PSK_GSL_MAX_BLOCKS=64 the slice join and the contig join walk 47 listed blocks per entry. The same hits."""
    def run(extra):
        env = {k: v for k, v in os.environ.items() if not k.startswith("PSK_")}
        env.update(extra)
        o = subprocess.check_output([sys.executable, "-c", SCATTERED], env=env, timeout=1200).decode().split()
        return int(o[0]), int(o[1]), o[2], int(o[3])
    base = run({"PSK_GSI_SLICE": "0", "PSK_GSI_JOIN": "0"})      # no index walk at all
    assert base[0] >= 241 * 40 and base[1] >= 3000 * 30 and base[3] == 0, base
    forced = run({"PSK_GSI_SLICE": "1", "PSK_GSL_MAX_BLOCKS": "64"})
    assert forced[:3] == base[:3] and forced[3] > 0, (forced, base)
    default = run({})
    assert default[:3] == base[:3], (default, base)


def test_slice_join_with_more_pairs_than_one_entry_holds(oracle):
    """The seed-index join by (query, slice) waves where a query has more passing references than one wave's LDS holds cursors for (300 > 256: two entries per query,
    the heads kernel's second 64-pair group partly filled), genomes of several contigs and a planted repeat (seeds with two anchors in one pair: the heads kernel cannot
    place those slices' chunk heads from the bitmap, the emit walk rewrites their rows) - against the per-pair join, and eight sampled hits against the oracle."""
    base = _run(ONE_FAMILY, {})
    assert base[0] > 80000      # (the far ends of the family - 12 % apart - fall below the screen or the aligned fraction)
    assert _run(ONE_FAMILY, {"PSK_GSI_SLICE": "0"}) == base
    assert _run(ONE_FAMILY, {"PSK_GSL_STAGE": "0"}) == base
    assert _run(ONE_FAMILY, {"PSK_PIPELINE": "1", "PSK_BATCH_ITEMS_LOG2": "21"}) == base      # two batches in flight, entries of one query in different batches
    import numpy as np
    import pyskani_amd as psk
    lut = np.frombuffer(b"ACGT", np.uint8)
    rng = np.random.default_rng(77)

    def mutate(a, d):
        b = a.copy(); m = rng.random(len(a)) < d; b[m] = (b[m] + rng.integers(1, 4, int(m.sum()), dtype=np.uint8)) & 3; return b
    anc = rng.integers(0, 4, 90_000, dtype=np.uint8)
    rep = rng.integers(0, 4, 2_500, dtype=np.uint8)

    def member(j):
        a = mutate(anc, 0.0004 * j)
        if j % 3 == 0:
            a = np.concatenate([a[:30_000], rep, a[30_000:60_000], mutate(rep, 0.01), a[60_000:]])
        parts = [a] if j % 4 else [a[:41_000], a[41_000:41_600], a[41_600:]]
        return [lut[x].tobytes() for x in parts]
    genomes = [(f"m{j}", member(j)) for j in range(300)]
    db = psk.Database(compression=30, marker_compression=200)
    db.sketch_many([(n, *c) for n, c in genomes])
    pick = np.random.default_rng(11)
    qs = [int(x) for x in pick.choice(300, 8, replace=False)]
    res = db.query_many([(genomes[q][0], *genomes[q][1]) for q in qs], learned_ani=False)
    for q, hits in zip(qs, res):
        assert len(hits) > 256      # (more pairs than one entry holds)
        h = hits[int(pick.integers(0, len(hits)))] if q % 2 else next(x for x in hits if x.reference_name == f"m{(q // 3) * 3}")      # (half of them against a reference with the repeat)
        r = int(h.reference_name[1:])
        want = oracle.chain(oracle.Sketch(genomes[r][1], c=30, marker_c=200), oracle.Sketch(genomes[q][1], c=30, marker_c=200))
        for f in ("n_anchors", "n_chunks", "n_intervals", "covered_query", "covered_ref", "sum_chain_anchors", "sum_chunk_seeds"):
            assert int(h._raw[f]) == int(getattr(want, f)), (q, r, f, int(h._raw[f]), int(getattr(want, f)))
        assert abs(h.identity - want.ani) < 1e-6 and abs(h.query_fraction - want.af_query) < 1e-6 and abs(h.reference_fraction - want.af_ref) < 1e-6


def test_rescue_prefilter_agrees_at_scale():
    base = _run(RESCUE, {})                         # 3 600 rescued contigs x 300 references: 2^20 pairs and more, the prefilter's default range
    assert base[0] > 4800 * 20
    # (the default joins these many small pairs through the references' probe tables, hands the DP its rows by chunk length, selects and reduces the
    # pairs of a handful of rows by one lane each, takes the anchor offsets from the join's running counts; switches of different stages share a run)
    for extra in ({"PSK_PREFILTER": "0"},
                  {"PSK_GSI_ONEPASS": "0", "PSK_DP_PRUNE": "0"},                                              # the index join with its count pass; every DP scoring its whole band
                  {"PSK_GSI_STAGE": "0"},                                                                   # every anchor of the index join its own 16-byte store (default: pairs of 32 bytes)
                  {"PSK_BSI_SMALL": "0"},                                                                   # the database-wide seed index in one walk instead of its blocks of 256 references
                  {"PSK_GSI_JOIN": "0"},                                                                    # the probe-table join instead of the database-wide seed index
                  {"PSK_GSI_JOIN": "0", "PSK_PREFILTER": "0"},
                  {"PSK_PREFILTER": "1", "PSK_JOIN_PAIRS": "0"},
                  {"PSK_PROBE": "0", "PSK_PREFILTER": "0", "PSK_SELECT_TINY": "0", "PSK_REDUCE_TINY": "0"},      # joins through the k-mer indexes, a wave per pair in selection and reduce
                  {"PSK_ROW_SORT": "0", "PSK_PROBE_LOCAL": "0", "PSK_ROUND_QUERIES": "700"},                  # rows in table order, offsets by a scan over the items, seven rounds of queries
                  {"PSK_CHAIN_WAVE_REG": "1", "PSK_PREFILTER": "1"},                                        # the small launch's register-window DP forced on the large batch
                  {"PSK_CHAIN_QUAD_DEEP": "0", "PSK_SELECT_TINY": "0"}):                                     # c = 30 (band 83): the wave-per-chunk LDS-ring DP
        assert _run(RESCUE, extra) == base, extra


def test_large_batch_sample_matches_oracle(oracle):
    """Closes the transitive link: 40 random (query, reference) pairs of the SAME 320-genome all-vs-all batch the
    digest tests run - i.e. hits produced by the large-batch paths (per-pair emit with the chunk table, XCD turns, live lists) -
    recomputed by the CPU oracle, every chaining integer and ANI / AF compared."""
    import numpy as np
    import pyskani_amd as psk
    lut = np.frombuffer(b"ACGT", np.uint8)
    rng = np.random.default_rng(77)

    def mutate(a, d):
        b = a.copy(); m = rng.random(len(a)) < d; b[m] = (b[m] + rng.integers(1, 4, int(m.sum()), dtype=np.uint8)) & 3; return b
    anc = [rng.integers(0, 4, 600_000, dtype=np.uint8) for _ in range(8)]
    genomes = [(f"g{f}_{j}", lut[mutate(anc[f], 0.002 * j)].tobytes()) for f in range(8) for j in range(40)]
    db = psk.Database()
    db.sketch_many(genomes)
    res = db.query_many(genomes, learned_ani=False)
    assert sum(len(h) for h in res) >= 320 * 40
    by_name = {n: g for n, g in genomes}
    pick = np.random.default_rng(5)
    checked = 0
    for qi in pick.choice(len(genomes), 40, replace=False):
        hits = res[int(qi)]
        h = hits[int(pick.integers(0, len(hits)))]
        want = oracle.chain(oracle.Sketch([by_name[h.reference_name]]), oracle.Sketch([genomes[int(qi)][1]]))
        for f in ("n_anchors", "n_chunks", "n_intervals", "covered_query", "covered_ref", "sum_chain_anchors", "sum_chunk_seeds"):
            assert int(h._raw[f]) == int(getattr(want, f)), (genomes[int(qi)][0], h.reference_name, f, int(h._raw[f]), int(getattr(want, f)))
        assert abs(h.identity - want.ani) < 1e-6 and abs(h.query_fraction - want.af_query) < 1e-6 and abs(h.reference_fraction - want.af_ref) < 1e-6
        checked += 1
    assert checked == 40


MANY_REFS = COMMON + r"""
import time, torch
N = int(os.environ.get("PSK_TEST_MANY_REFS", "70000"))
fam = N // 100
anc = rng.integers(0, 4, (fam, 2800), dtype=np.uint8)
refs = []
for i in range(N):
    a = anc[i // 100].copy(); m = rng.random(2800) < 0.0003 * (i % 100); a[m] = (a[m] + 1) & 3
    refs.append((f"r{i}", lut[a[: 2000 + (i * 7) % 800]].tobytes()))
contigs = []
for j in range(2000):
    i = int(rng.integers(0, N)); a = anc[i // 100].copy(); m = rng.random(2800) < 0.01; a[m] = (a[m] + 2) & 3
    contigs.append((f"c{j}", lut[a[100:2700]].tobytes()))
db = psk.Database(compression=30, marker_compression=200)
db.sketch_many(refs)
t0 = time.perf_counter(); res = db.query_many(contigs, learned_ani=False); t1 = time.perf_counter()
free0 = torch.cuda.mem_get_info()[0]
res2 = db.query_many(contigs, learned_ani=False); t2 = time.perf_counter()
free1 = torch.cuda.mem_get_info()[0]
n, d = digest(res)
assert digest(res2) == (n, d)
import pickle
pick = np.random.default_rng(3).choice(2000, 40, replace=False)
sample = []
for j in pick:
    hs = res[int(j)]
    if hs:
        h = hs[len(hs) // 2]
        sample.append((int(j), h.reference_name, {f: int(h._raw[f]) for f in ("n_anchors", "n_chunks", "n_intervals", "covered_query", "covered_ref", "sum_chain_anchors", "sum_chunk_seeds")}, h.identity, h.query_fraction))
if os.environ.get("PSK_TEST_SAMPLE"):
    pickle.dump((sample, {n: g for n, g in refs if any(n == s[1] for s in sample)}, {j: contigs[j][1] for j, *_ in sample}), open(os.environ["PSK_TEST_SAMPLE"], "wb"))
import ctypes as C
lk = C.c_uint64()
db._lib.psk_ctx_join_work(db._ctx._h, C.byref(lk), None, None, None, 0)
print(n, d, free0 - free1, round(t2 - t1, 3), lk.value)
"""


def test_database_beyond_the_seed_index_limit(oracle, tmp_path):
    """the reference loop has no limit on the number of references (lib.rs:617-637). The database-wide seed index carries 16-bit reference
    ids and stops at 65 536 references; the BLOCKED index (block-local ids, 64-bit block offsets) does not, and since round 6 a larger database is joined - and its rescued contigs
    prefiltered - through it. 70 000 references of 2-2.8 kb, 2 000 contig queries (every contig has fewer than 20 markers: rescued, i.e. screened against EVERY reference): the
    index walks ran (the library's lookup counter), the same hits from a second call, free device memory flat across it, the same hits with the index joins switched off
    (PSK_GSI_JOIN=0: probe-table join + per-reference prefilter, the route of such databases until round 5), 40 sampled hits recomputed by the oracle."""
    import pickle
    sample_file = str(tmp_path / "sample.pkl")
    env = dict(os.environ, PSK_TEST_SAMPLE=sample_file)
    for k in ("PSK_GSI_JOIN", "PSK_PROBE", "PSK_PREFILTER", "PSK_BSI_SMALL", "PSK_GSI"):
        env.pop(k, None)
    out = subprocess.check_output([sys.executable, "-c", "import os\n" + MANY_REFS], env=env, timeout=1500).decode().split()
    assert int(out[0]) > 2000 * 20 and int(out[2]) < (64 << 20) and int(out[4]) > 0, out      # hits; device memory the second call kept; index lookups
    env.pop("PSK_TEST_SAMPLE")
    o2 = subprocess.check_output([sys.executable, "-c", "import os\n" + MANY_REFS], env=dict(env, PSK_GSI_JOIN="0"), timeout=1500).decode().split()
    assert o2[:2] == out[:2] and int(o2[4]) == 0, (o2, out)      # the same hits without any index walk
    sample, refs, contigs = pickle.load(open(sample_file, "rb"))
    assert len(sample) >= 30
    for j, rname, ints, ani, afq in sample:
        want = oracle.chain(oracle.Sketch([refs[rname]], c=30, marker_c=200), oracle.Sketch([contigs[j]], c=30, marker_c=200))
        for f, v in ints.items():
            assert v == int(getattr(want, f)), (j, rname, f, v, int(getattr(want, f)))
        assert abs(ani - want.ani) < 1e-6 and abs(afq - want.af_query) < 1e-6


MANY_BLOCKS = COMMON + r"""
import os, pickle
# 18 000 references of ~72 kb at c = 30 (~2 400 seeds each: the slice join's range) in families of 50 consecutive references: 71 index blocks of 256 references,
# family 327 (references 16 350 - 16 399) straddles the boundary between blocks 63 and 64 - the second 64-bit word of the walks' block masks
N, FAM, L, CC, MC = (int(x) for x in os.environ.get("PSK_TEST_SHAPE", "18000 50 72000 30 200").split())
EDGE = (N // 16384 - (N % 16384 < 300)) * 16384 if N < 65536 else 65536      # the reference index the test is about: 16 384 (index blocks 63 | 64) / 65 536 (beyond the 16-bit ids)
def near(a, n_mut):
    b = a.copy(); p = rng.integers(0, len(a), n_mut); b[p] = (b[p] + rng.integers(1, 4, n_mut, dtype=np.uint8)) & 3; return b
refs = []
for f in range(N // FAM):
    a = rng.integers(0, 4, L + 500, dtype=np.uint8)
    for j in range(FAM):
        refs.append((f"r{f * FAM + j}", lut[near(a, 30 * j)[: L + (j * 37) % 500]].tobytes()))
db = psk.Database(compression=CC, marker_compression=MC)
db.sketch_many(refs)
# genome queries: whole families around the boundary and in the last blocks, and every 40th genome of the rest
qg = sorted(set(list(range(EDGE - 84, EDGE + 66)) + list(range(N - 150, N)) + list(range(0, 100)) + list(range(0, N, 40))))
genomes = [refs[i] for i in qg]
res_g = db.query_many(genomes, learned_ani=False)
# contig queries (< 2 048 seeds: the contig join), cut from references all over the database, half of them from blocks >= 64
contigs = []
for j in range(6000):
    i = int(rng.integers(EDGE, N)) if j % 2 else int(rng.integers(0, N)); g = np.frombuffer(refs[i][1], np.uint8)
    ln = int(rng.integers(L // 30, L // 8)); st = int(rng.integers(0, len(g) - ln))
    c = g[st:st + ln].copy(); p = rng.integers(0, ln, ln // 100); c[p] = lut[rng.integers(0, 4, len(p))]
    contigs.append((f"c{j}", c.tobytes()))
res_c = db.query_many(contigs, learned_ani=False)
ng, dg = digest(res_g); nc, dc = digest(res_c)
if os.environ.get("PSK_TEST_SAMPLE"):
    pick = np.random.default_rng(3)
    sample = []
    ints = ("n_anchors", "n_chunks", "n_intervals", "covered_query", "covered_ref", "sum_chain_anchors", "sum_chunk_seeds")
    for kind, queries, res in (("g", genomes, res_g), ("c", contigs, res_c)):
        cand = [x for x in range(len(queries)) if res[x] and any(int(h.reference_name[1:]) >= EDGE for h in res[x])]
        for x in pick.choice(cand, 12, replace=False):
            hs = [h for h in res[int(x)] if int(h.reference_name[1:]) >= EDGE]
            h = hs[int(pick.integers(0, len(hs)))]
            sample.append((queries[int(x)][1], refs[int(h.reference_name[1:])][1], {f: int(h._raw[f]) for f in ints}, h.identity, h.query_fraction, h.reference_fraction, CC, MC))
    pickle.dump(sample, open(os.environ["PSK_TEST_SAMPLE"], "wb"))
import ctypes as C
lk, vis = C.c_uint64(), C.c_uint64()
db._lib.psk_ctx_join_work(db._ctx._h, C.byref(lk), C.byref(vis), None, None, 0)
print(ng + nc, dg + dc, lk.value, vis.value)
"""


def test_index_blocks_beyond_the_first_sixty_four(oracle, tmp_path):
    """databases of 16 385 - 65 536 references are joined through index blocks 64 - 255, which the walks reach through the second to fourth word of their
    block masks (slice_join.hip gsl_walk_kernel, join.hip gsi_join_kernel) - code no test ran. 18 000 references (71 blocks), a family astride blocks 63 | 64, genome queries
    (slice join) and contig queries (contig join) whose passing references sit in blocks >= 64: the index walks ran (the library's lookup counter), the hits equal those of
    the per-pair / probe-table joins that use no index, and 24 sampled hits against references in blocks >= 64 are recomputed by the oracle (lib.rs:617-657)."""
    import pickle
    sample_file = str(tmp_path / "sample.pkl")
    env = dict(os.environ, PSK_TEST_SAMPLE=sample_file)
    for k in ("PSK_GSI_JOIN", "PSK_GSI_SLICE", "PSK_BSI_SMALL", "PSK_PIPELINE", "PSK_PROBE", "PSK_PREFILTER"):
        env.pop(k, None)
    out = subprocess.check_output([sys.executable, "-c", MANY_BLOCKS], env=env, timeout=1500).decode().split()
    n_hits, dig, lookups = int(out[0]), out[1], int(out[2])
    assert n_hits > 500 * 40 and lookups > 0, out      # (the index joins count their lookups; the joins without an index leave the counter alone)
    env.pop("PSK_TEST_SAMPLE")
    for extra in ({"PSK_GSI_SLICE": "0", "PSK_GSI_JOIN": "0"}, {"PSK_BSI_SMALL": "0", "PSK_GSL_STAGE": "0"}):
        o2 = subprocess.check_output([sys.executable, "-c", MANY_BLOCKS], env=dict(env, **extra), timeout=1500).decode().split()
        assert (int(o2[0]), o2[1]) == (n_hits, dig), (extra, o2, out)
        if extra.get("PSK_GSI_JOIN") == "0":
            assert int(o2[2]) == 0, o2      # no index walk took part in the cross-check
    _check_sample(oracle, sample_file)


def _check_sample(oracle, sample_file):
    import pickle
    sample = pickle.load(open(sample_file, "rb"))
    assert len(sample) == 24
    for q, r, ints, ani, afq, afr, cc, mc in sample:
        want = oracle.chain(oracle.Sketch([r], c=cc, marker_c=mc), oracle.Sketch([q], c=cc, marker_c=mc))
        for f, v in ints.items():
            assert v == int(getattr(want, f)), (f, v, int(getattr(want, f)))
        assert abs(ani - want.ani) < 1e-6 and abs(afq - want.af_query) < 1e-6 and abs(afr - want.af_ref) < 1e-6


def test_genomes_beyond_65536_references_go_through_the_blocked_index(oracle, tmp_path):
    """the blocked seed index has block-local reference ids and 64-bit block offsets since round 6 - no bound on the references of a database but memory -
    and the walks take their blocks from per-entry block tables instead of four 64-bit masks. 66 000 references of ~21 kb at c = 10 (2 100 seeds: the slice join's range; 258
    index blocks), a family astride reference 65 536, genome queries (slice join) and contig queries (contig join: blocks only - such a database has no database-wide index):
    the index walks ran, the hits equal those of the joins that use no index, 24 sampled hits against references beyond 65 536 recomputed by the oracle (lib.rs:617-657)."""
    sample_file = str(tmp_path / "sample.pkl")
    env = dict(os.environ, PSK_TEST_SAMPLE=sample_file, PSK_TEST_SHAPE="66000 50 21000 10 100")
    for k in ("PSK_GSI_JOIN", "PSK_GSI_SLICE", "PSK_BSI_SMALL", "PSK_PIPELINE", "PSK_PROBE", "PSK_PREFILTER", "PSK_GSI"):
        env.pop(k, None)
    out = subprocess.check_output([sys.executable, "-c", MANY_BLOCKS], env=env, timeout=1800).decode().split()
    n_hits, dig, lookups = int(out[0]), out[1], int(out[2])
    assert n_hits > 500 * 40 and lookups > 0, out
    env.pop("PSK_TEST_SAMPLE")
    o2 = subprocess.check_output([sys.executable, "-c", MANY_BLOCKS], env=dict(env, PSK_GSI_SLICE="0", PSK_GSI_JOIN="0"), timeout=1800).decode().split()
    assert (int(o2[0]), o2[1]) == (n_hits, dig) and int(o2[2]) == 0, (o2, out)
    _check_sample(oracle, sample_file)


MIXED = r"""
import ctypes as C, os, sys, time
sys.path.insert(0, %r)
import torch
import bench as B
dev = torch.device("cuda:0")
eng = B.Engine(0)
def free_gb(): return torch.cuda.mem_get_info()[0] / 2**30
n = 2000
anc_lens, fam_of = B.family_layout(3, n, n // 100)
buf, offs, lens = B.make_genomes(torch, dev, 3, 31, list(range(n)), fam_of, anc_lens, variant="plain")
torch.cuda.synchronize()
names = (C.c_char_p * n)(*[f"g{i}".encode() for i in range(n)])
c_off, c_len, gfc, nn = eng.layout(offs, lens, None)
out = eng.sketch_device_c(buf.data_ptr(), c_off, c_len, gfc, nn)
db = eng.make_db(names, out, nn)
os.environ["PSK_PIPELINE"] = "1"      # two lanes, each with its own chain scratch
h1 = eng.query_many(db, out, nn)
del os.environ["PSK_PIPELINE"]
eng.lib.psk_db_destroy(db); del buf; torch.cuda.empty_cache()      # (the database owns the sketches it was given)
held = free_gb()
g = 4
buf, offs, lens, gfc_l = B.make_big_genomes(torch, dev, g, 8, 40_000_000, 2, seed=5)      # 4 genomes of 8 x 40 Mb: pairs of > 2^20 seeds, the Gb-scale plan
torch.cuda.synchronize()
names = (C.c_char_p * g)(*[f"m{i}".encode() for i in range(g)])
c_off, c_len, gfc, nn = eng.layout(offs, lens, gfc_l)
hits, ms = [], []
for rep in range(3):
    handles = eng.sketch_device_c(buf.data_ptr(), c_off, c_len, gfc, nn)
    db = eng.make_db(names, handles, nn)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    hits.append(eng.query_many(db, handles, nn)); ms.append(1e3 * (time.perf_counter() - t0))
    eng.lib.psk_db_destroy(db)
after = free_gb()
# the same Gb-scale job in a context that never ran the all-vs-all
eng2 = B.Engine(0)
handles = eng2.sketch_device_c(buf.data_ptr(), c_off, c_len, gfc, nn)
db = eng2.make_db(names, handles, nn)
fresh = eng2.query_many(db, handles, nn)
print(h1, hits[0], hits[1], hits[2], fresh, round(held, 1), round(after, 1), " ".join(f"{x:.0f}" for x in ms))
""" % (ROOT,)


def test_gb_scale_call_after_an_all_vs_all_on_the_same_context():
    """VERDICT r5 item 2c (was profiles/scripts/r5_trim_lanes.py): an all-vs-all on two lanes leaves each lane holding its chain scratch; a Gb-scale call on the same
    context sizes its batches by the free memory and takes the idle lane's scratch back (psk_trim_idle_lanes). The Gb-scale call must return the hits a fresh context
    returns, three times over, and the library must end up holding less memory than the all-vs-all left it with."""
    out = subprocess.check_output([sys.executable, "-c", MIXED], env={k: v for k, v in os.environ.items() if not k.startswith("PSK_")}, timeout=1500).decode().split()
    h1, a, b, c, fresh = (int(x) for x in out[:5])
    assert h1 >= 2000 * 50 and a == b == c == fresh and a >= 4, out


def test_prefilter_scratch_returns_to_the_pool():
    """the seed prefilter's scratch block was a PoolScratch local without a destructor and leaked one block per
    query_many round. Free device memory must stay flat over repeated calls with the prefilter forced on."""
    code = COMMON + r"""
import torch
anc = [rng.integers(0, 4, 300_000, dtype=np.uint8) for _ in range(4)]
refs = [(f"r{f}_{j}", lut[mutate(anc[f], 0.004 * j)].tobytes()) for f in range(4) for j in range(10)]
contigs = []
for i in range(600):
    a = anc[i % 4]; L = int(rng.integers(1200, 3000)); st = int(rng.integers(0, len(a) - L))
    contigs.append((f"c{i}", lut[mutate(a[st:st + L], 0.01)].tobytes()))
db = psk.Database(compression=30, marker_compression=200)
db.sketch_many(refs)
n0 = sum(len(h) for h in db.query_many(contigs, learned_ani=False))
db.query_many(contigs, learned_ani=False)
free0 = torch.cuda.mem_get_info()[0]
for _ in range(12):
    assert sum(len(h) for h in db.query_many(contigs, learned_ani=False)) == n0
free1 = torch.cuda.mem_get_info()[0]
print(n0, free0 - free1)
"""
    env = dict(os.environ, PSK_PREFILTER="1")
    out = subprocess.check_output([sys.executable, "-c", "import torch\n" + code], env=env, timeout=900).decode().split()
    assert int(out[0]) > 600 and int(out[1]) < (8 << 20), out      # the leak was the prefilter block (tens of MB) per call
