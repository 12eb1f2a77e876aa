"""CPU: the bench's contract line stays inside the driver's stdout tail, and `--gpus N` decides to start its own ranks.

Round 3's line was 21 kB; the driver keeps an 8 kB tail and could not parse it (BENCH_r03.json: parsed = null). The mock below is
that very line (profiles/r3/r3z_bench_default.json): the compact form of it must fit in 4 kB and still carry the contract's keys."""
import json
import os
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)


@pytest.fixture(scope="module")
def bench():
    import importlib
    return importlib.import_module("bench")


def _mock_full():
    return json.load(open(os.path.join(ROOT, "profiles", "r3", "r3z_bench_default.json")))


def test_compact_line_fits_the_driver_tail_and_keeps_the_contract(bench):
    full = _mock_full()
    assert len(json.dumps(full)) > 16000                      # the mock really is the line that broke the parser
    full["copy_bw"] = {"GBps": 4321.0}
    full["roofline"]["frac_of_copy_bw"] = 0.4
    line = bench.compact_line(full, "profiles/r4/bench_full_search_x.json")
    text = json.dumps(line)
    assert len(text) < bench.LINE_BUDGET == 4096, len(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["value"] == pytest.approx(full["value"], rel=1e-4) and line["ms_per_step"] == pytest.approx(full["ms_per_step"], rel=1e-4)
    for k in ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "launches", "algorithmic_bytes_per_launch", "frac_of_copy_bw"):
        assert k in line["roofline"], k
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in line["cpu_baseline"], k
    assert line["cpu_baseline"]["all_cores"]["cores"] == full["cpu_baseline"]["all_cores"]["cores"]
    wl = line["extras"]["workloads"]
    assert set(wl) == {k for k, v in full["extras"]["workloads"].items() if v}
    for k, w in wl.items():
        if "ms_per_step" in full["extras"]["workloads"][k]:
            assert set(w) >= {"ms_per_step", "value", "unit", "hits", "roofline"}, (k, w)
            assert set(w["roofline"]) <= {"kernel", "frac", "traffic"}
    assert "model" not in line["config"] and "workload" in line["config"]


def test_contract_line_of_round_6_leads_with_the_contract_job(bench):
    """the line the driver parses is the 10 000 x 10 000 job (strong scaling) with host_to_host, the oracle check and the single-GPU scaling emulation beside it,
    the former headline under extras.workloads.search_1k - and it still fits the 4 kB budget. The mock is a full object a default run wrote (profiles/r6)."""
    full = json.load(open(os.path.join(ROOT, "profiles", "r6", "r6z_bench_default_full.json")))
    line = bench.compact_line(full, "profiles/r6/bench_full_allvsall_x.json")
    assert len(json.dumps(line)) < bench.LINE_BUDGET
    assert line["scaling"] == "strong" and "10000 x 10000" in line["config"]["workload"] and line["config"]["genomes"] == 10000
    assert line["unit"] == "genome-pairs/s" and line["value"] == pytest.approx(1e8 / (line["ms_per_step"] * 1e-3), rel=1e-3)
    assert {"value", "ms_per_step", "vs_cpu_all_cores"} <= set(line["host_to_host"])
    assert line["extras"]["oracle_check"] == "8 random hits bit-exact"
    sm = line["extras"]["scaling_model"]
    assert "emulation" in sm["kind"] and set(sm["ranks"]) == {"2", "4", "8"} and all(1.0 < v["speedup_vs_1"] <= int(n) * 1.05 for n, v in sm["ranks"].items())
    assert "search_1k" in line["extras"]["workloads"] and line["extras"]["workloads"]["search_1k"]["scaling"] == "weak"
    assert line["roofline"]["kernel"] in bench.KERNELS and 0 < line["roofline"]["frac"] <= 1 and line["cpu_baseline"]["kind"] == "port"


def test_compact_line_never_exceeds_the_budget_even_with_many_workloads(bench):
    full = _mock_full()
    w = full["extras"]["workloads"]
    for i in range(40):
        w[f"copy{i}"] = dict(w["allvsall_10k"])
    assert len(json.dumps(bench.compact_line(full))) < bench.LINE_BUDGET


def test_gpus_n_without_a_launcher_spawns_its_own_ranks(bench):
    assert bench.spawn_decision(8, {}) is True
    assert bench.spawn_decision(2, {"PATH": "/bin"}) is True
    assert bench.spawn_decision(1, {}) is False                          # the default run stays one process
    assert bench.spawn_decision(8, {"WORLD_SIZE": "8", "RANK": "3"}) is False   # a rank started by the driver's torch.distributed.run never spawns
    assert bench.spawn_decision(8, {"WORLD_SIZE": "1"}) is False


def test_spawn_happens_before_torch_is_imported_or_hip_is_touched():
    """the launcher parent must not initialise the GPU: the decision sits above `import torch` in main()"""
    src = open(os.path.join(ROOT, "bench.py")).read()
    main = src[src.index("def main():"):]
    assert main.index("spawn_decision(") < main.index("import torch")
    assert "os.exec" not in src and "execv" not in src


def test_no_kernel_row_prices_above_the_hbm_peak():
    """a bracket that did not run the work its byte formula counts (the one-walk index join has no COUNT pass: ~0.05 ms in the `anchor` timer) came out at
    227 x the HBM peak. Such rows carry no fraction; every row that does stays at or below 1."""
    import bench
    kern = {k: (0.0, 0) for k in bench.KERNELS}
    kern.update(sketch_scan=(7.0, 2), anchor=(0.1, 16), anchor_emit=(100.0, 16), chain_chunk=(80.0, 16), select=(9.0, 16), pair_reduce=(7.0, 16))
    units = {"bases": 1.4e9, "c": 30, "marker_c": 200, "items": 5.7e9, "anchors": 1.77e9, "index_lookups": 3.3e7, "index_entries_visited": 1.77e9, "candidates": 4e7, "chunk_rows": 2e7, "chained_pairs": 1.7e7}
    table = bench.kernel_rooflines(kern, 2, units, {})
    assert "frac_of_hbm_peak" not in table["anchor"]                       # 21 GB in 0.05 ms: not what that bracket did
    assert 0 < table["anchor_emit"]["frac_of_hbm_peak"] <= 1 and "index" in table["anchor_emit"]["bytes"]
    for k, row in table.items():
        assert row.get("frac_of_hbm_peak", 0.0) <= 1.0, (k, row)
    assert "frac_of_hbm_peak" in table["select"] and "frac_of_hbm_peak" in table["pair_reduce"]      # the kernels that had no byte count
    merge = bench.kernel_rooflines(kern, 2, dict(units, index_lookups=0.0, index_entries_visited=0.0), {})
    assert "item" in merge["anchor_emit"]["bytes"]
