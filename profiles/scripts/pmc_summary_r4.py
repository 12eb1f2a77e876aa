#!/usr/bin/env python3
"""profiles/scripts/pmc_summary_r4.py - HBM traffic per unit of work of the chain-stage kernels, from the raw counter files of
profiles/scripts/r4_pmc.sh (gpurun_out/pmc/<tag>.{FETCH_SIZE,WRITE_SIZE}.txt: kernel, launches, counter total in KiB over 1 warm-up +
2 timed steps) and the unit counts of the same workloads (gpurun_out/pmc/r4_units_<workload>.json). Writes profiles/r4/pmc_kernels.json,
what bench.py's `roofline.traffic` scales by the run's own units.

Scale factors are the ones MEASURED in profiles/r4/r4k_pmc_calibration.md (profiles/micro/pmc_calib.hip), per kernel by its access shape:
  stream  coalesced loads of any width, and per-lane private runs of 16-byte records: FETCH_SIZE reports half -> x 2
  gather  isolated 64-byte lines (random probes, bucket-table reads, binary searches): FETCH_SIZE is exact -> x 1
  runs    short contiguous runs (a k-mer's entries in the database-wide seed index: 3-10 lines): between the two -> x 1 reported, x 2 kept as the upper bound
WRITE_SIZE is taken as it is (exact for coalesced stores; a scattered 8- or 16-byte store is counted as the 32 bytes it costs)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PMC = os.path.join(ROOT, "gpurun_out", "pmc")
STEPS = 3      # 1 warm-up + 2 timed steps in every counter pass
F = {"stream": 2.0, "gather": 1.0, "runs": 1.0}
PLAN = {
    "allvsall": ("r4_ava", {"anchor": ([("anchor_join4_kernel", "stream")], "item"), "anchor_emit": ([("anchor_emit_pairs_kernel", "stream")], "item"),
                            "chain_chunk": ([("chain_lane20_kernel", "stream")], "anchor"), "select": ([("select_kernel", "stream")], "anchor")}),
    "metagenome": ("r4_meta", {"anchor": ([("gsi_join_kernel<false>", "runs")], "item"), "anchor_emit": ([("gsi_join_kernel<true>", "runs")], "item"),
                               "chain_chunk": ([("chain_quad_deep_kernel", "stream"), ("chain_chunk_list_kernel", "stream")], "anchor")}),
    "mammalian": ("r4_mammal", {"anchor": ([("anchor_join4_kernel", "stream")], "item"),
                                "anchor_emit": ([("anchor_emit_expand_kernel", "stream"), ("anchor_next_kernel<1>", "gather"), ("anchor_next_kernel<2>", "gather"), ("chunk_hops_sliced_kernel", "gather"),
                                                 ("item_next_kernel", "gather"), ("chunk_hops_items_kernel", "stream")], "anchor"),
                                "chain_chunk": ([("chain_lane20x_kernel", "stream"), ("chain_chunk_list_kernel", "stream")], "anchor")}),
}


def read_counter(tag, counter):
    out = {}
    for line in open(os.path.join(PMC, f"{tag}.{counter}.txt")):
        k, n, v = line.rstrip("\n").split("\t")
        out[k.replace("void ", "").strip()] = (int(n), float(v) * 1024.0)      # KiB -> bytes
    return out


def main():
    result = {"_method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (profiles/scripts/r4_pmc.sh), KiB x 1024, per timed step; FETCH scaled per kernel by its access "
                         "shape with the factors measured in profiles/r4/r4k_pmc_calibration.md (stream x 2, gather x 1, runs x 1 with x 2 as the upper bound)"}
    for workload, (tag, timers) in PLAN.items():
        try:
            fetch, write = read_counter(tag, "FETCH_SIZE"), read_counter(tag, "WRITE_SIZE")
            line = json.load(open(os.path.join(PMC, f"r4_units_{workload}.json")))
        except (OSError, ValueError) as e:
            print("skip", workload, e, file=sys.stderr)
            continue
        work = line["extras"].get("chain_work_per_step") or line["extras"].get("chain_work_per_step_rank0")
        units = {"item": work["items"], "anchor": work["anchors"]}
        result[workload] = {}
        for timer, (kernels, unit) in timers.items():
            raw = sum(fetch.get(k, (0, 0.0))[1] for k, _ in kernels) / STEPS
            scaled = sum(fetch.get(k, (0, 0.0))[1] * F[shape] for k, shape in kernels) / STEPS
            w = sum(write.get(k, (0, 0.0))[1] for k, _ in kernels) / STEPS
            if units[unit] <= 0 or (raw == 0 and w == 0):
                continue
            result[workload][timer] = {"kernels": [k for k, _ in kernels], "shapes": [s for _, s in kernels], "unit": unit, "units_per_step": units[unit],
                                       "fetch_bytes_raw_per_step": raw, "fetch_bytes_scaled_per_step": scaled, "write_bytes_per_step": w,
                                       "bytes_per_unit": (scaled + w) / units[unit], "bytes_per_unit_upper": (2.0 * raw + w) / units[unit]}
    out = os.path.join(ROOT, "profiles", "r4", "pmc_kernels.json")
    json.dump(result, open(out, "w"), indent=1)
    print("wrote", out)
    for wl, t in result.items():
        if wl.startswith("_"):
            continue
        for k, v in t.items():
            print(f"{wl:11s} {k:12s} {v['bytes_per_unit']:8.2f} B/{v['unit']} (upper {v['bytes_per_unit_upper']:.2f})  fetch raw {v['fetch_bytes_raw_per_step'] / 1e9:.2f} GB, scaled {v['fetch_bytes_scaled_per_step'] / 1e9:.2f} GB, write {v['write_bytes_per_step'] / 1e9:.2f} GB per step")


if __name__ == "__main__":
    main()
