#!/usr/bin/env python3
"""profiles/scripts/pmc_summary_r6.py - HBM traffic per unit of work of every kernel that holds >= 3 % of a workload's step, from the raw counter files of
profiles/scripts/r6_pmc.sh (gpurun_out/pmc/<tag>.{FETCH_SIZE,WRITE_SIZE}.txt: kernel, launches, counter total in KiB over 1 warm-up + 2 timed steps) and the
unit counts of the same workloads (gpurun_out/pmc/r6_units_<workload>.json: the library's own work counters). Writes profiles/r6/pmc_kernels.json - what bench.py's
`roofline.traffic` scales by the run's own units - and profiles/r6/r6_pmc_sketch_scan.json (the headline kernel, with its VALU instruction count).

Scale factors: the ones MEASURED in profiles/r4/r4k_pmc_calibration.md (profiles/micro/pmc_calib.hip), per kernel by its access shape:
  stream  coalesced loads of any width, and per-lane private runs of 16-byte records: FETCH_SIZE reports half -> x 2
  gather  isolated 64-byte lines (random probes, bucket-table reads, binary searches): FETCH_SIZE is exact -> x 1
  runs    short contiguous runs (a k-mer's entries in the seed index: 3-10 lines): between the two -> x 1 reported, x 2 kept as the upper bound
WRITE_SIZE is taken as it is (exact for coalesced stores; a scattered 8- or 16-byte store is counted as the 32 bytes it costs)."""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PMC = os.path.join(ROOT, "gpurun_out", "pmc")
OUT = os.path.join(ROOT, "profiles", "r6")
STEPS = 3      # 1 warm-up + 2 timed steps in every counter pass ...
STEPS_OF = {"allvsall": 4, "allvsall10k": 2}      # (the 10 000 x 10 000 pass: no warm-up, 1 timed step + the table step)
TAG_UNITS = {"allvsall10k": "allvsall10k"}      # ... and, for the all-vs-all at N = 1, bench.py's one more step as ONE chain of launches (the per-kernel table step): four passes of the step under the counters
F = {"stream": 2.0, "gather": 1.0, "runs": 1.0}
W_COUNT, W_EMIT = "gsl_walk_kernel<false, false>", "gsl_walk_kernel<true, true>"
PLAN = {
    "search": ("r6_search", {"sketch_scan": ([("sketch_scan_kernel", "stream")], "base", False), "sketch_emit": ([("sketch_emit_kernel", "stream")], "base", False)}),
    "allvsall": ("r6_ava", {"anchor": ([(W_COUNT, "runs")], "anchor", True), "anchor_emit": ([(W_EMIT, "runs"), ("gsl_heads_kernel", "stream")], "anchor", True),
                            "chain_chunk": ([("chain_lane20_kernel", "stream")], "anchor", False), "select": ([("select_kernel", "stream"), ("chunk_seeds_kernel", "gather")], "candidate", False),
                            "pair_reduce": ([("pair_reduce_kernel", "stream"), ("pair_reduce_wave_kernel", "stream")], "row", False)}),
    "allvsall10k": ("r6_ava10k", {"anchor": ([(W_COUNT, "runs")], "anchor", True), "anchor_emit": ([(W_EMIT, "runs"), ("gsl_heads_kernel", "stream")], "anchor", True),
                               "chain_chunk": ([("chain_lane20_kernel", "stream")], "anchor", False), "select": ([("select_kernel", "stream"), ("chunk_seeds_kernel", "gather")], "candidate", False)}),
    "metagenome": ("r6_meta", {"anchor_emit": ([("gsi_join_kernel<true>", "runs")], "anchor", True),
                               "chain_chunk": ([("chain_quad_deep_kernel", "stream"), ("chain_chunk_list_kernel", "stream")], "anchor", False),
                               "select": ([("select_tiny_kernel", "stream"), ("select_kernel", "stream")], "candidate", False),
                               "pair_reduce": ([("pair_reduce_tiny_kernel", "stream"), ("pair_reduce_small_kernel", "stream")], "row", False)}),
    "mammalian": ("r6_mammal", {"anchor": ([("anchor_join4_kernel", "stream")], "item", False),
                                "anchor_emit": ([("anchor_emit_expand_kernel", "stream"), ("chunk_hops_items_kernel", "stream"), ("chunk_hops_sliced_kernel", "gather")], "anchor", False),
                                "chain_chunk": ([("chain_lane20x_kernel", "stream"), ("chain_chunk_list_kernel", "stream")], "anchor", False),
                                "select": ([("select_huge_kernel", "stream"), ("select_big_kernel", "stream")], "candidate", False),
                                "pair_reduce": ([("pair_reduce_kernel", "stream"), ("pair_reduce_large_kernel", "stream")], "row", False)}),
}


def read_counter(tag, counter):
    out = {}
    for line in open(os.path.join(PMC, f"{tag}.{counter}.txt")):
        k, n, v = line.rstrip("\n").split("\t")
        k = k.replace("void ", "").strip()
        n0, v0 = out.get(k, (0, 0.0))
        out[k] = (n0 + int(n), v0 + float(v) * 1024.0)      # KiB -> bytes
    return out


def main():
    os.makedirs(OUT, exist_ok=True)
    os.makedirs(os.path.join(OUT, "pmc_raw"), exist_ok=True)
    result = {"_method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (profiles/scripts/r6_pmc.sh), KiB x 1024, per timed step; FETCH scaled per kernel by its access "
                         "shape with the factors measured in profiles/r4/r4k_pmc_calibration.md (stream x 2, gather x 1, runs x 1 with x 2 as the upper bound); units = the library's own work counters "
                         "of the same workload (psk_ctx_work / psk_ctx_join_work); index_join: the timer held a walk of the seed index (bench.py prices it by lookups, index entries and anchors)"}
    only = [a for a in sys.argv[1:] if a in PLAN]      # `pmc_summary_r6.py mammalian`: that workload's passes were run again, the others keep their entries
    if only:
        result.update({k: v for k, v in json.load(open(os.path.join(OUT, "pmc_kernels.json"))).items() if not k.startswith("_")})
    for workload, (tag, timers) in PLAN.items():
        if only and workload not in only:
            continue
        try:
            fetch, write = read_counter(tag, "FETCH_SIZE"), read_counter(tag, "WRITE_SIZE")
            line = json.load(open(os.path.join(PMC, f"r6_units_{workload}.json")))
        except (OSError, ValueError) as e:
            print("skip", workload, e, file=sys.stderr)
            continue
        for c in ("FETCH_SIZE", "WRITE_SIZE"):
            shutil.copy(os.path.join(PMC, f"{tag}.{c}.txt"), os.path.join(OUT, "pmc_raw", f"{tag}.{c}.txt"))
        work = line["extras"].get("chain_work_per_step") or line["extras"].get("chain_work_per_step_rank0")
        bases = line["extras"].get("bases_sketched_per_s", 0.0) * line["ms_per_step"] * 1e-3
        units = {"item": work["items"], "anchor": work["anchors"], "candidate": work.get("candidates", 0.0), "row": work.get("chunk_rows", 0.0), "base": bases}
        result[workload] = {}
        for timer, (kernels, unit, index_join) in timers.items():
            raw = sum(fetch.get(k, (0, 0.0))[1] for k, _ in kernels) / STEPS_OF.get(workload, STEPS)
            scaled = sum(fetch.get(k, (0, 0.0))[1] * F[shape] for k, shape in kernels) / STEPS_OF.get(workload, STEPS)
            w = sum(write.get(k, (0, 0.0))[1] for k, _ in kernels) / STEPS_OF.get(workload, STEPS)
            if units[unit] <= 0 or (raw == 0 and w == 0):
                continue
            result[workload][timer] = {"kernels": [k for k, _ in kernels if k in fetch or k in write], "shapes": [s for k, s in kernels if k in fetch or k in write], "unit": unit, "units_per_step": units[unit], "index_join": index_join,
                                       "fetch_bytes_raw_per_step": raw, "fetch_bytes_scaled_per_step": scaled, "write_bytes_per_step": w,
                                       "bytes_per_unit": (scaled + w) / units[unit], "bytes_per_unit_upper": (2.0 * raw + w) / units[unit]}
    json.dump(result, open(os.path.join(OUT, "pmc_kernels.json"), "w"), indent=1)
    print("wrote", os.path.join(OUT, "pmc_kernels.json"))
    for wl, t in result.items():
        if wl.startswith("_"):
            continue
        for k, v in t.items():
            print(f"{wl:11s} {k:12s} {v['bytes_per_unit']:8.2f} B/{v['unit']} (upper {v['bytes_per_unit_upper']:.2f})  fetch raw {v['fetch_bytes_raw_per_step'] / 1e9:.2f} GB, scaled {v['fetch_bytes_scaled_per_step'] / 1e9:.2f} GB, write {v['write_bytes_per_step'] / 1e9:.2f} GB per step")
    # the headline kernel: traffic per base + VALU wave-instructions per launch (bench.py: roofline.valu)
    if only and "search" not in only:
        return
    try:
        sq = {}
        for line in open(os.path.join(PMC, "r6_search_sq.sq.txt")):
            k, c, n, tot, avg = line.rstrip("\n").split("\t")
            sq[(k.replace("void ", "").strip(), c)] = (int(n), float(tot))
        shutil.copy(os.path.join(PMC, "r6_search_sq.sq.txt"), os.path.join(OUT, "pmc_raw", "r6_search_sq.sq.txt"))
        s = result["search"]["sketch_scan"]
        launches = sq[("sketch_scan_kernel", "SQ_INSTS_VALU")][0]
        doc = {"kernel": "sketch_scan_kernel", "bases_per_launch": s["units_per_step"], "fetch_bytes_raw": s["fetch_bytes_raw_per_step"], "write_bytes": s["write_bytes_per_step"],
               "traffic_bytes_corrected": s["fetch_bytes_scaled_per_step"] + s["write_bytes_per_step"], "traffic_bytes_per_base": s["bytes_per_unit"],
               "valu_wave_instructions_per_launch": sq[("sketch_scan_kernel", "SQ_INSTS_VALU")][1] / launches,
               "grbm_gui_active_cycles_per_launch": sq[("sketch_scan_kernel", "GRBM_GUI_ACTIVE")][1] / sq[("sketch_scan_kernel", "GRBM_GUI_ACTIVE")][0],
               "source": "round 6: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE / SQ counters in separate passes (profiles/scripts/r6_pmc.sh) over `bench.py --workload search --steps 2 --warmup 1 --cpu-sample 0 --no-api` "
                         "(3 launches of 1 001 genomes); counters in KiB; FETCH x 2 for coalesced streams (profiles/r4/r4k_pmc_calibration.md). Raw files: profiles/r6/pmc_raw/r6_search*"}
        json.dump(doc, open(os.path.join(OUT, "r6_pmc_sketch_scan.json"), "w"), indent=1)
        print("wrote r6_pmc_sketch_scan.json:", doc["traffic_bytes_per_base"], "B per base,", doc["valu_wave_instructions_per_launch"] / doc["bases_per_launch"] * 64, "VALU wave-instructions per 64 bases")
    except (OSError, KeyError, ValueError) as e:
        print("no sketch_scan summary:", e, file=sys.stderr)


if __name__ == "__main__":
    main()
