#!/usr/bin/env python3
"""profiles/scripts/pmc_summary.py — HBM traffic per unit of work of the profiled kernels, from the raw counter files of
profiles/scripts/r3_pmc.sh (gpurun_out/pmc/<tag>.{FETCH_SIZE,WRITE_SIZE}.txt: kernel, launches, counter total in KiB over
1 warm-up + 2 timed steps) and the unit counts of the same workload (gpurun_out/pmc/r3_units_<workload>.json: chain-stage work per
step as bench.py reports it). Writes profiles/r3/pmc_kernels.json: {workload: {timer: {...}}} — what bench.py's `roofline.traffic`
scales by the run's own units. Method: MI355X_MICROARCH.md HBM section — separate passes; FETCH_SIZE / WRITE_SIZE in KiB; on gfx950
FETCH_SIZE counts half of a 16-byte-per-lane coalesced stream, so kernels whose reads are such streams get the x2 correction
(`fetch_x2`), the others are reported raw and marked uncalibrated."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
PMC = os.path.join(ROOT, "gpurun_out", "pmc")
STEPS = 3      # 1 warm-up + 2 timed steps in every counter pass
# timer of bench.py -> (kernels of that bracket in this workload, unit, reads are 16-byte-per-lane streams?)
PLAN = {
    "allvsall": ("r3_ava", {"anchor": (["anchor_join4_kernel"], "item", False), "anchor_emit": (["anchor_emit_pairs_kernel"], "item", False),
                            "chain_chunk": (["chain_lane20_kernel"], "anchor", True)}),
    "metagenome": ("r3_meta", {"anchor": (["anchor_join_probe_kernel"], "item", False), "anchor_emit": (["anchor_emit_packed4_kernel", "chunk_heads_kernel"], "item", False),
                               "chain_chunk": (["chain_quad_deep_kernel", "chain_chunk_list_kernel"], "anchor", True)}),
    "mammalian": ("r3_mammal", {"anchor": (["anchor_join4_kernel"], "item", False), "anchor_emit": (["anchor_emit_expand_kernel", "anchor_next_kernel", "chunk_hops_sliced_kernel"], "anchor", True),
                                "chain_chunk": (["chain_lane20x_kernel", "chain_chunk_list_kernel"], "anchor", True)}),
}


def read_counter(tag, counter):
    out = {}
    path = os.path.join(PMC, f"{tag}.{counter}.txt")
    for line in open(path):
        k, n, v = line.rstrip("\n").split("\t")
        name = k.replace("void ", "").split("<")[0].strip()      # template instances of one kernel are summed
        prev = out.get(name, (0, 0.0))
        out[name] = (prev[0] + int(n), prev[1] + float(v) * 1024.0)      # KiB -> bytes
    return out


def main():
    result = {}
    for workload, (tag, timers) in PLAN.items():
        try:
            fetch, write = read_counter(tag, "FETCH_SIZE"), read_counter(tag, "WRITE_SIZE")
            line = json.loads(open(os.path.join(PMC, f"r3_units_{workload}.json")).read().strip().splitlines()[-1])
        except (OSError, ValueError) as e:
            print("skip", workload, e, file=sys.stderr)
            continue
        work = line["extras"]["chain_work_per_step"]
        units = {"item": work["items"], "anchor": work["anchors"]}
        result[workload] = {}
        for timer, (kernels, unit, wide) in timers.items():
            f = sum(fetch.get(k, (0, 0.0))[1] for k in kernels) / STEPS
            w = sum(write.get(k, (0, 0.0))[1] for k in kernels) / STEPS
            if units[unit] <= 0 or (f == 0 and w == 0):
                continue
            total = (2.0 * f if wide else f) + w
            result[workload][timer] = {"kernels": kernels, "unit": unit, "units_per_step": units[unit], "fetch_bytes_raw_per_step": f, "write_bytes_per_step": w,
                                       "fetch_x2": wide, "bytes_per_unit": total / units[unit],
                                       "note": ("FETCH_SIZE x 2 (16-byte-per-lane streams, gfx950 correction) + WRITE_SIZE" if wide else
                                                "FETCH_SIZE raw + WRITE_SIZE; the kernel's reads are 4- and 8-byte accesses, for which the gfx950 FETCH_SIZE scale is uncalibrated: a lower bound")}
    out = os.path.join(ROOT, "profiles", "r3", "pmc_kernels.json")
    json.dump(result, open(out, "w"), indent=1)
    print("wrote", out)
    for wl, t in result.items():
        for k, v in t.items():
            print(f"{wl:11s} {k:12s} {v['bytes_per_unit']:8.2f} B/{v['unit']}  (fetch {v['fetch_bytes_raw_per_step'] / 1e9:.2f} GB raw, write {v['write_bytes_per_step'] / 1e9:.2f} GB per step)")


if __name__ == "__main__":
    main()
