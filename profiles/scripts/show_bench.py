#!/usr/bin/env python3
"""profiles/scripts/show_bench.py <bench line json> — the default bench line as a table (headline + extras.workloads)."""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
r = d["roofline"]
print(f"search: {d['ms_per_step']:.2f} ms/step = {d['value'] / 1e3:.1f} k pairs/s; roofline {r['kernel']} {r['achieved']:.0f} GB/s frac {r['frac']:.3f} avg launch {r['avg_launch_ms']:.3f} ms; clock {d['clock']['shader_clock_mhz']:.0f} MHz; cpu {d['cpu_baseline']['value']:.1f} / {d['cpu_baseline']['all_cores']['value']:.0f} pairs/s")
print("  kernels", {k: round(v, 2) for k, v in d["kernel_ms_per_step"].items()})
e = d["extras"]
print(f"  api {e.get('api_pairs_per_s', 0):.0f} pairs/s, host_ascii {e.get('host_ascii_pairs_per_s', 0):.0f} pairs/s; valu {r.get('valu')}")
for k, v in e.get("workloads", {}).items():
    if not v: continue
    if "ms_per_step" not in v:
        print(k, v); continue
    rr = v["roofline"]
    cb = v.get("cpu_baseline", {})
    print(f"{k}: {v['ms_per_step']:.1f} ms/step = {v['value']:.4g} {v['unit']}; hits {v['hits']}; roofline {rr['kernel']} {rr['achieved']:.0f} GB/s frac {rr['frac']:.3f} traffic {rr.get('traffic')}; clock {v['clock']['shader_clock_mhz']:.0f}; cpu {cb.get('value')} / {cb.get('all_cores', {}).get('value')}")
    print("   kernels", {a: round(b['ms_per_step'], 1) for a, b in v["kernel_roofline"].items()})
    print("   fracs", {a: round(b['frac_of_hbm_peak'], 3) for a, b in v["kernel_roofline"].items() if 'frac_of_hbm_peak' in b})
    if "oracle_check" in v: print("   oracle", v["oracle_check"]["result"], [p["n_anchors"] for p in v["oracle_check"]["pairs"]])
print("workloads wall", e.get("workloads_wall_s"))
