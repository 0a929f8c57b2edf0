#!/usr/bin/env python3
"""Brings the rocprofv3 passes of tools/prof_all.sh to per-unit figures and writes the round's profile summaries:
  prof_report.py <prof dir> <out prefix, e.g. profiles/r03a>
  -> <prefix>_kernel_stats_<leg>.csv            the profiler's per-kernel statistics (our kernels only)
     <prefix>_pmc_counters.json                 per kernel: instructions / cycles per unit, wave-cycle split, LDS conflicts
     <prefix>_pmc_traffic_<leg>.json            HBM bytes per unit (block / stream / 64 KiB of output), per kernel and summed
Corrections as MI355X_MICROARCH.md prescribes: FETCH_SIZE / WRITE_SIZE are in KB (x 1024); FETCH_SIZE x 2 on gfx950
(wide coalesced reads are tallied at half their bytes); WRITE_SIZE as is."""
import collections, csv, glob, json, os, sys
src, prefix = sys.argv[1], sys.argv[2]
CLOCK, CUS = 2.4e9, 256
LEGS = ["fht", "dhtgen", "inflate_zlib6", "inflate_wg", "inflate_own", "inflate_stream", "c5"]


def kname(s):
    return s.split("(")[0].replace("void ", "").strip()


def ours(k):
    # the engine's kernels, and the library kernels it launches itself (the radix sort of the inflate batches' job order)
    return "nxz" in k or "rocprim" in k or "hipcub" in k


def leg_info(leg, p):
    try:
        line = [l for l in open(os.path.join(src, "%s_%s.log" % (leg, p))) if l.startswith('{"leg"')]
        return json.loads(line[-1])
    except (OSError, IndexError, ValueError):
        return None


def pmc(leg, p):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for path in glob.glob(os.path.join(src, "%s_%s" % (leg, p), "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            k = kname(r["Kernel_Name"])
            if ours(k):
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                disp[k].add(r["Dispatch_Id"])
    return agg, {k: len(v) for k, v in disp.items()}


counters = {}
for leg in LEGS:
    info = leg_info(leg, "stats")
    if not info:
        continue
    units = info["units"] * info["passes"]                       # (the deflate pass that makes inflate_own's input counts only for its own kernels)
    stats = []
    for path in glob.glob(os.path.join(src, leg + "_stats", "**", "*kernel_stats.csv"), recursive=True):
        stats += [r for r in csv.DictReader(open(path)) if ours(r["Name"])]
    if stats:
        with open("%s_kernel_stats_%s.csv" % (prefix, leg), "w") as f:
            w = csv.DictWriter(f, fieldnames=list(stats[0].keys()))
            w.writeheader()
            w.writerows(stats)
    tot_ns = {kname(r["Name"]): float(r["TotalDurationNs"]) for r in stats}
    calls = {kname(r["Name"]): int(r["Calls"]) for r in stats}
    fetch, _ = pmc(leg, "fetch")
    write, _ = pmc(leg, "write")
    per = {}
    # which kernels belong to the leg's timed work (inflate_own runs one deflate pass to make its input)
    skip = ("nxzl77::", "nxze::", "nxzd::") if leg.startswith("inflate") else ()
    own_units = {"inflate_own": info["units"] * 2}.get(leg, units)
    for k in sorted(set(fetch) | set(write)):
        if k.startswith(skip):
            continue
        rd = fetch[k].get("FETCH_SIZE", 0.0) * 1024 * 2 / own_units
        wr = write[k].get("WRITE_SIZE", 0.0) * 1024 / own_units
        per[k] = {"read": round(rd), "write": round(wr), "avg_launch_ms": round(tot_ns.get(k, 0) / max(calls.get(k, 1), 1) * 1e-6, 4), "launches": calls.get(k)}
    total = sum(v["read"] + v["write"] for v in per.values())
    unit = {"fht": "64 KiB block", "dhtgen": "64 KiB block (corpus)", "inflate_zlib6": "stream of one 64 KiB block", "inflate_wg": "stream of one 64 KiB block", "inflate_own": "stream of one 64 KiB block",
            "inflate_stream": "64 KiB of output", "c5": "64 KiB block of the mixed batch, one whole step (compress + wrap + decompress + wrap)"}[leg]
    json.dump({"command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate runs) --output-format csv -- python3 tools/prof_workload.py %s" % leg,
               "unit": unit, "block_bytes": 65536, "units_per_pass": info["units"], "passes": info["passes"],
               "corrections": "KB units x 1024; FETCH_SIZE x 2 (gfx950, MI355X_MICROARCH.md HBM section); WRITE_SIZE as is",
               "algorithmic_bytes_per_unit": round(info["algorithmic_bytes_per_unit"]), "per_kernel": per, "traffic_bytes_per_block": total,
               "traffic_over_algorithmic": round(total / info["algorithmic_bytes_per_unit"], 3)},
              open("%s_pmc_traffic_%s.json" % (prefix, leg), "w"), indent=1)
    for p in ("sq1", "sq2"):
        agg, nd = pmc(leg, p)
        for k, v in agg.items():
            if k.startswith(skip):
                continue
            o = counters.setdefault(k + " [" + leg + "]", {"dispatches": nd[k], "units": own_units})
            for c, x in v.items():
                o[c + "_per_unit"] = round(x / own_units, 1)
            if k in tot_ns:
                o["cu_cycles_per_unit"] = round(tot_ns[k] * 1e-9 * CLOCK * CUS / own_units)
            if "SQ_WAVE_CYCLES" in v:
                o["wave_cycles_split"] = {c: round(v[c] / v["SQ_WAVE_CYCLES"], 3) for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY") if c in v}
            if "SQ_LDS_IDX_ACTIVE" in v and "SQ_LDS_BANK_CONFLICT" in v and v["SQ_LDS_IDX_ACTIVE"]:
                o["lds_bank_conflict_share_of_lds_cycles"] = round(v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"], 3)
                if "cu_cycles_per_unit" in o:
                    o["lds_busy_share_of_kernel_time"] = round(v["SQ_LDS_IDX_ACTIVE"] / own_units / o["cu_cycles_per_unit"], 3)
# keys bench.py's what_binds() reads
for k, o in list(counters.items()):
    if k.startswith("nxzl77::lz77_kernel<false>"):
        counters["nxzl77::lz77_kernel<false>"] = dict(o, SQ_INSTS_VALU_per_job=o.get("SQ_INSTS_VALU_per_unit"), cu_cycles_per_job=o.get("cu_cycles_per_unit"))
json.dump(counters, open(prefix + "_pmc_counters.json", "w"), indent=1, sort_keys=True)
print("wrote", prefix + "_*")
