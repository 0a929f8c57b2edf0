#!/usr/bin/env python3
"""Per-block figures from one profiling run of tools/bench_split.py under rocprofv3 (kernel stats pass + PMC passes,
each its own run as MI355X_MICROARCH.md's HBM / rocprofv3 sections prescribe).
usage: pmc_report.py <pmc_summary.json> <kernel_stats.csv> <jobs per launch> <out prefix>
writes <prefix>_pmc_counters.json, <prefix>_pmc_traffic_fht.json, <prefix>_pmc_traffic_dhtgen.json, <prefix>_kernel_stats.csv
Corrections: FETCH_SIZE / WRITE_SIZE are printed in KB (x1024); FETCH_SIZE x2 on gfx950 for wide coalesced reads (guide)."""
import csv, json, sys

pmc = json.load(open(sys.argv[1]))
stats = list(csv.DictReader(open(sys.argv[2])))
jobs = int(sys.argv[3])
prefix = sys.argv[4]
CLOCK = 2.4e9
CUS = 256
ours = [r for r in stats if "nxz" in r["Name"]]
with open(prefix + "_kernel_stats.csv", "w") as f:
    w = csv.DictWriter(f, fieldnames=list(stats[0].keys()))
    w.writeheader()
    for r in ours:
        w.writerow(r)
avg_ns = {r["Name"].split("(")[0].replace("void ", ""): float(r["AverageNs"]) for r in ours}
rep = {}
for k, v in pmc.items():
    d = v["dispatches"]
    n = d * jobs
    o = {"dispatches": d, "jobs_per_dispatch": jobs, "avg_ns": avg_ns.get(k)}
    if avg_ns.get(k):
        o["cu_cycles_per_job"] = round(avg_ns[k] * 1e-9 * CLOCK * CUS / jobs)
    for c in ("SQ_INSTS_LDS", "SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT",
              "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS"):
        if c in v:
            o[c + "_per_job"] = round(v[c] / n, 1)
    if "SQ_WAVE_CYCLES" in v:
        wc = v["SQ_WAVE_CYCLES"]
        o["wave_cycles_split"] = {c: round(v[c] / wc, 3) for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY") if c in v}
    if "SQ_LDS_IDX_ACTIVE" in v and "SQ_LDS_BANK_CONFLICT" in v:
        o["lds_bank_conflict_share_of_lds_cycles"] = round(v["SQ_LDS_BANK_CONFLICT"] / v["SQ_LDS_IDX_ACTIVE"], 3)
        if "cu_cycles_per_job" in o:
            # one LDS pipe per CU: cycles it is busy per job over the CU cycles a job takes
            o["lds_busy_share_of_kernel_time"] = round(v["SQ_LDS_IDX_ACTIVE"] / n / o["cu_cycles_per_job"], 3)
    rd = v.get("FETCH_SIZE", 0) * 1024 * 2 / n
    wr = v.get("WRITE_SIZE", 0) * 1024 / n
    o["hbm_read_bytes_per_job_corrected"] = round(rd)
    o["hbm_write_bytes_per_job"] = round(wr)
    rep[k] = o
json.dump(rep, open(prefix + "_pmc_counters.json", "w"), indent=1, sort_keys=True)
for name, lz, en, extra in (("fht", "nxzl77::lz77_kernel<false>", "nxze::encode_kernel<false>", []),
                            ("dhtgen", "nxzl77::lz77_kernel<true>", "nxze::encode_kernel<true>", ["nxzd::dhtgen_kernel"])):
    ks = [lz, en] + extra
    t = sum(rep[k]["hbm_read_bytes_per_job_corrected"] + rep[k]["hbm_write_bytes_per_job"] for k in ks if k in rep)
    json.dump({"command": "rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- python tools/bench_split.py 65536",
               "kernels": ks, "block_bytes": 65536, "jobs_per_launch": jobs,
               "corrections": "KB units x1024; FETCH_SIZE x2 (gfx950, MI355X_MICROARCH.md HBM section); WRITE_SIZE as is",
               "per_kernel": {k: {"read": rep[k]["hbm_read_bytes_per_job_corrected"], "write": rep[k]["hbm_write_bytes_per_job"]} for k in ks if k in rep},
               "traffic_bytes_per_block": t,
               "note": "sum over the kernels of the pipeline; the LZ77 kernel writes the block's tokens (two bitmaps and the match records, "
                       "NXZ_TOK_STRIDE apart) and the entropy kernel reads them and the source once more"},
              open("%s_pmc_traffic_%s.json" % (prefix, name), "w"), indent=1)
