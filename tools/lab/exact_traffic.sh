# HBM-side requests of a profiling leg by size class (the L2's memory-side counters), in two rocprofv3 runs of their own:
# bytes read = 32 x RDREQ_32B + 64 x RDREQ_64B + 128 x RDREQ_128B, bytes written = 64 x WRREQ_64B + 32 x (WRREQ - WRREQ_64B).
#   tools/lab/exact_traffic.sh <leg> <out dir>
leg=${1:-inflate_wg}; out=${2:-gpurun_out/exact}
mkdir -p $out; cd /tmp 2>/dev/null; export TMPDIR=/tmp; cd - >/dev/null
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum --output-format csv -d $out/rd -- python3 tools/prof_workload.py $leg > $out/rd.log 2>&1
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --output-format csv -d $out/wr -- python3 tools/prof_workload.py $leg > $out/wr.log 2>&1
python3 - $out <<'PY'
import csv, glob, json, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
for d in ("rd", "wr"):
    for path in glob.glob("%s/%s/**/*counter_collection.csv" % (out, d), recursive=True):
        for r in csv.DictReader(open(path)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").strip()
            agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
info = json.loads([l for l in open(out + "/rd.log") if l.startswith('{"leg"')][-1])
units = info["units"] * info["passes"]
res = {}
for k, v in agg.items():
    if "nxz" not in k: continue
    rd = 32 * v.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * v.get("TCC_EA0_RDREQ_64B_sum", 0) + 128 * v.get("TCC_EA0_RDREQ_128B_sum", 0)
    wr = 64 * v.get("TCC_EA0_WRREQ_64B_sum", 0) + 32 * (v.get("TCC_EA0_WRREQ_sum", 0) - v.get("TCC_EA0_WRREQ_64B_sum", 0))
    res[k] = {"read_bytes_per_unit": round(rd / units), "write_bytes_per_unit": round(wr / units), "requests_per_unit": {c: round(x / units, 1) for c, x in v.items()}}
json.dump({"leg": info["leg"], "units": info["units"], "passes": info["passes"], "algorithmic_bytes_per_unit": round(info["algorithmic_bytes_per_unit"]), "per_kernel": res,
           "how": "rocprofv3 --pmc TCC_EA0_RDREQ[_32B|_64B|_128B]_sum and TCC_EA0_WRREQ[_64B]_sum, runs of their own; bytes by request size"}, open(out + "/exact_traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf $out/rd $out/wr
