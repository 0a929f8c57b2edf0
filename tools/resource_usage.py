#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel of the engine, from hipcc's -Rpass-analysis=kernel-resource-usage
(cross-compiles for gfx950 without a GPU).  usage: resource_usage.py [--json]   -> a table (or JSON) on stdout.
tests/test_kernel_resources.py holds the budgets a round must not drift over (round 2: the stream-per-wave inflate
kernel went from 96 to 99 VGPRs unnoticed and lost a wavefront per SIMD: 72.9 -> 64.4 GiB/s)."""
import json, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "power-gzip_amd", "csrc")
FILES = ["nxz_lz77.hip", "nxz_encode.hip", "nxz_dhtgen.hip", "nxz_inflate.hip", "nxz_inflate_lanes.hip", "nxz_inflate_wg.hip", "nxz_inflate_cut.hip", "nxz_misc.hip", "nxz_blockfind.hip"]


def demangle(names):
    try:
        p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
        return p.stdout.splitlines() if p.returncode == 0 and p.stdout else names
    except OSError:
        return names


def collect():
    out = {}
    for f in FILES:
        p = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", CSRC, "--cuda-device-only",
                            "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, f), "-o", "/dev/null"], capture_output=True, text=True)
        if p.returncode != 0:
            raise SystemExit("hipcc failed on %s:\n%s" % (f, p.stderr[-2000:]))
        cur = None
        for line in p.stderr.splitlines():
            m = re.search(r"Function Name: (\S+)", line)
            if m:
                cur = {"file": f}
                out[m.group(1)] = cur
                continue
            m = re.search(r"remark: +(TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)", line)
            if m and cur is not None:
                cur[m.group(1).split(" [")[0]] = int(m.group(2))
    names = list(out)
    return {re.sub(r"\(.*", "", d).replace("void ", ""): out[n] for n, d in zip(names, demangle(names))}


if __name__ == "__main__":
    res = collect()
    if "--json" in sys.argv:
        print(json.dumps(res, indent=1, sort_keys=True))
    else:
        print("%-58s %5s %5s %7s %6s %6s %6s" % ("kernel", "VGPR", "SGPR", "scratch", "occup", "sSpill", "vSpill"))
        for k, v in sorted(res.items()):
            print("%-58s %5d %5d %7d %6d %6d %6d" % (k[:58], v.get("VGPRs", 0), v.get("TotalSGPRs", 0), v.get("ScratchSize", 0), v.get("Occupancy", 0), v.get("SGPRs Spill", 0), v.get("VGPRs Spill", 0)))
