"""The workgroup-per-stream inflate kernel (power-gzip_amd/csrc/nxz_inflate_wg.hip, the product source itself) run on the CPU:
tests/native/hip_cpu_shim.h gives a workgroup an OS thread per lane, barriers, ballots and shuffles, tests/native/inflate_wg_sim.cpp
feeds it streams made by system zlib -- text, zeros, periods, stored blocks, huffman-only, mixed; streams it must hand back (too
long, target too small, cut short, damaged) -- and compares every byte and every result record.  No GPU: this is the check that
the algorithm (pieces in rounds, records, bitmap, resolve) is right before the kernel ever reaches the device."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


PARAMS = [(1, 128, 16)] + ([(2, 160, 1024), (3, 250, 0)] if os.environ.get("NXZ_SIM_FULL") else [])   # (two minutes a run; nres: pieces left in a round up to which a wavefront walks each)


@pytest.mark.skipif(not os.path.exists(CLANG), reason="the kernel source uses clang builtins: needs ROCm's clang++")
@pytest.mark.parametrize("seed,pmin,nres", PARAMS)
def test_workgroup_inflate_kernel_on_the_cpu(tmp_path, seed, pmin, nres):
    exe = tmp_path / "inflate_wg_sim"
    subprocess.run([CLANG, "-O1", "-g", "-std=c++17", "-pthread", os.path.join(ROOT, "tests", "native", "inflate_wg_sim.cpp"),
                    "-o", str(exe), "-lz"], check=True)
    r = subprocess.run([str(exe), os.path.join(ROOT, "tests", "golden", "alice29.txt"), str(seed), str(pmin), str(nres), "70000"],
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout + r.stderr
