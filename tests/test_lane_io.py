"""The inflate kernel's per-lane bit reader / output writer, compiled for the host and fuzzed under
AddressSanitizer (the header is the product code itself, power-gzip_amd/csrc/nxz_lane_io.h)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_lane_io_fuzz(tmp_path):
    exe = tmp_path / "lane_io_fuzz"
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize=alignment",
                    "-I", os.path.join(ROOT, "power-gzip_amd", "csrc"),
                    os.path.join(ROOT, "tests", "native", "lane_io_fuzz.cpp"), "-o", str(exe)], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr
