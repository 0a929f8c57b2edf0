"""nx_inflate's large-call path keeps its device buffer sets per DEVICE (advisor finding of round 3: one pool for all
contexts leaked a set's buffers and stream whenever callers on different devices took turns).  The device side is
played by tests/native/slot_pool_host.cpp under the CPU model of the host layer."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_devices_take_turns_without_leaking(tmp_path):
    exe = tmp_path / "slot_pool_host"
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-rdynamic", os.path.join(ROOT, "tests", "native", "slot_pool_host.cpp"), "-o", str(exe),
                    "-L", os.path.join(ROOT, "oracle"), "-lnxz_amd_model", "-Wl,-rpath," + os.path.join(ROOT, "oracle"), "-lz", "-lpthread"], check=True)
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("ok "), r.stdout + r.stderr
