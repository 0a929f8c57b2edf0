"""Short seeded soaks of every kernel family, collected by pytest so that the driver runs them too (round 3 ran the
soak scripts by hand only).  Each is the script of the same name with a small budget (<= 20 s of GPU work); the long
runs stay what they were: python tests/soak_*.py, results in profiles/rNN_soak.txt."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(script, *args, timeout=600):
    env = dict(os.environ, GRAFT_REPO_ROOT=ROOT)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", script)] + [str(a) for a in args], env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert p.returncode == 0, (p.stdout[-1500:], p.stderr[-1500:])
    return p.stdout


def test_deflate_blocks_against_the_oracle():
    """random kinds, sizes at tile / piece edges, histories: the LZ77 + entropy kernels == oracle, bit for bit"""
    assert "SOAK OK" in _run("soak_gpu.py", 2, 600)


def test_deflate_is_stable_pass_after_pass_and_round_trips():
    """the same 8192 blocks compressed pass after pass: every pass equal to the first (a race shows as an unstable
    block), inflated back on the device"""
    out = _run("soak_deflate_roundtrip_gpu.py", 8192, 10)
    assert "deflate-unstable blocks 0, round-trip failures 0" in out, out[-800:]


def test_inflate_kernels_against_the_oracle():
    """random zlib streams (levels, strategies, flushes, histories, cut-off tails) through the batched inflate kernels (a stream per lane with and without the fixed-code-only kernel, a stream per wave twice)"""
    assert "SOAK OK" in _run("soak_inflate_gpu.py", 1)


def test_one_stream_and_stream_parts():
    """long streams of mixed pieces through nxz_inflate_stream and, in steps of random sizes, through nx_inflate"""
    assert "SOAK OK" in _run("soak_stream_gpu.py", 6)
