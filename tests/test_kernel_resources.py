"""Register / scratch budgets of the HIP kernels, checked at build time without a GPU (hipcc cross-compiles for
gfx950; -Rpass-analysis=kernel-resource-usage).  Round 2 lost 12 % on the zlib -6 inflate leg because the
stream-per-wave kernel drifted from 96 to 99 VGPRs -- a wavefront per SIMD -- and only a slow bench showed it
(housekeeping item 8).  The budgets are the occupancy steps of MI355X_MICROARCH.md's register table:
<= 64 VGPRs: 8 waves per SIMD, 96: 5, 128: 4.  The full table of a round is kept in profiles/rNN_kernel_resource_usage.txt
(tools/resource_usage.py)."""
import importlib.util
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# kernel -> (max VGPRs, max scratch bytes per lane)
BUDGET = {
    "nxzl77::lz77_kernel<false, false, false>": (128, 64),   # 1024 threads per workgroup: 128 is the cap
    "nxzl77::lz77_kernel<true, false, false>": (128, 64),
    "nxzl77::lz77_kernel<false, true, false>": (128, 64),    # the fixed-Huffman form that writes the finished block
    "nxzl77::lz77_kernel<true, false, true>": (128, 64),     # the opt-in form that also makes the table and encodes
    "nxze::encode_kernel<false, false>": (64, 0),     # seven workgroups of 256 threads per CU
    "nxze::encode_kernel<true, true>": (64, 0),       # a caller's table (symbols may be missing: checked)
    "nxze::encode_kernel<true, false>": (64, 0),      # the table the device made of the block's own counts
    "nxzd::dhtgen_kernel": (64, 0),
    "nxzi::inflate_kernel<true, false>": (96, 0),     # a stream per wave, the target as window: five waves per SIMD
    "nxzi::inflate_kernel<true, true>": (96, 0),
    "nxzi::inflate_kernel<false, false>": (128, 0),   # window in LDS: LDS bounds the occupancy, not registers
    "nxzi::inflate_kernel<false, true>": (128, 0),
    "nxzl::inflate_lanes_kernel": (128, 48),          # a stream per lane, any block type: four waves per SIMD
    "nxzl::inflate_lanes_fixed_kernel": (80, 32),     # ... stored and fixed-code blocks only: six
    "nxzw::inflate_wg_kernel<false>": (128, 128),     # a stream per workgroup of 1024 threads: 128 is the cap; its phases are functions of their own
                                                      # (the scratch: the registers those functions save on entry, none in their loops)
    "nxzl::cksum_kernel<0>": (96, 0),
    "nxzl::cksum_kernel<1>": (96, 0),               # the WRAP function code: the same pass, storing as it goes
    "nxzl::cksum_kernel<2>": (96, 0),               # checksums and the outputs to the callers' pinned targets (nxu_run_job's rounds)
    "nxzb::find_blocks_kernel": (96, 0),
    "nxzi::token_sync_kernel": (72, 0),               # (LDS bounds it at five wavefronts per SIMD: 72 registers allow seven)
    "nxzi::block_tables_kernel": (64, 0),
    "nxzb::resolve_kernel": (64, 0),
    "nxzb::window_chain_kernel<true>": (128, 0),      # 1024 threads and 64 KiB of LDS per workgroup: two per CU whatever the registers
    "nxzb::window_chain_kernel<false>": (128, 0),
    "nxzb::compose_maps_kernel<true>": (128, 0),
    "nxzb::compose_maps_kernel<false>": (128, 0),
    "nxzb::check_headers_kernel": (96, 0),
    "nxz::pack_stream_kernel": (64, 0),
    "nxz::wrap_kernel": (64, 0),
}


@pytest.fixture(scope="module")
def usage():
    spec = importlib.util.spec_from_file_location("resource_usage", os.path.join(ROOT, "tools", "resource_usage.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.collect()


def test_every_kernel_stays_inside_its_register_budget(usage):
    over = []
    for k, (vmax, smax) in BUDGET.items():
        assert k in usage, (k, sorted(usage))
        u = usage[k]
        if u["VGPRs"] > vmax or u.get("ScratchSize", 0) > smax:
            over.append((k, u["VGPRs"], vmax, u.get("ScratchSize", 0), smax))
    assert not over, over


def test_no_kernel_is_missing_from_the_recorded_table(usage):
    """the table committed for the round lists every kernel the sources define (so a new kernel gets a line -- and,
    if it is on the hot path, a budget)"""
    rec = [f for f in sorted(os.listdir(os.path.join(ROOT, "profiles"))) if f.endswith("_kernel_resource_usage.txt")]
    assert rec, "run tools/resource_usage.py > profiles/rNN_kernel_resource_usage.txt"
    text = open(os.path.join(ROOT, "profiles", rec[-1])).read()
    missing = [k for k in usage if k[:58] not in text]
    assert not missing, missing
