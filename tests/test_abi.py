"""CPU tests of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/nxz_engine.h declares, and the wire structures have the reference's layout.
No compute calls (no GPU here)."""
import ctypes as C
import importlib
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pkg = importlib.import_module("power-gzip_amd")

# offsets probed from the reference's inc_nx/nxu.h with offsetof() (recorded as data)
REF_LAYOUT = {
    "sizeof(nxz_crb_t)": 256, "sizeof(nxz_crb_cpb_t)": 2048,
    "sizeof(nxz_dde_t)": 16, "sizeof(nxz_csb_t)": 16,
    "offsetof(nxz_crb_t, source)": 16, "offsetof(nxz_crb_t, target)": 32, "offsetof(nxz_crb_t, csb)": 240,
    "offsetof(nxz_crb_cpb_t, cpb)": 256,
    "offsetof(nxz_cpb_t, in_crc_le)": 4, "offsetof(nxz_cpb_t, in_w2_be)": 8, "offsetof(nxz_cpb_t, in_w3_be)": 12,
    "offsetof(nxz_cpb_t, in_dht)": 16, "offsetof(nxz_cpb_t, out_adler_be)": 384, "offsetof(nxz_cpb_t, out_crc_le)": 388,
    "offsetof(nxz_cpb_t, out_w2_be)": 392, "offsetof(nxz_cpb_t, out_w3_be)": 396, "offsetof(nxz_cpb_t, u)": 400,
    "offsetof(nxz_cpb_t, u.d.out_spbc_decomp_be)": 688, "offsetof(nxz_cpb_t, out_spbc_with_count_be)": 1664,
    "offsetof(nxz_dev_t, paste_addr)": 32, "offsetof(nxz_dev_t, fd)": 40, "offsetof(nxz_dev_t, function)": 44,
    "sizeof(nxz_batch_job_t)": 48, "sizeof(nxz_batch_result_t)": 32, "sizeof(nxz_batch_dht_t)": 296,
}


def test_wire_layout_matches_reference(tmp_path):
    src = ['#include <stdio.h>', '#include <stddef.h>', '#include "nxz_engine.h"', 'int main(void){']
    for k in REF_LAYOUT:
        src.append('printf("%s=%%zu\\n", (size_t)%s);' % (k.replace('"', ''), k))
    src.append("return 0;}")
    c = tmp_path / "layout.c"
    c.write_text("\n".join(src))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-I", os.path.join(ROOT, "include"), str(c), "-o", str(exe)], check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout
    got = dict(line.rsplit("=", 1) for line in out.strip().splitlines())
    for k, v in REF_LAYOUT.items():
        assert int(got[k]) == v, k


def test_library_exports_every_declared_symbol():
    p = pkg.lib_path()
    assert os.path.exists(p), "build first: __graft_entry__.build()"
    lib = C.CDLL(p)
    hdr = open(os.path.join(ROOT, "include", "nxz_engine.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(nx[uz]?_\w+|__crc32_vpmsum|nx_\w+)\s*\(", hdr))
    names |= {"tb_freq"}
    names = {n for n in names if not n.endswith("_t")}
    assert {"nx_function_begin", "nx_function_end", "nxu_run_job", "nx_wait_ticks", "__crc32_vpmsum",
            "nxz_batch_compress", "nxz_batch_decompress", "nxz_batch_wrap", "nxz_ctx_create"} <= names
    for n in sorted(names):
        assert hasattr(lib, n), "missing export: " + n
    assert C.c_uint64.in_dll(lib, "tb_freq").value == 512000000


def test_host_side_helpers_without_gpu():
    """__crc32_vpmsum and nx_wait_ticks are pure host code (lib/crc32_ppc.c:22-67 wraps the former)."""
    import zlib
    L = pkg.engine.load_library()
    data = bytes(range(256)) * 37 + b"tail"
    for init in (0, 0x12345678):
        # crc32_ppc: crc = ~crc; crc = __crc32_vpmsum(crc, p, len); return ~crc
        raw = L.__crc32_vpmsum((~init) & 0xffffffff, data, len(data))
        assert (~raw) & 0xffffffff == zlib.crc32(data, init)
    t = L.nx_wait_ticks(100, 5, 0)
    assert t >= 105


def test_engine_refuses_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(pkg.EngineError):
        pkg.Engine(0)


def test_preload_library_exports_the_references_versioned_abi():
    """f1: libnxz_preload.so (the nx_* layer and the unprefixed zlib names in one library, like the
    reference's libnxz.so) defines exactly the symbols of the reference's ABI dump (test/libnxz.abi, carried
    as tests/golden/libnxz_abi_symbols.json), each under the version node lib/Versions gives it -- plus
    deflateParams, which the reference leaves to real zlib (test/zlib.supp) and this library answers itself."""
    import json
    import subprocess
    want = {(s["name"], s["version"]) for s in json.load(open(os.path.join(ROOT, "tests", "golden", "libnxz_abi_symbols.json")))["symbols"]}
    assert len(want) == 76
    out = subprocess.run(["readelf", "--dyn-syms", "-W", os.path.join(ROOT, "power-gzip_amd", "libnxz_preload.so")],
                         capture_output=True, text=True, check=True).stdout
    got = set()
    for line in out.splitlines():
        f = line.split()
        if len(f) < 8 or f[6] == "UND" or f[4] != "GLOBAL" or f[3] not in ("FUNC", "OBJECT"):
            continue
        name, _, ver = f[7].partition("@@")
        if f[3] == "OBJECT" and name.startswith(("LIBNXZ_", "ZLIB_")):
            continue                                       # the version nodes themselves
        got.add((name, ver))
    assert got - want == {("deflateParams", "")}, sorted(got - want)
    assert want - got == set(), sorted(want - got)


def test_device_selection_policy_spreads_threads_over_the_gpus():
    """Row (e) / verdict item 6: a caller that names no device (NX_GZIP_DEV_NUM unset: "nx_id -1 means open any",
    /root/reference lib/nx_zlib.c:568-576,1281-1287 -- the reference takes the unit nearest the calling CPU, so an
    N-thread process uses every engine) gets a device per THREAD: the first thread the current device, the others
    the next devices in turn; an explicit ordinal pins.  The policy is a pure function, called here with made-up
    device counts (no GPU needed)."""
    import ctypes as C
    L = pkg.engine.load_library()
    f = L.nxz_pick_device
    f.restype = C.c_int
    f.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint, C.c_int]
    # one thread: what it always got -- the current device
    assert f(-1, 8, 0, 0, 1) == 0 and f(-1, 8, 5, 0, 1) == 5
    # 16 threads on 8 GPUs: every GPU gets two of them
    got = [f(-1, 8, 0, t, 1) for t in range(16)]
    assert sorted(got) == sorted(list(range(8)) * 2)
    # starting from the current device, wrapping
    assert [f(-1, 4, 2, t, 1) for t in range(5)] == [2, 3, 0, 1, 2]
    # the policy switched off (NXZ_DEVICE_POLICY=current): everybody on the current device
    assert {f(-1, 8, 3, t, 0) for t in range(16)} == {3}
    # an explicit ordinal pins, whatever the thread; out of range is refused
    assert {f(6, 8, 0, t, 1) for t in range(16)} == {6}
    assert f(8, 8, 0, 0, 1) == -1 and f(0, 0, 0, 0, 1) == -1
    # one GPU: everything on it
    assert {f(-1, 1, 0, t, 1) for t in range(9)} == {0}
