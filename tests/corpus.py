"""Real-data corpus for the metric's kind of data (BASELINE.json: "Silesia 64KiB blocks").

load(block=65536) -> list of (class, name, bytes) cut at `block` bytes per file (the last chunk
of a file is short), from the first source that is present:

  1. $SILESIA_DIR: the 12 files of the Silesia corpus (dickens ... xml, optionally with the
     ".uncompressed" suffix the reference's downloader leaves: oct/download.sh), each verified
     against the sha256 the reference pins in /root/reference/oct/silesia-*.source (line 3 of
     every file; the hash is of the decompressed file).  The hashes below are those values,
     carried as data.  A file whose hash differs is refused.
  2. else a deterministic fallback that exists on any box of this image: alice29.txt (the
     reference's samples/alice29.txt, committed as the data fixture tests/golden/alice29.txt,
     sha256 pinned in SURVEY 8(d)) plus the first 4 MiB of a recorded list of system files of
     several kinds (ELF, XML, JSON, tables, sources, msgpack, font, base64; since round 5 also an
     image-like kind -- binary tables of 16- and 32-bit values that zlib -1 brings down to 1.5-1.8 : 1,
     what Silesia's mr / x-ray / sao are -- and a packed kind that hardly shrinks at all: a zip and a
     compressed code object).  Every entry carries
     the sha256 of the bytes used; a file that is missing or differs is skipped and reported.

Test infrastructure and bench.py's corpus leg only.
"""
import hashlib
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SILESIA_SHA256 = {   # /root/reference/oct/silesia-<name>.source, line 3
    "dickens": "b24c37886142e11d0ee687db6ab06f936207aa7f2ea1fd1d9a36763c7a507e6a",
    "mozilla": "657fc3764b0c75ac9de9623125705831ebbfbe08fed248df73bc2dc66e2a963b",
    "mr": "68637ed52e3e4860174ed2dc0840ac77d5f1a60abbcb13770d5754e3774d53e6",
    "nci": "fc63a31770947b8c2062d3b19ca94c00485a232bb91b502021948fee983e1635",
    "ooffice": "e7ee013880d34dd5208283d0d3d91b07f442e067454276095ded14f322a656eb",
    "osdb": "60f027179302ca3ad87c58ac90b6be72ec23588aaa7a3b7fe8ecc0f11def3fa3",
    "reymont": "0eac0114a3dfe6e2ee1f345a0f79d653cb26c3bc9f0ed79238af4933422b7578",
    "samba": "93ba07bc44d8267789c1d911992f40b089ffa2140b4a160fac11ccae9a40e7b2",
    "sao": "c2d0ea2cc59d4c21b7fe43a71499342a00cbe530a1d5548770e91ecd6214adcc",
    "webster": "6a68f69b26daf09f9dd84f7470368553194a0b294fcfa80f1604efb11143a383",
    "x-ray": "7de9fce1405dc44ae5e6813ed21cd5751e761bd4265655a005d39b9685d1c9ad",
    "xml": "0e82e54e695c1938e4193448022543845b33020c8be6bf3bf3ead2224903e08c",
}
SILESIA_CLASS = {"dickens": "text", "mozilla": "exe", "mr": "image", "nci": "database", "ooffice": "exe", "osdb": "database",
                 "reymont": "pdf-text", "samba": "source", "sao": "binary", "webster": "text", "x-ray": "image", "xml": "xml"}

ALICE29 = ("text", os.path.join(ROOT, "tests", "golden", "alice29.txt"), 152089,
           "7467306ee0feed4971260f3c87421154a05be571d944e9cb021a5713700c38f0")
FALLBACK_CAP = 4 << 20
FALLBACK = [   # (class, path, bytes used, sha256 of those bytes) -- recorded in this image, ROCm 7.2.0 / Ubuntu 22.04
    ALICE29,
    ("text", "/usr/lib/python3.10/pydoc_data/topics.py", 745585, "7f6750d35de084e727f24822bc65b5576ec607e1a8ef85d423b6fa38d8bb3301"),
    ("xml", "/usr/share/mime/packages/freedesktop.org.xml", 2376295, "0269019c2ce7bcb5608c651c1418097f0d312d4c7b68ecc35bce570147bf727e"),
    ("elf", "/usr/lib/x86_64-linux-gnu/libc.so.6", 2220400, "9b846df01200e522f8c57c1c0e4435777180d6fa031fa2baa6fa34f68562c4d9"),
    ("elf", "/usr/lib/x86_64-linux-gnu/libcrypto.so.3", 4194304, "12db6185ceaf39095a5c21dc73b8ee6c8a8dc89fd8759343dc6a430d30ebaa9a"),
    ("table", "/opt/rocm/share/miopen/db/gfx90a68.db.txt", 4194304, "115f9f8f24e53467f0a9fb7990ba256c73c430c8643c33048cf8846a4a32a4d2"),
    ("table", "/usr/share/perl/5.34.0/Unicode/Collate/allkeys.txt", 1939332, "a3255d45b7af97f4dc14fb8364d7573b434425e5c58cacf00d16901ce081c78d"),
    ("json", "/usr/local/lib/python3.10/dist-packages/dash/html/metadata.json", 484981, "e840d1f4ed664aa665e67fbc0bafc87ced7820affd266d662d9edb374fff2bcb"),
    ("msgpack", "/opt/rocm/lib/hipblaslt/library/TensileLibrary_SS_SS_HA_Bias_SAV_UA_Type_SS_Contraction_l_Ailk_Bljk_Cijk_Dijk_gfx942.dat", 4194304, "8c82f80d970b41a336a48c25667b83fb0c861387deab1f684ef6400ed81d3ad8"),
    ("font", "/usr/share/fonts/truetype/dejavu/DejaVuSans.ttf", 757076, "690243adfefe0ce154b547db6205794bd30ac4277275179517a90994f4980648"),
    ("base64", "/etc/ssl/certs/ca-certificates.crt", 222392, "3a8b34c06e15fb1172bb8d0b7bc2eaf83433ebfd4bfc467e1099468fe4a51061"),
    ("source", "/usr/lib/python3.10/typing.py", 92557, "ec7b7f73fc92827c78a7d2aff90cffe070530cad6c693460165c26f76d195f41"),
    ("source", "/opt/rocm/include/hip/amd_detail/amd_hip_runtime.h", 14276, "e51a973fd5dd9e07300cb7c5a51fbd2059531abc33f67a45d7fe7cdef79552c8"),
    ("image", "/usr/lib/x86_64-linux-gnu/libicudata.so.70", 4194304, "a8419487114b78f91ca7454ee4444efca74934a7b3a130fe6911b8745022aa63"),
    ("image", "/usr/lib/x86_64-linux-gnu/gconv/libCNS.so", 473096, "626ff0bd6f6c82866cd5be2ba12c01586566123f9cde41a5a1be9d3e3d9f4d94"),
    ("packed", "/usr/local/lib/python3.10/dist-packages/scipy/special/tests/data/boost.npz", 1270643, "d73ecbbb51654522342ba0470a6263a9684e617c2b8374565fe3a79593f4b231"),
    ("packed", "/opt/rocm/lib/hipblaslt/library/TensileLibrary_B8F8_SB8F8_HA_Bias_SAB_SAV_UA_Type_B8S_HPA_Contraction_l_Ailk_Bljk_Cijk_Dijk_gfx950.co", 1227962, "77e54e7a7b33c8bd91b5a88ff0c7af69046eb4e8f925086e72c3c363ecc7d27b"),
]

# The twelve files of the Silesia corpus: published size in bytes, and the class of THIS corpus whose measured rate stands in
# for it when bench.py says what the engine would read on Silesia (config.silesia_weighted_GiB_s).  Where no class is close
# the slowest binary one is taken (osdb: a MySQL table file; reymont: a PDF -- Polish text above 0x7f, rule 3c does not see
# it as text).
SILESIA_BYTES = {"dickens": 10192446, "mozilla": 51220480, "mr": 9970564, "nci": 33553445, "ooffice": 6152192, "osdb": 10085684,
                 "reymont": 6627202, "samba": 21606400, "sao": 7251944, "webster": 41458703, "x-ray": 8474240, "xml": 5345280}
SILESIA_AS_FALLBACK_CLASS = {"dickens": "text", "webster": "text", "mozilla": "elf", "ooffice": "elf", "osdb": "elf", "reymont": "elf",
                             "samba": "source", "nci": "table", "xml": "xml", "mr": "image", "x-ray": "image", "sao": "image"}


def _cut(cls, name, data, block):
    return [(cls, name, data[i:i + block]) for i in range(0, len(data), block)]


def load_silesia(directory, block=65536):
    """the 12 Silesia files from `directory`, sha256-checked; raises ValueError on a wrong or missing file"""
    out = []
    for name, want in SILESIA_SHA256.items():
        path = None
        for cand in (name, name + ".uncompressed", "silesia-" + name + ".uncompressed"):
            p = os.path.join(directory, cand)
            if os.path.isfile(p):
                path = p
                break
        if path is None:
            raise ValueError("Silesia file %s not found in %s" % (name, directory))
        data = open(path, "rb").read()
        if hashlib.sha256(data).hexdigest() != want:
            raise ValueError("%s: sha256 differs from the pin of oct/silesia-%s.source" % (path, name))
        out += _cut(SILESIA_CLASS[name], name, data, block)
    return out


def load_fallback(block=65536):
    """(blocks, report): the recorded fallback files that are present and unchanged"""
    out, report = [], {"used": [], "skipped": []}
    for cls, path, size, want in FALLBACK:
        try:
            data = open(path, "rb").read(FALLBACK_CAP)
        except OSError:
            report["skipped"].append(os.path.basename(path) + " (missing)")
            continue
        if len(data) != size or hashlib.sha256(data).hexdigest() != want:
            report["skipped"].append(os.path.basename(path) + " (differs from the recorded bytes)")
            continue
        report["used"].append(os.path.basename(path))
        out += _cut(cls, os.path.basename(path), data, block)
    return out, report


def load(block=65536):
    """(name, blocks, report).  name: 'silesia' or 'fallback'"""
    d = os.environ.get("SILESIA_DIR")
    if d:
        return "silesia", load_silesia(d, block), {"dir": d, "files": list(SILESIA_SHA256)}
    blocks, report = load_fallback(block)
    if not blocks:
        raise RuntimeError("no corpus: set SILESIA_DIR or restore tests/golden/alice29.txt")
    return "fallback", blocks, report
