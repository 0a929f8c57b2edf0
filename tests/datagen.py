"""Seeded synthetic inputs shared by tests and bench (no reference files needed at run time).

Recipes follow the reference's own generators (values restated, not copied):
  text33  -- uniform text over a 33-symbol alphabet, /root/reference/test/test_utils.c:22-28,152-161
  lz      -- makedata-style LZ copies, /root/reference/samples/makedata.c:36-73: seed half of the
             buffer with random bytes of a reduced alphabet, then repeat {dist in [1,dist_max],
             len in [16, len_max+15]} copies
"""
import random

ALPHABET33 = b"abcdefghijklmnopqrstuvwxyz .,;!?\n"
assert len(ALPHABET33) == 33


def make_block(kind: str, n: int, seed: int = 0) -> bytes:
    rnd = random.Random((0x9E3779B97F4A7C15 ^ seed) & 0xFFFFFFFFFFFFFFFF)
    if kind == "zeros":
        return bytes(n)
    if kind == "random":
        return rnd.randbytes(n)
    if kind == "text33":
        return bytes(rnd.choices(ALPHABET33, k=n))
    if kind == "alice":
        return ALICE_LIKE(n, seed)
    if kind == "lz":
        buf = bytearray(rnd.choices(ALPHABET33, k=max(1, n // 2)))
        len_max = rnd.randrange(10, 250)
        dist_max = rnd.randrange(1, 65537)
        while len(buf) < n:
            dist = rnd.randrange(1, min(dist_max, len(buf)) + 1)
            ln = rnd.randrange(16, len_max + 16)
            for _ in range(ln):
                buf.append(buf[-dist])
        return bytes(buf[:n])
    if kind == "periodic":
        # a short random period repeated: every position sits inside one very long match
        period = rnd.choice([2, 3, 5, 7, 63, 64, 65, 300, 511, 512, 513, 4099])
        unit = rnd.randbytes(period)
        return (unit * (n // period + 1))[:n]
    if kind == "binary":
        # two symbols in runs: every hash slot collides, runs of all lengths
        out = bytearray()
        while len(out) < n:
            out += bytes([rnd.choice(b"ab")]) * rnd.choice([1, 1, 2, 3, 4, 5, 8, 9, 17, 40, 41, 260, 600])
        return bytes(out[:n])
    if kind == "sparse":
        # long copies of far-away text with single-byte edits: long matches, broken chains,
        # candidates that change in the middle of a match
        buf = bytearray(rnd.choices(ALPHABET33, k=max(1, n // 4)))
        while len(buf) < n:
            src = rnd.randrange(0, len(buf))
            ln = rnd.choice([9, 12, 24, 39, 40, 41, 72, 258, 259, 300, 700, 1500])
            chunk = bytearray(buf[src:src + ln])
            if chunk and rnd.random() < 0.5:
                chunk[rnd.randrange(len(chunk))] ^= 1
            buf += chunk
        return bytes(buf[:n])
    raise ValueError(kind)


_WORDS = ("the of and a to in is you that it he was for on are as with his they I at be this have from or one "
          "had by word but not what all were we when your can said there use an each which she do how their if "
          "will up other about out many then them these so some her would make like him into time has look two "
          "more write go see number no way could people my than first water been call who oil its now find long "
          "down day did get come made may part Alice rabbit queen hatter").split()


def ALICE_LIKE(n: int, seed: int = 1) -> bytes:
    """English-like text (Zipf-ish word choice) -- a stand-in with alice29-like statistics."""
    rnd = random.Random(seed)
    weights = [1.0 / (i + 1) for i in range(len(_WORDS))]
    out = []
    size = 0
    while size < n:
        w = rnd.choices(_WORDS, weights)[0]
        sep = rnd.choice([" ", " ", " ", " ", ", ", ". ", "\n"])
        out.append(w + sep)
        size += len(w) + len(sep)
    return "".join(out).encode()[:n]
