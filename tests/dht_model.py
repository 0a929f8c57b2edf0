"""CPU model of the device-side dhtgen (power-gzip_amd/csrc/nxz_dhtgen.hip), step for step.

The kernel builds a dynamic-Huffman header from 286 + 30 symbol counts the way the
reference's dhtgen() does (lib/nx_dhtgen.c:945-1034) but in a form a 64-lane wavefront can
run: rank sort instead of qsort, two-queue merge that records parents, code lengths by
pointer jumping, canonical codes by per-length ranks, and a run-length coder in which every
position of the length array decides by itself which symbol (if any) it emits.  This module
restates those steps with plain loops so that the reformulation can be checked against the
oracle's nxo_dhtgen (itself pinned to the reference binary) without a GPU.  Test
infrastructure only.
"""

CL_LEN = [5, 7, 6, 5, 5, 4, 4, 3, 3, 3, 3, 4, 5, 5, 4, 7, 6, 5, 6]
CL_ORDER = [16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15]


def _bitrev(v, n):
    r = 0
    for i in range(n):
        r |= ((v >> i) & 1) << (n - 1 - i)
    return r


def canon(lens):
    """canonical codes (bit-reversed) by per-length ranks: code = next[len] + #earlier symbols of that length"""
    cnt = [0] * 16
    for l in lens:
        cnt[l] += 1
    cnt[0] = 0
    nxt = [0] * 16
    c = 0
    for b in range(1, 16):
        c = (c + cnt[b - 1]) << 1
        nxt[b] = c
    codes = []
    seen = [0] * 16
    for l in lens:
        if l:
            codes.append(_bitrev(nxt[l] + seen[l], l))
            seen[l] += 1
        else:
            codes.append(0)
    return codes


def lengths(hist, nsym):
    """length-limited code lengths of one alphabet; hist is modified like the reference does"""
    limit = 1 << 14
    while True:
        s = sum(hist[:nsym])
        div = (s + limit - 1) // limit
        if div:
            for i in range(nsym):
                hist[i] = (hist[i] + div - 1) // div
        limit = limit * 3 // 4
        lens = [0] * nsym
        # rank sort of the used symbols by (count, symbol)
        used = [i for i in range(nsym) if hist[i]]
        n = len(used)
        if n == 0:
            return lens
        if n == 1:
            lens[used[0]] = 1
            return lens
        key = {i: (hist[i] << 9) | i for i in used}
        L = [0] * n
        S = [0] * n
        for i in used:
            r = sum(1 for j in used if key[j] < key[i])
            L[r] = hist[i]
            S[r] = i
        # two-queue merge, parents recorded (node k is the parent "k")
        N = []
        lpar = [0] * n
        npar = [0] * (n - 1)
        li = ni = 0
        while (n - li) + (len(N) - ni) > 1:
            w = 0
            k = len(N)
            for _ in range(2):
                if li < n and (ni >= len(N) or L[li] <= N[ni]):
                    w += L[li]; lpar[li] = k; li += 1
                else:
                    w += N[ni]; npar[ni] = k; ni += 1
            N.append(w)
        root = len(N) - 1
        # depth of node k below the root by pointer jumping: 5 rounds cover depth < 32
        d = [0 if k == root else 1 for k in range(len(N))]
        p = [root if k == root else npar[k] for k in range(len(N))]
        for _ in range(5):
            nd = [d[k] + d[p[k]] for k in range(len(N))]
            np_ = [p[p[k]] for k in range(len(N))]
            d, p = nd, np_
        deep = any(p[k] != root for k in range(len(N)))
        maxd = 0
        for x in range(n):
            dl = d[lpar[x]] + 1
            maxd = max(maxd, dl)
            lens[S[x]] = dl
        if not deep and maxd <= 15:
            return lens


def rle_emit(lens):
    """every position decides alone: returns a list of (symbol, extra_value, extra_bits) or None per position"""
    total = len(lens)
    out = [None] * total
    start = [0] * total
    end = [0] * total
    for i in range(total):
        start[i] = i if (i == 0 or lens[i] != lens[i - 1]) else start[i - 1]
    for i in range(total - 1, -1, -1):
        end[i] = i + 1 if (i == total - 1 or lens[i + 1] != lens[i]) else end[i + 1]
    for i in range(total):
        v = lens[i]
        k = i - start[i]
        R = end[i] - start[i]
        if v:
            if k == 0:
                out[i] = (v, 0, 0)
            else:
                kk = k - 1
                g0 = kk - kk % 6
                g = min(6, R - 1 - g0)
                if g >= 3:
                    if kk == g0:
                        out[i] = (16, g - 3, 2)
                else:
                    out[i] = (v, 0, 0)
        else:
            g0 = k - k % 138
            g = min(138, R - g0)
            if g >= 11:
                if k == g0:
                    out[i] = (18, g - 11, 7)
            elif g >= 3:
                if k == g0:
                    out[i] = (17, g - 3, 3)
            else:
                out[i] = (0, 0, 0)
    return out


def dhtgen(ll, d):
    """(bytes, dhtlen, ll_lens, d_lens) from 286 + 30 counts (lists, modified in place)"""
    ll_lens = lengths(ll, 286)
    d_lens = lengths(d, 30)
    lens = ll_lens + d_lens
    clcode = canon(CL_LEN)
    acc = 0
    n = 0

    def put(v, b):
        nonlocal acc, n
        acc |= v << n
        n += b
    put(286 - 257, 5); put(30 - 1, 5); put(19 - 4, 4)
    for i in range(19):
        put(CL_LEN[CL_ORDER[i]], 3)
    for e in rle_emit(lens):
        if e is not None:
            s, xv, xb = e
            put(clcode[s], CL_LEN[s])
            if xb:
                put(xv, xb)
    return acc.to_bytes((n + 7) // 8, "little"), n, ll_lens, d_lens
