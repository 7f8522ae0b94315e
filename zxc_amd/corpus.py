"""Deterministic synthetic corpora shaped like the benchmark inputs BASELINE.json names.

There is no silesia.tar / enwik9 on the box and no network, so the workload is
generated: ``synth_silesia(n)`` lays out byte classes in the member proportions of
the real silesia.tar (text, executables, 16-bit images, chemical-database text,
fixed-width DB rows, source code, XML, a binary star catalogue) so that the
reference encoder at level 3 / 64 KiB blocks lands at a silesia-like ratio and
mostly GLO blocks. ``ZXC_CORPUS_DIR`` overrides with real files.

Everything is seeded (numpy PCG64 keyed by a splitmix64 of the segment id) so the
same bytes come out on every box.
"""
import os

import numpy as np

_GOLDEN = 0x9E3779B97F4A7C15
_MASK = (1 << 64) - 1


def _splitmix64(x: int) -> int:
    x = (x + _GOLDEN) & _MASK
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _MASK
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK
    return z ^ (z >> 31)


def _rng(seed: int, seg: int) -> np.random.Generator:
    return np.random.Generator(np.random.PCG64(_splitmix64(seed ^ (seg * 0x100000001B3))))


def _concat_by_index(pool: np.ndarray, starts: np.ndarray, lens: np.ndarray) -> np.ndarray:
    """Concatenate pool[starts[i] : starts[i]+lens[i]] for all i (vectorised gather)."""
    total = int(lens.sum())
    ends = np.cumsum(lens)
    begin = ends - lens
    idx = np.arange(total, dtype=np.int64)
    idx += np.repeat(starts - begin, lens)
    return pool[idx]


def _vocab(rng, n_words, alphabet, min_len, max_len, sep):
    lens = rng.integers(min_len, max_len + 1, n_words)
    letters = rng.choice(np.frombuffer(alphabet, dtype=np.uint8), int(lens.sum()),
                         p=None).astype(np.uint8)
    # every word carries its trailing separator so a gather yields running text
    seps = rng.choice(np.frombuffer(sep, dtype=np.uint8), n_words)
    out = np.empty(int(lens.sum()) + n_words, dtype=np.uint8)
    ends = np.cumsum(lens + 1)
    starts = ends - (lens + 1)
    body_idx = np.arange(int(lens.sum()), dtype=np.int64) + np.repeat(np.arange(n_words), lens)
    out[body_idx] = letters
    out[ends - 1] = seps
    return out, starts, lens + 1


def gen_text(n: int, rng, n_words=50000, zipf_a=1.07, phrase_frac=0.30) -> np.ndarray:
    """English-like text: Zipf vocabulary + reused multi-word phrases."""
    pool, starts, lens = _vocab(rng, n_words, b"etaoinshrdlcumwfgypbvkjxqz" * 2 + b"eeeaaooiitnn",
                                2, 11, b"      ,.\n ")
    ranks = np.arange(1, n_words + 1, dtype=np.float64)
    p = ranks ** (-zipf_a)
    cdf = np.cumsum(p / p.sum())
    n_tok = int(n / 5.2) + 64
    w = np.searchsorted(cdf, rng.random(n_tok)).astype(np.int64).clip(0, n_words - 1)
    # phrases: short runs of word ids reused verbatim (creates LZ matches of 10-40 B)
    n_phr = 6000
    phr_len = rng.integers(2, 7, n_phr)
    phr_start = rng.integers(0, n_tok - 8, n_phr)
    sel = np.flatnonzero(rng.random(n_tok) < (phrase_frac / 4.0))
    pp = 1.0 / np.arange(1, n_phr + 1)
    pid = np.searchsorted(np.cumsum(pp / pp.sum()), rng.random(sel.size)).clip(0, n_phr - 1)
    for k in range(6):  # overwrite w[sel+k] with the phrase's k-th word where the phrase is long enough
        m = phr_len[pid] > k
        tgt = sel[m] + k
        ok = tgt < n_tok
        w[tgt[ok]] = w[(phr_start[pid[m]] + k)[ok]]
    out = _concat_by_index(pool, starts[w], lens[w])
    return out[:n] if out.size >= n else np.resize(out, n)


def gen_source(n: int, rng) -> np.ndarray:
    """Source-code / XML-like: templated lines with identifiers from a small Zipf pool."""
    ids_pool, ids_s, ids_l = _vocab(rng, 2000, b"abcdefghijklmnopqrstuvwxyz_", 3, 14, b"_")
    ids_l = ids_l - 1  # drop separator
    templates = [b"    if (%s != NULL) {\n", b"        return %s;\n", b"static int %s(void *ctx, size_t n)\n{\n",
                 b"    %s = 0;\n", b"#include <%s.h>\n", b"  <item name=\"%s\" value=\"0\"/>\n",
                 b"    for (i = 0; i < %s; i++) {\n", b"}\n\n", b"    /* %s */\n", b"  </%s>\n", b"  <%s>\n",
                 b"    %s->next = NULL;\n"]
    n_lines = n // 24 + 16
    t = rng.integers(0, len(templates), n_lines)
    p = 1.0 / np.arange(1, 2001) ** 1.1
    ident = np.searchsorted(np.cumsum(p / p.sum()), rng.random(n_lines)).clip(0, 1999)
    # vectorised assembly: pre/post halves of each template around the identifier
    pre = [tp.split(b"%s")[0] for tp in templates]
    post = [tp.split(b"%s")[1] if b"%s" in tp else b"" for tp in templates]
    has = np.array([b"%s" in tp for tp in templates])
    blob = b"".join(pre) + b"".join(post)
    blob = np.frombuffer(blob, dtype=np.uint8)
    pre_l = np.array([len(x) for x in pre]); post_l = np.array([len(x) for x in post])
    pre_s = np.cumsum(pre_l) - pre_l
    post_s = pre_l.sum() + np.cumsum(post_l) - post_l
    pool = np.concatenate([blob, ids_pool])
    seg_s = np.stack([pre_s[t], blob.size + ids_s[ident], post_s[t]], axis=1).reshape(-1)
    seg_l = np.stack([pre_l[t], np.where(has[t], ids_l[ident], 0), post_l[t]], axis=1).reshape(-1)
    out = _concat_by_index(pool, seg_s, seg_l)
    return out[:n] if out.size >= n else np.resize(out, n)


def gen_records(n: int, rng, rec=128) -> np.ndarray:
    """Fixed-width DB rows: incrementing key + low-cardinality columns + some noise."""
    rows = n // rec + 1
    a = np.zeros((rows, rec), dtype=np.uint8)
    key = np.arange(rows, dtype=np.uint64)
    a[:, 0:8] = key.view(np.uint8).reshape(rows, 8)
    for c in range(8, rec - 16, 8):
        card = int(rng.integers(3, 400))
        vals = rng.integers(0, 256, (card, 8), dtype=np.uint8)
        a[:, c:c + 8] = vals[rng.integers(0, card, rows)]
    a[:, rec - 16:rec - 8] = rng.integers(0, 256, (rows, 8), dtype=np.uint8)  # incompressible column
    a[:, rec - 8:] = 0x20
    return a.reshape(-1)[:n]


def gen_exe(n: int, rng) -> np.ndarray:
    """Executable-like (mozilla/ooffice class): instruction idioms drawn Zipf-style from a pool
    of 6000 short fragments, interleaved with skewed "immediate" bytes."""
    n_frag = 6000
    flen = rng.integers(3, 14, n_frag)
    hist = rng.dirichlet(np.full(256, 0.25))
    pool = rng.choice(256, int(flen.sum()), p=hist).astype(np.uint8)
    fstart = np.cumsum(flen) - flen
    imm_len = 4096
    imm = rng.integers(0, 256, imm_len, dtype=np.uint8)
    pool = np.concatenate([pool, imm])
    k = int(n / 6.0) + 64
    p = 1.0 / np.arange(1, n_frag + 1) ** 0.95
    f = np.searchsorted(np.cumsum(p / p.sum()), rng.random(k)).clip(0, n_frag - 1)
    ilen = rng.choice(np.array([0, 0, 1, 1, 2, 4]), k)                  # immediate bytes after each fragment
    istart = int(flen.sum()) + rng.integers(0, imm_len - 8, k)
    seg_s = np.stack([fstart[f], istart], axis=1).reshape(-1)
    seg_l = np.stack([flen[f], ilen], axis=1).reshape(-1)
    out = _concat_by_index(pool, seg_s, seg_l)
    # fresh random immediates so they do not all match the small pool
    m = rng.random(out.size) < 0.06
    out = out.copy()
    out[m] = rng.integers(0, 256, int(m.sum()), dtype=np.uint8)
    return out[:n] if out.size >= n else np.resize(out, n)


def gen_image16(n: int, rng, width=512) -> np.ndarray:
    """16-bit medical-image-like (mr / x-ray class): smooth field, few grey levels per region,
    little-endian u16; background rows repeat."""
    px = n // 2 + 1
    rows = px // width + 1
    y = np.cumsum(rng.normal(0, 1.0, rows))[:, None]
    x = np.cumsum(rng.normal(0, 1.0, width))[None, :]
    img = 900 + 6 * (y + x) + 300 * np.sin(y / 25.0) * np.cos(x / 33.0)
    img = (np.round(img / 8.0) * 8.0) + rng.integers(0, 2, (rows, width)) * 8
    bg = (np.abs(np.sin(y / 40.0)) < 0.25) & (np.abs(x) > -1)           # dark background bands
    img = np.where(bg, 16.0 + rng.integers(0, 2, (rows, width)) * 8, img)
    img = np.clip(img, 0, 4095).astype("<u2")
    return img.view(np.uint8).reshape(-1)[:n]


def gen_catalogue(n: int, rng, rec=28) -> np.ndarray:
    """Binary star-catalogue-like: 28-byte records of slowly varying float32 + ids."""
    rows = n // rec + 1
    a = np.zeros((rows, rec), dtype=np.uint8)
    for c in range(0, 20, 4):
        v = np.cumsum(rng.normal(0, 1e-3, rows)).astype("<f4")
        a[:, c:c + 4] = v.view(np.uint8).reshape(rows, 4)
    a[:, 20:24] = np.arange(rows, dtype="<u4").view(np.uint8).reshape(rows, 4)
    cls = rng.integers(0, 256, (40, 4), dtype=np.uint8)               # spectral-class-like column
    a[:, 24:28] = cls[rng.integers(0, 40, rows)]
    a[:, 0:2] = 0
    a[:, 4:6] = 0
    return a.reshape(-1)[:n]


def gen_chem(n: int, rng) -> np.ndarray:
    """nci-like: highly repetitive fixed-format chemical-table text."""
    atoms = [b"C", b"N", b"O", b"H", b"S", b"Cl"]
    lines = []
    for _ in range(512):
        x, y, z = rng.integers(-9, 10, 3)
        at = atoms[int(rng.integers(0, len(atoms)))]
        lines.append(b"   %2d.%04d   %2d.%04d    0.0000 %-3s 0  0  0  0  0  0  0  0  0  0  0  0\n" %
                     (x, int(rng.integers(0, 10000)), y, int(rng.integers(0, 10000)), at))
    pool = np.frombuffer(b"".join(lines), dtype=np.uint8)
    ll = np.array([len(x) for x in lines]); ls = np.cumsum(ll) - ll
    k = n // int(ll.mean()) + 8
    p = 1.0 / np.arange(1, 513) ** 0.9
    pick = np.searchsorted(np.cumsum(p / p.sum()), rng.random(k)).clip(0, 511)
    out = _concat_by_index(pool, ls[pick], ll[pick])
    return out[:n] if out.size >= n else np.resize(out, n)


# (class, fraction of silesia.tar's 211 947 520 bytes) — member sizes from the corpus' own listing
_SILESIA_MIX = [
    ("text", 0.048),     # dickens
    ("exe", 0.242),      # mozilla
    ("image16", 0.047),  # mr
    ("chem", 0.158),     # nci
    ("exe", 0.029),      # ooffice
    ("records", 0.048),  # osdb
    ("text", 0.031),     # reymont
    ("source", 0.102),   # samba
    ("catalogue", 0.034),  # sao
    ("text", 0.196),     # webster
    ("source", 0.025),   # xml
    ("image16", 0.040),  # x-ray
]
_GEN = {"enwik": None, "text": gen_text, "exe": gen_exe, "image16": gen_image16, "chem": gen_chem,
        "records": gen_records, "source": gen_source, "catalogue": gen_catalogue}

SILESIA_BYTES = 211947520


def synth_silesia(n: int = SILESIA_BYTES, seed: int = 0) -> bytes:
    """n bytes in silesia.tar's class mix (or the real file from ZXC_CORPUS_DIR)."""
    d = os.environ.get("ZXC_CORPUS_DIR")
    if d and os.path.exists(os.path.join(d, "silesia.tar")):
        with open(os.path.join(d, "silesia.tar"), "rb") as f:
            return f.read(n)
    parts = []
    total = sum(f for _, f in _SILESIA_MIX)
    done = 0
    for i, (cls, frac) in enumerate(_SILESIA_MIX):
        m = n - done if i == len(_SILESIA_MIX) - 1 else int(n * frac / total)
        if m <= 0:
            continue
        parts.append(np.ascontiguousarray(_GEN[cls](m, _rng(seed, i))[:m]))
        done += m
    return np.concatenate(parts).tobytes()


def synth_text(n: int, seed: int = 1) -> bytes:
    """enwik9-like: text + XML mix (config 3, the match-finder workload)."""
    a = gen_text(int(n * 0.8), _rng(seed, 100))
    b = gen_source(n - a.size, _rng(seed, 101))
    return np.concatenate([a, b]).tobytes()[:n]


# ---- the reference's own deterministic test generators (tests/test_common.c:10-145) -------------
def ref_seek_data(n: int, seed: int = 0) -> bytes:
    """fill_seek_data: (seed + 17*i + (i>>8)) & 255  (tests/test_common.c:139-143)."""
    i = np.arange(n, dtype=np.uint64)
    return ((seed + 17 * i + (i >> 8)) & 255).astype(np.uint8).tobytes()


def ref_lorem(n: int) -> bytes:
    """Lorem-ipsum loop (LZ-friendly), like the reference's gen_lz_data (tests/test_common.c)."""
    s = (b"Lorem ipsum dolor sit amet, consectetur adipiscing elit, sed do eiusmod tempor incididunt "
         b"ut labore et dolore magna aliqua. ")
    return (s * (n // len(s) + 1))[:n]


def small_offset_pattern(n: int) -> bytes:
    """'ABCDE' period-5 run: forces 8-bit offsets and overlapping matches."""
    return (b"ABCDE" * (n // 5 + 1))[:n]


def period300(n: int, seed: int = 7) -> bytes:
    """300-byte random period: forces 16-bit offsets with long matches."""
    p = _rng(seed, 300).integers(0, 256, 300, dtype=np.uint8).tobytes()
    return (p * (n // 300 + 1))[:n]


# ---- tiled corpus for the benchmark ("tiled with a per-tile 64-bit xor-rotated seed so no two
# tiles are identical") -------------------------------------------------------------------------------------
# One tile = silesia.tar's class mix, cut to a whole number of 64 KiB blocks, so tiles (and the archives made
# from them) concatenate into ONE seekable corpus whose block i lives in tile i // TILE_BLOCKS. A tile is
# generated in independent chunks of at most CHUNK_BYTES (own generator state each), so that a process pool
# fills it in parallel; tile t of seed s is always the same bytes whatever the pool size.
TILE_BLOCKS = SILESIA_BYTES // 65536          # 3234 blocks of 64 KiB
TILE_BYTES = TILE_BLOCKS * 65536              # 211 943 424 B (silesia.tar's size minus its last 4096 B)
CHUNK_BYTES = 8 << 20


def tile_seed(tile: int, seed: int = 0) -> int:
    """64-bit xor-rotated per-tile seed."""
    x = (_GOLDEN ^ ((tile + 1) * 0xD6E8FEB86659FD93) ^ seed) & _MASK
    r = 17 + (tile % 31)
    return ((x << r) | (x >> (64 - r))) & _MASK


def tile_chunks(tile: int, seed: int = 0):
    """[(class, n_bytes, rng_seed, rng_segment)] — the independent pieces of one tile, in layout order."""
    ts = tile_seed(tile, seed)
    total = sum(f for _, f in _SILESIA_MIX)
    out = []
    done = 0
    for i, (cls, frac) in enumerate(_SILESIA_MIX):
        m = TILE_BYTES - done if i == len(_SILESIA_MIX) - 1 else int(TILE_BYTES * frac / total)
        done += m
        k = 0
        while m > 0:
            c = min(m, CHUNK_BYTES)
            out.append((cls, c, ts, i * 64 + k))
            m -= c
            k += 1
    return out


def gen_enwik(n: int, rng) -> np.ndarray:
    """enwik9-like chunk: 80 % text + 20 % XML / source (config 3, the match-finder workload)."""
    a = gen_text(int(n * 0.8), rng)
    b = gen_source(n - a.size, rng)
    return np.concatenate([a, b])[:n]


def enwik_chunks(n: int, seed: int = 1):
    """[(class, n_bytes, rng_seed, rng_segment)] covering n bytes of unique enwik-like text in CHUNK_BYTES pieces."""
    out, k = [], 0
    while n > 0:
        c = min(n, CHUNK_BYTES)
        out.append(("enwik", c, tile_seed(k, seed), k))
        n -= c
        k += 1
    return out


def gen_chunk(task) -> bytes:
    """Pool worker: one chunk of a tile (numpy only, no GPU state)."""
    cls, n, ts, seg = task
    return np.ascontiguousarray(_GEN[cls](n, _rng(ts, seg))[:n]).tobytes()


def synth_silesia_tile(tile: int, seed: int = 0, pool=None) -> bytes:
    """TILE_BYTES bytes of tile `tile` (pool: anything with .map, e.g. multiprocessing.Pool)."""
    tasks = tile_chunks(tile, seed)
    parts = pool.map(gen_chunk, tasks) if pool is not None else [gen_chunk(t) for t in tasks]
    return b"".join(parts)


_GEN["enwik"] = gen_enwik
