#!/usr/bin/env python3
"""One-off differential fuzz (GPU box): random bit flips / byte stomps / truncations of every synth archive,
device decode vs the oracle: same accept/reject, same error code, same bytes. Usage: fuzzdiff.py [iterations]"""
import json, os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import zxc_amd, oracle_py
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1500
man = json.load(open(os.path.join(ROOT, "tests/golden/MANIFEST.json")))["synth"]
O = oracle_py.Oracle(); rng = random.Random(2026)
names = [n for n in man if man[n]["size"] <= 400000]
bad = 0; acc = 0
for it in range(iters):
    name = rng.choice(names); meta = man[name]
    m = bytearray(open(os.path.join(ROOT, "tests/golden/synth", name + ".zxc"), "rb").read())
    kind = rng.randrange(4)
    if kind == 0:
        for _ in range(rng.choice((1, 1, 2, 4))): m[rng.randrange(16, len(m))] ^= 1 << rng.randrange(8)
    elif kind == 1:
        p = rng.randrange(16, len(m)); m[p:p + rng.randrange(1, 9)] = bytes(rng.randrange(256) for _ in range(rng.randrange(1, 9)))
    elif kind == 2:
        p = rng.randrange(16, len(m) - 8); m[p] = rng.choice((0, 0xFF, 0x80, 0x7F, 0xE0))
    else:
        cut = rng.randrange(20, len(m)); del m[cut:cut + rng.randrange(1, 64)]
    m = bytes(m); ck = bool(meta["checksum"]) and rng.random() < 0.5
    a, ao = O.decompress(m, meta["size"], checksum=ck)
    b, bo = zxc_amd.decompress(m, meta["size"], checksum=ck, raise_on_error=False)
    if a != b or (a >= 0 and ao != bo):
        bad += 1
        print("MISMATCH", name, "kind", kind, "oracle", a, "gpu", b, flush=True)
        if bad <= 3: open(os.path.join(ROOT, "gpurun_out", f"fuzz_fail_{bad}.zxc"), "wb").write(m)
    acc += a >= 0
print(f"{iters} mutants, {acc} still valid, {bad} mismatches")
