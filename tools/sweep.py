#!/usr/bin/env python3
"""One-off parity sweep (GPU box): every data class x level 1..7 x block size 4 KiB..2 MiB, reference encoder ->
device decoder (buffer API and seekable API), byte compare; plus device encoder -> reference decoder."""
import hashlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import zxc_amd, oracle_py
from zxc_amd import corpus
ref = oracle_py.Ref()
mib = int(sys.argv[1]) if len(sys.argv) > 1 else 3
classes = {c: corpus._GEN[c](mib << 20, corpus._rng(int(os.environ.get("SWEEP_SEED", "7")), 3)).tobytes() for c in corpus._GEN}
classes["zeros"] = bytes(mib << 20)
classes["random"] = np.random.default_rng(5).integers(0, 256, mib << 20, dtype=np.uint8).tobytes()
classes["mix"] = corpus.synth_silesia(mib << 20, seed=11)
if os.environ.get("SWEEP_SUBSET"):  # (tests/test_gpu_pipeline.py: three classes)
    classes = {k: classes[k] for k in ("mix", "zeros", "random")}
    part_len = min(700001, (mib << 20) - 12345)
else:
    part_len = 700001
bad = 0; n = 0; t0 = time.time()
for cname, data in classes.items():
    for level in range(1, 8):
        for bs in (4096, 65536, 524288, 2097152):
            for ck in ((False, True) if bs == 65536 else (False,)):
                comp = ref.compress(data, level, bs, True, ck)
                out = zxc_amd.decompress(comp, checksum=ck, raise_on_error=False)
                ok = isinstance(out, tuple) and out[0] == len(data) and out[1] == data
                if ok and bs == 65536:
                    s = zxc_amd.Seekable(comp); part = s.decompress_range(12345, part_len); s.close()
                    ok = part == data[12345:12345 + part_len]
                n += 1
                if not ok:
                    bad += 1; print("FAIL decode", cname, level, bs, ck, out[0] if isinstance(out, tuple) else None, flush=True)
        for bs in (4096, 65536, 524288, 2097152):  # device encoder -> reference decoder, every level and block size
            comp = zxc_amd.compress(data, level, bs, True, level == 5)
            rc, out = ref.decompress(comp, len(data), checksum=(level == 5))
            n += 1
            if rc != len(data) or out != data:
                bad += 1; print("FAIL encode", cname, level, bs, rc, flush=True)
print(f"{n} cases, {bad} failures, {time.time() - t0:.0f} s")
