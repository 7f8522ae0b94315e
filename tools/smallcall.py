#!/usr/bin/env python3
"""Latency of SMALL host-API calls (pageable buffers): zxc_compress / zxc_decompress of 64 KiB .. 64 MiB of text and one
zxc_compress_block / zxc_decompress_block — what a call costs besides its launch (round 5: all of them work in the staging arenas,
no hipMalloc / hipFree per call). Usage: smallcall.py"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def main():
    import torch  # noqa: F401  (one HIP runtime in the process)
    import zxc_amd.api as api
    import oracle_py
    from zxc_amd import corpus
    L = api.lib()
    L.zxc_mi355x_set_device(0)
    B = oracle_py.bind_block_api(L)
    data = b"".join(corpus.gen_chunk(c) for c in corpus.enwik_chunks(64 << 20, seed=1))[:64 << 20]
    cap = int(L.zxc_compress_bound(len(data)))
    cbuf = C.create_string_buffer(cap)
    dbuf = C.create_string_buffer(len(data) + 4096)

    def best(fn, n=7):
        fn()
        b = 1e9
        for _ in range(n):
            t = time.perf_counter()
            r = fn()
            b = min(b, time.perf_counter() - t)
        return r, b
    for size in (64 << 10, 1 << 20, 16 << 20, 64 << 20):
        o = api._CompressOpts(level=3, block_size=65536, seekable=0)
        src = data[:size]
        n, tc = best(lambda: L.zxc_compress(src, size, cbuf, cap, C.byref(o)))
        assert n > 0
        r, td = best(lambda: L.zxc_decompress(cbuf, n, dbuf, size, None))
        assert r == size and dbuf.raw[:size] == src
        print(f"{size >> 10:7d} KiB: zxc_compress {tc * 1e3:8.3f} ms ({size / tc / 1e9:6.2f} GB/s)   zxc_decompress {td * 1e3:8.3f} ms ({size / td / 1e9:6.2f} GB/s)", flush=True)
    cctx = B.zxc_create_cctx(None)
    dctx = B.zxc_create_dctx()
    o = oracle_py.CompressOpts(level=3, block_size=65536)
    blk = data[:65536]
    n, tc = best(lambda: B.zxc_compress_block(cctx, blk, 65536, cbuf, cap, C.byref(o)))
    assert n > 0
    blkc = cbuf.raw[:n]
    r, td = best(lambda: B.zxc_decompress_block(dctx, blkc, n, dbuf, 65536 + 2112, None))
    assert r == 65536 and dbuf.raw[:65536] == blk
    print(f"one 64 KiB block: zxc_compress_block {tc * 1e3:.3f} ms   zxc_decompress_block {td * 1e3:.3f} ms")


if __name__ == "__main__":
    main()
