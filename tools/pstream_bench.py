#!/usr/bin/env python3
"""Push-streaming throughput on the GPU box (pageable host buffers, one thread): 1 GiB of the bench's text through zxc_cstream_*
and back through zxc_dstream_*, fed in chunks of CHUNK MiB per call, next to zxc_compress / zxc_decompress on the same buffers.
Usage: pstream_bench.py [total MiB] [chunk MiB ...]; ZXC_MI355X_PSTREAM_WINDOW_MIB sets the launch window."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # one HIP runtime in the process (tests/conftest.py)
import zxc_amd.api as api

def main():
    total = (int(sys.argv[1]) if len(sys.argv) > 1 else 1024) << 20
    chunks = [int(a) for a in sys.argv[2:]] or [1, 8, 32, 128]
    L = api._bind_pstream(api.lib())
    L.zxc_mi355x_set_device(0)
    import multiprocessing as mp
    from zxc_amd import corpus
    pool = mp.get_context("spawn").Pool(max(1, min(32, os.cpu_count() or 1)))
    data = b"".join(pool.map(corpus.gen_chunk, corpus.enwik_chunks(total, seed=1)))[:total]  # the bench's encode text
    pool.close()
    total = len(data)
    src = C.create_string_buffer(data, total)
    cap = int(api.lib().zxc_compress_bound(total))
    arc = C.create_string_buffer(cap)
    dec = C.create_string_buffer(total + 4096)

    def run_c(chunk):
        o = api._CompressOpts(level=3, block_size=65536, checksum_enabled=0)
        cs = L.zxc_cstream_create(C.byref(o))
        out = api._OutBuf(C.addressof(arc), cap, 0)
        t = time.perf_counter()
        off = 0
        while off < total:
            n = min(chunk, total - off)
            inb = api._InBuf(C.addressof(src) + off, n, 0)
            while inb.pos < inb.size:
                r = L.zxc_cstream_compress(cs, C.byref(out), C.byref(inb))
                assert r >= 0, r
            off += n
        while True:
            r = L.zxc_cstream_end(cs, C.byref(out))
            assert r >= 0, r
            if r == 0:
                break
        dt = time.perf_counter() - t
        L.zxc_cstream_free(cs)
        return dt, out.pos

    def run_d(chunk, csize):
        ds = L.zxc_dstream_create(None)
        out = api._OutBuf(C.addressof(dec), total + 4096, 0)
        t = time.perf_counter()
        off = 0
        while not L.zxc_dstream_finished(ds):
            n = min(chunk, csize - off)
            inb = api._InBuf(C.addressof(arc) + off, n, 0)
            r = L.zxc_dstream_decompress(ds, C.byref(out), C.byref(inb))
            assert r >= 0 and (r > 0 or inb.pos > 0 or L.zxc_dstream_finished(ds)), (r, inb.pos)
            off += inb.pos
        dt = time.perf_counter() - t
        L.zxc_dstream_free(ds)
        return dt, out.pos

    print(f"{total >> 20} MiB of text, level 3, 64 KiB blocks, window {os.environ.get('ZXC_MI355X_PSTREAM_WINDOW_MIB', '128')} MiB")
    for ch in chunks:
        best_c = best_d = 1e9
        for _ in range(3):
            dt, csize = run_c(ch << 20)
            best_c = min(best_c, dt)
        # feed the decoder compressed chunks of the same ratio
        dch = max(1 << 16, int((ch << 20) * csize / total))
        for _ in range(3):
            dt, dsize = run_d(dch, csize)
            best_d = min(best_d, dt)
        assert dsize == total and dec.raw[:1 << 20] == data[:1 << 20] and C.string_at(C.addressof(dec) + total - 4096, 4096) == data[-4096:]
        print(f"chunk {ch:4d} MiB: cstream {total / best_c / 1e9:6.2f} GB/s of source (ratio {total / csize:.3f}), dstream {total / best_d / 1e9:6.2f} GB/s decoded", flush=True)
    o = api._CompressOpts(level=3, block_size=65536, seekable=0)
    best = 1e9
    for _ in range(3):
        t = time.perf_counter(); n = api.lib().zxc_compress(src, total, arc, cap, C.byref(o)); best = min(best, time.perf_counter() - t)
    print(f"zxc_compress   {total / best / 1e9:6.2f} GB/s of source")
    best = 1e9
    for _ in range(3):
        t = time.perf_counter(); r = api.lib().zxc_decompress(arc, n, dec, total, None); best = min(best, time.perf_counter() - t)
    assert r == total
    print(f"zxc_decompress {total / best / 1e9:6.2f} GB/s decoded")


if __name__ == "__main__":
    main()
