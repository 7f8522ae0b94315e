#!/usr/bin/env python3
"""Regenerates the dictionary of the reference's golden cases 09_block_dict / 12_glo_huffman_dict
(reference tests/format/golden_cases.h:64-245) so those two frozen archives can be DECODED by the
parity tests instead of only rejected with DICT_REQUIRED:

* gc_dict.bin      the fixed dictionary content (golden_cases.h:177-180)
* gc_dict_huf.bin  the 128-byte shared literal table = zxc_train_dict_huf(payload 12, dict), computed
                   by the unmodified reference (oracle/_ref), like gc_dict_huf_table() (:226-245)
* MANIFEST.json    decoded_sha256 / dict fields of the two cases; the decoded bytes must equal the
                   payload generators restated below (:184-223), checked here with the reference decoder.

Runs in the build container only (needs oracle/_ref); the outputs are committed fixtures.
"""
import ctypes as C
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))

DICT = (b"GET /api/v1/users/ HTTP/1.1\r\nHost: api.example.com\r\n"
        b"Accept: application/json\r\nUser-Agent: zxc-client\r\n")


def lcg(s):
    return (s * 1103515245 + 12345) & 0xFFFFFFFF


def dict_payload():  # gc_make_dict_payload
    req = (b"GET /api/v1/users/4242/profile HTTP/1.1\r\nHost: api.example.com\r\n"
           b"Accept: application/json\r\nUser-Agent: zxc-client\r\n\r\n")
    return bytes(req[i % len(req)] for i in range(4096))


def huffman_dict_payload():  # gc_make_huffman_dict_payload
    cap, st, out = 4096, 0x5EEDCAFE, b""
    while len(out) + 160 < cap:
        st = lcg(st); user = st % 100000
        st = lcg(st); session = st
        st = lcg(st); page = st % 64
        out += (b"GET /api/v1/users/%u/profile?session=%08x&page=%u HTTP/1.1\r\n"
                b"Host: api.example.com\r\nAccept: application/json\r\n"
                b"User-Agent: zxc-client\r\n\r\n" % (user, session, page))
    return out


def main():
    import oracle_py
    ref = oracle_py.Ref()
    L = ref.lib
    p12 = huffman_dict_payload()
    huf = C.create_string_buffer(128)
    samples = (C.c_char_p * 1)(p12)
    sizes = (C.c_size_t * 1)(len(p12))
    L.zxc_train_dict_huf.restype = C.c_int
    L.zxc_train_dict_huf.argtypes = [C.POINTER(C.c_char_p), C.POINTER(C.c_size_t), C.c_size_t, C.c_char_p, C.c_size_t, C.c_char_p]
    assert L.zxc_train_dict_huf(samples, sizes, 1, DICT, len(DICT), huf) == 0
    open(os.path.join(HERE, "format", "gc_dict.bin"), "wb").write(DICT)
    open(os.path.join(HERE, "format", "gc_dict_huf.bin"), "wb").write(huf.raw)
    man = json.load(open(os.path.join(HERE, "MANIFEST.json")))
    for name, payload, table in (("09_block_dict.zxc", dict_payload(), None), ("12_glo_huffman_dict.zxc", p12, huf.raw)):
        comp = open(os.path.join(HERE, "format", name), "rb").read()
        o = oracle_py.DecompressOpts(checksum_enabled=0)
        keep = (C.create_string_buffer(DICT, len(DICT)), C.create_string_buffer(table, 128) if table else None)
        o.dict = C.cast(keep[0], C.c_void_p); o.dict_size = len(DICT)
        o.dict_huf = C.cast(keep[1], C.c_void_p) if table else None
        out = C.create_string_buffer(len(payload) + 64)
        rc = L.zxc_decompress(comp, len(comp), out, len(payload), C.byref(o))
        assert rc == len(payload) and out.raw[:rc] == payload, (name, rc)
        man["format"][name]["decoded_sha256"] = hashlib.sha256(payload).hexdigest()
        man["format"][name]["dict"] = "gc_dict.bin"
        man["format"][name]["dict_huf"] = "gc_dict_huf.bin" if table else None
        print(name, "decodes with the regenerated dictionary:", rc, "bytes")
    json.dump(man, open(os.path.join(HERE, "MANIFEST.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
