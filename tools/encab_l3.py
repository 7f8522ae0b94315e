#!/usr/bin/env python3
"""Encoder A/B at one level on the bench's text (GPU box): python tools/encab_l3.py libA.so libB.so ...  (ENC_LEVEL, default 3; ENC_MIB, default 256)
One child process per library variant (ZXC_LIB_VARIANT): GB/s of source, ratio, a hash of the block sizes."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--one":
    sys.path.insert(0, ROOT)
    import ctypes as C, hashlib, torch, zxc_amd
    from zxc_amd import corpus
    bs = 65536; mib = int(os.environ.get("ENC_MIB", "256")); level = int(os.environ.get("ENC_LEVEL", "3"))
    data = corpus.synth_text(64 << 20, seed=1)
    dev = torch.device("cuda", 0)
    base = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev)
    d_src = torch.cat([base.repeat(mib // 64), torch.zeros(256, dtype=torch.uint8, device=dev)])
    n = d_src.numel() - 256; nb = (n + bs - 1) // bs
    L = zxc_amd.lib()
    stride = L.zxc_mi355x_encode_slot_stride(bs)
    d_slots = torch.empty(nb * stride, dtype=torch.uint8, device=dev)
    d_sizes = torch.zeros(nb, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    def step():
        rc = L.zxc_mi355x_encode_blocks_device(C.c_void_p(d_src.data_ptr()), n, bs, level, 0, C.c_void_p(d_slots.data_ptr()), C.c_void_p(d_sizes.data_ptr()), C.c_void_p(stream))
        assert rc == 0, rc
    step(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record(); step(); step(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 2)
    h = hashlib.sha256(d_sizes.cpu().numpy().tobytes()).hexdigest()[:12]
    print(f"{os.environ.get('ZXC_LIB_VARIANT','libzxc_mi355x.so'):24s} L{level}: {n>>20} MiB in {best:8.2f} ms = {n/best/1e6:7.1f} GB/s  ratio {n/int(d_sizes.sum().item()):.4f} sizes-sha {h}", flush=True)
elif __name__ == "__main__":
    for lib in sys.argv[1:]:
        env = dict(os.environ); env["ZXC_LIB_VARIANT"] = lib; env["ZXC_TOOLS_AB"] = "1"
        try:
            subprocess.run([sys.executable, os.path.abspath(__file__), "--one"], env=env, timeout=float(os.environ.get("AB_TIMEOUT", "120")))
        except subprocess.TimeoutExpired:
            print(f"{lib:24s} TIMEOUT", flush=True)
