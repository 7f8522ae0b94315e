#!/usr/bin/env python3
"""Where a chunk's wall time goes in the encoder (GPU box): ZXC_TOOLS_AB=1 ZXC_LIB_VARIANT=libzxc_clk.so python tools/encclk.py  (ENC_LEVEL, default 3)
Needs the -DEXP_ENC_CLOCKS build of the encode kernel and the shim (zxc_encode_kernel.hip: ENC_T). One launch over 256 MiB of text;
prints every phase's share of the summed per-wave clocks and the clocks per chunk of 64 positions."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if __name__ == "__main__":
    import ctypes as C, torch, zxc_amd
    from zxc_amd import corpus
    bs = 65536; level = int(os.environ.get("ENC_LEVEL", "3"))
    data = corpus.synth_text(64 << 20, seed=1)
    dev = torch.device("cuda", 0)
    base = torch.frombuffer(bytearray(data), dtype=torch.uint8).to(dev)
    d_src = torch.cat([base.repeat(4), torch.zeros(256, dtype=torch.uint8, device=dev)])
    n = d_src.numel() - 256; nb = (n + bs - 1) // bs
    L = zxc_amd.lib()
    stride = L.zxc_mi355x_encode_slot_stride(bs)
    d_slots = torch.empty(nb * stride, dtype=torch.uint8, device=dev)
    d_sizes = torch.zeros(nb, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    def step():
        rc = L.zxc_mi355x_encode_blocks_device(C.c_void_p(d_src.data_ptr()), n, bs, level, 0, C.c_void_p(d_slots.data_ptr()), C.c_void_p(d_sizes.data_ptr()), C.c_void_p(stream))
        assert rc == 0, rc
    out = (C.c_ulonglong * 8)()
    step(); torch.cuda.synchronize()
    assert L.zxc_mi355x_exp_enc_clocks(out) == 0
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record(); step(); e1.record(); torch.cuda.synchronize()
    assert L.zxc_mi355x_exp_enc_clocks(out) == 0
    names = ["set-up (tables cleared)", "own bytes + hash + head lookup + publish", "chain links + candidate requests until the bytes are here",
             "first compares + second 16 bytes", "extension beyond 32 bytes", "round bookkeeping", "scalar parse", "emission"]
    tot = sum(out); chunks = n / 64
    print(f"level {level}: {n >> 20} MiB in {e0.elapsed_time(e1):.2f} ms (instrumented build), {tot / chunks:.0f} clocks per chunk of 64 positions")
    for k in range(8):
        print(f"  {names[k]:60s} {100 * out[k] / tot:5.1f} %   {out[k] / chunks:7.0f} clocks per chunk")
