import os, sys, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch, zxc_amd
from zxc_amd import corpus
bs = 65536; mib = 256
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
for level in (1, 3, 5, 7):
    def step():
        rc = L.zxc_mi355x_encode_blocks_device(C.c_void_p(d_src.data_ptr()), n, bs, level, 0, C.c_void_p(d_slots.data_ptr()), C.c_void_p(d_sizes.data_ptr()), C.c_void_p(stream))
        assert rc == 0, rc
    step(); torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(2): step()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 2
    import hashlib
    h = hashlib.sha256(d_sizes.cpu().numpy().tobytes()).hexdigest()[:12]
    print(f"{os.environ.get('ZXC_LIB_VARIANT','default'):22s} L{level}: {n>>20} MiB in {ms:8.2f} ms = {n/ms/1e6:7.1f} GB/s  ratio {n/int(d_sizes.sum().item()):.4f} sizes-sha {h}", flush=True)
