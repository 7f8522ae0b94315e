#!/usr/bin/env python3
"""The bench's host-API leg on its own (GPU box): zxc_compress / zxc_decompress of a 1 GiB frame and the seekable range calls on pageable
host buffers, PCIe both ways inside the timed region. ZXC_MI355X_PLAIN_COPIES=1: plain hipMemcpy instead of the copy engine."""
import argparse, json, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
if __name__ == "__main__":
    import torch, bench, zxc_amd
    zxc_amd.lib().zxc_mi355x_set_device(0)
    print(json.dumps(bench.host_api_run(argparse.Namespace(block_size=65536), torch.device("cuda", 0))))
