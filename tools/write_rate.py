#!/usr/bin/env python3
"""What a store-only kernel sustains on this box, next to the read+write copy rate: the ceiling a write-dominated kernel (the first layer:
20 MB read, 436 MB written) can be held to.  usage (GPU box): python tools/write_rate.py > gpurun_out/write_rate.txt"""
import pathlib
import sys
sys.path.insert(0, str(pathlib.Path(__file__).resolve().parent.parent))
import torch
from supervised_dispnet_amd import _lib


def rate(fn, nbytes, reps=8):
    best = 0.0
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        e1.synchronize()
        best = max(best, nbytes / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    return best


def main():
    dev = torch.device("cuda:0")
    st = torch.cuda.current_stream().cuda_stream
    for mb in (436, 1024, 4096):
        n = mb * (1 << 20) // 4
        a = torch.empty(n, dtype=torch.float32, device=dev)
        b = torch.empty(n, dtype=torch.float32, device=dev)
        fill = rate(lambda: _lib.call("dn_fill", a.data_ptr(), 1.0, n, st), 4.0 * n)
        tfill = rate(lambda: a.fill_(2.0), 4.0 * n)
        copy = rate(lambda: _lib.call("dn_ubench_copy", a.data_ptr(), b.data_ptr(), n, st), 8.0 * n)
        pat = [rate(lambda m=m: _lib.call("dn_ubench_store", a.data_ptr(), n, m, st), 4.0 * n) for m in (0, 1)]
        print("%5d MiB: 16-byte stores, 1 KiB contiguous per instruction %7.0f GB/s | 64-byte segments at 256-byte stride (4 instructions per row) %7.0f GB/s" % (mb, pat[0], pat[1]))
        print("%5d MiB: dn_fill (store only) %7.0f GB/s | torch fill_ %7.0f GB/s | dn_ubench_copy (read + write) %7.0f GB/s" % (mb, fill, tfill, copy))
        del a, b


if __name__ == "__main__":
    main()
