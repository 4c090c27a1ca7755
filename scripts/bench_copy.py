"""Practical HBM rates of this box next to the 8 TB/s peak the roofline uses: device-to-device copy (read + write),
fill (write only) and a reduction (read only) over 8 GiB, as torch runs them.  usage: python3 scripts/bench_copy.py"""
import time

import torch


def rate(fn, bytes_moved, reps=10):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return bytes_moved * reps / (time.perf_counter() - t0) / 1e9


def main():
    n = 1 << 30
    a = torch.empty(n, dtype=torch.int64, device="cuda")
    b = torch.empty(n, dtype=torch.int64, device="cuda")
    a.fill_(3)
    print("copy  (8 GiB read + 8 GiB write): %7.0f GB/s" % rate(lambda: b.copy_(a), 16 * n))
    print("fill  (8 GiB write):              %7.0f GB/s" % rate(lambda: b.fill_(7), 8 * n))
    print("sum   (8 GiB read):               %7.0f GB/s" % rate(lambda: a.sum(), 8 * n))


main()
