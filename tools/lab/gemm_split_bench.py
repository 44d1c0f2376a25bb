#!/usr/bin/env python
"""Lab: the split-precision panel GEMM next to the exact one at the path's shapes (POEM-medium, batch 32)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from poem_v2_amd import hip  # noqa: E402


def timeit(fn, n=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def main():
    dev = "cuda:0"
    L = hip.lib()
    for M, N, K, act in ((131072, 1536, 256, 0), (131072, 256, 256, 0), (25568, 768, 256, 0), (25568, 1280, 256, 2),
                         (25568, 256, 256, 0), (1048576, 256, 384, 1)):
        x = torch.randn(M, K, device=dev)
        w = torch.randn(N, K, device=dev) / K ** 0.5
        b = torch.randn(N, device=dev)
        wp = hip.pack_linear(w)
        nt = (N + 31) // 32
        img = torch.empty(nt * 32 * K * 4, dtype=torch.uint8, device=dev)
        sc = torch.empty(nt, dtype=torch.float32, device=dev)
        hip.check(L.poem_pack_split_gemm(hip.ptr(w), N, K, img.data_ptr(), sc.data_ptr(), hip.stream()))
        y = torch.empty(M, N, device=dev)
        te = timeit(lambda: hip.check(L.poem_gemm(hip.ptr(x), K, wp.data_ptr(), hip.ptr(b), None, 0, hip.ptr(y), N, M, N, K, act, hip.stream())))
        ts = timeit(lambda: hip.check(L.poem_gemm_split(hip.ptr(x), K, img.data_ptr(), sc.data_ptr(), hip.ptr(b), None, 0, hip.ptr(y), N, M, N, K, act, hip.stream())))
        fl = 2.0 * M * N * K
        by = 4.0 * (M * K + M * N)
        print(f"M={M} N={N} K={K} act={act}: exact {te:.3f} ms ({fl / te / 1e9:.0f} TFLOP/s)  split {ts:.3f} ms "
              f"({fl / ts / 1e9:.0f} TFLOP/s-equivalent, {by / ts / 1e6:.0f} GB/s of x + y)", flush=True)


if __name__ == "__main__":
    main()
