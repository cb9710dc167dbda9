#!/usr/bin/env python3
"""Which kernels the vendor library runs for the DiT / square shapes (names only; run under rocprofv3 --kernel-trace)."""
import torch
for M, N, K in [(8192, 8192, 8192), (4992, 6144, 2048), (4992, 2048, 2048), (4992, 8192, 2048), (4992, 2048, 8192)]:
    x = torch.randn(M, K, device="cuda").bfloat16(); w = torch.randn(N, K, device="cuda").bfloat16()
    for _ in range(3): y = torch.nn.functional.linear(x, w)
    torch.cuda.synchronize()
