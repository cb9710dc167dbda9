import sys, math, torch
sys.path.insert(0, "candle-video_amd"); sys.path.insert(0, "tools")
import ltxhip
from microbench import timeit
for M, N, K in [(1, 2048, 256), (1, 2048, 2048), (1, 12288, 2048), (128, 2048, 4096), (128, 2048, 2048), (128, 4096, 2048), (4992, 2048, 128), (4992, 128, 2048), (1, 4096, 256), (1, 512, 512)]:
    x = torch.randn(M, K, device="cuda").bfloat16(); w = (torch.randn(N, K, device="cuda") / math.sqrt(K)).bfloat16(); b = torch.randn(N, device="cuda").bfloat16()
    fn = lambda: ltxhip.ops.linear(x, w, b)
    t = min(timeit(fn, iters=20, warm=3) for _ in range(3))
    byt = (M * K + N * K + M * N) * 2
    print(M, N, K, "us", round(t * 1000, 1), "GB/s", round(byt / t / 1e6, 0), "plan", ltxhip.ops.gemm_plan(M, N, K), flush=True)
