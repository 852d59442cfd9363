"""What the vendor GEMM (torch.matmul -> hipBLASLt / rocBLAS, hand-scheduled assembly kernels) reaches on THIS box for the GEMM
shapes of the batch-64 backbone's K >= 2304 convolutions, on the same kind of random bf16 data — a measured ceiling for
csrc/conv_bf16.hip's implicit-GEMM tile (MEASUREMENT ONLY: nothing on the product path calls a library GEMM).
    M x N x K: 61504 x 1024 x 4608 (layer3's shortcut conv), 61504 x 256 x 2304 (its conv2), 61504 x 512 x 2304 (layer2's)"""
import torch
dev = 'cuda:0'
for M, N, K in ((61504, 1024, 4608), (61504, 256, 2304), (61504, 512, 2304), (8192, 8192, 8192)):
    a = torch.randn(M, K, device=dev).bfloat16()
    for layout in ('NT', 'NN'):
        w = (torch.randn(N, K, device=dev) * 0.02).bfloat16()
        b = w.t() if layout == 'NT' else w.t().contiguous()
        for _ in range(3): torch.matmul(a, b)
        torch.cuda.synchronize()
        best = 1e9
        for rep in range(3):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): torch.matmul(a, b)
            e1.record(); torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 10 * 1e3)
        print('matmul bf16 %6d x %5d x %5d %s: %8.1f us  %7.1f TFLOP/s' % (M, N, K, layout, best, 2.0 * M * N * K / best / 1e6), flush=True)
