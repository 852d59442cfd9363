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

# the same CONVOLUTIONS through the vendor stack (torch.nn.functional.conv2d -> MIOpen), channels-last, for reference
import torch.nn.functional as F


def conv_ref(name, N, H, Cin, Cout, k, stride, pad, dil, dtype, reps=10):
    x = torch.randn(N, Cin, H, H, device=dev, dtype=dtype).contiguous(memory_format=torch.channels_last)
    w = (torch.randn(Cout, Cin, k, k, device=dev, dtype=dtype) * 0.02).contiguous(memory_format=torch.channels_last)
    try:
        for _ in range(3): y = F.conv2d(x, w, None, stride, pad, dil)
        torch.cuda.synchronize()
    except Exception as e:
        print('conv2d %s: failed (%s)' % (name, str(e)[:80])); return
    best = 1e9
    for rep in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): F.conv2d(x, w, None, stride, pad, dil)
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps * 1e3)
    oh = y.shape[2]
    print('conv2d (MIOpen) %-44s %8.1f us  %7.1f TFLOP/s' % (name, best, 2.0 * N * oh * oh * Cout * k * k * Cin / best / 1e6), flush=True)


bf = torch.bfloat16
conv_ref('bf16 b64 layer3 shortcut 3x3 512->1024', 64, 31, 512, 1024, 3, 1, 1, 1, bf)
conv_ref('bf16 b64 layer3 conv2 3x3 d2 256->256', 64, 31, 256, 256, 3, 1, 2, 2, bf)
conv_ref('bf16 b64 layer2 shortcut 3x3/s2 256->512', 64, 63, 256, 512, 3, 2, 0, 1, bf)
conv_ref('bf16 b64 layer3 conv3 1x1 256->1024', 64, 31, 256, 1024, 1, 1, 0, 1, bf)
conv_ref('bf16 b64 layer3 conv1 1x1 1024->256', 64, 31, 1024, 256, 1, 1, 0, 1, bf)
f32 = torch.float32
conv_ref('fp32 b1 layer3 shortcut 3x3 512->1024', 1, 31, 512, 1024, 3, 1, 1, 1, f32, reps=30)
conv_ref('fp32 b1 layer3 conv2 3x3 d2 256->256', 1, 31, 256, 256, 3, 1, 2, 2, f32, reps=30)
conv_ref('fp32 Conf_Fusion 3x3 256->512 on 7 x 25x25', 7, 25, 256, 512, 3, 1, 1, 1, f32, reps=30)
conv_ref('fp32 tower 3x3 256->256 on 25x25', 1, 25, 256, 256, 3, 1, 1, 1, f32, reps=30)
