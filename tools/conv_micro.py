"""MIOpen conv2d vs GEMM formulations for the ResNet101 shapes at 928x1600 (6 images)."""
import torch, torch.nn.functional as F

def t(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it

N = 6
for name, cin, cout, h, w, k in [("l1 1x1 256->64", 256, 64, 232, 400, 1), ("l1 3x3 64", 64, 64, 232, 400, 3),
                                 ("l1 1x1 64->256", 64, 256, 232, 400, 1), ("l2 1x1 512->128", 512, 128, 116, 200, 1),
                                 ("l2 3x3 128", 128, 128, 116, 200, 3), ("l2 1x1 128->512", 128, 512, 116, 200, 1),
                                 ("l3 1x1 1024->256", 1024, 256, 58, 100, 1), ("l3 3x3 256", 256, 256, 58, 100, 3),
                                 ("l3 1x1 256->1024", 256, 1024, 58, 100, 1), ("l4 1x1 2048->512", 2048, 512, 29, 50, 1),
                                 ("l4 3x3 512", 512, 512, 29, 50, 3), ("stem 7x7", 3, 64, 928, 1600, 7)]:
    x = torch.randn(N, cin, h, w, device="cuda")
    wgt = torch.randn(cout, cin, k, k, device="cuda")
    stride = 2 if k == 7 else 1
    ms = t(lambda: F.conv2d(x, wgt, padding=k // 2, stride=stride))
    fl = 2 * N * (h // stride) * (w // stride) * cin * cout * k * k
    line = f"{name:20s} conv2d {ms:7.3f} ms {fl/ms/1e9:7.1f} TF/s"
    xl = x.contiguous(memory_format=torch.channels_last); wl = wgt.contiguous(memory_format=torch.channels_last)
    ms2 = t(lambda: F.conv2d(xl, wl, padding=k // 2, stride=stride))
    line += f" | NHWC conv2d {ms2:7.3f} ms {fl/ms2/1e9:6.1f} TF/s"
    if k == 1:
        xm = xl.permute(0, 2, 3, 1)            # [N,H,W,C] view
        w2 = wgt.view(cout, cin)
        ms3 = t(lambda: F.linear(xm, w2))
        line += f" | NHWC linear {ms3:7.3f} ms {fl/ms3/1e9:6.1f} TF/s"
        xc = x.view(N, cin, h * w)
        ms4 = t(lambda: torch.matmul(w2, xc))   # NCHW: W[Cout,Cin] @ X[N,Cin,HW]
        line += f" | NCHW bmm {ms4:7.3f} ms {fl/ms4/1e9:6.1f} TF/s"
    else:
        pass
    print(line, flush=True)
