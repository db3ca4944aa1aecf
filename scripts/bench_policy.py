"""Timing of the PO4AO-shaped policy network of bench.py's C3 line (stock PyTorch-ROCm convolutions) under the settings a trainer
could switch on: MIOpen auto-tuning (cudnn.benchmark), channels_last.  python scripts/bench_policy.py [n_envs] [n_act]"""
import sys, time
import torch
import torch.nn as nn
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
A = int(sys.argv[2]) if len(sys.argv) > 2 else 41
dev = torch.device("cuda", 0)


def make():
    torch.manual_seed(5)
    net = nn.Sequential(nn.Conv2d(39, 64, 3, padding=1), nn.LeakyReLU(), nn.Conv2d(64, 64, 3, padding=1), nn.LeakyReLU(), nn.Conv2d(64, 1, 3, padding=1))
    return net.to(dev).eval()


def timeit(net, x, n=30):
    with torch.no_grad():
        for _ in range(5):
            net(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            net(x)
        torch.cuda.synchronize()
    return 1e3 * (time.perf_counter() - t0) / n


x = torch.randn(N, 39, A, A, device=dev)
print("default              %.3f ms" % timeit(make(), x))
torch.backends.cudnn.benchmark = True
print("cudnn.benchmark      %.3f ms" % timeit(make(), x))
net = make().to(memory_format=torch.channels_last)
xc = x.contiguous(memory_format=torch.channels_last)
print("benchmark + NHWC     %.3f ms" % timeit(net, xc))
with torch.autocast("cuda", dtype=torch.bfloat16):
    print("benchmark + bf16     %.3f ms (information only: the reference policy runs in float32)" % timeit(make(), x))
