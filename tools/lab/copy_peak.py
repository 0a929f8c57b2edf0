import importlib, os, sys, torch
sys.path.insert(0, "/root/repo"); 
pkg = importlib.import_module("power-gzip_amd")
eng = pkg.Engine(0)
n = 1 << 30
a = torch.empty(n, dtype=torch.uint8, device=eng.dev); b = torch.empty(n, dtype=torch.uint8, device=eng.dev); a.zero_()
def t(f, name):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): f()
    e1.record(); torch.cuda.synchronize()
    print(name, round(2.0 * n * 5 / (e0.elapsed_time(e1) * 1e-3) / 1e9, 1), "GB/s")
t(lambda: b.copy_(a), "torch memcpy")
t(lambda: eng.copy_device(b, a), "engine copy16")
a32 = a.view(torch.float32); b32 = b.view(torch.float32)
t(lambda: torch.add(a32, 0.0, out=b32), "torch add kernel f32")
