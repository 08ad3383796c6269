"""Practical f64 matrix ceiling on this box: vendor DGEMM (rocBLAS/hipBLASLt through torch) at a
large square size and at the bank's own shape (129 x 400^3 batched).  Context for roofline.frac:
the nominal 78.6 TFLOP/s assumes the 2.4 GHz boost clock under a full f64 MFMA load."""
import json, time, torch
dev = torch.device("cuda", 0)
def bench(fn, flops, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return flops * n / (time.perf_counter() - t0) / 1e12
out = {}
for N in (2048, 4096, 8192):
    a = torch.randn(N, N, dtype=torch.float64, device=dev); b = torch.randn(N, N, dtype=torch.float64, device=dev)
    out[f"dgemm_{N}"] = bench(lambda: a @ b, 2.0 * N ** 3)
a = torch.randn(129, 400, 400, dtype=torch.float64, device=dev); b = torch.randn(400, 400, dtype=torch.float64, device=dev)
out["bmm_129x400_shared_B"] = bench(lambda: a @ b, 2.0 * 129 * 400 ** 3, 50)
b2 = torch.randn(129, 400, 400, dtype=torch.float64, device=dev)
out["bmm_129x400"] = bench(lambda: torch.bmm(a, b2), 2.0 * 129 * 400 ** 3, 50)
a = torch.randn(129, 416, 416, dtype=torch.float64, device=dev); b2 = torch.randn(129, 416, 416, dtype=torch.float64, device=dev)
out["bmm_129x416"] = bench(lambda: torch.bmm(a, b2), 2.0 * 129 * 416 ** 3, 50)
print(json.dumps({k: round(v, 2) for k, v in out.items()}))
