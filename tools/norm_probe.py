import torch, time
dev = torch.device("cuda", 0)
for shape in [(12, 1056, 5632), (12, 1056, 3168), (12, 1056, 2816), (12, 1056, 1056)]:
    X = torch.randn(*shape, device=dev).bfloat16()
    def t(f, n=20):
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): r = f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3, r
    a, ra = t(lambda: X.norm(dim=(-2, -1), keepdim=True))
    b, rb = t(lambda: torch.linalg.vector_norm(X.view(shape[0], -1), dim=1))
    c, rc = t(lambda: torch.linalg.vector_norm(torch.linalg.vector_norm(X.view(shape[0], 1024, -1), dim=2, dtype=torch.float32), dim=1))
    d, rd = t(lambda: torch.linalg.vector_norm(X.view(shape[0], -1), dim=1, dtype=torch.float32))
    print(shape, f"norm(dim=(-2,-1)) {a:.0f} us | vector_norm flat {b:.0f} | two-stage fp32 {c:.0f} | flat fp32 {d:.0f}",
          float((ra.flatten().float() - rc.flatten()).abs().max() / rc.abs().max()))
