import torch, time
x = torch.empty(1<<31, dtype=torch.float32, device="cuda")  # 8 GiB
x.fill_(1.0)
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for name, fn in (("sum (read only)", lambda: x.sum()), ("max (read only)", lambda: x.max()), ("fill (write only)", lambda: x.fill_(2.0)), ("copy (r+w)", None)):
    if fn is None:
        y = torch.empty_like(x); fn = lambda: y.copy_(x); bytes_ = 2 * x.numel() * 4
    else:
        bytes_ = x.numel() * 4
    fn(); torch.cuda.synchronize()
    ev0.record()
    for _ in range(5): fn()
    ev1.record(); torch.cuda.synchronize()
    print(name, round(5 * bytes_ / (ev0.elapsed_time(ev1) * 1e-3) / 1e9, 1), "GB/s")
