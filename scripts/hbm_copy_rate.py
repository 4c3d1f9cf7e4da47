"""What this box sustains for plain device-to-device copies of the sizes this path moves per kernel
(read + write counted): python scripts/hbm_copy_rate.py"""
import torch
for mb in (20, 75, 150, 500, 2000):
    n = mb * 1024 * 1024 // 8
    a = torch.empty(n, dtype=torch.float64, device="cuda"); b = torch.ones(n, dtype=torch.float64, device="cuda")
    for _ in range(5): a.copy_(b)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 50
    e0.record()
    for _ in range(reps): a.copy_(b)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("copy %5d MB -> %5d MB moved: %.1f us, %.2f TB/s" % (mb, 2 * mb, ms * 1e3, 2 * mb * 1.048576e6 / (ms * 1e-3) / 1e12))
