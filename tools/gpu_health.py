"""Is this GPU deterministic?  The same elementwise expression and the same matmul, recomputed: every run must give the same bits.
(round 6: one box of the pool returned different results run to run — and then memory access faults in kernels that had passed the
whole suite for rounds; this check separates a sick device from a bug.)"""
import torch
torch.manual_seed(0)
x = torch.randn(1 << 26, device="cuda")
r = (x * 1.5 + 2).sin()
bad_e = sum(int(((x * 1.5 + 2).sin() != r).any()) for _ in range(20))
n_e = int(((x * 1.5 + 2).sin() != r).sum())
a = torch.randn(4096, 4096, device="cuda"); b = torch.randn(4096, 4096, device="cuda")
_ = a @ b
ref = a @ b
bad_m = sum(int(((a @ b) != ref).any()) for _ in range(20))
y = torch.arange(1 << 26, device="cuda", dtype=torch.int32)
bad_i = sum(int(((y * 3 + 1) != (y * 3 + 1)).any()) for _ in range(20))
print(f"gpu_health: elementwise mismatching runs {bad_e}/20 (last run: {n_e} elements), matmul {bad_m}/20, integer {bad_i}/20", flush=True)
