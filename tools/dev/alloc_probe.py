import time, torch
torch.cuda.init(); torch.cuda.synchronize()
for gb in (1, 8, 8, 32, 32):
    t = time.perf_counter(); x = torch.empty(gb << 30, dtype=torch.uint8, device="cuda"); torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"[alloc] {gb} GB: {dt * 1e3:.1f} ms = {dt * 1e3 / gb:.1f} ms/GB", flush=True)
    del x; torch.cuda.empty_cache()
