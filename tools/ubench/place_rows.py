import torch, time
dev = torch.device("cuda", 0)
H, W = 4320, 7680
for dtype, name in ((torch.uint8, "u8"), (torch.float64, "f64")):
    frame = torch.empty((H, W, 3), dtype=dtype, device=dev)
    rows = H - H // 8
    staging = torch.randint(0, 200, (rows, W, 3), device=dev).to(dtype)
    idx = torch.randperm(H, device=dev)[:rows].sort().values
    def run(f, s, reps=20):
        for _ in range(3):
            f.index_copy_(0, idx, s)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            f.index_copy_(0, idx, s)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    ms = run(frame, staging)
    nbytes = staging.numel() * staging.element_size()
    print("%s index_copy_ as is: %.3f ms  (%.0f GB/s moved)" % (name, ms, 2 * nbytes / ms / 1e6))
    if dtype == torch.uint8:
        f2 = frame.view(H, W * 3).view(torch.int64)
        s2 = staging.view(rows, W * 3).view(torch.int64)
        ms = run(f2, s2)
        print("%s index_copy_ through an int64 view: %.3f ms  (%.0f GB/s moved)" % (name, ms, 2 * nbytes / ms / 1e6))
        f3 = frame.view(H, W * 3).view(torch.complex128)
        s3 = staging.view(rows, W * 3).view(torch.complex128)
        ms = run(f3, s3)
        print("%s index_copy_ through a 16-byte view: %.3f ms  (%.0f GB/s moved)" % (name, ms, 2 * nbytes / ms / 1e6))
        ref = frame.clone(); ref.zero_(); ref.index_copy_(0, idx, staging)
        chk = torch.zeros_like(frame); chk.view(H, W * 3).view(torch.int64).index_copy_(0, idx, s2)
        print("views agree:", bool(torch.equal(ref, chk)))
