import time, torch
for m in (224, 152):
    A = torch.randn(m, m, dtype=torch.float64); A = A @ A.T
    for th in (1, 2, 4, 8, 16):
        torch.set_num_threads(th)
        torch.linalg.eigh(A); t = time.time()
        for _ in range(20): torch.linalg.eigh(A)
        print(m, "threads", th, "eigh %.2f ms" % ((time.time() - t) / 20 * 1e3))
