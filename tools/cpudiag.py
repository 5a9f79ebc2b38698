import os, time, torch
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), flush=True)
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us"):
    try: print(f, open(f).read().strip(), flush=True)
    except Exception as e: print(f, "n/a", flush=True)
print("loadavg", open("/proc/loadavg").read().strip(), flush=True)
a = torch.randn(4096, 4096)
for n in (8, 16, 32, len(os.sched_getaffinity(0))):
    torch.set_num_threads(n)
    t0 = time.perf_counter(); (a @ a).sum().item(); t1 = time.perf_counter()
    t0 = time.perf_counter(); (a @ a).sum().item(); t1 = time.perf_counter()
    print("threads", n, "4096^3 matmul s", round(t1 - t0, 3), flush=True)
