import multiprocessing as mp, time, os
def burn(_):
    t0 = time.process_time(); x = 0
    while time.process_time() - t0 < 1.0: x += 1
    return x
if __name__ == "__main__":
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/cpuset.cpus.effective"):
        try: print(f, open(f).read().strip())
        except Exception as e: print(f, "n/a")
    print("sched_getaffinity", len(os.sched_getaffinity(0)))
    for n in (8, 16, 32, 64, 128):
        t0 = time.time()
        with mp.Pool(n) as p: p.map(burn, range(n))
        dt = time.time() - t0
        print("procs %d: 1 cpu-s each took %.2fs wall -> effective parallelism %.1f" % (n, dt, n / dt))
