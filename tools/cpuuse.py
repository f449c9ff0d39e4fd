import os, subprocess, resource, sys, time
c = sys.argv[1]
t = time.time()
subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--concurrency", c, "--steps", "960"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
w = time.time() - t
ru = resource.getrusage(resource.RUSAGE_CHILDREN)
print(f"conc {c}: wall {w:.1f} s, cpu {ru.ru_utime + ru.ru_stime:.1f} s (user {ru.ru_utime:.1f}, sys {ru.ru_stime:.1f}) -> {(ru.ru_utime + ru.ru_stime) / w:.1f} cores average incl. setup")
