"""Where the visible GPU hangs (NUMA node) and what the sequential evaluation costs from each node's cores: the zero-copy
evaluation reads X from, and writes its results and completion word to, pinned host memory — allocated on the node of the
core that first touches it."""
import glob, os, subprocess, sys
def read(p):
    try: return open(p).read().strip()
    except Exception as e: return "?(%s)" % e
nodes = sorted(glob.glob("/sys/devices/system/node/node[0-9]*"))
print("nodes:", [(os.path.basename(n), read(n + "/cpulist")) for n in nodes])
for c in sorted(glob.glob("/sys/class/drm/card[0-9]*")):
    if os.path.exists(c + "/device/numa_node"):
        print(c, "numa_node", read(c + "/device/numa_node"), "vendor", read(c + "/device/vendor"))
print("this process may run on", len(os.sched_getaffinity(0)), "cpus")
if len(sys.argv) > 1 and sys.argv[1] == "run":
    for n in nodes:
        cl = read(n + "/cpulist")
        cpus = set()
        for part in cl.split(","):
            if "-" in part:
                a, b = part.split("-"); cpus |= set(range(int(a), int(b) + 1))
            elif part.strip().isdigit(): cpus.add(int(part))
        cpus &= os.sched_getaffinity(0)
        if not cpus: continue
        code = ("import os, sys, runpy; os.sched_setaffinity(0, %r); "
                "sys.argv = ['bench.py', '--steps', '200', '--warmup', '20', '--no-stage-timing']; "
                "runpy.run_path('bench.py', run_name='__main__')" % (sorted(cpus),))
        for rep in range(2):
            r = subprocess.run([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
            print(os.path.basename(n), r.stdout.decode().strip()[-120:] or r.stderr.decode().strip()[-300:])
