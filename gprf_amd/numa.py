"""Run next to the GPU.  The sequential evaluation is zero-copy: the kernels read X from, and write the result and the
completion word into, pinned host memory, and the host polls that word.  `pin_to_gpu_node` narrows the calling process's CPU
affinity to the cores of the GPU's NUMA node (what `numactl --cpunodebind` would do from outside); call it before the first
`GPRF(...)` is built.  Measured on the MI355X pool: no difference on most boxes, 0.51-0.61 ms against 0.39 per evaluation
on one ("Run-to-run spread") — cheap insurance, not a cure for every slow run."""
import os


def _node_cpus(node):
    cpus = set()
    for part in open("/sys/devices/system/node/node%d/cpulist" % node).read().strip().split(","):
        a, _, b = part.partition("-")
        if a.strip().isdigit():
            cpus |= set(range(int(a), int(b or a) + 1))
    return cpus


def gpu_numa_node(device=0):
    """NUMA node of HIP device `device` from its PCI address, or -1 when the platform does not say"""
    try:
        import torch
        p = torch.cuda.get_device_properties(device)
        path = "/sys/bus/pci/devices/%04x:%02x:%02x.0/numa_node" % (p.pci_domain_id, p.pci_bus_id, p.pci_device_id)
        return int(open(path).read().strip())
    except Exception:
        return -1


def pin_to_gpu_node(device=0):
    """-> (node, number of cpus now allowed), or (-1, 0) when nothing was changed"""
    if os.environ.get("GPRF_NUMA_PIN", "1") == "0" or not hasattr(os, "sched_setaffinity"):
        return -1, 0
    node = gpu_numa_node(device)
    if node < 0:
        return -1, 0
    try:
        cpus = _node_cpus(node) & os.sched_getaffinity(0)
        if not cpus:
            return -1, 0
        os.sched_setaffinity(0, cpus)
        return node, len(cpus)
    except Exception:
        return -1, 0
