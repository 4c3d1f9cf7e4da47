"""Exposed memory round trips in the compiled kernels: every `s_waitcnt vmcnt(0)` (or lgkmcnt(0) behind an s_load) that sits
between two vector-memory loads is a full memory latency that nothing overlaps.  Per kernel: loads, stores, vmcnt(0) waits, and
the longest run of load -> vmcnt(0) -> load -> vmcnt(0) ... without anything else in flight.
   hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o k.s gprf_amd/csrc/gprf_<stage>.hip   (potrf | solve | mgrad | big | fill | tables)
   python scripts/isa_serial_loads.py k.s [name filter]"""
import re, sys

def kernels(text):
    cur, name = None, None
    for line in text.split("\n"):
        m = re.match(r"^(_Z\w+):\s", line)
        if m:
            name, cur = m.group(1), []
        elif cur is not None:
            cur.append(line)
            if "s_endpgm" in line:
                yield name, cur
                cur = None

def main(path, flt=""):
    for name, lines in kernels(open(path).read()):
        if flt not in name: continue
        loads = stores = w0 = 0
        chain = best = 0        # consecutive (load..., vmcnt(0)) groups
        pending = False
        for l in lines:
            s = l.strip()
            if s.startswith(("global_load", "buffer_load", "flat_load")) and "lds" not in s.split()[0]:
                loads += 1; pending = True
            elif s.startswith(("global_store", "buffer_store", "flat_store")):
                stores += 1
            elif s.startswith("s_waitcnt") and "vmcnt(0)" in s:
                w0 += 1
                if pending: chain += 1; best = max(best, chain); pending = False
            elif s.startswith(("v_mfma", "s_barrier", "s_cbranch_scc", "s_cbranch_vcc")) or re.match(r"^\.LBB", s):
                if s.startswith("v_mfma") or s.startswith("s_barrier"): chain = 0
        print("%-64s loads %4d stores %4d vmcnt(0) %4d  longest load->wait(0) run %3d" % (name[:64], loads, stores, w0, best))

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
