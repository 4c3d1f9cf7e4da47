"""Per-kernel register / LDS / scratch use from the compiler's metadata:
   hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o k.s gprf_amd/csrc/gprf_<stage>.hip   (potrf | solve | mgrad | big | fill | tables)
   python scripts/isa_resources.py k.s [name filter]"""
import re, sys

def main(path, flt=""):
    t = open(path).read()
    meta = t[t.index("amdhsa.kernels:"):]
    for blk in meta.split("  - .agpr_count:")[1:]:
        g = lambda k: re.search(r"\.%s:\s+(\S+)" % k, blk).group(1)
        name = g("name")
        if flt in name:
            print("%-70s vgpr %3s agpr %3s sgpr %3s lds %6s scratch %4s" % (
                name[:70], g("vgpr_count"), blk.split()[0], g("sgpr_count"), g("group_segment_fixed_size"),
                g("private_segment_fixed_size")))

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
