"""Instruction histogram of one kernel's ISA, split at its s_barrier's:
   hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o k.s gprf_amd/csrc/gprf_<stage>.hip   (potrf | solve | mgrad | big | fill | tables)
   python scripts/isa_phases.py k.s k_potrf_reg"""
import sys

def main(path, name):
    lines = open(path).read().split('\n')
    st = [i for i, l in enumerate(lines) if name in l and l.startswith('_Z') and '; @' in l][0]
    en = [i for i, l in enumerate(lines) if i > st and '.Lfunc_end' in l][0]
    body = lines[st:en]
    seg, cur = [], []
    for l in body:
        cur.append(l)
        if 's_barrier' in l:
            seg.append(cur); cur = []
    seg.append(cur)
    def cnt(sg, *keys): return sum(any(k in l for k in keys) for l in sg)
    print("lines", len(body))
    for i, sg in enumerate(seg):
        print(i, len(sg), "scratch", cnt(sg, 'scratch_'), "mfma", cnt(sg, 'v_mfma'), "dpp", cnt(sg, 'dpp'),
              "accrd", cnt(sg, 'accvgpr_read'), "accwr", cnt(sg, 'accvgpr_write'), "wl", cnt(sg, 'v_writelane'),
              "rl", cnt(sg, 'v_readlane'), "dsr", cnt(sg, 'ds_read'), "dsw", cnt(sg, 'ds_write'),
              "gld", cnt(sg, 'global_load'), "gst", cnt(sg, 'global_store'), "wait", cnt(sg, 's_waitcnt'),
              "nop", cnt(sg, 's_nop'), "f64", cnt(sg, '_f64'), "br", cnt(sg, 's_cbranch'))

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
