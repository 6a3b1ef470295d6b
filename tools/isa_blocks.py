"""Basic-block census of one kernel in a -save-temps ISA dump: per block the instruction mix (v_mad_u64_u32, other VALU, DS,
global/scratch, s_waitcnt, s_nop).  Usage: python tools/isa_blocks.py build/mpe_pair2048-...gfx950.s 'Cfg<2048,29,18,4>' [min_insts]"""
import re
import sys


def main():
    path, want = sys.argv[1], sys.argv[2]
    mn = int(sys.argv[3]) if len(sys.argv) > 3 else 60
    m = re.match(r"Cfg<(\d+),(\d+),(\d+),(\d+)>", want)
    tag = "CfgILi%sELi%sELi%sELi%sE" % m.groups() if m else want
    lines = open(path).read().split("\n")
    start = next(i for i, l in enumerate(lines) if l.startswith("_ZN3mpe") and tag in l and "pair_modexp_kernel" in l and ":" in l and not l.startswith("\t"))
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    blocks, cur = [], ["entry", []]
    blocks.append(cur)
    for l in lines[start + 1:end]:
        if re.match(r"^\.LBB\d+_\d+:", l):
            cur = [l.split(":")[0], []]
            blocks.append(cur)
        elif l.startswith("\t") and not l.strip().startswith((".", ";")):
            cur[1].append(l.strip())
    tot = dict(n=0, mad=0, valu=0)
    for name, ins in blocks:
        c = lambda *p: sum(1 for x in ins if x.startswith(p))
        if len(ins) >= mn:
            print(f"{name:12s} insts {len(ins):5d}  mad {c('v_mad_u64_u32'):5d}  valu {c('v_'):5d}  ds {c('ds_'):4d}  glob {c('global_', 'buffer_'):4d}  "
                  f"scratch {c('scratch_'):3d}  waitcnt {c('s_waitcnt'):3d}  nop {c('s_nop'):3d}  salu {c('s_') - c('s_waitcnt') - c('s_nop'):4d}")
    return blocks


if __name__ == "__main__":
    main()
