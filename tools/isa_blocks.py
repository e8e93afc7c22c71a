#!/usr/bin/env python3
"""Static instruction mix of one kernel, per basic block, from hipcc --save-temps assembly.

  hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -Ika9q_sdr_amd/csrc --save-temps -c ka9q_sdr_amd/csrc/kq_full16k.hip -o /tmp/x.o
  python3 tools/isa_blocks.py kq_full16k-hip-amdgcn-amd-amdhsa-gfx950.s k_filter_full16k [min_instructions] [--ops BLOCK]

Prints, for every basic block of at least `min_instructions`, the number of vector-ALU instructions (packed ones
separately), scalar, LDS and global-memory instructions.  With --ops BLOCK the opcode histogram of that block.
The dynamic count of a path is the sum over the blocks it runs through; compare with SQ_INSTS_VALU / SQ_WAVES.
"""
import collections
import re
import sys


def kernel_lines(path, name):
    out, on = [], False
    for ln in open(path):
        s = ln.strip()
        if not on:
            if re.match(r"^_Z\w*%s\w*:" % re.escape(name), s):
                on = True
            continue
        if s.startswith("s_endpgm"):
            # the last s_endpgm of the function ends it; .Lfunc_end follows
            out.append(s)
            continue
        if s.startswith(".Lfunc_end"):
            break
        out.append(s)
    return out


def main():
    path, name = sys.argv[1], sys.argv[2]
    args = sys.argv[3:]
    ops_of = None
    if "--ops" in args:
        i = args.index("--ops")
        ops_of = args[i + 1]
        del args[i : i + 2]
    min_n = int(args[0]) if args else 25
    blocks = [["entry", collections.Counter()]]
    for s in kernel_lines(path, name):
        m = re.match(r"^(\.LBB\d+_\d+):", s)
        if m:
            blocks.append([m.group(1), collections.Counter()])
            continue
        if not s or s[0] in ";.":
            continue
        op = s.split()[0]
        blocks[-1][1][op] += 1
        if op.startswith(("s_cbranch", "s_branch")):  # the fall-through after a branch is a block of its own
            blocks.append(["%s+%d" % (blocks[-1][0].split("+")[0], len(blocks)), collections.Counter()])
    tot = collections.Counter()
    for name_, c in blocks:
        n = sum(c.values())
        cls = collections.Counter()
        for k, v in c.items():
            if k.startswith("v_pk_"):
                cls["pk"] += v
            if k.startswith("v_"):
                cls["valu"] += v
            elif k.startswith("s_"):
                cls["salu"] += v
            elif k.startswith("ds_"):
                cls["lds"] += v
            elif k.startswith(("global_", "buffer_", "flat_", "scratch_")):
                cls["vmem"] += v
        tot.update(cls)
        if ops_of == name_:
            for k, v in c.most_common():
                print("    %-28s %d" % (k, v))
        if n >= min_n and ops_of is None:
            print("%-12s total %5d  valu %5d (packed %4d)  salu %4d  lds %3d  vmem %3d" %
                  (name_, n, cls["valu"], cls["pk"], cls["salu"], cls["lds"], cls["vmem"]))
    if ops_of is None:
        print("whole kernel (static): valu %d (packed %d) salu %d lds %d vmem %d" %
              (tot["valu"], tot["pk"], tot["salu"], tot["lds"], tot["vmem"]))


if __name__ == "__main__":
    main()
