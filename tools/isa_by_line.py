"""Static instruction counts of one kernel by source line, from hipcc's assembly with line tables:

    hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -std=c++17 --cuda-device-only \
        -gline-tables-only -S raytracing-one-weekend_amd/csrc/wrt.hip -o /tmp/wrt_g.s
    python tools/isa_by_line.py /tmp/wrt_g.s _Z8k_renderILb0ELb1ELb1EEv12RenderLaunch [first-last ...]

Every instruction is booked to the source line of the innermost .loc in force (inlined code keeps the
line of the function it came from).  Without ranges: totals per file:line, largest first.  With ranges
(`hpp:838-921`, `hip:195-252`): VALU / SALU / SMEM / VMEM / LDS / branch counts per range, so that the
static size of a phase (all branches together) can be compared between builds.
"""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "scratch_", "buffer_")):
        return "vmem"
    return "other"


def main():
    path, kernel = sys.argv[1], sys.argv[2]
    ranges = []
    for a in sys.argv[3:]:
        f, r = a.split(":")
        lo, hi = r.split("-")
        ranges.append((a, f, int(lo), int(hi)))
    files = {}
    inside = False
    where = ("?", 0)
    per_line = collections.defaultdict(collections.Counter)
    ops_by_line = collections.defaultdict(collections.Counter)
    for line in open(path):
        m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]*)"', line)
        if m:
            files[int(m.group(1))] = m.group(2)
            continue
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"', line)
        if m and int(m.group(1)) not in files:
            files[int(m.group(1))] = m.group(2)
            continue
        if line.startswith(kernel + ":"):
            inside = True
            continue
        if inside and line.startswith(".Lfunc_end"):
            break
        if not inside:
            continue
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", line)
        if m:
            where = (files.get(int(m.group(1)), m.group(1)), int(m.group(2)))
            continue
        m = re.match(r"\s+([a-z][a-z0-9_]+)\b", line)
        if not m or line.lstrip().startswith((".", ";")):
            continue
        op = m.group(1)
        kind = classify(op)
        per_line[where][kind] += 1
        ops_by_line[where][op] += 1
    if not ranges:
        total = collections.Counter()
        for c in per_line.values():
            total.update(c)
        print("total", dict(total))
        rows = sorted(per_line.items(), key=lambda kv: -kv[1]["valu"])
        for (f, l), c in rows[:80]:
            print(f"{f.split('/')[-1]:18s}:{l:5d}  valu {c['valu']:5d} salu {c['salu']:4d} smem {c['smem']:3d} "
                  f"vmem {c['vmem']:3d} lds {c['lds']:3d} br {c['branch']:3d}")
        return
    for name, f, lo, hi in ranges:
        tot = collections.Counter()
        ops = collections.Counter()
        for (ff, l), c in per_line.items():
            if f in ff and lo <= l <= hi:
                tot.update(c)
                ops.update(ops_by_line[(ff, l)])
        top = ", ".join(f"{o} {n}" for o, n in ops.most_common(12))
        print(f"{name:20s} valu {tot['valu']:5d} salu {tot['salu']:4d} smem {tot['smem']:3d} vmem {tot['vmem']:3d} "
              f"lds {tot['lds']:3d} br {tot['branch']:3d} | {top}")


if __name__ == "__main__":
    main()
