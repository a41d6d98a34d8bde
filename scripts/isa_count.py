#!/usr/bin/env python3
"""Static instruction mix of the kernels in a device assembly file.

usage: hipcc -O3 -std=c++20 --offload-arch=gfx950 --offload-device-only -S x.hip -o x.s
       scripts/isa_count.py x.s [name-substring ...]
Counts are per kernel body as written (loops are counted once): packed / other VALU, LDS, vector
memory, scalar memory, waits, barriers, plus the register and LDS figures of the descriptor."""
import re
import subprocess
import sys


def demangle(n):
    try:
        d = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip()
        d = d.replace("(anonymous namespace)::", "")
        depth, out = 0, ""
        for ch in d:  # cut the argument list, keep template arguments
            if ch == "(" and depth == 0:
                break
            depth += ch == "<"
            depth -= ch == ">"
            out += ch
        return out.replace("void ", "").replace("rr::", "").replace("(anonymous namespace)::", "")
    except OSError:
        return n


def main():
    path, pats = sys.argv[1], sys.argv[2:]
    txt = open(path).read().splitlines()
    body, cur, meta = {}, None, {}
    for ln in txt:
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            cur = m.group(1)
            body[cur] = []
            continue
        if cur and ln.strip().startswith(".end_amdhsa_kernel"):
            cur = None
        if cur is not None:
            s = ln.strip()
            if s and not s.startswith((".", ";", "//")) and not s.endswith(":"):
                body[cur].append(s.split()[0])
    name = None
    for ln in txt:
        m = re.match(r"\s+\.name:\s+(\S+)", ln)
        if m:
            name = m.group(1)
            meta.setdefault(name, {})
        for key in ("vgpr_count", "sgpr_count", "group_segment_fixed_size", "vgpr_spill_count", "private_segment_fixed_size"):
            m = re.match(r"\s+\.%s:\s+(\d+)" % key, ln)
            if m and name:
                meta[name][key] = int(m.group(1))
    print(f"{'kernel':48s} {'pk':>5s} {'valu':>5s} {'lds':>4s} {'vmem':>4s} {'smem':>4s} {'salu':>5s} {'wait':>4s} {'bar':>3s} {'vgpr':>4s} {'sgpr':>4s} {'lds B':>6s} {'spill':>5s}")
    for k, ins in body.items():
        d = demangle(k)
        if pats and not any(p in d for p in pats):
            continue
        if k not in meta:
            continue
        c = dict(pk=0, valu=0, lds=0, vmem=0, smem=0, salu=0, wait=0, bar=0)
        for i in ins:
            if i.startswith("v_pk_"):
                c["pk"] += 1
            elif i.startswith("v_"):
                c["valu"] += 1
            elif i.startswith("ds_"):
                c["lds"] += 1
            elif i.startswith(("global_", "buffer_", "flat_", "scratch_")):
                c["vmem"] += 1
            elif i.startswith("s_load") or i.startswith("s_buffer_load"):
                c["smem"] += 1
            elif i.startswith("s_waitcnt"):
                c["wait"] += 1
            elif i.startswith("s_barrier"):
                c["bar"] += 1
            elif i.startswith("s_"):
                c["salu"] += 1
        m = meta[k]
        print(f"{d[-48:]:48s} {c['pk']:5d} {c['valu']:5d} {c['lds']:4d} {c['vmem']:4d} {c['smem']:4d} {c['salu']:5d} {c['wait']:4d} {c['bar']:3d} "
              f"{m.get('vgpr_count', 0):4d} {m.get('sgpr_count', 0):4d} {m.get('group_segment_fixed_size', 0):6d} {m.get('vgpr_spill_count', 0):5d}")


if __name__ == "__main__":
    main()
