"""Load / wait / store sequence of every kernel in an ISA listing (hipcc -S --cuda-device-only ...): finds global loads that the compiler
serialised (a load directly followed by `s_waitcnt vmcnt(0)`, typically one per conditional load: each `if (ok) v = load` of an unrolled
batch becomes its own basic block) and loads issued behind stores (vmcnt is ONE in-order counter: such a load's wait also waits for the
store).  usage: python tools/isa_scan.py file.s [kernel-name-substring]   ->  per kernel: loads, waits by count, the L/W/S/b sequence."""
import re
import sys
from collections import Counter


def kernels(text):
    cur, name = [], None
    for line in text.split("\n"):
        m = re.match(r"^(_Z\w+):\s", line)
        if m:
            if name:
                yield name, cur
            name, cur = m.group(1), []
        elif name:
            if line.startswith("\t.end_amdhsa_kernel") or line.startswith(".Lfunc_end"):
                yield name, cur
                name, cur = None, []
            else:
                cur.append(line.strip())
    if name:
        yield name, cur


def main():
    text = open(sys.argv[1]).read()
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    for name, lines in kernels(text):
        if want not in name:
            continue
        seq = []
        for l in lines:
            if l.startswith(("global_load", "buffer_load", "flat_load")):
                seq.append("Ld" if "lds" in l else "L")
            elif l.startswith(("global_store", "buffer_store", "flat_store")):
                seq.append("S")
            elif l.startswith("s_waitcnt") and "vmcnt" in l:
                seq.append("W" + re.search(r"vmcnt\((\d+)\)", l).group(1))
            elif l.startswith("s_cbranch"):
                seq.append("b")
            elif l.startswith("s_barrier"):
                seq.append("|")
        loads = sum(1 for s in seq if s[0] == "L")
        serial = sum(1 for a, b in zip(seq, seq[1:]) if a[0] == "L" and b == "W0")
        behind = sum(1 for a, b in zip(seq, seq[1:]) if a == "S" and b[0] == "L")
        waits = Counter(s for s in seq if s[0] == "W")
        print(f"{name}: loads {loads}, load->vmcnt(0) pairs {serial}, store->load pairs {behind}, waits {dict(waits)}")
        if want:
            print("   " + " ".join(seq))


if __name__ == "__main__":
    main()
