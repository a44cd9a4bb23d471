#!/usr/bin/env python3
"""Prints the memory / wait skeleton of one kernel's main (outermost, nested) loop from a hipcc -S listing:
    python tools/isa_skeleton.py listing.s k_reg_radixIdLi8ELi8ELi8E
Lines kept: barriers, s_waitcnt vmcnt, scratch traffic, loop headers; runs of global/buffer loads and stores are counted.
A `vmcnt(0)` inside the loop that is not the collection point of the prefetch is a stall on the previous tile's stores."""
import re, sys
lst, key = sys.argv[1], sys.argv[2]
lines = open(lst).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*" + re.escape(key) + r"\w*:", l))
end = next(i for i in range(start, len(lines)) if ".amdhsa_kernel" in lines[i])
body = lines[start:end]
# main loop: first depth-1 header that has child loops
main = next((i for i, l in enumerate(body) if "Loop Header: Depth=1" in l and i + 1 < len(body) and "Child Loop" in body[i + 1]), 0)
nl = ns = 0
def flush():
    global nl, ns
    if nl or ns:
        print(f"        ... {nl} loads, {ns} stores")
    nl = ns = 0
for i in range(main, len(body)):
    l = body[i]
    if re.search(r"(global|buffer)_load", l): nl += 1; continue
    if re.search(r"(global|buffer)_store", l): ns += 1; continue
    if re.search(r"s_barrier|s_waitcnt vmcnt|scratch_|Loop Header", l):
        flush()
        print(f"{i:5d}: {l.strip()}")
flush()
for l in lines[end:end + 60]:
    if re.search(r"NumVgprs|ScratchSize|Occupancy", l): print(l.strip())
