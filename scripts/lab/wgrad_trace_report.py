#!/usr/bin/env python3
"""Per-workgroup residency of the NHWC weight-gradient kernel from its lab trace (OMNIHD_WGRAD_NHWC_TRACE=file): for the last launch
in the file — start / end spread, duration by XCC, how many workgroups run at a time."""
import sys, collections
lines = open(sys.argv[1]).read().splitlines()
starts = [i for i, l in enumerate(lines) if l.startswith("launch")]
i0 = starts[-1]
print(lines[i0])
rows = [tuple(int(v) for v in l.split()) for l in lines[i0 + 1:] if l and not l.startswith("launch")]
t0 = min(r[1] for r in rows)
dur = [(r[2] - r[1]) / 100.0 for r in rows]         # us (100 MHz ticks)
beg = [(r[1] - t0) / 100.0 for r in rows]
end = [(r[2] - t0) / 100.0 for r in rows]
print(f"workgroups {len(rows)}; kernel span {max(end):.1f} us; start: min {min(beg):.1f} max {max(beg):.1f} us; duration: min {min(dur):.1f} mean {sum(dur)/len(dur):.1f} max {max(dur):.1f} us")
by = collections.defaultdict(list)
for r, d, b, e in zip(rows, dur, beg, end):
    by[r[4] & 0xf].append((d, b, e))
for x in sorted(by):
    v = by[x]
    print(f"  XCC {x}: {len(v):3d} workgroups, start {min(b for _, b, _ in v):7.1f}..{max(b for _, b, _ in v):7.1f} us, duration mean {sum(d for d, _, _ in v)/len(v):7.1f} max {max(d for d, _, _ in v):7.1f}, last end {max(e for _, _, e in v):7.1f}")
cus = collections.Counter(((r[4] & 0xf), (r[3] >> 13) & 7, (r[3] >> 12) & 1, (r[3] >> 8) & 0xf) for r in rows)
print(f"distinct (xcc, se, sh, cu): {len(cus)}; workgroups per CU: {collections.Counter(cus.values())}")
late = sorted(beg)[-8:]
print("latest starts (us):", " ".join(f"{v:.1f}" for v in late))
