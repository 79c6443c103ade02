#!/usr/bin/env python3
"""One call out of a launch-by-launch timeline written by tools/trace_lastcall.py when its window holds several calls of the
single-proof probe (with the TranscriptRng chain started ahead of prove() the device is never idle between two calls, so the idle-gap
rule of trace_lastcall.py no longer separates them):  python tools/timeline_call.py <timeline.txt> [k]  prints the k-th call
(default 2: the first one of a process has no n to start its chain with), from its first one-commitment kernel to its K_assemble,
times from the call's start, runs of one kernel on one queue collapsed into one line (count, summed duration)."""
import re
import sys


def main():
    path, k = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 2
    L = open(path).read().split("\n")
    ends = [i for i, l in enumerate(L) if " ms " in l and "K_assemble" in l]
    if len(ends) < k:
        sys.exit("%s holds %d call(s)" % (path, len(ends)))
    a = (ends[k - 2] + 1) if k > 1 else 0
    while a < ends[k - 1] and "k_commit_wave" not in L[a] and k > 1:
        a += 1
    row = re.compile(r"\s*([\d.]+) ms\s+(q\d+)\s+(.*?)\s+([\d.]+) us\s+idle")
    t0, prev, cnt, first, tot, out = None, None, 0, 0.0, 0.0, []
    for l in L[a:ends[k - 1] + 1]:
        m = row.match(l)
        if not m:
            continue
        if t0 is None:
            t0 = float(m.group(1))
        key = (m.group(2), m.group(3).strip())
        if key != prev:
            if prev:
                out.append("%10.3f ms  %-4s %-52s x%-4d %10.1f us" % (first, prev[0], prev[1][:52], cnt, tot))
            prev, cnt, first, tot = key, 0, float(m.group(1)) - t0, 0.0
        cnt += 1
        tot += float(m.group(4))
    if prev:
        out.append("%10.3f ms  %-4s %-52s x%-4d %10.1f us" % (first, prev[0], prev[1][:52], cnt, tot))
    print("# call %d of %s: start of the call's first launch | queue | kernel | launches in a row | their summed duration" % (k, path))
    print("\n".join(out))


if __name__ == "__main__":
    main()
