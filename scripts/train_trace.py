"""Kernel sequence of ONE training step from a rocprofv3 kernel trace (scripts/train_trace.py <kernel_trace.csv> [step]):
launches in start order with short names, run-length encoded, plus per-name launch counts -- to see which torch glue kernels
(fills, copies, casts) sit between the native calls.  Steps are cut at the optimizer's multi_tensor_apply launches."""
import csv
import re
import sys
from collections import Counter


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"at::native::", "", name)
    m = re.match(r"([A-Za-z_0-9:]+)(<[^(]{0,60})?", name)
    return (m.group(1) + (m.group(2) or ""))[:70] if m else name[:70]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = [short(r["Kernel_Name"]) for r in rows]
    cuts = [i for i, n in enumerate(names) if "multi_tensor_apply" in n]
    ends = [c for i, c in enumerate(cuts) if i + 1 == len(cuts) or cuts[i + 1] - c > 50]       # last optimizer launch of a step
    want = int(sys.argv[2]) if len(sys.argv) > 2 else len(ends) - 2
    a, b = ends[want] + 1, ends[want + 1] + 1
    seq = names[a:b]
    dur = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[a:b]]
    print("step %d: %d launches, %.2f ms of kernels, %.2f ms wall" % (want, len(seq), sum(dur) / 1e3,
          (int(rows[b - 1]["End_Timestamp"]) - int(rows[a]["Start_Timestamp"])) / 1e6))
    i = 0
    while i < len(seq):
        j = i
        while j < len(seq) and seq[j] == seq[i]:
            j += 1
        print("%5d x%-3d %8.1f us  %s" % (i, j - i, sum(dur[i:j]), seq[i]))
        i = j
    # where the GPU waits for the host: gaps between the end of one kernel and the start of the next (single stream)
    st = [int(r["Start_Timestamp"]) for r in rows[a:b]]
    en = [int(r["End_Timestamp"]) for r in rows[a:b]]
    gaps, hi = [], en[0]
    for j in range(1, len(st)):
        if st[j] > hi:
            gaps.append(((st[j] - hi) / 1e3, j))
        hi = max(hi, en[j])
    print("---- idle: %.2f ms in %d gaps; > 20 us: %.2f ms in %d; > 100 us: %.2f ms in %d" % (
        sum(g for g, _ in gaps) / 1e3, len(gaps), sum(g for g, _ in gaps if g > 20) / 1e3, sum(1 for g, _ in gaps if g > 20),
        sum(g for g, _ in gaps if g > 100) / 1e3, sum(1 for g, _ in gaps if g > 100)))
    # idle by 100-launch window (which part of the step is host-bound)
    for w0 in range(0, len(seq), 100):
        gw = sum(g for g, j in gaps if w0 <= j < w0 + 100)
        kw = sum(dur[w0:w0 + 100])
        print("  launches %4d-%4d: kernels %7.1f us, idle %7.1f us   first: %s" % (w0, min(w0 + 100, len(seq)) - 1, kw, gw, seq[w0]))
    for g, j in sorted(gaps, reverse=True)[:25]:
        print("  gap %7.1f us before launch %4d %s (after %s)" % (g, j, seq[j], seq[j - 1]))
    print("---- counts")
    cnt, tot = Counter(seq), Counter()
    for n, d in zip(seq, dur):
        tot[n] += d
    for n, c in cnt.most_common():
        print("%5d %9.1f us  %s" % (c, tot[n], n))


if __name__ == "__main__":
    main()
