"""Join scripts/micro/fetch_calib's known byte counts with the FETCH_SIZE / WRITE_SIZE passes of scripts/fetch_calib.sh."""
import collections
import csv
import glob
import sys

rb = sys.argv[1]
want = collections.OrderedDict()
for line in open("/tmp/fc_FETCH_SIZE.log"):
    if line.startswith("BYTES"):
        _, k, v = line.split()
        want[k] = int(v)


def key_of(name):
    for k in want:
        base = k.split("<")[0]
        if not name.startswith(base) and ("void " + base) not in name and (" " + base) not in name and base not in name:
            continue
        if "<" in k:
            if ("<" + k.split("<")[1]) in name.replace(" ", ""):
                return k
        elif base in name:
            return k
    return None


def load(c, counter=None, scale=1024.0):
    agg = collections.defaultdict(list)
    fs = glob.glob("/tmp/fc_%s/**/fc_counter_collection.csv" % c, recursive=True)
    if not fs:
        return {}
    for r in csv.DictReader(open(fs[0])):
        k = key_of(r["Kernel_Name"])
        if k and (counter is None or r["Counter_Name"] == counter):
            agg[k].append(float(r["Counter_Value"]) * scale)
    return {k: sum(v) / len(v) for k, v in agg.items()}


f, w = load("FETCH_SIZE"), load("WRITE_SIZE")
print("row_bytes %s: bytes touched once (slab 3 GiB >> Infinity Cache) against the counters (KB x 1024, as reported -- NO doubling)" % rb)
print("%-14s %14s %14s %8s %14s %8s" % ("kernel", "bytes", "FETCH_SIZE", "ratio", "WRITE_SIZE", "ratio"))
for k in want:
    print("%-14s %14d %14.0f %8.3f %14.0f %8.3f" % (k, want[k], f.get(k, 0), f.get(k, 0) / want[k], w.get(k, 0), w.get(k, 0) / want[k]))

req = {c: load("REQ", "TCC_EA0_RDREQ%s_sum" % c, 1.0) for c in ("", "_32B", "_64B", "_128B")}
if req[""]:
    print("request-size counters (TCC_EA0_RDREQ{,_32B,_64B,_128B}_sum): bytes = 32 n32 + 64 n64 + 128 n128")
    print("%-14s %12s %12s %12s %12s %14s %8s" % ("kernel", "RDREQ", "32B", "64B", "128B", "bytes", "ratio"))
    for k in want:
        n, n32, n64, n128 = (req[c].get(k, 0) for c in ("", "_32B", "_64B", "_128B"))
        b = 32 * n32 + 64 * n64 + 128 * n128
        print("%-14s %12.0f %12.0f %12.0f %12.0f %14.0f %8.3f" % (k, n, n32, n64, n128, b, b / want[k]))
