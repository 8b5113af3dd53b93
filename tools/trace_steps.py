#!/usr/bin/env python3
"""Per-step kernel summary of a rocprofv3 --kernel-trace CSV for steps that voxelise a BATCH of clouds: a step starts at every
`batch`-th vox_insert launch.  usage: trace_steps.py <kernel_trace.csv> <batch> [last_n_steps] [--top N] [--group]"""
import collections, csv, re, sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    return name


def main():
    a = [x for x in sys.argv[1:] if not x.startswith("--")]
    top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 50
    if "--top" in sys.argv:
        a.remove(str(top))
    rows = list(csv.DictReader(open(a[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    batch = int(a[1])
    last = int(a[2]) if len(a) > 2 else 3
    first = [i for i, r in enumerate(rows) if "vox_insert" in r["Kernel_Name"]][::batch]
    lo, hi = len(first) - 1 - last, len(first) - 2
    tot, cnt = collections.Counter(), collections.Counter()
    busy = wall = nk = 0
    for s in range(lo, hi + 1):
        seg = rows[first[s]:first[s + 1]]
        wall += int(rows[first[s + 1]]["Start_Timestamp"]) - int(seg[0]["Start_Timestamp"])
        for r in seg:
            d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
            busy += d
            k = short(r["Kernel_Name"])
            tot[k] += d
            cnt[k] += 1
        nk += len(seg)
    ns = hi - lo + 1
    print("steps %d..%d: per step wall(profiled) %.2f ms, kernel busy %.2f ms, %d kernel launches" % (lo, hi, wall / ns / 1e6, busy / ns / 1e6, nk // ns))
    for k, v in tot.most_common(top):
        print("%8.1f us/step %7.1f calls avg %8.1f us %5.1f%%  %s" % (v / ns / 1e3, cnt[k] / ns, v / cnt[k] / 1e3, 100.0 * v / busy, k[:110]))


if __name__ == "__main__":
    main()
