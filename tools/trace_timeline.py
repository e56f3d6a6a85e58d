"""Per-frame GPU timeline from a rocprofv3 --kernel-trace CSV (scripts/trace_gpu.sh): busy/idle, overlap, per-queue order.

    python tools/trace_timeline.py gpurun_out/t1_kernel_trace.csv [--frames 3]
"""
import argparse
import csv
import re


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:40]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("csv")
    ap.add_argument("--frames", type=int, default=2)
    a = ap.parse_args()
    rows = list(csv.DictReader(open(a.csv)))
    ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Queue_Id"]) for r in rows]
    ev.sort()
    vols = [i for i, e in enumerate(ev) if e[2].startswith("corr_volume")]
    # steady state: last third of the volume launches
    i0, i1 = vols[len(vols) * 2 // 3], vols[-2]
    nfr = sum(1 for i in vols if i0 <= i < i1)
    t0, t1 = ev[i0][0], ev[i1][0]
    print(f"steady window: {nfr} frames, {(t1 - t0) / nfr / 1e3:.1f} us/frame")
    seg = [e for e in ev if t0 <= e[0] < t1]
    # union busy time + concurrency histogram
    pts = sorted([(s, 1) for s, e, _, _ in seg] + [(min(e, t1), -1) for s, e, _, _ in seg])
    busy = {0: 0, 1: 0, 2: 0, 3: 0}
    cur, last = 0, t0
    for t, d in pts:
        busy[min(cur, 3)] += t - last
        cur += d
        last = t
    tot = t1 - t0
    print("concurrency: " + "  ".join(f"{k}{'+' if k == 3 else ''} kernels {v / tot * 100:.1f}%" for k, v in busy.items()))
    agg = {}
    for s, e, n, q in seg:
        d = agg.setdefault((n, q), [0, 0])
        d[0] += 1
        d[1] += e - s
    print("per frame, by kernel (queue): calls, avg us, total us/frame")
    for (n, q), (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"  {n:40s} q{q:>2} {c / nfr:5.1f} x {t / c / 1e3:8.2f} = {t / nfr / 1e3:8.1f}")
    print(f"\nfirst {a.frames} frames of the window (us relative):")
    tend = ev[vols[vols.index(i0) + a.frames]][0]
    for s, e, n, q in seg:
        if s >= tend:
            break
        print(f"  {(s - t0) / 1e3:9.1f} -> {(e - t0) / 1e3:9.1f}  ({(e - s) / 1e3:7.1f})  q{q:>2} {n}")


if __name__ == "__main__":
    main()
