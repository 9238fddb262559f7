#!/usr/bin/env python3
"""GPU busy fraction and per-kernel concurrency from a rocprofv3 kernel trace CSV.

usage: trace_busy.py <dir-with-*_kernel_trace.csv>
Prints, for the window between the first and last kernel: wall time, union of
kernel intervals (busy), sum of kernel durations, and per-kernel count/avg/sum.
"""
import csv, glob, sys, collections

def main(d):
    files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
    iv = []
    last_render = 0
    per = collections.defaultdict(lambda: [0, 0])
    for f in files:
        for r in csv.DictReader(open(f)):
            s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
            iv.append((s, e))
            if "synth_render" in r["Kernel_Name"]: last_render = max(last_render, e)
            k = r["Kernel_Name"].split("(")[0][:48]
            per[k][0] += 1; per[k][1] += e - s
    iv.sort()
    busy = 0; cs, ce = iv[0]
    for s, e in iv[1:]:
        if s > ce: busy += ce - cs; cs, ce = s, e
        else: ce = max(ce, e)
    busy += ce - cs
    wall = max(e for _, e in iv) - iv[0][0]
    tot = sum(e - s for s, e in iv)
    print(f"kernels {len(iv)}  wall {wall/1e6:.1f} ms  busy(union) {busy/1e6:.1f} ms ({busy/wall:.2%})  sum {tot/1e6:.1f} ms  avg concurrency when busy {tot/busy:.2f}")
    # the bench itself: everything after the last frame of the synthetic input was rendered
    run = [(s, e) for s, e in iv if s >= last_render]
    if last_render and run:
        b2 = 0; cs, ce = run[0]
        for s, e in run[1:]:
            if s > ce: b2 += ce - cs; cs, ce = s, e
            else: ce = max(ce, e)
        b2 += ce - cs
        w2 = max(e for _, e in run) - run[0][0]
        t2 = sum(e - s for s, e in run)
        print(f"after the input rendering (warm-up + timed steps): wall {w2/1e6:.1f} ms  busy(union) {b2/1e6:.1f} ms ({b2/w2:.2%})  sum {t2/1e6:.1f} ms  avg concurrency when busy {t2/b2:.2f}")
    for k, (n, t) in sorted(per.items(), key=lambda x: -x[1][1]):
        print(f"  {k:48s} n={n:7d} avg={t/n/1e3:9.1f} us  sum={t/1e6:9.1f} ms ({t/tot:.1%})")

if __name__ == "__main__":
    main(sys.argv[1])
