"""Idle-gap attribution from a rocprofv3 kernel trace CSV: python tools/gaps.py <kernel_trace.csv> [steps]"""
import csv, sys, collections
path, steps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].split("<")[0]))
rows.sort()
marks = [i for i, r in enumerate(rows) if r[2].endswith("k_render_loss")] or [i for i, r in enumerate(rows) if r[2].endswith("k_loss")]   # one per step
i0, i1 = marks[-steps - 1], marks[-1]
win = rows[i0:i1]
t0, t1 = win[0][0], rows[i1][0]
busy = 0; gaps = collections.Counter(); gapn = collections.Counter(); kern = collections.Counter(); kn = collections.Counter()
cur_end = win[0][0]
prev = None
for s, e, n in win:
    kern[n] += e - s; kn[n] += 1
    if s > cur_end:
        gaps[(prev, n)] += s - cur_end; gapn[(prev, n)] += 1
    if e > cur_end:
        busy += e - max(s, cur_end); cur_end = e; prev = n
tot = t1 - t0
print("window %.3f ms/step, busy %.3f ms/step, idle %.3f ms/step, %d launches/step" % (tot / steps / 1e6, busy / steps / 1e6, (tot - busy) / steps / 1e6, len(win) // steps))
print("-- top gaps (us/step, count/step, avg us)")
for (a, b), g in gaps.most_common(25):
    print("%8.1f %6.2f %7.1f  %s -> %s" % (g / steps / 1e3, gapn[(a, b)] / steps, g / gapn[(a, b)] / 1e3, a, b))
print("-- kernels (us/step, count/step)")
for n, g in kern.most_common(30):
    print("%8.1f %6.2f  %s" % (g / steps / 1e3, kn[n] / steps, n))
# -- exclusive time: the part of a launch during which nothing else runs on the device (what a kernel hidden under another one does
# not have); sum over names = the part of `busy` with exactly one kernel resident
ev = []
for s, e, n in win:
    ev.append((s, 1, n)); ev.append((e, -1, n))
ev.sort(key=lambda x: (x[0], x[1]))
live = collections.Counter(); nlive = 0; last = ev[0][0]; excl = collections.Counter(); multi = 0
for t, d, n in ev:
    if t > last:
        if nlive == 1:
            excl[next(k for k, v in live.items() if v > 0)] += t - last
        elif nlive > 1:
            multi += t - last
    last = t
    live[n] += d; nlive += d
print("-- exclusive time (us/step): one kernel resident %.1f, two or more %.1f" % (sum(excl.values()) / steps / 1e3, multi / steps / 1e3))
for n, g in excl.most_common(30):
    print("%8.1f  of %8.1f  %s" % (g / steps / 1e3, kern[n] / steps / 1e3, n))
cp = sorted(e - s for s, e, n in win if n.endswith("copyBuffer"))
if cp:
    print("-- __amd_rocclr_copyBuffer durations (us): n/step %.2f, p10 %.1f, p50 %.1f, p90 %.1f, max %.1f" % (
        len(cp) / steps, cp[len(cp) // 10] / 1e3, cp[len(cp) // 2] / 1e3, cp[len(cp) * 9 // 10] / 1e3, cp[-1] / 1e3))
