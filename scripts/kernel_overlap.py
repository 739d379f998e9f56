"""Overlap structure of the frame kernels from a rocprofv3 kernel trace (kernel_trace.csv): per kernel the mean duration,
and for the steady part of the run the share of wall time during which k kernels (and which) are running at once.
usage: python scripts/kernel_overlap.py <kernel_trace.csv>"""
import collections, csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if any(k in r["Kernel_Name"] for k in ("setup_kernel", "raster_kernel", "resolve_surface"))]
ev = []
short = lambda n: "surface" if "resolve_surface" in n else "raster" if "raster_kernel" in n else "setup"
dur = collections.defaultdict(list)
for r in rows:
    b, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    ev.append((b, e, short(r["Kernel_Name"])))
ev.sort()
# steady part: the middle half of the launches
n = len(ev)
mid = ev[n // 4: 3 * n // 4]
t0, t1 = mid[0][0], mid[-1][1]
for b, e, k in mid:
    dur[k].append(e - b)
print("kernels in the window:", {k: len(v) for k, v in dur.items()}, "window %.3f ms" % ((t1 - t0) / 1e6))
for k, v in dur.items():
    print("  %-8s mean %.1f us  min %.1f  max %.1f" % (k, sum(v) / len(v) / 1e3, min(v) / 1e3, max(v) / 1e3))
frames = len(dur["surface"])
print("  wall time per frame %.1f us" % ((t1 - t0) / 1e3 / max(frames, 1)))
points = []
for b, e, k in mid:
    points.append((b, 1, k)); points.append((e, -1, k))
points.sort()
state = collections.Counter()
share = collections.Counter()
last = points[0][0]
for t, d, k in points:
    key = tuple(sorted((kk, c) for kk, c in state.items() if c))
    share[key] += t - last
    last = t
    state[k] += d
total = sum(share.values())
for key, v in sorted(share.items(), key=lambda kv: -kv[1])[:14]:
    print("  %5.1f %%  %s" % (100.0 * v / total, " + ".join("%d %s" % (c, k) for k, c in key) or "idle"))
