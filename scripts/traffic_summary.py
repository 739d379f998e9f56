"""Per-kernel HBM-side bytes per launch from two rocprofv3 counter_collection.csv files (read pass,
write pass).  Read requests are tallied by size (32 / 64 / 128 B counters; requests in none of
the sized counters count as 64 B); write requests are 64 B or 32 B."""
import collections, csv, json, sys

def load(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        disp[k].add(r["Dispatch_Id"])
    return {k: {c: v / len(disp[k]) for c, v in cs.items()} for k, cs in agg.items()}, {k: len(v) for k, v in disp.items()}

rd, n_rd = load(sys.argv[1])
wr, n_wr = load(sys.argv[2])
out = {}
for k in rd:
    if not any(s in k for s in ("vertex_kernel", "setup_kernel", "setup_views_kernel", "raster_kernel", "raster_views_kernel", "resolve_surface")):
        continue
    r, w = rd[k], wr.get(k, {})
    n32, n64, n128, n = r.get("TCC_EA0_RDREQ_32B_sum", 0), r.get("TCC_EA0_RDREQ_64B_sum", 0), r.get("TCC_EA0_RDREQ_128B_sum", 0), r.get("TCC_EA0_RDREQ_sum", 0)
    other = max(n - n32 - n64 - n128, 0)
    read_bytes = 32 * n32 + 64 * (n64 + other) + 128 * n128
    w64, wn = w.get("TCC_EA0_WRREQ_64B_sum", 0), w.get("TCC_EA0_WRREQ_sum", 0)
    write_bytes = 64 * w64 + 32 * max(wn - w64, 0)
    hit, miss = w.get("TCC_HIT_sum", 0), w.get("TCC_MISS_sum", 0)
    out[k] = {"launches": n_rd[k], "read_requests": {"32B": n32, "64B": n64, "128B": n128, "all": n}, "read_bytes": read_bytes,
              "write_requests": {"64B": w64, "all": wn}, "write_bytes": write_bytes, "hbm_bytes": read_bytes + write_bytes,
              "l2_hit_rate": hit / (hit + miss) if hit + miss else None}
print(json.dumps(out, indent=1))
