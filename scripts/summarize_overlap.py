"""Reads a rocprofv3 kernel + memory-copy trace of one rank of a sharded run and reports how much of the
exchange (host-staged copies under gloo, RCCL kernels under nccl) ran while scan kernels were running."""
import csv, glob, os, sys
d = sys.argv[1]
kf = sorted(glob.glob(os.path.join(d, "*", "*_kernel_trace.csv")))[-1]
mf = sorted(glob.glob(os.path.join(d, "*", "*_memory_copy_trace.csv")))
scans, others, exch = [], [], []
for r in csv.DictReader(open(kf)):
    n = r["Kernel_Name"]
    iv = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]))
    if "scan_log" in n or "split_owner" in n or "line_count" in n:
        scans.append(iv)
    elif ("ccl" in n.lower() or "nccl" in n.lower()) and "rocclr" not in n:
        exch.append(iv)
    elif "tsx::" in n:
        others.append(iv)
if mf:
    for r in csv.DictReader(open(mf[-1])):
        b = int(r.get("Bytes", r.get("Size", "0")) or 0) if any(k in r for k in ("Bytes", "Size")) else 1 << 30
        iv = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]))
        if b >= (8 << 20):
            exch.append(iv)
def overlap(a, bs):
    t = 0
    for s, e in bs:
        t += max(0, min(a[1], e) - max(a[0], s))
    return t
tot_exch = sum(e - s for s, e in exch)
ov = sum(overlap(x, scans) for x in exch)
print("scan-phase kernels: %d, %.2f ms;  exchange intervals (copies >= 8 MiB or *ccl* kernels): %d, %.2f ms"
      % (len(scans), sum(e - s for s, e in scans) / 1e6, len(exch), tot_exch / 1e6))
print("exchange time that ran while a scan-phase kernel of a later window was running: %.2f ms (%.0f %%)"
      % (ov / 1e6, 100.0 * ov / max(tot_exch, 1)))
print("partition + build kernels: %d, %.2f ms" % (len(others), sum(e - s for s, e in others) / 1e6))
