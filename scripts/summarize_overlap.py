"""Reads a rocprofv3 kernel + memory-copy trace of one rank of a sharded run and reports how much of the
exchange (host-staged copies under gloo, RCCL kernels under nccl) ran while scan kernels were running."""
import csv, glob, os, sys
d = sys.argv[1]
kf = sorted(glob.glob(os.path.join(d, "*", "*_kernel_trace.csv")))[-1]
mf = sorted(glob.glob(os.path.join(d, "*", "*_memory_copy_trace.csv")))
scans, others, exch, walks = [], [], [], []
named = []
for r in csv.DictReader(open(kf)):
    n = r["Kernel_Name"]
    iv = (int(r["Start_Timestamp"]), int(r["End_Timestamp"]))
    named.append((iv, n))
    if any(t in n for t in ("scan_log", "split_owner", "line_count", "strip_desc", "desc_pack", "desc_prefix")):
        scans.append(iv)   # what a window does BEFORE its exchange (key exchange: scan + split; description exchange: describe + pack)
    elif "walk_part" in n or "walk_log" in n:
        walks.append(iv)   # description exchange: what a window does AFTER its exchange
    elif ("ccl" in n.lower() or "nccl" in n.lower()) and "rocclr" not in n:
        exch.append(iv)
    elif "rocclr_copyBuffer" in n:   # gloo: tensor.cpu() / copy_() of pageable memory run as copy kernels, chunk by chunk
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
print("scan-phase kernels: %d, %.2f ms;  exchange intervals (*ccl* kernels, staging copy kernels of gloo, copies >= 8 MiB): %d, %.2f ms"
      % (len(scans), sum(e - s for s, e in scans) / 1e6, len(exch), tot_exch / 1e6))
print("exchange time that ran while a scan-phase kernel of a later window was running: %.2f ms (%.0f %%)"
      % (ov / 1e6, 100.0 * ov / max(tot_exch, 1)))
if walks:
    ovw = sum(overlap(x, walks) for x in exch)
    print("walk kernels (description exchange, after a window's all-gather): %d, %.2f ms; exchange time that ran while a walk of an "
          "earlier window was running: %.2f ms (%.0f %%)" % (len(walks), sum(e - s for s, e in walks) / 1e6, ovw / 1e6,
                                                            100.0 * ovw / max(tot_exch, 1)))
print("partition + build kernels: %d, %.2f ms" % (len(others), sum(e - s for s, e in others) / 1e6))

# Under gloo the transfer itself is CPU work (invisible here); what the GPU trace shows of an exchange is its staging
# copies before and after.  Per step (a step ends with its build kernel): the hull of the staging copies >= 20 us, and how
# much of the scan-phase kernel time lies INSIDE it -- a step that scanned everything first and exchanged afterwards
# would show 0 %, W windows with the exchange of window i behind the scan of window i+1 about (W-1)/W.
builds = sorted(e for (s_, e), n in ((iv, n) for iv, n in named) if "build_segments" in n)
big = [x for x in exch if x[1] - x[0] >= 20000]
prev = 0
for i, be in enumerate(builds):
    ex = [x for x in big if prev < x[0] < be]
    sc = [x for x in scans if prev < x[0] < be]
    wk = [x for x in walks if prev < x[0] < be]
    if ex and sc:
        h0, h1 = min(x[0] for x in ex), max(x[1] for x in ex)
        inside = sum(max(0, min(h1, e) - max(h0, s_)) for s_, e in sc)
        tot = sum(e - s_ for s_, e in sc)
        line = "step %d: exchange activity spans %.1f ms; scan-phase kernel time inside that span: %.2f of %.2f ms (%.0f %%)" % (
            i, (h1 - h0) / 1e6, inside / 1e6, tot / 1e6, 100.0 * inside / max(tot, 1))
        if wk:
            wi = sum(max(0, min(h1, e) - max(h0, s_)) for s_, e in wk)
            wt = sum(e - s_ for s_, e in wk)
            line += "; walk time inside: %.2f of %.2f ms (%.0f %%)" % (wi / 1e6, wt / 1e6, 100.0 * wi / max(wt, 1))
        print(line)
    prev = be
