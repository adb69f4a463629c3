#!/usr/bin/env python3
"""bench.py -- k-mers/sec inserted, k=31, synthetic FASTQ, --mode=HIP hot path.

One "step" = one pass of the hot path over one batch: clear the table, then
FASTQ scan -> 2-bit encode -> hash -> dedup -> insert for ~1e9 k-mers per GPU of
synthetic reads (generateFakeSequences.py shape) already resident in HBM; with
N > 1 ranks (one process per GPU, reads sharded, weak scaling) the step ends
with the table merge over RCCL.  Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md)


def cpu_baseline(k, seed, max_seconds=40, budget_seconds=110):
    """Time the real reference (oracle/_ref/tsxCount_ref) on a bounded sample of the same synthetic reads, on this
    box's host cores: `--mode=OMP` at the host's core count (north_star: "CAS/OMP path") and `--mode=CAS`, the better
    of the runs that exit 0 is reported with its mode, every attempt is listed.

    What is reported: k-mers of the sample / (wall time of the run - wall time of the same binary on an EMPTY input),
    i.e. process start-up, option parsing, the random-matrix set-up and the allocation of the 2^23-slot table are
    measured and subtracted.  `cores` = the OpenMP threads the run used, `host_cores` = os.cpu_count() of the box.  The
    reference's parallel modes are not robust (unsynchronised std::set inserts and retry loops: CAS live-locks at 16
    threads on this input and sometimes crashes at 8), so every attempt has a time limit and thread counts go down."""
    from tsxcount_amd import synth
    ref_bin = os.path.join(ROOT, "oracle", "_ref", "tsxCount_ref")
    attempts, good = [], []
    host_cores = os.cpu_count() or 1
    t_start = time.time()

    def run_ref(td, text, threads, mode):
        path = os.path.join(td, "sample.fastq")
        with open(path, "wb") as f:
            f.write(text)
        # 2k+s must be a multiple of 8 for the reference's byte-wise CAS
        # stores to stay aligned (TSXHashMapCAS.h:141-232): k=31 -> s=2.
        cmd = [ref_bin, "--input=" + path, "--k=%d" % k, "--l=23", "--s=%d" % ((-2 * k) % 8 or 8), "--mode=" + mode,
               "--threads=%d" % threads]
        t0 = time.time()
        try:
            rc = subprocess.run(cmd, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                                timeout=max_seconds).returncode
        except subprocess.TimeoutExpired:
            rc = "timeout"
        return rc, time.time() - t0

    if os.path.exists(ref_bin):
        plan = [("OMP", min(host_cores, 64), 3000), ("OMP", 16, 3000), ("CAS", 8, 3000), ("CAS", 8, 3000),
                ("CAS", 4, 1500), ("CAS", 2, 800), ("CAS", 1, 400)]
        for mode, threads, n_reads in plan:
            if time.time() - t_start > budget_seconds or (good and mode == "CAS" and any(g["mode"] == "CAS" for g in good)):
                break
            threads = max(1, min(threads, host_cores))
            text = synth.fastq(seed, 0, n_reads)
            nrand, na = synth.read_lengths(seed, 0, n_reads)
            kmers = int(((nrand + na) - k + 1).clip(min=0).sum())
            with tempfile.TemporaryDirectory() as td:
                rc0, dt0 = run_ref(td, b"", threads, mode)
                rc, dt = run_ref(td, text, threads, mode)
            if rc == 0 and rc0 == 0 and dt > dt0:
                good.append({"value": kmers / (dt - dt0), "unit": "k-mers/s", "cores": threads, "host_cores": host_cores,
                             "kind": "reference", "mode": mode, "seconds": round(dt, 2),
                             "startup_seconds_subtracted": round(dt0, 3),
                             "sample": "%d synthetic reads (%d k-mers), k=%d, tsxCount --mode=%s --l=23 --s=%d "
                                       "--threads=%d; wall time minus the wall time of an empty-input run"
                                       % (n_reads, kmers, k, mode, (-2 * k) % 8 or 8, threads)})
            attempts.append({"mode": mode, "threads": threads, "reads": n_reads, "rc": rc, "seconds": round(dt, 1),
                             "kmers_per_s": round(kmers / (dt - dt0), 1) if (rc == 0 and rc0 == 0 and dt > dt0) else None})
        if good:
            best = max(good, key=lambda g: g["value"])
            best["attempts"] = attempts
            return best
    # fall back to the C restatement (single core)
    from oracle.oracle import Oracle
    n_reads = 12000
    text = synth.fastq(seed, 0, n_reads)
    nrand, na = synth.read_lengths(seed, 0, n_reads)
    kmers = int(((nrand + na) - k + 1).clip(min=0).sum())
    o = Oracle(k, 25, 2, seed=1)
    t0 = time.time()
    o.count_fastq(text)
    dt = time.time() - t0
    return {"value": kmers / dt, "unit": "k-mers/s", "cores": 1, "host_cores": host_cores, "kind": "port",
            "seconds": round(dt, 2),
            "sample": "%d synthetic reads (%d k-mers), k=%d, oracle/tsx_oracle.c serial, in-process "
                      "(count phase only)" % (n_reads, kmers, k),
            "attempts": attempts}


def run_check(args, m, T, TD, dist, torch, dist_on, sharded, world, rank, red, dev, local_rank, text, nbytes,
              kmers_rank, kmers_total, first):
    """The --check of the synthetic workload, outside the timed region (tsxcount_amd/verify.py):
    totals read back from the table, the analytic polyA count, exact lookups of a sample of
    reads, and (one GPU) a second table filled through the other insert path."""
    from tsxcount_amd import verify
    detail = {}
    st = m.stats()
    ok = st["insert_failures"] == 0 and st["overflow_failures"] == 0 and st["lock_timeouts"] == 0
    detail["failure_counters"] = {k2: st[k2] for k2 in ("insert_failures", "overflow_failures", "lock_timeouts")}

    def allsum(v):
        if not dist_on:
            return int(v)
        t = torch.tensor([int(v)], dtype=torch.int64, device=red)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return int(t.item())

    if args.workload == "zipf":
        # everything the table must hold follows from the generator (tsxcount_amd.synth.ZipfExpect): the number of
        # occurrences and of DISTINCT k-mers, the count of every k-mer of a sample of reads and of the hottest k-mer
        from tsxcount_amd import synth
        ex = synth.ZipfExpect(args.seed, args.reads, args.zipf_read_len, synth.zipf_thresholds(args.zipf_templates, args.zipf_a), args.k)
        detail["totals"] = {"kmers_in_reads": ex.total, "scanned": st["kmers_added"], "sum_of_counts_in_table": st["count_sum"],
                            "distinct_expected": ex.distinct, "distinct_in_table": st["distinct"]}
        ok = ok and st["kmers_added"] == ex.total == st["count_sum"] and st["distinct"] == ex.distinct
        ids = verify.sample_read_ids(0, args.reads, args.check_reads)
        seqs, cnt = ex.sample(ids)
        got = m.getKmerCounts(T.encode_many(seqs, args.k))
        detail["sample"] = {"reads": int(len(ids)), "looked_up": int(len(seqs)), "unequal": int((got != cnt).sum()),
                            "largest_expected_count": int(cnt.max())}
        ok = ok and len(seqs) > 0 and detail["sample"]["unequal"] == 0
        t, p, c = ex.hottest()
        hot = int(m.getKmerCounts(T.encode_many([synth.zipf_template(args.seed, t, p, args.k)], args.k))[0])
        detail["hottest_kmer"] = {"expected": c, "table": hot}
        ok = ok and hot == c
        import ctypes
        dbg8 = (ctypes.c_uint64 * 8)()
        m._lib.tsx_hip_debug_counters.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_uint64)]
        m._lib.tsx_hip_debug_counters(m.handle, dbg8)
        detail["skew"] = {"records_off_the_fast_route": st["fallback_inserts"], "deferred_list_entries": int(dbg8[7]),
                          "overflow_carries": st["overflow_carries"]}
        if not args.no_cross_check and m.layout.table_bytes <= (32 << 30):
            other = "partitioned" if args.path == "atomic" else "atomic"
            detail["cross"] = verify.cross_check(m, text.data_ptr(), nbytes, other, device=local_rank)
            ok = ok and detail["cross"]["ok"]
        return ok, detail
    # totals: every k-mer of the reads was scanned, and the counts held by the table(s) add up to them
    # (minimizer exchange: the owners add the homopolymer totals a second time to that counter)
    scanned = allsum(st["kmers_added"]) if ((world == 1 or sharded) and not getattr(args, "mini", False)) else None
    count_sum = allsum(st["count_sum"])
    detail["totals"] = {"kmers_in_reads": kmers_total, "scanned": scanned, "sum_of_counts_in_table": count_sum}
    ok = ok and count_sum == kmers_total and (scanned is None or scanned == kmers_total)
    # polyA: analytic count from the generator (all ranks' reads) vs the table (its owner answers)
    _, _, npolya = T.synth_sizes(args.seed, first, args.reads, args.k, want_polya=True)
    polya_expect = allsum(npolya)
    polya_got = allsum(m.getKmerCount("A" * args.k))
    detail["polyA"] = {"expected": polya_expect, "table": polya_got}
    ok = ok and polya_got == polya_expect
    # sample of rank 0's reads, looked up on every rank (non-owners answer 0), summed
    ids = verify.sample_read_ids(0, args.reads, args.check_reads)
    kmers, mult, safe = verify.sample_expectations(args.seed, args.k, ids)
    got = m.getKmerCounts(kmers)
    if dist_on:
        g = torch.from_numpy(got.astype("int64")).to(red)
        dist.all_reduce(g, op=dist.ReduceOp.SUM)
        got = g.cpu().numpy().astype("uint64")
    detail["sample"] = dict(verify.judge_sample(got, mult, safe), reads=int(len(ids)))
    ok = ok and detail["sample"]["below_sample_multiplicity"] == 0 and detail["sample"]["safe_unequal"] == 0 \
        and detail["sample"]["looked_up"] > 0
    # second table through the other insert path (single table only)
    if not dist_on and not args.no_cross_check and m.layout.table_bytes <= (32 << 30):   # (a second table of that size)
        other = "partitioned" if args.path == "atomic" else "atomic"
        detail["cross"] = verify.cross_check(m, text.data_ptr(), nbytes, other, device=local_rank)
        ok = ok and detail["cross"]["ok"]
    if dist_on:
        # every distinct k-mer lives on exactly one rank
        detail["distinct_total"] = allsum(st["distinct"])
        ok = ok and 0 < detail["distinct_total"] <= kmers_total
        okt = torch.tensor([1 if ok else 0], dtype=torch.int64, device=red)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        ok = bool(int(okt.item()))
    return ok, detail


def self_launch(n):
    """Start n ranks of this script through torch.distributed.run as a child process; returns its exit code.
    Nothing here imports torch or touches the GPU."""
    import socket
    sock = socket.socket()
    sock.bind(("127.0.0.1", 0))
    port = sock.getsockname()[1]
    sock.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    # the ranks read their options from the environment: the launcher's own parser trips over script options that
    # abbreviate one of its own ("--l" is a prefix of --log-dir, --local-addr, ...)
    env["TSX_BENCH_ARGV"] = json.dumps(sys.argv[1:])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)]
    sys.stderr.write("bench.py: starting %d ranks: %s\n" % (n, " ".join(cmd)))
    sys.stderr.flush()
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE)
    lines = []
    for raw in child.stdout:               # rank 0 prints ONE JSON line; anything else on stdout goes to stderr
        line = raw.decode("utf-8", "replace")
        if line.lstrip().startswith("{") and '"metric"' in line:
            lines.append(line)
        else:
            sys.stderr.write(line)
    rc = child.wait()
    for line in lines[-1:]:
        sys.stdout.write(line)
    sys.stdout.flush()
    if rc == 0 and not lines:
        sys.stderr.write("bench.py: the ranks exited 0 without a result line\n")
        return 1
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--k", type=int, default=31)
    ap.add_argument("--l", "--table-bits", dest="l", type=int, default=30, help="log2 table slots per GPU (load factor 0.75 at the default input)")
    ap.add_argument("--reads", type=int, default=1087000, help="synthetic reads per GPU (~1e9 k-mers at k=31)")
    ap.add_argument("--seed", type=int, default=20261004)
    ap.add_argument("--path", default="auto", choices=["auto", "atomic", "partitioned"])
    ap.add_argument("--merge", default="auto", choices=["auto", "mini", "shard", "tables"],
                    help="N > 1: mini = owner of a k-mer = f(its minimizer), masked strip descriptions travel, every GPU a whole "
                         "table of what it owns (20 <= k <= 32, any N <= 16); shard = ONE table sharded by slot range (N a "
                         "power of two, k <= 32); tables = per-GPU tables merged afterwards (any k); auto = mini from 4 GPUs "
                         "on (by the one-GPU simulations of a step, DESIGN.md section 6), else shard, else tables")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="nccl = RCCL (one GPU per rank); gloo only to rehearse N > 1 on a single GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--check-reads", type=int, default=1000, help="reads whose k-mers the check looks up one by one")
    ap.add_argument("--no-cross-check", action="store_true",
                    help="skip the second table (other insert path) of the check")
    ap.add_argument("--workload", default="reads", choices=["reads", "zipf"],
                    help="reads: generateFakeSequences.py shape (the BASELINE metric); zipf: BASELINE config 4 -- windows of "
                         "Zipf(1.2)-picked template sequences (one GPU; give --k 63)")
    ap.add_argument("--zipf-read-len", type=int, default=250)
    ap.add_argument("--zipf-templates", type=int, default=1 << 20)
    ap.add_argument("--zipf-a", type=float, default=1.2)
    ap.add_argument("--force-dist", action="store_true",
                    help="run the N > 1 code path (process group, sharded table, collectives) even at world size 1: "
                         "the only way to push the RCCL leg through its API on a 1-GPU box")
    argv = sys.argv[1:]
    if not argv and "TSX_BENCH_ARGV" in os.environ:      # a rank started by self_launch()
        argv = json.loads(os.environ["TSX_BENCH_ARGV"])
    args = ap.parse_args(argv)

    # One process per GPU.  Under a launcher (python -m torch.distributed.run sets WORLD_SIZE) this process is one
    # rank.  A bare `python bench.py --gpus N` with N > 1 starts the N ranks itself -- as CHILD processes, before
    # torch or the GPU is touched here (no exec from a process that has initialised the GPU) -- relays rank 0's
    # JSON line and exits with the children's code: one command runs the job, like the reference's CLI
    # (src/mains/main.cpp:404-507).  A mismatch inside a launcher is refused.
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        raise SystemExit(self_launch(args.gpus))
    env_world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != env_world:
        sys.stderr.write("bench.py: --gpus %d but the launcher set WORLD_SIZE=%d. Launch one rank per GPU:\n"
                         "  python -m torch.distributed.run --nnodes=1 --nproc-per-node %d --master-addr 127.0.0.1 "
                         "--master-port 29671 bench.py --gpus %d --steps %d --warmup %d\n"
                         % (args.gpus, env_world, args.gpus, args.gpus, args.steps, args.warmup))
        raise SystemExit(2)

    import torch
    import torch.distributed as dist
    import tsxcount_amd as T
    from tsxcount_amd import distributed as TD

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1 or args.force_dist
    if dist_on:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29671")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # stdout carries ONE JSON line: RCCL's start-up banner ("RCCL version : ...") goes to stderr
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
            if args.backend == "nccl" and torch.cuda.is_available():
                torch.cuda.set_device(local_rank)
                warm = torch.zeros((1,), dtype=torch.int64, device="cuda")
                dist.all_reduce(warm)   # communicator set-up happens (and prints) here
                torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
            os.close(saved_stdout)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if args.backend == "gloo":
        local_rank = local_rank % max(1, torch.cuda.device_count())
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    red = torch.device("cpu") if args.backend == "gloo" else dev  # where the tiny reductions live

    # synthetic input straight into HBM; every rank owns its own read shard
    first = rank * args.reads
    zipf = args.workload == "zipf"
    if zipf:
        if dist_on:
            raise SystemExit("bench.py: --workload zipf is a one-GPU configuration (BASELINE config 4)")
        from tsxcount_amd import synth
        zthr = synth.zipf_thresholds(args.zipf_templates, args.zipf_a)
        nbytes = T.synth_zipf_device(args.seed, args.reads, args.zipf_read_len, zthr)
        kmers_rank = args.reads * max(0, args.zipf_read_len - args.k + 1)
        text = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev)
        text[nbytes:] = 10
        torch.cuda.synchronize(dev)
        T.synth_zipf_device(args.seed, args.reads, args.zipf_read_len, zthr, text.data_ptr(), nbytes, device=local_rank)
    else:
        nbytes, kmers_rank, _ = T.synth_sizes(args.seed, first, args.reads, args.k)
        text = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev)
        torch.cuda.synchronize(dev)
        T.synth_fastq_device(args.seed, first, args.reads, args.k, text.data_ptr(), nbytes, device=local_rank)

    # N > 1: ONE table sharded by slot range over the GPUs (2^(l + log2 N) slots in all, so the
    # load factor per GPU is the same at every N: weak scaling).  Keys travel to their owner
    # through one RCCL all-to-all before they are built; see tsxcount_amd/distributed.py.
    can_mini = dist_on and 20 <= args.k <= 32 and world <= 16
    can_shard = dist_on and (world & (world - 1)) == 0 and args.k <= 32
    mode = args.merge
    if mode == "auto":
        mode = "mini" if (can_mini and world >= 4) else ("shard" if can_shard else ("mini" if can_mini else "tables"))
    if (mode == "mini" and not can_mini) or (mode == "shard" and not can_shard):
        raise SystemExit("bench.py: --merge %s does not fit k = %d on %d GPUs" % (mode, args.k, world))
    mini = dist_on and mode == "mini"
    sharded = dist_on and mode == "shard"
    args.mini = mini
    bits = world.bit_length() - 1 if sharded else 0
    m = T.TSXHashMapHIP(args.l, 0, args.k, device=local_rank, shard_bits=bits, shard_index=rank if sharded else 0)
    m.set_path(args.path)
    sc = TD.ShardedCounter(m, nbytes) if sharded else (TD.MinimizerCounter(m, nbytes) if mini else None)

    def step():
        m.clear()
        if sharded or mini:
            sc.step(text.data_ptr(), nbytes)
        else:
            m.countFastqDevice(text.data_ptr(), nbytes)
            if dist_on:
                TD.merge_tables(m)  # per-GPU tables merged afterwards (any N, any k)
            else:
                m.sync()

    for _ in range(args.warmup):
        step()
    m.set_timing(True)
    if dist_on:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize(dev)
    if dist_on:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    stage, launches = m.get_stage_timing()
    scan_ms, count_ms = stage["line"], stage["scan"]
    build_ms = stage["level1"] + stage["level2"] + stage["build"] + stage["post"]
    m.set_timing(False)

    # max over ranks
    if dist_on:
        t = torch.tensor([elapsed], dtype=torch.float64, device=red)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        kt = torch.tensor([kmers_rank], dtype=torch.int64, device=red)
        dist.all_reduce(kt, op=dist.ReduceOp.SUM)
        kmers_total = int(kt.item())
    else:
        kmers_total = kmers_rank

    st = m.stats()
    check_ok, check_detail = run_check(args, m, T, TD, dist, torch, dist_on, sharded, world, rank, red, dev, local_rank,
                                       text, nbytes, kmers_rank, kmers_total, first)

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = kmers_total * args.steps / elapsed
        # Roofline.  Per k-mer the path must at least read its share of the text and read +
        # write one 8-byte slot: algorithmic bytes = text bytes + 16 B x k-mer occurrences
        # (DESIGN.md section 3).  Reported for the dominant kernel (the scan kernel, which in
        # the partitioned path reads the text and writes one 8-byte key per logged k-mer) and
        # for the whole device path of a step.
        # stage times are sums over the timed calls: one call per step, or (sharded table) one scan call per window
        # plus one build call per step -- per-step figures either way
        # (a table above 2^32 slots is built slab by slab: one timing tuple per text window and per slab, see count_slabs)
        slab_bits = max(0, args.l - 14 - 18) if (m.wk == 1 and m.layout.entry_limbs == 1 and not dist_on) else 0
        pieces = args.steps if (sharded or mini or slab_bits) else max(launches, 1)
        partitioned = build_ms / pieces > 0.5
        keys_logged = st["distinct"] if partitioned else 0
        # algorithmic bytes of each stage of one launch: the scan kernel reads the
        # text and writes one 8-byte key per logged k-mer (atomic path: reads the text, reads + writes one
        # slot per k-mer); a radix level reads and writes every key once; the build reads every key once
        # and writes every slot of the table once.
        # a logged record is 1, 2 or 4 words (3-limb keys travel as 4), a slot entry_limbs words
        rec_b = 8.0 * {1: 1, 2: 2, 3: 4, 4: 4}[m.wk]
        slot_b = 8.0 * m.layout.entry_limbs
        table_bytes = slot_b * (1 << args.l)
        stage_bytes = {"scan": nbytes + (rec_b * keys_logged if partitioned else 2 * slot_b * kmers_rank),
                       "level1": 2 * rec_b * keys_logged, "level2": 2 * rec_b * keys_logged,
                       "build": rec_b * keys_logged + table_bytes}
        # walk fused with radix level 1 (walk_part_kernel): there is no level-1 launch, its stage time is ~0
        fused = partitioned and m.wk == 1 and stage["level1"] / pieces < 0.05
        names = {"scan": (("strip_desc_kernel + walk_part_kernel" if fused else "strip_desc_kernel + walk_log_kernel") if m.wk == 1
                          else "strip_desc_wide_kernel<%d> + walk_log_wide_kernel<%d>" % (m.wk, m.wk)) if partitioned
                 else "count_fastq_kernel<%d>" % m.wk,
                 "level1": "partition_ring_kernel (level 1)", "level2": "partition_ring_kernel (level 2)",
                 "build": "build_segments_stream_kernel" if (m.wk == 1 and m.layout.entry_limbs == 1)
                 else "build_segments_wide_stream_kernel<%d>" % m.wk}
        if sharded and sc._mode() == "desc":
            # description exchange: "scan" = strip_desc_kernel + pack of every window, "level1" = the walks over the
            # descriptions of all GPUs (with the later windows' descriptions interleaved), then level 2 and the build
            names["scan"] = "strip_desc_kernel + desc_pack_kernel"
            flt = os.environ.get("TSX_HIP_SHARD_FILTER", "auto")
            names["level1"] = ("%s (descriptions of all %d GPUs)"
                               % ("walk_log_kernel with owner filter + partition_ring_kernel (level 1)"
                                  if (flt == "1" or (flt == "auto" and world >= 4)) else "walk_part_kernel", world))
            stage_bytes["scan"] = nbytes + nbytes // 2 * 2          # text read; descriptions written and packed
            stage_bytes["level1"] = world * (nbytes // 2) + rec_b * keys_logged
        if mini:
            # minimizer exchange: "scan" = strip_desc_kernel + desc_owner_split_kernel of every window, "level1" = the walks over
            # the lists this GPU was sent, then level 2 and the build
            names["scan"] = "strip_desc_kernel + desc_owner_split_kernel"
            names["level1"] = "walk_part_kernel (the descriptions this GPU owns, from all %d GPUs)" % world
            lists = 16 * (sc.last.get("received_descriptions", 0))
            stage_bytes["scan"] = nbytes + nbytes // 2 * 2 + lists       # text read; descriptions written, read, lists written
            stage_bytes["level1"] = lists + rec_b * keys_logged
        if slab_bits:
            names["scan"] = "strip_desc_kernel + desc_pack_kernel (every text window described once)"
            names["level1"] = "walk_part_kernel with slab filter (%d slabs x all descriptions)" % (1 << slab_bits)
            stage_bytes["scan"] = nbytes + nbytes // 2 * 2
            stage_bytes["level1"] = (1 << slab_bits) * (nbytes // 4) + rec_b * keys_logged
        stage_ms = {k2: stage[k2] / pieces for k2 in names}
        dom = max(stage_ms, key=lambda k2: stage_ms[k2])   # the kernel a step spends most time in
        kern_ms = stage_ms[dom]
        kern_bytes = stage_bytes[dom]
        achieved = kern_bytes / (kern_ms * 1e-3) / 1e9 if kern_ms > 0 else 0.0
        path_bytes = nbytes + 2 * slot_b * kmers_rank
        path_ms = (scan_ms + count_ms + build_ms) / pieces
        traffic = None
        prof = {}
        pmc = os.path.join(ROOT, "profiles", "round3_pmc.json" if args.k == 31 else "round3_pmc_k%d.json" % args.k)
        if not os.path.exists(pmc) and args.k == 31:
            pmc = os.path.join(ROOT, "profiles", "round2_pmc.json")
        if os.path.exists(pmc):
            try:
                prof = json.load(open(pmc))
                pp = prof.get("partitioned" if partitioned else "atomic", {})
                traffic = pp.get("stages", {}).get(dom) if partitioned else pp.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        if mini:
            how = (", owner of a k-mer = f(its minimizer): strip descriptions masked per owner travel by RCCL send/recv, "
                   "every GPU walks what it owns into a table of its own")
        elif sharded and sc._mode() == "desc":
            how = ", table sharded by slot range, strip descriptions all-gathered over RCCL, every GPU walks all and keeps what it owns"
        elif sharded:
            how = ", table sharded by slot range, keys exchanged by one RCCL all-to-all per window"
        else:
            how = ", per-GPU tables merged over RCCL all-to-all"
        out = {
            "metric": ("k-mers/sec inserted, k=%d, Zipf-skewed synthetic reads, 1 GPU; --check pass" % args.k) if zipf else
                      "k-mers/sec inserted, k=%d, 1e9 synthetic k-mers, 1/2/4/8 GPU; --check pass" % args.k,
            "value": value, "unit": "k-mers/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "rccl_ranks": (dist.get_world_size() if dist_on else 0), "backend": (dist.get_backend() if dist_on else None),
            "config": {"workload": ("Zipf(%.2f) windows of %d bases over %d templates (BASELINE config 4), "
                                    % (args.zipf_a, args.zipf_read_len, args.zipf_templates) if zipf else
                                    "synthetic FASTQ (generateFakeSequences.py shape), ") + "%d reads/GPU = %d k-mers/GPU, "
                                   "k=%d, table 2^%d slots/GPU, %s insert path%s%s"
                                   % (args.reads, kmers_rank, args.k, args.l,
                                      "partitioned" if partitioned else "atomic",
                                      (", built in %d slabs of 2^32 slots" % (1 << slab_bits)) if slab_bits else "",
                                      how if world > 1 else ""),
                       "k": args.k, "l": args.l, "kmers_per_gpu": kmers_rank, "fastq_bytes_per_gpu": nbytes,
                       "distinct_rank0": st["distinct"], "check": "pass" if check_ok else "FAIL",
                       "check_detail": check_detail},
            "roofline": {"bound": "hbm", "kernel": names[dom],
                         "achieved": achieved,
                         "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "kernel_ms": kern_ms, "algorithmic_bytes_per_launch": kern_bytes,
                         "line_pass_ms": scan_ms / pieces, "partition_build_ms": build_ms / pieces,
                         "exchange_gap_ms": stage["gap"] / pieces, "inserts_after_build_ms": stage["post"] / pieces,
                         "stages": {k2: {"kernel": names[k2], "ms": stage_ms[k2],
                                         "algorithmic_bytes": stage_bytes[k2],
                                         "achieved": stage_bytes[k2] / (stage_ms[k2] * 1e-3) / 1e9 if stage_ms[k2] > 0 else 0.0}
                                    for k2 in names if stage_ms[k2] > 0.05},
                         "whole_path": {"algorithmic_bytes": path_bytes, "device_ms": path_ms,
                                        "achieved": path_bytes / (path_ms * 1e-3) / 1e9 if path_ms > 0 else 0.0,
                                        "frac": (path_bytes / (path_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if path_ms > 0 else 0.0}},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.k, args.seed)
        elif world > 1:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    m.close()
    if dist_on:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
