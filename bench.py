#!/usr/bin/env python3
"""bench.py -- the headline metric of BASELINE.json on MI355X:
input GB/s, compress + decompress, level 5, batched 128 KB strings.

One step = one pass of the hot path over one device-resident batch:
CompressDeflate(level 5) of every stream, then DecompressDeflate of every result.
Inputs are generated on the device (synthetic G-text) and are resident in HBM
before the timed region; nothing crosses PCIe inside it.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--streams S] [--level L]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Prints ONE JSON line on rank 0.  `roofline` is computed live for the dominant kernel of the
timed region from HIP events on the library's stream; `cpu_baseline` is the CPU oracle
(a C port of the reference algorithm, the reference itself being Lua and no Lua VM present)
on a bounded sample of the same batch.
"""
import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

STREAM_LEN = 131072
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8 TB/s spec
SEED = 0x4C69624465666C61


def gen_text_device(torch, count, length, device, seed):
    """G-text on the device: Zipf(1.1) words over the 4096-word vocabulary of synth.py,
    punctuation every 60-80 bytes.  Returns uint8 tensor (count*length)."""
    import numpy as np
    from libdeflate_amd import synth
    wl, wt = synth._vocab()
    nw = len(wl)
    table = np.full((nw, 11), 32, dtype=np.uint8)
    mask = np.zeros((nw, 11), dtype=bool)
    for i in range(nw):
        table[i, : wl[i]] = wt[i, : wl[i]]
        mask[i, : wl[i] + 1] = True
    g = torch.Generator(device=device)
    g.manual_seed(seed & 0x7FFFFFFFFFFFFFFF)
    p = torch.arange(1, nw + 1, dtype=torch.float64, device=device) ** -1.1
    cdf = torch.cumsum(p / p.sum(), 0).to(torch.float32)
    t_table = torch.from_numpy(table).to(device)
    t_mask = torch.from_numpy(mask).to(device)
    total = count * length
    out = torch.empty(total, dtype=torch.uint8, device=device)
    filled = 0
    chunk_words = 1 << 24
    while filled < total:
        u = torch.rand(chunk_words, device=device, generator=g)
        idx = torch.searchsorted(cdf, u).clamp_(max=nw - 1)
        flat = t_table[idx][t_mask[idx]]
        take = min(flat.numel(), total - filled)
        out[filled: filled + take] = flat[:take]
        filled += take
    # sentence / line breaks
    nbreak = total // 60 + 2
    gaps = torch.randint(60, 81, (nbreak,), device=device, generator=g)
    pos = torch.cumsum(gaps, 0)
    pos = pos[pos < total]
    punct = torch.tensor(list(b".,;\n\n"), dtype=torch.uint8, device=device)
    out[pos] = punct[torch.randint(0, 5, (pos.numel(),), device=device, generator=g)]
    return out


def cpu_baseline(host_rows, level, seconds_target=12.0):
    """The CPU oracle (C port of the reference's algorithm) on the host cores, round trip,
    over a bounded sample of the batch.  Returns dict for the JSON line."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import ldtest as T
    # every host core this process may use: the affinity mask, capped by the share the box grants one GPU
    # (OMP_NUM_THREADS = 16 there, while os.cpu_count() shows the whole 8-GPU machine)
    visible = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    share = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    cores = max(1, min(visible, share) if share > 0 else visible)
    rows = [r.tobytes() for r in host_rows]

    def work(s):
        c, _ = T.o_compress(s, level=level)
        d, left = T.o_decompress(c, cap=len(s))
        assert d == s and left == 0
        return len(s)

    # calibrate on one stream, then size the sample to ~seconds_target of wall time
    t0 = time.time()
    work(rows[0])
    per = max(time.time() - t0, 1e-4)
    nsample = int(max(cores, min(len(rows), seconds_target * cores / per)))
    sample = rows[:nsample]
    t0 = time.time()
    with ThreadPoolExecutor(max_workers=cores) as ex:  # ctypes releases the GIL inside the oracle
        nbytes = sum(ex.map(work, sample))
    dt = time.time() - t0
    return {"value": nbytes / dt / 1e9, "unit": "GB/s", "cores": cores, "host_cpus_visible": visible, "kind": "port",
            "sample": "%d of the batch's 128 KiB streams, compress(level %d)+decompress by oracle/ld_oracle.c, "
                      "%d threads (all cores granted to this GPU: OMP_NUM_THREADS=%s of %d visible), %.1f s; the reference itself "
                      "(Lua) cannot run here: LuaJIT baseline unavailable, its published figure is 4.1 MB/s per core"
                      % (nsample, level, cores, os.environ.get("OMP_NUM_THREADS", "unset"), visible, dt)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--streams", type=int, default=8192, help="streams per GPU (each 131072 B)")
    ap.add_argument("--stream-bytes", type=int, default=STREAM_LEN, help="bytes per stream (default 131072; 4096 = configs[1] shape)")
    ap.add_argument("--level", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-levels", action="store_true", help="skip the one-step runs at levels 1 and 8")
    ap.add_argument("--no-extra-rows", action="store_true",
                    help="skip the secondary entries (configs[1], configs[3], C5, host API) measured after the headline")
    ap.add_argument("--rendezvous-only", action="store_true",
                    help="launcher check (CPU, gloo): start the ranks, run the barrier / MAX-over-ranks protocol around "
                         "empty steps and print the line with value null; no GPU work")
    args = ap.parse_args()

    # shard.py is loaded by path: importing the package would dlopen the HIP runtime and libdeflate_hip.so, and the
    # launcher below must stay a process that has loaded nothing of the GPU stack
    import importlib.util
    spec = importlib.util.spec_from_file_location("ldh_shard", os.path.join(ROOT, "libdeflate_amd", "shard.py"))
    shard = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(shard)
    rank, local_rank, world = shard.env_world()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N`: this process becomes the launcher of N rank processes (one per GPU).  It has
        # made no GPU call (neither torch nor the library is loaded) and makes none: it waits, stops the other ranks
        # when one fails, and hands on the exit code.  Under a profiler the profiler's preloaded library has already
        # initialised the GPU in this process, and starting children from it is the launcher hop this pool forbids.
        preload = os.environ.get("LD_PRELOAD", "")
        if any(k.startswith(("ROCP_", "ROCPROF")) for k in os.environ) or "rocprof" in preload:
            sys.exit("bench.py: --gpus %d under a profiler: profile ONE rank instead (python bench.py --gpus 1, or "
                     "set RANK/LOCAL_RANK/WORLD_SIZE=1 yourself)" % args.gpus)
        sys.exit(shard.launch_ranks([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], args.gpus,
                                    timeout=float(os.environ.get("LDH_LAUNCH_TIMEOUT", "3000"))))
    if args.gpus != world:
        sys.exit("bench.py: --gpus %d but WORLD_SIZE is %d" % (args.gpus, world))
    if args.rendezvous_only:
        import torch
        dist = shard.init("gloo")
        dt = shard.timed_steps(lambda: None, args.steps, args.warmup, lambda: None, dist, torch, "cpu")
        ok = shard.all_true(True, dist, torch, "cpu")
        if rank == 0:
            print(json.dumps({"metric": "rendezvous only", "value": None, "n_gpus": world, "steps": args.steps,
                              "warmup": args.warmup, "ms_per_step": dt * 1e3 / max(args.steps, 1), "verified": ok}), flush=True)
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        return

    import numpy as np
    import torch
    # one rank per GPU over RCCL ("nccl").  LDH_BENCH_BACKEND=gloo is the rehearsal of the N > 1 path on a box with
    # fewer GPUs than ranks: ranks share devices (LOCAL_RANK modulo the device count) and the rendezvous runs on CPU
    # tensors; everything else is the same code.
    backend = os.environ.get("LDH_BENCH_BACKEND", "nccl")
    ndev = max(torch.cuda.device_count(), 1)
    if backend == "nccl" and local_rank >= ndev:
        sys.exit("bench.py: rank %d has no GPU (%d visible)" % (local_rank, ndev))
    local_rank %= ndev
    dist = shard.init(backend, torch.device("cuda", local_rank))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    rdev = dev if backend == "nccl" else torch.device("cpu")  # where the rendezvous tensors live

    from libdeflate_amd import _native as N
    from libdeflate_amd import batch as B
    N.init(local_rank)

    S, L = args.streams, args.stream_bytes
    data = gen_text_device(torch, S, L, dev, SEED ^ rank)
    off = torch.arange(S + 1, dtype=torch.int64, device=dev) * L
    bt = B.DeviceBatch(data, off, L)
    stride = N.lib.ldh_compress_bound(L)
    comp = torch.empty(S * stride, dtype=torch.uint8, device=dev)
    comp_len = torch.zeros(S, dtype=torch.int32, device=dev)
    pad = torch.zeros(S, dtype=torch.uint8, device=dev)
    back = torch.empty(S * L, dtype=torch.uint8, device=dev)
    back_len = torch.zeros(S, dtype=torch.int32, device=dev)
    left = torch.zeros(S, dtype=torch.int32, device=dev)
    status = torch.zeros(S, dtype=torch.int32, device=dev)

    def step():
        B.device_compress(bt, comp, stride, comp_len, pad, fmt=0, level=args.level)
        B.device_decompress_slots(comp, stride, comp_len, S, back, L, back_len, left, status, fmt=0)

    def dev_sync():
        B.sync()
        torch.cuda.synchronize()

    # the library enqueues on its own stream: torch's kernels that produced `data`, `off` and the fills above must
    # have completed before the first *_dev call reads them
    torch.cuda.synchronize()
    # untimed warm-up, then the profiled + timed steps (barrier + sync on both sides, MAX over ranks)
    shard.timed_steps(step, 0, args.warmup, dev_sync, dist, torch, rdev)
    B.profile(True)
    B.profile_reset()
    dt = shard.timed_steps(step, args.steps, 0, dev_sync, dist, torch, rdev)
    prof = B.profile_report()
    B.profile(False)

    # SURVEY 8.d also wants levels 1 and 8 reported: one verified step each, outside the timed region
    other = {}
    if not args.no_other_levels:
        for lv in (1, 8):
            if lv == args.level:
                continue
            B.profile(True)
            B.profile_reset()
            B.device_compress(bt, comp, stride, comp_len, pad, fmt=0, level=lv)
            B.device_decompress_slots(comp, stride, comp_len, S, back, L, back_len, left, status, fmt=0)
            dev_sync()
            pr = B.profile_report()
            B.profile(False)
            t_c = sum(v[0] for k, v in pr.items() if k != "k_inflate")
            t_d = sum(v[0] for k, v in pr.items() if k == "k_inflate")
            good = bool(torch.equal(back, data)) and int(status.abs().sum()) == 0
            if rank == 0:  # and, like the headline, four streams' bytes against the oracle (a valid but different
                # parse would pass the round trip)
                sys.path.insert(0, os.path.join(ROOT, "tests"))
                import ldtest as T
                lens4 = comp_len[:4].cpu().numpy()
                cbuf4 = comp[: 4 * stride].cpu().numpy()
                rows4 = data[: 4 * L].view(4, L).cpu().numpy()
                for i in range(4):
                    want, _ = T.o_compress(rows4[i].tobytes(), level=lv)
                    good = good and cbuf4[i * stride: i * stride + int(lens4[i])].tobytes() == want
            other[str(lv)] = {"compress_GBps": S * L / 1e6 / max(t_c, 1e-9), "decompress_GBps": S * L / 1e6 / max(t_d, 1e-9),
                              "roundtrip_GBps": S * L / 1e6 / max(t_c + t_d, 1e-9),
                              "compress_ratio": S * L / max(int(comp_len.sum().item()), 1), "verified": good}
        # leave the buffers as the timed level produced them (the checks below read them)
        step()
        dev_sync()

    # correctness inside the same run: exact round trip of every stream, oracle bytes on a sample
    ok = bool(torch.equal(back, data)) and int(status.abs().sum()) == 0 and int(left.sum()) == 0
    comp_bytes = int(comp_len.sum().item())
    nsamp = 4
    host_rows = data[: max(nsamp, 4096) * L].view(-1, L)[: 4096 if not args.no_cpu_baseline else nsamp].cpu().numpy()
    if rank == 0:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import ldtest as T
        lens = comp_len[:nsamp].cpu().numpy()
        cbuf = comp[: nsamp * stride].cpu().numpy()
        for i in range(nsamp):
            want, _ = T.o_compress(host_rows[i].tobytes(), level=args.level)
            ok = ok and cbuf[i * stride: i * stride + int(lens[i])].tobytes() == want
    ok = shard.all_true(ok, dist, torch, rdev)

    extra = {}
    if rank == 0 and world == 1 and not args.no_extra_rows and L == STREAM_LEN:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_rows as R
        del comp, back  # (the rows allocate their own buffers)
        extra["configs1_small_strings"] = R.row_small_strings(torch, dev, SEED, T=T)
        extra["configs3_zlib_dictionary"] = R.row_zlib_dictionary(
            torch, dev, data, min(S, 2048), L, os.path.join(ROOT, "tests", "golden", "dictionary32768.txt"), level=args.level, T=T)
        extra["configs4_mixed_inflate_print"] = R.row_mixed_inflate_print(torch, dev, data, S, L)
        extra["host_api_pcie_inclusive"] = R.row_host_api(torch, data, min(S, 2048), L, level=args.level)
        ok = ok and all(v.get("verified", True) if "levels" not in v else all(x["verified"] for x in v["levels"].values())
                        for v in extra.values())
    if rank == 0:
        in_bytes = S * L * args.steps * world
        value = in_bytes / dt / 1e9
        # dominant kernel of the timed region
        name, (ms, launches) = max(prof.items(), key=lambda kv: kv[1][0])
        compress_kernels = ("k_sort_links", "k_seg_links", "k_match_lazy", "k_parse_lazy", "k_lz_greedy", "k_tokens_literal",
                            "k_histogram", "k_huffman", "k_plan", "k_bitpack")
        # algorithmic bytes per launch: compress N_in + N_out, decompress N_comp + N_out
        alg = S * L + comp_bytes
        avg_s = ms / 1e3 / max(launches, 1)
        achieved = alg / avg_s / 1e9
        # HBM-side bytes per launch of that kernel: PMC passes (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE) of this
        # same command, committed under profiles/ (a process cannot collect its own counters); null for any
        # other workload than the one they were taken on
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                pt = json.load(f)
            if pt.get("workload") == "%dx%d level %d" % (S, L, args.level):
                traffic = pt["kernels"][name]["total"]
        except (OSError, KeyError, ValueError):
            traffic = None
        line = {
            "metric": "input GB/s compress+decompress, level %d, batched %d KB strings" % (args.level, L // 1024),
            "value": value, "unit": "GB/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u8", "data": "synthetic",
            "config": {"workload": "%d x %d B G-text streams per GPU (SURVEY 8.d), CompressDeflate level %d + "
                                   "DecompressDeflate, device-resident" % (S, L, args.level),
                       "streams_per_gpu": S, "stream_bytes": L, "level": args.level, "seed": hex(SEED),
                       "compress_ratio": S * L / max(comp_bytes, 1), "verified": ok,
                       "other_levels_one_step_kernel_time": other,
                       "phase_ms_per_step": {k: v[0] / args.steps for k, v in sorted(prof.items())},
                       "compress_GBps": S * L * args.steps / 1e6 / max(sum(v[0] for k, v in prof.items() if k in compress_kernels), 1e-9),
                       "decompress_GBps": S * L * args.steps / 1e6 / max(sum(v[0] for k, v in prof.items() if k not in compress_kernels), 1e-9)},
            "roofline": {"bound": "hbm", "kernel": name, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg, "avg_launch_ms": avg_s * 1e3},
        }
        if extra:
            line["other_configs"] = extra
        if not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(host_rows, args.level)
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        sys.exit(3)


if __name__ == "__main__":
    main()
