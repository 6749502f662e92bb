#!/usr/bin/env python
"""Headline benchmark: MoCoGAN training clips/sec on synthetic MUG-shape batches (BASELINE.json).

One "step" = one full update_core iteration (D_I, D_V and G forward/backward + three Adam
updates, in-kernel Philox noise) on a per-GPU batch of (B,3,16,64,64) clips already resident in
HBM.  N > 1 ranks (launched by torch.distributed.run) shard the global batch B*N data-parallel
and all-reduce gradients over RCCL; scaling is weak (per-GPU batch fixed).

Prints ONE JSON line (rank 0).  Besides the contract fields it carries
  roofline     : the VideoDiscriminator Conv3d implicit-GEMM kernels (78 % of the step's FLOPs):
                 algorithmic FLOPs per step / summed launch durations measured with HIP events on
                 the launch stream during the timed steps, against the fp32 MFMA peak;
  cpu_baseline : the NumPy restatement of the Chainer-CPU algorithm (oracle/, im2col + BLAS, fp32)
                 timed on this host's cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP32_MFMA_TFLOPS = 157.3          # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32, dense
PEAK_BF16_MFMA_TFLOPS = 2516.8         # MI355X_MICROARCH.md: bf16 MFMA, dense (16x the f32 MFMA rate)
PEAK_F32X3_TFLOPS = PEAK_BF16_MFMA_TFLOPS / 6.0   # 'f32x3': six bf16 products per fp32 product -> 419.5 algorithmic TFLOP/s
HEADLINE_DTYPE = 'f32x3'               # configs[1]: fp32 results; the products on the bf16 pipe (round 4's review allowed it as the headline)
LINE_LIMIT = 4096                      # bytes of the one stdout line (the driver parses it; round 3's 51 KB line was not parsed)
DETAIL_FILE = 'bench_detail.json'      # everything else: per-layer tables, tile choices, full secondary records


def dv_conv_flops_per_clip():
    """Algorithmic FLOPs (2*MAC, true channel counts) of the VideoDiscriminator conv layers dc1..dc4
    for one clip: (forward per call, by layer)."""
    chans = [3, 64, 128, 256, 512]
    t, h = 16, 64
    out = []
    for l in range(4):
        to, ho = t - 3, h // 2
        out.append(2.0 * to * ho * ho * 64 * chans[l] * chans[l + 1])
        t, h = to, ho
    return out


def dv_conv_flops_per_step(batch):
    """What one iteration launches on D_V's strided convolutions: 2 forwards (real, fake); for D's own
    loss wgrad x2 on dc1..4 and dgrad x2 on dc2..4; for G's loss dgrad x1 on dc1..4."""
    f = dv_conv_flops_per_clip()
    fwd = 2 * sum(f)
    wgrad = 2 * sum(f)
    dgrad = 2 * sum(f[1:]) + sum(f)
    return batch * (fwd + wgrad + dgrad), batch * fwd, batch * wgrad, batch * dgrad


def dv_conv_algorithmic_bytes_per_step(batch, dtype='f32'):
    """Bytes the D_V conv family has to move if every tensor crossed HBM exactly once per launch:
    fprop reads x_l and writes y_l; wgrad reads x_l and g_l and writes dW; dgrad reads g_l and writes gx_l.
    fp32 networks: 4 bytes per element; bf16 networks: 2 bytes for every tensor but the 4-channel clip side and dW."""
    chans = [4, 64, 128, 256, 512]
    t, h = 16, 64
    x_b, y_b, w_b = [], [], []
    es = 2.0 if dtype == 'bf16' else 4.0
    for l in range(4):
        to, ho = t - 3, h // 2
        x_b.append((4.0 if l == 0 else es) * t * h * h * chans[l])
        y_b.append(es * to * ho * ho * chans[l + 1])
        w_b.append(4.0 * 64 * chans[l] * chans[l + 1])
        t, h = to, ho
    fprop = 2 * batch * (sum(x_b) + sum(y_b)) + sum(w_b)
    wgrad = 2 * batch * (sum(x_b) + sum(y_b)) + sum(w_b)
    dgrad = 2 * batch * (sum(y_b[1:]) + sum(x_b[1:])) + batch * (sum(y_b) + sum(x_b)) + 2 * sum(w_b)
    if dtype == 'f32x3':
        # the INPUT operands of dc2..dc4's launches are read as three bf16 terms (6 instead of 4 bytes per value); outputs stay fp32
        xin, yin, win = sum(x_b[1:]), sum(y_b[1:]), sum(w_b[1:])
        extra = 0.5 * (2 * batch * xin + win)                       # fprop reads x, w
        extra += 0.5 * (2 * batch * (xin + yin))                    # wgrad reads x, y
        extra += 0.5 * (3 * batch * yin + 2 * win)                  # dgrad (two launches at 2n / n clips) reads y, w
        return fprop + wgrad + dgrad + extra
    return fprop + wgrad + dgrad


def pmc_traffic(batch, dtype='f32'):
    """HBM-side traffic of the D_V conv launches of one step, from the committed rocprofv3 --pmc summary
    (FETCH_SIZE / WRITE_SIZE cannot be read from inside the process; tools/pmc_traffic.py documents the
    collection).  Scaled linearly from the profiled batch.  Returns (bytes_per_step | None, source)."""
    names = ('r05_dv_conv_traffic_bf16.json', 'r04_dv_conv_traffic_bf16.json', 'r03_dv_conv_traffic_bf16.json') if dtype == 'bf16' else \
        ('r06_dv_conv_traffic_f32x3.json', 'r05_dv_conv_traffic_f32x3.json', 'r04_dv_conv_traffic_f32x3.json', 'r03_dv_conv_traffic_f32x3.json') if dtype == 'f32x3' else \
        ('r05_dv_conv_traffic.json', 'r04_dv_conv_traffic.json', 'r03_dv_conv_traffic.json', 'r02_dv_conv_traffic.json', 'r01_dv_conv_traffic.json')
    for name in names:                                                              # newest collection first
        try:
            d = json.load(open(os.path.join(ROOT, 'profiles', name)))
            return d['dv_conv_hbm_bytes_per_step'] * batch / d['batch'], 'profiles/%s (PMC, batch %d)' % (name, d['batch'])
        except Exception:
            continue
    return None, None


def _cpu_time_shape(channels, dim_zl, batch, warmup, steps):
    """median seconds per update_core iteration of the fp32 NumPy port at one input shape"""
    import numpy as np
    from oracle import net as onet, updater as oupd
    rng = np.random.RandomState(0)
    gen = onet.init_generator(rng, dim_zl=dim_zl, out_channels=channels)
    di = onet.init_discriminator(rng, 2, channels, 1)
    dv = onet.init_discriminator(rng, 3, channels, 1)
    og, oi, ov = (oupd.new_adam_state(q) for q in (gen, di, dv))
    x = rng.uniform(-1, 1, (batch, channels, 16, 64, 64)).astype(np.float32)
    t_real = rng.randint(0, dim_zl, batch) if dim_zl else None
    times = []
    for s in range(warmup + steps):
        rnd = oupd.draw_step_randomness(rng, 'normal', batch, channels, dim_zl=dim_zl)
        t0 = time.time()
        oupd.update_core('normal', gen, di, dv, og, oi, ov, x, t_real, rnd, dim_zl=dim_zl)
        if s >= warmup:
            times.append(time.time() - t0)
    times.sort()
    return times[len(times) // 2], times


def cpu_baseline(batch, warmup, steps):
    """the fp32 NumPy port (oracle/: im2col + BLAS sgemm, the algorithm class of the
    Chainer CPU path) timed on this host's cores at batch `batch`, `warmup` untimed + `steps` timed iterations,
    median -- at the MUG shape 16x3x64x64 (C2; `value`) and at the Moving-MNIST shape 16x1x64x64 (C1)."""
    try:
        from threadpoolctl import threadpool_info
        threads = max([i.get('num_threads', 1) for i in threadpool_info()] or [1])
    except Exception:
        threads = os.cpu_count()
    med3, t3 = _cpu_time_shape(3, 6, batch, warmup, steps)
    med1, t1 = _cpu_time_shape(1, 0, batch, warmup, steps)
    return {"value": batch / med3, "unit": "clips/s", "cores": int(threads), "kind": "port", "batch": batch, "timed_iterations": steps,
            "blas_threads": int(threads), "host_cpus": os.cpu_count(),
            "c1_moving_mnist_shape": {"value": batch / med1, "unit": "clips/s", "shape": "16x1x64x64", "batch": batch,
                                      "median_s_per_iteration": med1},
            "sample": "CPU restatement of the Chainer-CPU algorithm (not Chainer itself): update_core at batch %d, full "
                      "width (n_filters=64), fp32 NumPy im2col+BLAS; %d warm-up + %d timed iterations, median "
                      "%.2f s (16x3x64x64, min %.2f max %.2f); the 16x1x64x64 shape under c1_moving_mnist_shape; "
                      "%d BLAS threads on a %d-cpu host"
                      % (batch, warmup, steps, med3, t3[0], t3[-1], int(threads), os.cpu_count())}


def self_launch(n, argv):
    """Parent of `bench.py --gpus N` (N > 1, no WORLD_SIZE): starts N ranks with torch.distributed.run as a CHILD
    process (never exec: nothing here has initialised the GPU, and nothing will), relays the one JSON line of rank 0
    and returns the child's exit code.  Fewer visible GPUs than ranks is an error, not a hang."""
    import socket
    import subprocess
    rehearsal = os.environ.get('MCG_SINGLE_DEVICE') == '1' or os.environ.get('MCG_BENCH_DRYRUN') == '1'
    if not rehearsal:
        import torch
        ndev = torch.cuda.device_count()               # counts devices without initialising HIP
        if ndev < n:
            sys.stderr.write('bench.py --gpus %d: only %d GPU(s) visible on this node\n' % (n, ndev))
            return 2
    with socket.socket() as sock:
        sock.bind(('127.0.0.1', 0))
        port = sock.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')  # RCCL needs dmabuf IPC on this driver
    env.setdefault('OMP_NUM_THREADS', '8')
    limit = float(os.environ.get('MCG_BENCH_LAUNCH_TIMEOUT', '1500'))
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, start_new_session=True, text=True)
    try:
        out, _ = proc.communicate(timeout=limit)
    except subprocess.TimeoutExpired:
        import signal
        os.killpg(proc.pid, signal.SIGKILL)            # exactly the process group started above
        proc.wait()
        sys.stderr.write('bench.py --gpus %d: the ranks did not finish within %.0f s\n' % (n, limit))
        return 3
    lines = [l for l in out.splitlines() if l.startswith('{"metric"')]
    if lines:
        print(lines[-1], flush=True)
    else:
        sys.stdout.write(out)
    if proc.returncode == 0 and not lines:
        sys.stderr.write('bench.py --gpus %d: rank 0 printed no result line\n' % n)
        return 4
    return proc.returncode


def dry_run(args, world, rank):
    """MCG_BENCH_DRYRUN=1 (CPU-container test of the launcher only): rendezvous over gloo, one MAX all-reduce as in
    the timed region's epilogue, rank 0 prints a line marked dry_run -- no kernels run, nothing is measured."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([float(rank + 1)], dtype=torch.float64)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('gloo', rank=rank, world_size=world)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"metric": "training clips/sec (16\u00d73\u00d764\u00d764)", "value": None, "unit": "clips/s",
                          "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "dry_run": True,
                          "max_over_ranks_check": float(t)}))
    return 0


def _round(v, nd=4):
    return round(v, nd) if isinstance(v, float) else v


def compact_line(out, secondary, cpu, detail_path):
    """The ONE stdout line: the contract fields, `config`, `roofline`, `cpu_baseline` and one short record per secondary workload --
    at most LINE_LIMIT bytes.  Per-layer tables, tile choices and the full secondary records live in DETAIL_FILE."""
    rl = out.get("roofline") or {}
    line = {k: _round(out[k]) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "ms_per_step_median",
                                        "higher_is_better", "scaling", "vs_baseline", "dtype", "data") if k in out}
    cfg = out.get("config") or {}
    line["config"] = {k: cfg[k] for k in ("workload", "variant", "per_gpu_batch", "global_batch", "parallelism", "side_streams", "sync_bn",
                                          "input_ready_early") if k in cfg}
    line["roofline"] = {k: _round(rl[k]) for k in ("bound", "achieved", "peak", "unit", "frac", "kernel", "kernel_ms_per_step", "traffic",
                                                   "algorithmic_bytes", "traffic_source", "algorithmic_gflop_per_step") if k in rl}
    if "by_pass" in rl:
        line["roofline"]["tflops_by_pass"] = {k: _round(v["tflops"], 1) for k, v in rl["by_pass"].items()}
    if cpu is not None:
        line["cpu_baseline"] = {"value": _round(cpu["value"]), "unit": cpu["unit"], "cores": cpu["cores"], "kind": cpu["kind"],
                                "sample": "oracle/ (NumPy im2col+BLAS fp32 restatement of the Chainer-CPU algorithm, not Chainer), "
                                          "update_core at batch %d, full width, median of %d timed iterations"
                                          % (cpu.get("batch", 0), cpu.get("timed_iterations", 0))}
    sec = []
    for s_ in secondary:
        if s_ is None:
            continue
        r = {"workload": (s_.get("config") or {}).get("workload", "")[-40:], "dtype": s_.get("dtype")}
        c = s_.get("config") or {}
        if "per_gpu_batch" in c:
            r["workload"] = "%s batch %d/GPU %s" % (c.get("variant"), c["per_gpu_batch"], c["workload"][c["workload"].rfind("(BASELINE"):])
        if "error" in s_:
            r["error"] = s_["error"][:160]
        else:
            r.update(value=_round(s_["value"], 1), ms_per_step=_round(s_["ms_per_step"], 3), ms_median=_round(s_.get("ms_per_step_median"), 3),
                     frac=_round(s_["roofline"]["frac"]), peak=_round(s_["roofline"]["peak"], 1))
        sec.append(r)
    if sec:
        line["secondary"] = sec
    if "dist" in out:
        line["dist"] = {k: out["dist"][k] for k in ("backend", "world_size")}
    line["losses"] = {k: _round(v) for k, v in (out.get("losses") or {}).items()}
    line["detail"] = os.path.basename(detail_path) if detail_path else None
    text = json.dumps(line)
    if len(text) > LINE_LIMIT:                       # never exceed the bound: drop the optional parts, longest first
        for k in sorted((k for k in ("secondary", "losses", "dist") if k in line), key=lambda k: -len(json.dumps(line[k]))):
            line.pop(k, None)
            text = json.dumps(line)
            if len(text) <= LINE_LIMIT:
                break
    if len(text) > LINE_LIMIT:                       # still too long (a string grew): cut the long strings, keep every contract field
        def clip(v):
            return v[:120] if isinstance(v, str) else {k: clip(x) for k, x in v.items()} if isinstance(v, dict) else v
        line = clip(line)
        text = json.dumps(line)
    if len(text) > LINE_LIMIT:                       # last resort: the contract fields and the pointer to the detail file -- a line is ALWAYS printed
        line = {k: line[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                     "vs_baseline", "dtype", "data", "detail") if k in line}
        text = json.dumps(line)
    return text


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50, help='timed iterations (BASELINE.md section 4: >= 50; `value` = all of them over the wall time, '
                                                          'the median step time is reported beside it)')
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=32, help='clips per GPU (BASELINE config C2: 32)')
    ap.add_argument('--model', default='normal', choices=['normal', 'cgan', 'infogan'])
    ap.add_argument('--dtype', default=HEADLINE_DTYPE, choices=['f32', 'bf16', 'f32x3'],
                    help="MFMA operand type of the conv GEMMs.  f32x3 (default, the headline since round 5 -- round 4's review, item 7): "
                         "BASELINE configs[1] with the wide convolutions' fp32 products formed on the bf16 matrix pipe (operands as three "
                         "bf16 terms, six bf16 products per fp32 product, fp32 accumulate: fp32 results, held to the fp32 tolerances by "
                         "the same tests); f32 = the same iteration on the fp32 MFMA (first secondary line); bf16 = configs[2] (use with "
                         "--batch 256), fp32 accumulation / parameters / Adam")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--autotune', type=int, default=1,
                    help='1 (default): the first launch of each conv geometry times the 7 tile candidates once (warm-up)')
    ap.add_argument('--sync-bn', type=int, default=0, help='N > 1 only: 1 = synchronised BatchNorm (not the headline semantics)')
    ap.add_argument('--tiles', default='', help='JSON of tile choices to start from (e.g. for a profiler pass without tuning launches)')
    ap.add_argument('--save-tiles', default='', help='write the tile choices of this run to this JSON file')
    ap.add_argument('--overlap', type=int, default=None,
                    help='1 (default): the headline pass places the ImageDiscriminator update and the weight-gradient GEMMs on side '
                         'HIP streams; 0: one stream throughout.  (Rounds 1-3 defaulted to 0 for bf16 at batch >= 128, where side '
                         'streams lost with the kernels of that time; re-measured in round 4 at batch 256: 12.1 k against 11.6 k '
                         'clips/s with them.)  The roofline pass is always one-stream.')
    ap.add_argument('--secondary', type=int, default=1,
                    help='1 (default): when the headline workload is configs[1] (no --model/--dtype/--batch), also time '
                         'configs[2] (bf16, batch 256) and configs[3] (infogan) -- or, on 8 GPUs, configs[4] (128 clips per GPU) -- '
                         'for a few steps each and report them under "secondary" on the same line')
    ap.add_argument('--secondary-steps', type=int, default=20, help='timed iterations of each secondary workload (20: as the headline; with 8 the fill\n'
                    '                    and drain of the first / last iteration cost the short runs 2-4 %%)')
    ap.add_argument('--cpu-sample-batch', type=int, default=8, help='BASELINE.md section 3: batch 8')
    ap.add_argument('--cpu-sample-steps', type=int, default=5, help='timed iterations (median reported)')
    ap.add_argument('--cpu-sample-warmup', type=int, default=3)
    args = ap.parse_args()
    if args.overlap is None:
        args.overlap = int(os.environ.get('MCG_OVERLAP', '1'))

    if os.environ.get('MCG_DEBUG_HANG'):
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ['MCG_DEBUG_HANG']), exit=True)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world == 1 and args.gpus > 1:
        # `python bench.py --gpus N` without a launcher: this process (which must not touch the GPU) becomes the
        # launcher -- one rank per GPU under torch.distributed.run -- and relays rank 0's line and the exit code.
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))
    if world != args.gpus:
        raise SystemExit('bench.py --gpus %d was started with WORLD_SIZE=%d' % (args.gpus, world))
    if os.environ.get('MCG_BENCH_DRYRUN') == '1':
        raise SystemExit(dry_run(args, world, rank))
    # The contract is ONE line on stdout.  Native libraries write to file descriptor 1 behind Python's back -- RCCL prints a five-line
    # version banner when its first communicator is created (seen in the one-rank rehearsal) -- so until the line is ready everything
    # that goes to descriptor 1 lands on stderr.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')   # (RCCL between processes needs dmabuf IPC on this driver; set before HIP starts)
    import torch
    import torch.distributed as dist
    import mocogan_chainer_amd.hiplib as hl
    import mocogan_chainer_amd.step as mstep

    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the HIP path has no CPU fallback')
    # MCG_SINGLE_DEVICE=1 + MCG_DIST_BACKEND=gloo: rehearse the N > 1 code path on a one-GPU box
    # (every rank on cuda:0, gradients exchanged through gloo); never used for reported numbers.
    rehearsal = os.environ.get('MCG_SINGLE_DEVICE') == '1'
    torch.cuda.set_device(0 if rehearsal else local_rank)
    exchange = None
    # MCG_DP_REHEARSE_NCCL=1 (one GPU, one rank): the data-parallel code path -- process group over nccl (= RCCL), tile-table broadcast,
    # bucketed gradient all-reduce on its stream, the timing collectives -- with a world of ONE; never used for reported numbers.
    dp = world > 1 or os.environ.get('MCG_DP_REHEARSE_NCCL') == '1'
    if dp:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        dist.init_process_group(os.environ.get('MCG_DIST_BACKEND', 'nccl'), rank=rank, world_size=world)
        exchange = mstep.GradExchange(force=world == 1)
    hl.load()
    hl.set_autotune(bool(args.autotune))
    if args.tiles:
        hl.load_tile_choices(args.tiles)

    def barrier():
        if dp:
            dist.barrier()
        torch.cuda.synchronize()

    def measure(model, dtype, B, steps, warmup, overlap):
        """One workload: builds the three networks, runs the roofline pass (one stream, HIP events around every conv
        launch) and the timed headline pass; returns the fields of a result line."""
        if exchange is not None:
            # every rank runs the SAME tile codes: geometries the shipped table does not hold are tuned by one local
            # iteration on throw-away networks, then rank 0's choices are broadcast (a straggler sets the step time)
            mstep.pretune_and_share_tiles(exchange, model, dtype, B, rank)
        gen, di, dv = mstep.make_models(model, num_labels=6, seed=0)           # identical init on every rank
        if exchange is not None:                                                # ... and made identical by construction
            for net in (gen, di, dv):
                exchange.broadcast_params([net.fp.p, net.fp.m, net.fp.v] + list(net.running.values()))
        # input_ready_early: the batch is resident, so the next iteration's real chain may start under the end of this one -- the
        # schedule model.updater.Updater runs too when its iterator stays a batch ahead (trainer.PrefetchIterator hands over the
        # copy's event); MCG_INPUT_EARLY=0 switches it off.  It only matters where the two-chain schedule is on (>= 64 clips).
        early = os.environ.get('MCG_INPUT_EARLY', '1') == '1'
        ts = mstep.TrainStep(model, gen, di, dv, exchange=exchange, seed=1234, rank=rank, precision=dtype, overlap=False,
                              sync_bn=bool(args.sync_bn), input_ready_early=early)
        g = torch.Generator(device='cuda')
        g.manual_seed(rank)
        x_real = torch.rand((B, 3, 16, 64, 64), device='cuda', generator=g) * 2 - 1   # synthetic U(-1,1), resident in HBM
        t_real = torch.randint(0, 6, (B,), device='cuda', dtype=torch.int32, generator=g)
        # Pass 1 (roofline): one stream, every conv launch bracketed by HIP events on that stream -- kernels run
        # alone, so a launch's duration is the kernel's own time (this is what rocprofv3 --stats of
        # `bench.py --overlap 0` reports too).
        for _ in range(warmup):
            ts.run(x_real, t_real)
        barrier()
        hl.timing_begin()
        timing = None
        try:
            t0 = time.perf_counter()
            for _ in range(steps):
                ts.run(x_real, t_real)
            barrier()
            dt_serial_instr = time.perf_counter() - t0
        finally:
            timing = hl.timing_end()                   # (an exception must not leave the event records armed for the next workload)
        # Pass 2 (headline): un-instrumented, EXACTLY `steps` iterations between barrier + synchronize, with the
        # side-stream placement unless --overlap 0.  Same kernels, same arithmetic, same results.
        ts.set_overlap(bool(overlap))
        chains_before = mstep.chain_iterations
        # (untimed: the side streams' first iterations allocate -- blocks handed to another stream return to the caching allocator only
        #  after that stream's work, so the pools take ~10 iterations to settle; with 3 of them the batch-256 line scattered 9.9-12.1 k)
        for _ in range(max(warmup, 10) if overlap else 0):
            ts.run(x_real, t_real)
        barrier()
        # (one event per iteration on the main stream, which joins the side streams at the end of every iteration: no host
        #  synchronisation inside the timed region, the median step time comes from consecutive events afterwards)
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
        t0 = time.perf_counter()
        marks[0].record()
        for i in range(steps):
            ts.run(x_real, t_real)
            marks[i + 1].record()
        barrier()
        dt_local = time.perf_counter() - t0
        step_ms = sorted(marks[i].elapsed_time(marks[i + 1]) for i in range(steps))
        ms_median = step_ms[len(step_ms) // 2]
        tall = torch.tensor([dt_local], device='cuda', dtype=torch.float64)
        per_rank = [dt_local]
        if dp:
            gathered = [torch.zeros_like(tall) for _ in range(world)]
            dist.all_gather(gathered, tall)
            per_rank = [float(t) for t in gathered]
            dist.all_reduce(tall, op=dist.ReduceOp.MAX)
        dt_best = float(tall)
        losses = ts.losses()
        early_used = bool(early and overlap and mstep.chain_iterations > chains_before)
        del ts, gen, di, dv, x_real
        torch.cuda.empty_cache()
        if rank != 0:
            return None
        ms_per_step = dt_best / steps * 1e3
        value = B * world * steps / dt_best
        tot, f_fwd, f_wg, f_dg = dv_conv_flops_per_step(B)
        dv_ms = {k: timing.get('D_V.' + k, (0, 0.0)) for k in ('fprop', 'wgrad', 'dgrad')}
        dv_split_ms = timing.get('D_V.split', (0, 0.0))[1] / steps      # 'f32x3': the passes that split the GEMM operands count as GEMM time
        dv_total_ms = sum(v[1] for v in dv_ms.values()) / steps + dv_split_ms
        achieved = tot / (dv_total_ms * 1e-3) / 1e12 if dv_total_ms > 0 else 0.0
        kern = {}
        for k, fl in (('fprop', f_fwd), ('wgrad', f_wg), ('dgrad', f_dg)):
            n_l, ms = dv_ms[k]
            kern[k] = {"launches_per_step": n_l / steps, "ms_per_step": ms / steps,
                       "tflops": fl / (ms / steps * 1e-3) / 1e12 if ms > 0 else 0.0}
        all_conv_ms = sum(v[1] for k, v in timing.items() if ' N=' not in k) / steps
        by_layer = {}
        for k, (n_l, ms) in sorted(timing.items()):
            if ' N=' not in k:
                continue
            f = dict(kv.split('=') for kv in k.split()[1:])
            N_, T_, H_, Ci_, Co_ = (int(f[x]) for x in ('N', 'T', 'H', 'Ci', 'Co'))
            kt_ = 4 if T_ > 1 else 1
            gflop = 2.0 * N_ * (T_ - kt_ + 1) * (H_ // 2) ** 2 * kt_ * 16 * min(Ci_, 3 if Ci_ == 4 else Ci_) * Co_ / 1e9
            by_layer[k] = {"launches_per_step": n_l / steps, "ms_per_launch": ms / n_l,
                           "tflops": gflop * n_l / ms if ms > 0 else 0.0}
        traffic, traffic_src = pmc_traffic(B, dtype)
        # f32x3 executes six bf16 MFMA products per algorithmic fp32 product: its roof in ALGORITHMIC TFLOP/s is the dense bf16 peak / 6
        peak = PEAK_BF16_MFMA_TFLOPS if dtype == 'bf16' else PEAK_F32X3_TFLOPS if dtype == 'f32x3' else PEAK_FP32_MFMA_TFLOPS
        cfg_name = "configs[2]" if (dtype == 'bf16' and B == 256 and model == 'normal') else \
            "configs[1]" if (dtype == 'f32' and B == 32 and model == 'normal') else \
            "configs[1], fp32 products on the bf16 pipe" if (dtype == 'f32x3' and B == 32 and model == 'normal') else \
            "configs[3]" if (dtype == 'f32' and B == 32 and model == 'infogan') else \
            "configs[3], fp32 products on the bf16 pipe" if (dtype == 'f32x3' and B == 32 and model == 'infogan') else \
            "configs[4]" if (B == 128 and world == 8 and model == 'normal') else "off-list variant of configs[1]"
        per_rank_ms = sorted(t / steps * 1e3 for t in per_rank)
        return {
            "metric": "training clips/sec (16\u00d73\u00d764\u00d764)", "value": value, "unit": "clips/s", "n_gpus": world,
            "steps": steps, "warmup": warmup, "ms_per_step": ms_per_step, "ms_per_step_median": ms_median,
            "step_ms_min_max": [step_ms[0], step_ms[-1]], "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic" + (" (single-device rehearsal)" if rehearsal else ""),
            "config": {"workload": "MUG-shape synthetic (B,3,16,64,64) U(-1,1), one update_core iteration per step "
                                   "(BASELINE.json %s)" % cfg_name, "variant": model,
                       "per_gpu_batch": B, "global_batch": B * world, "n_filters": 64, "dim_zl": 6,
                       "parallelism": "dp%d" % world, "side_streams": bool(overlap), "input_ready_early": early_used,
                       "sync_bn": bool(args.sync_bn)},
            "roofline": {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s",
                         "frac": achieved / peak, "dv_conv3d_mfma_util_pct": 100.0 * achieved / peak, "traffic": traffic,
                         "algorithmic_bytes": dv_conv_algorithmic_bytes_per_step(B, dtype),
                         "traffic_unit": "bytes per step (memory side of L2, incl. Infinity-Cache hits)",
                         "traffic_source": traffic_src,
                         "traffic_measured_in_run": False,      # PMC counters need rocprofv3 around the process: see traffic_source
                         "kernel": "VideoDiscriminator Conv3d implicit-GEMM family (%s<FpropP|DgradP|WgradP>), "
                                   % ("gemm_kernel" if dtype == 'f32' else "gemm_bf16_v2_kernel SPLIT / gemm_kernel" if dtype == 'f32x3' else "gemm_bf16_kernel") +
                                   "dc1..dc4, all launches of one step",
                         "algorithmic_gflop_per_step": tot / 1e9, "kernel_ms_per_step": dv_total_ms, "by_pass": kern,
                         **({"operand_split_ms_per_step": dv_split_ms,
                             "note": "fp32 products on the bf16 matrix pipe: operands as three bf16 terms, six bf16 MFMA products per "
                                     "fp32 product (DESIGN.md); `peak` = dense bf16 MFMA peak / 6 in algorithmic TFLOP/s",
                             "x_fp32_mfma_peak": achieved / PEAK_FP32_MFMA_TFLOPS} if dtype == 'f32x3' else {}),
                         "all_conv_kernels_ms_per_step": all_conv_ms,
                         "by_network_and_pass_ms_per_step": {k: v[1] / steps for k, v in sorted(timing.items()) if ' N=' not in k},
                         "by_layer": by_layer,
                         "measured": "HIP events around every launch of the family during %d one-stream iterations "
                                     "(%.3f ms/step with the event records); the headline pass %s"
                                     % (steps, dt_serial_instr / steps * 1e3,
                                        "overlaps independent kernels on side streams, which stretches individual launches"
                                        if overlap else "is one-stream too"),
                         "dv_conv_share_of_step_time": dv_total_ms / (dt_serial_instr / steps * 1e3)},
            "dist": {"backend": dist.get_backend() if dp else None, "world_size": dist.get_world_size() if dp else 1,
                     "per_rank_ms_per_step": {"min": per_rank_ms[0], "median": per_rank_ms[len(per_rank_ms) // 2],
                                              "max": per_rank_ms[-1]}},
            "losses": losses,
        }

    def default_overlap(dtype, B):
        return int(os.environ.get('MCG_OVERLAP', '1'))          # (rounds 1-3: off for bf16 at batch >= 128; re-measured in round 4: on is +4 %)

    def all_ranks_ok(ok):
        """MIN over the ranks of a local success flag (one small all-reduce every rank reaches)"""
        if not dp:
            return ok
        f = torch.tensor([1.0 if ok else 0.0], device='cuda')
        dist.all_reduce(f, op=dist.ReduceOp.MIN)
        return bool(f.item() > 0.5)

    def preflight(model, dtype, B):
        """N > 1 only: ONE local iteration of the workload (no collective inside) on throw-away networks.  A rank that cannot run
        it (out of memory, a geometry the library refuses) says so HERE, where every rank still reaches the flag's all-reduce --
        inside measure() the other ranks would be left waiting in RCCL."""
        ok = True
        try:
            gen, di, dv = mstep.make_models(model, num_labels=6, seed=0)
            ts = mstep.TrainStep(model, gen, di, dv, seed=0, rank=rank, precision=dtype)
            ts.run(torch.zeros((B, 3, 16, 64, 64), device='cuda'), torch.zeros(B, dtype=torch.int32, device='cuda'))
            torch.cuda.synchronize()
            del ts, gen, di, dv
        except Exception as exc:                                   # noqa: BLE001
            sys.stderr.write('bench.py rank %d: preflight of %s %s batch %d failed: %r\n' % (rank, model, dtype, B, exc))
            ok = False
        torch.cuda.empty_cache()
        return all_ranks_ok(ok)

    out = measure(args.model, args.dtype, args.batch, args.steps, args.warmup, args.overlap)
    # The other single-GPU workloads BASELINE.json names ride on the same line (a few steps each): configs[2] (bf16
    # networks, batch 256) and configs[3] (--model infogan, batch 32); on 8 GPUs configs[4] (global batch 1024).
    # MCG_BENCH_REHEARSAL_BATCH (single-device rehearsals only, tests/test_gpu_dp.py): the per-rank batch that stands in for the
    # headline's 32 and the 8-GPU secondaries' 128, so that a one-GPU box can walk the `world == 8` branch below in seconds
    reh_b = int(os.environ.get('MCG_BENCH_REHEARSAL_BATCH', '0')) if rehearsal else 0
    headline_cfg = args.model == 'normal' and args.dtype == HEADLINE_DTYPE and args.batch == (reh_b or 32)
    sec_b = reh_b or 128
    secondary = []

    def also(model, dtype, B):
        """a secondary workload must never cost the headline its line.  One GPU: a failure is recorded in its place.  N > 1: the
        workload only starts when every rank has run it once locally (preflight); after that an exception is a real bug and is
        raised -- swallowing it on one rank would leave the others in a collective."""
        if dp:
            if not preflight(model, dtype, B):
                if rank == 0:
                    secondary.append({"config": {"workload": "%s %s batch %d" % (model, dtype, B)}, "dtype": dtype,
                                      "error": "preflight failed on at least one rank (see stderr)"})
                return
            secondary.append(measure(model, dtype, B, args.secondary_steps, 3, default_overlap(dtype, B)))
            return
        try:
            secondary.append(measure(model, dtype, B, args.secondary_steps, 3, default_overlap(dtype, B)))
        except Exception as exc:                                   # noqa: BLE001
            secondary.append({"config": {"workload": "%s %s batch %d" % (model, dtype, B)}, "dtype": dtype, "error": repr(exc)[:300]})
    if args.secondary and headline_cfg:
        other = 'f32' if HEADLINE_DTYPE == 'f32x3' else 'f32x3'
        if world == 1:
            also('normal', other, 32)          # configs[1] again in the other form of its fp32 arithmetic (fp32 MFMA / bf16 pipe): first
            also('normal', 'bf16', 256)
            also('infogan', HEADLINE_DTYPE, 32)
        elif world == 8 or os.environ.get('MCG_BENCH_SECONDARY_DP') == '1':
            also('normal', HEADLINE_DTYPE, sec_b)
            also('normal', other, sec_b)
    if rank == 0 and args.save_tiles:
        hl.save_tile_choices(args.save_tiles)
    if rank == 0:
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(args.cpu_sample_batch, args.cpu_sample_warmup, args.cpu_sample_steps)
        tiles = {"%s N=%d T=%d H=%d Ci=%d Co=%d p=%d" % (k[0], k[1], k[2], k[3], k[5], k[6], k[9]): v
                 for k, v in sorted(hl.tile_choices().items(), key=str)}
        detail = dict(out, secondary=[s_ for s_ in secondary if s_ is not None], tile_choices=tiles, cpu_baseline=cpu)
        detail_path = os.environ.get('MCG_BENCH_DETAIL', os.path.join(ROOT, DETAIL_FILE))
        try:
            with open(detail_path, 'w') as f:
                json.dump(detail, f, indent=1)
        except OSError as exc:
            sys.stderr.write('bench.py: could not write %s: %r\n' % (detail_path, exc))
            detail_path = None
        sys.stdout.flush()
        os.dup2(real_stdout, 1)
        print(compact_line(out, secondary, cpu, detail_path), flush=True)
        os.dup2(2, 1)
    if dp:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
