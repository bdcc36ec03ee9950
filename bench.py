#!/usr/bin/env python3
"""Training-throughput benchmark of the Dense_U_Net_lidar hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1 without a launcher: starts the N ranks itself)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One "step" = forward + per-pixel BCE + metrics + backward + (DP: gradient all-reduce) + Adam on one synthetic
minibatch that is already resident in HBM.  Rank 0 prints ONE JSON line (Measurement).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

DENSENETS = {121: (32, (6, 12, 24, 16), 64), 169: (32, (6, 12, 32, 32), 64), 201: (32, (6, 12, 48, 32), 64)}
# BASELINE.json configs.  C2 (configs[1]) is the 1-GPU configuration the metric is quoted on.
CONFIGS = {
    "c1": dict(depth=121, cbb=1, s2=0, batch=1, H=256, W=384, dtype="fp32", name="C1 d121 no-fusion 1x3x256x384"),
    "c2": dict(depth=121, cbb=1, s2=3, batch=4, H=1280, W=1920, dtype="fp16", name="C2 d121 early-fusion 6ch b4 1280x1920"),
    "c3": dict(depth=121, cbb=3, s2=3, batch=4, H=1280, W=1920, dtype="fp16", name="C3 d121 mid-fusion(3) b4/GPU 1280x1920"),
    "c4": dict(depth=169, cbb=3, s2=3, batch=2, H=1280, W=1920, dtype="fp16", name="C4 d169 mid-fusion(3) b2/GPU 1280x1920"),
    "c5": dict(depth=201, cbb=3, s2=3, batch=8, H=640, W=960, dtype="bf16", name="C5 d201 mid-fusion(3) b8/GPU 640x960 mixed bf16"),
}
PEAK_MFMA_TFLOPS = {"fp16": 2500.0, "bf16": 2500.0, "fp32": 157.3}  # dense, MI355X_MICROARCH.md
DTYPE_LABEL = {"fp16": "f16", "bf16": "bf16", "fp32": "f32"}
EVENT_PASSES = 3
PEAK_HBM_GBS = 8000.0


def metric_label(c):
    """BASELINE.json's metric for the 1280x1920 RGB+LiDAR configs; the other configs say what they measured."""
    inputs = "RGB+LiDAR" if c["s2"] else "RGB"
    return f"training images/sec at {c['H']}x{c['W']} {inputs}"


def make_config(c):
    from dmmfods_amd.utils.Dense_U_Net_lidar_helper import get_config
    cfg = get_config("/tmp/dmmfods_bench")
    k, bc, nif = DENSENETS[c["depth"]]
    cfg.model.growth_rate, cfg.model.block_config, cfg.model.num_init_features = k, bc, nif
    cfg.model.concat_before_block_num, cfg.model.stream_2_in_channels = c["cbb"], c["s2"]
    return cfg


def synthetic_batch(c, device, seed):
    """RGB ~ U[0,255]; LiDAR ~90 % exact zeros, rest U[0,255]; targets (U > 0.9)."""
    g = torch.Generator(device=device).manual_seed(seed)
    B, H, W = c["batch"], c["H"], c["W"]
    rgb = torch.rand(B, 3, H, W, device=device, generator=g) * 255.0
    s2 = max(c["s2"], 1)
    lidar = torch.rand(B, s2, H, W, device=device, generator=g) * 255.0
    lidar = lidar * (torch.rand(B, s2, H, W, device=device, generator=g) > 0.9)
    tgt = (torch.rand(B, 3, H, W, device=device, generator=g) > 0.9).float()
    return rgb, lidar, tgt


ENC_1X1 = __import__("re").compile(r"/(f|s2)\.(b\d+\.l\d+\.conv1|transition\d+\.conv)$|/concat_module\.conv$")


def collect_profile(model, plan, K, ops_out=None, enc=None):
    """Per-class sums of the per-launch event times; `enc` (a dict) receives the forward time / FLOPs of the DenseNet encoder's
    1x1 convolutions (dense-layer bottlenecks, transitions, the fusion module): north_star's MFMA target is quoted on them."""
    from dmmfods_amd import _lib
    L = _lib.lib()
    classes = {}
    for which in (0, 1):
        n = L.dmm_plan_profile_num_ops(plan.handle, which)
        ms = (C.c_double * n)()
        passes = C.c_int()
        _lib.check(L.dmm_plan_profile_collect(plan.handle, which, ms, n, C.byref(passes)))
        if passes.value == 0:
            continue
        for i in range(n):
            label, fl, by = C.c_char_p(), C.c_double(), C.c_double()
            L.dmm_plan_profile_op(plan.handle, which, i, C.byref(label), C.byref(fl), C.byref(by))
            cls = (label.value or b"").decode().split("/")[0] or "other"
            if enc is not None and which == 0 and ENC_1X1.search((label.value or b"").decode()):
                enc["ms"] = enc.get("ms", 0.0) + ms[i] / passes.value
                enc["flops"] = enc.get("flops", 0.0) + fl.value
            if ops_out is not None and ms[i] > 0:
                ops_out.append((ms[i] / passes.value, (label.value or b"").decode(), fl.value, by.value))
            e = classes.setdefault(cls, dict(ms=0.0, launches=0, flops=0.0, bytes=0.0))
            e["ms"] += ms[i]
            e["launches"] += passes.value
            e["flops"] += fl.value * passes.value
            e["bytes"] += by.value * passes.value
    return classes


def dominant_class(classes):
    """The kernel class with the largest share of the serial pass.  "other" (unlabelled housekeeping launches: finalize kernels,
    memsets) is not a kernel and cannot be bracketed by the class filter, so it never qualifies."""
    named = {k: v for k, v in classes.items() if k != "other"} or classes
    return max(named.items(), key=lambda kv: kv[1]["ms"])


def _source_stamp():
    from tools.src_hash import source_hash
    return source_hash()


def measured_traffic(cls, workload):
    """HBM bytes per launch of a kernel class from the committed rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE, see
    tools/pmc_traffic.py); bench.py cannot run the profiler around itself, so the figure comes from profiles/ and is only
    used when it was collected on the same workload.  Returns (bytes per launch, file, PMC class, stale): `stale` when the file
    carries no source stamp or one of another build (tools/src_hash.py) - the figure is then reported under another key."""
    import glob
    import json as _json
    for path in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "*", "*pmc_hbm_traffic.json")), reverse=True):
        try:
            d = _json.load(open(path))
        except Exception:  # noqa: BLE001
            continue
        if (d.get("config"), d.get("batch"), d.get("dtype")) != workload:
            continue
        # rocprof sees kernel names, not the plan's labels: the generic weight-gradient kernel (normal and transposed form) is one
        # class there ("wgrad*.nN"); it stands for the plan's "wgrad.nN" when the launch counts of the step agree
        e = d.get("classes", {}).get(cls) or d.get("classes", {}).get(cls.replace("wgrad.", "wgrad*."))
        pmc_cls = cls
        if e is None and cls.startswith("bw1."):   # one kernel, labelled by the padded input width in the plan: the PMC class is "bw1"
            e, pmc_cls = d.get("classes", {}).get("bw1"), "bw1 (all launches of the kernel: the profiler sees the kernel name, not the plan's n128 / n64 label)"
        if e:
            return (e["traffic_bytes_per_launch"], os.path.relpath(path, os.path.dirname(os.path.abspath(__file__))), pmc_cls,
                    d.get("source_sha16") != _source_stamp())
    return None


def in_step_kernel_sum(workload):
    """Sum of all kernel durations per step, both streams, from the committed rocprofv3 --kernel-trace --stats summary of this
    workload (profiles/*/<config>_b<batch>_kernel_stats.json, written by tools/profile_round.sh next to the CSV; bench.py cannot
    wrap itself in the profiler).  Returns (ms per step, file, stale) - see measured_traffic."""
    import glob
    import json as _json
    here = os.path.dirname(os.path.abspath(__file__))
    for path in sorted(glob.glob(os.path.join(here, "profiles", "*", "*kernel_stats.json")), reverse=True):
        try:
            d = _json.load(open(path))
        except Exception:  # noqa: BLE001
            continue
        if (d.get("config"), d.get("batch"), d.get("dtype")) == workload and d.get("steps"):
            return round(d["total_kernel_ms"] / d["steps"], 3), os.path.relpath(path, here), d.get("source_sha16") != _source_stamp()
    return None


def roofline_block(classes, dtype, workload=None, timed=None):
    if not classes:
        return None, None
    peak_f = PEAK_MFMA_TFLOPS[dtype] * 1e12
    peak_b = PEAK_HBM_GBS * 1e9
    tot_ms = sum(e["ms"] for e in classes.values())
    ideal_ms = 0.0
    table = []
    for cls, e in classes.items():
        t = e["ms"] / 1e3
        tf, tb = e["flops"] / peak_f, e["bytes"] / peak_b
        ideal_ms += max(tf, tb) * 1e3
        table.append(dict(kernel=cls, ms_total=round(e["ms"], 3), launches=e["launches"],
                          share=round(e["ms"] / tot_ms, 4) if tot_ms else 0,
                          tflops=round(e["flops"] / t / 1e12, 2) if t else 0, gbs=round(e["bytes"] / t / 1e9, 1) if t else 0,
                          bound="mfma" if tf > tb else "hbm", frac=round(max(tf, tb) / t, 4) if t else 0))
    table.sort(key=lambda r: -r["ms_total"])
    cls, e = dominant_class(classes)
    e_full = e
    alone_ms = e["ms"] / max(e["launches"], 1)
    if timed and cls in timed and timed[cls]["launches"] and timed[cls]["ms"] > 0:
        e = timed[cls]  # the same launches, bracketed inside the timed region (other streams running beside them)
    t = e["ms"] / 1e3
    tf, tb = e["flops"] / peak_f, e["bytes"] / peak_b
    if tf > tb:
        roof = dict(bound="mfma", achieved=round(e["flops"] / t / 1e12, 3), peak=PEAK_MFMA_TFLOPS[dtype], unit="TFLOP/s")
    else:
        roof = dict(bound="hbm", achieved=round(e["bytes"] / t / 1e9, 2), peak=PEAK_HBM_GBS, unit="GB/s")
    roof["frac"] = round(roof["achieved"] / roof["peak"], 4)
    roof["traffic"] = None
    tr = measured_traffic(cls, workload)
    roof["traffic"] = None
    if tr is not None:
        # a PMC summary taken on ANOTHER build of the library is not this run's traffic: quoted under its own key, `traffic` stays null
        roof["traffic_of_another_build" if tr[3] else "traffic"] = tr[0]
        roof["traffic_source"], roof["traffic_class"] = tr[1], tr[2]
    roof["kernel"] = cls
    roof["avg_launch_ms"] = round(e["ms"] / max(e["launches"], 1), 4)
    roof["avg_launch_ms_alone"] = round(alone_ms, 4)
    # the same class in the serial pass (no other stream beside it): the kernel's own efficiency
    roof["frac_alone"] = round(roof["frac"] * (e["ms"] / max(e["launches"], 1)) / alone_ms, 4) if alone_ms else None
    roof["event_launches"] = e["launches"]
    roof["alg_flops_per_launch"] = e["flops"] / max(e["launches"], 1)
    roof["alg_bytes_per_launch"] = e["bytes"] / max(e["launches"], 1)
    roof["share_of_step"] = round(e_full["ms"] / tot_ms, 4) if tot_ms else None
    roof["per_layer_roofline_frac_of_step"] = round(ideal_ms / tot_ms, 4) if tot_ms else None
    return roof, table


def cpu_baseline(min_seconds=10.0, max_steps=60):
    """The CPU oracle (oracle/restatement.py, kind 'port') timed on this host: BASELINE.json configs[0] (C1)."""
    from oracle import restatement as R
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    # the GPU box grants a 16-core share per GPU whatever os.cpu_count() says; oversubscribing makes torch crawl
    ncores = max(1, min(avail, int(os.environ.get("DMM_CPU_THREADS", "16"))))
    torch.set_num_threads(ncores)
    print(f"[bench] cpu_baseline: oracle on {ncores} threads ...", file=sys.stderr, flush=True)
    arch = R.densenet_arch(121, concat_before_block_num=1, stream_2_in_channels=0)
    P = R.make_state(arch, seed=123)
    tr = R.Trainer(arch, P)
    rgb, lidar, tgt = R.make_inputs(arch, 1, 256, 384, seed=0)
    t_start = time.time()
    tr.step(rgb, lidar, tgt)  # warm-up
    print(f"[bench] cpu_baseline: warm-up step {time.time() - t_start:.2f} s", file=sys.stderr, flush=True)
    times = []
    t_timed = time.time()
    while len(times) < max_steps and (len(times) < 5 or time.time() - t_timed < min_seconds):  # a bounded ~10 s sample
        t0 = time.time()
        tr.step(rgb, lidar, tgt)
        times.append(time.time() - t0)
    best = min(times)
    median = sorted(times)[len(times) // 2]
    return dict(value=round(1.0 / best, 4), unit="img/s", cores=torch.get_num_threads(), cpu_model=cpu_model(), kind="port",
                sample=f"C1 d121 no-fusion 1x3x256x384 fp32 fwd+BCE+bwd+Adam, {len(times)} steps ({sum(times):.1f} s) after 1 warm-up, best {best:.3f} / median {median:.3f} s/step "
                       f"(= {109.8 / best:.1f} conv GFLOP/s); the GPU workload is 25.4x more conv FLOPs per image")


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start the N rank processes ourselves, as a child
    `python -m torch.distributed.run`, BEFORE this process touches the GPU, and pass its exit code on."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    print("[bench] no RANK in the environment: " + " ".join(cmd), file=sys.stderr, flush=True)
    return subprocess.call(cmd)


def conv0_dgrad_gflop(c):
    """a training step is 3x the forward conv FLOPs minus the data gradient of the stem convolution(s), which is never
    needed (2 * 64 * Cin * 49 * H/2 * W/2 per image and stem)."""
    stems = [(3 + c["s2"]) if c["cbb"] == 1 else 3] + ([c["s2"]] if c["cbb"] > 1 else [])
    return sum(2.0 * 64 * cin * 49 * (c["H"] // 2) * (c["W"] // 2) for cin in stems) / 1e9


def percentile(xs, q):
    xs = sorted(xs)
    if not xs:
        return None
    k = (len(xs) - 1) * q
    lo, hi = int(k), min(int(k) + 1, len(xs) - 1)
    return xs[lo] + (xs[hi] - xs[lo]) * (k - lo)


class Workload:
    """One BASELINE configuration resident on this rank: model, optimizer, synthetic batch, and the training step."""

    def __init__(self, c, device, rank, distributed, force):
        from dmmfods_amd.graphs.models.Dense_U_Net_lidar import Dense_U_Net_lidar
        from dmmfods_amd.optim import FusedAdam
        from dmmfods_amd.parallel import GradAllReduce, broadcast_parameters
        self.c = c
        torch.manual_seed(123)  # identical weights on every rank (reference agent seed, H:179)
        self.model = Dense_U_Net_lidar(make_config(c), compute_dtype=c["dtype"]).to(device).train()
        self.opt = FusedAdam(self.model)
        self.reducer = GradAllReduce(self.model, force=force) if distributed else None
        if distributed:
            broadcast_parameters(self.model, src=0, force=force)   # every rank starts from rank 0's weights and running statistics
        self.rgb, self.lidar, self.tgt = synthetic_batch(c, device, seed=rank)
        self.overlap_comm = not os.environ.get("DMM_NO_COMM_OVERLAP")
        self.tail_events = None   # [(backward enqueued, collectives joined)] event pairs of the timed steps (N > 1)

    def step(self):
        from dmmfods_amd.parallel import GradAllReduce
        m = self.model
        with torch.no_grad():
            m(self.rgb, self.lidar)
        met = m.loss_backward(self.tgt)      # enqueues the whole backward; returns before the GPU has run it
        if self.reducer is not None:
            if self.tail_events is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()                  # = the end of this rank's backward on the compute stream
            if self.overlap_comm:            # one all-reduce per gradient bucket, each behind the bucket's readiness event
                GradAllReduce.wait(self.reducer.reduce_overlapped())
            else:
                self.reducer.all_reduce()
            if self.tail_events is not None:
                e1.record()                  # = the compute stream has joined the last collective: e1 - e0 = un-overlapped tail
                self.tail_events.append((e0, e1))
        self.opt.step()
        return met

    @property
    def plan(self):
        return self.model._last[0]


def timed_region(w, steps, distributed, device):
    """EXACTLY `steps` steps between barrier + synchronize on both sides; per-step durations from events recorded on the
    compute stream between steps (no host synchronisation inside the region).  Returns (elapsed s, max over ranks; per-step ms)."""
    import torch.distributed as dist
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(steps + 1)]
    if distributed:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks[0].record()
    met = None
    for i in range(steps):
        met = w.step()
        marks[i + 1].record()
    torch.cuda.synchronize()
    if distributed:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if distributed:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = t.item()
    per_step = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]
    return elapsed, per_step, met


def step_stats(per_step):
    return {"median": round(percentile(per_step, 0.5), 3), "p10": round(percentile(per_step, 0.1), 3),
            "p90": round(percentile(per_step, 0.9), 3), "n": len(per_step)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)    # >= 10 warm-up + >= 30 timed steps
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default="c2", choices=sorted(CONFIGS))
    ap.add_argument("--also", default=None, choices=sorted(CONFIGS) + ["none"],
                    help="N > 1 only: a second workload measured after the main one and reported under 'also' (default c3, BASELINE's 8-GPU d121 configuration)")
    ap.add_argument("--dtype", default=None, choices=["fp16", "bf16", "fp32"])
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true")
    ap.add_argument("--table", action="store_true", help="also print the per-kernel-class table to stderr")
    ap.add_argument("--ops", type=int, default=0, help="print the N most expensive launches (per step) to stderr")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ and not os.environ.get("DMM_FORCE_DIST"):
        sys.exit(spawn_ranks(args.gpus))

    # Native libraries write to file descriptor 1 (RCCL prints its version banner there when the first communicator comes up).
    # The contract is ONE JSON line on stdout, so everything else that lands on fd 1 goes to stderr and the line is written to
    # the saved descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE is {world}: refusing to report a {world}-rank run as {args.gpus} GPUs", file=sys.stderr)
        sys.exit(2)
    distributed = world > 1 or bool(os.environ.get("DMM_FORCE_DIST"))  # the second form exercises the N>1 code path on one GPU
    # Rehearsal of the N > 1 code path on a ONE-GPU box (tests/test_dp_gpu.py): DMM_DIST_BACKEND=gloo DMM_DIST_SAME_DEVICE=1 puts
    # every rank on cuda:0 and exchanges through gloo (RCCL refuses two ranks on one device).  Never the measured configuration.
    backend = os.environ.get("DMM_DIST_BACKEND", "nccl")
    dev_index = 0 if os.environ.get("DMM_DIST_SAME_DEVICE") else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:   # (only the single-process DMM_FORCE_DIST form gets here without one)
            os.environ["MASTER_PORT"] = str(free_port())
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from dmmfods_amd import _lib

    c = dict(CONFIGS[args.config])
    if args.dtype:
        c["dtype"] = args.dtype
    if args.batch:
        c["batch"] = args.batch
    force = world == 1 and distributed  # DMM_FORCE_DIST: the N > 1 code path, collectives included, on one GPU
    w = Workload(c, device, rank, distributed, force)
    step = w.step

    for _ in range(args.warmup):
        step()
    plan = w.plan
    L = _lib.lib()
    full_classes, ops_list, dom_prefix = None, None, None
    if not args.no_profile:
        # (1) one untimed pass with an event pair around EVERY launch, serialised on one stream: the per-class table and the
        #     choice of the dominant kernel class; (2) the timed region brackets only that class (two event records per
        #     selected launch), so the step being timed runs as in production.
        _lib.check(L.dmm_plan_profile_filter(plan.handle, None))
        _lib.check(L.dmm_plan_profile_begin(plan.handle, 1))
        step()
        torch.cuda.synchronize()
        ops_list = [] if args.ops else None
        enc = {}
        full_classes = collect_profile(w.model, plan, 1, ops_list, enc)
        if full_classes and rank == 0:
            dom_prefix = (dominant_class(full_classes)[0] + "/").encode()
        if distributed:
            obj = [dom_prefix]
            dist.broadcast_object_list(obj, src=0)
            dom_prefix = obj[0]
        _lib.check(L.dmm_plan_profile_filter(plan.handle, dom_prefix))
        # the event pairs are recorded in the first EVENT_PASSES steps of the timed region only (each pair costs ~5 us of stream time)
        _lib.check(L.dmm_plan_profile_begin(plan.handle, min(EVENT_PASSES, args.steps)))
    if distributed:
        w.tail_events = []
    elapsed, per_step, met = timed_region(w, args.steps, distributed, device)
    loss = met["loss_per_class"].sum().item()
    comm = None
    if distributed and w.tail_events:
        tails = [a.elapsed_time(b) for a, b in w.tail_events]
        bk = w.model.grad_buckets()
        comm = {"backend": backend, "rccl_world": world if backend == "nccl" else None, "overlap": bool(w.overlap_comm),
                "buckets": len(bk), "bucket_mb": [round(n * 4 / 2 ** 20, 1) for _, n in bk],
                "all_reduce_mb_per_step": round(sum(n for _, n in bk) * 4 / 2 ** 20, 1),
                # time the compute stream spent waiting for collectives after ITS OWN backward had finished (rank 0): the part of
                # the exchange that backward did not hide, including the wait for slower ranks
                "unoverlapped_tail_ms": step_stats(tails)}
    w.tail_events = None

    # The value of the weight-gradient stream as a number (world 1): the same step with every launch on ONE stream, right behind
    # the timed region.  (dmm_set_option("overlap_wgrad") is read at every launch list; the plan is not rebuilt.)
    schedule = None
    timed_classes = None
    if rank == 0 and not args.no_profile and full_classes:
        timed_classes = collect_profile(None, plan, args.steps)       # (before anything else runs on the plan)
    if world == 1 and not args.no_profile and not os.environ.get("DMM_NO_OVERLAP"):
        _lib.check(L.dmm_plan_profile_begin(plan.handle, 0))          # no event pairs in these steps
        _lib.check(L.dmm_set_option(b"overlap_wgrad", 0))
        try:
            for _ in range(2):
                w.step()
            el1, ps1, _ = timed_region(w, 6, False, device)
        finally:
            _lib.check(L.dmm_set_option(b"overlap_wgrad", 1))
        schedule = {"two_stream_ms_per_step": round(elapsed / args.steps * 1e3, 3), "single_stream_ms_per_step": round(el1 / 6 * 1e3, 3),
                    "single_stream_steps": 6}

    also = None
    also_name = args.also or ("c3" if world > 1 and args.config == "c2" else "none")
    if world > 1 and also_name != "none" and also_name != args.config:
        # BASELINE's multi-GPU configurations are the mid-fusion networks; `value` stays on the 1-GPU workload so that the 1 -> N
        # series is one weak-scaling curve, and the named 8-GPU configuration is measured right behind it on the same ranks
        c2nd = dict(CONFIGS[also_name])
        del w
        torch.cuda.empty_cache()
        w2 = Workload(c2nd, device, rank, distributed, force)
        for _ in range(args.warmup):
            w2.step()
        w2.tail_events = []
        el2, ps2, _ = timed_region(w2, args.steps, distributed, device)
        tails2 = [a.elapsed_time(b) for a, b in w2.tail_events]
        also = {"workload": c2nd["name"], "dtype": DTYPE_LABEL[c2nd["dtype"]], "value": round(c2nd["batch"] * world * args.steps / el2, 3),
                "unit": "img/s", "ms_per_step": round(el2 / args.steps * 1e3, 3), "step_ms": step_stats(ps2),
                "per_gpu_batch": c2nd["batch"], "unoverlapped_tail_ms": step_stats(tails2)}

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = c["batch"] * world * args.steps / elapsed
        roof, table = (None, None)
        if not args.no_profile and full_classes:
            timed = timed_classes
            roof, table = roofline_block(full_classes, c["dtype"], (args.config, c["batch"], DTYPE_LABEL[c["dtype"]]), timed)
            if ops_list:
                ops_list.sort(reverse=True)
                for ms_, lab, fl, by in ops_list[:args.ops]:
                    print(f"{ms_:9.3f} ms  {lab:48s} {fl / ms_ / 1e9 if ms_ else 0:8.1f} TF/s {by / ms_ / 1e6 if ms_ else 0:8.1f} GB/s", file=sys.stderr)
        fwd_flops_img = plan.flops_forward / c["batch"]
        train_flops_img = 3 * fwd_flops_img - conv0_dgrad_gflop(c) * 1e9   # SURVEY 8(d) convention
        out = {
            "metric": metric_label(c),
            "value": round(value, 3), "unit": "img/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE_LABEL[c["dtype"]], "data": "synthetic",
            "config": {"workload": c["name"], "key": args.config, "per_gpu_batch": c["batch"], "global_batch": c["batch"] * world,
                       "height": c["H"], "width": c["W"], "storage_dtype": c["dtype"], "accumulate": "fp32",
                       "parallelism": f"dp{world}", "weights": "random-init (reference init, seed 123)",
                       "fwd_conv_gflop_per_img": round(fwd_flops_img / 1e9, 1),
                       "train_conv_gflop_per_img": round(train_flops_img / 1e9, 1)},
            "step_ms": step_stats(per_step),          # per-step durations (HIP events between steps on the compute stream)
            "final_loss_sum": loss,
            "achieved_conv_tflops": round(train_flops_img * value / 1e12, 2),
            # MFMA-only fraction of the step = (sum of conv FLOPs / dense MFMA peak) / measured step time
            "mfma_only_frac_of_step": round(train_flops_img * value / 1e12 / (world * PEAK_MFMA_TFLOPS[c["dtype"]]), 4),
            "roofline": roof,
        }
        if schedule is not None:
            # serial sum: the per-launch event times of the untimed one-stream pass (every launch carries ~4 us of event-pair
            # overhead); in-step sum: rocprofv3 --kernel-trace --stats of this command, total kernel time / steps, both streams
            if full_classes:
                schedule["serial_kernel_sum_ms"] = round(sum(e["ms"] for e in full_classes.values()), 3)
            ins = in_step_kernel_sum((args.config, c["batch"], DTYPE_LABEL[c["dtype"]]))
            if ins is not None:   # (from a committed rocprofv3 summary; of another build of the library: under its own key)
                schedule["in_step_kernel_sum_ms_of_another_build" if ins[2] else "in_step_kernel_sum_ms"] = ins[0]
                schedule["in_step_source"] = ins[1]
            out["schedule"] = schedule
        if comm is not None:
            out["comm"] = comm
        if also is not None:
            out["also"] = [also]
        if not args.no_profile and full_classes and enc.get("ms"):
            # forward time of the encoder's 1x1 convolutions in the serial per-launch pass vs the dense MFMA peak
            out["encoder_1x1"] = {"fwd_ms": round(enc["ms"], 3), "fwd_gflop": round(enc["flops"] / 1e9, 1),
                                  "tflops": round(enc["flops"] / enc["ms"] / 1e9, 1),
                                  "mfma_frac": round(enc["flops"] / enc["ms"] / 1e9 / PEAK_MFMA_TFLOPS[c["dtype"]], 4)}
        if table and args.table:
            for r in table:
                print(json.dumps(r), file=sys.stderr)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        else:
            out["cpu_baseline"] = None
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
