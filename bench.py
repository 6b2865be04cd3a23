#!/usr/bin/env python3
"""bench.py -- BASELINE config 2: SegManaged(ConvNeXt-T + ASPP), 512x512 crop, bf16 compute, 16 images per GPU, full training
step (forward + backward + SyncBN / gradient all-reduce over RCCL + fused AdamW + running mIoU), synthetic data.

  python bench.py --gpus N --steps K --warmup W        (N>1: launched by torch.distributed.run, one rank per GPU)

Rank 0 prints ONE JSON line (see the driver contract); `roofline` describes the dominant kernel measured live with HIP
events on the launch stream, `cpu_baseline` is the CPU oracle (a "port": the reference itself needs TensorFlow) timed on
this box's host cores on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0       # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16 MFMA
TRAIN_GFLOP_PER_IMAGE = 148.6  # 24.77 GMAC fwd x 2 x 3 (OS32)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--fp32", action="store_true", help="fp32 storage (parity mode); the headline number is bf16")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--graph-step", action="store_true", help="(default on one GPU since round 4; kept for old command lines)")
    ap.add_argument("--eager-step", action="store_true", help="enqueue every launch of the timed steps from the host instead of replaying the step from one HIP graph")
    ap.add_argument("--roofline-steps", type=int, default=3, help="eager steps in front of the timed region whose dominant-GEMM launches carry HIP event pairs")
    ap.add_argument("--probe-native", action="store_true",
                    help="(internal) the probe job of choose_exchange(): two eager + two replayed steps with the stream-ordered RCCL exchange")
    ap.add_argument("--check-launch", action="store_true",
                    help="rendezvous + one all-reduce only (gloo when there is no GPU): tests the --gpus N self-launch path on CPU")
    return ap.parse_args()


def visible_gpu_count():
    """GPUs this process may use, WITHOUT a HIP / torch.cuda call (the launcher parent must not hold the device): the visibility list when
    one is exported, else the kfd topology (a node with SIMDs is a GPU; CPU nodes report simd_count 0).  None when neither source exists."""
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            return len([t for t in v.split(",") if t.strip() != ""])
    root = "/sys/class/kfd/kfd/topology/nodes"
    if not os.path.isdir(root):
        return None
    n = 0
    for node in os.listdir(root):
        try:
            with open(os.path.join(root, node, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
        except (OSError, ValueError):
            continue
    return n


def _cpulist(text):
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out += list(range(int(a), int(b or a) + 1))
    return out


def gpu_numa_nodes():
    """NUMA node of every GPU in kfd order, from sysfs only (no HIP call): kfd topology node -> PCI address (domain, location_id) ->
    /sys/bus/pci/devices/<bdf>/numa_node.  [] when the topology is not readable (containers without /sys/class/kfd)."""
    root = "/sys/class/kfd/kfd/topology/nodes"
    out = []
    try:
        nodes = sorted(os.listdir(root), key=int)
    except (OSError, ValueError):
        return out
    for node in nodes:
        try:
            with open(os.path.join(root, node, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) <= 0:
                continue
            loc, dom = int(props.get("location_id", "0")), int(props.get("domain", "0"))
            bdf = f"{dom:04x}:{(loc >> 8) & 0xff:02x}:{(loc >> 3) & 0x1f:02x}.{loc & 7:x}"
            with open(f"/sys/bus/pci/devices/{bdf}/numa_node") as f:
                out.append(int(f.read().strip()))
        except (OSError, ValueError):
            out.append(-1)
    return out


def pin_to_gpu_numa(local_rank, ranks_on_node):
    """Pin this rank to cores of its GPU's NUMA node (a contiguous share when several ranks sit on one node).  In the c10d + eager fallback every
    rank's host thread enqueues ~7 ms of launches per 8 ms step: a thread that migrates across sockets, or eight ranks packed on one socket,
    starves the GPUs.  Affinity of THIS process only, set before it touches the GPU (never an exec); ISEG_BENCH_PIN=0 switches it off.  Returns a
    short description for the JSON line, or None when nothing was changed."""
    if os.environ.get("ISEG_BENCH_PIN", "1") == "0" or not hasattr(os, "sched_setaffinity") or ranks_on_node <= 1:
        return None      # (one rank: nothing competes for the cores, and the CPU baseline wants all of them)
    try:
        allowed = sorted(os.sched_getaffinity(0))
        numa = gpu_numa_nodes()
        vis = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")
        order = [int(t) for t in vis.split(",") if t.strip().isdigit()] if vis else list(range(len(numa)))
        gpu = order[local_rank] if local_rank < len(order) else local_rank
        node = numa[gpu] if gpu < len(numa) else -1
        cores = allowed
        if node >= 0:
            with open(f"/sys/devices/system/node/node{node}/cpulist") as f:
                on_node = [c for c in _cpulist(f.read()) if c in set(allowed)]
            if on_node:
                cores = on_node
        # ranks that share the node (or the whole allowed set) take equal contiguous shares, at least two cores each
        peers = [r for r in range(ranks_on_node) if (numa[order[r]] if r < len(order) and order[r] < len(numa) else -1) == node] or [local_rank]
        share = max(2, len(cores) // max(1, len(peers)))
        k = peers.index(local_rank) if local_rank in peers else 0
        mine = cores[k * share:(k + 1) * share] or cores
        os.sched_setaffinity(0, mine)
        return f"gpu {gpu} numa {node}: cores {mine[0]}-{mine[-1]} ({len(mine)})"
    except (OSError, ValueError, IndexError) as e:
        print(f"bench.py: NUMA pinning skipped ({type(e).__name__}: {e})", file=sys.stderr)
        return None


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves.  This parent never touches the GPU (no HIP call, no
    torch.cuda.*: the device count comes from the visibility list / kfd topology), the children are fresh interpreters with RANK /
    LOCAL_RANK / WORLD_SIZE / MASTER_* set (never an exec of a process that initialised the GPU).  Rank 0 inherits stdout and prints the JSON
    line; every rank's stderr (and the other ranks' stdout) goes to its own log file, whose tail is shown when a rank fails.  A wall-clock
    watchdog (ISEG_BENCH_TIMEOUT_S, default 900) kills the ranks that are left and exits non-zero, so a hung rank cannot hang the caller;
    a rank that dies takes the others down after a short grace period instead of leaving them in a collective."""
    import socket
    import subprocess
    import tempfile

    n = args.gpus
    have = visible_gpu_count()
    if not args.check_launch and have is not None and have < n:
        print(f"bench.py: --gpus {n} but this node exposes {have} GPU(s)", file=sys.stderr)
        return 2
    sock, sock2 = socket.socket(), socket.socket()      # two free ports: the job's rendezvous and the exchange probe's (choose_exchange)
    sock.bind(("127.0.0.1", 0))
    sock2.bind(("127.0.0.1", 0))
    port, probe_port = sock.getsockname()[1], sock2.getsockname()[1]
    sock.close()
    sock2.close()
    log_dir = os.environ.get("ISEG_BENCH_LOG_DIR") or tempfile.mkdtemp(prefix="iseg_bench_")
    os.makedirs(log_dir, exist_ok=True)
    limit = float(os.environ.get("ISEG_BENCH_TIMEOUT_S", "900"))
    procs, logs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   ISEG_BENCH_PROBE_PORT=os.environ.get("ISEG_BENCH_PROBE_PORT", str(probe_port)),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        log = open(os.path.join(log_dir, f"rank{r}.log"), "wb")
        logs.append(log)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else log, stderr=log))
    t0 = time.time()
    failed_at = None
    verdict = 0
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            break
        now = time.time()
        if any(c not in (None, 0) for c in codes) and failed_at is None:
            failed_at = now
        hung = now - t0 > limit
        if hung or (failed_at is not None and now - failed_at > 20.0):
            why = f"no result after {limit:.0f} s" if hung else "a rank failed and the others did not exit within 20 s"
            print(f"bench.py: {why}; terminating the remaining ranks (the children started here, by pid)", file=sys.stderr)
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            deadline = time.time() + 10.0
            for p in procs:
                try:
                    p.wait(timeout=max(0.1, deadline - time.time()))
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
            verdict = 4 if hung else 1
            break
        time.sleep(0.2)
    for log in logs:
        log.close()
    codes = [p.returncode for p in procs]
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad or verdict:
        print(f"bench.py: ranks failed (rank, exit code): {bad}; per-rank logs in {log_dir}", file=sys.stderr)
        for r, _ in bad[:4]:
            try:
                with open(os.path.join(log_dir, f"rank{r}.log"), "rb") as f:
                    tail = f.read()[-1500:].decode("utf-8", "replace")
                print(f"---- rank {r} (tail) ----\n{tail}", file=sys.stderr)
            except OSError:
                pass
        return verdict or 1
    return 0


def joined_ranks(device):
    """the number of ranks that actually take part in collectives: an all-reduce of ones, not the environment's word for it"""
    from iseg_amd import dist

    one = torch.ones(1, dtype=torch.float32, device=device)
    dist.all_reduce_sum(one)
    return int(round(float(one.item())))


def check_launch(args):
    from iseg_amd import dist

    affinity = pin_to_gpu_numa(int(os.environ.get("LOCAL_RANK", "0")), max(args.gpus, 1))
    use_gpu = (visible_gpu_count() or 0) >= max(args.gpus, 1) and torch.cuda.is_available()
    if os.environ.get("ISEG_BENCH_TEST_HANG") == str(os.environ.get("RANK", "0")):      # (tests/test_bench_launch.py: the watchdog)
        time.sleep(3600)
    if os.environ.get("ISEG_BENCH_TEST_FAIL") == str(os.environ.get("RANK", "0")):
        sys.exit(7)
    # (the launch check probes only when a test scripts the probe's outcome: its other cases are about the launcher itself)
    exchange = choose_exchange(args) if (use_gpu or os.environ.get("ISEG_BENCH_TEST_PROBE")) else "not probed"
    dist.init(backend=None if use_gpu else "gloo")
    dev = torch.device("cuda", dist.local_rank()) if use_gpu else torch.device("cpu")
    if os.environ.get("ISEG_BENCH_TEST_SPLIT_DECISION") == str(os.environ.get("RANK", "0")):      # (tests: this rank's parent lost its probe to the limit)
        os.environ["ISEG_DIST_NATIVE"] = "0"
        exchange = "c10d work objects (test: split decision)"
    exchange = agree_exchange(exchange, dev)
    n = joined_ranks(dev)
    if n != args.gpus:
        print(f"bench.py: {n} ranks joined, --gpus {args.gpus}", file=sys.stderr)
        sys.exit(3)
    dist.barrier()
    if dist.rank() == 0:
        print(json.dumps({"metric": "launch_check", "n_gpus": n, "backend": torch.distributed.get_backend() if n > 1 else "none",
                          "exchange": exchange, "host_affinity": affinity}))
    if dist.is_initialized():
        torch.distributed.destroy_process_group()


PROBE_MARK = "ISEG_PROBE_NATIVE_OK"


def probe_port():
    """rendezvous port of the probe job, the same on every rank and known to be free when it was chosen: (1) ISEG_BENCH_PROBE_PORT -- bench.py's own
    launcher binds a second free port and exports it; (2) under torch.distributed.run the agent's TCP store (MASTER_ADDR:MASTER_PORT, no GPU
    involved): rank 0 asks the kernel for a free port and publishes it, the others read it; (3) MASTER_PORT + 23 when neither exists."""
    p = os.environ.get("ISEG_BENCH_PROBE_PORT")
    if p:
        return int(p)
    base = int(os.environ.get("MASTER_PORT", "29500"))
    if os.environ.get("TORCHELASTIC_USE_AGENT_STORE") == "True":
        try:
            import datetime
            import socket

            store = torch.distributed.TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), base, is_master=False,
                                               timeout=datetime.timedelta(seconds=60))
            key = "iseg_bench_probe_port_" + os.environ.get("TORCHELASTIC_RESTART_COUNT", "0")
            if os.environ.get("RANK", "0") == "0":
                sock = socket.socket()
                sock.bind(("", 0))
                free = sock.getsockname()[1]
                sock.close()
                store.set(key, str(free))
            return int(store.get(key))
        except Exception as e:      # noqa: BLE001 -- any store problem: the documented fallback
            print(f"bench.py: probe port through the agent store failed ({type(e).__name__}: {e}); using MASTER_PORT + 23", file=sys.stderr)
    return base + 23


def agree_exchange(exchange, device):
    """The ranks decided alone (choose_exchange: each from its own probe child, under its own wall-clock limit), so start-up skew around the limit
    could leave one rank with the marker and another with a killed child -- a mixed native / c10d job hangs in its first collective.  Once the
    process group is up and BEFORE the first SyncBN message or gradient bucket: all-reduce(MIN) of the local decision over c10d; every rank then
    runs what the slowest one can run.  (iseg_amd.dist reads ISEG_DIST_NATIVE at every collective, so setting it here is in time.)"""
    if not torch.distributed.is_available() or not torch.distributed.is_initialized() or torch.distributed.get_world_size() <= 1:
        return exchange
    mine = 1.0 if os.environ.get("ISEG_DIST_NATIVE", "0") not in ("0", "") else 0.0
    flag = torch.tensor([mine], dtype=torch.float32, device=device)
    torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
    agreed = float(flag.item())
    if agreed != mine:      # this rank's probe passed, another rank's did not
        os.environ["ISEG_DIST_NATIVE"] = "0"
        print(f"bench.py: rank {os.environ.get('RANK', '?')}: another rank's probe did not pass; all ranks take c10d + eager step", file=sys.stderr)
        return "c10d work objects (another rank's native-exchange probe did not pass: agreed by all-reduce(MIN) of the ranks' decisions)"
    return exchange


def choose_exchange(args):
    """Data-parallel runs (N > 1): which exchange does the measured job use?  The fast one -- SyncBN messages and gradient buckets enqueued on
    explicit HIP streams through the C ABI's own RCCL communicator (ISEG_DIST_NATIVE=1), which makes the whole step one HIP graph -- has never met
    two ranks on hardware, and a mismatch inside a replayed graph hangs without a watchdog.  So every rank, BEFORE it touches the GPU, starts a
    fresh child (same RANK / WORLD_SIZE, its own rendezvous port) that runs the probe job: two eager and two replayed steps with that exchange,
    then a c10d all-reduce(MIN) of the ranks' success flags, then the marker line.  The marker -- a value every rank computed from the same
    all-reduce -- decides, not the child's exit code, so all ranks take the same branch: marker seen -> native + graph replay; anything else
    (crash, timeout: the child is killed by pid after ISEG_BENCH_PROBE_TIMEOUT_S, default 240 s) -> c10d work objects + eager step.
    An explicit ISEG_DIST_NATIVE in the environment is the user's decision and skips the probe."""
    import subprocess

    if args.gpus <= 1:
        return "none (one rank)"
    if os.environ.get("ISEG_DIST_NATIVE") is not None:
        return "stream-ordered RCCL through the C ABI (ISEG_DIST_NATIVE set by the caller)" if os.environ["ISEG_DIST_NATIVE"] not in ("0", "") \
            else "c10d work objects (ISEG_DIST_NATIVE=0 set by the caller)"
    limit = float(os.environ.get("ISEG_BENCH_PROBE_TIMEOUT_S", "240"))
    port = probe_port()
    env = dict(os.environ, ISEG_DIST_NATIVE="1", MASTER_PORT=str(port))
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", str(args.gpus), "--probe-native", "--batch", str(args.batch), "--size", str(args.size)]
    if args.fp32:      # the probe must exercise the exchange the measured job runs (fp32 storage changes every message and bucket)
        cmd.append("--fp32")
    if args.check_launch:
        cmd.append("--check-launch")
    try:
        child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    except OSError as e:
        print(f"bench.py: probe could not start ({e})", file=sys.stderr)
        return "c10d work objects (probe could not start)"
    import signal

    def _term(signum, frame):      # the launcher's watchdog terminates this rank: take the probe child along (by pid)
        child.kill()
        sys.exit(143)

    prev = signal.signal(signal.SIGTERM, _term)
    try:
        out, _ = child.communicate(timeout=limit)
        why = f"exit code {child.returncode}"
    except subprocess.TimeoutExpired:
        child.kill()      # (the child this rank started, by pid)
        out, _ = child.communicate()
        why = f"no result after {limit:.0f} s"
    signal.signal(signal.SIGTERM, prev)
    text = out.decode("utf-8", "replace") if out else ""
    if PROBE_MARK in text:
        os.environ["ISEG_DIST_NATIVE"] = "1"
        return "stream-ordered RCCL through the C ABI (probe passed on every rank)"
    print(f"bench.py: rank {os.environ.get('RANK', '?')}: native-exchange probe did not pass ({why}); falling back to c10d + eager step\n"
          f"---- probe output (tail) ----\n{text[-1200:]}", file=sys.stderr)
    os.environ["ISEG_DIST_NATIVE"] = "0"
    return f"c10d work objects (native-exchange probe did not pass: {why})"


def probe_native(args):
    """the probe job (child of choose_exchange): see there.  On a box without GPUs (--check-launch, tests/test_bench_launch.py) the training
    steps are skipped and ISEG_BENCH_TEST_PROBE = ok | fail | hang scripts the outcome of rank 1."""
    from iseg_amd import dist

    rank = int(os.environ.get("RANK", "0"))
    ok = 1.0
    if args.check_launch:
        mode = os.environ.get("ISEG_BENCH_TEST_PROBE", "ok")
        if rank == 1 and mode == "fail":
            os._exit(5)
        if rank == 1 and mode == "hang":
            time.sleep(3600)
        dist.init(backend="gloo")
        flag = torch.tensor([ok], dtype=torch.float32)
    else:
        from iseg_amd.data import synthetic_batch
        from iseg_amd.graphs import GraphedTrainStep

        try:
            strategy, model, trainer = build_trainer(args)
            x, y = synthetic_batch(args.batch, args.size, args.size, seed=100 + rank)
            x, y = x.cuda(), y.cuda()
            losses = [float(trainer.train_step(x, y)[0]) for _ in range(2)]
            step = GraphedTrainStep(trainer, warmup=0)
            if not step._eligible(x):
                raise RuntimeError("the data-parallel step is not capturable with this exchange")
            losses += [float(step(x, y)[0]) for _ in range(3)]
            torch.cuda.synchronize()
            if not all(v == v and abs(v) < 1e4 for v in losses):
                raise RuntimeError(f"losses {losses}")
        except Exception as e:      # noqa: BLE001 -- any failure means: do not use this exchange
            print(f"probe rank {rank}: {type(e).__name__}: {e}", flush=True)
            ok = 0.0
        flag = torch.tensor([ok], dtype=torch.float32, device="cuda")
    torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MIN)
    if float(flag.item()) == 1.0:
        print(PROBE_MARK, flush=True)
    sys.stdout.flush()
    os._exit(0)      # no teardown: the decision has been printed, a communicator that hangs on exit must not undo it


def build_trainer(args):
    from iseg_amd.core_env import common_env_setup
    from iseg_amd.core_optimizer import get_optimizer
    from iseg_amd.core_train import CoreTrain
    from iseg_amd.heads import convnext_tiny_aspp
    from iseg_amd.modelhelper import model_common_setup

    from iseg_amd import dist

    # ISEG_DIST_SINGLE_RANK_COLLECTIVES=1 on one GPU: the mirrored strategy with a world of one rank, i.e. every SyncBN message and gradient
    # bucket goes through RCCL -- the cost of the data-parallel plumbing itself, measurable without a second GPU
    one_device = args.gpus == 1 and not dist._forced()
    strategy = common_env_setup(use_one_device_strategy=one_device, mixed_precision=not args.fp32, random_seed=0)
    model = convnext_tiny_aspp(num_class=21, output_stride=32, build_input_size=(args.size, args.size))
    helper = model_common_setup(model, restore_checkpoint=False)
    helper.set_optimizer(get_optimizer(strategy, initial_lr=1e-4, end_lr=0.0, epoch_steps=1000, train_epoch=30, optimizer="adamw",
                                       adamw_weight_decay=0.05))
    trainer = CoreTrain(helper, None).create_trainable_model(21, ignore_label=255, batch_size=args.batch * strategy.num_replicas_in_sync)
    return strategy, model, trainer


def time_kernel(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(iters):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / iters * 1e-3   # seconds per launch


def _gemm_group_model(key):
    """algorithmic flops and HBM bytes of one launch of a timed GEMM group (operands read once, outputs written once)"""
    _, akc, bkc, M, N, K, act, has_pre, has_res, has_aux, adt, ddt, split = key[:13]
    ea = 2 if adt == torch.bfloat16 else 4
    ed = 2 if ddt == torch.bfloat16 else 4
    nbytes = (M * K + K * N) * ea + M * N * ed * (1 + int(has_pre)) + M * N * ed * (int(has_res) + int(has_aux))
    if len(key) > 13 and key[13] == 9:      # the weight-gradient PAIR launch: two products of these sizes (Z [4C, C] and dW1 [C, 4C]) over K rows
        return 4.0 * M * N * K, 2 * nbytes
    return 2.0 * M * N * K, nbytes


def _kernel_label(key):
    _, akc, bkc, M, N, K = key[:6]
    split = key[12]
    variant = key[13] if len(key) > 13 else 0
    act, has_pre, has_res, has_aux = key[6:10]
    orient = {(1, 0): "forward x[M,K] @ kernel[K,N]", (1, 1): "a[M,K] @ b[N,K]^T, both K-contiguous (data gradient, or forward on the K-contiguous kernel copy)",
              (0, 0): "wgrad x[K,M]^T @ dy[K,N]"}[(akc, bkc)]
    epi = {0: "", 1: "relu", 2: "gelu", 3: "gelu'(aux)", 4: "relu'(aux)", 5: "x aux"}.get(int(act), f"act{act}")
    epi = " + ".join(t for t in (epi, "second output" if has_pre else "", "residual" if has_res else "") if t)
    if variant == 9:
        return (f"iseg_mm::gemm_bf16_dma_tn_pair_kernel (both weight gradients of an un-fused ConvNeXt block in one launch: Z = gelu(h)^T dout [{M} x {N}] and "
                f"dW1 = y2^T dH [{N} x {M}] over {K} pixel rows, split-K slabs)")
    name = "iseg_mm::gemm_bf16_kernel" if not variant else \
        "iseg_mm::gemm_bf16_dma_kernel<%s>" % {1: "128x64,4 stages", 2: "256x128,3 stages", 3: "128x128,2 stages", 4: "128x128,3 stages",
                                              5: "256x128,3 stages,persistent", 6: "256x192,2 stages"}[variant]
    if variant == 2 and _dma_symbol(key)[0] and "ELi32EE" in _dma_symbol(key)[0]:
        name = "iseg_mm::gemm_bf16_dma_kernel<256x128,3 stages of 32 K,two workgroups per CU>"
    return f"{name} ({orient}; M={M} N={N} K={K}{', split-K slabs' if split else ''}{'; epilogue ' + epi if epi else ''})"


def _dma_symbol(key):
    """(mangled-name fragment, total grid size) of the LDS-DMA GEMM instantiation a timed group runs on, or (None, None)"""
    _, akc, bkc, M, N, K, act, has_pre, has_res, has_aux = key[:10]
    variant = key[13] if len(key) > 13 else 0
    shapes = {2: (4, 2, 3, 4), 6: (4, 2, 2, 6), 4: (2, 2, 3, 4)}      # variant -> (WM, WN, NS, FN) of gemm_dma.h dispatch_dma
    if variant == 9:      # the weight-gradient pair launch: (tiles of both problems) x (splits of one resident round over 256 CUs) x 512 threads
        tiles = 2 * -(-M // 256) * -(-N // 128)
        want = max(1, min(256 // tiles, K // 512))
        kps = -(-(-(-K // want)) // 128) * 128
        return "gemm_bf16_dma_tn_pair_kernel", tiles * -(-K // kps) * 512
    if variant not in shapes:
        return None, None
    wm, wn, ns, fn = shapes[variant]
    ek = 1 if (int(act) == 2 and has_pre) else 2 if int(act) == 5 else 3 if (int(act) == 0 and has_res) else 4 if (int(act) == 0 and not has_aux) else 0
    bm, bn = wm * 64, wn * fn * 16
    tiles = -(-M // bm) * -(-N // bn)
    kt = 1 if K % 64 else 0      # (the K-tail instantiation, gemm_dma.h)
    # (round 6: the 256 x 128 form runs 32-deep ring stages, two workgroups per CU, for whole-K problems with K <= 512 -- gemm_dma.h dispatch_dma)
    bks = 32 if (variant == 2 and K <= 512 and K % 32 == 0 and K >= 96 and os.environ.get("ISEG_GEMM_DMA_BK32", "1") != "0") else 64
    if os.environ.get("ISEG_GEMM_DMA_BK32") == "2" and variant == 2 and K % 32 == 0 and K >= 96:
        bks = 32
    return f"gemm_bf16_dma_kernelILi{wm}ELi{wn}ELi{ns}EDF16bLb0ELi{fn}ELi{ek}ELb{kt}ELi{bks}EE", tiles * wm * wn * 64


def pick_dominant_memory(report):
    """the non-GEMM group (today: the depthwise 7 x 7 launches, instrumented in kernels.dwconv2d) with the largest total time, or None"""
    totals = {k: v[0] * v[1] for k, v in report.items() if k[0] != "gemm"}
    return max(totals, key=totals.get) if totals else None


def roofline_memory(key, sec, launches, steps, step_seconds):
    """`roofline_memory`: the largest non-GEMM launch group of the step against the HBM roofline (round-5 verdict item 7: over the whole trace the top
    row is a depthwise kernel, not a GEMM).  Algorithmic bytes = the activation read once + written once (+ the residual-branch gradient read once
    for the data-gradient form); the 49 K fp32 weights are noise.  The vector-pipe rate is reported beside it: a 7 x 7 depthwise pass is 49 MACs per
    4 bytes, so at 64 T MAC/s (v_pk_fma_f32 peak, EXPERIMENTS.md) its arithmetic needs about as long as its bytes do."""
    _, N, H, W, C, K, dil, flip, has_add, dtype = key
    es = 2 if dtype == torch.bfloat16 else 4
    elems = N * H * W * C
    nbytes = elems * es * (2 + int(has_add))
    macs = elems * K * K
    ach = nbytes / sec / 1e9
    return {"kernel": (f"depthwise {K}x{K} {'data gradient (+ residual-branch gradient)' if flip else 'forward'}: dwconv_fwd_dma_kernel / dwconv7_mfma_kernel "
                       f"(csrc/dwconv.hip, csrc/dwconv_mfma.hip), N={N} {H}x{W}x{C}"),
            "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
            "launch_us": round(sec * 1e6, 2), "launches_per_step": launches / steps, "algorithmic_bytes_per_launch": int(nbytes),
            "share_of_step": round(sec * launches / steps / step_seconds, 4),
            "valu_tmacs": round(macs / sec / 1e12, 2), "valu_peak_tmacs": 64.0, "traffic": None}


def pick_dominant(report):
    """the GEMM group with the largest total time; several stage-2 GEMM groups sit within a few per cent of each other, so among the
    groups within 10 % of the maximum prefer one whose launches a PMC pass can attribute (see _dma_symbol).  (The largest non-GEMM group gets its
    own entry: pick_dominant_memory / roofline_memory.)"""
    report = {k: v for k, v in report.items() if k[0] == "gemm"}
    totals = {k: v[0] * v[1] for k, v in report.items()}
    top = max(totals.values())
    near = sorted((k for k, t in totals.items() if t >= 0.9 * top), key=lambda k: -totals[k])
    for k in near:      # prefer a group whose instantiation rocprof can name on its own (LDS-DMA kernel, epilogue kind in the symbol)
        if _dma_symbol(k)[0] is not None:
            return k
    return near[0]


def roofline_from_timer(report, steps, survey=None):
    """the GEMM launch group (kernel template + shape) with the largest share of the step; `report` holds the live
    HIP-event timings of the timed region, `survey` (all GEMM groups, one warm-up step) gives its share of GEMM time"""
    report = {k: v for k, v in report.items() if k[0] == "gemm"}
    if survey is not None:
        survey = {k: v for k, v in survey.items() if k[0] == "gemm"}
    best = max(report.items(), key=lambda kv: kv[1][0] * kv[1][1])
    key, (sec, launches) = best
    flops, nbytes = _gemm_group_model(key)
    _, akc, bkc, M, N, K, act, has_pre, has_res, has_aux, adt, ddt, split = key[:13]
    intensity = flops / nbytes
    ridge = MFMA_BF16_PEAK_TF * 1e12 / (HBM_PEAK_GBS * 1e9)
    if survey is not None and key in survey:
        share = survey[key][0] * survey[key][1] / sum(v[0] * v[1] for v in survey.values())
    else:
        share = sec * launches / sum(v[0] * v[1] for v in report.values())
    common = {"kernel": _kernel_label(key),
              "launches_per_step": launches / steps, "launch_us": round(sec * 1e6, 2),
              "share_of_gemm_time": round(share, 4), "algorithmic_bytes_per_launch": int(nbytes),
              "algorithmic_flops_per_launch": int(flops), "flop_per_byte": round(intensity, 1), "traffic": None}
    # HBM bytes per launch from the PMC counters: they cannot be read inside this process, so the figure comes from the separate rocprofv3
    # --pmc FETCH_SIZE / WRITE_SIZE passes over this same command (tools/collect_evidence.sh -> profiles/r06_pmc.json, corrections per
    # MI355X_MICROARCH.md), looked up by kernel symbol (the epilogue kind is part of it since round 3) and grid; null when the file was
    # taken on another source tree or holds no such kernel.
    sym, grid = _dma_symbol(key)
    doc, why = _pmc_on_this_tree()
    if doc is not None and sym is not None:
        for rec in doc.get("kernels", []):
            if sym in rec["kernel"] and int(rec["grid"]) == grid:
                common["traffic"] = rec["traffic"]
                common["profiled_launch_us"] = rec["avg_us"]
                common["traffic_source"] = f"{PMC_FILE} kernel {sym} grid {grid} (rocprofv3 --pmc FETCH_SIZE, WRITE_SIZE; separate passes; same source stamp)"
                break
    elif why:
        common["traffic_source"] = why
    if intensity >= ridge:
        ach = flops / sec / 1e12
        common.update({"bound": "mfma", "achieved": round(ach, 1), "peak": MFMA_BF16_PEAK_TF, "unit": "TFLOP/s",
                       "frac": round(ach / MFMA_BF16_PEAK_TF, 4), "hbm_gbs": round(nbytes / sec / 1e9, 1)})
    else:
        ach = nbytes / sec / 1e9
        common.update({"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4),
                       "tflops": round(flops / sec / 1e12, 1)})
    return common


PMC_FILE = os.path.join("profiles", "r06_pmc.json")


def _pmc_on_this_tree():
    """profiles/r06_pmc.json (rocprofv3 FETCH_SIZE / WRITE_SIZE / SQ passes over this same command, tools/collect_evidence.sh) when it was taken
    on EXACTLY this source tree (tools/source_stamp.py), else (None, why)"""
    try:
        with open(os.path.join(ROOT, PMC_FILE)) as f:
            doc = json.load(f)
    except (OSError, ValueError):
        return None, f"{PMC_FILE} not found"
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from source_stamp import source_stamp

    have, now = (doc.get("stamp") or {}).get("source_sha16"), source_stamp()
    if have != now:
        return None, f"{PMC_FILE} was taken on source stamp {have}, this tree is {now}: counters not reported"
    return doc, None


def step_fractions(args, ips_per_gpu, sec_per_step):
    """the WHOLE step against both peaks, beside the single-kernel `roofline`: matrix-core fraction from the model's algorithmic flops
    (TRAIN_GFLOP_PER_IMAGE) at the measured rate; HBM fraction and matrix-core busy fraction from the step's PMC passes -- null unless the
    counters on file were taken on this very source tree, at the default batch and size"""
    out = {"mfma_frac": round(ips_per_gpu * TRAIN_GFLOP_PER_IMAGE / 1e3 / MFMA_BF16_PEAK_TF, 4), "hbm_frac": None, "traffic_bytes": None,
           "mfma_busy_frac": None}
    if args.batch == 16 and args.size == 512 and not args.fp32:
        doc, why = _pmc_on_this_tree()
        if doc is None:
            out["traffic_source"] = why
        else:
            st = doc["step"]
            out["traffic_bytes"] = int(st["traffic_bytes"])
            out["hbm_frac"] = round(st["traffic_bytes"] / sec_per_step / (HBM_PEAK_GBS * 1e9), 4)
            out["mfma_busy_frac"] = st.get("mfma_occupancy")
            out["traffic_source"] = (f"{PMC_FILE} (rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | SQ_VALU_MFMA_BUSY_CYCLES, separate passes); stamp "
                                     f"{doc['stamp']['source_sha16']} = this tree, taken at commit {doc['stamp'].get('git_head')}")
    return out


def cpu_baseline(args):
    """The CPU oracle (oracle/, a port: the TensorFlow reference cannot run here) on the host cores, on bounded samples:
      value / sample   the flagship's train step -- forward, mean ignore-label CE, backward -- in fp32, one image at the benchmark resolution;
      cfg1_train       BASELINE configs[0], the reference's own CPU-runnable case: ResNet-50 + ASPP, 256 x 256, batch 2, the same train step;
      cfg2_fp32_forward_ms   the fp32 forward of the flagship on the two-image parity batch (what the logits / argmax parity tests compare with)."""
    from oracle import models as OM
    from iseg_amd.data import synthetic_batch
    from iseg_amd.heads import convnext_tiny_aspp, resnet50_aspp
    from iseg_amd import nn

    import contextlib

    def build(factory, **kw):
        prev = nn.device()
        nn.set_device("cpu")
        try:
            with contextlib.redirect_stdout(sys.stderr):      # (build-time chatter: stdout carries the JSON line only)
                return factory(num_class=21, **kw)
        finally:
            nn.set_device(prev if prev.type != "cpu" else None)

    def grad_weights(w):
        return {k: (v.clone().requires_grad_(True) if not k.endswith(("moving_mean", "moving_variance")) else v) for k, v in w.items()}

    def timed(fn, budget_s, max_n):
        fn()
        t0 = time.time()
        n = 0
        while time.time() - t0 < budget_s and n < max_n:
            fn()
            n += 1
        return (time.time() - t0) / n, n

    from oracle import host_threads

    cores = host_threads.apply()      # the container's real core budget (cgroup quota): `cores` is what the baseline ran on
    w = OM.export_weights(build(convnext_tiny_aspp, build_input_size=(args.size, args.size)), dtype=torch.float32)
    x, y = synthetic_batch(1, args.size, args.size, seed=0)

    def step():
        OM.mean_ce_loss(OM.convnext_aspp_forward(grad_weights(w), x, training=True)["logits"], y).backward()

    dt, n = timed(step, 10.0, 8)
    out = {"value": round(1.0 / dt, 3), "unit": "images/s", "cores": cores, "kind": "port",
           "sample": f"{n} train steps (fwd+bwd, fp32, torch-CPU oracle) of ConvNeXt-T+ASPP on 1 image {args.size}x{args.size}"}
    x2, _ = synthetic_batch(2, args.size, args.size, seed=31)

    def fwd():
        with torch.no_grad():
            OM.convnext_aspp_forward(w, x2, training=False)

    dtf, nf = timed(fwd, 5.0, 6)
    out["cfg2_fp32_forward_ms"] = {"value": round(dtf * 1e3, 1), "sample": f"{nf} forward passes (fp32, torch-CPU oracle) of the flagship on the 2-image parity batch {args.size}x{args.size}"}
    w1 = OM.export_weights(build(resnet50_aspp, build_input_size=(256, 256)), dtype=torch.float32)
    x1, y1 = synthetic_batch(2, 256, 256, seed=4)

    def step1():
        OM.mean_ce_loss(OM.resnet_aspp_forward(grad_weights(w1), x1, training=True)["logits"], y1).backward()

    dt1, n1 = timed(step1, 6.0, 12)
    out["cfg1_train"] = {"value": round(2.0 / dt1, 3), "unit": "images/s",
                         "sample": f"{n1} train steps (fwd+bwd, fp32, torch-CPU oracle) of BASELINE configs[0]: ResNet-50+ASPP, 256x256, batch 2"}
    return out


def main():
    args = parse()
    env_world = int(os.environ.get("WORLD_SIZE", "0") or 0)
    if args.gpus > 1 and env_world == 0:
        sys.exit(self_launch(args))
    if env_world not in (0, args.gpus) or (args.gpus == 1 and env_world > 1):
        print(f"bench.py: --gpus {args.gpus} but the launcher set WORLD_SIZE={env_world}", file=sys.stderr)
        sys.exit(2)
    if args.probe_native:
        return probe_native(args)
    if args.check_launch:
        return check_launch(args)
    affinity = pin_to_gpu_numa(int(os.environ.get("LOCAL_RANK", "0")), max(args.gpus, 1))      # (before any GPU call: affinity only, no exec)
    exchange = choose_exchange(args)      # (N > 1: before this process touches the GPU; sets ISEG_DIST_NATIVE for the measured job)
    from iseg_amd import dist
    from iseg_amd.data import synthetic_batch

    if args.gpus > 1:      # the ranks agree on the exchange before the first collective of the job (see agree_exchange)
        dist.init()
        exchange = agree_exchange(exchange, torch.device("cuda", dist.local_rank()))

    # stdout carries the ONE JSON line and nothing else: until that line is printed, file descriptor 1 points at stderr, so neither the
    # reference's build-time chatter mirrored by the host code ("Use the random seed ...") nor a native library's banner (RCCL prints its
    # version to stdout when the first communicator comes up) can land in front of it
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    strategy, model, trainer = build_trainer(args)
    rank = dist.rank()
    world = joined_ranks(torch.device("cuda", dist.local_rank()))
    if args.gpus != world:
        print(f"bench.py: --gpus {args.gpus} but {world} rank(s) joined the process group", file=sys.stderr)
        sys.exit(3)
    x, y = synthetic_batch(args.batch, args.size, args.size, seed=100 + rank)
    x, y = x.cuda(), y.cuda()
    from iseg_amd import kernels as K

    # Roofline: the last warm-up step times every GEMM launch with HIP events on the launch stream to find the dominant
    # (template, shape) group.  Eager headline (--eager-step, or data parallel): the timed region brackets one launch in four of that group.
    # Replayed headline (default on one GPU): `--roofline-steps` eager steps right in front of the timed region bracket EVERY launch of the
    # group -- event-record nodes inside a captured graph carry no timestamps on ROCm 7.2 (EXPERIMENTS.md), and the kernels, shapes and buffers
    # are the same ones the replay runs.
    from iseg_amd.graphs import GraphedTrainStep

    replay = (not args.eager_step) and GraphedTrainStep(trainer)._eligible(x)
    want_roofline = rank == 0 and not args.no_roofline
    survey = None
    for i in range(args.warmup):
        if want_roofline and i == args.warmup - 1:
            survey = K.KernelTimer()
            K.KERNEL_TIMER[0] = survey
        trainer.train_step(x, y)
    K.KERNEL_TIMER[0] = None
    timer = None
    survey_report = None
    if want_roofline:
        dominant = dominant_mem = None
        if survey is not None:
            survey_report = survey.report()
            dominant = pick_dominant(survey_report)
            dominant_mem = pick_dominant_memory(survey_report)
        # eager headline: one launch in four of the dominant group carries an event pair (>= 40 samples over the default 20 steps): every pair idles
        # the stream for a few microseconds, and 19 pairs per step had taxed the headline by 2.5 %
        timer = K.KernelTimer(only=[k for k in (dominant, dominant_mem) if k is not None] or None, every=1 if replay else 4)
        K.KERNEL_TIMER[0] = timer
    step_fn = trainer.train_step
    roofline_steps = args.steps
    if replay:
        if timer is not None:
            roofline_steps = max(1, args.roofline_steps)
            for _ in range(roofline_steps):
                trainer.train_step(x, y)
            torch.cuda.synchronize()
            K.KERNEL_TIMER[0] = None
        # the whole step replayed from one HIP graph (iseg_amd/graphs.py): capture happens here, outside the timed region
        step_fn = GraphedTrainStep(trainer, warmup=0)
        try:
            step_fn(x, y)
            step_fn(x, y)
            # the synthetic batch lives in the captured step's input buffers from here on (what an on-device input pipeline does: it writes
            # the batch where the step reads it): no per-step device-to-device copy of the 50 MB image tensor, exactly as in the eager step
            bufs = step_fn.input_buffers(x, y)
            if bufs is not None:
                bufs[0].copy_(x)
                bufs[1].copy_(y)
                x, y = bufs
        except Exception as e:      # a capture that fails must not cost the measurement: the eager step is the same kernels, enqueued by the host
            print(f"bench.py: capturing the step into a HIP graph failed ({type(e).__name__}: {e}); timing the eager step instead", file=sys.stderr)
            torch.cuda.synchronize()
            step_fn, replay = trainer.train_step, False
            step_fn(x, y)
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    enqueue = 0.0      # host time inside the step calls (nothing in the loop waits for the GPU): what this rank's CPU needs to keep its GPU fed
    for _ in range(args.steps):
        te = time.perf_counter()
        losses = step_fn(x, y)
        enqueue += time.perf_counter() - te
    torch.cuda.synchronize()
    dist.barrier()
    elapsed = time.perf_counter() - t0
    K.KERNEL_TIMER[0] = None
    t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
    enq = torch.zeros(world, dtype=torch.float64, device="cuda")
    enq[rank] = enqueue / args.steps * 1e3
    if world > 1:
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        torch.distributed.all_reduce(enq, op=torch.distributed.ReduceOp.SUM)
    elapsed = float(t.item())
    host_enqueue_ms = [round(float(v), 3) for v in enq.tolist()]
    loss_val = float(losses[0])
    if rank != 0:
        return
    ips = args.batch * world * args.steps / elapsed
    res = {
        "metric": "train_images_per_sec (ConvNeXt-T+ASPP 512x512 crop, fwd+bwd+AdamW, whole job)",
        "value": round(ips, 2), "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if args.fp32 else "bf16", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[1]: ConvNeXt-T + ASPP (output stride 32), {args.size}x{args.size}, {args.batch} img/GPU, "
                               "random-init weights, drop_path 0.1, dropout 0.1, SyncBN, AdamW, running mIoU",
                   "global_batch": args.batch * world, "parallelism": f"dp{world}"},
        "images_per_sec_per_gpu": round(ips / world, 2),
        "mfma_roofline_frac": round(ips / world * TRAIN_GFLOP_PER_IMAGE / 1e3 / MFMA_BF16_PEAK_TF, 4),
        "final_loss": round(loss_val, 5),
        "exchange": exchange,
        "host_enqueue_ms_per_step": host_enqueue_ms,      # per rank; graph replay: one launch call, eager: every kernel's enqueue
        "host_affinity": affinity,      # rank 0's own (every rank pins itself the same way)
        "step_mode": ("hip-graph replay of the whole step (one graph launch per step)" if replay and any(e.get("graph") is not None for e in getattr(step_fn, "entries", {}).values())
                      else "eager (every kernel enqueued from the host)"),
    }
    res["step"] = step_fractions(args, ips / world, elapsed / args.steps)
    if timer is not None:
        rep = timer.report()
        res["roofline"] = roofline_from_timer(rep, roofline_steps, survey_report)
        mem = [(k, v) for k, v in rep.items() if k[0] != "gemm"]
        if mem:
            k, (sec, launches) = max(mem, key=lambda kv: kv[1][0] * kv[1][1])
            res["roofline_memory"] = roofline_memory(k, sec, launches, roofline_steps, elapsed / args.steps)
        res["roofline"]["measured_in"] = (f"{roofline_steps} eager steps directly in front of the timed region, every launch of the group bracketed by HIP events on the launch stream"
                                          if replay else "the timed region, one launch in four of the group bracketed by HIP events on the launch stream")
    if world == 1 and not args.no_cpu_baseline:
        res["cpu_baseline"] = cpu_baseline(args)
    sys.stdout.flush()
    os.dup2(real_stdout, 1)
    os.close(real_stdout)
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
