"""Data-parallel plumbing: one process per GPU, torch.distributed over RCCL (backend "nccl" on ROCm) / gloo on CPU.

Replaces distribution/distribution_utils.py:75-95,158-169 (tf.distribute.MirroredStrategy + ReplicaContext.all_reduce):
  * SyncBN statistics: one packed [2C+1] fp32 all-reduce per layer in forward, one [2C] in backward (functional.py);
  * gradients: the flat fp32 gradient buffer is cut into contiguous buckets; a bucket's all-reduce(sum) is launched
    asynchronously as soon as every parameter in it has its gradient enqueued, so RCCL traffic over xGMI overlaps the
    rest of the backward pass.  xGMI is a point-to-point mesh (7 links/GPU), so few large messages are used.
"""
import os

import torch
import torch.distributed as td


def is_initialized():
    return td.is_available() and td.is_initialized()


def world_size():
    return td.get_world_size() if is_initialized() else 1


def rank():
    return td.get_rank() if is_initialized() else 0


def local_rank():
    return int(os.environ.get("LOCAL_RANK", "0"))


def _forced():
    """ISEG_DIST_SINGLE_RANK_COLLECTIVES=1: a world of ONE rank still creates the process group and issues every collective (SyncBN
    messages, async gradient buckets, broadcasts).  A 1-GPU box can this way drive the real RCCL backend -- stream ordering of the
    async handles, in-place reduction of flat-buffer slices -- that otherwise only an 8-GPU node would touch."""
    return os.environ.get("ISEG_DIST_SINGLE_RANK_COLLECTIVES", "0") == "1"


def active():
    """True when collectives have to be issued"""
    return is_initialized() and (td.get_world_size() > 1 or _forced())


def init(backend=None):
    """Initialise from the torchrun environment (RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT)."""
    if is_initialized():
        return
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    if ws <= 1 and not _forced():
        return
    os.environ.setdefault("RANK", "0")
    if backend is None:
        backend = os.environ.get("ISEG_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
    if torch.cuda.is_available():
        torch.cuda.set_device(local_rank())
        from . import kernels

        kernels._DEVICE_INDEX[0] = None      # the launch path caches the device index (kernels.stream)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29500")
    td.init_process_group(backend=backend, rank=int(os.environ["RANK"]), world_size=ws)


# ---------------------------------------------------------------------------------------------------------
# Stream-ordered exchange (ISEG_DIST_NATIVE=1): the collectives go through the C ABI's own RCCL communicator (csrc/comm.hip:
# iseg_comm_* / iseg_allreduce_sum) instead of c10d work objects.  A collective is then ONE enqueue on an explicit HIP stream -- nothing the
# host waits for, no work object, no watchdog thread -- which is what makes a data-parallel training step capturable into a HIP graph
# (iseg_amd/graphs.py): SyncBN messages ride the compute stream between the kernels that produce and consume them, gradient buckets a side
# stream that forks from / joins the compute stream through events (graph edges when captured).  c10d stays the rendezvous (it carries the
# 128-byte RCCL id to the other ranks) and the default exchange; "emulate" runs the same scheduling with a blocking c10d primitive (gloo on the
# CPU: tests/test_dist_gloo.py).  Reference: distribution/distribution_utils.py:75-95,158-169 (MirroredStrategy's NCCL all-reduce).
# ---------------------------------------------------------------------------------------------------------
_NATIVE = {"comm": None, "comm_side": None, "side": None, "join": None}


def native_mode():
    """'' (c10d work objects), 'rccl' (C-ABI communicator, stream-ordered) or 'emulate' (stream-ordered scheduling over a blocking c10d call)"""
    v = os.environ.get("ISEG_DIST_NATIVE", "0")
    if v in ("1", "rccl"):
        return "rccl" if torch.cuda.is_available() else "emulate"
    return "emulate" if v == "emulate" else ""


def _native_comm(which="comm"):
    """the C-ABI communicators of this process, created on first use: rank 0 draws the id, c10d broadcasts it.  Two of them: "comm" carries the
    SyncBN messages on the compute stream, "comm_side" the gradient buckets on the side stream -- operations on ONE RCCL communicator are
    serialised in host-issue order whatever streams they are enqueued on, so a bucket sum sharing the SyncBN communicator would sit in front of the
    backward pass's next SyncBN message instead of overlapping it.  Every rank creates both in the same order (first use is the first forward
    SyncBN message / the first bucket of the first step, identical on all ranks)."""
    if _NATIVE[which] is None:
        # BOTH communicators come up at the first request for EITHER, in one fixed order ("comm", then "comm_side"): communicator creation is a
        # collective (an id broadcast over c10d + ncclCommInitRank), so a rank that met its first gradient bucket before its first SyncBN message
        # (a model whose first trainable layer has no SyncBN in front of it on one rank's code path, a resumed step) must not pair its
        # "comm_side" creation with another rank's "comm" creation (tests/test_dist_gloo.py::test_native_communicators_come_up_in_one_order)
        for name in ("comm", "comm_side"):
            if _NATIVE[name] is None:
                _NATIVE[name] = _create_native_comm()
    return _NATIVE[which]


def _create_native_comm():
    import ctypes as C

    from . import _hip

    L = _hip.lib()
    uid = torch.zeros(128, dtype=torch.uint8)
    if rank() == 0:
        buf = C.create_string_buffer(128)
        _hip.check(L.iseg_comm_unique_id(buf), "iseg_comm_unique_id")
        uid = torch.frombuffer(bytearray(buf.raw), dtype=torch.uint8).clone()
    if world_size() > 1:
        dev = torch.device("cuda", torch.cuda.current_device()) if td.get_backend() == "nccl" else torch.device("cpu")
        u = uid.to(dev)
        td.broadcast(u, 0)
        uid = u.cpu()
    comm = C.c_void_p()
    _hip.check(L.iseg_comm_init(C.byref(comm), world_size(), rank(), bytes(uid.numpy().tobytes())), "iseg_comm_init")
    return comm


def _stream_all_reduce(t, raw_stream, which="comm"):
    """in-place sum over the ranks, enqueued on `raw_stream` (a hipStream_t handle); returns at once"""
    mode = native_mode()
    if mode == "rccl":
        from . import _hip
        from . import kernels as K

        if not t.is_contiguous():
            raise ValueError("stream-ordered all-reduce needs a contiguous tensor (a slice of the flat buffers is)")
        code = K.F32 if t.dtype == torch.float32 else K.BF16 if t.dtype == torch.bfloat16 else None
        if code is None or not t.is_cuda:      # integer counts (confusion matrices) and host tensors: not for the C ABI's device entry; take the c10d call
            if t.is_cuda and torch.cuda.is_current_stream_capturing():
                raise RuntimeError(f"stream-ordered all-reduce of a {t.dtype} tensor under graph capture: only float32 / bfloat16 go through the C ABI's "
                                   "communicator, and a c10d work object cannot be captured")
            td.all_reduce(t, op=td.ReduceOp.SUM)
            return
        _hip.check(_hip.lib().iseg_allreduce_sum(_native_comm(which), K.ptr(t), t.numel(), code, raw_stream), "iseg_allreduce_sum")
    else:
        td.all_reduce(t, op=td.ReduceOp.SUM)


def all_reduce_sum(t, async_op=False):
    if not active():
        return None
    if native_mode() and not async_op:
        from . import kernels as K

        _stream_all_reduce(t, K.stream() if t.is_cuda else None)      # on the compute stream: ordered behind its producer, in front of its consumer
        return None
    return td.all_reduce(t, op=td.ReduceOp.SUM, async_op=async_op)


def broadcast(t, src=0):
    if active():
        td.broadcast(t, src)


def barrier():
    if active():
        td.barrier()


class GradReducer:
    """Bucketed, backward-overlapped all-reduce(sum) over the flat gradient buffer of a ParamStore."""

    def __init__(self, store, bucket_bytes=48 << 20, head_bytes=(4 << 20, 16 << 20)):
        """Buckets are contiguous runs of the flat buffer in model order.  The FIRST buckets hold the first layers, whose gradients are
        ready last: nothing is left to overlap their all-reduce with, so they are small (head_bytes: 4 MB, then 16 MB) and only the
        later ones -- ready early in the backward pass -- take the full bucket_bytes."""
        env = os.environ.get("ISEG_DP_BUCKETS_MB")      # experiments: "head0,head1,...,bucket" in MiB, e.g. "8,1024"
        if env:
            mb = [int(float(v) * (1 << 20)) for v in env.split(",")]
            head_bytes, bucket_bytes = tuple(mb[:-1]), mb[-1]
        self.store = store
        self.buckets = []      # (lo, hi, n_params)
        self.bucket_of = {}    # id(param) -> bucket index
        lo, cnt, acc = None, 0, 0
        for p, off, n in store.segments:
            if lo is None:
                lo = off
            self.bucket_of[id(p)] = len(self.buckets)
            cnt += 1
            acc += store.padded(n) * 4
            cap = head_bytes[len(self.buckets)] if len(self.buckets) < len(head_bytes) else bucket_bytes
            if acc >= min(cap, bucket_bytes):
                self.buckets.append((lo, off + store.padded(n), cnt))
                lo, cnt, acc = None, 0, 0
        if cnt:
            last = store.segments[-1]
            end = last[1] + store.padded(last[2])
            if self.buckets and acc < bucket_bytes // 4:      # a small remainder rides the previous bucket instead of costing a collective
                plo, _, pcnt = self.buckets[-1]
                self.buckets[-1] = (plo, end, pcnt + cnt)
                for p, off, n in store.segments[-cnt:]:
                    self.bucket_of[id(p)] = len(self.buckets) - 1
            else:
                self.buckets.append((lo, end, cnt))
        self.uses = None       # id(param) -> ready() calls per step, learnt from the first step
        self.reset()

    def reset(self):
        self.seen = {}                                  # id(param) -> ready() calls this step
        self.launched = [False] * len(self.buckets)
        self.handles = []
        if self.uses is None:
            self.pending = None                         # first step: nothing is launched before finish()
        else:
            self.pending = [0] * len(self.buckets)      # parameters of the bucket that still owe a gradient contribution
            for pid, n in self.uses.items():
                if n > 0:
                    self.pending[self.bucket_of[pid]] += 1

    def _launch(self, b):
        lo, hi, _ = self.buckets[b]
        self.launched[b] = True
        from . import kernels as K

        K.deferred_flush()      # the all-reduce reads the gradient buffer: reductions still queued into it must be enqueued first
        if native_mode():
            self._launch_stream_ordered(self.store.flat_g[lo:hi])
        else:
            self.handles.append(all_reduce_sum(self.store.flat_g[lo:hi], async_op=True))

    def _launch_stream_ordered(self, t):
        """the bucket's sum on the side stream, behind everything the compute stream has enqueued so far (an event = a graph edge under
        capture); finish() joins the side stream back.  On the CPU (emulation over gloo) there are no streams: the call is the blocking one."""
        if not t.is_cuda:
            _stream_all_reduce(t, None)
            return
        if _NATIVE["side"] is None:
            _NATIVE["side"] = torch.cuda.Stream(device=t.device)
        side = _NATIVE["side"]
        fork = torch.cuda.Event()
        fork.record()                      # on the current (compute / capture) stream
        side.wait_event(fork)
        _stream_all_reduce(t, side.cuda_stream, "comm_side")
        self.side_used = True

    def ready(self, *params):
        """called by an operator's backward once the kernels that write these parameters' gradients are enqueued.  A parameter
        used k times in the forward pass (a layer applied twice, a shared table) reports k times; the bucket goes out when every
        parameter in it has reported as often as it did in the first step, which only counts (static graphs: the count is the
        same every step).  A report for a bucket that is already on the wire would mean a silently wrong sum, so it raises."""
        if not active():
            return
        for p in params:
            if p is None:
                continue
            pid = id(p)
            b = self.bucket_of.get(pid)
            if b is None:
                continue
            c = self.seen[pid] = self.seen.get(pid, 0) + 1
            if self.launched[b]:
                raise RuntimeError(f"GradReducer: gradient of {getattr(p, 'iseg_name', '?')} reported after its bucket was all-reduced "
                                   "(the model used it more often than in the first step); call relearn() after changing the graph")
            if self.pending is not None and c == self.uses.get(pid, 0):
                self.pending[b] -= 1
                if self.pending[b] == 0:
                    self._launch(b)

    def relearn(self):
        self.uses = None
        self.reset()

    def finish(self):
        """launch whatever was not triggered by ready() and make the compute stream wait for all buckets"""
        if active():
            if self.uses is None:
                self.uses = dict(self.seen)
            for b in range(len(self.buckets)):
                if not self.launched[b]:
                    self._launch(b)
            for h in self.handles:
                if h is not None:
                    h.wait()
            if getattr(self, "side_used", False):      # stream-ordered buckets: the optimizer step waits for the side stream
                join = torch.cuda.Event()
                join.record(_NATIVE["side"])
                torch.cuda.current_stream().wait_event(join)
                self.side_used = False
        self.reset()


_ACTIVE_REDUCER = [None]


def set_active_reducer(r):
    _ACTIVE_REDUCER[0] = r


def grads_ready(*params):
    r = _ACTIVE_REDUCER[0]
    if r is not None:
        r.ready(*params)      # (queued parameter-gradient reductions are flushed when a bucket actually goes out: GradReducer._launch)
