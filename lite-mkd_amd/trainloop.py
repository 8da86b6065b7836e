"""The per-episode training loop of the reference (trainwandb.py:111-188 train, :190-287 train_task,
:359-417 test, :419-443 prepare_task) with its quirks kept:
  * `iteration` is incremented before use and the optimizer fires when (iteration+1) % tasks_per_batch == 0,
    so the first step comes after tasks_per_batch-1 episodes (:125,141-143);
  * scheduler.step() runs every episode (:145) — milestones are in episodes;
  * only the CE term carries the 1/16 accumulation scaling (distillers.py:326-335).
wandb / checkpoint-file side effects are replaced by an optional `log` callback."""
import numpy as np
import torch

from . import ops
from .parallel import EarlyAllReduce, FlatParams, world_size
from ._lib import lib
import ctypes


PIPELINE_DEPTH = 1          # PipelinedEpisodes: forwards queued ahead of the oldest pending backward
REPACK_AT_STEP = True
EARLY_ALLREDUCE = True      # world > 1: all-reduce the bucket's tail under the last backward pass of an optimizer interval (parallel.EarlyAllReduce)


class FusedOptimizer:
    """SGD (no momentum; trainwandb.py:103-104, options.py:72-73) or Adam over the flat buffers."""

    def __init__(self, module, opt="sgd", lr=1e-4):
        self.bucket = FlatParams(module)
        self.opt, self.lr, self.steps = opt, float(lr), 0
        self.early = EarlyAllReduce(self.bucket, module)      # tail of the gradient bucket all-reduced under the last backward pass (world > 1)
        if opt == "adam":
            self.m = torch.zeros_like(self.bucket.flat)
            self.v = torch.zeros_like(self.bucket.flat)
        elif opt != "sgd":
            raise KeyError(opt)

    def expect_step(self):
        """call before the LAST episode of an optimizer interval (the loop knows: (iteration + 1) % every == 0 after it): under episode
        parallelism the all-reduce of the bucket's tail - last trunk stage, heads, matcher - is then issued as soon as the backward pass
        has left the last stage, and runs beside the rest of it (parallel.EarlyAllReduce).  Harmless on one process."""
        if EARLY_ALLREDUCE and world_size() > 1:
            self.early.arm()
            ops.GRAD_READY_HOOK = self.early.hook
        else:
            ops.GRAD_READY_HOOK = None

    def zero_grad(self):
        ops.join_all_streams()              # weight gradients (ops.SIDE_WGRAD) and direct parameter gradients (side / auxiliary streams) in flight
        self.bucket.zero_grad()

    def step(self):
        b = self.bucket
        ops.join_all_streams()              # every stream that writes the gradient buffer or its shadow, or reads the packs re-packed below
        ops.GRAD_READY_HOOK = None
        upto = self.early.finish()          # the tail went out during the backward pass: wait for it, all-reduce the rest here
        b.fold_shadow()                     # gradients the side / auxiliary streams' kernels accumulated directly (ops.DIRECT_PARAM_GRAD);
                                            # unconditionally: the flag may have been switched off since they were written
                                            # (the early part folded and zeroed its own tail of the shadow)
        b.allreduce_grads(upto)
        self.steps += 1
        ops.WEIGHT_EPOCH[0] += 1            # invalidates the packed-weight cache (raw-pointer update below)
        s = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        P = lambda t: ctypes.c_void_p(t.data_ptr())       # noqa: E731
        if self.opt == "sgd":
            lib().call("lmkd_sgd_step", P(b.flat), P(b.grad), ctypes.c_float(self.lr), b.numel, 0, s)
        else:
            lib().call("lmkd_adam_step", P(b.flat), P(b.grad), P(self.m), P(self.v), ctypes.c_float(self.lr),
                       ctypes.c_float(0.9), ctypes.c_float(0.999), ctypes.c_float(1e-8), self.steps, b.numel, 0, s)
        if REPACK_AT_STEP:
            # every cached fragment-order pack of the convolution weights in ONE launch behind the update (otherwise each is re-packed
            # by three launches at its first use in the next forward: ~120 launches per step, one step per 2 episodes at world 8)
            ops.refresh_packs()


class MultiStepLR:
    """torch.optim.lr_scheduler.MultiStepLR(milestones, gamma=0.1) on FusedOptimizer.lr.  The reference calls scheduler.step()
    once per episode (trainwandb.py:145), so its milestones ([20000, 40000], options.py) count EPISODES.  Under episode
    parallelism every rank runs the loop over its own episodes; one step() here therefore advances the count by the world size:
    the milestones stay in global episodes and the learning-rate schedule of an N-GPU run matches the 1-GPU reference run."""

    def __init__(self, optimizer, milestones, gamma=0.1, episodes_per_step=None):
        self.opt, self.milestones, self.gamma, self.n = optimizer, sorted(milestones), gamma, 0
        self.base = optimizer.lr
        self.inc = world_size() if episodes_per_step is None else episodes_per_step

    def step(self):
        self.n += self.inc
        self.opt.lr = self.base * self.gamma ** sum(1 for m in self.milestones if m <= self.n)


def prepare_task(task_dict, device, images_to_device=True):
    """trainwandb.py:419-443"""
    context_images, context_labels = task_dict["support_set"][0], task_dict["support_labels"][0]
    target_images, target_labels = task_dict["target_set"][0], task_dict["target_labels"][0]
    context_teacher_feature = task_dict["support_set_feature_teacher"][0]
    target_teacher_feature = task_dict["target_set_feature_teacher"][0]
    real_target_labels = task_dict.get("real_target_labels", [None])[0]
    batch_class_list = task_dict.get("batch_class_list", [None])[0]
    dev = torch.device(device)
    if dev.type != "cuda":
        if images_to_device:
            context_images, target_images = context_images.to(dev), target_images.to(dev)
            context_teacher_feature, target_teacher_feature = context_teacher_feature.to(dev), target_teacher_feature.to(dev)
        return (context_images, target_images, context_teacher_feature, target_teacher_feature, context_labels.to(dev),
                target_labels.long().to(dev), real_target_labels, batch_class_list)
    # host tensors go up through pinned memory, non-blocking (ops.h2d_async): a pageable .to(device) stalls the host until the
    # stream has drained, i.e. once per episode
    if images_to_device:
        context_images = ops.h2d_async(context_images, dev)
        target_images = ops.h2d_async(target_images, dev)
        context_teacher_feature = ops.h2d_async(context_teacher_feature, dev)
        target_teacher_feature = ops.h2d_async(target_teacher_feature, dev)
    cpu_labels = context_labels if context_labels.device.type == "cpu" else None
    context_labels = _labels_to_device(context_labels, dev, None)
    if cpu_labels is not None and context_labels is not cpu_labels:
        ops.note_cpu_labels(context_labels, cpu_labels)
    target_labels = _labels_to_device(target_labels, dev, torch.int64)
    return (context_images, target_images, context_teacher_feature, target_teacher_feature, context_labels,
            target_labels, real_target_labels, batch_class_list)


# The frozen teacher head needs only the teacher features of the episode: it is queued on a stream of its own BEFORE the student's
# forward, so its ~40 small kernels run beside the student's trunk instead of after it (the loss waits for both).
TEACHER_STREAM = True


def _tensors(obj):
    if torch.is_tensor(obj):
        yield obj
    elif isinstance(obj, dict):
        for v in obj.values():
            yield from _tensors(v)
    elif isinstance(obj, (list, tuple)):
        for v in obj:
            yield from _tensors(v)


def _teacher_forward(teacher, ctf, labels, ttf, way):
    """-> (teacher output dict, stream to join before the outputs are used | None)"""
    if not (TEACHER_STREAM and ctf.is_cuda) or torch.is_grad_enabled() and any(p.requires_grad for p in teacher.parameters()):
        return teacher(ctf, labels, ttf), None
    main, aux = torch.cuda.current_stream(ctf.device), ops.aux_stream(ctf.device)
    # per-episode tensors both heads share are created on the main stream first (class plan: small H2D copies + one cat)
    if labels.dim() == 1 and ctf.dim() == 3:
        ops.get_plan(labels, way).full_rowmap(ctf.shape[0] + ttf.shape[0])
    aux.wait_stream(main)
    with torch.cuda.stream(aux):
        out = teacher(ctf, labels, ttf)
    for t in (ctf, ttf, labels):
        t.record_stream(aux)
    for t in _tensors(out):
        t.record_stream(main)
    return out, aux


_LABEL_CACHE = {}      # (id of the CPU tensor, _version, dtype) -> (weakref, device copy): labels of a resident episode pool go up once


def _labels_to_device(t, dev, dtype):
    if t.is_cuda:
        return t.to(dtype) if dtype is not None and t.dtype != dtype else t
    import weakref
    key = (id(t), t._version, dtype)
    hit = _LABEL_CACHE.get(key)
    if hit is not None and hit[0]() is t:
        return hit[1]
    d = ops.h2d_async(t, dev, dtype)
    if len(_LABEL_CACHE) > 256:
        _LABEL_CACHE.clear()
    _LABEL_CACHE[key] = (weakref.ref(t), d)
    return d


def train_task(task_dict, student, teacher, distiller, accuracy_fn, config):
    """trainwandb.py:190-287 for the logits-based distillers."""
    out = _train_task_prepared(prepare_task(task_dict, config.device), student, teacher, distiller, accuracy_fn, config)
    release_inputs(task_dict)
    return out


def _train_task_prepared(prepared, student, teacher, distiller, accuracy_fn, config):
    task_loss, task_accuracy = _episode_forward(prepared, student, teacher, distiller, accuracy_fn, config)
    task_loss.backward(retain_graph=False)
    ops.h2_fence_step()      # fp32h2: the range fence judges this episode's tensors (asynchronous; a no-op in the other arithmetics)
    return task_loss.detach(), task_accuracy, {"accuracy": task_accuracy}


def _episode_forward(prepared, student, teacher, distiller, accuracy_fn, config):
    """forward + loss + accuracy of one episode (trainwandb.py:190-283) -> (task_loss with its autograd graph, accuracy)"""
    (context_images, target_images, context_teacher_feature, target_teacher_feature, context_labels,
     target_labels, _, _) = prepared
    teacher_model_dict, joined = _teacher_forward(teacher, context_teacher_feature, context_labels, target_teacher_feature, config.way)
    model_dict = student(context_images, context_labels, target_images)
    if joined is not None:
        torch.cuda.current_stream(config.device).wait_stream(joined)
    target_logits = model_dict["logits"]
    teacher_logits = teacher_model_dict["logits"]
    if config.distill_name == "KL_feature":         # trainwandb.py:209-226: the features travel inside the logits dicts
        target_logits = {"logits": target_logits,
                         "feature": torch.cat([model_dict["context_features"], model_dict["target_features"]], 0)}
        teacher_logits = {"logits": teacher_logits,
                          "feature": torch.cat([context_teacher_feature, target_teacher_feature], 0)}
    loss = getattr(distiller, config.distill_name)(target_logits, teacher_logits, target_labels)
    task_loss = loss["loss"]
    if config.distill_name == "KL_feature":         # :243-244
        target_logits = target_logits["logits"]
    if isinstance(target_logits, dict) and "kl" in target_logits and "ce" in target_logits:
        task_accuracy, _ = ops.accuracy(target_logits["kl"], target_logits["ce"], target_labels)     # :247-257,278
    elif isinstance(target_logits, dict):
        task_accuracy = accuracy_fn(target_logits["kl"], target_labels)
    else:
        task_accuracy = accuracy_fn(target_logits, target_labels)
    return task_loss, task_accuracy


class PipelinedEpisodes:
    """Software pipelining ACROSS episodes: the forward of episode i + 1 is queued (on the other stream set, ops.set_lane) before the
    backward of episode i, so the two run side by side on the GPU.  Within one episode the launch sequence has stretches the chip
    cannot fill - the matcher / loss section between the trunk's forward and backward (~2 ms of 126-workgroup GEMMs and
    wavefront-reduction kernels on an otherwise idle chip), the stem / pooling passes, every convolution launch's last partial round
    of workgroups (10-20 % of a launch at 200 frames) - and the only other work available is the second trunk call's.  Two
    episodes in flight double that supply.  Nothing else changes: each episode runs the same kernels on the same data; forwards stay
    in program order among themselves (BatchNorm running statistics), backwards too (every gradient accumulates in the order of the
    sequential loop, so the flat gradient buffer is bit-identical to it), weights only change in the optimizer step, before which
    flush() drains the pipeline.

        pipe = PipelinedEpisodes(student, teacher, distiller, accuracy_fn, config)
        for task_dict in loader:
            done = pipe.push(task_dict)          # -> (loss, accuracy) of the PREVIOUS episode, or None for the first
            if optimizer step due: pipe.flush(); optimizer.step(); optimizer.zero_grad()
        pipe.flush()                             # -> (loss, accuracy) of the last episode; the caller's stream waits for everything"""

    def __init__(self, student, teacher, distiller, accuracy_fn, config, depth=None):
        self.student, self.teacher, self.distiller, self.accuracy_fn, self.config = student, teacher, distiller, accuracy_fn, config
        # depth: how many forwards may be queued ahead of the oldest pending backward (1: the forward of episode i + 1 beside the backward
        # of episode i; 2: the forward of episode i + 2 starts as soon as that of i + 1 is queued - it fills the stretch in which the
        # longer backward of episode i is still running - at the price of a third episode's activations in memory)
        self.depth = int(depth if depth is not None else PIPELINE_DEPTH)
        self.queue = []
        self.bwd_done = None
        self.count = 0

    @property
    def pending(self):
        return self.queue[-1] if self.queue else None

    def push(self, task_dict):
        dev = self.config.device
        caller = torch.cuda.current_stream(dev)
        lane = self.count % (self.depth + 1)
        self.count += 1
        ops.set_lane(lane)
        try:
            main = ops.lane_main(dev)
            main.wait_stream(caller)                 # the optimizer's weight update / zeroed gradients, the caller's input tensors
            with torch.cuda.stream(main):
                prepared = prepare_task(task_dict, dev)
                loss, acc = _episode_forward(prepared, self.student, self.teacher, self.distiller, self.accuracy_fn, self.config)
            self.queue.append((loss, acc, lane, task_dict))
            return self._backward(self.queue.pop(0)) if len(self.queue) > self.depth else None
        finally:
            ops.set_lane(0)

    def _backward(self, item):
        loss, acc, lane, task_dict = item
        dev = self.config.device
        ops.set_lane(lane)
        try:
            main = ops.lane_main(dev)
            with torch.cuda.stream(main):
                if self.bwd_done is not None:
                    main.wait_event(self.bwd_done)       # gradients accumulate in episode order
                loss.backward(retain_graph=False)
                ops.h2_fence_step()      # fp32h2: the range fence judges this episode's tensors (asynchronous)
                # the lane's side / auxiliary streams wrote parameter gradients directly (autograd got None and joined nothing): the
                # event that orders the NEXT backward's accumulations must cover them
                d = torch.device(dev)
                for table in (ops._side_streams, ops._aux_streams):
                    s = table.get((d.type, ops.device_index(d), lane))
                    if s is not None:
                        main.wait_stream(s)
                ev = torch.cuda.Event()
                ev.record(main)
                self.bwd_done = ev
                release_inputs(task_dict)      # a StreamedEpisodes loader may now re-stage the episode's input set (behind every stream that read it)
        finally:
            ops.set_lane(0)
        return loss.detach(), acc

    def join(self):
        """make the caller's current stream wait for every backward pass run so far - and with it for the loss / accuracy tensors push() and
        flush() have returned (they are written on the lane streams: a caller that reads them, float(loss), must join first; the host runs
        several episodes ahead of the device)"""
        if self.bwd_done is not None:
            torch.cuda.current_stream(self.config.device).wait_event(self.bwd_done)

    def flush(self):
        """run the pending backward passes (oldest first); afterwards the caller's current stream waits for every episode pushed so far.
        -> (loss, accuracy) of the last episode, or a list of them when more than one was pending, or None"""
        outs = []
        while self.queue:
            outs.append(self._backward(self.queue.pop(0)))
        self.join()
        if not outs:
            return None
        return outs[0] if len(outs) == 1 else outs


class StreamedEpisodes:
    """The loader side of the loop (trainwandb.py:87-88: a DataLoader worker; :419-443 prepare_task; video_reader.py:474-485 the
    loader's output), overlapped with the compute that is timed: wraps a HOST-side episode source and yields task_dicts whose frames and
    teacher features already sit in device memory.

    source.host_episode() -> {"frames": uint8 [F, H, W, 3] (decoded frames of every video of the episode, support videos first, one
    resolution), "params": [(flip, x1, y1)] per video (GpuFrameTransform.draw), "features": float [N, L, 2048], "support_labels",
    "target_labels" (+ optional "real_target_labels", "batch_class_list"), "ns": number of support videos}.  A prefetch THREAD runs it
    one or two episodes ahead (np.load, frame picking and pinning happen off the training thread); the training thread only enqueues:
    one H2D copy of the uint8 frames (a quarter of the fp32 tensors' bytes) + one of the features on a COPY stream, the frame transform
    (Resize -> crop / flip -> ToTensor, video_transform.GpuFrameTransform.batch) there too, into one of `sets` static input sets.  The
    compute stream waits for the set's `ready` event; the copy stream, before it overwrites a set, for the events the loop recorded on
    every stream that read it (release(): the lane's streams and the weight-gradient stream after the episode's backward pass was
    queued - nothing on the compute side ever waits for the loader).  With cross-episode pipelining three episodes are in flight
    (backward i, forward i + 1, staging i + 2) and the loader stages one episode ahead of the loop's request: hence sets >= 4."""

    def __init__(self, source, config, device, sets=4, prefetch=2):
        import queue
        import threading
        from .video_transform import GpuFrameTransform
        self.source, self.c, self.device = source, config, torch.device(device)
        self.tf = GpuFrameTransform(config.img_size, self.device)
        self.copy = torch.cuda.Stream(device=self.device)
        self.nsets = max(3, int(sets))
        self.sets = [None] * self.nsets
        self.dataset = getattr(source, "dataset", source)
        self._q = queue.Queue(maxsize=max(1, int(prefetch)))
        # a ring of PINNED host buffers the prefetch thread copies the episodes into (pinning a fresh 92 MB tensor per episode costs tens of
        # milliseconds of hipHostMalloc: 15.8 instead of 49 episodes/s); a slot is reused once the H2D copy that read it has passed
        self._pins = [None] * (max(1, int(prefetch)) + 2)
        self._stop = threading.Event()
        self._thread = None
        self._threading = threading
        self.length = getattr(source, "length", 10 ** 9)
        self.staged = 0

    # ---- host side (prefetch thread)
    def _produce(self):
        try:
            n = 0
            while not self._stop.is_set() and n < self.length:
                h = dict(self.source.host_episode())
                if h["frames"].is_pinned() and h["features"].is_pinned() and h["features"].dtype is torch.float32:
                    n += 1      # the source decodes into pinned memory of its own (and keeps it alive until the episode has been staged)
                    while not self._stop.is_set():
                        try:
                            self._q.put(h, timeout=0.1)
                            break
                        except Exception:
                            continue
                    continue
                slot = self._pins[n % len(self._pins)]
                if slot is None or slot["frames"].shape != h["frames"].shape or slot["features"].shape != h["features"].shape:
                    slot = self._pins[n % len(self._pins)] = {"frames": torch.empty(h["frames"].shape, dtype=torch.uint8).pin_memory(),
                                                              "features": torch.empty(h["features"].shape, dtype=torch.float32).pin_memory(),
                                                              "event": None}
                if slot["event"] is not None:
                    slot["event"].synchronize()            # (host wait in the prefetch thread: the copy that read this slot has finished)
                    slot["event"] = None
                slot["frames"].copy_(h["frames"])
                slot["features"].copy_(h["features"])
                h["frames"], h["features"], h["_slot"] = slot["frames"], slot["features"], slot
                n += 1
                while not self._stop.is_set():
                    try:
                        self._q.put(h, timeout=0.1)
                        break
                    except Exception:      # queue.Full
                        continue
        except BaseException as e:      # surfaces in the training thread
            self._q.put(e)
        finally:
            self._q.put(None)

    def _set(self, k, h):
        s = self.sets[k]
        F_, H, W, _ = h["frames"].shape
        N, L, D = h["features"].shape
        S = self.c.img_size
        if s is None or s["u8"].shape != h["frames"].shape or s["feat"].shape != h["features"].shape:
            s = self.sets[k] = {"u8": torch.empty((F_, H, W, 3), dtype=torch.uint8, device=self.device),
                                "x": torch.empty((F_, S, S, 4), dtype=torch.float32, device=self.device),
                                "feat": torch.empty((N, L, D), dtype=torch.float32, device=self.device),
                                "ready": torch.cuda.Event(), "free": []}
            for t in (s["u8"], s["x"], s["feat"]):
                ops._audit.audit_ok(t, "static input set: released to the copy stream through the events of release()")
        return s

    def _stage(self, h):
        k = self.staged % self.nsets
        self.staged += 1
        s = self._set(k, h)
        L = h["features"].shape[1]
        with torch.cuda.stream(self.copy):
            for ev in s["free"]:
                self.copy.wait_event(ev)                   # every stream that read this set last time has passed
            s["free"] = []
            s["u8"].copy_(h["frames"], non_blocking=True)
            s["feat"].copy_(h["features"], non_blocking=True)
            if h.get("_slot") is not None:                 # the pinned slot is free again once these copies have run
                ev = torch.cuda.Event()
                ev.record(self.copy)
                h["_slot"]["event"] = ev
            self.tf.batch(s["u8"], h["params"], L, out=s["x"])
            s["ready"].record(self.copy)
        ns = int(h["ns"])
        td = {"support_set": s["x"][:ns * L], "target_set": s["x"][ns * L:],
              "support_set_feature_teacher": s["feat"][:ns], "target_set_feature_teacher": s["feat"][ns:],
              "support_labels": h["support_labels"], "target_labels": h["target_labels"],
              "real_target_labels": h.get("real_target_labels", h["target_labels"]),
              "batch_class_list": h.get("batch_class_list", torch.arange(self.c.way).float())}
        td = {key: v.unsqueeze(0) for key, v in td.items()}
        td["_input_set"] = (self, k)
        return td

    def release(self, k):
        """the episode that read set k has been queued completely (forward and backward): record where every stream that may still read
        the set stands - the copy stream waits for these before it overwrites the set"""
        s = self.sets[k]
        if s is None:
            return
        # the streams of the lane the episode ran on (release() is called under that lane: PipelinedEpisodes._backward, train_task) and the
        # weight-gradient stream - not every stream of every lane (twice as many events to record and for the copy stream to wait for)
        streams = [torch.cuda.current_stream(self.device)] + list(ops._WG_STREAM.values())
        for table in (ops._side_streams, ops._aux_streams, ops._lane_mains):
            streams += [st for key, st in table.items() if key[2] == ops.LANE[0]]
        seen = set()
        for st in streams:
            if st.cuda_stream in seen:
                continue
            seen.add(st.cuda_stream)
            ev = torch.cuda.Event()
            ev.record(st)
            s["free"].append(ev)

    def __iter__(self):
        if self._thread is None:
            self._thread = self._threading.Thread(target=self._produce, daemon=True)
            self._thread.start()
        # staging runs ONE episode ahead of the loop: the copy + transform of episode i + 1 are queued (on the copy stream) before the loop
        # queues episode i's forward pass, so they execute under it instead of in front of episode i + 1 (round 5: 0.90 -> see
        # profiles/r05_bench.json `stream_inputs` of the resident rate); one more input set is in flight for it (sets >= 4)
        held = None
        while True:
            h = self._q.get()
            if isinstance(h, BaseException):
                raise h
            td = self._stage(h) if h is not None else None
            if held is not None:
                torch.cuda.current_stream(self.device).wait_event(self.sets[held["_input_set"][1]]["ready"])
                yield held
            if td is None:
                return
            held = td

    def close(self):
        self._stop.set()


def release_inputs(task_dict):
    """tell a StreamedEpisodes loader that the episode of this task_dict has been queued completely (a no-op for any other loader)"""
    tag = task_dict.get("_input_set") if isinstance(task_dict, dict) else None
    if tag is not None:
        tag[0].release(tag[1])


def init_model(config):
    """trainwandb.py:53-68: Student / Teacher from the registry names in `config`, moved to config.device"""
    from .model.model_select import Student, Teacher
    student, teacher = Student(config), Teacher(config)
    return student.to(config.device), teacher.to(config.device)


class TestAccuracies:
    """utils.TestAccuracies of the reference as far as make() uses it: the list of test sets a test run reports on"""

    def __init__(self, test_sets):
        self.datasets = list(test_sets)


def make(config, video_loader=None, base_seed=0, schedule=None):
    """trainwandb.py:78-109: everything the training loop needs, built from the reference's args namespace in the reference's order -
    -> (student, teacher, video_loader, distillers, accuracy_fn, test_accuracies, optimizer, scheduler), the reference's tuple.
    Differences that are the point of this build: the optimizer is FusedOptimizer (SGD / Adam kernels on the flat parameter and
    gradient buffers; under torch.distributed its step() all-reduces the gradient bucket) and, because the dataset / video I/O layer
    is outside the hot path (video_reader.VideoDataset: host JPEG decoding), the default `video_loader` is SyntheticEpisodes with the
    dataset's shapes and dtypes - pass the real DataLoader to train on data.  The replicas' initial weights are broadcast from rank 0."""
    from .distillers import Distiller
    from .parallel import rank as _rank
    from .utils import aggregate_accuracy
    if schedule is not None:      # the arithmetic mode and every schedule switch in one object (schedule.Schedule), installed process-wide
        schedule.apply()
    student, teacher = init_model(config)
    test_set = [config.dataset]
    if video_loader is None:
        video_loader = SyntheticEpisodes(config, base_seed=base_seed, rank=_rank(), device=config.device)
    distillers = Distiller(config.distill_name, config.cfg, config.device)
    accuracy_fn = aggregate_accuracy
    test_accuracies = TestAccuracies(test_set)
    if config.opt not in ("adam", "sgd"):
        raise KeyError(config.opt)
    optimizer = FusedOptimizer(student, config.opt, config.learning_rate)
    optimizer.bucket.broadcast_params(0)
    scheduler = MultiStepLR(optimizer, milestones=config.sch, gamma=0.1)
    optimizer.zero_grad()
    return student, teacher, video_loader, distillers, accuracy_fn, test_accuracies, optimizer, scheduler


class GraphedEpisode:
    """train_task as a captured hipGraph (torch.cuda.CUDAGraph): forward + loss + backward of one episode on the three HIP streams
    become ONE graph launch - ~600 kernel launches, their Python glue and the autograd engine leave the per-episode host path
    (10-13 ms of host time per episode eagerly: profiles/r02_host_bound.txt; bf16 tensors ran host-bound).

    What changes from episode to episode has to live in device memory the graph reads:
      * frames, teacher features, labels: the graph is captured on the episode's OWN device tensors and cached by their addresses
        (a resident pool of episodes, as in bench.py, replays without any copy); episodes that arrive in new tensors are copied
        into the static tensors of one generic graph;
      * the class plan (support labels -> class-sorted row map) is a device tensor built before the capture / refreshed in place;
        its STRUCTURE (shots per class) is baked into the graph, an episode with another structure runs eagerly;
      * dropout seeds: ops.SeedSlots - the mask kernels read their seeds from device memory, the host stages the seeds of the next
        replay (the same draws from torch's generator, in the same order, as the eager path: results are bit-identical);
      * packed weights: re-packed IN PLACE after every optimizer step (ops.refresh_packs), outside the graph.
    The first episode with a new key runs eagerly (warm-up: allocator, kernel attributes), the second is captured and replayed.
    Gradients accumulate into the same flat buffers as in the eager path; optimizer, scheduler and all-reduce stay outside."""

    KEYS = ("support_set", "target_set", "support_set_feature_teacher", "target_set_feature_teacher", "support_labels", "target_labels")

    def __init__(self, student, teacher, distiller, accuracy_fn, config, max_graphs=4):
        self.student, self.teacher, self.distiller, self.accuracy_fn, self.config = student, teacher, distiller, accuracy_fn, config
        self.max_graphs = max_graphs
        self.graphs = {}        # key -> entry dict
        self.seen = set()
        self.replays = self.eager = 0

    def _key(self, task_dict):
        """identity of the episode's tensors (the frames and teacher features must already be device tensors; the small label tensors
        may live on the host - prepare_task moves them before the capture)"""
        ts = [task_dict[k] for k in self.KEYS]
        if not all(t.is_cuda for t in ts[:4]):
            return None
        return tuple(t.data_ptr() for t in ts) + tuple(tuple(t.shape) for t in ts)

    def _seeds(self, n):
        from .model.classifiers.TRX_2fcsup import TemporalCrossTransformer as T
        return [T.draw_dropout_seed() for _ in range(n)]

    def _capture(self, task_dict, key):
        cfg = self.config
        prepared = prepare_task(task_dict, cfg.device)
        labels = prepared[4]
        plan = ops.get_plan(labels, cfg.way)                       # built (H2D copies) outside the capture; the capture finds it cached
        plan.full_rowmap(prepared[2].shape[0] + prepared[3].shape[0])
        plan.full_rowmap(prepared[0].shape[0] // cfg.seq_len + prepared[1].shape[0] // cfg.seq_len)
        streams = [ops.side_stream(cfg.device), ops.aux_stream(cfg.device)] + list(ops._WG_STREAM.values())
        ops.refresh_packs(streams)
        slots = ops.SeedSlots(cfg.device)
        g = torch.cuda.CUDAGraph()
        prev_sync = ops.SYNC_WGRAD_AT_BACKWARD_END
        torch.cuda.synchronize()
        ops.SEED_SLOTS = slots
        ops.SYNC_WGRAD_AT_BACKWARD_END = True                      # every forked stream joins the capture stream before it ends
        ops.amax_pool_reset()                                      # fp32h2: the words of the tensors' maxima come from a pool zeroed inside THIS graph
        try:
            with torch.cuda.graph(g):
                loss, acc, _ = _train_task_prepared(prepared, self.student, self.teacher, self.distiller, self.accuracy_fn, cfg)
                ops.wait_weight_grads()
        finally:
            ops.SEED_SLOTS = None
            ops.SYNC_WGRAD_AT_BACKWARD_END = prev_sync
        # the entry HOLDS the episode's tensors (their addresses cannot be recycled while it lives) and remembers their in-place
        # versions: a replay is only valid for these very tensors with unchanged contents (_resident)
        ent = {"graph": g, "slots": slots, "loss": loss, "acc": acc, "prepared": prepared, "plan": plan, "counts": tuple(plan.counts),
               "task": task_dict, "versions": tuple(task_dict[k]._version for k in self.KEYS)}
        self.graphs[key] = ent
        return ent

    def _eager(self, task_dict):
        self.eager += 1
        return train_task(task_dict, self.student, self.teacher, self.distiller, self.accuracy_fn, self.config)

    def _resident(self, ent, task_dict):
        """is this call about the very tensors the entry was captured on, unmodified since?  (The key is made of addresses: a loader that
        hands out FRESH device tensors gets recycled allocator addresses, and one that refills static buffers in place keeps them - in
        both cases the captured class plan, label copies and host-side label derivations would be stale.)"""
        return (all(task_dict[k] is ent["task"][k] for k in self.KEYS)
                and tuple(task_dict[k]._version for k in self.KEYS) == ent["versions"])

    def _replay(self, ent):
        ops.wait_weight_grads()                                        # an eager episode's weight gradients may still be in flight
        ops.refresh_packs()
        ent["slots"].stage(self._seeds(ent["slots"].used))
        ent["graph"].replay()
        self.replays += 1
        loss, acc = ent["loss"].clone(), ent["acc"].clone()            # the graph's output tensors are overwritten by the next replay
        return loss, acc, {"accuracy": acc}

    def _generic(self, task_dict):
        """episodes that arrive in NEW tensors (a data loader): one graph on static copies of the inputs; the new episode's tensors are
        copied in, its class plan (computed on the host from the labels, as ops.ClassPlan does) overwrites the static plan's device
        tensors.  An episode whose class structure differs from the captured one runs eagerly."""
        cfg = self.config
        shapes = tuple(tuple(task_dict[k].shape) for k in self.KEYS)
        ent = self.graphs.get(("generic",) + shapes)
        lab = task_dict["support_labels"][0]
        plan = ops.ClassPlan(lab, cfg.way, lab.detach().to("cpu"))
        if ent is None:
            if ("generic",) + shapes not in self.seen:
                self.seen.add(("generic",) + shapes)
                return self._eager(task_dict)
            static = {k: v.detach().to(cfg.device).clone() if k in self.KEYS else v for k, v in task_dict.items()}
            ent = self._capture(static, ("generic",) + shapes)
            return self._replay(ent)
        if tuple(plan.counts) != ent["counts"] or plan.classes != ent["plan"].classes:
            return self._eager(task_dict)
        for k in self.KEYS:
            ent["task"][k].copy_(task_dict[k], non_blocking=True)
        ent["plan"]._packed.copy_(plan._packed)      # class ids, row map and its full-row-map views in one device copy
        # prepare_task's derived tensor (target labels as int64) lives in the static `prepared` tuple
        ent["prepared"][5].copy_(ent["task"]["target_labels"][0].long())
        return self._replay(ent)

    # ---- several episodes per graph.  A one-episode graph must join the weight-gradient stream before it ends, so its ~4 ms tail no
    # longer runs under the next episode's forward (the replay loop is ~5 % slower than the eager loop on the GPU side).  The episodes
    # between two optimizer steps depend on each other only through the gradient buffers, which the eager loop orders on the
    # weight-gradient stream anyway: captured as ONE graph they keep that overlap and join once, where the optimizer would wait too.
    def run_interval(self, task_dicts):
        """the episodes of one optimizer interval, resident in device tensors (keyed like __call__): first sight eager, then captured
        as one graph, then one replay per call.  -> [(loss, accuracy, info)] per episode, bit-identical to the eager loop"""
        keys = tuple(self._key(t) for t in task_dicts)
        if any(k is None for k in keys):
            return [self(t) for t in task_dicts]
        ikey = ("interval",) + keys
        ent = self.graphs.get(ikey)
        if ent is None:
            if ikey not in self.seen:
                self.seen.add(ikey)
                return [self._eager(t) for t in task_dicts]
            ent = self._capture_interval(task_dicts, ikey)
        ops.wait_weight_grads()
        ops.refresh_packs()
        ent["slots"].stage(self._seeds(ent["slots"].used))
        ent["graph"].replay()
        self.replays += len(task_dicts)
        out = [(l.clone(), a.clone()) for l, a in ent["out"]]
        return [(l, a, {"accuracy": a}) for l, a in out]

    def _capture_interval(self, task_dicts, ikey):
        cfg = self.config
        prepared = [prepare_task(t, cfg.device) for t in task_dicts]
        plans = []
        for pr in prepared:
            plan = ops.get_plan(pr[4], cfg.way)
            plan.full_rowmap(pr[2].shape[0] + pr[3].shape[0])
            plan.full_rowmap(pr[0].shape[0] // cfg.seq_len + pr[1].shape[0] // cfg.seq_len)
            plans.append(plan)
        streams = [ops.side_stream(cfg.device), ops.aux_stream(cfg.device)] + list(ops._WG_STREAM.values())
        ops.refresh_packs(streams)
        slots = ops.SeedSlots(cfg.device, n=16 * len(prepared))
        g = torch.cuda.CUDAGraph()
        prev_sync = ops.SYNC_WGRAD_AT_BACKWARD_END
        torch.cuda.synchronize()
        ops.SEED_SLOTS = slots
        ops.SYNC_WGRAD_AT_BACKWARD_END = False                     # the weight-gradient stream runs on under the next episode's forward ...
        ops.amax_pool_reset()
        out = []
        try:
            with torch.cuda.graph(g):
                for pr, plan in zip(prepared, plans):
                    ops._plan_cache[0], ops._plan_cache[1], ops._plan_cache[2] = pr[4], cfg.way, plan      # (get_plan's one-entry identity cache)
                    loss, acc, _ = _train_task_prepared(pr, self.student, self.teacher, self.distiller, self.accuracy_fn, cfg)
                    out.append((loss, acc))
                ops.wait_weight_grads()                            # ... and joins the capture stream once, at the end
        finally:
            ops.SEED_SLOTS = None
            ops.SYNC_WGRAD_AT_BACKWARD_END = prev_sync
        ent = {"graph": g, "slots": slots, "out": out, "prepared": prepared, "plans": plans, "tasks": list(task_dicts)}
        self.graphs[ikey] = ent
        return ent

    def __call__(self, task_dict):
        key = self._key(task_dict)
        if key is None:
            return self._generic(task_dict)
        ent = self.graphs.get(key)
        if ent is None:
            if key not in self.seen:                                   # first sight: eager (also the warm-up of the capture)
                if sum(1 for k in self.seen if k[0] != "generic") >= self.max_graphs:
                    return self._generic(task_dict)                    # device tensors, but new ones every time (a data loader)
                self.seen.add(key)
                return self._eager(task_dict)
            if sum(1 for k in self.graphs if k[0] != "generic") >= self.max_graphs:
                return self._generic(task_dict)
            ent = self._capture(task_dict, key)
        elif not self._resident(ent, task_dict):
            # same addresses, other tensors or modified contents: not the resident episode this graph was captured on
            del self.graphs[key]
            self.seen.discard(key)
            return self._generic(task_dict)
        return self._replay(ent)


def save_checkpoint(student, iteration, config, bn_stats=None):
    """trainwandb.py:171-180: {'iteration', 'model_state_dict'} -> <save_dir>/<yyyymmddHHMM><mode><iteration>.pt (the
    file load_student / the reference's test.py read).  RANK-LOCAL: writes a file from this process's weights, no collective
    (a caller that follows the reference's `if rank == 0: save_checkpoint(...)` pattern cannot deadlock).  bn_stats: optional
    {key: tensor} replacing the BatchNorm running statistics in the saved copy - under episode parallelism the loop passes
    parallel.sync_bn_running_stats(student), the statistics pooled over the ranks (that call IS collective)."""
    import os
    import time
    os.makedirs(config.save_dir, exist_ok=True)
    path = os.path.join(config.save_dir, "%s%s%d.pt" % (time.strftime("%Y%m%d%H%M", time.localtime(time.time())), config.mode, iteration))
    bn_stats = bn_stats or {}
    sd = {k: (bn_stats[k] if k in bn_stats else v).detach().cpu() for k, v in student.state_dict().items()}
    torch.save({"iteration": iteration, "model_state_dict": sd}, path)
    return path


def checkpoint_all_ranks(student, iteration, config):
    """COLLECTIVE: pool the BatchNorm running statistics over the ranks (one all-reduce), then rank 0 writes the checkpoint.
    -> path on rank 0, None elsewhere"""
    from .parallel import rank as _rank, sync_bn_running_stats
    pooled = sync_bn_running_stats(student)
    return save_checkpoint(student, iteration, config, pooled) if _rank() == 0 else None


def _crossed(n_before, n_after, period):
    """did the count of processed GLOBAL episodes pass a multiple of `period` between n_before and n_after?"""
    return period > 0 and n_after // period > n_before // period


def train(student, teacher, video_loader, distiller, optimizer, scheduler, accuracy_fn, config, log=None, schedule=None):
    """trainwandb.py:111-188 (optimizer cadence, print, checkpoint every save_freq, test at test_iters).
    Returns (losses, accuracies) as python floats.

    Under episode parallelism (world W > 1) every rank runs this loop over its own episodes and EVERY counter of the reference
    stays in GLOBAL episodes: the loop ends after ceil(training_iterations / W) local iterations, the optimizer fires every
    tasks_per_batch / W local iterations (tasks_per_batch global episodes; W must divide it), MultiStepLR advances by W per
    local iteration, and print_freq / save_freq / test_iters trigger when the global count (local iteration x W) passes them.
    At W = 1 all of this reduces to the reference's conditions literally."""
    if schedule is not None:      # run the loop under this schedule.Schedule (restored afterwards); without one: the three switches below
        with schedule.applied():
            cfg2 = _with(config, side_wgrad=schedule.side_wgrad, direct_param_grad=schedule.direct_param_grad,
                         pipeline_episodes=schedule.pipeline_episodes)
            return train(student, teacher, video_loader, distiller, optimizer, scheduler, accuracy_fn, cfg2, log)
    losses, accuracies = [], []
    world = world_size()
    if config.tasks_per_batch % world != 0:
        raise ValueError("tasks_per_batch = %d is not divisible by the world size %d: the optimizer step could not keep the reference's "
                         "%d-episode gradient accumulation" % (config.tasks_per_batch, world, config.tasks_per_batch))
    total_iterations = -(-config.training_iterations // world)
    every = max(1, config.tasks_per_batch // world)
    iteration = 0
    # FusedOptimizer waits for the side-stream weight gradients itself (step / zero_grad), so backward() need not: the next
    # episode's forward then overlaps the tail of the previous episode's weight gradients
    sync_prev, side_prev, direct_prev = ops.SYNC_WGRAD_AT_BACKWARD_END, ops.SIDE_WGRAD, ops.DIRECT_PARAM_GRAD
    ops.SYNC_WGRAD_AT_BACKWARD_END = not isinstance(optimizer, FusedOptimizer)
    ops.SIDE_WGRAD = getattr(config, "side_wgrad", True)      # this loop owns the optimizer: weight gradients on their own stream
    # ... and BatchNorm / Linear / TRX parameter gradients added into .grad by the kernels themselves (the optimizer folds the shadow)
    ops.DIRECT_PARAM_GRAD = isinstance(optimizer, FusedOptimizer) and getattr(config, "direct_param_grad", True)
    try:
        return _train_loop(student, teacher, video_loader, distiller, optimizer, scheduler, accuracy_fn, config, log, losses,
                           accuracies, total_iterations, every, iteration, world)
    finally:
        ops.join_all_streams()
        if isinstance(optimizer, FusedOptimizer):
            optimizer.bucket.fold_shadow()          # leave complete gradients in .grad for whoever reads them next
        ops.SYNC_WGRAD_AT_BACKWARD_END, ops.SIDE_WGRAD, ops.DIRECT_PARAM_GRAD = sync_prev, side_prev, direct_prev


def _with(config, **kw):
    import copy
    c = copy.copy(config)
    for k, v in kw.items():
        setattr(c, k, v)
    return c


def _train_loop(student, teacher, video_loader, distiller, optimizer, scheduler, accuracy_fn, config, log, losses, accuracies,
                total_iterations, every, iteration, world=1):
    # cross-episode pipelining (schedule.Schedule.pipeline_episodes): the forward of this episode is queued BEFORE the backward of the
    # previous one; results arrive one episode late and are bit-identical to the sequential loop (PipelinedEpisodes)
    pipe = (PipelinedEpisodes(student, teacher, distiller, accuracy_fn, config)
            if getattr(config, "pipeline_episodes", False) and torch.device(config.device).type == "cuda" else None)

    def took(res):
        for r in (res if isinstance(res, list) else [res]):
            if r is not None:
                losses.append(r[0])
                accuracies.append(r[1])
    for task_dict in video_loader:
        if iteration >= total_iterations:
            break
        iteration += 1
        torch.set_grad_enabled(True)
        if (((iteration + 1) % every == 0) or (iteration == (total_iterations - 1))) and hasattr(optimizer, "expect_step"):
            optimizer.expect_step()          # this episode's backward completes the interval's gradients
        if pipe is not None:
            took(pipe.push(task_dict))
        else:
            took(train_task(task_dict, student, teacher, distiller, accuracy_fn, config)[:2])
        if ((iteration + 1) % every == 0) or (iteration == (total_iterations - 1)):
            if pipe is not None:
                took(pipe.flush())           # the step needs the gradients of every episode up to this one
            optimizer.step()
            optimizer.zero_grad()
        scheduler.step()
        # the reference's (iteration + 1) % period == 0 tests, on the global episode count g = (iteration + 1) * world
        g0, g1 = iteration * world, (iteration + 1) * world
        last = (iteration + 1) == total_iterations
        if log is not None and _crossed(g0, g1, config.print_freq) and losses:      # (pipelined: results arrive one episode late)
            n = max(1, config.print_freq // world)
            if pipe is not None:
                pipe.join()                  # the logged tensors were written on the lane streams
            log(iteration, float(torch.stack(losses[-n:]).mean()), float(torch.stack(accuracies[-n:]).mean()))
        if _crossed(g0, g1, config.save_freq) and not last:
            if pipe is not None:
                took(pipe.flush())           # the BatchNorm running statistics of the last forward are written on a lane stream
            checkpoint_all_ranks(student, iteration, config)
        if any(g0 < t <= g1 for t in getattr(config, "test_iters", ())) and not last:
            if pipe is not None:
                took(pipe.flush())
            accuracy_dict = test(student, video_loader, accuracy_fn, config)
            if log is not None:
                log(iteration, accuracy_dict, None)
    if pipe is not None:
        took(pipe.flush())
    return [float(x) for x in losses], [float(x) for x in accuracies]


def test_task(task_dict, model, accuracy_fn, config):
    """trainwandb.py:289-357 (student branch)."""
    (context_images, target_images, _, _, context_labels, target_labels, _, _) = prepare_task(task_dict, config.device)
    logits = model(context_images, context_labels, target_images)["logits"]
    if isinstance(logits, dict) and "kl" in logits and "ce" in logits:
        acc, _ = ops.accuracy(logits["kl"], logits["ce"], target_labels)
    elif isinstance(logits, dict):
        acc = accuracy_fn(logits["kl"], target_labels)
    else:
        acc = accuracy_fn(logits, target_labels)
    return {"test_accuracy": acc}


def test(model, video_loader, accuracy_fn, config):
    """trainwandb.py:359-417: eval mode, no_grad, mean accuracy x100 and 95 % CI = 196*std/sqrt(n)."""
    model.eval()
    accuracies = []
    with torch.no_grad():
        if hasattr(video_loader, "dataset"):
            video_loader.dataset.train = False
        iteration = 0
        for task_dict in video_loader:
            if iteration >= config.num_test_tasks:
                break
            iteration += 1
            accuracies.append(test_task(task_dict, model, accuracy_fn, config)["test_accuracy"].item())
        if hasattr(video_loader, "dataset"):
            video_loader.dataset.train = True
    model.train()
    accuracy = np.array(accuracies).mean() * 100.0
    confidence = (196.0 * np.array(accuracies).std()) / np.sqrt(len(accuracies))
    return {config.dataset: {"accuracy": accuracy, "confidence": confidence}}


class SyntheticEpisodes:
    """Stand-in for VideoDataset + DataLoader(batch_size=1) (video_reader.py:398-485): yields task_dicts with
    the same keys/shapes/dtypes, frames ~ U[0,1), teacher features ~ N(0,1), shuffled float labels.
    Episode e of rank r uses seed base + r*10**6 + e."""

    def __init__(self, config, base_seed=0, rank=0, device="cpu", length=10 ** 9, train=True):
        self.c, self.base, self.rank, self.device, self.length = config, base_seed, rank, device, length
        self.train = train
        self.dataset = self

    def episode(self, e):
        c = self.c
        g = torch.Generator(device="cpu").manual_seed(self.base + self.rank * 10 ** 6 + e)
        q = c.query_per_class if self.train else c.query_per_class_test
        ns, nq, L, S = c.way * c.shot, c.way * q, c.seq_len, c.img_size
        sl = torch.arange(c.way).repeat_interleave(c.shot)[torch.randperm(ns, generator=g)].float()
        tl = torch.arange(c.way).repeat_interleave(q)[torch.randperm(nq, generator=g)].float()
        if str(self.device) != "cpu":
            gd = torch.Generator(device=self.device).manual_seed(self.base + self.rank * 10 ** 6 + e)
            rnd = lambda *s: torch.rand(*s, generator=gd, device=self.device)          # noqa: E731
            nrm = lambda *s: torch.randn(*s, generator=gd, device=self.device)         # noqa: E731
        else:
            rnd = lambda *s: torch.rand(*s, generator=g)                               # noqa: E731
            nrm = lambda *s: torch.randn(*s, generator=g)                              # noqa: E731
        d = {"support_set": rnd(ns * L, 3, S, S), "target_set": rnd(nq * L, 3, S, S),
             "support_set_feature_teacher": nrm(ns, L, 2048), "target_set_feature_teacher": nrm(nq, L, 2048),
             "support_labels": sl, "target_labels": tl, "real_target_labels": tl.clone(),
             "batch_class_list": torch.arange(c.way).float()}
        return {k: v.unsqueeze(0) for k, v in d.items()}

    def __iter__(self):
        for e in range(self.length):
            yield self.episode(e)
