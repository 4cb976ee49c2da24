"""hipGraph capture of the refinement path for fixed shapes (launch-bound small batches).

One eval forward is a dozen kernel launches plus a few memsets issued from Python through ctypes; at the batch
sizes the reference's eval drivers use (tens of crops) the host issue time, not the GPU, sets the latency.
`CapturedRefine(model, *example_inputs)` records `model.refine(...)` once into a hipGraph (torch.cuda.CUDAGraph =
hipStreamBeginCapture on ROCm; lib3dal_hip.so only ever enqueues on the stream it is given and allocates
nothing, so its launches are captured as they are) and replays it per call. Requirements: eval mode, the device
sampler (the NumPy sampler needs a host round trip in the middle), fixed (B, N).
"""
import collections

import torch

from ._heads import Workspace


class CapturedRefine:
    """The recorded graph holds RAW pointers to the model's workspace and to its packed-weight blobs. Both are kept
    alive here for as long as the graph exists: the capture runs on a workspace of its own (an eager call with a
    larger batch may re-allocate the model's, never this one), and the blobs are referenced together with the
    PackedCache stamps they were built from. Every call compares the stamps first; if the weights changed
    (load_state_dict, an optimizer step, invalidate_packed()) the graph is recorded again instead of replaying the
    stale weights. `recaptures` counts how often that happened."""

    def __init__(self, model, *example_inputs):
        if model.training:
            raise RuntimeError("capture the eval-mode path: call model.eval() first")
        if getattr(model, "sampler", "device") != "device":
            raise RuntimeError("graph capture needs sampler='device' (the NumPy sampler synchronises with the host)")
        self.model = model
        # static input buffers in the callers' layout (same strides as the examples, e.g. point-major pts)
        self.inputs = [None if t is None else torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=t.device)
                       .copy_(t) for t in example_inputs]
        self._ws = Workspace()                              # the captured launches' own workspace
        self.recaptures = -1
        self._capture()

    def _capture(self):
        model = self.model
        saved_ws = model._ws
        model._ws = self._ws
        try:
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):                   # warm-up outside the capture: packs weights, sizes workspace
                model.refine(*self.inputs)
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.boxes = model.refine(*self.inputs)
        finally:
            model._ws = saved_ws
        self._stamps, self._blobs = model._cache.snapshot()     # the blobs the graph points into stay referenced
        self._ws_buf = self._ws.buf
        self._settings = (model.precision, model.seed, model.item_offset)
        self._keep = dict(model.last) if hasattr(model, "last") else None
        self.recaptures += 1

    def _stale(self):
        m = self.model
        if (m.precision, m.seed, m.item_offset) != self._settings:
            return True
        # the blobs themselves must be the ones the graph points into: after invalidate_packed() the cache re-packs
        # into NEW tensors whose stamps can equal the old ones (a .data write leaves pointer and version untouched)
        if any(m._cache._blob.get(k) is not v for k, v in self._blobs.items()):
            return True
        return m._cache._stamp != self._stamps or not m._cache.current()

    def __call__(self, *inputs):
        """copies the inputs into the captured buffers, replays, returns the (B,7) boxes (a buffer that the next
        call overwrites)"""
        for dst, src in zip(self.inputs, inputs):
            if dst is not None:
                dst.copy_(src)
        if self._stale():
            self._capture()
        self.graph.replay()
        return self.boxes


class StreamPipe:
    """Consecutive `model.refine()` calls on `depth` HIP streams, each with a workspace of its own.

    At the eval drivers' batch (64 crops x 4096 points, static_eval.py:299) one call is two big kernels and ten
    small ones (the per-crop FC layers, the sampler, the fills, the box decode: 4-27 us each, 86 us of a 1.55 ms
    call) during which most of the chip idles, and the big kernels' last wave of workgroups leaves CUs empty too.
    Batches are independent, so the next call can fill those holes: `submit()` enqueues a call on the next stream
    (it waits for what the caller's stream has produced so far: the inputs), `collect(keep)` hands back finished
    boxes in submission order after making the caller's stream wait for them, leaving `keep` calls in flight.

        pipe = StreamPipe(model)                      # eval mode; any sampler that needs no host round trip
        for batch in batches:
            pipe.submit(*batch)
            for boxes in pipe.collect(keep=1): ...
        for boxes in pipe.collect(): ...
    """

    def __init__(self, model, depth=2):
        if model.training:
            raise RuntimeError("StreamPipe runs the eval-mode path: call model.eval() first")
        if getattr(model, "sampler", "device") != "device":
            raise RuntimeError("StreamPipe needs sampler='device' (the NumPy sampler synchronises with the host)")
        if depth < 1:
            raise ValueError("depth must be >= 1")
        self.model = model
        self.streams = [torch.cuda.Stream() for _ in range(depth)]
        self.workspaces = [Workspace() for _ in range(depth)]
        self.pending = collections.deque()
        self.submitted = 0

    def submit(self, *inputs):
        k = self.submitted % len(self.streams)
        stream, model = self.streams[k], self.model
        stream.wait_stream(torch.cuda.current_stream())
        saved = model._ws
        model._ws = self.workspaces[k]                      # (a call still running on the other stream uses the other one)
        try:
            with torch.cuda.stream(stream):
                boxes = model.refine(*inputs)
                done = torch.cuda.Event()
                done.record(stream)
        finally:
            model._ws = saved
        for t in inputs:
            if torch.is_tensor(t) and t.is_cuda:
                t.record_stream(stream)                     # the caller may free them while the call is in flight
        self.pending.append((boxes, done))
        self.submitted += 1

    def collect(self, keep=0):
        out = []
        cur = torch.cuda.current_stream()
        while len(self.pending) > keep:
            boxes, done = self.pending.popleft()
            cur.wait_event(done)
            boxes.record_stream(cur)
            out.append(boxes)
        return out


class CapturedTrainStep:
    """One whole training step — forward, criterion, backward, optimizer — as a hipGraph.

    At the train drivers' batch the step issues several hundred launches (per layer: GEMM, two-stage reductions,
    per-channel epilogues; the optimizer; the criterion) and a third of its wall time is their host-side issue.
    Capturing the step once and replaying it removes that. Requirements: the device sampler (no host round trip),
    fixed shapes, an optimizer built with `capturable=True` (torch.optim.Adam supports it), and a criterion made of
    torch ops. `step_fn(*inputs)` must run forward + criterion and return the scalar loss; backward and
    `optimizer.step()` are added here. Capture BEFORE the parameters take part in any eager backward (PyTorch binds a
    parameter's gradient accumulator to the stream of its first backward; one bound to the default stream cannot be
    used from the capture stream). The object-point draw and the Dropout draw are part of the recording: Dropout
    advances with torch's graph-safe RNG on every replay, the device sampler's key is a kernel argument and stays
    what it was at capture time.

        cap = CapturedTrainStep(model, optimizer, step_fn, *example_inputs)
        loss = cap(*batch)            # copies the batch into the captured buffers, replays, returns the loss tensor

    optimizer_in_graph=False (round 5): forward, criterion and backward are the graph; `optimizer.step()` runs eagerly
    behind every replay, with the optimizer exactly as the train drivers build it (static_train.py:220, no
    `capturable=True`). Why: a capturable Adam keeps its step counters on the device and computes the two bias
    corrections with `_foreach_pow(scalar, [0-dim step tensors])`, which has no multi-tensor path — 120 single-element
    kernels per step (one per parameter and beta) plus 8 more multi-tensor launches than the plain optimizer; inside a
    replay every kernel node costs at least ~4.5 us of stream time, so the captured step was SLOWER than the eager one
    (6.93 vs 6.55 ms; profiles/r05_train_graph.json: 324 vs 185 dispatches, 0.75 ms of them those kernels). With the
    optimizer outside, the replay is the eager step's kernels without its launch gaps and the host issues one replay
    and the optimizer's ~14 launches per step instead of ~185.
    """

    def __init__(self, model, optimizer, step_fn, *example_inputs, warmup=3, optimizer_in_graph=True):
        if not model.training:
            raise RuntimeError("capture the train-mode path: call model.train() first")
        if getattr(model, "sampler", "device") != "device":
            raise RuntimeError("graph capture needs sampler='device' (the NumPy sampler synchronises with the host)")
        self.inputs = [None if t is None else torch.empty_strided(t.shape, t.stride(), dtype=t.dtype, device=t.device)
                       .copy_(t) for t in example_inputs]
        self.optimizer, self.optimizer_in_graph = optimizer, optimizer_in_graph
        # optimizer outside the graph: the warm-up runs forward + backward with NO optimizer step, so it must leave no
        # trace an eager run would not have — the BatchNorm running statistics / num_batches_tracked and the
        # device-resident draw counters (train.draw_step) it advances are put back afterwards. With the optimizer inside,
        # the warm-up iterations are real training steps on the example batch and everything moves together.
        keep = None
        if not optimizer_in_graph:
            mods = list(model.modules())
            keep = ([(b, b.detach().clone()) for b in model.buffers()],
                    [(m.__dict__["_draw_step"], m.__dict__["_draw_step"].clone()) for m in mods if "_draw_step" in m.__dict__], mods)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):                                  # also sizes every scratch buffer
                optimizer.zero_grad(set_to_none=True)
                step_fn(*self.inputs).backward()
                if optimizer_in_graph:
                    optimizer.step()
            if keep is not None:
                with torch.no_grad():
                    for t, was in keep[0] + keep[1]:
                        t.copy_(was)
                    for m in keep[2]:                                # counters the warm-up itself created: back to their start
                        if "_draw_step" in m.__dict__ and not any(m.__dict__["_draw_step"] is t for t, _ in keep[1]):
                            m.__dict__["_draw_step"].zero_()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad(set_to_none=True)
        with torch.cuda.graph(self.graph):
            self.loss = step_fn(*self.inputs)
            self.loss.backward()                                     # (gradients: the graph's own buffers, rewritten by every replay)
            if optimizer_in_graph:
                optimizer.step()

    def __call__(self, *inputs):
        for dst, src in zip(self.inputs, inputs):
            if dst is not None:
                dst.copy_(src)
        self.graph.replay()
        if not self.optimizer_in_graph:
            self.optimizer.step()
        return self.loss
