"""Launch plans: the training step recorded once as a table of C-ABI calls and re-issued from C (``csrc/plan.hip``).

``Trainer.step`` (utils/train.py:148-170 upstream) is a fixed sequence of ~490 launches on two streams.  Issued from Python it costs
6-7 ms of interpreter + ctypes time per ~9.4 ms GPU step, so a slower host makes the step host-bound; a replayed hipGraph has no host
cost but its executor serialises the two stream branches.  A plan keeps the eager step exactly as it is — same
entry points, same arguments, same two ``hipStream_t`` s, RCCL calls issued from the host between segments — and walks it in C:
one ``ddpm_plan_run`` per segment.

Recording: ``LaunchPlan.record(body)`` runs ``body(cut)`` ONCE, eagerly (the recorded step is a real step), while ``_hip.call`` appends a
copy of every call.  ``cut(fn)`` closes the current segment: ``fn`` (a communicator call) runs now and again at that point of every replay.
Everything a recorded call addresses must keep its address:
* on the GPU the body runs inside a private allocator pool (``torch.cuda.MemPool``) owned by the plan, so the activations / gradients it
  allocated and released keep their memory reserved for the replays — nothing else can be handed those addresses;
* tensors the engine allocates are also registered through ``_hip.retain`` and kept by the plan when there is no pool (host-emulated
  runs in the CPU test-suite, where replay loops over the recorded calls in Python through the same ``_hip._invoke``).
Step-varying scalars must be device-resident (learning rate / bias corrections / EMA weight / dropout seed word: ``hyper_dev``), the
contract the captured-graph form already imposed.  Anything in the body that is NOT a ``_hip.call`` (a torch op) would silently be left
out of the replays: the direct step draws (t, noise) outside the recorded region and has no other torch arithmetic
(``tests/test_configs_gpu.py::test_captured_training_step_equals_the_eager_step`` holds replayed steps to eager steps; the host-emulated
suite checks that a recorded body contains no torch arithmetic).
"""
import contextlib
import ctypes
import gc
import struct

import torch

from . import _hip

__all__ = ["LaunchPlan"]

_MASK64 = (1 << 64) - 1


def _words(name, args):
    """Arguments of one call as the 64-bit words ddpm_plan_append takes (see include/ddpm_hip.h)."""
    types = _hip.PROTOTYPES[name]
    if len(types) != len(args):
        raise TypeError(f"{name}: {len(args)} arguments recorded, {len(types)} declared")
    out = []
    for ty, v in zip(types, args):
        if ty is _hip.F:
            out.append(struct.unpack("<I", struct.pack("<f", float(v)))[0])
        elif ty is _hip.I:
            out.append(int(v) & 0xFFFFFFFF)
        else:                                   # pointers (None = NULL), 64-bit integers
            out.append((0 if v is None else int(v)) & _MASK64)
    return out


class LaunchPlan:
    def __init__(self, device):
        self.device = torch.device(device)
        self.segments = [[]]            # [[(entry point, args), ...], ...]
        self.callbacks = []             # callbacks[i] runs after segment i
        self.keep = []                  # tensors whose addresses the entries hold (see _hip.retain)
        self.retains = self.device.type != "cuda"     # on the GPU the private pool keeps the memory: holding every tensor alive until the
                                                       # recording ends would make the pool as large as the SUM of the step's allocations
        self.pool = None
        self._c = None                  # handle of the C-side copy (GPU only)
        self.recorded = False
        self.build_error = None         # why the C-side copy could not be built (then the plan must not be replayed)

    # ------------------------------------------------------------------ recording
    def add(self, name, args):
        self.segments[-1].append((name, args))

    def cut(self, fn):
        """End the segment being recorded; ``fn`` runs here now and at this point of every replay."""
        self.callbacks.append(fn)
        self.segments.append([])
        fn()

    def record(self, body):
        assert not self.recorded
        on_gpu = self.device.type == "cuda"
        if on_gpu:
            self.pool = torch.cuda.MemPool()
            scope = torch.cuda.use_mem_pool(self.pool, device=self.device)
        else:
            scope = contextlib.nullcontext()
        # Releasing an allocator pool (an earlier plan's, a captured graph's) while allocations are routed to another one trips an
        # internal assertion of the caching allocator — and such owners usually sit in reference cycles (trainer <-> step object), so
        # it is the cyclic collector that frees them, whenever it happens to run.  Collect NOW, then keep the collector off for the
        # duration of the recording (what torch.cuda.graph() does around a capture, for the same reason).
        gc.collect()
        gc_was_on = gc.isenabled()
        gc.disable()
        prev = _hip.record_into(self)
        try:
            with scope:
                body(self.cut)
        finally:
            _hip.record_into(prev)
            if gc_was_on:
                gc.enable()
        if on_gpu:
            try:
                # The recorded addresses stay valid only while the private pool keeps every block the body allocated and released
                # (retains is False on the GPU).  That rests on torch.cuda.MemPool holding its own reference after use_mem_pool's scope
                # (true for torch 2.10): checked here, so that a build where the scope's exit drops the last reference fails LOUDLY — the
                # plan is marked unusable and the step stays eager — instead of replaying into memory that went back to the allocator.
                count = getattr(self.pool, "use_count", None)
                if callable(count) and count() < 1:
                    raise RuntimeError("the plan's MemPool lost its last reference when the recording scope closed")
                self._build_native()
            except Exception as e:           # (the body has run: the step is complete either way)
                self.build_error = f"{type(e).__name__}: {e}"
        self.recorded = True
        return self

    def _build_native(self):
        lib = _hip.lib()
        h = lib.ddpm_plan_create()
        if not h:
            raise RuntimeError("ddpm_plan_create failed")
        self._c = ctypes.c_void_p(h)
        for i, seg in enumerate(self.segments):
            for name, args in seg:
                w = _words(name, args)
                arr = (ctypes.c_ulonglong * len(w))(*w)
                rc = lib.ddpm_plan_append(self._c, name.encode(), ctypes.cast(arr, ctypes.c_void_p), len(w))
                if rc < 0:
                    raise RuntimeError(f"ddpm_plan_append({name}) failed: the executor does not know this entry point with {len(w)} arguments")
            if i + 1 < len(self.segments):
                lib.ddpm_plan_cut(self._c)

    # ------------------------------------------------------------------ replay
    def replay(self):
        if self._c is not None:
            lib = _hip.lib()
            for i in range(len(self.segments)):
                rc = lib.ddpm_plan_run(self._c, i)
                if rc != 0:
                    idx = ctypes.c_int(-1)
                    name = lib.ddpm_plan_failed_entry(self._c, ctypes.cast(ctypes.pointer(idx), ctypes.c_void_p))
                    raise RuntimeError(f"launch plan: {name.decode() if name else '?'} (entry {idx.value}) failed: {_hip._ERR.get(rc, rc)}")
                if i < len(self.callbacks):
                    self.callbacks[i]()
        else:
            for i, seg in enumerate(self.segments):
                for name, args in seg:
                    _hip._invoke(name, args)
                if i < len(self.callbacks):
                    self.callbacks[i]()

    @property
    def launches(self):
        return sum(len(s) for s in self.segments)

    def __del__(self):
        c, self._c = getattr(self, "_c", None), None
        if c is not None:
            try:
                _hip.lib().ddpm_plan_destroy(c)
            except Exception:
                pass
