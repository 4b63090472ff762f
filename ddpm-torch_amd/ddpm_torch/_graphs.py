"""hipGraph plumbing shared by the captured training step and the captured sampling step.

A *segmented* graph is a list of hipGraphs captured back to back from ONE pass over a Python body, with an optional
host callback between consecutive segments.  The body receives ``cut(fn)``: it ends the segment being captured,
remembers ``fn`` and opens the next segment.  ``replay()`` then launches segment 0, calls its ``fn``, launches
segment 1, ... — which is how the data-parallel training step keeps its RCCL all-reduces OUT of the captured work
(they are issued eagerly between two segments, exactly where the eager backward issues them) while everything
else — ~1000 kernel launches per step — is replayed from a handful of graph launches.

All segments share one private memory pool and are always replayed in capture order on one stream, so activations
allocated in one segment and released in a later one keep their addresses from replay to replay.
"""
import gc

import torch

__all__ = ["SegmentedGraph"]


class SegmentedGraph:
    def __init__(self, device):
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.pool = torch.cuda.graph_pool_handle()
        self.segments = []              # [(CUDAGraph, callback or None)]
        self._open = None
        self._on_abort = None
        self._generators = []

    def register_generator(self, gen):
        """A non-default torch.Generator consumed inside the body (its state is advanced on every replay)."""
        if gen is not None:
            self._generators.append(gen)

    # ------------------------------------------------------------------ capture
    def _begin(self):
        g = torch.cuda.CUDAGraph()
        # Registering a generator with its first graph creates the generator's graph-state tensors (seed / offset words that every
        # capture and every replay then rewrites IN PLACE), and capture_begin registers the device's default generator the same way.
        # Created under torch.inference_mode (the sampler captures inside @inference_mode) they would be inference tensors, and the
        # next capture or replay from ordinary code — the training step after the first epoch's sample grid — would die with
        # "Inplace update to inference tensor outside InferenceMode".  So: always create them as ordinary tensors.
        with torch.inference_mode(False):
            for gen in self._generators:
                g.register_generator_state(gen)
            # thread_local: other host threads (the communicator's watchdog polls events) must not invalidate the capture
            g.capture_begin(pool=self.pool, capture_error_mode="thread_local")
        self._open = g

    def _end(self, callback):
        g, self._open = self._open, None
        g.capture_end()
        self.segments.append((g, callback))

    def _cut(self, callback):
        self._end(callback)
        self._begin()

    def capture(self, body, on_abort=None):
        """Run ``body(cut)`` once under stream capture (nothing executes).  On failure the partial capture is discarded and
        the exception propagates; the caller falls back to eager execution.  ``on_abort()``: called on the capture stream before the
        broken capture is ended — the body's owner joins whatever streams it had forked into the capture (a capture with unjoined work
        cannot be ended: the streams would stay in capture mode, the registered generators in their in-graph mode, and the caching
        allocator would keep routing allocations to this graph's pool)."""
        self._on_abort = on_abort
        assert not self.segments and self._open is None
        # (an allocator pool released in the middle of a capture — by the cyclic collector freeing an earlier graph or launch plan —
        #  trips an internal assertion of the caching allocator: collect first, keep the collector off while capturing)
        gc.collect()
        gc_was_on = gc.isenabled()
        gc.disable()
        try:
            return self._capture(body)
        finally:
            if gc_was_on:
                gc.enable()

    def _capture(self, body):
        torch.cuda.synchronize(self.device)
        cur = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            self._begin()
            try:
                body(self._cut)
                self._end(None)
            except BaseException:
                if self._open is not None:
                    if self._on_abort is not None:
                        try:
                            self._on_abort()
                        except Exception:
                            pass
                    try:
                        self._open.capture_end()
                    except Exception:
                        # the capture was invalidated: capture_end stops at hipStreamEndCapture and never tells the caching allocator
                        # that allocations are no longer routed to this graph's pool (the process would then abort at exit on the
                        # allocator's "captures_underway.empty()" assertion, after an otherwise successful fallback)
                        try:
                            torch._C._cuda_endAllocateToPool(self.device.index if self.device.index is not None else torch.cuda.current_device(), self.pool)
                        except Exception:
                            pass
                    self._open = None
                self.segments.clear()
                raise
        cur.wait_stream(self.stream)
        return self

    # ------------------------------------------------------------------ replay
    def replay(self):
        for g, callback in self.segments:
            g.replay()
            if callback is not None:
                callback()

    @property
    def launches(self):
        return len(self.segments)
