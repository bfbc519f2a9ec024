"""Two (or more) batches in flight on one GPU.

One explanation step has phases with very different machine use: the encoder's activation chain and the reverse walk
are matrix-core bound and fill all 256 CUs, the decoder replay / decoder LRP in between are chains of small
latency-bound launches that leave most of the chip idle, and every launch of the walk ends in a partially filled last
round of workgroups.  `LRPPipeline` owns N independent `lrp_handle`s (own caches, own side stream) and issues
consecutive batches round-robin on N HIP streams, so batch i+1's encode / decoder phases run under batch i's walk.
Nothing is shared between the handles but the (read-only) caller tensors; results are bit-identical to a single handle.
[MI355X] 32 images x 10 words: 40.1 ms/step with one handle, 37.5 with two (three: no further gain), 24 GB of HBM each.
"""
import torch

from .engine import LRPEngine


class LRPPipeline(object):
    def __init__(self, n_handles=2, **engine_kwargs):
        if n_handles < 1:
            raise ValueError("n_handles must be >= 1")
        self.engines = [LRPEngine(**engine_kwargs) for _ in range(n_handles)]
        dev = self.engines[0].device
        self.streams = [torch.cuda.Stream(device=dev) for _ in range(n_handles)]
        self._done = [None] * n_handles          # event after the last batch issued on each slot
        self._next = 0

    @property
    def next_slot(self):
        """Index of the handle / stream the next `explain_batch` goes to (so the caller can pick that slot's output)."""
        return self._next

    def reset(self):
        """Wait for everything in flight and start again at slot 0."""
        self.synchronize()
        self._next = 0

    def set_weights(self, weights):
        for e in self.engines:
            e.set_weights(weights)

    def set_weights_from_device(self, weights):
        for e in self.engines:
            e.set_weights_from_device(weights)

    def set_precision(self, mode):
        for e in self.engines:
            e.set_precision(mode)

    def set_fast_layers(self, mask):
        for e in self.engines:
            e.set_fast_layers(mask)

    def explain_batch(self, images, captions, img_idx, tpos, out=None):
        """One step (encode -> decoder replay -> per-token heat-maps) on the next handle's stream; returns
        (out, slot).  The result is complete on `streams[slot]` only: before reading it on another stream call
        `wait(slot)` (stream-side wait on the batch's event, no host block) or `synchronize()`.  A result tensor
        allocated here (out=None) is registered with the caller's stream, so the caching allocator does not hand its
        memory out again while the caller still uses it.  The caller's input tensors must stay alive until then."""
        k = self._next
        self._next = (k + 1) % len(self.engines)
        eng, st = self.engines[k], self.streams[k]
        caller = torch.cuda.current_stream(eng.device)
        st.wait_stream(caller)                                        # inputs produced on the caller's stream
        with torch.cuda.stream(st):
            eng.encode_images(images)
            eng.decoder_forward(captions)
            res = eng.explain_tokens(img_idx, tpos, out=out)
            ev = torch.cuda.Event()
            ev.record(st)
        self._done[k] = ev
        if out is None:
            res[0].record_stream(caller)
        return res[0], k

    def wait(self, slot=None):
        """Make the CURRENT stream wait for the last batch of `slot` (default: of every slot) — what a consumer on
        another stream needs before touching the result; the host does not block."""
        cur = torch.cuda.current_stream(self.engines[0].device)
        for k in (range(len(self.engines)) if slot is None else [slot]):
            if self._done[k] is not None:
                cur.wait_event(self._done[k])

    def synchronize(self):
        for st in self.streams:
            st.synchronize()
