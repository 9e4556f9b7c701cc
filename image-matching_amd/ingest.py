"""Pipelined image ingest for the matching hot path (datasets/SSHIDataset.py:14-29 does
imread -> cv2.resize -> /255 on the host and a blocking `.to(device)` per pair, superpoint_glue_test.py:74-75).

Here the host only hands over the decoded uint8 images: they are packed into a pinned staging buffer, copied
host->device on a side stream while the previous batch is being matched, and resized + normalised by the
`resize_u8_unit` kernel on that same stream.  `depth` staging slots; an event per slot hands the fp32 batch to the
compute stream, a second event returns the slot once the consumer is done with it."""
import torch


class IngestPipeline:
    def __init__(self, engine, batch, src_hw, dst_hw=None, depth=2):
        self.engine, self.B, self.src_hw = engine, batch, tuple(src_hw)
        self.dst_hw = tuple(dst_hw) if dst_hw is not None else self.src_hw
        dev = engine.device
        self.copy_stream = torch.cuda.Stream(dev)
        Hs, Ws = self.src_hw
        H, W = self.dst_hw
        self.slots = [{
            "pinned": torch.empty(batch, Hs, Ws, dtype=torch.uint8).pin_memory(),
            "dev_u8": torch.empty(batch, Hs, Ws, dtype=torch.uint8, device=dev),
            "out": torch.empty(batch, 1, H, W, dtype=torch.float32, device=dev),
            "ready": torch.cuda.Event(), "free": torch.cuda.Event(), "used": False,
        } for _ in range(depth)]
        self.next = 0

    def staging(self):
        """The next slot's pinned (batch,Hs,Ws) uint8 buffer as a numpy view, so a decoder can write frames straight
        into page-locked memory (then call `submit_staged(n)`); blocks until that slot's previous batch was consumed."""
        slot = self.slots[self.next]
        if slot["used"]:
            slot["free"].synchronize()             # the consumer of this slot's previous batch has finished
        return slot["pinned"].numpy()

    def submit(self, images):
        """images: sequence of <= batch uint8 (Hs,Ws) numpy arrays / tensors.  Returns a ticket for `take`."""
        n = len(images)
        if n > self.B:
            raise ValueError(f"{n} images for a batch of {self.B}")
        self.staging()
        pinned = self.slots[self.next]["pinned"]
        for i, im in enumerate(images):
            t = im if torch.is_tensor(im) else torch.from_numpy(im)
            if tuple(t.shape) != self.src_hw or t.dtype != torch.uint8:
                raise ValueError(f"image {i}: expected uint8 {self.src_hw}, got {t.dtype} {tuple(t.shape)}")
            pinned[i].copy_(t)
        return self.submit_staged(n)

    def submit_staged(self, n):
        """Ship the first n frames of the buffer `staging()` returned."""
        slot = self.slots[self.next]
        self.next = (self.next + 1) % len(self.slots)
        with torch.cuda.stream(self.copy_stream):
            slot["dev_u8"][:n].copy_(slot["pinned"][:n], non_blocking=True)
            self.engine.ingest(slot["dev_u8"][:n], self.dst_hw, out=slot["out"][:n])
            slot["ready"].record(self.copy_stream)
        slot["used"], slot["n"] = True, n
        return slot

    def take(self, ticket):
        """Make the current (compute) stream wait for the batch; returns (n,1,H,W) float32."""
        torch.cuda.current_stream(self.engine.device).wait_event(ticket["ready"])
        return ticket["out"][:ticket["n"]]

    def release(self, ticket):
        """Call after the last kernel that reads the batch has been enqueued on the current stream."""
        ticket["free"].record(torch.cuda.current_stream(self.engine.device))
