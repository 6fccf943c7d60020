"""The MJPEG sink of the Streamer modes, on the device.

Mirror of the two encode sites of the reference's ``MJPEGStreamer`` (reference streamer.py:249-256
``_encoder_loop`` and 285-291 ``encode_jpeg``): there the float32 HWC frame ``make_sbs`` copied to the host is
channel-flipped and handed to ``cv2.imencode('.jpg', bgr, [IMWRITE_JPEG_QUALITY, quality])``.  Here the frame
stays on the GPU: ``encode_jpeg`` takes what ``make_sbs`` / ``pipeline`` produced (uint8 or float32 0..255, RGB,
HWC) and returns the JPEG bytes; only those bytes (≈1-3 MB instead of a 25-50 MB float frame) cross PCIe.
Same names, argument meaning and failure behaviour as the reference's method (``b""`` for ``None``).
"""
from __future__ import annotations

from typing import List, Optional

import numpy as np
import torch

from . import _lib, ops


def _to_device(arr) -> torch.Tensor:
    t = torch.from_numpy(np.ascontiguousarray(arr)) if isinstance(arr, np.ndarray) else arr
    if t.dtype not in (torch.uint8, torch.float32):
        t = t.float()
    if not t.is_cuda:
        if not torch.cuda.is_available():
            raise _lib.D2SError("encode_jpeg needs a ROCm device (no CPU path in this package)")
        t = t.cuda()
    return t


def encode_jpeg_batch(frames, quality: int = 90) -> List[bytes]:
    """[B,H,W,3] RGB frames -> list of JPEG byte strings (one host copy of the packed bytes)."""
    t = _to_device(frames)
    out, sizes = ops.jpeg_encode(t, quality)
    sz = sizes.cpu().tolist()
    if any(s < 0 for s in sz):                                        # incompressible frame: retry with the hard bound
        out, sizes = ops.jpeg_encode(t, quality, out_stride=ops.jpeg_bound(t.shape[-3], t.shape[-2])[0])
        sz = sizes.cpu().tolist()
    n = max(sz)
    host = out[:, :n].cpu().numpy()
    return [host[b, :s].tobytes() for b, s in enumerate(sz)]


def encode_jpeg(arr, quality: int = 90) -> bytes:
    """``MJPEGStreamer.encode_jpeg`` (reference streamer.py:285-291): one HWC RGB frame -> JPEG bytes."""
    if arr is None:
        return b""
    return encode_jpeg_batch(_to_device(arr).unsqueeze(0), quality)[0]


class MJPEGEncoder:
    """The encode half of ``MJPEGStreamer`` (reference streamer.py:37-44, 230-257): fixed quality, latest-frame
    semantics.  The reference encodes on a separate thread (``_encoder_loop``) while the main loop produces the next
    frame; here the encode is queued on the encoder's own HIP stream, so its small kernels fill the gaps of the next
    frame's model pass instead of extending the frame time.  The HTTP half is out of scope."""

    def __init__(self, quality: int = 90):
        self.quality = int(quality)
        self.encoded_frame: Optional[bytes] = None
        self._stream: Optional[torch.cuda.Stream] = None
        self._ws: Optional[torch.Tensor] = None        # this encoder's own scratch: it runs beside other streams' encodes
        self._pending = None

    def set_frame(self, frame) -> None:
        """Queue the encode of one HWC RGB frame (device tensor or numpy) behind the work already issued on the current
        stream; returns at once.  The frame's memory must stay untouched until ``wait()`` (or the next ``set_frame``)."""
        t = _to_device(frame)
        if t.dim() != 3:
            raise ValueError("MJPEGEncoder.set_frame takes ONE HWC frame (the reference's encoder thread holds one raw_frame, "
                             "streamer.py:230-257); use encode_jpeg_batch for a batch")
        if self._stream is None or self._stream.device != t.device:
            self._stream = torch.cuda.Stream(device=t.device)
            self._ws = None
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream(t.device))
        need = ops.jpeg_bound(t.shape[-3], t.shape[-2])[1] + 256
        with torch.cuda.stream(self._stream):
            if self._ws is None or self._ws.numel() < need:
                # (re)allocated on the ENCODER's stream: the old scratch may still be read by the encode queued there, and the
                # caching allocator only reuses a block for work ordered behind the stream it was allocated on
                self._ws = torch.empty(need, dtype=torch.uint8, device=t.device)
            self._stream.wait_event(ready)
            out, sizes = ops.jpeg_encode(t, self.quality, workspace=self._ws)
            done = torch.cuda.Event()
            done.record()
        t.record_stream(self._stream)
        self._pending = (t, out, sizes, done)

    def busy_event(self) -> Optional[torch.cuda.Event]:
        """Event that fires when the queued encode has finished reading its frame (wait on it before overwriting it)."""
        return self._pending[3] if self._pending else None

    def wait(self) -> Optional[bytes]:
        """Finish the queued encode and publish it as ``encoded_frame`` (what ``_generate`` serves, streamer.py:259-283)."""
        if self._pending is None:
            return self.encoded_frame
        t, out, sizes, done = self._pending
        self._pending = None
        done.synchronize()
        n = int(sizes[0])
        if n < 0:                                                       # incompressible frame: hard bound, synchronously
            self.encoded_frame = encode_jpeg(t, self.quality)
        else:
            self.encoded_frame = out[0, :n].cpu().numpy().tobytes()
        return self.encoded_frame
