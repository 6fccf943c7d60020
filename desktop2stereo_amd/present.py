"""Device-resident hand-off of produced frames to the display path.

The reference's viewer uploads every frame to OpenGL with a host synchronise plus a device-to-device copy into a mapped
PBO (reference viewer.py:1584-1712, 2399-2428).  ``PresentRing`` is the replacement seam: the consumer owns a ring of
device buffers (here torch tensors; in a GL application the mapped pointers of its PBOs, or GL buffer ids through
``bind_gl_buffer``), the producer writes its output straight into the slot it acquired (``out=`` of ``Engine.pipeline`` /
``ops.make_sbs``) and publishes it; waits are HIP events between the two streams (libd2s_hip.so ``d2s_present_*``).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import check


class PresentRing:
    def __init__(self, shape: Tuple[int, ...], dtype=torch.uint8, slots: int = 3, device: int = 0):
        if not torch.cuda.is_available():
            raise _lib.D2SError("PresentRing needs a ROCm device")
        self.lib = _lib.load()
        self.device = torch.device("cuda", device)
        self._h = C.c_void_p()
        check(self.lib.d2s_present_create(device, slots, C.byref(self._h)), "d2s_present_create")
        # the consumer's buffers (stand-ins for mapped PBOs): the library only borrows the pointers
        self.buffers = [torch.empty(shape, dtype=dtype, device=self.device) for _ in range(slots)]
        for i, b in enumerate(self.buffers):
            check(self.lib.d2s_present_bind(self._h, i, C.c_void_p(b.data_ptr()), b.numel() * b.element_size()), "d2s_present_bind")

    def bind_gl_buffer(self, slot: int, gl_buffer: int):
        """Register an OpenGL buffer object for slot `slot` (hipGraphicsGLRegisterBuffer, WRITE_DISCARD -- what the
        reference's CUDART_GL.register_buffer does, viewer.py:287-300).  Needs a current GL context; raises otherwise."""
        check(self.lib.d2s_present_bind_gl_buffer(self._h, slot, int(gl_buffer)), "d2s_present_bind_gl_buffer")

    def acquire(self, stream: Optional[torch.cuda.Stream] = None) -> Tuple[int, torch.Tensor]:
        st = stream or torch.cuda.current_stream(self.device)
        slot, ptr, n = C.c_int(), C.c_void_p(), C.c_uint64()
        check(self.lib.d2s_present_acquire(self._h, C.c_void_p(st.cuda_stream), C.byref(slot), C.byref(ptr), C.byref(n)), "d2s_present_acquire")
        return slot.value, self.buffers[slot.value]

    def publish(self, slot: int, stream: Optional[torch.cuda.Stream] = None):
        st = stream or torch.cuda.current_stream(self.device)
        check(self.lib.d2s_present_publish(self._h, slot, C.c_void_p(st.cuda_stream)), "d2s_present_publish")

    def consume(self, stream: Optional[torch.cuda.Stream] = None, host_wait: bool = False) -> Tuple[int, torch.Tensor, int]:
        """Latest published slot; `stream` (default: current) waits for its ready event on the device, or the host does."""
        st = C.c_void_p(-1 & 0xFFFFFFFFFFFFFFFF) if host_wait else C.c_void_p((stream or torch.cuda.current_stream(self.device)).cuda_stream)
        slot, ptr, seq = C.c_int(), C.c_void_p(), C.c_uint64()
        check(self.lib.d2s_present_consume(self._h, st, C.byref(slot), C.byref(ptr), C.byref(seq)), "d2s_present_consume")
        return slot.value, self.buffers[slot.value], seq.value

    def release(self, slot: int, stream: Optional[torch.cuda.Stream] = None):
        st = stream or torch.cuda.current_stream(self.device)
        check(self.lib.d2s_present_release(self._h, slot, C.c_void_p(st.cuda_stream)), "d2s_present_release")

    def close(self):
        if getattr(self, "_h", None):
            self.lib.d2s_present_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
