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


class _DevMem:
    """`nbytes` of device memory at `ptr` as a __cuda_array_interface__ object (torch.as_tensor aliases it, no copy)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


class PresentRing:
    """shape / dtype: layout of a frame.  With a shape the ring allocates its own slots (torch tensors standing in for the
    consumer's buffers); shape=None: slots are bound later (`bind_gl_buffer`, or `bind` of a consumer-owned pointer)."""

    def __init__(self, shape: Optional[Tuple[int, ...]] = None, dtype=torch.uint8, slots: int = 3, device: int = 0):
        if not torch.cuda.is_available():
            raise _lib.D2SError("PresentRing needs a ROCm device")
        self.lib = _lib.load()
        self.device = torch.device("cuda", device)
        self.shape, self.dtype = (tuple(shape) if shape is not None else None), dtype
        self._h = C.c_void_p()
        check(self.lib.d2s_present_create(device, slots, C.byref(self._h)), "d2s_present_create")
        # the consumer's buffers: the library only borrows the pointers.  None = bound to a GL buffer / not bound yet
        self.buffers = [None] * slots
        if shape is not None:
            for i in range(slots):
                self.bind(i, torch.empty(shape, dtype=dtype, device=self.device))

    def bind(self, slot: int, buf: torch.Tensor):
        """Lend a consumer-owned device tensor to slot `slot` (kept alive by the ring)."""
        if not buf.is_cuda or not buf.is_contiguous():
            raise ValueError("PresentRing.bind: a contiguous ROCm device tensor")
        check(self.lib.d2s_present_bind(self._h, slot, C.c_void_p(buf.data_ptr()), buf.numel() * buf.element_size()), "d2s_present_bind")
        self.buffers[slot] = buf

    def bind_gl_buffer(self, slot: int, gl_buffer: int):
        """Register an OpenGL buffer object for slot `slot` (hipGraphicsGLRegisterBuffer, WRITE_DISCARD -- what the
        reference's CUDART_GL.register_buffer does, viewer.py:287-300).  Needs a current GL context; raises otherwise.
        The buffer is mapped while the producer owns the slot (acquire .. publish) and unmapped, stream-ordered, at publish."""
        check(self.lib.d2s_present_bind_gl_buffer(self._h, slot, int(gl_buffer)), "d2s_present_bind_gl_buffer")
        self.buffers[slot] = None                   # acquire() hands out a view of the mapped pointer instead

    def _view(self, slot: int, ptr: Optional[int], nbytes: int) -> Optional[torch.Tensor]:
        """The tensor the caller reads / writes for `slot`: the bound tensor, or a view over the pointer the library reports
        (the mapped PBO of a GL-bound slot), shaped like a frame when the ring knows the frame layout."""
        if self.buffers[slot] is not None:
            return self.buffers[slot]
        if not ptr:
            return None                             # GL-bound slot after publish: the consumer sources the GL buffer itself
        t = torch.as_tensor(_DevMem(ptr, nbytes), device=self.device)
        if self.shape is not None:
            n = 1
            for d in self.shape:
                n *= d
            n *= torch.empty((), dtype=self.dtype).element_size()
            if n > nbytes:
                raise _lib.D2SError(f"slot {slot}: the GL buffer holds {nbytes} bytes, a frame needs {n}")
            t = t[:n].view(self.dtype).view(self.shape)
        return t

    def acquire(self, stream: Optional[torch.cuda.Stream] = None) -> Tuple[int, torch.Tensor]:
        st = stream or torch.cuda.current_stream(self.device)
        slot, ptr, n = C.c_int(), C.c_void_p(), C.c_uint64()
        check(self.lib.d2s_present_acquire(self._h, C.c_void_p(st.cuda_stream), C.byref(slot), C.byref(ptr), C.byref(n)), "d2s_present_acquire")
        try:
            return slot.value, self._view(slot.value, ptr.value, n.value)
        except Exception:
            self.cancel(slot.value, st)            # e.g. the mapped GL buffer is smaller than a frame: the slot must not stay acquired
            raise

    def cancel(self, slot: int, stream: Optional[torch.cuda.Stream] = None):
        """Give an acquired slot back without publishing it (the producer failed between acquire and publish)."""
        st = stream or torch.cuda.current_stream(self.device)
        check(self.lib.d2s_present_cancel(self._h, slot, C.c_void_p(st.cuda_stream)), "d2s_present_cancel")

    def produce(self, fn, stream: Optional[torch.cuda.Stream] = None):
        """acquire -> fn(slot_tensor) -> publish, cancelling the slot if fn raises.  Returns the slot index."""
        slot, buf = self.acquire(stream)
        try:
            fn(buf)
        except Exception:
            self.cancel(slot, stream)
            raise
        self.publish(slot, stream)
        return slot

    def publish(self, slot: int, stream: Optional[torch.cuda.Stream] = None):
        st = stream or torch.cuda.current_stream(self.device)
        check(self.lib.d2s_present_publish(self._h, slot, C.c_void_p(st.cuda_stream)), "d2s_present_publish")

    def consume(self, stream: Optional[torch.cuda.Stream] = None, host_wait: bool = False) -> Tuple[int, Optional[torch.Tensor], int]:
        """Latest published slot; `stream` (default: current) waits for its ready event on the device, or the host does.
        The tensor is None for a GL-bound slot (unmapped at publish: the consumer reads the GL buffer object)."""
        st = C.c_void_p(-1 & 0xFFFFFFFFFFFFFFFF) if host_wait else C.c_void_p((stream or torch.cuda.current_stream(self.device)).cuda_stream)
        slot, ptr, seq = C.c_int(), C.c_void_p(), C.c_uint64()
        check(self.lib.d2s_present_consume(self._h, st, C.byref(slot), C.byref(ptr), C.byref(seq)), "d2s_present_consume")
        buf = self.buffers[slot.value]
        return slot.value, buf, seq.value

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
