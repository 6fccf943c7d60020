"""GPU: f4, the device-resident hand-off to the display path (d2s_present_*, reference viewer.py:1584-1712, 2399-2428).
The producer writes d2s_pipeline's output straight into consumer-owned ring slots; a consumer on its own stream and host
thread reads the latest published slot.  No host synchronisation on the producer side, no extra copy."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
torch = pytest.importorskip("torch")


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.fail("gpu test selected but no ROCm device is visible")
    return torch.device("cuda", 0)


def test_present_ring_producer_consumer(dev):
    from desktop2stereo_amd import ops, synth
    from desktop2stereo_amd.config import MODELS, PipelineParams, engine_shape
    from desktop2stereo_amd.present import PresentRing
    from desktop2stereo_amd.weights import make_weights
    cfg = MODELS["tiny"]
    H, W, res = 270, 480, 140
    h, w, _ = engine_shape(H, W, res)
    p = PipelineParams(depth_resolution=res)
    sp = ops.sbs_params(p.ipd, p.depth_strength, p.convergence, "Full-SBS", True)
    oh, ow = ops.sbs_shape(H, W, sp)
    eng = ops.Engine(cfg, make_weights(cfg, 0), h, w, 1, "fp32")
    frames = [torch.from_numpy(synth.structured_frame(H, W, s)[None]).to(dev) for s in range(6)]
    want = [eng.pipeline(f, p, sp).clone() for f in frames]                     # ordinary call: library-independent output tensor
    torch.cuda.synchronize()        # the engine is one stream at a time: its workspaces must be idle before `prod` (non-blocking) uses it
    ring = PresentRing((1, oh, ow, 3), torch.uint8, slots=3)        # triple buffering: {held by the consumer, latest published, being written}
    with pytest.raises(Exception):
        ring.consume()                                                           # nothing published yet: loud
    prod, cons = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    seen, errors = {}, []
    published = threading.Semaphore(0)

    def consumer():
        try:
            got_last = 0
            while got_last < len(frames):
                published.acquire()
                slot, buf, seq = ring.consume(cons)                              # device-side wait on the slot's ready event
                with torch.cuda.stream(cons):
                    seen[seq] = buf.clone()                                      # the "display": reads the slot on its own stream
                ring.release(slot, cons)
                got_last = max(got_last, seq)
        except Exception as e:                                                   # surfaced in the main thread
            errors.append(e)

    t = threading.Thread(target=consumer)
    t.start()
    with torch.cuda.stream(prod):
        for f in frames:
            slot, buf = ring.acquire(prod)                                       # waits (on the device) for the consumer's release
            eng.pipeline(f, p, sp, out=buf)                                      # written in place: no copy
            ring.publish(slot, prod)
            published.release()
    t.join(60)
    assert not t.is_alive() and not errors, errors
    torch.cuda.synchronize()
    assert seen, "consumer saw nothing"
    for seq, buf in seen.items():                                                # latest-frame semantics: every frame it saw is intact
        assert torch.equal(buf, want[seq - 1]), seq
    assert max(seen) == len(frames)
    # host-wait form (a GL consumer before it sources the PBO)
    slot, buf, seq = ring.consume(host_wait=True)
    assert seq == len(frames) and torch.equal(buf, want[-1])
    # slot states: a held slot cannot be re-bound or published, an unacquired slot cannot be published, release needs a held slot
    with pytest.raises(Exception):
        ring.publish(slot)
    with pytest.raises(Exception):
        ring.bind(slot, torch.empty((1, oh, ow, 3), dtype=torch.uint8, device=dev))
    ring.release(slot)
    with pytest.raises(Exception):
        ring.release(slot)
    # two acquires without a publish in between hand out two different slots
    s1, b1 = ring.acquire()
    s2, b2 = ring.acquire()
    assert s1 != s2 and b1.data_ptr() != b2.data_ptr()
    ring.publish(s1); ring.publish(s2)
    # a producer that fails between acquire and publish gives the slot back (d2s_present_cancel): the ring never loses slots.  With
    # one slot held by the consumer, far more failures than slots in a row must leave the ring usable, and a cancelled slot is never
    # consumable (the consumer keeps seeing the last PUBLISHED frame).
    slot_c, buf_c, seq_c = ring.consume(host_wait=True)
    for _ in range(8):
        with pytest.raises(ZeroDivisionError):
            ring.produce(lambda b: 1 / 0)
    s3, _ = ring.acquire()
    ring.cancel(s3)
    with pytest.raises(Exception):
        ring.cancel(s3)                                                          # not acquired any more
    with pytest.raises(Exception):
        ring.publish(s3)                                                         # ... and not publishable
    ring.release(slot_c)
    assert ring.consume(host_wait=True)[2] in (seq_c, seq_c + 1, seq_c + 2)      # still a published frame, none of the cancelled ones
    ring.release(ring.consume(host_wait=True)[0])
    done = ring.produce(lambda b: b.fill_(7))
    slot_d, buf_d, seq_d = ring.consume(host_wait=True)
    assert slot_d == done and int(buf_d.max()) == 7 and int(buf_d.min()) == 7
    ring.release(slot_d)
    # a view over a pointer the library reports (what a GL-bound slot hands out): aliases the memory, no copy
    from desktop2stereo_amd.present import _DevMem
    src = torch.arange(64, dtype=torch.uint8, device=dev)
    view = torch.as_tensor(_DevMem(src.data_ptr(), 64), device=dev)
    view[:4] = 200
    torch.cuda.synchronize()
    assert src[:4].tolist() == [200] * 4 and view.data_ptr() == src.data_ptr()
    # GL registration needs a GL context: on a compute node it must fail loudly, never silently
    with pytest.raises(Exception):
        ring.bind_gl_buffer(0, 12345)
    ring.close(); eng.close()
