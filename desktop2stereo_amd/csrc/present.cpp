// f4: device-resident hand-off of a produced frame (packed stereo frame, depth) to the display path.
//
// The reference's viewer gets frames into OpenGL like this (viewer.py:1584-1712, 2399-2428, CUDART_GL :232-345):
//     torch.cuda.current_stream().synchronize()           <- host stalls until the frame is complete
//     hipGraphicsMapResources(pbo) -> hipMemcpy(D2D) into the mapped pointer -> hipGraphicsUnmapResources
//     glTexSubImage2D(... from the PBO)
// i.e. one host synchronisation and one extra device copy of the whole frame per frame.  Here the consumer lends its
// buffers to the producer instead: a ring of N consumer-owned device allocations (plain device pointers -- the mapped
// pointer of a registered GL PBO is one; d2s_present_bind_gl_buffer does the registration the reference does through
// ctypes), d2s_pipeline / d2s_make_sbs write their output straight into the acquired slot, and the hand-off is two HIP events
// per slot ("ready" recorded on the producer's stream, "released" on the consumer's): no host synchronisation, no copy,
// frame i+1 is produced while frame i is displayed.  Latest-frame semantics like the viewer's update_frame.
#include "common.h"

#include <dlfcn.h>
#include <mutex>
#include <vector>

struct d2s_present {
    struct Slot {
        void* ptr = nullptr;
        uint64_t bytes = 0;
        void* gl_resource = nullptr;              // hipGraphicsResource* when bound to a GL buffer
        bool mapped = false;
        hipEvent_t ready = nullptr, released = nullptr;
        uint64_t seq = 0;                          // publish sequence number (0: never published)
        bool has_released = false;
        bool held = false;                         // between consume and release: the consumer's stream may still read it
        bool writing = false;                      // between acquire and publish: the producer's stream is (about to be) writing it
    };
    int device = 0;
    std::vector<Slot> slots;
    int next = 0;                                  // next slot the producer acquires
    uint64_t seq = 0;
    std::mutex mu;                                 // producer and consumer are different host threads (main.py:232-262)
};

using namespace d2s;

namespace {
// the GL interop entry points of libamdhip64, resolved at run time like the reference's ctypes binding (no GL headers here)
typedef hipError_t (*reg_fn)(void**, unsigned, unsigned);
typedef hipError_t (*map_fn)(int, void**, hipStream_t);
typedef hipError_t (*ptr_fn)(void**, size_t*, void*);
typedef hipError_t (*unreg_fn)(void*);
template <typename F> F sym(const char* name) { return (F)dlsym(RTLD_DEFAULT, name); }
}  // namespace

extern "C" int d2s_present_create(int device_id, int slots, d2s_present** out) {
    D2S_REQUIRE(out && slots >= 1 && slots <= 16, "slots must be in [1,16]");
    D2S_ON_DEVICE(device_id);
    d2s_present* p = new d2s_present();
    p->device = device_id;
    p->slots.resize(slots);
    for (auto& s : p->slots) {
        hipError_t err = hipEventCreateWithFlags(&s.ready, hipEventDisableTiming);
        if (err == hipSuccess) err = hipEventCreateWithFlags(&s.released, hipEventDisableTiming);
        if (err != hipSuccess) {                   // give back what exists so far
            for (auto& t : p->slots) {
                if (t.ready) (void)hipEventDestroy(t.ready);
                if (t.released) (void)hipEventDestroy(t.released);
            }
            delete p;
            return hip_fail(err, "hipEventCreateWithFlags", __FILE__, __LINE__);
        }
    }
    *out = p;
    return D2S_OK;
}

extern "C" int d2s_present_bind(d2s_present* p, int slot, void* dev_ptr, uint64_t bytes) {
    D2S_REQUIRE(p && slot >= 0 && slot < (int)p->slots.size() && dev_ptr && bytes > 0, "bad argument");
    std::lock_guard<std::mutex> lk(p->mu);
    auto& s = p->slots[slot];
    D2S_REQUIRE(!s.gl_resource, "slot is bound to a GL buffer");
    D2S_REQUIRE(!s.held && !s.writing, "slot is in use (acquired or consumed): release / publish it first");
    s.ptr = dev_ptr; s.bytes = bytes; s.seq = 0; s.has_released = false;
    return D2S_OK;
}

extern "C" int d2s_present_bind_gl_buffer(d2s_present* p, int slot, unsigned gl_buffer) {
    D2S_REQUIRE(p && slot >= 0 && slot < (int)p->slots.size(), "bad argument");
    D2S_ON_DEVICE(p->device);
    static reg_fn reg = sym<reg_fn>("hipGraphicsGLRegisterBuffer");
    if (!reg) { set_error("hipGraphicsGLRegisterBuffer is not available in this HIP runtime"); return D2S_E_UNSUPPORTED; }
    std::lock_guard<std::mutex> lk(p->mu);
    auto& s = p->slots[slot];
    D2S_REQUIRE(!s.held && !s.writing, "slot is in use (acquired or consumed): release / publish it first");
    void* res = nullptr;
    const unsigned WRITE_DISCARD = 2;              // hipGraphicsRegisterFlagsWriteDiscard: the producer overwrites the whole buffer (viewer.py:293)
    hipError_t err = reg(&res, gl_buffer, WRITE_DISCARD);
    if (err != hipSuccess || !res) return hip_fail(err != hipSuccess ? err : hipErrorInvalidValue, "hipGraphicsGLRegisterBuffer (is a GL context current on this thread?)", __FILE__, __LINE__);
    s.gl_resource = res; s.ptr = nullptr; s.bytes = 0; s.mapped = false; s.seq = 0; s.has_released = false;
    return D2S_OK;
}

// producer: the next slot of the ring.  The producer's stream waits (on the device) until the consumer released that slot.
extern "C" int d2s_present_acquire(d2s_present* p, void* producer_stream, int* slot, void** dev_ptr, uint64_t* bytes) {
    D2S_REQUIRE(p && slot && dev_ptr, "null pointer");
    D2S_ON_DEVICE(p->device);
    std::lock_guard<std::mutex> lk(p->mu);
    // Triple buffering: never the slot the consumer holds (its release event does not exist yet, the device cannot wait for it),
    // and -- while another one is free -- not the latest published slot either, so the consumer always finds a frame.  With
    // two slots and a consumer holding one, the other is re-used even if it is the latest: d2s_present_consume then reports
    // "nothing published" until the next publish (use >= 3 slots when producer and consumer run concurrently).
    const int n = (int)p->slots.size();
    int latest = -1;
    for (int k = 0; k < n; ++k)
        if (p->slots[k].seq > 0 && (latest < 0 || p->slots[k].seq > p->slots[latest].seq)) latest = k;
    int i = -1, fallback = -1;
    for (int k = 0; k < n; ++k) {
        const int c = (p->next + k) % n;
        if (p->slots[c].held || p->slots[c].writing) continue;     // the consumer reads it / an earlier acquire is still unpublished
        if (c != latest) { i = c; break; }
        if (fallback < 0) fallback = c;
    }
    if (i < 0) i = fallback;
    if (i < 0) { set_error("d2s_present_acquire: every slot is held by the consumer or acquired and not yet published"); return D2S_E_STATE; }
    auto& s = p->slots[i];
    s.seq = 0;                                      // being rewritten: not consumable until published again
    if (s.gl_resource && !s.mapped) {              // map for the time the producer writes (unmapped again at publish)
        static map_fn mapr = sym<map_fn>("hipGraphicsMapResources");
        static ptr_fn getp = sym<ptr_fn>("hipGraphicsResourceGetMappedPointer");
        if (!mapr || !getp) { set_error("HIP-GL interop entry points missing"); return D2S_E_UNSUPPORTED; }
        D2S_HIP(mapr(1, &s.gl_resource, (hipStream_t)producer_stream));
        size_t n = 0;
        D2S_HIP(getp(&s.ptr, &n, s.gl_resource));
        s.bytes = n; s.mapped = true;
    }
    if (!s.ptr) { set_error("d2s_present_acquire: slot has no buffer bound"); return D2S_E_STATE; }
    if (s.has_released) D2S_HIP(hipStreamWaitEvent((hipStream_t)producer_stream, s.released, 0));
    s.writing = true;
    p->next = (i + 1) % (int)p->slots.size();
    *slot = i; *dev_ptr = s.ptr;
    if (bytes) *bytes = s.bytes;
    return D2S_OK;
}

// producer: everything queued on producer_stream so far (the kernels that wrote the slot) precedes "ready"
extern "C" int d2s_present_publish(d2s_present* p, int slot, void* producer_stream) {
    D2S_REQUIRE(p && slot >= 0 && slot < (int)p->slots.size(), "bad argument");
    D2S_ON_DEVICE(p->device);
    std::lock_guard<std::mutex> lk(p->mu);
    auto& s = p->slots[slot];
    if (!s.writing) { set_error("d2s_present_publish: the slot was not acquired (or was published already)"); return D2S_E_STATE; }
    s.writing = false;                              // a failure below leaves the slot unpublished (seq 0) but acquirable again
    if (s.gl_resource && s.mapped) {
        static map_fn unmap = sym<map_fn>("hipGraphicsUnmapResources");
        if (!unmap) { set_error("hipGraphicsUnmapResources missing"); return D2S_E_UNSUPPORTED; }
        s.mapped = false;
        D2S_HIP(unmap(1, &s.gl_resource, (hipStream_t)producer_stream));     // stream-ordered: GL sees the buffer after the writes
    }
    D2S_HIP(hipEventRecord(s.ready, (hipStream_t)producer_stream));
    s.seq = ++p->seq;
    return D2S_OK;
}

// producer: give an acquired slot back WITHOUT publishing it (the frame was not produced: the pipeline raised, the mapped buffer
// turned out too small, ...).  The slot becomes acquirable again with its old contents marked unusable (seq stays 0, so the consumer
// never sees it); a GL-bound slot is unmapped, stream-ordered, like at publish.  Without this an abandoned slot would stay
// `writing` forever -- excluded from acquire, not re-bindable -- and two such failures with a consumer holding the third slot would
// stall a 3-slot ring for good.
extern "C" int d2s_present_cancel(d2s_present* p, int slot, void* producer_stream) {
    D2S_REQUIRE(p && slot >= 0 && slot < (int)p->slots.size(), "bad argument");
    D2S_ON_DEVICE(p->device);
    std::lock_guard<std::mutex> lk(p->mu);
    auto& s = p->slots[slot];
    if (!s.writing) { set_error("d2s_present_cancel: the slot is not acquired"); return D2S_E_STATE; }
    s.writing = false;                              // first: whatever fails below, the slot is not lost
    if (s.gl_resource && s.mapped) {
        static map_fn unmap = sym<map_fn>("hipGraphicsUnmapResources");
        s.mapped = false;
        if (!unmap) { set_error("hipGraphicsUnmapResources missing"); return D2S_E_UNSUPPORTED; }
        D2S_HIP(unmap(1, &s.gl_resource, (hipStream_t)producer_stream));
    }
    return D2S_OK;
}

// consumer: the most recently published slot (latest-frame semantics).  consumer_stream waits on the device for "ready";
// consumer_stream == (void*)-1: wait on the HOST instead (what a GL consumer needs before it sources the PBO).
// *dev_ptr: the slot's device pointer; NULL for a GL-bound slot (publish unmapped it -- the consumer sources the GL buffer
// object itself, e.g. glTexSubImage2D from the bound PBO).
extern "C" int d2s_present_consume(d2s_present* p, void* consumer_stream, int* slot, void** dev_ptr, uint64_t* seq) {
    D2S_REQUIRE(p && slot && dev_ptr, "null pointer");
    D2S_ON_DEVICE(p->device);
    hipEvent_t ready = nullptr;
    {
        // the lock covers the slot choice only: a host wait under it would stall the producer's acquire / publish for a
        // whole GPU frame time (they are different host threads)
        std::lock_guard<std::mutex> lk(p->mu);
        int best = -1;
        for (int i = 0; i < (int)p->slots.size(); ++i)
            if (p->slots[i].seq > 0 && (best < 0 || p->slots[i].seq > p->slots[best].seq)) best = i;
        if (best < 0) { set_error("d2s_present_consume: nothing published yet"); return D2S_E_STATE; }
        auto& s = p->slots[best];
        s.held = true;                              // until d2s_present_release: the producer skips it
        ready = s.ready;                            // (not re-recorded while the slot is held)
        *slot = best; *dev_ptr = s.gl_resource ? nullptr : s.ptr;
        if (seq) *seq = s.seq;
    }
    hipError_t err = consumer_stream == (void*)-1 ? hipEventSynchronize(ready) : hipStreamWaitEvent((hipStream_t)consumer_stream, ready, 0);
    if (err != hipSuccess) {
        std::lock_guard<std::mutex> lk(p->mu);
        p->slots[*slot].held = false;
        return hip_fail(err, "wait for the slot's ready event", __FILE__, __LINE__);
    }
    return D2S_OK;
}

// consumer: its reads of the slot (queued on consumer_stream so far) are done -> the producer may overwrite it
extern "C" int d2s_present_release(d2s_present* p, int slot, void* consumer_stream) {
    D2S_REQUIRE(p && slot >= 0 && slot < (int)p->slots.size(), "bad argument");
    D2S_ON_DEVICE(p->device);
    std::lock_guard<std::mutex> lk(p->mu);
    auto& s = p->slots[slot];
    if (!s.held) { set_error("d2s_present_release: the slot is not held by the consumer"); return D2S_E_STATE; }
    if (consumer_stream != (void*)-1) D2S_HIP(hipEventRecord(s.released, (hipStream_t)consumer_stream));
    s.has_released = consumer_stream != (void*)-1;
    s.held = false;
    return D2S_OK;
}

extern "C" int d2s_present_destroy(d2s_present* p) {
    if (!p) return D2S_OK;
    D2S_ON_DEVICE(p->device);
    static unreg_fn unreg = sym<unreg_fn>("hipGraphicsUnregisterResource");
    for (auto& s : p->slots) {
        if (s.gl_resource && unreg) (void)unreg(s.gl_resource);
        if (s.ready) (void)hipEventDestroy(s.ready);
        if (s.released) (void)hipEventDestroy(s.released);
    }
    delete p;
    return D2S_OK;
}
