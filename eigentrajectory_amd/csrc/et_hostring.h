// et_hostring.h -- a lagged, non-blocking look at the device-side et_kmeans_state from the host.
//
// The Lloyd loops (et_kmeans_fit, et_kmeans_fit_sharded) never wait for the convergence flag inside the loop:
// every few iterations the state block is copied to a pinned ring slot, and only a copy that has ARRIVED (event
// query) is looked at; launches that were queued after convergence are no-ops on the device.
//
// This is the one place where the library keeps memory of its own (include/eigentraj.h, "Ownership"): per host
// thread and device 4 x sizeof(et_kmeans_state) + 64 bytes of pinned host memory and 4 events, created on first use
// and released when that host thread ends.  No device memory is ever allocated by the library.
#pragma once

#include <vector>

#include "et_common.h"

namespace et {

class StateRing {
  public:
    static constexpr int kSlots = 4;

    // the ring of the calling host thread for the current device (nullptr + *rc set on a HIP error)
    static StateRing *get(int *rc) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) {
            *rc = ET_ERR_HIP;
            return nullptr;
        }
        // per host thread: released when the thread ends (the Python side runs fits from short-lived pool threads)
        struct Holder {
            std::vector<StateRing *> v;
            ~Holder() {
                for (StateRing *r : v) delete r;
            }
        };
        static thread_local Holder holder;
        std::vector<StateRing *> &rings = holder.v;
        if ((int)rings.size() <= dev) rings.resize(dev + 1, nullptr);
        if (!rings[dev]) {
            StateRing *r = new StateRing();
            // (+ one cache line for the mailbox the chained Lloyd kernel writes its progress into, see mailbox())
            if (hipHostMalloc((void **)&r->slots_, sizeof(et_kmeans_state) * kSlots + 64, hipHostMallocDefault) != hipSuccess) {
                delete r;
                *rc = ET_ERR_HIP;
                return nullptr;
            }
            r->mail_ = reinterpret_cast<volatile unsigned long long *>(r->slots_ + kSlots);
            if (hipHostGetDevicePointer((void **)&r->mail_dev_, (void *)r->mail_, 0) != hipSuccess) r->mail_dev_ = nullptr;
            for (int i = 0; i < kSlots; ++i) {
                if (hipEventCreateWithFlags(&r->ev_[i], hipEventDisableTiming) != hipSuccess) {
                    delete r;
                    *rc = ET_ERR_HIP;
                    return nullptr;
                }
                ++r->n_ev_;
            }
            rings[dev] = r;
        }
        rings[dev]->posted_ = rings[dev]->seen_ = 0;
        *rc = ET_OK;
        return rings[dev];
    }

    // enqueue a copy of the device state; when the ring is full, first wait for the oldest copy.  *done |= that flag.
    int post(const et_kmeans_state *dev_state, hipStream_t st, bool *done) {
        if (posted_ - seen_ == kSlots) {
            ET_HIP_TRY(hipEventSynchronize(ev_[seen_ % kSlots]));
            *done = *done || slots_[seen_ % kSlots].done != 0;
            ++seen_;
        }
        ET_HIP_TRY(hipMemcpyAsync(&slots_[posted_ % kSlots], dev_state, sizeof(et_kmeans_state), hipMemcpyDeviceToHost, st));
        ET_HIP_TRY(hipEventRecord(ev_[posted_ % kSlots], st));
        ++posted_;
        return ET_OK;
    }

    int pending() const { return posted_ - seen_; }

    // block until the oldest outstanding copy has arrived and look at it.  A sharded loop uses ONLY this (never
    // poll): which copy a rank looks at must not depend on timing, or the ranks would stop enqueueing their
    // collectives at different iterations.
    int wait_oldest(bool *done) {
        if (seen_ == posted_) return ET_OK;
        ET_HIP_TRY(hipEventSynchronize(ev_[seen_ % kSlots]));
        *done = *done || slots_[seen_ % kSlots].done != 0;
        ++seen_;
        return ET_OK;
    }

    // look at every copy that has arrived (never blocks)
    void poll(bool *done) {
        while (seen_ < posted_ && hipEventQuery(ev_[seen_ % kSlots]) == hipSuccess) {
            *done = *done || slots_[seen_ % kSlots].done != 0;
            ++seen_;
        }
    }

    // Mailbox: 8 bytes of the pinned block that workgroup 0 of every chained Lloyd launch overwrites with
    // (done << 63) | iterations applied -- a plain store over the host link, no copy packet and no event in the stream
    // (a state copy every 4 launches cost a 4.6 us copy kernel and ~8 us of dispatch gaps each: 0.2 ms per 100
    // iterations).  The single-GPU loop reads it to stop launching after convergence and to stay a bounded number of
    // launches ahead of the device.  NOT for the sharded loop: what a rank sees here depends on timing.
    unsigned long long *mailbox_device() const { return mail_dev_; }
    void mailbox_reset() {
        if (mail_) *mail_ = 0ull;
    }
    bool mailbox_done() const { return mail_ && (*mail_ >> 63) != 0; }
    long long mailbox_iter() const { return mail_ ? (long long)(*mail_ & 0x7fffffffffffffffull) : 0; }

    // the pinned block and the events go back to the runtime with the owning thread (errors are ignored: at process
    // exit the runtime may already be gone)
    ~StateRing() {
        for (int i = 0; i < n_ev_; ++i) (void)hipEventDestroy(ev_[i]);
        if (slots_) (void)hipHostFree(slots_);
    }

  private:
    int n_ev_ = 0;
    volatile unsigned long long *mail_ = nullptr;
    unsigned long long *mail_dev_ = nullptr;
    et_kmeans_state *slots_ = nullptr;
    hipEvent_t ev_[kSlots];
    int posted_ = 0, seen_ = 0;
};

}  // namespace et
