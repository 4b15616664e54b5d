// stub_twflow.cpp — a stand-in for libtwflow.so's C ABI that computes nothing: every pair answers a canned vector
// list.  ONLY for CPU-side builds of the host layer: the ThreadSanitizer build (`make tsan`; GPU sanitizer runs are not
// available) and the host-layer ceiling measurement (`make queue_stub`, tests/test_host_queue.py): twhost.cpp's queue,
// consumers, pump and dispose path with 8 consumers on 8 pretend devices.  Never linked into the product.
//
// Two timing models:
//   default              every tw_wait sleeps 50 us (TSAN runs: just enough to interleave threads)
//   TW_STUB_BATCH_MS=T   each pretend device is a FIFO server with the MEASURED service time of the real engine:
//                        a batch of `slots` pairs takes T ms (a partial batch its share), batches of one engine run one
//                        after the other, and tw_wait returns when the batch's completion time has passed (a timer,
//                        not a sleep per call) — VERDICT r4 #5: 31.3 ms per 128 pairs per device at 1080p.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <atomic>
#include <chrono>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/twflow.h"

typedef std::chrono::steady_clock stub_clock;
struct tw_engine {
    int device, cap;
    std::mutex m;
    std::map<tw_ticket, int> open;  // ticket -> width (echoed as the vector's x)
    tw_ticket next = 1;
    // timer model (TW_STUB_BATCH_MS): the open batch and the completion time of every launched one
    double batch_ms = 0;
    long cur_batch = 0;
    int cur_count = 0;
    std::map<tw_ticket, long> batch_of;
    std::map<long, stub_clock::time_point> done_at;
    std::map<long, int> waiters;  // pairs of a launched batch not collected yet
    stub_clock::time_point free_at = stub_clock::now();
    void launch_locked()
    {
        if (!cur_count) return;
        const stub_clock::time_point now = stub_clock::now();
        const stub_clock::time_point start = now > free_at ? now : free_at;
        const auto service = std::chrono::nanoseconds((long long)(batch_ms * 1e6 * cur_count / (cap > 0 ? cap : 1)));
        free_at = start + service;
        done_at[cur_batch] = free_at;
        waiters[cur_batch] = cur_count;
        cur_batch++;
        cur_count = 0;
    }
};

static std::atomic<int> g_engines{0};

extern "C" {
void tw_default_params(tw_params* p)
{
    p->pyrScale = 0.5; p->pyrLevels = 3; p->winSize = 30; p->pyrIterations = 3; p->polyN = 7; p->polySigma = 1.5; p->flags = 256;
}
int tw_device_count(void)
{
    const char* ev = getenv("TW_STUB_DEVICES");
    return ev ? atoi(ev) : 8;
}
// TW_STUB_PCI="0000:01:00.0,0000:81:00.0,...": bus id of pretend device i (cycled); unset: "0000:<i>0:00.0"
tw_status tw_device_pci_bus_id(int device, char* buf, int cap)
{
    if (!buf || cap < 16 || device < 0 || device >= tw_device_count()) return TW_E_DEVICE;
    const char* ev = getenv("TW_STUB_PCI");
    if (ev && ev[0]) {
        std::string s(ev);
        std::vector<std::string> ids;
        size_t p = 0;
        while (p <= s.size()) {
            const size_t q = s.find(',', p);
            ids.push_back(s.substr(p, q == std::string::npos ? std::string::npos : q - p));
            if (q == std::string::npos) break;
            p = q + 1;
        }
        snprintf(buf, (size_t)cap, "%s", ids[(size_t)device % ids.size()].c_str());
    } else {
        snprintf(buf, (size_t)cap, "0000:%x0:00.0", device & 0xf);
    }
    return TW_OK;
}
tw_status tw_host_alloc(tw_engine*, size_t bytes, void** hptr)
{
    *hptr = malloc(bytes);
    return *hptr ? TW_OK : TW_E_NOMEM;
}
tw_status tw_host_free(tw_engine*, void* hptr)
{
    free(hptr);
    return TW_OK;
}
tw_status tw_set_option(tw_engine*, int, int) { return TW_OK; }
tw_status tw_prof_select(tw_engine*, int, int) { return TW_OK; }
tw_status tw_prof_read(tw_engine*, int, double* ms, int* n)
{
    if (ms) *ms = 0;
    if (n) *n = 0;
    return TW_OK;
}
const char* tw_strerror(tw_status) { return "stub"; }
const char* tw_last_error(const tw_engine*) { return ""; }
int tw_grid_capacity(int w, int h, int span) { return span > 0 ? ((h + span - 1) / span) * ((w + span - 1) / span) : 0; }
tw_status tw_engine_create(int device, const tw_params*, int slots, tw_engine** out)
{
    tw_engine* e = new tw_engine();
    e->device = device;
    e->cap = slots;
    if (const char* ev = getenv("TW_STUB_BATCH_MS")) e->batch_ms = atof(ev);
    g_engines++;
    *out = e;
    return TW_OK;
}
void tw_engine_destroy(tw_engine* e)
{
    g_engines--;
    delete e;
}
tw_status tw_submit_u8(tw_engine* e, const uint8_t* a, const uint8_t* b, int w, int h, ptrdiff_t, int, double, tw_ticket* t)
{
    if (!a || !b || w < 1 || h < 1) return TW_E_BAD_PARAMETER;
    std::lock_guard<std::mutex> lk(e->m);
    *t = e->next++;
    e->open[*t] = w + (a[0] != b[0] ? 1000000 : 0);
    if (e->batch_ms > 0) {
        e->batch_of[*t] = e->cur_batch;
        if (++e->cur_count >= e->cap) e->launch_locked();  // a full batch starts by itself, like the engine's
    }
    return TW_OK;
}
tw_status tw_submit_png8(tw_engine* e, const uint8_t* a, int cha, const uint8_t* b, int chb, int w, int h, int span, double thr,
                         tw_ticket* t)
{
    // filtered rows: byte 0 is a filter type, byte 1 the first sample — compare those like tw_submit_u8 compares pixel 0
    return tw_submit_u8(e, a + (cha ? 1 : 0), b + (chb ? 1 : 0), w, h, w, span, thr, t);
}
tw_status tw_flush(tw_engine* e)
{
    if (e->batch_ms > 0) {
        std::lock_guard<std::mutex> lk(e->m);
        e->launch_locked();
    }
    return TW_OK;
}
tw_status tw_wait(tw_engine* e, tw_ticket t, tw_vector* out, int cap, int* n, float* seconds)
{
    int w;
    {
        std::lock_guard<std::mutex> lk(e->m);
        auto it = e->open.find(t);
        if (it == e->open.end()) return TW_E_BAD_PARAMETER;
        w = it->second;
        e->open.erase(it);
    }
    if (e->batch_ms > 0) {
        stub_clock::time_point until;
        {
            std::lock_guard<std::mutex> lk(e->m);
            const long b = e->batch_of[t];
            e->batch_of.erase(t);
            if (b == e->cur_batch) e->launch_locked();  // waiting for a pair of the open batch starts it
            until = e->done_at[b];
            if (--e->waiters[b] == 0) {
                e->waiters.erase(b);
                e->done_at.erase(b);
            }
        }
        std::this_thread::sleep_until(until);  // returns at once for every later pair of a finished batch
    } else {
        std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
    const bool differs = w >= 1000000;
    if (n) *n = differs ? 1 : 0;
    if (differs && out && cap > 0) {
        out[0].x = w - 1000000;
        out[0].y = e->device;
        out[0].dx = 6.0;
        out[0].dy = -7.5;
    }
    if (seconds) *seconds = 1e-4f;
    return TW_OK;
}
}
