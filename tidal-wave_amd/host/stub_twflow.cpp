// stub_twflow.cpp — a stand-in for libtwflow.so's C ABI that computes nothing: every pair answers a canned vector
// list after a short sleep.  ONLY for the ThreadSanitizer build of the host layer (`make tsan`), which must run
// on the CPU (GPU sanitizer runs are not available): it lets twhost.cpp's queue, consumers, pump and dispose path
// run under TSAN with 8 consumers on 8 pretend devices.  Never linked into the product.
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

struct tw_engine {
    int device, cap;
    std::mutex m;
    std::map<tw_ticket, int> open;  // ticket -> width (echoed as the vector's x)
    tw_ticket next = 1;
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
    return TW_OK;
}
tw_status tw_submit_png8(tw_engine* e, const uint8_t* a, int cha, const uint8_t* b, int chb, int w, int h, int span, double thr,
                         tw_ticket* t)
{
    // filtered rows: byte 0 is a filter type, byte 1 the first sample — compare those like tw_submit_u8 compares pixel 0
    return tw_submit_u8(e, a + (cha ? 1 : 0), b + (chb ? 1 : 0), w, h, w, span, thr, t);
}
tw_status tw_flush(tw_engine*) { return TW_OK; }
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
    std::this_thread::sleep_for(std::chrono::microseconds(50));
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
