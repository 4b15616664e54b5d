// numa_driver.cpp — prints where twhost::Manager placed its consumers (stub backend, CPU only): one JSON line with
// every consumer's device, PCI bus id, NUMA node and the CPUs its thread may run on after placement.
// tests/test_host_queue.py builds a pretend /sys tree (TW_SYSFS_ROOT), names pretend devices (TW_STUB_DEVICES,
// TW_STUB_PCI) and checks that consumer i sits on the CPUs of device (i % devices)'s node — SURVEY.md 8(e) "pin each
// worker + its staging buffers to the GPU's NUMA node"; the device mapping is /root/reference/src/consumer.cpp:18-24.
#include <stdio.h>
#include <stdlib.h>

#include <condition_variable>
#include <mutex>

#include "twhost.h"

using namespace twhost;

int main(int argc, char** argv)
{
    const int threads = argc > 1 ? atoi(argv[1]) : 4;
    const int jobs = argc > 2 ? atoi(argv[2]) : 64;
    std::mutex m;
    std::condition_variable cv;
    long done = 0;
    bool completed = false;
    Observer o;
    o.onNext = [&](const Response&) { std::lock_guard<std::mutex> lk(m); done++; cv.notify_all(); };
    o.onError = [&](const std::string&) { std::lock_guard<std::mutex> lk(m); done++; cv.notify_all(); };
    o.onCompleted = [&](const Report&) { std::lock_guard<std::mutex> lk(m); completed = true; cv.notify_all(); };
    Parameter p;
    tw_default_params(&p.optParam);
    p.numThreads = threads;
    p.batch = 4;
    Manager* mg = new Manager(o);
    mg->start(p);
    mg->waitReady();
    std::vector<uint8_t> a(64 * 48, 10), b(64 * 48, 20);
    for (int j = 0; j < jobs; j++) {
        RawPair r;
        r.expect = a.data();
        r.target = b.data();
        r.width = 64;
        r.height = 48;
        r.stride = 64;
        mg->requestRaw("a", "b", r);
    }
    {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return done >= jobs; });
    }
    mg->stop();
    {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return completed; });
    }
    const std::vector<ConsumerStats> st = mg->consumerStats();  // consumers have joined: final numbers
    long pairs = 0;
    printf("{\"consumers\": [");
    for (size_t i = 0; i < st.size(); i++) {
        const ConsumerStats& s = st[i];
        pairs += s.pairs;
        printf("%s{\"id\": %d, \"device\": %d, \"pci\": \"%s\", \"numa_node\": %d, \"pairs\": %ld, \"cpus\": [", i ? ", " : "",
               s.id, s.device, s.pciBusId.c_str(), s.numaNode, s.pairs);
        for (size_t c = 0; c < s.cpus.size(); c++) printf("%s%d", c ? ", " : "", s.cpus[c]);
        printf("]}");
    }
    printf("], \"pairs\": %ld, \"main_thread_cpus\": %zu}\n", pairs, this_thread_cpus().size());
    delete mg;
    return pairs == jobs ? 0 : 1;
}
