// bench_queue.cpp — queue-sharded throughput driver (BASELINE.json configs[3]: "batch 2048x 1080p pairs sharded
// across 8xMI355X via broker/manager queue").
//
// One process, one twhost::Manager, min(numThreads, deviceCount * perDevice) consumers on ONE shared request
// queue — the reference's only parallelism strategy (/root/reference/src/manager.cpp:55-59 spawns the consumers,
// :68-78 enqueues, src/consumer.cpp:42-94 is the worker loop).  The pairs are in memory (page-locked buffers by
// default, twhost::RawPair), so neither the file system nor a decoder is in the way: what is measured is the
// queue, the per-GPU consumers, the uploads over PCIe and the engines.  Needs no edits for 8 GPUs: --devices 0
// uses every device the box has.
//
//   bench_queue --pgm-dir DIR [--pairs 2048] [--devices 0] [--per-device 1] [--batch 128] [--warmup-batches 2]
//               [--pinned 1] [--files 0] [--span 10] [--threshold 5] [--prof 1]
// DIR holds pair_<i>_a.pgm / pair_<i>_b.pgm (i = 0..; or .png), written by bench.py / tests from tidal-wave_amd/synth.py.
// Prints one JSON line.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <vector>

#include "../tidal-wave_amd/host/twhost.h"

using namespace twhost;

namespace {
struct Args {
    std::string dir;
    int pairs = 2048, devices = 0, per_device = 1, batch = 128, warmup_batches = 2, pinned = 1, files = 0, span = 10;
    int prof = 1;  // event-bracket the level-0 blur / polyexp launches of every consumer's engine
    double threshold = 5.0;
    int syn_w = 0, syn_h = 0, syn_n = 4;  // --synthetic WxH [--synthetic-pairs N]: random pairs made in memory, no files
};

bool parse(int argc, char** argv, Args& a)
{
    for (int i = 1; i < argc; i++) {
        auto val = [&](const char* name) -> const char* {
            if (strcmp(argv[i], name) != 0 || i + 1 >= argc) return nullptr;
            return argv[++i];
        };
        const char* v;
        if ((v = val("--pgm-dir"))) a.dir = v;
        else if ((v = val("--pairs"))) a.pairs = atoi(v);
        else if ((v = val("--devices"))) a.devices = atoi(v);
        else if ((v = val("--per-device"))) a.per_device = atoi(v);
        else if ((v = val("--batch"))) a.batch = atoi(v);
        else if ((v = val("--warmup-batches"))) a.warmup_batches = atoi(v);
        else if ((v = val("--pinned"))) a.pinned = atoi(v);
        else if ((v = val("--files"))) a.files = atoi(v);
        else if ((v = val("--span"))) a.span = atoi(v);
        else if ((v = val("--prof"))) a.prof = atoi(v);
        else if ((v = val("--threshold"))) a.threshold = atof(v);
        else if ((v = val("--synthetic"))) {
            if (sscanf(v, "%dx%d", &a.syn_w, &a.syn_h) != 2 || a.syn_w < 1 || a.syn_h < 1) return false;
        } else if ((v = val("--synthetic-pairs"))) a.syn_n = atoi(v);
        else {
            fprintf(stderr, "unknown argument %s\n", argv[i]);
            return false;
        }
    }
    return (!a.dir.empty() || a.syn_w > 0) && a.pairs > 0 && a.batch > 0 && a.syn_n > 0;
}

struct Img {
    std::vector<uint8_t> pageable;
    uint8_t* data = nullptr;
    int w = 0, h = 0;
};
}  // namespace

int main(int argc, char** argv)
{
    Args a;
    if (!parse(argc, argv, a)) {
        fprintf(stderr, "usage: bench_queue --pgm-dir DIR [--pairs N] [--devices N] [--per-device K] [--batch B] ...\n");
        return 2;
    }
    if (a.devices > 0 && !getenv("HIP_VISIBLE_DEVICES")) {
        // restrict the run to the first `devices` GPUs before the first HIP call (consumer i uses device i % count)
        std::string vis;
        for (int i = 0; i < a.devices; i++) vis += (i ? "," : "") + std::to_string(i);
        setenv("HIP_VISIBLE_DEVICES", vis.c_str(), 1);
    }
    const int ndev_all = tw_device_count();
    if (ndev_all < 1) {
        fprintf(stderr, "bench_queue needs a HIP device: the product path has no CPU fallback\n");
        return 3;
    }
    const int ndev = a.devices > 0 ? std::min(a.devices, ndev_all) : ndev_all;

    // the distinct pairs of the directory, decoded once
    tw_params prm;
    tw_default_params(&prm);
    tw_engine* alloc_eng = nullptr;  // only used for page-locked allocations
    if (a.pinned && !a.files && tw_engine_create(0, &prm, 1, &alloc_eng) != TW_OK) {
        fprintf(stderr, "cannot create an engine on device 0\n");
        return 3;
    }
    std::vector<Img> imgs;
    std::vector<std::string> names;
    if (a.syn_w > 0) {
        // in-memory pairs without a directory (the host-layer ceiling runs on the stub backend): `syn_n` distinct pairs of
        // LCG noise, every second pair identical (the stub answers one vector when pixel 0 differs)
        unsigned sd = 12345u;
        for (int i = 0; i < 2 * a.syn_n; i++) {
            Img im;
            im.w = a.syn_w;
            im.h = a.syn_h;
            const size_t n = (size_t)im.w * im.h;
            void* hp = nullptr;
            if (alloc_eng) {
                if (tw_host_alloc(alloc_eng, n, &hp) != TW_OK) {
                    fprintf(stderr, "tw_host_alloc failed\n");
                    return 3;
                }
                im.data = (uint8_t*)hp;
            } else {
                im.pageable.resize(n);
                im.data = im.pageable.data();
            }
            if ((i & 1) && ((i >> 1) & 1)) {
                memcpy(im.data, imgs[(size_t)i - 1].data, n);
            } else {
                for (size_t k = 0; k < n; k++) {
                    sd = sd * 1664525u + 1013904223u;
                    im.data[k] = (uint8_t)(sd >> 24);
                }
            }
            imgs.push_back(std::move(im));
            char nm[64];
            snprintf(nm, sizeof(nm), "synthetic_%d_%c", i >> 1, (i & 1) ? 'b' : 'a');
            names.push_back(nm);
        }
    }
    for (int i = 0; a.syn_w == 0; i++) {
        bool ok = true;
        for (int q = 0; q < 2 && ok; q++) {
            char path[4096];
            snprintf(path, sizeof(path), "%s/pair_%d_%c.pgm", a.dir.c_str(), i, q ? 'b' : 'a');
            Img im;
            if (!load_gray(path, im.pageable, im.w, im.h)) {
                // (--files 1 also takes PNG pairs: the decode pool then runs its half-decode + device path)
                snprintf(path, sizeof(path), "%s/pair_%d_%c.png", a.dir.c_str(), i, q ? 'b' : 'a');
                if (!load_gray(path, im.pageable, im.w, im.h)) {
                    ok = false;
                    break;
                }
            }
            im.data = im.pageable.data();
            if (alloc_eng) {
                void* hp = nullptr;
                if (tw_host_alloc(alloc_eng, im.pageable.size(), &hp) != TW_OK) {
                    fprintf(stderr, "tw_host_alloc failed\n");
                    return 3;
                }
                memcpy(hp, im.pageable.data(), im.pageable.size());
                im.pageable.clear();
                im.pageable.shrink_to_fit();
                im.data = (uint8_t*)hp;
            }
            imgs.push_back(std::move(im));
            names.push_back(path);
        }
        if (!ok) break;
    }
    if (imgs.size() % 2) {
        imgs.pop_back();
        names.pop_back();
    }
    const int distinct = (int)imgs.size() / 2;
    if (distinct < 1) {
        fprintf(stderr, "no pair_<i>_a.pgm / pair_<i>_b.pgm under %s\n", a.dir.c_str());
        return 2;
    }

    std::mutex m;
    std::condition_variable cv;
    long done = 0, errors = 0, flagged = 0;
    bool completed = false;
    Report final_report;
    std::string first_error;
    Observer obs;
    obs.onNext = [&](const Response& r) {
        std::lock_guard<std::mutex> lk(m);
        done++;
        flagged += (long)r.vectors.size();
        cv.notify_all();
    };
    obs.onError = [&](const std::string& reason) {
        std::lock_guard<std::mutex> lk(m);
        done++;
        errors++;
        if (first_error.empty()) first_error = reason;
        cv.notify_all();
    };
    obs.onCompleted = [&](const Report& rep) {
        std::lock_guard<std::mutex> lk(m);
        final_report = rep;
        completed = true;
        cv.notify_all();
    };

    Parameter p;
    p.span = a.span;
    p.threshold = a.threshold;
    p.optParam = prm;
    p.consumersPerDevice = std::max(1, a.per_device);
    p.numThreads = ndev * p.consumersPerDevice;  // one consumer per (device, slot): src/manager.cpp:55-59
    p.batch = a.batch;
    p.profileKernels = a.prof != 0;
    Manager* mg = new Manager(obs);
    mg->start(p);
    const int consumers = mg->consumerCount();
    // Readiness barrier: every consumer has bound its device, placed itself on the GPU's NUMA node and
    // created its engine before the first job is pushed — HIP initialisation is serialised per process, and a
    // consumer still inside tw_engine_create would otherwise pay for it inside the timed region.
    mg->waitReady();

    auto push = [&](long j) {
        const int k = (int)(j % distinct);
        const Img &ia = imgs[2 * k], &ib = imgs[2 * k + 1];
        if (a.files) {
            mg->request(names[2 * k], names[2 * k + 1]);
        } else {
            RawPair rp;
            rp.expect = ia.data;
            rp.target = ib.data;
            rp.width = ia.w;
            rp.height = ia.h;
            rp.stride = ia.w;
            mg->requestRaw(names[2 * k], names[2 * k + 1], rp);
        }
    };
    auto wait_done = [&](long target) {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return done >= target; });
    };

    // warm-up: rounds of batch x consumers jobs until EVERY consumer has run at least warmup_batches full batches'
    // worth of pairs (plan, workspaces, first-launch costs); a consumer that got none in a round gets its chance in
    // the next one (a consumer takes at most its share of a short queue)
    long warm = 0;
    int rounds = 0;
    const long want_each = (long)std::max(0, a.warmup_batches) * a.batch;  // 0: no warm-up at all (tests)
    for (; rounds < 32; rounds++) {
        long least = want_each;
        for (const ConsumerStats& cs : mg->consumerStats()) least = std::min(least, cs.pairs);
        if (least >= want_each) break;
        const long n = (long)a.batch * consumers;
        for (long j = 0; j < n; j++) push(warm + j);
        warm += n;
        wait_done(warm);
    }
    bool all_warm = true;
    for (const ConsumerStats& cs : mg->consumerStats()) all_warm = all_warm && cs.pairs >= want_each;
    const long flagged_warm = flagged;
    mg->markEpoch();  // every consumer restarts its counters (and drops the warm-up's kernel events) at its next job

    const auto t0 = std::chrono::steady_clock::now();
    for (long j = 0; j < a.pairs; j++) push(warm + j);
    wait_done(warm + a.pairs);
    const auto t1 = std::chrono::steady_clock::now();
    const double sec = std::chrono::duration<double>(t1 - t0).count();
    const long long t0_ns = std::chrono::duration_cast<std::chrono::nanoseconds>(t0.time_since_epoch()).count();

    mg->stop();
    {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return completed; });
    }
    const std::vector<ConsumerStats> stats = mg->consumerStats();  // the consumers have joined: final numbers
    const PumpStats pump = mg->pumpStats();
    delete mg;
    if (alloc_eng) {
        for (Img& im : imgs) tw_host_free(alloc_eng, im.data);
        tw_engine_destroy(alloc_eng);
    }
    printf("{\"pairs\": %d, \"seconds\": %.6f, \"pairs_per_s\": %.2f, \"devices\": %d, \"consumers\": %d, "
           "\"engine_batch\": %d, \"distinct_pairs\": %d, \"width\": %d, \"height\": %d, \"input\": \"%s\", "
           "\"errors\": %ld, \"flagged_vectors\": %ld, \"report\": {\"request\": %d, \"data\": %d, \"error\": %d}, "
           "\"first_error\": \"%s\", \"warmup_pairs\": %ld, \"warmup_rounds\": %d, \"all_consumers_warm\": %s, "
           "\"pump\": {\"delivered\": %ld, \"latency_mean_us\": %.2f, \"latency_max_us\": %.2f, \"callback_busy_ms\": %.3f}, "
           "\"per_consumer\": [",
           a.pairs, sec, a.pairs / sec, ndev, consumers, a.batch, distinct, imgs[0].w, imgs[0].h,
           a.files ? "pgm files" : (a.pinned ? "page-locked host buffers" : "pageable host buffers"), errors,
           flagged - flagged_warm, final_report.requestCount, final_report.dataCount, final_report.errorCount,
           first_error.c_str(), warm, rounds, all_warm ? "true" : "false", pump.delivered, pump.meanUs, pump.maxUs,
           pump.busyMs);
    for (size_t i = 0; i < stats.size(); i++) {
        const ConsumerStats& cs = stats[i];
        // idle = the part of the timed region this consumer did not work in: before its first job, after its last
        // response, and blocked on an empty queue in between (at N = 8 a starved consumer must show)
        const double start_ms = cs.firstJobNs ? (double)(cs.firstJobNs - t0_ns) * 1e-6 : sec * 1e3;
        const double active_ms = cs.firstJobNs && cs.lastResponseNs > cs.firstJobNs
                                     ? (double)(cs.lastResponseNs - cs.firstJobNs) * 1e-6 - cs.waitMs : 0.0;
        const double idle_ms = std::max(0.0, sec * 1e3 - active_ms);
        printf("%s{\"id\": %d, \"device\": %d, \"pci\": \"%s\", \"numa_node\": %d, \"cpus\": %zu, \"pairs\": %ld, "
               "\"batches\": %ld, \"blur_l0_ms\": %.4f, \"blur_l0_launches\": %d, \"polyexp_l0_ms\": %.4f, "
               "\"polyexp_l0_launches\": %d, \"first_job_ms\": %.3f, \"wait_ms\": %.3f, \"idle_ms\": %.3f, "
               "\"idle_frac\": %.4f}",
               i ? ", " : "", cs.id, cs.device, cs.pciBusId.c_str(), cs.numaNode, cs.cpus.size(), cs.pairs, cs.batches,
               cs.profMs[0], cs.profLaunches[0], cs.profMs[1], cs.profLaunches[1], start_ms, cs.waitMs, idle_ms,
               idle_ms / (sec * 1e3));
    }
    printf("]}\n");
    return errors ? 1 : 0;
}
