// tsan_driver.cpp — drives twhost::Manager (8 consumers on the stub backend of stub_twflow.cpp) under
// ThreadSanitizer: 1 000 in-memory jobs from two producer threads, a few bad requests, then a second manager that is
// disposed while its consumers are busy.  The reference's hazards this replaces: unsynchronised isRunning flags
// (/root/reference/src/message_queue.h:94-96, src/consumer.h:47, src/manager.h:63).  Exit code 0 and no TSAN
// report = pass (tests/test_host_tsan.py).
#include <stdio.h>

#include <chrono>
#include <condition_variable>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "twhost.h"

using namespace twhost;

struct Sink {
    std::mutex m;
    std::condition_variable cv;
    long data = 0, err = 0, hits = 0;
    bool completed = false;
    Report rep;
};

static Observer observer(Sink& s)
{
    Observer o;
    o.onNext = [&s](const Response& r) {
        std::lock_guard<std::mutex> lk(s.m);
        s.data++;
        s.hits += (long)r.vectors.size();
        s.cv.notify_all();
    };
    o.onError = [&s](const std::string&) {
        std::lock_guard<std::mutex> lk(s.m);
        s.err++;
        s.cv.notify_all();
    };
    o.onCompleted = [&s](const Report& rep) {
        std::lock_guard<std::mutex> lk(s.m);
        s.rep = rep;
        s.completed = true;
        s.cv.notify_all();
    };
    return o;
}

int main(int argc, char** argv)
{
    std::vector<uint8_t> a(64 * 48, 10), b(64 * 48, 10), c(64 * 48, 99);
    Parameter p;
    tw_default_params(&p.optParam);
    p.numThreads = 8;
    p.batch = 16;
    int rc = 0;
    {
        Sink s;
        Manager* mg = new Manager(observer(s));
        mg->start(p);
        if (mg->consumerCount() != 8) { fprintf(stderr, "expected 8 consumers, got %d\n", mg->consumerCount()); rc = 1; }
        mg->waitReady();
        mg->markEpoch();  // the readiness / epoch / stats paths of the queue driver, raced against the producers below
        std::thread watcher([&] {
            for (int i = 0; i < 200; i++) {
                long n = 0;
                for (const ConsumerStats& cs : mg->consumerStats()) n += cs.pairs;
                if (n > 1008) { fprintf(stderr, "stats: %ld pairs\n", n); rc = 1; }
                std::this_thread::sleep_for(std::chrono::microseconds(200));
            }
        });
        const int N = 1000, BAD = 7;
        auto producer = [&](int lo, int hi) {
            for (int j = lo; j < hi; j++) {
                RawPair r;
                r.expect = a.data();
                r.target = (j % 3 == 0) ? c.data() : b.data();  // every third pair "differs"
                r.width = 64;
                r.height = 48;
                r.stride = 64;
                mg->requestRaw("a", "b", r);
            }
        };
        std::thread t1(producer, 0, N / 2), t2(producer, N / 2, N);
        for (int j = 0; j < BAD; j++) mg->request("/nonexistent/expected.png", "/nonexistent/target.png");
        mg->request("", "x");
        t1.join();
        t2.join();
        watcher.join();
        {
            std::unique_lock<std::mutex> lk(s.m);
            s.cv.wait(lk, [&] { return s.data + s.err >= N + BAD + 1; });
        }
        mg->stop();
        {
            std::unique_lock<std::mutex> lk(s.m);
            s.cv.wait(lk, [&] { return s.completed; });
        }
        delete mg;
        const long want_hits = (N + 2) / 3;
        if (s.data != N || s.err != BAD + 1 || s.hits != want_hits || s.rep.requestCount != N + BAD + 1 ||
            s.rep.dataCount != N || s.rep.errorCount != BAD + 1) {
            fprintf(stderr, "counts: data %ld err %ld hits %ld (want %d %d %ld), report %d/%d/%d\n", s.data, s.err, s.hits,
                    N, BAD + 1, want_hits, s.rep.requestCount, s.rep.dataCount, s.rep.errorCount);
            rc = 1;
        }
    }
    {
        // dispose while the consumers are busy: queued jobs are dropped, completion still arrives
        Sink s;
        Manager* mg = new Manager(observer(s));
        mg->start(p);
        for (int j = 0; j < 4000; j++) {
            RawPair r;
            r.expect = a.data();
            r.target = b.data();
            r.width = 64;
            r.height = 48;
            r.stride = 64;
            mg->requestRaw("a", "b", r);
        }
        mg->stop();
        mg->stop();  // idempotent
        {
            std::unique_lock<std::mutex> lk(s.m);
            s.cv.wait(lk, [&] { return s.completed; });
        }
        delete mg;
        if (s.rep.requestCount != 4000 || s.data > 4000) { fprintf(stderr, "dispose: report %d, data %ld\n", s.rep.requestCount, s.data); rc = 1; }
    }
    if (argc > 1) {
        // FILES through the decode pool (round 4): argv[1] holds e.png / t.png (same size, first pixel differs), t5.png
        // (3 px narrower: the size-reconcile path), pal.png (palette: decoded on the host) and bad.png (damaged).  Batches
        // of pairs the device can finish are inflated straight into the page-locked arena (the fast path); a batch with
        // any other pair takes the decode-then-copy path; three batches are in flight per consumer.
        const std::string d = argv[1];
        Sink s;
        Parameter pf = p;
        pf.numThreads = 4;
        pf.batch = 8;
        Manager* mg = new Manager(observer(s));
        mg->start(pf);
        const int N = 400;
        int want_err = 0;
        long want_hits_min = 0;
        for (int j = 0; j < N; j++) {
            const int kind = (j / 16) % 2 == 0 ? j % 2 : j % 5;  // runs of 16 all-eligible pairs, then mixed ones
            const char* t = kind == 0 ? "e.png" : kind == 1 ? "t.png" : kind == 2 ? "t5.png" : kind == 3 ? "pal.png" : "bad.png";
            if (kind == 4) want_err++;
            if (kind == 1) want_hits_min++;
            mg->request(d + "/e.png", d + "/" + t);
        }
        {
            std::unique_lock<std::mutex> lk(s.m);
            s.cv.wait(lk, [&] { return s.data + s.err >= N; });
        }
        mg->stop();
        {
            std::unique_lock<std::mutex> lk(s.m);
            s.cv.wait(lk, [&] { return s.completed; });
        }
        delete mg;
        if (s.err != want_err || s.data != N - want_err || s.hits < want_hits_min) {
            fprintf(stderr, "files: data %ld err %ld hits %ld (want %d %d >= %ld)\n", s.data, s.err, s.hits, N - want_err,
                    want_err, want_hits_min);
            rc = 1;
        }
    }
    printf("tsan driver: %s\n", rc ? "FAILED" : "ok");
    return rc;
}
