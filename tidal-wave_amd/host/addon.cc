// addon.cc — N-API binding of the `TidalWave` class (reference: src/broker.{h,cpp}, src/main.cpp — written there
// against the node-0.10 V8 API, which node >= 12 no longer has).  Same class name, methods, argument checks,
// event names and payload key order; events are delivered on the JS thread through a thread-safe function.
#include <node_api.h>
#include <string.h>

#include <string>

#include "twhost.h"

using namespace twhost;

namespace {

struct Event {
    enum Kind { DATA, ERROR_, FINISH } kind;
    Response res;
    std::string reason;
    Report report;
};

struct Broker {
    napi_env env = nullptr;
    napi_ref self = nullptr;  // strong reference: the JS object lives until 'finish' has been emitted
    napi_threadsafe_function tsfn = nullptr;
    Manager* manager = nullptr;
};

#define NAPI_OK(call)                     \
    do {                                  \
        if ((call) != napi_ok) return nullptr; \
    } while (0)

void set_str(napi_env env, napi_value o, const char* k, const std::string& v)
{
    napi_value s;
    napi_create_string_utf8(env, v.c_str(), v.size(), &s);
    napi_set_named_property(env, o, k, s);
}
void set_i32(napi_env env, napi_value o, const char* k, int v)
{
    napi_value n;
    napi_create_int32(env, v, &n);
    napi_set_named_property(env, o, k, n);
}
void set_f64(napi_env env, napi_value o, const char* k, double v)
{
    napi_value n;
    napi_create_double(env, v, &n);
    napi_set_named_property(env, o, k, n);
}

// Broker::convertResult, src/broker.cpp:161-188 (key order preserved)
napi_value convert_result(napi_env env, const Response& r)
{
    napi_value o;
    napi_create_object(env, &o);
    set_str(env, o, "status", r.status);
    set_i32(env, o, "span", r.span);
    set_f64(env, o, "threshold", r.threshold);
    set_str(env, o, "expect_image", r.expect_image);
    set_str(env, o, "target_image", r.target_image);
    set_f64(env, o, "time", r.time);
    set_i32(env, o, "height", r.height);
    set_i32(env, o, "width", r.width);
    napi_value arr;
    napi_create_array_with_length(env, r.vectors.size(), &arr);
    for (size_t i = 0; i < r.vectors.size(); i++) {
        napi_value v;
        napi_create_object(env, &v);
        set_i32(env, v, "x", r.vectors[i].x);
        set_i32(env, v, "y", r.vectors[i].y);
        set_f64(env, v, "dx", r.vectors[i].dx);
        set_f64(env, v, "dy", r.vectors[i].dy);
        napi_set_element(env, arr, (uint32_t)i, v);
    }
    napi_set_named_property(env, o, "vector", arr);
    return o;
}

// runs on the JS thread: this.emit(name, payload)  (src/broker.cpp:44-86)
void deliver(napi_env env, napi_value /*js_cb*/, void* context, void* data)
{
    Broker* b = (Broker*)context;
    Event* ev = (Event*)data;
    if (env && b->self) {
        napi_handle_scope scope;
        napi_open_handle_scope(env, &scope);
        napi_value self, emit, name, payload;
        napi_get_reference_value(env, b->self, &self);
        napi_create_object(env, &payload);
        const char* nm = "data";
        if (ev->kind == Event::DATA) {
            payload = convert_result(env, ev->res);
        } else if (ev->kind == Event::ERROR_) {
            nm = "error";
            set_str(env, payload, "status", "ERROR");
            set_str(env, payload, "reason", ev->reason);
        } else {
            nm = "finish";
            set_i32(env, payload, "request", ev->report.requestCount);
            set_i32(env, payload, "data", ev->report.dataCount);
            set_i32(env, payload, "error", ev->report.errorCount);
        }
        napi_create_string_utf8(env, nm, NAPI_AUTO_LENGTH, &name);
        napi_valuetype t = napi_undefined;
        if (napi_get_named_property(env, self, "emit", &emit) == napi_ok && napi_typeof(env, emit, &t) == napi_ok &&
            t == napi_function) {
            napi_value argv[2] = {name, payload}, ret;
            napi_call_function(env, self, emit, 2, argv, &ret);
            bool pending = false;
            napi_is_exception_pending(env, &pending);
            if (pending) {
                napi_value ex;
                napi_get_and_clear_last_exception(env, &ex);
                napi_fatal_exception(env, ex);
            }
        }
        if (ev->kind == Event::FINISH) {
            napi_delete_reference(env, b->self);
            b->self = nullptr;
            napi_release_threadsafe_function(b->tsfn, napi_tsfn_release);
        }
        napi_close_handle_scope(env, scope);
    }
    delete ev;
}

// Broker::getInt32OrDefault / getNumberOrDefault, src/broker.cpp:190-209: wrong JS type => default
int get_i32(napi_env env, napi_value opts, const char* k, int def)
{
    napi_valuetype t;
    if (!opts || napi_typeof(env, opts, &t) != napi_ok || t != napi_object) return def;
    bool has = false;
    napi_value v;
    if (napi_has_named_property(env, opts, k, &has) != napi_ok || !has) return def;
    if (napi_get_named_property(env, opts, k, &v) != napi_ok || napi_typeof(env, v, &t) != napi_ok || t != napi_number)
        return def;
    double d;
    napi_get_value_double(env, v, &d);
    if (!(d >= -2147483648.0 && d <= 2147483647.0) || d != (double)(int)d || (d == 0 && 1 / d < 0)) return def;  // IsInt32
    return (int)d;
}
double get_f64(napi_env env, napi_value opts, const char* k, double def)
{
    napi_valuetype t;
    if (!opts || napi_typeof(env, opts, &t) != napi_ok || t != napi_object) return def;
    bool has = false;
    napi_value v;
    if (napi_has_named_property(env, opts, k, &has) != napi_ok || !has) return def;
    if (napi_get_named_property(env, opts, k, &v) != napi_ok || napi_typeof(env, v, &t) != napi_ok || t != napi_number)
        return def;
    double d;
    napi_get_value_double(env, v, &d);
    return d;
}

void finalize(napi_env, void* data, void*)
{
    Broker* b = (Broker*)data;
    if (b->manager) {
        b->manager->stop();
        delete b->manager;  // joins the pump and the consumers
    }
    delete b;
}

// Broker::createInstance, src/broker.cpp:101-123
napi_value construct(napi_env env, napi_callback_info info)
{
    size_t argc = 1;
    napi_value argv[1] = {nullptr}, self;
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, &self, nullptr));
    napi_value opts = argc >= 1 ? argv[0] : nullptr;
    Parameter p;
    tw_default_params(&p.optParam);
    p.threshold = get_f64(env, opts, "threshold", 5.0);
    p.span = get_i32(env, opts, "span", 10);
    // The reference's default is 4 consumer threads (src/broker.cpp:109).  Here a consumer is a GPU worker, so a caller
    // that passes no numThreads gets one per device the box has (on an 8-GPU node the default used to
    // drive four cards); 4 only when no device answers (every job then reports the missing device).
    const int ndev_default = tw_device_count();
    p.numThreads = get_i32(env, opts, "numThreads", ndev_default > 0 ? ndev_default : 4);
    p.optParam.pyrScale = get_f64(env, opts, "pyrScale", 0.5);
    p.optParam.pyrLevels = get_i32(env, opts, "pyrLevels", 3);
    p.optParam.winSize = get_i32(env, opts, "winSize", 30);
    p.optParam.pyrIterations = get_i32(env, opts, "pyrIterations", 3);
    p.optParam.polyN = get_i32(env, opts, "polyN", 7);
    p.optParam.polySigma = get_f64(env, opts, "polySigma", 1.5);
    p.optParam.flags = get_i32(env, opts, "flags", 256);  // cv::OPTFLOW_FARNEBACK_GAUSSIAN

    Broker* b = new Broker();
    b->env = env;
    NAPI_OK(napi_wrap(env, self, b, finalize, nullptr, nullptr));
    NAPI_OK(napi_create_reference(env, self, 1, &b->self));
    napi_value rname;
    napi_create_string_utf8(env, "tidalwave", NAPI_AUTO_LENGTH, &rname);
    NAPI_OK(napi_create_threadsafe_function(env, nullptr, nullptr, rname, 0, 1, nullptr, nullptr, b, deliver, &b->tsfn));
    Observer obs;
    obs.onNext = [b](const Response& r) {
        Event* ev = new Event();
        ev->kind = Event::DATA;
        ev->res = r;
        napi_call_threadsafe_function(b->tsfn, ev, napi_tsfn_blocking);
    };
    obs.onError = [b](const std::string& why) {
        Event* ev = new Event();
        ev->kind = Event::ERROR_;
        ev->reason = why;
        napi_call_threadsafe_function(b->tsfn, ev, napi_tsfn_blocking);
    };
    obs.onCompleted = [b](const Report& rep) {
        Event* ev = new Event();
        ev->kind = Event::FINISH;
        ev->report = rep;
        napi_call_threadsafe_function(b->tsfn, ev, napi_tsfn_blocking);
    };
    b->manager = new Manager(obs);
    b->manager->start(p);
    return self;
}

bool get_string(napi_env env, napi_value v, std::string& out)
{
    napi_valuetype t;
    if (napi_typeof(env, v, &t) != napi_ok || t != napi_string) return false;
    size_t n = 0;
    napi_get_value_string_utf8(env, v, nullptr, 0, &n);
    out.resize(n);
    napi_get_value_string_utf8(env, v, &out[0], n + 1, &n);
    return true;
}

// Broker::requestCalc, src/broker.cpp:125-150
napi_value calc(napi_env env, napi_callback_info info)
{
    size_t argc = 3;
    napi_value argv[3], self;
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, &self, nullptr));
    if (argc != 2) {
        napi_throw_type_error(env, nullptr, "2 arguments expected");
        return nullptr;
    }
    std::string a, b;
    if (!get_string(env, argv[0], a)) {
        napi_throw_type_error(env, nullptr, "Wrong arguments(expect_image)");
        return nullptr;
    }
    if (!get_string(env, argv[1], b)) {
        napi_throw_type_error(env, nullptr, "Wrong arguments(target_image)");
        return nullptr;
    }
    Broker* br = nullptr;
    NAPI_OK(napi_unwrap(env, self, (void**)&br));
    if (br && br->manager) br->manager->request(a, b);
    return nullptr;
}

// Broker::requestDispose, src/broker.cpp:152-158
napi_value dispose(napi_env env, napi_callback_info info)
{
    napi_value self;
    NAPI_OK(napi_get_cb_info(env, info, nullptr, nullptr, &self, nullptr));
    Broker* br = nullptr;
    NAPI_OK(napi_unwrap(env, self, (void**)&br));
    if (br && br->manager) br->manager->stop();
    return nullptr;
}

// decodeGray(path) -> {width, height, data: Buffer} | null : the host-side cv::imread(…, GRAYSCALE) stand-in,
// exported for the decode parity tests (not part of the reference's surface)
napi_value decode_gray(napi_env env, napi_callback_info info)
{
    size_t argc = 1;
    napi_value argv[1], self;
    NAPI_OK(napi_get_cb_info(env, info, &argc, argv, &self, nullptr));
    std::string path;
    napi_value nul;
    napi_get_null(env, &nul);
    if (argc < 1 || !get_string(env, argv[0], path)) return nul;
    std::vector<uint8_t> img;
    int w = 0, h = 0;
    if (!load_gray(path, img, w, h)) return nul;
    napi_value o, buf;
    void* dst = nullptr;
    napi_create_object(env, &o);
    napi_create_buffer_copy(env, img.size(), img.data(), &dst, &buf);
    set_i32(env, o, "width", w);
    set_i32(env, o, "height", h);
    napi_set_named_property(env, o, "data", buf);
    return o;
}

napi_value init(napi_env env, napi_value exports)
{
    napi_property_descriptor props[] = {
        {"calc", nullptr, calc, nullptr, nullptr, nullptr, napi_default, nullptr},
        {"dispose", nullptr, dispose, nullptr, nullptr, nullptr, napi_default, nullptr},
    };
    napi_value cls;
    NAPI_OK(napi_define_class(env, "TidalWave", NAPI_AUTO_LENGTH, construct, nullptr, 2, props, &cls));
    NAPI_OK(napi_set_named_property(env, exports, "TidalWave", cls));
    napi_value fn;
    NAPI_OK(napi_create_function(env, "decodeGray", NAPI_AUTO_LENGTH, decode_gray, nullptr, &fn));
    NAPI_OK(napi_set_named_property(env, exports, "decodeGray", fn));
    napi_value n;
    napi_create_int32(env, tw_device_count(), &n);
    napi_set_named_property(env, exports, "deviceCount", n);
    return exports;
}

}  // namespace

NAPI_MODULE(tidalwave, init)
