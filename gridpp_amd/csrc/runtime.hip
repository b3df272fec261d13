// Runtime, point sets and neighbour queries of libgridpp_hip.so.
#include "oi_common.h"
#include <algorithm>
#include <thread>
#include <mutex>
#include <atomic>
#include <memory>

namespace gpp {

std::recursive_mutex& api_mutex_ref() { static std::recursive_mutex m; return m; }   // one compute call at a time (GPP_TRY, common.h)
static thread_local std::string g_last_error;
void set_error(const char* msg) { g_last_error = msg ? msg : ""; }
int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_last_error = buf;
    return code;
}

static hipStream_t g_stream = nullptr, g_stream2 = nullptr, g_stream3 = nullptr, g_stream4 = nullptr;
static std::atomic<unsigned long long> g_next_serial{1};   // one counter for every instantiation of make_points<T>
static std::atomic<int> g_live_handles{0};   // gpp_points alive (gpp_set_device refuses to switch under them)
static int g_device = -1;
static std::mutex g_mutex;

void ensure_device() {
    std::lock_guard<std::mutex> lock(g_mutex);
    if(g_device >= 0) return;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if(e != hipSuccess || count == 0) throw Error{GPP_ENODEVICE, "no HIP device visible (libgridpp_hip has no CPU path)"};
    int dev = 0;
    (void)hipGetDevice(&dev);
    g_device = dev;
    GPP_HIP(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
}
hipStream_t stream() {
    ensure_device();
    return g_stream;
}
hipStream_t stream2() {
    ensure_device();
    std::lock_guard<std::mutex> lock(g_mutex);
    if(!g_stream2) GPP_HIP(hipStreamCreateWithFlags(&g_stream2, hipStreamNonBlocking));
    return g_stream2;
}
hipStream_t stream3() {
    ensure_device();
    std::lock_guard<std::mutex> lock(g_mutex);
    if(!g_stream3) GPP_HIP(hipStreamCreateWithFlags(&g_stream3, hipStreamNonBlocking));
    return g_stream3;
}
hipStream_t stream4() {
    ensure_device();
    std::lock_guard<std::mutex> lock(g_mutex);
    if(!g_stream4) GPP_HIP(hipStreamCreateWithFlags(&g_stream4, hipStreamNonBlocking));
    return g_stream4;
}

}   // namespace gpp

using namespace gpp;

extern "C" const char* gpp_last_error(void) { return g_last_error.c_str(); }
extern "C" const char* gpp_version(void) { return "0.8.0.dev1+mi355x.r1"; }

extern char** environ;
void gpp_release_ensi_workspace();   // ensi.hip
void gpp_release_oi_workspace();     // oi.hip
void gpp_oi_drain_pending();         // oi.hip: completes the GPP_ASYNC calls of the calling thread

// ---- path overrides: the one test hook (common.h: path_env) ---------------------------------------------------------------------
namespace gpp {
static std::mutex g_override_mutex;
// (name, value): the value strings are interned -- every value ever set stays alive in `override_values` (a few bytes per
//  gpp_set_path_override call of a test process) -- so the pointer path_override() hands out stays valid whatever another thread
//  sets or clears afterwards; callers may hold it across their whole call (it used to point into the table entry)
static std::vector<std::pair<std::string, const char*>>& override_table() { static std::vector<std::pair<std::string, const char*>> t; return t; }
static const char* intern_override_value(const char* v) {
    static std::vector<std::unique_ptr<std::string>> override_values;
    override_values.emplace_back(new std::string(v));
    return override_values.back()->c_str();
}
const char* path_override(const char* name) {
    std::lock_guard<std::mutex> lock(g_override_mutex);
    for(auto& kv : override_table()) if(kv.first == name) return kv.second;
    return nullptr;
}
}
// name = one of the GPP_* switches of the library (they choose between implementations with identical results), value = its setting,
// NULL to clear it.  Process-wide.
extern "C" int gpp_set_path_override(const char* name, const char* value) {
    GPP_TRY
    if(!name || strncmp(name, "GPP_", 4) != 0) invalid("override names start with GPP_");
    std::lock_guard<std::mutex> lock(g_override_mutex);
    auto& t = override_table();
    for(size_t i = 0; i < t.size(); i++)
        if(t[i].first == name) {
            if(value) { if(strcmp(t[i].second, value) != 0) t[i].second = intern_override_value(value); }
            else t.erase(t.begin() + (long)i);
            return GPP_OK;
        }
    if(value) t.emplace_back(name, intern_override_value(value));
    return GPP_OK;
    GPP_CATCH
}
// the overrides that are set, comma separated; returns how many
extern "C" int gpp_active_overrides(char* buf, int len) {
    GPP_TRY
    std::string out;
    int n = 0;
    {
        std::lock_guard<std::mutex> lock(g_override_mutex);
        for(auto& kv : override_table()) { if(!out.empty()) out += ","; out += kv.first; n++; }
    }
    if(buf && len > 0) { strncpy(buf, out.c_str(), (size_t)len - 1); buf[len - 1] = 0; }
    return n;
    GPP_CATCH
}

#ifdef GPP_POISON
extern "C" int gpp_debug_poison_oi_workspace(int byte);                      // oi.hip
extern "C" int gpp_debug_poison_ensi_workspace(int byte);                    // ensi.hip (optimal_interpolation_ensi and _ensi_multi)
extern "C" int gpp_debug_poison_nbh_workspace(int byte, int keep_padding);   // neighbourhood.hip
// every call-to-call workspace of the calling thread (diagnostic build only, tools/hostile/build.sh)
extern "C" int gpp_debug_poison_workspaces(int byte, int keep_padding) {
    int rc = gpp_debug_poison_oi_workspace(byte);
    if(rc == GPP_OK) rc = gpp_debug_poison_ensi_workspace(byte);
    if(rc == GPP_OK) rc = gpp_debug_poison_nbh_workspace(byte, keep_padding);
    return rc;
}
#endif

// ---- the pool of staging buffers (common.h: Staged) -- touched under the API lock only ----------------------------------------------------
namespace gpp {
namespace {
struct StageSlot { void* p; size_t cap; };
std::vector<StageSlot> g_stage_free;
size_t g_stage_bytes = 0;
constexpr size_t STAGE_KEEP_ONE = (size_t)256 << 20, STAGE_KEEP_ALL = (size_t)1 << 30;
}
void* stage_borrow(size_t bytes, size_t* cap) {
    int best = -1;
    for(int i = 0; i < (int)g_stage_free.size(); i++)   // the smallest kept buffer that holds the request (and is not more than four times too large)
        if(g_stage_free[i].cap >= bytes && g_stage_free[i].cap <= 4 * bytes + 4096 && (best < 0 || g_stage_free[i].cap < g_stage_free[best].cap)) best = i;
    void* p = nullptr;
    if(best >= 0) {
        p = g_stage_free[best].p; *cap = g_stage_free[best].cap;
        g_stage_bytes -= *cap;
        g_stage_free.erase(g_stage_free.begin() + best);
    }
    else {
        const size_t want = (bytes + 4095) & ~(size_t)4095;
        if(hipMalloc(&p, want) != hipSuccess) {   // out of memory with buffers lying idle in the pool: give them back and try again
            (void)hipGetLastError();
            stage_release_all();
            GPP_HIP(hipMalloc(&p, want));
        }
        *cap = want;
    }
#ifdef GPP_POISON
    GPP_HIP(hipMemsetAsync(p, 0xFF, *cap, stream()));   // (diagnostic build: whatever the last borrower left is as hostile as a fresh allocation's 0xFF)
#endif
    return p;
}
void stage_return(void* p, size_t cap) {
    if(!p) return;
    if(cap > STAGE_KEEP_ONE) { (void)hipFree(p); return; }
    g_stage_free.push_back({p, cap});
    g_stage_bytes += cap;
    while(g_stage_bytes > STAGE_KEEP_ALL || g_stage_free.size() > 24) {   // the largest goes first
        int big = 0;
        for(int i = 1; i < (int)g_stage_free.size(); i++) if(g_stage_free[i].cap > g_stage_free[big].cap) big = i;
        (void)hipFree(g_stage_free[big].p);
        g_stage_bytes -= g_stage_free[big].cap;
        g_stage_free.erase(g_stage_free.begin() + big);
    }
}
void stage_release_all() {
    for(auto& sl : g_stage_free) (void)hipFree(sl.p);
    g_stage_free.clear();
    g_stage_bytes = 0;
}
}   // namespace gpp

// frees the thread's large call-to-call workspaces (they grow on demand and are otherwise kept for the next call)
extern "C" int gpp_release_workspaces(void) {
    GPP_TRY
    gpp_oi_drain_pending();
    // (deferred calls of ANOTHER thread are not in this thread's queue: everything in flight on the three streams ends before a buffer goes)
    GPP_HIP(hipStreamSynchronize(stream()));
    if(g_stream2) GPP_HIP(hipStreamSynchronize(g_stream2));
    if(g_stream3) GPP_HIP(hipStreamSynchronize(g_stream3));
    if(g_stream4) GPP_HIP(hipStreamSynchronize(g_stream4));
    stage_release_all();
    gpp_release_ensi_workspace();
    gpp_release_oi_workspace();
    return GPP_OK;
    GPP_CATCH
}

extern "C" int gpp_device_count(int* count) {
    GPP_TRY
    if(!count) invalid("count is NULL");
    int c = 0;
    if(hipGetDeviceCount(&c) != hipSuccess) c = 0;
    *count = c;
    return GPP_OK;
    GPP_CATCH
}
extern "C" int gpp_set_device(int device) {
    GPP_TRY
    std::lock_guard<std::mutex> lock(g_mutex);
    int count = 0;
    if(hipGetDeviceCount(&count) != hipSuccess || count == 0) throw Error{GPP_ENODEVICE, "no HIP device visible"};
    if(device < 0 || device >= count) invalid("device index out of range");
    GPP_HIP(hipSetDevice(device));
    if(g_device != device) {
        // every handle, workspace and the stream live on the device that was current at the first call: moving to another one
        // afterwards would mix pointers of two devices
        if(g_stream && g_live_handles.load() > 0) invalid("gpp_set_device: the device cannot change while point sets / fields created on the current one are alive");
        if(g_stream) (void)hipStreamSynchronize(g_stream);
        if(g_stream2) (void)hipStreamSynchronize(g_stream2);
        if(g_stream3) (void)hipStreamSynchronize(g_stream3);
        if(g_stream4) (void)hipStreamSynchronize(g_stream4);
        stage_release_all();   // (staging buffers of the device that is left)
        if(g_stream) { (void)hipStreamDestroy(g_stream); g_stream = nullptr; }
        if(g_stream2) { (void)hipStreamDestroy(g_stream2); g_stream2 = nullptr; }
        if(g_stream3) { (void)hipStreamDestroy(g_stream3); g_stream3 = nullptr; }
        if(g_stream4) { (void)hipStreamDestroy(g_stream4); g_stream4 = nullptr; }
        g_device = device;
        GPP_HIP(hipStreamCreateWithFlags(&g_stream, hipStreamNonBlocking));
    }
    return GPP_OK;
    GPP_CATCH
}
extern "C" int gpp_get_stream(void** s) {
    GPP_TRY
    if(!s) invalid("stream is NULL");
    *s = (void*)stream();
    return GPP_OK;
    GPP_CATCH
}
// Page-locked host memory for result arrays: a device-to-host copy into it is one DMA transfer at link speed, while a copy
// into freshly allocated pageable memory is bounded by its first-touch page faults (measured: 64 MB in 7 ms, i.e. 9 GB/s).
extern "C" int gpp_host_alloc(size_t bytes, void** out) {
    GPP_TRY
    if(!out) invalid("NULL argument");
    ensure_device();
    *out = nullptr;
    GPP_HIP(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    return GPP_OK;
    GPP_CATCH
}
extern "C" int gpp_host_free(void* p) {
    GPP_TRY
    if(p) GPP_HIP(hipHostFree(p));
    return GPP_OK;
    GPP_CATCH
}

extern "C" int gpp_synchronize(void) {
    GPP_TRY
    GPP_HIP(hipStreamSynchronize(stream()));
    // (the streams beside the library stream carry work of deferred optimal-interpolation calls only; created on first use)
    if(g_stream2) GPP_HIP(hipStreamSynchronize(g_stream2));
    if(g_stream3) GPP_HIP(hipStreamSynchronize(g_stream3));
    if(g_stream4) GPP_HIP(hipStreamSynchronize(g_stream4));
    return GPP_OK;
    GPP_CATCH
}

// ---- coordinates (src/api/util.cpp:583-624) -----------------------------------
// Host libm on purpose: x/y/z are float32 roundings of double cos/sin products; using the
// same libm as the reference build keeps every coordinate (and therefore every float32
// distance, radius membership and nearest-neighbour index) bit-identical.
static const double kRadiusEarth = 6.378137e6;   // include/gridpp.h:55

static bool convert_range(const float* lats, const float* lons, int i0, int i1, int type, float* x, float* y, float* z) {
    for(int i = i0; i < i1; i++) {
        float lat = lats[i], lon = lons[i];
        bool ok = (type == GPP_CARTESIAN) ? is_valid(lat) : (is_valid(lat) && lat >= -90.001 && lat <= 90.001);
        if(!ok || !is_valid(lon)) return false;
        if(type == GPP_CARTESIAN) { x[i] = lon; y[i] = lat; z[i] = 0; }
        else {
            double lonr = M_PI / 180 * lon, latr = M_PI / 180 * lat;
            x[i] = (float)(std::cos(latr) * std::cos(lonr) * kRadiusEarth);
            y[i] = (float)(std::cos(latr) * std::sin(lonr) * kRadiusEarth);
            z[i] = (float)(std::sin(latr) * kRadiusEarth);
        }
    }
    return true;
}
static void convert_all(const float* lats, const float* lons, int n, int type, float* x, float* y, float* z) {
    if(type != GPP_GEODETIC && type != GPP_CARTESIAN) invalid("Unknown coordinate type");
    int nt = (int)std::min<long>(std::max(1u, std::thread::hardware_concurrency()), std::max(1, n / 65536));
    bool ok = true;
    if(nt <= 1) ok = convert_range(lats, lons, 0, n, type, x, y, z);
    else {
        std::vector<std::thread> th;
        std::vector<char> oks(nt, 1);
        for(int t = 0; t < nt; t++) {
            int i0 = (int)((long)n * t / nt), i1 = (int)((long)n * (t + 1) / nt);
            th.emplace_back([=, &oks]() { oks[t] = convert_range(lats, lons, i0, i1, type, x, y, z); });
        }
        for(auto& t : th) t.join();
        for(char c : oks) ok = ok && c;
    }
    if(!ok) invalid("Invalid coords");   // util.cpp:596-600
}

extern "C" int gpp_convert_coordinates(const float* lats, const float* lons, int n, int type, float* x, float* y, float* z) {
    GPP_TRY
    if(n < 0) invalid("n < 0");
    convert_all(lats, lons, n, type, x, y, z);
    return GPP_OK;
    GPP_CATCH
}

// ---- the same conversion on the device, for large point sets --------------------------------------------------
// x/y/z must equal the host libm values bit for bit.  Both libm's and the device library's sin / cos are accurate to a
// couple of ulp(double); the float32 rounding of two doubles that close differs only if a float32 rounding boundary
// lies between them.  So the kernel computes in double, rounds, and lists every point one of whose coordinates lies
// within 64 ulp(double) of such a boundary (about 3 in 10^8); the host recomputes those few with libm and patches them.
#pragma clang fp contract(off)
__device__ __forceinline__ bool d_near_float_boundary(const double v) {
    const float f = (float)v;
    if(!(fabs(v) < 3.0e38)) return true;
    const double fd = (double)f;
    const float fn = (v >= fd) ? nextafterf(f, INFINITY) : nextafterf(f, -INFINITY);   // the neighbour on v's side
    const double mid = 0.5 * (fd + (double)fn);
    return fabs(v - mid) <= fabs(v) * 1.5e-14 + 1e-300;
}
__global__ void k_convert(const float* __restrict__ lats, const float* __restrict__ lons, int n, int type,
                          float* __restrict__ x, float* __restrict__ y, float* __restrict__ z,
                          int* __restrict__ flagged, int cap, int* __restrict__ counters) {   // counters[0] = #flagged, [1] = invalid input
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n) return;
    const float lat = lats[i], lon = lons[i];
    const bool vlat = !isnan(lat) && !isinf(lat), vlon = !isnan(lon) && !isinf(lon);
    const bool ok = (type == GPP_CARTESIAN) ? vlat : (vlat && (double)lat >= -90.001 && (double)lat <= 90.001);
    if(!ok || !vlon) { counters[1] = 1; return; }
    if(type == GPP_CARTESIAN) { x[i] = lon; y[i] = lat; z[i] = 0; return; }
    const double lonr = M_PI / 180 * (double)lon, latr = M_PI / 180 * (double)lat;
    const double cl = cos(latr), sl = sin(latr), co = cos(lonr), so = sin(lonr);
    const double xd = cl * co * 6.378137e6, yd = cl * so * 6.378137e6, zd = sl * 6.378137e6;
    x[i] = (float)xd; y[i] = (float)yd; z[i] = (float)zd;
    if(d_near_float_boundary(xd) || d_near_float_boundary(yd) || d_near_float_boundary(zd)) {
        const int k = atomicAdd(&counters[0], 1);
        if(k < cap) flagged[k] = i;
    }
}
__global__ void k_patch(const int* __restrict__ idx, const float* __restrict__ v, int m, float* __restrict__ x, float* __restrict__ y, float* __restrict__ z) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if(k >= m) return;
    const int i = idx[k];
    x[i] = v[3 * k]; y[i] = v[3 * k + 1]; z[i] = v[3 * k + 2];
}
// returns false when the device result cannot be trusted (too many boundary cases): the caller converts on the host
__global__ void k_cast_f64(const double* __restrict__ in, int n, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n) out[i] = (float)in[i];   // the rounding numpy's astype(float32) / PyArray_CastToType does (swig/vector.i:42-55)
}
// host array (float or double) -> float buffer in HBM
static void upload_as_float(DevBuf<float>& dst, const float* src, int n) { dst.upload(src, n); }
static void upload_as_float(DevBuf<float>& dst, const double* src, int n) {
    DevBuf<double> tmp;
    tmp.upload(src, n);
    dst.get(n);
    hipLaunchKernelGGL(k_cast_f64, dim3((n + 255) / 256), dim3(256), 0, stream(), tmp.p, n, dst.p);
    GPP_HIP(hipGetLastError());
    GPP_HIP(hipStreamSynchronize(stream()));   // tmp dies here
}
// lats / lons: the caller's arrays (they become the persistent d_lat / d_lon of the set)
template <class T>
static bool convert_all_device(gpp_points* p, const T* lats, const T* lons) {
    const int n = p->n;
    DevBuf<int> d_flag, d_cnt;
    const int cap = 1 << 16;
    upload_as_float(p->d_lat, lats, n); upload_as_float(p->d_lon, lons, n);
    p->latlon_on_device = true;
    p->d_x.get(n); p->d_y.get(n); p->d_z.get(n);
    d_flag.get(cap); d_cnt.get(2);
    GPP_HIP(hipMemsetAsync(d_cnt.p, 0, 2 * sizeof(int), stream()));
    hipLaunchKernelGGL(k_convert, dim3((n + 255) / 256), dim3(256), 0, stream(), p->d_lat.p, p->d_lon.p, n, p->type, p->d_x.p, p->d_y.p, p->d_z.p, d_flag.p, cap, d_cnt.p);
    GPP_HIP(hipGetLastError());
    int cnt[2] = {0, 0};
    GPP_HIP(hipMemcpyAsync(cnt, d_cnt.p, sizeof(cnt), hipMemcpyDeviceToHost, stream()));
    GPP_HIP(hipStreamSynchronize(stream()));
    if(cnt[1]) invalid("Invalid coords");   // util.cpp:596-600
    if(cnt[0] > cap) return false;
    if(cnt[0] > 0) {
        std::vector<int> idx(cnt[0]);
        GPP_HIP(hipMemcpy(idx.data(), d_flag.p, sizeof(int) * cnt[0], hipMemcpyDeviceToHost));
        std::vector<float> v(3 * (size_t)cnt[0]);
        for(int k = 0; k < cnt[0]; k++) {
            float lat = (float)lats[idx[k]], lon = (float)lons[idx[k]];
            convert_range(&lat, &lon, 0, 1, p->type, &v[3 * k], &v[3 * k + 1], &v[3 * k + 2]);
        }
        DevBuf<float> d_v;
        DevBuf<int> d_i;
        d_v.upload(v.data(), v.size()); d_i.upload(idx.data(), idx.size());
        hipLaunchKernelGGL(k_patch, dim3((cnt[0] + 255) / 256), dim3(256), 0, stream(), d_i.p, d_v.p, cnt[0], p->d_x.p, p->d_y.p, p->d_z.p);
        GPP_HIP(hipGetLastError());
        GPP_HIP(hipStreamSynchronize(stream()));
    }
    return true;
}

// ---- point sets ---------------------------------------------------------------
void gpp_points::ensure_host_xyz() {
    if(host_xyz) return;
    x.resize(n); y.resize(n); z.resize(n);
    GPP_HIP(hipMemcpy(x.data(), d_x.p, sizeof(float) * n, hipMemcpyDeviceToHost));
    GPP_HIP(hipMemcpy(y.data(), d_y.p, sizeof(float) * n, hipMemcpyDeviceToHost));
    GPP_HIP(hipMemcpy(z.data(), d_z.p, sizeof(float) * n, hipMemcpyDeviceToHost));
    host_xyz = true;
}
void gpp_points::to_device() {
    if(on_device) return;
    if(host_xyz) {   // (a set converted on the device has its x / y / z there already)
        d_x.upload(x.data(), n);
        d_y.upload(y.data(), n);
        d_z.upload(z.data(), n);
    }
    if(host_fields) {   // (otherwise make_points put them there)
        d_elev.upload(elevs.data(), n);
        d_laf.upload(lafs.data(), n);
    }
    GPP_HIP(hipStreamSynchronize(gpp::stream()));
    on_device = true;
}
void gpp_points::ensure_host_fields() {
    if(host_fields) return;
    lats.resize(n); lons.resize(n); elevs.resize(n); lafs.resize(n);
    GPP_HIP(hipMemcpy(lats.data(), d_lat.p, sizeof(float) * n, hipMemcpyDeviceToHost));
    GPP_HIP(hipMemcpy(lons.data(), d_lon.p, sizeof(float) * n, hipMemcpyDeviceToHost));
    GPP_HIP(hipMemcpy(elevs.data(), d_elev.p, sizeof(float) * n, hipMemcpyDeviceToHost));
    GPP_HIP(hipMemcpy(lafs.data(), d_laf.p, sizeof(float) * n, hipMemcpyDeviceToHost));
    host_fields = true;
}
float gpp_points::lat_at(int i) {
    if(host_fields) return lats[i];
    float v;
    GPP_HIP(hipMemcpy(&v, d_lat.p + i, sizeof(float), hipMemcpyDeviceToHost));
    return v;
}
float gpp_points::lon_at(int i) {
    if(host_fields) return lons[i];
    float v;
    GPP_HIP(hipMemcpy(&v, d_lon.p + i, sizeof(float), hipMemcpyDeviceToHost));
    return v;
}
void gpp_free_obs_index(gpp_obs_index*);
struct gpp_nn_index { int unused; };
void gpp_free_nn_index(gpp_nn_index* p) { delete p; }
gpp_points::~gpp_points() {
    g_live_handles.fetch_sub(1);
    if(obs_index) gpp_free_obs_index(obs_index);
    if(nn_index) gpp_free_nn_index(nn_index);
}

__global__ void k_fill_f(float* out, int n, float v) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n) out[i] = v;
}
template <class T>
static gpp_points* make_points(const T* lats, const T* lons, const T* elevs, const T* lafs, int n, int ny, int nx, int type) {
    if(n < 0) invalid("negative size");
    if(n > 0 && (!lats || !lons)) invalid("lats/lons are NULL");
    std::unique_ptr<gpp_points> p(new gpp_points);
    p->serial = g_next_serial.fetch_add(1); g_live_handles.fetch_add(1);
    p->n = n; p->ny = ny; p->nx = nx; p->type = type;
    // do the vertical / land-area-fraction factors of a structure function vary over this point set at all?
    auto uniform = [](const T* v, int m) {   // absent (all NaN: points.cpp:23-30, grid.cpp:41-54), all invalid, or all valid and equal
        if(!v || m == 0) return true;
        const float first = (float)v[0];
        const bool inv0 = std::isnan(first) || std::isinf(first);
        for(int i = 0; i < m; i++) {
            const float e = (float)v[i];
            const bool inv = std::isnan(e) || std::isinf(e);
            if(inv != inv0 || (!inv && e != first)) return false;
        }
        return true;
    };
    p->elev_uniform = uniform(elevs, n); p->laf_uniform = uniform(lafs, n);
    // large sets: conversion on the device straight from the caller's arrays; neither x / y / z nor the four fields get a host
    // copy unless a host-side function asks for one
    bool done = false;
    if(n >= (1 << 16) && !path_env("GPP_HOST_CONVERT")) {
        if(type != GPP_GEODETIC && type != GPP_CARTESIAN) invalid("Unknown coordinate type");
        ensure_device();
        p->host_xyz = false;
        done = convert_all_device(p.get(), lats, lons);
        if(!done) { p->host_xyz = true; p->latlon_on_device = false; }
    }
    if(done) {
        p->host_fields = false;
        if(elevs) upload_as_float(p->d_elev, elevs, n);
        else { p->d_elev.get(n); hipLaunchKernelGGL(k_fill_f, dim3((n + 255) / 256), dim3(256), 0, stream(), p->d_elev.p, n, NAN); }
        if(lafs) upload_as_float(p->d_laf, lafs, n);
        else { p->d_laf.get(n); hipLaunchKernelGGL(k_fill_f, dim3((n + 255) / 256), dim3(256), 0, stream(), p->d_laf.p, n, NAN); }
        GPP_HIP(hipGetLastError());
        GPP_HIP(hipStreamSynchronize(stream()));   // the caller's arrays may go away after this call
        p->on_device = true;
        return p.release();
    }
    p->lats.assign(lats, lats + n);      // (element-wise conversion to float for double sources)
    p->lons.assign(lons, lons + n);
    if(elevs) p->elevs.assign(elevs, elevs + n); else p->elevs.assign(n, NAN);
    if(lafs) p->lafs.assign(lafs, lafs + n); else p->lafs.assign(n, NAN);
    p->x.resize(n); p->y.resize(n); p->z.resize(n);
    convert_all(p->lats.data(), p->lons.data(), n, type, p->x.data(), p->y.data(), p->z.data());
    return p.release();
}

extern "C" int gpp_points_create(const float* lats, const float* lons, const float* elevs, const float* lafs, int n, int type, gpp_points** out) {
    GPP_TRY
    if(!out) invalid("out is NULL");
    *out = make_points(lats, lons, elevs, lafs, n, 0, 0, type);
    return GPP_OK;
    GPP_CATCH
}
extern "C" int gpp_grid_create(const float* lats, const float* lons, const float* elevs, const float* lafs, int ny, int nx, int type, gpp_points** out) {
    GPP_TRY
    if(!out) invalid("out is NULL");
    if(ny < 0 || nx < 0) invalid("negative grid size");
    if((long)ny * nx > 0x7fffffffL) invalid("grid too large");
    *out = make_points(lats, lons, elevs, lafs, ny * nx, ny, nx, type);
    return GPP_OK;
    GPP_CATCH
}
// The same from float64 arrays (what numpy hands over by default): large sets are cast to float32 on the device, which costs one
// upload instead of numpy's single-threaded astype on 8-byte elements
extern "C" int gpp_points_create_f64(const double* lats, const double* lons, const double* elevs, const double* lafs, int n, int type, gpp_points** out) {
    GPP_TRY
    if(!out) invalid("out is NULL");
    *out = make_points(lats, lons, elevs, lafs, n, 0, 0, type);
    return GPP_OK;
    GPP_CATCH
}
extern "C" int gpp_grid_create_f64(const double* lats, const double* lons, const double* elevs, const double* lafs, int ny, int nx, int type, gpp_points** out) {
    GPP_TRY
    if(!out) invalid("out is NULL");
    if(ny < 0 || nx < 0) invalid("negative grid size");
    if((long)ny * nx > 0x7fffffffL) invalid("grid too large");
    *out = make_points(lats, lons, elevs, lafs, ny * nx, ny, nx, type);
    return GPP_OK;
    GPP_CATCH
}
extern "C" int gpp_points_destroy(gpp_points* p) {
    GPP_TRY
    gpp_oi_drain_pending();   // (a pending GPP_ASYNC call of this thread may still use the handle)
    // A deferred call of another thread is not in this thread's queue, and its kernels read the handle's arrays and its remembered list of
    // declined tiles: whatever is in flight on the three streams ends before they are freed (a handle nothing was enqueued
    // for costs three calls that return at once).
    if(p && g_stream) {
        GPP_HIP(hipStreamSynchronize(g_stream));
        if(g_stream2) GPP_HIP(hipStreamSynchronize(g_stream2));
        if(g_stream3) GPP_HIP(hipStreamSynchronize(g_stream3));
        if(g_stream4) GPP_HIP(hipStreamSynchronize(g_stream4));
    }
    delete p;
    return GPP_OK;
    GPP_CATCH
}
extern "C" int gpp_points_size(const gpp_points* p, int* n, int* ny, int* nx, int* type) {
    GPP_TRY
    if(!p) invalid("points is NULL");
    if(n) *n = p->n;
    if(ny) *ny = p->ny;
    if(nx) *nx = p->nx;
    if(type) *type = p->type;
    return GPP_OK;
    GPP_CATCH
}
extern "C" int gpp_points_get(const gpp_points* p, int field, float* out) {
    GPP_TRY
    if(!p) invalid("points is NULL");
    const std::vector<float>* v = nullptr;
    if(field >= 4) const_cast<gpp_points*>(p)->ensure_host_xyz();
    else const_cast<gpp_points*>(p)->ensure_host_fields();
    switch(field) {
        case 0: v = &p->lats; break;
        case 1: v = &p->lons; break;
        case 2: v = &p->elevs; break;
        case 3: v = &p->lafs; break;
        case 4: v = &p->x; break;
        case 5: v = &p->y; break;
        case 6: v = &p->z; break;
        default: invalid("unknown field");
    }
    if(p->n) memcpy(out, v->data(), sizeof(float) * p->n);
    return GPP_OK;
    GPP_CATCH
}

// ---- radius query for a single location (host; API completeness, not the hot path) ----
// Exact restatement of kdtree.cpp:39-60 + within_radius :247-260 on the float32 x/y/z.
#pragma clang fp contract(off)
// (KDTree::get_neighbours / get_closest_neighbours for single locations: radius.hip)

// ---- nearest neighbour (device, brute force over the point set; one wave per query) ----
// Metric = float32 squared chord distance in the reference's operation order (Boost
// comparable_distance on point<float,3>); ties -> lowest index (R-tree order is unspecified).
__global__ __launch_bounds__(256) void k_nearest_bruteforce(const float* __restrict__ px, const float* __restrict__ py,
                                                            const float* __restrict__ pz, int n,
                                                            const float* __restrict__ qx, const float* __restrict__ qy,
                                                            const float* __restrict__ qz, int nq, int include_match,
                                                            int* __restrict__ out) {
    int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    int lane = threadIdx.x & 63;
    if(wave >= nq) return;
    float x = qx[wave], y = qy[wave], z = qz[wave];
    float best = INFINITY;
    int besti = 0x7fffffff;
    for(int i = lane; i < n; i += 64) {
        float ax = px[i], ay = py[i], az = pz[i];
        if(!include_match && ax == x && ay == y && az == z) continue;   // kdtree.cpp:265-270
        float dx = ax - x, dy = ay - y, dz = az - z;
        float s = dx * dx + dy * dy;
        s = s + dz * dz;
        if(s < best || (s == best && i < besti)) { best = s; besti = i; }
    }
    for(int off = 32; off > 0; off >>= 1) {
        float ob = __shfl_xor(best, off);
        int oi = __shfl_xor(besti, off);
        if(ob < best || (ob == best && oi < besti)) { best = ob; besti = oi; }
    }
    if(lane == 0) out[wave] = (besti == 0x7fffffff) ? -1 : besti;
}

// Same metric and tie rule through the bin index of the point set (gpp_obs_index): rings of bins around the query's bin are
// visited until no unvisited bin can hold a point at least as close (points of ring r+1 are >= r bin widths away in the
// projection, hence in 3-D).  One thread per query.
__global__ __launch_bounds__(256) void k_nearest_binned(const float4* __restrict__ sgeo, const float2* __restrict__ smeta,
                                                        const int* __restrict__ bin_start, int axis_a, int axis_b, int nbx, int nby,
                                                        float amin, float bmin, float inv_s,
                                                        const float* __restrict__ qx, const float* __restrict__ qy, const float* __restrict__ qz,
                                                        int nq, int include_match, int* __restrict__ out) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if(q >= nq) return;
    const float x = qx[q], y = qy[q], z = qz[q];
    const float qa = axis_a == 0 ? x : (axis_a == 1 ? y : z);
    const float qb = axis_b == 1 ? y : (axis_b == 2 ? z : x);
    const int cbx = min(max((int)floorf((qa - amin) * inv_s), 0), nbx - 1);
    const int cby = min(max((int)floorf((qb - bmin) * inv_s), 0), nby - 1);
    const float sbin = 1.0f / inv_s;
    float best = INFINITY;
    int besti = 0x7fffffff;
    auto visit = [&](int row, int xa, int xb) {
        if(row < 0 || row >= nby) return;
        xa = max(xa, 0); xb = min(xb, nbx - 1);
        if(xa > xb) return;
        const int js = bin_start[row * nbx + xa], je = bin_start[row * nbx + xb + 1];
        for(int j = js; j < je; ++j) {
            const float4 g = sgeo[j];
            if(!include_match && g.x == x && g.y == y && g.z == z) continue;   // kdtree.cpp:265-270
            const float dx = g.x - x, dy = g.y - y, dz = g.z - z;
            float s = dx * dx + dy * dy;
            s = s + dz * dz;
            const int o = __float_as_int(smeta[j].y);
            if(s < best || (s == best && o < besti)) { best = s; besti = o; }
        }
    };
    const int rmax = max(max(cbx, nbx - 1 - cbx), max(cby, nby - 1 - cby));
    for(int r = 0; r <= rmax; ++r) {
        if(r >= 2) { const float lb = (float)(r - 1) * sbin * 0.999f; if(best < lb * lb) break; }
        if(r == 0) visit(cby, cbx, cbx);
        else {
            visit(cby - r, cbx - r, cbx + r);
            visit(cby + r, cbx - r, cbx + r);
            for(int row = cby - r + 1; row <= cby + r - 1; ++row) { visit(row, cbx - r, cbx - r); visit(row, cbx + r, cbx + r); }
        }
    }
    out[q] = (besti == 0x7fffffff) ? -1 : besti;
}

void gpp_nearest_device(gpp_points* p, const float* d_qx, const float* d_qy, const float* d_qz, int nq, int include_match, int* d_out) {
    p->to_device();
    if(nq == 0) return;
    if(p->n > 2048 && !path_env("GPP_NN_BRUTE")) {
        gpp_obs_index* ix = gpp_build_obs_index(p);
        hipLaunchKernelGGL(k_nearest_binned, dim3((nq + 255) / 256), dim3(256), 0, stream(), ix->d_sgeo.p, ix->d_smeta.p, ix->d_bin_start.p,
                           ix->axis_a, ix->axis_b, ix->nbx, ix->nby, ix->amin, ix->bmin, ix->inv_s, d_qx, d_qy, d_qz, nq, include_match, d_out);
        GPP_HIP(hipGetLastError());
        return;
    }
    int waves_per_block = 4;
    int blocks = (nq + waves_per_block - 1) / waves_per_block;
    hipLaunchKernelGGL(k_nearest_bruteforce, dim3(blocks), dim3(256), 0, stream(), p->d_x.p, p->d_y.p, p->d_z.p, p->n,
                       d_qx, d_qy, d_qz, nq, include_match, d_out);
    GPP_HIP(hipGetLastError());
}

extern "C" int gpp_points_nearest_neighbour(gpp_points* p, const float* qlats, const float* qlons, int nq, int include_match, int* indices) {
    GPP_TRY
    if(!p) invalid("points is NULL");
    if(nq < 0) invalid("nq < 0");
    if(nq == 0) return GPP_OK;
    if(p->n == 0) { for(int i = 0; i < nq; i++) indices[i] = -1; return GPP_OK; }   // points.cpp:55-61
    std::vector<float> qx(nq), qy(nq), qz(nq);
    convert_all(qlats, qlons, nq, p->type, qx.data(), qy.data(), qz.data());
    DevBuf<float> dx, dy, dz;
    DevBuf<int> dout;
    dx.upload(qx.data(), nq); dy.upload(qy.data(), nq); dz.upload(qz.data(), nq);
    dout.get(nq);
    gpp_nearest_device(p, dx.p, dy.p, dz.p, nq, include_match, dout.p);
    GPP_HIP(hipMemcpyAsync(indices, dout.p, sizeof(int) * nq, hipMemcpyDeviceToHost, stream()));
    GPP_HIP(hipStreamSynchronize(stream()));
    return GPP_OK;
    GPP_CATCH
}

__global__ void k_gather(const float* __restrict__ values, const int* __restrict__ idx, int n, float* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n) out[i] = idx[i] >= 0 ? values[idx[i]] : NAN;
}
__global__ void k_fill(float* out, int n, float v) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n) out[i] = v;
}

// values [nt][n_from] -> out [nt][nq]: the index is looked up once per location, the levels are gathered with unit-stride
// writes (nearest.cpp:47-66 makes the same split "because we do not want time to be the inner loop")
__global__ void k_gather_levels(const float* __restrict__ values, const int* __restrict__ idx, int nq, size_t nfrom, int nt,
                                float* __restrict__ out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= nq) return;
    const int j = idx[i];
    for(int t = 0; t < nt; ++t) out[(size_t)t * nq + i] = j >= 0 ? values[(size_t)t * nfrom + j] : NAN;
}
__global__ void k_fill_long(float* out, size_t n, float v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n) out[i] = v;
}

extern "C" int gpp_nearest_levels(gpp_points* from, gpp_points* to, const float* values, int nt, float* out, int mem) {
    GPP_TRY
    if(!from || !to) invalid("points is NULL");
    if(from->type != to->type) invalid("Coordinate types must be the same");
    if(nt < 0) invalid("negative number of time levels");
    int nq = to->n;
    if(nq == 0 || nt == 0) return GPP_OK;
    const size_t nout = (size_t)nt * nq;
    OutField o;
    o.bind(out, nout, mem);
    if(from->n == 0) {   // nearest.cpp:132-134
        hipLaunchKernelGGL(k_fill_long, dim3((unsigned)((nout + 255) / 256)), dim3(256), 0, stream(), o.d, nout, NAN);
        GPP_HIP(hipGetLastError());
    }
    else {
        if(!values) invalid("values is NULL");
        InField v;
        v.bind(values, (size_t)nt * from->n, mem);
        to->to_device();
        DevBuf<int> idx;
        idx.get(nq);
        gpp_nearest_device(from, to->d_x.p, to->d_y.p, to->d_z.p, nq, 1, idx.p);
        if(nt == 1) hipLaunchKernelGGL(k_gather, dim3((nq + 255) / 256), dim3(256), 0, stream(), v.d, idx.p, nq, o.d);
        else hipLaunchKernelGGL(k_gather_levels, dim3((nq + 255) / 256), dim3(256), 0, stream(), v.d, idx.p, nq, (size_t)from->n, nt, o.d);
        GPP_HIP(hipGetLastError());
        o.finish();
        GPP_HIP(hipStreamSynchronize(stream()));   // staged buffers die with this scope
        return GPP_OK;
    }
    o.finish();
    GPP_HIP(hipStreamSynchronize(stream()));
    return GPP_OK;
    GPP_CATCH
}

extern "C" int gpp_nearest(gpp_points* from, gpp_points* to, const float* values, float* out, int mem) {
    return gpp_nearest_levels(from, to, values, 1, out, mem);
}
