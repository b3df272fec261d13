// Shared host-side definitions of libgridpp_hip.so (not part of the C-ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>
#include <mutex>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <vector>
#include "../../include/gridpp_hip.h"

namespace gpp {

// ---- errors ---------------------------------------------------------------
struct Error {
    int code;
    std::string msg;
};
void set_error(const char* msg);
int fail(int code, const char* fmt, ...);

#define GPP_HIP(expr)                                                                              \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if(e_ != hipSuccess)                                                                       \
            throw gpp::Error{GPP_ERUNTIME, std::string(#expr) + ": " + hipGetErrorString(e_)};     \
    } while(0)

// Wrap the body of every extern "C" entry point: no exception crosses the ABI, and ONE call runs at a time.  The reference's objects
// are immutable and its queries const, so it may be called from several host threads at once (src/api/oi.cpp:221-233 does so itself);
// here the handles carry lazily built state (device copies, the observation index, the memo of the last OI call) and every call
// goes to the one library stream, so the library serialises the calls itself (a recursive lock: entry points call each other).
// Workspaces, statistics and the error message are per thread.  tests/test_gpu_threads.py.
std::recursive_mutex& api_mutex_ref();
#define GPP_TRY try { std::lock_guard<std::recursive_mutex> gpp_api_lock_(gpp::api_mutex_ref());
#define GPP_CATCH                                                             \
    }                                                                         \
    catch(const gpp::Error& e) { gpp::set_error(e.msg.c_str()); return e.code; } \
    catch(const std::exception& e) { gpp::set_error(e.what()); return GPP_ERUNTIME; } \
    catch(...) { gpp::set_error("Unknown exception"); return GPP_ERUNTIME; }

[[noreturn]] inline void invalid(const std::string& m) { throw Error{GPP_EINVAL, m}; }
[[noreturn]] inline void runtime(const std::string& m) { throw Error{GPP_ERUNTIME, m}; }

// ---- switches --------------------------------------------------------------
// timing_env(): environment switches that SKIP work, print statistics or change accuracy (GPP_OI_DEBUG, GPP_ENSI_DEBUG,
// GPP_SCAN_STATS, GPP_ENSI_STATS) exist only in diagnostic builds (-DGPP_TIMING_SWITCHES, tools/variant.sh): the product
// library never reads them and GPP_DBG() is the constant 0 there.
// path_env(): switches that choose between implementations returning the same results (tests force the rarely taken paths
// with them, A/B timings compare them).  gpp_active_overrides() lists the ones that are set; bench.py refuses to run with any.
#ifdef GPP_TIMING_SWITCHES
inline const char* timing_env(const char* n) { return getenv(n); }
#define GPP_DBG(a, bits) ((a).debug & (bits))
#else
inline const char* timing_env(const char*) { return nullptr; }
#define GPP_DBG(a, bits) 0
#endif
// (round 4: NOT the environment any more -- the library reads no environment variable.  An override exists only after an explicit
//  gpp_set_path_override(name, value) call, the hook the tests and the A/B tools use; the Python mirror forwards the GPP_* variables of
//  its process to it once at load time so that `GPP_OI_NO_UNION=1 python tools/...` keeps working.)
const char* path_override(const char* name);   // runtime.hip; NULL when not set
inline const char* path_env(const char* n) { return path_override(n); }

// ---- device runtime ----------------------------------------------------------
hipStream_t stream();   // library stream (created on first use, after the device is chosen)
hipStream_t stream3();  // and a third (status copies of deferred calls)
hipStream_t stream4();  // and a fourth (every other band of the banded host path of optimal_interpolation: the last, partly filled round of a band's workgroups beside the next band)
hipStream_t stream2();  // a second one: work that runs BESIDE the library stream inside one call (ordered against it with events)
void ensure_device();   // throws GPP_ENODEVICE when no GPU is visible

void stage_release_all();   // (runtime.hip: the pool of staging buffers, see Staged below)
template <class T>
struct DevBuf {   // owning HBM buffer, grows on demand, never shrinks
    T* p = nullptr;
    size_t cap = 0;
    unsigned long long gen = 0;   // counts allocations: a cache of "what this buffer holds" keys on it (a new allocation may come back at the old address)
    ~DevBuf() { if(p) (void)hipFree(p); }
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    T* get(size_t n) {
        if(n > cap) {
            T* old = p;
            p = nullptr;
            cap = 0;   // (a failed hipMalloc below must not leave a capacity behind a null pointer)
            if(old) GPP_HIP(hipFree(old));
            if(hipMalloc((void**)&p, (n ? n : 1) * sizeof(T)) != hipSuccess) {   // out of memory while the staging pool holds idle buffers (up to 1 GiB): give them back, once
                (void)hipGetLastError();
                p = nullptr;
                stage_release_all();
                GPP_HIP(hipMalloc((void**)&p, (n ? n : 1) * sizeof(T)));
            }
            cap = n;
            gen++;
#ifdef GPP_POISON
            // diagnostic build (tools/hostile/build.sh): a fresh allocation is as hostile as a poisoned workspace (hipMalloc tends to
            // hand out zero pages, which hide a read of something never written)
            GPP_HIP(hipMemsetAsync(p, 0xFF, (n ? n : 1) * sizeof(T), stream()));
#endif
        }
        return p;
    }
#ifdef GPP_POISON
    void poison(int byte) { if(p && cap) GPP_HIP(hipMemsetAsync(p, byte, cap * sizeof(T), stream())); }
#endif
    void release() {
        if(p) (void)hipFree(p);
        p = nullptr; cap = 0;
        gen++;
    }
    void upload(const T* h, size_t n) {
        get(n);
        if(n) GPP_HIP(hipMemcpyAsync(p, h, n * sizeof(T), hipMemcpyHostToDevice, stream()));
    }
};

// A field argument of the C-ABI resolved to a device pointer: either the caller's
// HBM pointer (GPP_MEM_DEVICE) or a staged copy (GPP_MEM_HOST).
static __global__ void k_stage_f64(const double* __restrict__ in, size_t n, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n) out[i] = (float)in[i];   // the rounding of the reference's PyArray_CastToType (swig/vector.i:42-55)
}
// Staging buffers of the host-side fields of a call (round 5): borrowed from a small pool (runtime.hip) instead of one hipMalloc + hipFree per field and
// call -- for a 1000 x 1000 field those cost as much as the two copies over PCIe.  All work of the library goes to its one stream in call order, so a buffer
// returned at the end of a call (even one that failed) is not written by the next borrower before its last reader has run.  Buffers of up to 256 MiB are
// kept, 1 GiB in all (the largest go first); gpp_release_workspaces() and gpp_set_device() empty the pool.
void* stage_borrow(size_t bytes, size_t* cap);
void stage_return(void* p, size_t cap);
void stage_release_all();
template <class T>
struct Staged {
    T* p = nullptr;
    size_t capb = 0;
    Staged() = default;
    Staged(const Staged&) = delete;
    Staged& operator=(const Staged&) = delete;
    ~Staged() { if(p) stage_return(p, capb); }
    T* get(size_t n) {
        const size_t bytes = (n ? n : 1) * sizeof(T);
        if(bytes > capb) {
            if(p) stage_return(p, capb);
            p = nullptr; capb = 0;
            p = static_cast<T*>(stage_borrow(bytes, &capb));
        }
        return p;
    }
    void upload(const T* h, size_t n) {
        get(n);
        if(n) GPP_HIP(hipMemcpyAsync(p, h, n * sizeof(T), hipMemcpyHostToDevice, stream()));
    }
};

struct InField {
    Staged<float> staged;
    const float* d = nullptr;
    void bind(const float* src, size_t n, int mem) {
        if(!src) { d = nullptr; return; }
        if(mem & GPP_MEM_DEVICE) d = src;
        else if(mem & GPP_HOST_F64) {   // the host array holds doubles: one upload + a cast on the device
            Staged<double> wide;
            wide.upload(reinterpret_cast<const double*>(src), n);
            staged.get(n);
            if(n) hipLaunchKernelGGL(k_stage_f64, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream(), wide.p, n, staged.p);
            GPP_HIP(hipGetLastError());
            GPP_HIP(hipStreamSynchronize(stream()));   // `wide` is released here
            d = staged.p;
        }
        else { staged.upload(src, n); d = staged.p; }
    }
};
struct OutField {
    Staged<float> staged;
    float* d = nullptr;
    float* host = nullptr;
    size_t n = 0;
    void bind(float* dst, size_t n_, int mem) {
        n = n_;
        if(!dst) { d = nullptr; host = nullptr; return; }
        if(mem & GPP_MEM_DEVICE) { d = dst; host = nullptr; }
        else { d = staged.get(n); host = dst; }
    }
    void finish() {
        if(host && n) GPP_HIP(hipMemcpyAsync(host, d, n * sizeof(float), hipMemcpyDeviceToHost, stream()));
    }
};

inline bool is_valid(float v) { return !std::isnan(v) && !std::isinf(v); }   // src/api/util.cpp:16-18

}   // namespace gpp

// ---- point set ---------------------------------------------------------------
struct gpp_obs_index;   // oi.hip: bin-sorted observation block
struct gpp_nn_index;    // runtime.hip: uniform-cell index for nearest-neighbour queries

// runtime.hip: nearest-neighbour indices of nq device-resident query points in p (exact metric, ties -> lowest index)
void gpp_nearest_device(gpp_points* p, const float* d_qx, const float* d_qy, const float* d_qz, int nq, int include_match, int* d_out);

struct gpp_points {
    int n = 0, ny = 0, nx = 0, type = GPP_GEODETIC;
    std::vector<float> lats, lons, elevs, lafs, x, y, z;   // host copies (float32, as the reference stores them)
    gpp::DevBuf<float> d_x, d_y, d_z, d_elev, d_laf;       // HBM-resident SoA
    bool on_device = false;
    gpp::DevBuf<float> d_lat, d_lon;                       // lat / lon in HBM, uploaded on first use (bilinear.hip)
    bool latlon_on_device = false;
    void latlon_to_device();
    bool host_xyz = true;            // x, y, z are present on the host (large sets are converted on the device and
    void ensure_host_xyz();          // downloaded only when a host-side function asks for them)
    // large sets keep lats / lons / elevs / lafs in HBM only (d_lat, d_lon, d_elev, d_laf): the four host vectors are filled by
    // ensure_host_fields() the first time a host-side function asks for them (a 4000 x 4000 grid would otherwise spend most of
    // its construction time page-faulting 256 MB of copies nobody reads)
    bool host_fields = true;
    void ensure_host_fields();
    int tile_wshift = -1;            // gpp_tile_wshift's answer for this grid (computed once)
    float lat_at(int i);             // single elements without materialising the vectors
    float lon_at(int i);
    bool elev_uniform = true, laf_uniform = true;   // every point has the same elevation / laf (or none has one)
    // memo of the last OI call with this point set as the background: did k_oi_union pay? (same observations handle and
    // structure scales -> same geometry -> same answer; the observation VALUES do not matter)
    // (keyed on the observation set's serial number, not its address: a new handle may reuse the address of a destroyed one)
    struct { unsigned long long points_id = 0; float h = 0, v = 0, w = 0; int kh = -1, kv = -1, kw = -1, cv = -1; int max_points = -1; float declined = 0; int leftover = -1; int n1 = 0;
             // the declined tiles themselves (round 5): a list in HBM, one flag byte per tile for the first pass, and the list's length as a device
             // int -- the list passes of the NEXT call with this geometry run from it on a second stream while the first pass skips those tiles
             gpp::DevBuf<int> list, count; gpp::DevBuf<unsigned char> flags; int nlist = 0, list_ntiles = 0; } union_memo;   // (leftover: 4-cell items the last call left to k_oi; n1: tiles its first pass declined)
    unsigned long long serial = 0;   // unique per handle, assigned at creation
    gpp_obs_index* obs_index = nullptr;
    gpp_nn_index* nn_index = nullptr;
    void to_device();
    ~gpp_points();
};
