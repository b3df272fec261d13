// Radius-query consumers on the device: count (src/api/count.cpp:6-66), gridding / gridding_nearest
// (src/api/gridding.cpp:6-131) and the batched form of KDTree::get_neighbours(_with_distance) / get_num_neighbours
// (src/api/kdtree.cpp:39-64,241-260).
//
// All of them walk the point set's bin index (gpp_obs_index, the same one the OI kernels and the nearest-neighbour search
// use): one thread per output location visits the bins that overlap the axis-aligned box +-radius in the two indexed
// axes and applies the reference's test to every point in them -- strictly inside the box, then chord length <= radius
// in float32.  Variable-length results go through a CSR layout: a counting pass, a device-wide exclusive scan (rocPRIM),
// a filling pass.  Statistics of a location's values are the sequential float loops of util.cpp:19-178 (row_stats.h).
#include "common.h"
#include "oi_common.h"
#include "row_stats.h"
#include <rocprim/rocprim.hpp>
#include <algorithm>

using namespace gpp;

namespace {

struct IxView {
    const float4* sgeo;
    const float2* smeta;
    const int* bin_start;
    int axis_a, axis_b, nbx, nby;
    float amin, bmin, inv_s;
};
IxView view_of(gpp_obs_index* ix) {
    return IxView{ix->d_sgeo.p, ix->d_smeta.p, ix->d_bin_start.p, ix->axis_a, ix->axis_b, ix->nbx, ix->nby, ix->amin, ix->bmin, ix->inv_s};
}

__device__ __forceinline__ int bin_of(float v, float lo, float inv_s, int nb) {
    const float f = floorf((v - lo) * inv_s);
    return (int)fminf(fmaxf(f, 0.0f), (float)(nb - 1));
}

// f(sorted position, original index, distance) for every point the reference's get_neighbours would return
template <class F>
__device__ __forceinline__ void visit_radius(const IxView& ix, float x, float y, float z, float radius, bool include_match, F f) {
    if(!(radius > 0)) return;   // an empty or NaN box holds nothing strictly inside
    const float lox = x - radius, hix = x + radius, loy = y - radius, hiy = y + radius, loz = z - radius, hiz = z + radius;
    const float alo = ix.axis_a == 0 ? lox : (ix.axis_a == 1 ? loy : loz), ahi = ix.axis_a == 0 ? hix : (ix.axis_a == 1 ? hiy : hiz);
    const float blo = ix.axis_b == 1 ? loy : (ix.axis_b == 2 ? loz : lox), bhi = ix.axis_b == 1 ? hiy : (ix.axis_b == 2 ? hiz : hix);
    const int bx0 = bin_of(alo, ix.amin, ix.inv_s, ix.nbx), bx1 = bin_of(ahi, ix.amin, ix.inv_s, ix.nbx);
    const int by0 = bin_of(blo, ix.bmin, ix.inv_s, ix.nby), by1 = bin_of(bhi, ix.bmin, ix.inv_s, ix.nby);
    for(int row = by0; row <= by1; ++row) {
        const int js = ix.bin_start[row * ix.nbx + bx0], je = ix.bin_start[row * ix.nbx + bx1 + 1];
        for(int j = js; j < je; ++j) {
            const float4 g = ix.sgeo[j];
            if(!(g.x > lox && g.x < hix && g.y > loy && g.y < hiy && g.z > loz && g.z < hiz)) continue;   // kdtree.cpp:46,53
            const float dx = g.x - x, dy = g.y - y, dz = g.z - z;
            const float d = sqrtf(dx * dx + dy * dy + dz * dz);                                          // kdtree.cpp:189-194
            if(!(include_match ? d <= radius : (d <= radius && d > 0))) continue;                        // kdtree.cpp:247-260
            f(j, __float_as_int(ix.smeta[j].y), d);
        }
    }
}

__global__ __launch_bounds__(256) void k_radius_count(IxView ix, const float* __restrict__ qx, const float* __restrict__ qy,
                                                      const float* __restrict__ qz, int nq, float radius, int include_match,
                                                      int* __restrict__ cnt, float* __restrict__ cnt_f) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if(q >= nq) return;
    int c = 0;
    visit_radius(ix, qx[q], qy[q], qz[q], radius, include_match != 0, [&](int, int, float) { ++c; });
    if(cnt) cnt[q] = c;
    if(cnt_f) cnt_f[q] = (float)c;
}

// CSR fill: idx / dist / val (each optional) at [offset[q], offset[q] + count)
__global__ __launch_bounds__(256) void k_radius_fill(IxView ix, const float* __restrict__ qx, const float* __restrict__ qy,
                                                     const float* __restrict__ qz, int q0, int nq, float radius, int include_match,
                                                     const long long* __restrict__ offset, long long base, int* __restrict__ idx,
                                                     float* __restrict__ dist, const float* __restrict__ values, float* __restrict__ val) {
    const int q = q0 + blockIdx.x * blockDim.x + threadIdx.x;
    if(q >= q0 + nq) return;
    long long w = offset[q] - base;
    visit_radius(ix, qx[q], qy[q], qz[q], radius, include_match != 0, [&](int, int orig, float d) {
        if(idx) idx[w] = orig;
        if(dist) dist[w] = d;
        if(val) val[w] = values[orig];
        ++w;
    });
}

__device__ __forceinline__ float segment_statistic(const float* row, int n, int statistic) {
    if(statistic == GPP_MEDIAN) return row_quantile(row, n, 0.5f);   // util.cpp:96-105 (calc_quantile 0 / 0.5 / 1)
    return row_statistic(row, n, statistic);
}

// gridding.cpp:23-31 / :52-59 (require_some = false) and :92-97 / :122-126 (require_some = true)
__global__ __launch_bounds__(256) void k_segment_statistic(const float* __restrict__ val, const long long* __restrict__ offset, long long base,
                                                           const int* __restrict__ cnt, int q0, int nq, int min_num, int statistic,
                                                           int require_some, float* __restrict__ out) {
    const int q = q0 + blockIdx.x * blockDim.x + threadIdx.x;
    if(q >= q0 + nq) return;
    const int c = cnt[q];
    float r = NAN;
    if((!require_some || c > 0) && (min_num <= 0 || c >= min_num)) r = segment_statistic(val + (offset[q] - base), c, statistic);
    out[q] = r;
}

// gridding for the statistics that stream (everything but Median): one pass, the accumulators of util.cpp:22-76 fed in
// the walk's order -- the same order the CSR path stores, so both paths give the same bits.
__global__ __launch_bounds__(256) void k_radius_statistic(IxView ix, const float* __restrict__ qx, const float* __restrict__ qy,
                                                          const float* __restrict__ qz, int nq, float radius,
                                                          const float* __restrict__ values, int min_num, int statistic,
                                                          float* __restrict__ out) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if(q >= nq) return;
    int c = 0, count = 0;
    float total = 0, total2 = 0, K = NAN, m = NAN;
    const bool spread = statistic == GPP_STD || statistic == GPP_VARIANCE, lo = statistic == GPP_MIN, hi = statistic == GPP_MAX;
    visit_radius(ix, qx[q], qy[q], qz[q], radius, true, [&](int, int orig, float) {
        ++c;
        const float v = values[orig];
        if(!nv(v)) return;
        if(spread) {
            if(!nv(K)) K = v;
            const float d = v - K;
            total += d; total2 += d * d;
        }
        else if(lo || hi) {
            if(!nv(m)) m = v;
            else if(lo ? v < m : v > m) m = v;
        }
        else total += v;
        ++count;
    });
    float r = NAN;
    if(min_num <= 0 || c >= min_num) {
        if(statistic == GPP_COUNT) r = (float)count;
        else if(lo || hi) r = m;
        else if(count > 0) {
            if(spread) {
                const float mean = total / (float)count, mean2 = total2 / (float)count;
                float var = mean2 - mean * mean;
                if(var < 0) var = 0;
                r = statistic == GPP_STD ? sqrtf(var) : var;
            }
            else if(statistic == GPP_MEAN) r = total / (float)count;
            else if(statistic == GPP_SUM) r = total;
        }
    }
    out[q] = r;
}

// The same pass with one wavefront per location: the 64 lanes test 64 consecutive points of a bin row at once (coalesced
// float4 loads), the hits are then fed to the accumulators one by one in point order through v_readlane, so the result has
// the same bits as the one-thread walk while the loads no longer form a serial latency chain.
__global__ __launch_bounds__(256) void k_radius_statistic_wave(IxView ix, const float* __restrict__ qx, const float* __restrict__ qy,
                                                               const float* __restrict__ qz, int nq, float radius,
                                                               const float* __restrict__ values, int min_num, int statistic,
                                                               float* __restrict__ out) {
    const int q = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if(q >= nq) return;
    const float x = qx[q], y = qy[q], z = qz[q];
    int c = 0, count = 0;
    float total = 0, total2 = 0, K = NAN, m = NAN;
    const bool spread = statistic == GPP_STD || statistic == GPP_VARIANCE, lo = statistic == GPP_MIN, hi = statistic == GPP_MAX;
    if(radius > 0) {
        const float lox = x - radius, hix = x + radius, loy = y - radius, hiy = y + radius, loz = z - radius, hiz = z + radius;
        const float alo = ix.axis_a == 0 ? lox : (ix.axis_a == 1 ? loy : loz), ahi = ix.axis_a == 0 ? hix : (ix.axis_a == 1 ? hiy : hiz);
        const float blo = ix.axis_b == 1 ? loy : (ix.axis_b == 2 ? loz : lox), bhi = ix.axis_b == 1 ? hiy : (ix.axis_b == 2 ? hiz : hix);
        const int bx0 = bin_of(alo, ix.amin, ix.inv_s, ix.nbx), bx1 = bin_of(ahi, ix.amin, ix.inv_s, ix.nbx);
        const int by0 = bin_of(blo, ix.bmin, ix.inv_s, ix.nby), by1 = bin_of(bhi, ix.bmin, ix.inv_s, ix.nby);
        for(int row = by0; row <= by1; ++row) {
            const int js = ix.bin_start[row * ix.nbx + bx0], je = ix.bin_start[row * ix.nbx + bx1 + 1];
            for(int base = js; base < je; base += 64) {
                const int j = base + lane;
                bool in = false;
                float v = 0;
                if(j < je) {
                    const float4 g = ix.sgeo[j];
                    if(g.x > lox && g.x < hix && g.y > loy && g.y < hiy && g.z > loz && g.z < hiz) {
                        const float dx = g.x - x, dy = g.y - y, dz = g.z - z;
                        in = sqrtf(dx * dx + dy * dy + dz * dz) <= radius;
                    }
                    if(in) v = values[__float_as_int(ix.smeta[j].y)];
                }
                unsigned long long mask = __ballot(in);
                c += __popcll(mask);
                while(mask) {
                    const int b = __ffsll((long long)mask) - 1;
                    mask &= mask - 1;
                    const float vb = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), b));
                    if(!nv(vb)) continue;
                    if(spread) {
                        if(!nv(K)) K = vb;
                        const float d = vb - K;
                        total += d; total2 += d * d;
                    }
                    else if(lo || hi) {
                        if(!nv(m)) m = vb;
                        else if(lo ? vb < m : vb > m) m = vb;
                    }
                    else total += vb;
                    ++count;
                }
            }
        }
    }
    if(lane != 0) return;
    float r = NAN;
    if(min_num <= 0 || c >= min_num) {
        if(statistic == GPP_COUNT) r = (float)count;
        else if(lo || hi) r = m;
        else if(count > 0) {
            if(spread) {
                const float mean = total / (float)count, mean2 = total2 / (float)count;
                float var = mean2 - mean * mean;
                if(var < 0) var = 0;
                r = statistic == GPP_STD ? sqrtf(var) : var;
            }
            else if(statistic == GPP_MEAN) r = total / (float)count;
            else if(statistic == GPP_SUM) r = total;
        }
    }
    out[q] = r;
}

__global__ void k_widen(const int* __restrict__ in, int n, long long* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n) out[i] = in[i];
}
__global__ void k_iota(int* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n) out[i] = i;
}
__global__ void k_histogram(const int* __restrict__ target, int n, int* __restrict__ cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n && target[i] >= 0) atomicAdd(&cnt[target[i]], 1);
}
__global__ void k_gather_sorted(const float* __restrict__ values, const int* __restrict__ order, int n, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n) out[i] = values[order[i]];
}
__global__ void k_fill_value(float* out, size_t n, float v) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if(i < n) out[i] = v;
}

// entries per filling pass (2^28 = 1 GiB of float values; GPP_CSR_CAP overrides it so that the tests can reach the chunked path)
long long csr_cap() {
    const char* e = path_env("GPP_CSR_CAP");
    const long long v = e ? atoll(e) : 0;
    return v > 0 ? v : (1ll << 28);
}
#define CSR_CAP csr_cap()

// exclusive scan of the per-location counts into 64-bit offsets [nq + 1]; returns the total
long long scan_counts(const int* cnt, int nq, DevBuf<long long>& wide, DevBuf<long long>& offset) {
    wide.get((size_t)nq + 1);
    offset.get((size_t)nq + 1);
    GPP_HIP(hipMemsetAsync(wide.p + nq, 0, sizeof(long long), stream()));
    hipLaunchKernelGGL(k_widen, dim3((nq + 255) / 256), dim3(256), 0, stream(), cnt, nq, wide.p);
    size_t sb = 0;
    GPP_HIP(rocprim::exclusive_scan((void*)nullptr, sb, wide.p, offset.p, 0LL, (size_t)(nq + 1), rocprim::plus<long long>(), stream()));
    DevBuf<char> tmp;
    tmp.get(sb);
    GPP_HIP(rocprim::exclusive_scan((void*)tmp.p, sb, wide.p, offset.p, 0LL, (size_t)(nq + 1), rocprim::plus<long long>(), stream()));
    long long total = 0;
    GPP_HIP(hipMemcpyAsync(&total, offset.p + nq, sizeof(long long), hipMemcpyDeviceToHost, stream()));
    GPP_HIP(hipStreamSynchronize(stream()));
    return total;
}

// Query chunks [q0, q1) whose CSR segments fit CSR_CAP entries (the offsets come to the host only when one pass is not enough)
std::vector<std::pair<int, int>> chunks_of(const DevBuf<long long>& offset, int nq, long long total) {
    std::vector<std::pair<int, int>> ch;
    if(total <= CSR_CAP) { ch.emplace_back(0, nq); return ch; }
    std::vector<long long> h((size_t)nq + 1);
    GPP_HIP(hipMemcpy(h.data(), offset.p, sizeof(long long) * h.size(), hipMemcpyDeviceToHost));
    int q0 = 0;
    while(q0 < nq) {
        int q1 = (int)(std::upper_bound(h.begin() + q0, h.end(), h[q0] + CSR_CAP) - h.begin()) - 1;
        if(q1 <= q0) q1 = q0 + 1;   // a single location with more than CSR_CAP neighbours still gets its own pass
        ch.emplace_back(q0, q1);
        q0 = q1;
    }
    return ch;
}

void check_same_type(gpp_points* a, gpp_points* b) {
    if(!a || !b) invalid("points is NULL");
}

}   // namespace

// count(input set, output locations, radius): out[i] = number of input points within radius of location i
extern "C" int gpp_count(gpp_points* from, gpp_points* to, float radius, float* out, int mem) {
    GPP_TRY
    ensure_device();
    check_same_type(from, to);
    const int nq = to->n;
    if(nq == 0) return GPP_OK;
    if(!out) invalid("out is NULL");
    OutField o;
    o.bind(out, nq, mem);
    if(from->n == 0) hipLaunchKernelGGL(k_fill_value, dim3((nq + 255) / 256), dim3(256), 0, stream(), o.d, (size_t)nq, 0.0f);
    else {
        to->to_device();
        gpp_obs_index* ix = gpp_build_obs_index(from);
        hipLaunchKernelGGL(k_radius_count, dim3((nq + 255) / 256), dim3(256), 0, stream(), view_of(ix), to->d_x.p, to->d_y.p, to->d_z.p, nq,
                           radius, 1, (int*)nullptr, o.d);
    }
    GPP_HIP(hipGetLastError());
    o.finish();
    GPP_HIP(hipStreamSynchronize(stream()));
    return GPP_OK;
    GPP_CATCH
}

extern "C" int gpp_gridding(gpp_points* to, gpp_points* from, const float* values, float radius, int min_num, int statistic,
                            float* out, int mem) {
    GPP_TRY
    ensure_device();
    check_same_type(from, to);
    if(!is_valid(radius) || radius < 0) invalid("radius must be >= 0");   // gridding.cpp:9-12
    if(min_num < 0) invalid("min_num must be >= 0");
    const int nq = to->n;
    if(nq == 0) return GPP_OK;
    OutField o;
    o.bind(out, nq, mem);
    DevBuf<int> cnt;
    cnt.get(nq);
    DevBuf<long long> wide, offset;
    DevBuf<float> val;
    if(from->n == 0) {   // every neighbour list is empty: calc_statistic of nothing (0 for Count, NaN otherwise), NaN if min_num > 0
        const float v = (min_num <= 0 && statistic == GPP_COUNT) ? 0.0f : NAN;
        hipLaunchKernelGGL(k_fill_value, dim3((nq + 255) / 256), dim3(256), 0, stream(), o.d, (size_t)nq, v);
        GPP_HIP(hipGetLastError());
    }
    else {
        if(!values) invalid("values is NULL");
        InField v;
        v.bind(values, from->n, mem);
        to->to_device();
        gpp_obs_index* ix = gpp_build_obs_index(from);
        const IxView iv = view_of(ix);
        const bool streams = statistic == GPP_MEAN || statistic == GPP_SUM || statistic == GPP_COUNT || statistic == GPP_STD ||
                             statistic == GPP_VARIANCE || statistic == GPP_MIN || statistic == GPP_MAX;
        if(streams && !path_env("GPP_GRIDDING_CSR")) {
            if(path_env("GPP_GRIDDING_THREAD"))
                hipLaunchKernelGGL(k_radius_statistic, dim3((nq + 255) / 256), dim3(256), 0, stream(), iv, to->d_x.p, to->d_y.p, to->d_z.p, nq, radius,
                                   v.d, min_num, statistic, o.d);
            else
                hipLaunchKernelGGL(k_radius_statistic_wave, dim3((nq + 3) / 4), dim3(256), 0, stream(), iv, to->d_x.p, to->d_y.p, to->d_z.p, nq, radius,
                                   v.d, min_num, statistic, o.d);
            GPP_HIP(hipGetLastError());
            o.finish();
            GPP_HIP(hipStreamSynchronize(stream()));
            return GPP_OK;
        }
        hipLaunchKernelGGL(k_radius_count, dim3((nq + 255) / 256), dim3(256), 0, stream(), iv, to->d_x.p, to->d_y.p, to->d_z.p, nq, radius, 1,
                           cnt.p, (float*)nullptr);
        GPP_HIP(hipGetLastError());
        const long long total = scan_counts(cnt.p, nq, wide, offset);
        for(auto ch : chunks_of(offset, nq, total)) {
            const int q0 = ch.first, n = ch.second - ch.first;
            long long base = 0, end = 0;
            if(total > CSR_CAP) {
                GPP_HIP(hipMemcpy(&base, offset.p + q0, sizeof(long long), hipMemcpyDeviceToHost));
                GPP_HIP(hipMemcpy(&end, offset.p + ch.second, sizeof(long long), hipMemcpyDeviceToHost));
            }
            else end = total;
            val.get((size_t)std::max<long long>(end - base, 1));
            hipLaunchKernelGGL(k_radius_fill, dim3((n + 255) / 256), dim3(256), 0, stream(), iv, to->d_x.p, to->d_y.p, to->d_z.p, q0, n, radius, 1,
                               offset.p, base, (int*)nullptr, (float*)nullptr, v.d, val.p);
            hipLaunchKernelGGL(k_segment_statistic, dim3((n + 255) / 256), dim3(256), 0, stream(), val.p, offset.p, base, cnt.p, q0, n, min_num,
                               statistic, 0, o.d);
            GPP_HIP(hipGetLastError());
        }
        o.finish();
        GPP_HIP(hipStreamSynchronize(stream()));
        return GPP_OK;
    }
    o.finish();
    GPP_HIP(hipStreamSynchronize(stream()));
    return GPP_OK;
    GPP_CATCH
}

extern "C" int gpp_gridding_nearest(gpp_points* to, gpp_points* from, const float* values, int min_num, int statistic, float* out, int mem) {
    GPP_TRY
    ensure_device();
    check_same_type(from, to);
    if(min_num < 0) invalid("min_num must be >= 0");   // gridding.cpp:68-69
    const int no = to->n, S = from->n;
    if(no == 0) {
        if(S > 0) runtime("gridding_nearest: no output location to assign the points to");   // the reference indexes an empty vector here
        return GPP_OK;
    }
    OutField o;
    o.bind(out, no, mem);
    if(S == 0) {
        hipLaunchKernelGGL(k_fill_value, dim3((no + 255) / 256), dim3(256), 0, stream(), o.d, (size_t)no, NAN);
        GPP_HIP(hipGetLastError());
        o.finish();
        GPP_HIP(hipStreamSynchronize(stream()));
        return GPP_OK;
    }
    if(!values) invalid("values is NULL");
    InField v;
    v.bind(values, S, mem);
    from->to_device();
    DevBuf<int> target, starget, order, iota, cnt;
    target.get(S); starget.get(S); order.get(S); iota.get(S); cnt.get(no);
    gpp_nearest_device(to, from->d_x.p, from->d_y.p, from->d_z.p, S, 1, target.p);   // gridding.cpp:85-90
    hipLaunchKernelGGL(k_iota, dim3((S + 255) / 256), dim3(256), 0, stream(), iota.p, S);
    // stable sort by target keeps the input order inside every location (the reference push_backs in input order)
    int bits = 1;
    while((1ll << bits) < no) bits++;
    size_t sb = 0;
    GPP_HIP(rocprim::radix_sort_pairs((void*)nullptr, sb, target.p, starget.p, iota.p, order.p, (size_t)S, 0u, (unsigned)bits, stream()));
    DevBuf<char> tmp;
    tmp.get(sb);
    GPP_HIP(rocprim::radix_sort_pairs((void*)tmp.p, sb, target.p, starget.p, iota.p, order.p, (size_t)S, 0u, (unsigned)bits, stream()));
    GPP_HIP(hipMemsetAsync(cnt.p, 0, sizeof(int) * no, stream()));
    hipLaunchKernelGGL(k_histogram, dim3((S + 255) / 256), dim3(256), 0, stream(), target.p, S, cnt.p);
    DevBuf<long long> wide, offset;
    scan_counts(cnt.p, no, wide, offset);
    DevBuf<float> val;
    val.get(S);
    hipLaunchKernelGGL(k_gather_sorted, dim3((S + 255) / 256), dim3(256), 0, stream(), v.d, order.p, S, val.p);
    hipLaunchKernelGGL(k_segment_statistic, dim3((no + 255) / 256), dim3(256), 0, stream(), val.p, offset.p, 0ll, cnt.p, 0, no, min_num, statistic,
                       1, o.d);
    GPP_HIP(hipGetLastError());
    o.finish();
    GPP_HIP(hipStreamSynchronize(stream()));
    return GPP_OK;
    GPP_CATCH
}

// Batched KDTree::get_neighbours(_with_distance): counts[nq] always; when indices / distances are given they receive the
// CSR payload (offsets[nq + 1], capacity `cap` entries) -- a call with cap too small returns the counts only and
// *total, so the caller can size the buffers and call again.  Indices of a location are in ascending order.
extern "C" int gpp_points_get_neighbours_batch(gpp_points* p, const float* qlats, const float* qlons, int nq, float radius, int include_match,
                                               int* counts, long long* offsets, int* indices, float* distances, long long cap, long long* total) {
    GPP_TRY
    ensure_device();
    if(!p || !counts || !total) invalid("NULL argument");
    if(nq < 0) invalid("nq < 0");
    *total = 0;
    if(nq == 0) return GPP_OK;
    if(p->n == 0) {
        for(int i = 0; i < nq; i++) counts[i] = 0;
        if(offsets) for(int i = 0; i <= nq; i++) offsets[i] = 0;
        return GPP_OK;
    }
    std::vector<float> qx(nq), qy(nq), qz(nq);
    if(gpp_convert_coordinates(qlats, qlons, nq, p->type, qx.data(), qy.data(), qz.data()) != GPP_OK) return GPP_EINVAL;
    DevBuf<float> dx, dy, dz;
    dx.upload(qx.data(), nq); dy.upload(qy.data(), nq); dz.upload(qz.data(), nq);
    p->to_device();
    gpp_obs_index* ix = gpp_build_obs_index(p);
    const IxView iv = view_of(ix);
    DevBuf<int> cnt;
    cnt.get(nq);
    hipLaunchKernelGGL(k_radius_count, dim3((nq + 255) / 256), dim3(256), 0, stream(), iv, dx.p, dy.p, dz.p, nq, radius, include_match, cnt.p,
                       (float*)nullptr);
    GPP_HIP(hipGetLastError());
    DevBuf<long long> wide, offset;
    const long long tot = scan_counts(cnt.p, nq, wide, offset);
    *total = tot;
    GPP_HIP(hipMemcpyAsync(counts, cnt.p, sizeof(int) * nq, hipMemcpyDeviceToHost, stream()));
    if(offsets) GPP_HIP(hipMemcpyAsync(offsets, offset.p, sizeof(long long) * ((size_t)nq + 1), hipMemcpyDeviceToHost, stream()));
    GPP_HIP(hipStreamSynchronize(stream()));
    if((!indices && !distances) || tot == 0 || tot > cap) return GPP_OK;
    DevBuf<int> didx;
    DevBuf<float> ddist;
    didx.get((size_t)tot);
    ddist.get((size_t)tot);
    hipLaunchKernelGGL(k_radius_fill, dim3((nq + 255) / 256), dim3(256), 0, stream(), iv, dx.p, dy.p, dz.p, 0, nq, radius, include_match, offset.p,
                       0ll, didx.p, ddist.p, (const float*)nullptr, (float*)nullptr);
    GPP_HIP(hipGetLastError());
    std::vector<int> hi((size_t)tot);
    std::vector<float> hd((size_t)tot);
    GPP_HIP(hipMemcpyAsync(hi.data(), didx.p, sizeof(int) * (size_t)tot, hipMemcpyDeviceToHost, stream()));
    GPP_HIP(hipMemcpyAsync(hd.data(), ddist.p, sizeof(float) * (size_t)tot, hipMemcpyDeviceToHost, stream()));
    GPP_HIP(hipStreamSynchronize(stream()));
    // present every location's list in ascending index order (the walk delivers bin order; the reference's own order is the
    // R-tree's, which is unspecified)
    std::vector<long long> ho((size_t)nq + 1);
    GPP_HIP(hipMemcpy(ho.data(), offset.p, sizeof(long long) * ho.size(), hipMemcpyDeviceToHost));
    std::vector<std::pair<int, float>> seg;
    for(int q = 0; q < nq; q++) {
        const long long a = ho[q], b = ho[q + 1];
        seg.clear();
        for(long long k = a; k < b; k++) seg.emplace_back(hi[k], hd[k]);
        std::sort(seg.begin(), seg.end());
        for(long long k = a; k < b; k++) {
            if(indices) indices[k] = seg[k - a].first;
            if(distances) distances[k] = seg[k - a].second;
        }
    }
    return GPP_OK;
    GPP_CATCH
}

// KDTree::get_neighbours / get_neighbours_with_distance (kdtree.cpp:39-60) for one location: the batch of one
extern "C" int gpp_points_get_neighbours(gpp_points* p, float lat, float lon, float radius, int include_match,
                                         int* indices, float* distances, int cap, int* count) {
    GPP_TRY
    ensure_device();
    if(!p || !count) invalid("NULL argument");
    long long total = 0;
    int c = 0;
    const int rc = gpp_points_get_neighbours_batch(p, &lat, &lon, 1, radius, include_match, &c, nullptr, indices, distances, cap, &total);
    *count = c;
    return rc;
    GPP_CATCH
}

namespace {
// float32 squared chord in the reference's operation order as a sortable key (non-negative floats order like their bits)
__global__ void k_dist2_keys(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, int n, float qx, float qy,
                             float qz, int include_match, unsigned* __restrict__ key, int* __restrict__ idx) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if(i >= n) return;
    const float px = x[i], py = y[i], pz = z[i];
    const float dx = px - qx, dy = py - qy, dz = pz - qz;
    float s2 = dx * dx + dy * dy;
    s2 = s2 + dz * dz;
    const bool skip = !include_match && px == qx && py == qy && pz == qz;   // kdtree.cpp:265-270
    key[i] = skip ? 0xffffffffu : __float_as_uint(s2);
    idx[i] = i;
}
}   // namespace

// KDTree::get_closest_neighbours (kdtree.cpp:82-103) for one location: the `num` nearest points by float32 squared chord
// distance, nearest first, ties -> lower index (the R-tree's order is unspecified): keys on the device, stable radix sort.
extern "C" int gpp_points_get_closest_neighbours(gpp_points* p, float lat, float lon, int num, int include_match, int* indices, int* count) {
    GPP_TRY
    if(!p || !count) invalid("NULL argument");
    *count = 0;
    if(num <= 0 || p->n == 0) return GPP_OK;
    ensure_device();
    float qx, qy, qz;
    if(gpp_convert_coordinates(&lat, &lon, 1, p->type, &qx, &qy, &qz) != GPP_OK) return GPP_EINVAL;
    p->to_device();
    const int n = p->n;
    DevBuf<unsigned> key, skey;
    DevBuf<int> idx, sidx;
    key.get(n); skey.get(n); idx.get(n); sidx.get(n);
    hipLaunchKernelGGL(k_dist2_keys, dim3((n + 255) / 256), dim3(256), 0, stream(), p->d_x.p, p->d_y.p, p->d_z.p, n, qx, qy, qz, include_match, key.p, idx.p);
    GPP_HIP(hipGetLastError());
    size_t sb = 0;
    GPP_HIP(rocprim::radix_sort_pairs((void*)nullptr, sb, key.p, skey.p, idx.p, sidx.p, (size_t)n, 0u, 32u, stream()));
    DevBuf<char> tmp;
    tmp.get(sb);
    GPP_HIP(rocprim::radix_sort_pairs((void*)tmp.p, sb, key.p, skey.p, idx.p, sidx.p, (size_t)n, 0u, 32u, stream()));
    const int k = std::min(num, n);
    std::vector<unsigned> hk(k);
    std::vector<int> hi(k);
    GPP_HIP(hipMemcpyAsync(hk.data(), skey.p, sizeof(unsigned) * k, hipMemcpyDeviceToHost, stream()));
    GPP_HIP(hipMemcpyAsync(hi.data(), sidx.p, sizeof(int) * k, hipMemcpyDeviceToHost, stream()));
    GPP_HIP(hipStreamSynchronize(stream()));
    int c = 0;
    for(int i = 0; i < k && hk[i] != 0xffffffffu; i++) indices[c++] = hi[i];
    *count = c;
    return GPP_OK;
    GPP_CATCH
}

// ---- distance (src/api/distance.cpp:6-120) -----------------------------------------------------------------------
namespace {
// KDTree::calc_distance (kdtree.cpp:107-133): planar distance, or the great-circle arc from double trigonometry on
// float32-rounded radians (deg2rad returns float, :195-197)
__device__ float d_calc_distance(float lat1, float lon1, float lat2, float lon2, int type) {
    if(type == GPP_CARTESIAN) {
        const float dx = lon1 - lon2, dy = lat1 - lat2;
        return sqrtf(dx * dx + dy * dy);
    }
    if(lat1 == lat2 && lon1 == lon2) return 0;
    const double lat1r = (float)(lat1 * M_PI / 180), lat2r = (float)(lat2 * M_PI / 180);
    const double lon1r = (float)(lon1 * M_PI / 180), lon2r = (float)(lon2 * M_PI / 180);
    const double ratio = cos(lat1r) * cos(lon1r) * cos(lat2r) * cos(lon2r) + cos(lat1r) * sin(lon1r) * cos(lat2r) * sin(lon2r) + sin(lat1r) * sin(lat2r);
    return (float)(acos(ratio) * 6.378137e6);
}

// The `num` nearest points of every location (float32 squared chord, ties -> lower index) by a ring search over the bin index
// with a sorted list of the best `num` candidates kept in HBM scratch, then the largest calc_distance among them.
__global__ __launch_bounds__(256) void k_knn_max_distance(IxView ix, const float* __restrict__ plat, const float* __restrict__ plon, int n,
                                                          const float* __restrict__ qx, const float* __restrict__ qy, const float* __restrict__ qz,
                                                          const float* __restrict__ qlat, const float* __restrict__ qlon, int nq, int num, int type,
                                                          int query_first, float* __restrict__ skey, int* __restrict__ sidx, float* __restrict__ out) {
    const int q = blockIdx.x * blockDim.x + threadIdx.x;
    if(q >= nq) return;
    const int k = min(num, n);
    float* key = skey + (size_t)q * k;
    int* idx = sidx + (size_t)q * k;
    const float x = qx[q], y = qy[q], z = qz[q];
    const float qa = ix.axis_a == 0 ? x : (ix.axis_a == 1 ? y : z), qb = ix.axis_b == 1 ? y : (ix.axis_b == 2 ? z : x);
    const int cbx = bin_of(qa, ix.amin, ix.inv_s, ix.nbx), cby = bin_of(qb, ix.bmin, ix.inv_s, ix.nby);
    const float sbin = 1.0f / ix.inv_s;
    int have = 0;
    auto visit = [&](int row, int xa, int xb) {
        if(row < 0 || row >= ix.nby) return;
        xa = max(xa, 0); xb = min(xb, ix.nbx - 1);
        if(xa > xb) return;
        const int js = ix.bin_start[row * ix.nbx + xa], je = ix.bin_start[row * ix.nbx + xb + 1];
        for(int j = js; j < je; ++j) {
            const float4 g = ix.sgeo[j];
            const float dx = g.x - x, dy = g.y - y, dz = g.z - z;
            float s2 = dx * dx + dy * dy;
            s2 = s2 + dz * dz;
            const int o = __float_as_int(ix.smeta[j].y);
            if(have == k && !(s2 < key[k - 1] || (s2 == key[k - 1] && o < idx[k - 1]))) continue;
            int p = have < k ? have : k - 1;          // insertion into the sorted list
            while(p > 0 && (s2 < key[p - 1] || (s2 == key[p - 1] && o < idx[p - 1]))) { key[p] = key[p - 1]; idx[p] = idx[p - 1]; --p; }
            key[p] = s2; idx[p] = o;
            if(have < k) ++have;
        }
    };
    const int rmax = max(max(cbx, ix.nbx - 1 - cbx), max(cby, ix.nby - 1 - cby));
    for(int r = 0; r <= rmax && k > 0; ++r) {
        if(r >= 2 && have == k) { const float lb = (float)(r - 1) * sbin * 0.999f; if(key[k - 1] < lb * lb) break; }
        if(r == 0) visit(cby, cbx, cbx);
        else {
            visit(cby - r, cbx - r, cbx + r);
            visit(cby + r, cbx - r, cbx + r);
            for(int row = cby - r + 1; row <= cby + r - 1; ++row) { visit(row, cbx - r, cbx - r); visit(row, cbx + r, cbx + r); }
        }
    }
    float max_dist = 0;
    for(int i = 0; i < have; ++i) {
        const int o = idx[i];
        const float d = query_first ? d_calc_distance(qlat[q], qlon[q], plat[o], plon[o], type) : d_calc_distance(plat[o], plon[o], qlat[q], qlon[q], type);
        if(d > max_dist) max_dist = d;
    }
    out[q] = max_dist;
}
}   // namespace

extern "C" int gpp_distance(gpp_points* from, gpp_points* to, int num, int query_first, float* out, int mem) {
    GPP_TRY
    if(!from || !to) invalid("points is NULL");
    if(from->type != to->type) invalid("Incompatible coordinate types");   // distance.cpp:7-8
    const int nq = to->n;
    if(nq == 0) return GPP_OK;
    ensure_device();
    OutField o;
    o.bind(out, nq, mem);
    const int k = std::max(0, std::min(num, from->n));
    if(k == 0) hipLaunchKernelGGL(k_fill_value, dim3((nq + 255) / 256), dim3(256), 0, stream(), o.d, (size_t)nq, 0.0f);
    else {
        to->to_device(); to->latlon_to_device();
        from->latlon_to_device();
        gpp_obs_index* ix = gpp_build_obs_index(from);
        DevBuf<float> skey;
        DevBuf<int> sidx;
        // scratch of k entries per location, in slabs of at most 2^28 entries
        const int slab = (int)std::max<long long>(1, std::min<long long>(nq, (1ll << 28) / k));
        skey.get((size_t)slab * k); sidx.get((size_t)slab * k);
        for(int q0 = 0; q0 < nq; q0 += slab) {
            const int m = std::min(slab, nq - q0);
            hipLaunchKernelGGL(k_knn_max_distance, dim3((m + 255) / 256), dim3(256), 0, stream(), view_of(ix), from->d_lat.p, from->d_lon.p, from->n,
                               to->d_x.p + q0, to->d_y.p + q0, to->d_z.p + q0, to->d_lat.p + q0, to->d_lon.p + q0, m, num, from->type, query_first,
                               skey.p, sidx.p, o.d + q0);
        }
    }
    GPP_HIP(hipGetLastError());
    o.finish();
    GPP_HIP(hipStreamSynchronize(stream()));
    return GPP_OK;
    GPP_CATCH
}

// ---- staticcorr_points (src/api/corr_points.cpp:26-131) -----------------------------------------------------------------
namespace {
// the knots a point keeps: inside its localization radius with corr_background > 0 -- counted (keys == NULL) or stored as
// sortable keys (rho bits, ~knot index) at the point's CSR offset
__global__ __launch_bounds__(256) void k_staticcorr_candidates(IxView ix, DevStructure st, const float* __restrict__ px, const float* __restrict__ py,
                                                               const float* __restrict__ pz, const float* __restrict__ pe, const float* __restrict__ pl,
                                                               int np, int* __restrict__ cnt, const long long* __restrict__ offset,
                                                               unsigned long long* __restrict__ keys) {
    const int y = blockIdx.x * blockDim.x + threadIdx.x;
    if(y >= np) return;
    const float x1 = px[y], y1 = py[y], z1 = pz[y], e1 = pe[y], l1 = pl[y];
    int c = 0;
    long long w = keys ? offset[y] : 0;
    visit_radius(ix, x1, y1, z1, st.R, true, [&](int j, int orig, float) {
        const float4 g = ix.sgeo[j];
        const float rho = d_corr(st, x1, y1, z1, e1, l1, g.x, g.y, g.z, g.w, ix.smeta[j].x, true);   // corr_background(p1, p2)
        if(!(rho > 0.0f)) return;
        if(keys) keys[w++] = ((unsigned long long)__float_as_uint(rho) << 32) | (unsigned)(~orig);
        ++c;
    });
    if(!keys) cnt[y] = c;
}
// out[y][knot] = rho for the kept knots: all of them, or the max_points largest keys (rho descending, ties -> lower index)
__global__ __launch_bounds__(256) void k_staticcorr_write(const unsigned long long* __restrict__ keys, const long long* __restrict__ offset,
                                                          const int* __restrict__ cnt, int np, int nS, int max_points, float* __restrict__ out) {
    const int y = blockIdx.x * blockDim.x + threadIdx.x;
    if(y >= np) return;
    const unsigned long long* k = keys + offset[y];
    const int n = cnt[y];
    unsigned long long thr = 0ull;   // keep keys >= thr
    if(max_points > 0 && n > max_points) {   // the max_points-th largest key, by bisection on the key value
        unsigned long long lo = 0ull, hi = ~0ull;   // largest t with #(keys >= t) >= max_points
        while(lo < hi) {
            const unsigned long long mid = lo + ((hi - lo) >> 1) + 1ull;
            int c = 0;
            for(int i = 0; i < n; ++i) c += (k[i] >= mid) ? 1 : 0;
            if(c >= max_points) lo = mid; else hi = mid - 1ull;
        }
        thr = lo;
    }
    float* row = out + (size_t)y * nS;
    for(int i = 0; i < n; ++i) {
        const unsigned long long key = k[i];
        if(key >= thr) row[~(unsigned)(key & 0xffffffffull)] = __uint_as_float((unsigned)(key >> 32));
    }
}
}   // namespace

extern "C" int gpp_staticcorr_points(gpp_points* points, gpp_points* knots, const gpp_structure* st, int max_points, float* out, int mem) {
    GPP_TRY
    if(max_points < 0) invalid("max_points must be >= 0");                                  // corr_points.cpp:34-35
    if(!points || !knots || !st) invalid("NULL argument");
    if(points->type != knots->type)
        invalid("Both background grid and observations points must be of same coordinate type (lat/lon or x/y)");   // :37-39
    const int nY = points->n, nS = knots->n;
    const size_t total = (size_t)nY * nS;
    if(total == 0) return GPP_OK;
    if(!out) invalid("out is NULL");
    ensure_device();
    DevStructure d = gpp_resolve_structure(st);
    if(d.fh) runtime("staticcorr_points: spatially varying structure functions are not supported on the GPU path yet");
    OutField o;
    o.bind(out, total, mem);
    GPP_HIP(hipMemsetAsync(o.d, 0, total * sizeof(float), stream()));                       // init_vec2(nY, nS, 0) (:44)
    points->to_device();
    gpp_obs_index* ix = gpp_build_obs_index(knots);
    const IxView iv = view_of(ix);
    DevBuf<int> cnt;
    cnt.get(nY);
    hipLaunchKernelGGL(k_staticcorr_candidates, dim3((nY + 255) / 256), dim3(256), 0, stream(), iv, d, points->d_x.p, points->d_y.p, points->d_z.p,
                       points->d_elev.p, points->d_laf.p, nY, cnt.p, (const long long*)nullptr, (unsigned long long*)nullptr);
    GPP_HIP(hipGetLastError());
    DevBuf<long long> wide, offset;
    const long long tot = scan_counts(cnt.p, nY, wide, offset);
    if(tot > 0) {
        DevBuf<unsigned long long> keys;
        keys.get((size_t)tot);
        hipLaunchKernelGGL(k_staticcorr_candidates, dim3((nY + 255) / 256), dim3(256), 0, stream(), iv, d, points->d_x.p, points->d_y.p, points->d_z.p,
                           points->d_elev.p, points->d_laf.p, nY, cnt.p, offset.p, keys.p);
        hipLaunchKernelGGL(k_staticcorr_write, dim3((nY + 255) / 256), dim3(256), 0, stream(), keys.p, offset.p, cnt.p, nY, nS, max_points, o.d);
        GPP_HIP(hipGetLastError());
        o.finish();
        GPP_HIP(hipStreamSynchronize(stream()));
        return GPP_OK;
    }
    o.finish();
    GPP_HIP(hipStreamSynchronize(stream()));
    return GPP_OK;
    GPP_CATCH
}
