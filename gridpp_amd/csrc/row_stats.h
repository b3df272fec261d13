// Per-row statistics with the reference's semantics (src/api/util.cpp:19-178): calc_statistic / calc_quantile on one
// contiguous row of floats, sequential in member order like the reference.  Shared by neighbourhood.hip and radius.hip.
#pragma once
#include "common.h"

#pragma clang fp contract(off)

__device__ __forceinline__ bool nv(float v) { return !isnan(v) && !isinf(v); }

// src/api/util.cpp:19-110 on one row held in LDS (stride 1), sequential like the reference
static __device__ float row_statistic(const float* row, int n, int statistic) {
    float value = NAN;
    if(statistic == GPP_MEAN || statistic == GPP_SUM || statistic == GPP_COUNT) {
        float total = 0; int count = 0;
        for(int i = 0; i < n; i++) { float v = row[i]; if(nv(v)) { total += v; count++; } }
        if(statistic == GPP_COUNT) value = (float)count;
        else if(count > 0) value = (statistic == GPP_MEAN) ? total / (float)count : total;
    }
    else if(statistic == GPP_STD || statistic == GPP_VARIANCE) {
        float total = 0, total2 = 0, K = NAN; int count = 0;
        for(int i = 0; i < n; i++) {
            float v = row[i];
            if(nv(v)) {
                if(!nv(K)) K = v;
                float d = v - K;
                total += d; total2 += d * d; count++;
            }
        }
        if(count > 0) {
            float mean = total / (float)count, mean2 = total2 / (float)count;
            float var = mean2 - mean * mean;
            if(var < 0) var = 0;
            value = (statistic == GPP_STD) ? sqrtf(var) : var;
        }
    }
    else if(statistic == GPP_MIN || statistic == GPP_MAX) {   // calc_quantile q = 0 / 1 (util.cpp:121-146)
        float m = NAN;
        for(int i = 0; i < n; i++) {
            float v = row[i];
            if(!nv(v)) continue;
            if(!nv(m)) m = v;
            else if(statistic == GPP_MIN ? v < m : v > m) m = v;
        }
        value = m;
    }
    return value;
}

// monotone map float -> uint32 (valid values only)
__device__ __forceinline__ unsigned f2ord(float f) {
    unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned o) {
    unsigned u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
    return __uint_as_float(u);
}
// util.cpp:158-176 given the two order statistics
__device__ __forceinline__ float quantile_from_order(float q, int N, float lowerValue, float upperValue, int lowerIndex, int upperIndex) {
    if(lowerIndex == upperIndex) return lowerValue;
    float lowerQuantile = (float)lowerIndex / (float)(N - 1);
    float upperQuantile = (float)upperIndex / (float)(N - 1);
    float f = (q - lowerQuantile) / (upperQuantile - lowerQuantile);
    return lowerValue + (upperValue - lowerValue) * f;
}
// k-th smallest (0-based) among the valid values of a lane-private row, by bisection on f2ord
static __device__ float row_kth(const float* row, int n, int k) {
    unsigned lo = 0, hi = 0xffffffffu;   // smallest o with count(ord <= o) >= k+1
    while(lo < hi) {
        unsigned mid = lo + ((hi - lo) >> 1);
        int c = 0;
        for(int i = 0; i < n; i++) { float v = row[i]; if(nv(v) && f2ord(v) <= mid) c++; }
        if(c >= k + 1) hi = mid; else lo = mid + 1;
    }
    return ord2f(lo);
}
static __device__ float row_quantile(const float* row, int n, float q) {   // util.cpp:111-178 (q already validated)
    if(!nv(q)) return NAN;
    int N = 0;
    for(int i = 0; i < n; i++) if(nv(row[i])) N++;
    if(N == 0) return NAN;
    if(q == 0) return row_statistic(row, n, GPP_MIN);
    if(q == 1) return row_statistic(row, n, GPP_MAX);
    float pos = q * (float)(N - 1);
    int lowerIndex = (int)floorf(pos), upperIndex = (int)ceilf(pos);
    float lv = row_kth(row, n, lowerIndex);
    float uv = (upperIndex == lowerIndex) ? lv : row_kth(row, n, upperIndex);
    return quantile_from_order(q, N, lv, uv, lowerIndex, upperIndex);
}

