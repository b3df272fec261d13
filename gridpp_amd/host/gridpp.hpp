// gridpp.hpp -- C++ host mirror of the reference's public API (include/gridpp.h) for the hot path only,
// header-only over the C-ABI of libgridpp_hip.so (include/gridpp_hip.h).
//
// Same namespace, type names, signatures, defaults and exception types as the reference, so code written
// against gridpp.h for this path compiles unchanged:
//   vec/vec2/vec3/ivec                         include/gridpp.h:25-41
//   Statistic, CoordinateType, MV              :49,88-100,120-123
//   Points, Grid, KDTree (queries)             :1746-2060
//   BarnesStructure (scalar form)              :2160-2185
//   optimal_interpolation(_full)(_ensi)        :162-294
//   neighbourhood*, get_neighbourhood_thresholds :588-716
//   nearest (all eight overloads)              :860-900
//   bilinear(Grid, Grid|Points, vec2|vec3)     :902-930
//   count, gridding, gridding_nearest          :938-1010
//   fill, fill_missing, doping_square/circle, neighbourhood_search, calc_gradient
//   calc_statistic / calc_quantile             :1454-1482
// Nested vectors are flattened once, handed to the C-ABI as host buffers (GPP_MEM_HOST) and un-flattened,
// exactly where the reference flattens them itself (src/api/oi.cpp:69-86).  Errors: GPP_EINVAL ->
// std::invalid_argument, everything else -> std::runtime_error (swig/gridpp.i:21-40 maps these to python).
#pragma once
#include <cmath>
#include <memory>
#include <stdexcept>
#include <iostream>
#include <string>
#include <vector>
#include "../../include/gridpp_hip.h"

namespace gridpp {
typedef std::vector<float> vec;
typedef std::vector<vec> vec2;
typedef std::vector<vec2> vec3;
typedef std::vector<int> ivec;
typedef std::vector<ivec> ivec2;
static const float MV = NAN;
static const double radius_earth = 6.378137e6;

enum Statistic { Mean = 0, Min = 10, Median = 20, Max = 30, Quantile = 40, Std = 50, Variance = 60, Sum = 70, Count = 80, RandomChoice = 90, Unknown = -1 };
enum CoordinateType { Geodetic = 0, Cartesian = 1 };

namespace detail {
inline void check(int rc) {
    if(rc == GPP_OK) return;
    std::string msg = gpp_last_error();
    if(rc == GPP_EINVAL) throw std::invalid_argument(msg);
    throw std::runtime_error(msg);
}
inline vec flatten(const vec2& a, size_t& Y, size_t& X) {
    Y = a.size(); X = Y ? a[0].size() : 0;
    vec f; f.reserve(Y * X);
    for(const auto& r : a) { if(r.size() != X) throw std::invalid_argument("ragged 2-D array"); f.insert(f.end(), r.begin(), r.end()); }
    return f;
}
inline vec flatten(const vec3& a, size_t& Y, size_t& X, size_t& E) {
    Y = a.size(); X = Y ? a[0].size() : 0; E = (Y && X) ? a[0][0].size() : 0;
    vec f; f.reserve(Y * X * E);
    for(const auto& r : a) { if(r.size() != X) throw std::invalid_argument("ragged 3-D array");
        for(const auto& c : r) { if(c.size() != E) throw std::invalid_argument("ragged 3-D array"); f.insert(f.end(), c.begin(), c.end()); } }
    return f;
}
inline vec2 unflatten(const vec& f, size_t Y, size_t X) {
    vec2 o(Y);
    for(size_t y = 0; y < Y; y++) o[y].assign(f.begin() + y * X, f.begin() + (y + 1) * X);
    return o;
}
inline vec3 unflatten(const vec& f, size_t Y, size_t X, size_t E) {
    vec3 o(Y);
    for(size_t y = 0; y < Y; y++) { o[y].resize(X); for(size_t x = 0; x < X; x++) o[y][x].assign(f.begin() + (y * X + x) * E, f.begin() + (y * X + x + 1) * E); }
    return o;
}
struct Handle {
    gpp_points* p = nullptr;
    ~Handle() { if(p) gpp_points_destroy(p); }
};
}   // namespace detail

inline bool is_valid(float value) { return !std::isnan(value) && !std::isinf(value); }   // src/api/util.cpp:16-18
inline std::string version() { return gpp_version(); }
// include/gridpp.h:1410, src/api/gridpp.cpp:11-43 (the reference's table has no "variance": Unknown, like every other name it does not know)
inline Statistic get_statistic(const std::string& name) {
    static const struct { const char* n; Statistic s; } table[] = {{"mean", Mean}, {"min", Min}, {"max", Max}, {"median", Median}, {"quantile", Quantile},
                                                                   {"std", Std}, {"sum", Sum}, {"count", Count}, {"randomchoice", RandomChoice}};
    for(const auto& e : table) if(name == e.n) return e.s;
    return Unknown;
}
inline void set_omp_threads(int) {}        // src/api/gridpp.cpp:184-207: no meaning on the GPU path
inline int get_omp_threads() { return 1; }
// messages (include/gridpp.h:1394-1430, src/api/gridpp.cpp:70-76, src/api/util.cpp:226-252)
inline int& debug_level_ref() { static int level = 0; return level; }
inline void set_debug_level(int level) { debug_level_ref() = level; }
inline int get_debug_level() { return debug_level_ref(); }
inline void debug(const std::string& s) { std::cout << s << std::endl; }
inline void warning(const std::string& s) { std::cout << "Warning: " << s << std::endl; }
inline void error(const std::string& s) { std::cout << "Error: " << s << std::endl; throw std::runtime_error(s); }
inline void future_deprecation_warning(const std::string& function, const std::string& other = "") {
    std::cout << "Future deprecation warning: " << function << " will be deprecated";
    if(other != "") std::cout << ", use " << other << " instead." << std::endl;
    else std::cout << "." << std::endl;
}

class Points {
  public:
    Points() : Points(vec(), vec()) {}
    Points(vec lats, vec lons, vec elevs = vec(), vec lafs = vec(), CoordinateType type = Geodetic) : mType(type) {
        size_t N = lats.size();
        if(lons.size() != N) throw std::invalid_argument("Cannot create points with unequal lat and lon sizes");
        if(elevs.size() != 0 && elevs.size() != N) throw std::invalid_argument("'elevs' must either be size 0 or the same size at lats/lons");
        if(lafs.size() != 0 && lafs.size() != N) throw std::invalid_argument("'lafs' must either be size 0 or the same size at lats/lons");
        mH = std::make_shared<detail::Handle>();
        detail::check(gpp_points_create(lats.data(), lons.data(), elevs.size() == N && N ? elevs.data() : nullptr,
                                        lafs.size() == N && N ? lafs.data() : nullptr, (int)N, (int)type, &mH->p));
        mN = (int)N;
    }
    int size() const { return mN; }
    CoordinateType get_coordinate_type() const { return mType; }
    vec get_lats() const { return field(0); }
    vec get_lons() const { return field(1); }
    vec get_elevs() const { return field(2); }
    vec get_lafs() const { return field(3); }
    ivec get_neighbours(float lat, float lon, float radius, bool include_match = true) const {
        ivec idx(mN > 0 ? mN : 1); int count = 0;
        detail::check(gpp_points_get_neighbours(mH->p, lat, lon, radius, include_match, idx.data(), nullptr, (int)idx.size(), &count));
        idx.resize(count);
        return idx;
    }
    int get_num_neighbours(float lat, float lon, float radius, bool include_match = true) const { return (int)get_neighbours(lat, lon, radius, include_match).size(); }
    int get_nearest_neighbour(float lat, float lon, bool include_match = true) const {
        int i = -1;
        detail::check(gpp_points_nearest_neighbour(mH->p, &lat, &lon, 1, include_match, &i));
        return i;
    }
    gpp_points* handle() const { return mH->p; }
  protected:
    vec field(int k) const { vec o(mN); detail::check(gpp_points_get(mH->p, k, o.data())); return o; }
    std::shared_ptr<detail::Handle> mH;
    int mN = 0;
    CoordinateType mType = Geodetic;
};
typedef Points KDTree;

class Grid {
  public:
    Grid() : Grid(vec2(), vec2()) {}
    Grid(vec2 lats, vec2 lons, vec2 elevs = vec2(), vec2 lafs = vec2(), CoordinateType type = Geodetic) : mType(type) {
        size_t Y, X, Y2, X2;
        vec la = detail::flatten(lats, Y, X), lo = detail::flatten(lons, Y2, X2);
        if(Y != Y2 || X != X2) throw std::invalid_argument("lats and lons must have the same shape");
        size_t Ye, Xe, Yl, Xl;
        vec el = detail::flatten(elevs, Ye, Xe), lf = detail::flatten(lafs, Yl, Xl);
        if(Y * X == 0) Y = X = 0;
        mH = std::make_shared<detail::Handle>();
        detail::check(gpp_grid_create(la.data(), lo.data(), (Ye == Y && Xe == X && Y * X) ? el.data() : nullptr,   // grid.cpp:41-54
                                      (Yl == Y && Xl == X && Y * X) ? lf.data() : nullptr, (int)Y, (int)X, (int)type, &mH->p));
        mY = (int)Y; mX = (int)X;
    }
    ivec size() const { return ivec{mY, mX}; }
    CoordinateType get_coordinate_type() const { return mType; }
    ivec get_nearest_neighbour(float lat, float lon, bool include_match = true) const {   // grid.cpp:76-82,108-114
        if(mY * mX == 0) return ivec();
        int i = -1;
        detail::check(gpp_points_nearest_neighbour(mH->p, &lat, &lon, 1, include_match, &i));
        return ivec{i / mX, i % mX};
    }
    bool get_box(float lat, float lon, int& Y1_out, int& X1_out, int& Y2_out, int& X2_out) const {   // grid.cpp:149-229
        int inside = 0, box[4] = {-1, -1, -1, -1};
        detail::check(gpp_grid_get_box(mH->p, &lat, &lon, 1, &inside, box));
        Y1_out = box[0]; X1_out = box[1]; Y2_out = box[2]; X2_out = box[3];
        return inside != 0;
    }
    gpp_points* handle() const { return mH->p; }
  private:
    std::shared_ptr<detail::Handle> mH;
    int mY = 0, mX = 0;
    CoordinateType mType = Geodetic;
};

// ---- structure functions (include/gridpp.h:2069-2343), scalar forms -------------------------------------------
class StructureFunction {
  public:
    virtual ~StructureFunction() {}
    const gpp_structure* c_struct() const { return &mS; }
    float localization_distance() const { float d; detail::check(gpp_structure_localization_distance(&mS, 0.0f, 0.0f, &d)); return d; }
  protected:
    StructureFunction() { mS = gpp_structure(); }
    void init(int kind, float h, float v, float w, float hmax) {
        if(!is_valid(v) || v < 0) throw std::invalid_argument("v must be >= 0");
        if(!is_valid(w) || w < 0) throw std::invalid_argument("w must be >= 0");
        mS = gpp_structure();
        mS.kind = kind; mS.h = h; mS.v = v; mS.w = w;
        detail::check(gpp_structure_min_rho(kind, h, hmax, &mS.min_rho));
    }
    gpp_structure mS;
};
class BarnesStructure : public StructureFunction { public: BarnesStructure(float h, float v = 0, float w = 0, float hmax = MV) { init(GPP_SK_BARNES, h, v, w, hmax); } };
class CressmanStructure : public StructureFunction { public: CressmanStructure(float h, float v = 0, float w = 0) { init(GPP_SK_CRESSMAN, h, v, w, MV); } };
class SoarStructure : public StructureFunction { public: SoarStructure(float h, float v = 0, float w = 0, float hmax = MV) { init(GPP_SK_SOAR, h, v, w, hmax); } };
class ToarStructure : public StructureFunction { public: ToarStructure(float h, float v = 0, float w = 0, float hmax = MV) { init(GPP_SK_TOAR, h, v, w, hmax); } };
class PowerlawStructure : public StructureFunction { public: PowerlawStructure(float h, float v = 0, float w = 0, float hmax = MV) { init(GPP_SK_POWERLAW, h, v, w, hmax); } };
class LinearStructure : public StructureFunction { public: LinearStructure(float h, float v = 0, float w = 0, float hmax = MV) { init(GPP_SK_LINEAR, h, v, w, hmax); } };
class MultipleStructure : public StructureFunction {   // src/api/structure.cpp:90-138
  public:
    MultipleStructure(const StructureFunction& sh, const StructureFunction& sv, const StructureFunction& sw) {
        const gpp_structure *h = sh.c_struct(), *v = sv.c_struct(), *w = sw.c_struct();
        mS = *h;
        mS.v = v->v; mS.w = w->w;
        mS.kind_v = (v->kind_v ? v->kind_v - 1 : v->kind) + 1;
        mS.kind_w = (w->kind_w ? w->kind_w - 1 : w->kind) + 1;
        mS.loc = sh.localization_distance();
        mS.flags = GPP_ST_HAS_LOC;
        mS.cv_dist = 0;
        // (a nested MultipleStructure contributes the field of its own vertical / laf component)
        mS.field = h->field; mS.field_v = v->kind_v ? v->field_v : v->field; mS.field_w = w->kind_w ? w->field_w : w->field;
    }
};
class CrossValidation : public StructureFunction {     // src/api/structure.cpp:910-944
  public:
    CrossValidation(const StructureFunction& structure, float dist) {
        if(!is_valid(dist) || dist < 0) throw std::invalid_argument("Invalid 'dist' in CrossValidation structure");
        mS = *structure.c_struct();
        mS.flags |= GPP_ST_CV;
        mS.cv_dist = dist;
    }
};

// ---- optimal interpolation (include/gridpp.h:162-248) -------------------------------------------------
inline vec optimal_interpolation_full(const Points& bpoints, const vec& background, const vec& bvariance, const Points& points,
                                      const vec& obs, const vec& obs_variance, const vec& background_at_points,
                                      const vec& bvariance_at_points, const StructureFunction& structure, int max_points,
                                      vec& analysis_variance, bool allow_extrapolation = true) {
    if(max_points < 0) throw std::invalid_argument("max_points must be >= 0");
    if(bpoints.get_coordinate_type() != points.get_coordinate_type())
        throw std::invalid_argument("Both background points and observations points must be of same coordinate type (lat/lon or x/y)");
    if((int)background.size() != bpoints.size()) throw std::invalid_argument("Input field is not the same size as the grid");
    if(background.size() != bvariance.size()) throw std::invalid_argument("Input bvariance is not the same size as the grid");
    if((int)obs.size() != points.size()) throw std::invalid_argument("Observations and points size mismatch");
    if((int)obs_variance.size() != points.size()) throw std::invalid_argument("Obs variance and points size mismatch");
    if((int)background_at_points.size() != points.size()) throw std::invalid_argument("Background and points size mismatch");
    if((int)bvariance_at_points.size() != points.size()) throw std::invalid_argument("Background variance and points size mismatch");
    if(points.size() == 0) return background;   // oi.cpp:189-190 (analysis_variance left untouched)
    vec out(background.size());
    analysis_variance.assign(background.size(), 0);
    detail::check(gpp_optimal_interpolation_full(bpoints.handle(), background.data(), bvariance.data(), points.handle(), obs.data(),
                                                 obs_variance.data(), background_at_points.data(), bvariance_at_points.data(),
                                                 structure.c_struct(), max_points, allow_extrapolation, out.data(),
                                                 analysis_variance.data(), GPP_MEM_HOST));
    return out;
}
inline vec optimal_interpolation(const Points& bpoints, const vec& background, const Points& points, const vec& pobs, const vec& pratios,
                                 const vec& pbackground, const StructureFunction& structure, int max_points, bool allow_extrapolation = true) {
    if(max_points < 0) throw std::invalid_argument("max_points must be >= 0");
    if(bpoints.get_coordinate_type() != points.get_coordinate_type())
        throw std::invalid_argument("Both background points and observations points must be of same coordinate type (lat/lon or x/y)");
    if((int)background.size() != bpoints.size()) throw std::invalid_argument("Input field is not the same size as the grid");
    if((int)pobs.size() != points.size()) throw std::invalid_argument("Observations and points size mismatch");
    if((int)pratios.size() != points.size()) throw std::invalid_argument("Ratios and points size mismatch");
    if((int)pbackground.size() != points.size()) throw std::invalid_argument("Background and points size mismatch");
    vec out(background.size());
    if(background.empty()) return out;
    detail::check(gpp_optimal_interpolation_full(bpoints.handle(), background.data(), nullptr, points.handle(), pobs.data(), pratios.data(),
                                                 pbackground.data(), nullptr, structure.c_struct(), max_points, allow_extrapolation,
                                                 out.data(), nullptr, GPP_MEM_HOST));
    return out;
}
namespace detail {
inline void check_grid_field(const Grid& g, size_t Y, size_t X) {
    if((int)Y != g.size()[0] || (int)X != g.size()[1]) throw std::invalid_argument("input field is not the same size as the grid");
}
}
inline vec2 optimal_interpolation(const Grid& bgrid, const vec2& background, const Points& points, const vec& pobs, const vec& pratios,
                                  const vec& pbackground, const StructureFunction& structure, int max_points, bool allow_extrapolation = true) {
    if(max_points < 0) throw std::invalid_argument("max_points must be >= 0");
    if(bgrid.get_coordinate_type() != points.get_coordinate_type())
        throw std::invalid_argument("Both background grid and observations points must be of same coordinate type (lat/lon or x/y)");
    size_t Y, X;
    vec bg = detail::flatten(background, Y, X);
    detail::check_grid_field(bgrid, Y, X);
    if((int)pobs.size() != points.size() || (int)pratios.size() != points.size() || (int)pbackground.size() != points.size())
        throw std::invalid_argument("Observations / ratios / background and points size mismatch");
    vec out(bg.size());
    if(!bg.empty())
        detail::check(gpp_optimal_interpolation_full(bgrid.handle(), bg.data(), nullptr, points.handle(), pobs.data(), pratios.data(),
                                                     pbackground.data(), nullptr, structure.c_struct(), max_points, allow_extrapolation,
                                                     out.data(), nullptr, GPP_MEM_HOST));
    return detail::unflatten(out, Y, X);
}
inline vec2 optimal_interpolation_full(const Grid& bgrid, const vec2& background, const vec2& bvariance, const Points& points, const vec& obs,
                                       const vec& obs_variance, const vec& background_at_points, const vec& bvariance_at_points,
                                       const StructureFunction& structure, int max_points, vec2& analysis_variance, bool allow_extrapolation = true) {
    if(max_points < 0) throw std::invalid_argument("max_points must be >= 0");
    if(bgrid.get_coordinate_type() != points.get_coordinate_type())
        throw std::invalid_argument("Both background grid and observations points must be of same coordinate type (lat/lon or x/y)");
    size_t Y, X, Yv, Xv;
    vec bg = detail::flatten(background, Y, X), bv = detail::flatten(bvariance, Yv, Xv);
    detail::check_grid_field(bgrid, Y, X);
    if(Y != Yv || X != Xv) throw std::invalid_argument("Input bvariance is not the same size as the grid");
    if((int)obs.size() != points.size() || (int)obs_variance.size() != points.size() || (int)background_at_points.size() != points.size() ||
       (int)bvariance_at_points.size() != points.size())
        throw std::invalid_argument("Observation arrays and points size mismatch");
    if(points.size() == 0) { analysis_variance = bvariance; return background; }
    vec out(bg.size()), var(bg.size());
    detail::check(gpp_optimal_interpolation_full(bgrid.handle(), bg.data(), bv.data(), points.handle(), obs.data(), obs_variance.data(),
                                                 background_at_points.data(), bvariance_at_points.data(), structure.c_struct(), max_points,
                                                 allow_extrapolation, out.data(), var.data(), GPP_MEM_HOST));
    analysis_variance = detail::unflatten(var, Y, X);
    return detail::unflatten(out, Y, X);
}
// include/gridpp.h:263-294
namespace detail {
// the two warnings the reference prints at the end of optimal_interpolation_ensi (src/api/oi_ensi.cpp:557-566)
inline void ensi_warnings() {
    gpp_ensi_stats st;
    if(gpp_ensi_last_stats(&st) != GPP_OK) return;
    if(st.condition_passthrough > 0) warning("Condition number error in " + std::to_string(st.condition_passthrough) + " points. Using raw values in those points.");
    if(st.real_part_passthrough > 0) warning("Could not find the real part of W in " + std::to_string(st.real_part_passthrough) + " points. Using raw values in those points.");
}
}
inline vec2 optimal_interpolation_ensi(const Points& bpoints, const vec2& background, const Points& points, const vec& pobs, const vec& psigmas,
                                       const vec2& pbackground, const StructureFunction& structure, int max_points, bool allow_extrapolation = true) {
    if(max_points < 0) throw std::invalid_argument("max_points must be >= 0");
    if(points.size() == 0) return background;
    if(bpoints.get_coordinate_type() != points.get_coordinate_type())
        throw std::invalid_argument("Both background and observations points must be of same coorindate type (lat/lon or x/y)");
    size_t N, E, S, E2;
    vec bg = detail::flatten(background, N, E), pbg = detail::flatten(pbackground, S, E2);
    if((int)N != bpoints.size()) throw std::invalid_argument("Input field is not the same size as the grid");
    if((int)pobs.size() != points.size() || (int)psigmas.size() != points.size() || (int)S != points.size())
        throw std::invalid_argument("Observations / sigmas / background and points size mismatch");
    if(E != E2) throw std::invalid_argument("Ensemble size mismatch");
    vec out(bg.size());
    detail::check(gpp_optimal_interpolation_ensi(bpoints.handle(), bg.data(), (int)E, points.handle(), pobs.data(), psigmas.data(), pbg.data(),
                                                 structure.c_struct(), max_points, allow_extrapolation, out.data(), GPP_MEM_HOST));
    detail::ensi_warnings();
    return detail::unflatten(out, N, E);
}
inline vec3 optimal_interpolation_ensi(const Grid& bgrid, const vec3& background, const Points& points, const vec& pobs, const vec& psigmas,
                                       const vec2& pbackground, const StructureFunction& structure, int max_points, bool allow_extrapolation = true) {
    if(max_points < 0) throw std::invalid_argument("max_points must be >= 0");
    if(points.size() == 0) return background;
    if(bgrid.size()[0] == 0 || bgrid.size()[1] == 0) throw std::invalid_argument("Grid size cannot be zero");
    if(bgrid.get_coordinate_type() != points.get_coordinate_type())
        throw std::invalid_argument("Both background grid and observations points must be of same coordinate type (lat/lon or x/y)");
    size_t Y, X, E, S, E2;
    vec bg = detail::flatten(background, Y, X, E), pbg = detail::flatten(pbackground, S, E2);
    detail::check_grid_field(bgrid, Y, X);
    if((int)pobs.size() != points.size() || (int)psigmas.size() != points.size() || (int)S != points.size())
        throw std::invalid_argument("Observations / sigmas / background and points size mismatch");
    if(E != E2) throw std::invalid_argument("Ensemble members in gridded background is not the same as in the point background");
    vec out(bg.size());
    detail::check(gpp_optimal_interpolation_ensi(bgrid.handle(), bg.data(), (int)E, points.handle(), pobs.data(), psigmas.data(), pbg.data(),
                                                 structure.c_struct(), max_points, allow_extrapolation, out.data(), GPP_MEM_HOST));
    detail::ensi_warnings();
    return detail::unflatten(out, Y, X, E);
}

// include/gridpp.h:311-441 (optimal_interpolation_ensi_multi_ebe / _ebesc / _utem; variant 1 / 2 / 3 of the C-ABI entry point)
namespace detail {
inline vec2 ensi_multi_points(int variant, gpp_points* bhandle, int bsize, CoordinateType btype, const vec& bratios, const vec2& background, const vec2* background_corr,
                              const Points& points, const vec* pobs1, const vec2* pobs2, const vec& pratios, const vec2& pbackground,
                              const vec2* pbackground_corr, const StructureFunction& structure, int max_points, bool allow_extrapolation) {
    if(max_points < 0) throw std::invalid_argument("max_points must be >= 0");
    if(btype != points.get_coordinate_type())
        throw std::invalid_argument("Both background and observations points must be of same coorindate type (lat/lon or x/y)");
    if((int)background.size() != bsize) throw std::invalid_argument("Input background field is not the same size as the grid");
    if(background_corr && (int)background_corr->size() != bsize) throw std::invalid_argument("Input background_corr field is not the same size as the grid");
    if((int)bratios.size() != bsize) throw std::invalid_argument("Bratios and grid size mismatch");
    if((int)(pobs1 ? pobs1->size() : pobs2->size()) != points.size()) throw std::invalid_argument("Observations and points exception mismatch");
    if((int)pratios.size() != points.size()) throw std::invalid_argument("Pratios and points size mismatch");
    if((int)pbackground.size() != points.size()) throw std::invalid_argument("Background and points size mismatch");
    if(pbackground_corr && (int)pbackground_corr->size() != points.size()) throw std::invalid_argument("Background_corr and points size mismatch");
    if(points.size() == 0) return background;
    size_t N, E, S, E2;
    vec bg = flatten(background, N, E), pbg = flatten(pbackground, S, E2), bgc, pbgc, po;
    if(background_corr) { size_t a, b; bgc = flatten(*background_corr, a, b); pbgc = flatten(*pbackground_corr, a, b); }
    if(pobs2) { size_t a, b; po = flatten(*pobs2, a, b); } else po = *pobs1;
    if(E != E2) throw std::invalid_argument("Ensemble size mismatch");
    vec out(bg.size());
    check(gpp_optimal_interpolation_ensi_multi(variant, bhandle, bratios.data(), bg.data(), background_corr ? bgc.data() : nullptr, (int)E,
                                               points.handle(), po.data(), pratios.data(), pbg.data(), pbackground_corr ? pbgc.data() : nullptr,
                                               structure.c_struct(), max_points, allow_extrapolation, out.data(), GPP_MEM_HOST));
    return unflatten(out, N, E);
}
}  // namespace detail
inline vec2 optimal_interpolation_ensi_multi_ebe(const Points& bpoints, const vec& bratios, const vec2& background, const vec2& background_corr,
                                                 const Points& obs_points, const vec2& pobs, const vec& pratios, const vec2& pbackground,
                                                 const vec2& pbackground_corr, const StructureFunction& structure, int max_points, bool allow_extrapolation = true) {
    return detail::ensi_multi_points(1, bpoints.handle(), bpoints.size(), bpoints.get_coordinate_type(), bratios, background, &background_corr, obs_points, nullptr, &pobs, pratios, pbackground, &pbackground_corr,
                                     structure, max_points, allow_extrapolation);
}
inline vec2 optimal_interpolation_ensi_multi_ebesc(const Points& bpoints, const vec& bratios, const vec2& background, const Points& obs_points, const vec2& pobs,
                                                   const vec& pratios, const vec2& pbackground, const StructureFunction& structure, int max_points,
                                                   bool allow_extrapolation = true) {
    return detail::ensi_multi_points(2, bpoints.handle(), bpoints.size(), bpoints.get_coordinate_type(), bratios, background, nullptr, obs_points, nullptr, &pobs, pratios, pbackground, nullptr, structure, max_points,
                                     allow_extrapolation);
}
inline vec2 optimal_interpolation_ensi_multi_utem(const Points& bpoints, const vec& bratios, const vec2& background, const vec2& background_corr,
                                                  const Points& obs_points, const vec& pobs, const vec& pratios, const vec2& pbackground,
                                                  const vec2& pbackground_corr, const StructureFunction& structure, int max_points, bool allow_extrapolation = true) {
    return detail::ensi_multi_points(3, bpoints.handle(), bpoints.size(), bpoints.get_coordinate_type(), bratios, background, &background_corr, obs_points, &pobs, nullptr, pratios, pbackground, &pbackground_corr,
                                     structure, max_points, allow_extrapolation);
}
// Grid overloads (src/api/oi_ensi_multi.cpp:34-327): the grid's points in row-major order, fields flattened the same way
namespace detail {
inline vec2 flat_field(const vec3& f) { vec2 o; for(const auto& row : f) for(const auto& c : row) o.push_back(c); return o; }
inline vec flat_field(const vec2& f) { vec o; for(const auto& row : f) for(float c : row) o.push_back(c); return o; }
inline vec3 unflat_field(const vec2& f, size_t Y, size_t X) { vec3 o(Y, vec2(X)); for(size_t y = 0; y < Y; y++) for(size_t x = 0; x < X; x++) o[y][x] = f[y * X + x]; return o; }
}  // namespace detail
inline vec3 optimal_interpolation_ensi_multi_ebe(const Grid& bgrid, const vec2& bratios, const vec3& background, const vec3& background_corr,
                                                 const Points& obs_points, const vec2& pobs, const vec& pratios, const vec2& pbackground,
                                                 const vec2& pbackground_corr, const StructureFunction& structure, int max_points, bool allow_extrapolation = true) {
    const vec2 bgc = detail::flat_field(background_corr);
    return detail::unflat_field(detail::ensi_multi_points(1, bgrid.handle(), bgrid.size()[0] * bgrid.size()[1], bgrid.get_coordinate_type(), detail::flat_field(bratios),
                                detail::flat_field(background), &bgc, obs_points, nullptr, &pobs, pratios, pbackground, &pbackground_corr, structure, max_points,
                                allow_extrapolation), background.size(), background.empty() ? 0 : background[0].size());
}
inline vec3 optimal_interpolation_ensi_multi_ebesc(const Grid& bgrid, const vec2& bratios, const vec3& background, const Points& obs_points, const vec2& pobs,
                                                   const vec& pratios, const vec2& pbackground, const StructureFunction& structure, int max_points,
                                                   bool allow_extrapolation = true) {
    return detail::unflat_field(detail::ensi_multi_points(2, bgrid.handle(), bgrid.size()[0] * bgrid.size()[1], bgrid.get_coordinate_type(), detail::flat_field(bratios),
                                detail::flat_field(background), nullptr, obs_points, nullptr, &pobs, pratios, pbackground, nullptr, structure, max_points, allow_extrapolation), background.size(), background.empty() ? 0 : background[0].size());
}
inline vec3 optimal_interpolation_ensi_multi_utem(const Grid& bgrid, const vec2& bratios, const vec3& background, const vec3& background_corr,
                                                  const Points& obs_points, const vec& pobs, const vec& pratios, const vec2& pbackground,
                                                  const vec2& pbackground_corr, const StructureFunction& structure, int max_points, bool allow_extrapolation = true) {
    const vec2 bgc = detail::flat_field(background_corr);
    return detail::unflat_field(detail::ensi_multi_points(3, bgrid.handle(), bgrid.size()[0] * bgrid.size()[1], bgrid.get_coordinate_type(), detail::flat_field(bratios),
                                detail::flat_field(background), &bgc, obs_points, &pobs, nullptr, pratios, pbackground, &pbackground_corr, structure, max_points,
                                allow_extrapolation), background.size(), background.empty() ? 0 : background[0].size());
}

// ---- multi-GPU: one process per GPU, contiguous row tiles, observation values broadcast from rank 0 over RCCL --------
// No counterpart in include/gridpp.h (the reference's parallelism is OpenMP).  Usage, on every rank:
//     gridpp::multi::init(rank, world, id);                      // id from multi::unique_id() on rank 0, distributed by the caller
//     multi::row_tile(Y, rank, world, r0, r1);                   // build Grid(lats[r0:r1], lons[r0:r1]) and the full Points
//     tile = multi::optimal_interpolation(tile_grid, tile_background, points, pobs, pratios, pbackground, structure, max_points);
// pobs / pratios / pbackground need valid contents on rank 0 only (the other ranks pass vectors of the right size).
namespace multi {
inline void row_tile(int ny, int rank, int world, int& row0, int& row1) { detail::check(gpp_row_tile(ny, rank, world, &row0, &row1)); }
inline std::string unique_id() { std::string id(128, '\0'); detail::check(gpp_comm_unique_id(&id[0])); return id; }
inline void init(int rank, int world, const std::string& id) {
    if(id.size() != 128) throw std::invalid_argument("multi::init: the id must hold 128 bytes");
    detail::check(gpp_comm_init(rank, world, id.data()));
}
inline void destroy() { detail::check(gpp_comm_destroy()); }
inline void broadcast(vec& values, int root = 0) { detail::check(gpp_comm_broadcast_host(values.data(), values.size() * sizeof(float), root)); }
inline vec2 optimal_interpolation(const Grid& tile_grid, const vec2& tile_background, const Points& points, vec& pobs, vec& pratios, vec& pbackground,
                                  const StructureFunction& structure, int max_points, bool allow_extrapolation = true) {
    if((int)pobs.size() != points.size() || (int)pratios.size() != points.size() || (int)pbackground.size() != points.size())
        throw std::invalid_argument("Observations / ratios / background and points size mismatch");
    vec block;                                   // one broadcast for the three vectors
    block.reserve(3 * pobs.size());
    block.insert(block.end(), pobs.begin(), pobs.end());
    block.insert(block.end(), pratios.begin(), pratios.end());
    block.insert(block.end(), pbackground.begin(), pbackground.end());
    broadcast(block, 0);
    const size_t S = pobs.size();
    std::copy(block.begin(), block.begin() + S, pobs.begin());
    std::copy(block.begin() + S, block.begin() + 2 * S, pratios.begin());
    std::copy(block.begin() + 2 * S, block.end(), pbackground.begin());
    return gridpp::optimal_interpolation(tile_grid, tile_background, points, pobs, pratios, pbackground, structure, max_points, allow_extrapolation);
}
}  // namespace multi

// ---- neighbourhood (include/gridpp.h:588-716) ---------------------------------------------------------------
namespace detail {
inline vec2 nb(const vec& f, size_t Y, size_t X, size_t E, int is3d, int halfwidth, Statistic statistic) {
    if(halfwidth < 0) throw std::invalid_argument("Half width must be > 0");
    if(statistic == Quantile) throw std::invalid_argument("Use neighbourhood_quantile for computing neighbourhood quantiles");
    if(Y * X * E == 0) return vec2();
    vec out(Y * X);
    check(gpp_neighbourhood(f.data(), (int)Y, (int)X, (int)E, is3d, halfwidth, (int)statistic, out.data(), GPP_MEM_HOST));
    return unflatten(out, Y, X);
}
inline vec2 brute(const vec& f, size_t Y, size_t X, size_t E, int halfwidth, int statistic, float q) {
    if(halfwidth < 0) throw std::invalid_argument("Half width must be > 0");
    if(Y * X * E == 0) return vec2();
    vec out(Y * X);
    check(gpp_neighbourhood_brute_force(f.data(), (int)Y, (int)X, (int)E, halfwidth, statistic, q, out.data(), GPP_MEM_HOST));
    return unflatten(out, Y, X);
}
inline vec2 qfast(const vec& f, size_t Y, size_t X, size_t E, int is3d, const vec2& quantile, int halfwidth, const vec& thresholds) {
    if(halfwidth < 0) throw std::invalid_argument("Half width must be > 0");
    if(Y * X * E == 0) return vec2();
    size_t Yq, Xq;
    vec q = flatten(quantile, Yq, Xq);
    if(!(Yq == 1 && Xq == 1) && !(Yq == Y && Xq == X)) throw std::invalid_argument("Quantile must be the same size as input, or size (1, 1)");
    vec out(Y * X);
    check(gpp_neighbourhood_quantile_fast(f.data(), (int)Y, (int)X, (int)E, is3d, q.data(), (int)q.size(), halfwidth, thresholds.data(),
                                          (int)thresholds.size(), out.data(), GPP_MEM_HOST));
    return unflatten(out, Y, X);
}
}
inline vec2 neighbourhood(const vec2& input, int halfwidth, Statistic statistic) { size_t Y, X; vec f = detail::flatten(input, Y, X); return detail::nb(f, Y, X, 1, 0, halfwidth, statistic); }
inline vec2 neighbourhood(const vec3& input, int halfwidth, Statistic statistic) { size_t Y, X, E; vec f = detail::flatten(input, Y, X, E); return detail::nb(f, Y, X, E, 1, halfwidth, statistic); }
inline vec2 neighbourhood_brute_force(const vec2& input, int halfwidth, Statistic statistic) { size_t Y, X; vec f = detail::flatten(input, Y, X); return detail::brute(f, Y, X, 1, halfwidth, statistic, 0); }
inline vec2 neighbourhood_brute_force(const vec3& input, int halfwidth, Statistic statistic) { size_t Y, X, E; vec f = detail::flatten(input, Y, X, E); return detail::brute(f, Y, X, E, halfwidth, statistic, 0); }
inline vec2 neighbourhood_quantile(const vec2& input, float quantile, int halfwidth) { size_t Y, X; vec f = detail::flatten(input, Y, X); return detail::brute(f, Y, X, 1, halfwidth, Quantile, quantile); }
inline vec2 neighbourhood_quantile(const vec3& input, float quantile, int halfwidth) { size_t Y, X, E; vec f = detail::flatten(input, Y, X, E); return detail::brute(f, Y, X, E, halfwidth, Quantile, quantile); }
inline vec2 neighbourhood_quantile_fast(const vec2& input, const vec2& quantile, int halfwidth, const vec& thresholds) { size_t Y, X; vec f = detail::flatten(input, Y, X); return detail::qfast(f, Y, X, 1, 0, quantile, halfwidth, thresholds); }
inline vec2 neighbourhood_quantile_fast(const vec2& input, float quantile, int halfwidth, const vec& thresholds) { return neighbourhood_quantile_fast(input, vec2(1, vec(1, quantile)), halfwidth, thresholds); }
inline vec2 neighbourhood_quantile_fast(const vec3& input, const vec2& quantile, int halfwidth, const vec& thresholds) { size_t Y, X, E; vec f = detail::flatten(input, Y, X, E); return detail::qfast(f, Y, X, E, 1, quantile, halfwidth, thresholds); }
inline vec2 neighbourhood_quantile_fast(const vec3& input, float quantile, int halfwidth, const vec& thresholds) { return neighbourhood_quantile_fast(input, vec2(1, vec(1, quantile)), halfwidth, thresholds); }
// deprecated aliases (include/gridpp.h:710-716, src/api/neighbourhood.cpp:541-552)
inline vec2 neighbourhood_ens(const vec3& input, int halfwidth, Statistic statistic) { future_deprecation_warning("neighbourhood_ens", "neighbourhood"); return neighbourhood(input, halfwidth, statistic); }
inline vec2 neighbourhood_quantile_ens(const vec3& input, float quantile, int halfwidth) { future_deprecation_warning("neighbourhood_quantile_ens", "neighbourhood_quantile"); return neighbourhood_quantile(input, quantile, halfwidth); }
inline vec2 neighbourhood_quantile_ens_fast(const vec3& input, float quantile, int radius, const vec& thresholds) { future_deprecation_warning("neighbourhood_quantile_ens_fast", "neighbourhood_quantile_fast"); return neighbourhood_quantile_fast(input, quantile, radius, thresholds); }
namespace detail {
inline vec thresholds(const vec& f, int num) {
    if(num <= 0) throw std::invalid_argument("num_thresholds must be > 0");
    if(f.empty()) return vec();
    vec out(num >= (int)f.size() ? f.size() : num); int count = 0;
    check(gpp_calc_even_quantiles(f.data(), (long)f.size(), num, 1, out.data(), &count, GPP_MEM_HOST));
    out.resize(count);
    return out;
}
}
inline vec get_neighbourhood_thresholds(const vec2& input, int num_thresholds) { size_t Y, X; return detail::thresholds(detail::flatten(input, Y, X), num_thresholds); }
inline vec get_neighbourhood_thresholds(const vec3& input, int num_thresholds) { size_t Y, X, E; return detail::thresholds(detail::flatten(input, Y, X, E), num_thresholds); }

// ---- nearest (include/gridpp.h:895; src/api/nearest.cpp:124-144) ------------------------------------------------
namespace detail {
// values [T][n] flattened -> [T][nq]; `sized` = the size check of the overload passed (src/api/util.cpp:427-438)
inline vec nearest_flat(gpp_points* from, size_t nfrom, gpp_points* to, size_t nq, const vec& v, size_t T, bool sized, const char* what) {
    if(!sized) throw std::invalid_argument(what);
    vec out(T * nq, MV);
    if(T * nq == 0 || nfrom == 0) return out;
    if(v.size() != T * nfrom) throw std::invalid_argument(what);
    check(gpp_nearest_levels(from, to, v.data(), (int)T, out.data(), GPP_MEM_HOST));
    return out;
}
inline size_t cells(const Grid& g) { return (size_t)g.size()[0] * g.size()[1]; }
inline bool fits(const Grid& g, size_t Y, size_t X) { return (int)Y == g.size()[0] && (int)X == g.size()[1]; }
const char* const GRID_MISMATCH = "Grid size is not the same as values";
const char* const POINTS_MISMATCH = "Points size is not the same as values";
}   // namespace detail
// the eight overloads of src/api/nearest.cpp:7-222
inline vec nearest(const Grid& igrid, const Points& opoints, const vec2& ivalues) {
    size_t Y, X;
    vec v = detail::flatten(ivalues, Y, X);
    return detail::nearest_flat(igrid.handle(), detail::cells(igrid), opoints.handle(), opoints.size(), v, 1, Y == 0 || detail::fits(igrid, Y, X), detail::GRID_MISMATCH);
}
inline vec2 nearest(const Grid& igrid, const Points& opoints, const vec3& ivalues) {
    size_t T, Y, X;
    vec v = detail::flatten(ivalues, T, Y, X);
    return detail::unflatten(detail::nearest_flat(igrid.handle(), detail::cells(igrid), opoints.handle(), opoints.size(), v, T,
                                                  T == 0 || Y == 0 || detail::fits(igrid, Y, X), detail::GRID_MISMATCH), T, opoints.size());
}
inline vec2 nearest(const Grid& igrid, const Grid& ogrid, const vec2& ivalues) {
    size_t Y, X;
    vec v = detail::flatten(ivalues, Y, X);
    return detail::unflatten(detail::nearest_flat(igrid.handle(), detail::cells(igrid), ogrid.handle(), detail::cells(ogrid), v, 1,
                                                  Y == 0 || detail::fits(igrid, Y, X), detail::GRID_MISMATCH), ogrid.size()[0], ogrid.size()[1]);
}
inline vec3 nearest(const Grid& igrid, const Grid& ogrid, const vec3& ivalues) {
    size_t T, Y, X;
    vec v = detail::flatten(ivalues, T, Y, X);
    return detail::unflatten(detail::nearest_flat(igrid.handle(), detail::cells(igrid), ogrid.handle(), detail::cells(ogrid), v, T,
                                                  T == 0 || Y == 0 || detail::fits(igrid, Y, X), detail::GRID_MISMATCH), T, ogrid.size()[0], ogrid.size()[1]);
}
inline vec nearest(const Points& ipoints, const Points& opoints, const vec& ivalues) {
    return detail::nearest_flat(ipoints.handle(), ipoints.size(), opoints.handle(), opoints.size(), ivalues, 1, (int)ivalues.size() == ipoints.size(), detail::POINTS_MISMATCH);
}
inline vec2 nearest(const Points& ipoints, const Points& opoints, const vec2& ivalues) {
    size_t T, N;
    vec v = detail::flatten(ivalues, T, N);
    return detail::unflatten(detail::nearest_flat(ipoints.handle(), ipoints.size(), opoints.handle(), opoints.size(), v, T,
                                                  T == 0 || (int)N == ipoints.size(), detail::POINTS_MISMATCH), T, opoints.size());
}
inline vec2 nearest(const Points& ipoints, const Grid& ogrid, const vec& ivalues) {
    return detail::unflatten(detail::nearest_flat(ipoints.handle(), ipoints.size(), ogrid.handle(), detail::cells(ogrid), ivalues, 1,
                                                  (int)ivalues.size() == ipoints.size(), detail::POINTS_MISMATCH), ogrid.size()[0], ogrid.size()[1]);
}
inline vec3 nearest(const Points& ipoints, const Grid& ogrid, const vec2& ivalues) {
    size_t T, N;
    vec v = detail::flatten(ivalues, T, N);
    return detail::unflatten(detail::nearest_flat(ipoints.handle(), ipoints.size(), ogrid.handle(), detail::cells(ogrid), v, T,
                                                  T == 0 || (int)N == ipoints.size(), detail::POINTS_MISMATCH), T, ogrid.size()[0], ogrid.size()[1]);
}

// ---- count / gridding (include/gridpp.h:938-1010; src/api/count.cpp:6-66, src/api/gridding.cpp:6-131) -------------
inline vec count(const Points& ipoints, const Points& opoints, float radius) {
    vec out(opoints.size(), 0);
    if(opoints.size()) detail::check(gpp_count(ipoints.handle(), opoints.handle(), radius, out.data(), GPP_MEM_HOST));
    return out;
}
inline vec count(const Grid& igrid, const Points& opoints, float radius) {
    vec out(opoints.size(), 0);
    if(opoints.size()) detail::check(gpp_count(igrid.handle(), opoints.handle(), radius, out.data(), GPP_MEM_HOST));
    return out;
}
inline vec2 count(const Points& ipoints, const Grid& ogrid, float radius) {
    vec out(detail::cells(ogrid), 0);
    if(out.size()) detail::check(gpp_count(ipoints.handle(), ogrid.handle(), radius, out.data(), GPP_MEM_HOST));
    return detail::unflatten(out, ogrid.size()[0], ogrid.size()[1]);
}
inline vec2 count(const Grid& igrid, const Grid& ogrid, float radius) {
    vec out(detail::cells(ogrid), 0);
    if(out.size()) detail::check(gpp_count(igrid.handle(), ogrid.handle(), radius, out.data(), GPP_MEM_HOST));
    return detail::unflatten(out, ogrid.size()[0], ogrid.size()[1]);
}
// distance (include/gridpp.h; src/api/distance.cpp:6-120): largest calc_distance to the `num` nearest points of the input set
inline vec distance(const Grid& grid, const Points& points, int num) {
    if(grid.get_coordinate_type() != points.get_coordinate_type()) throw std::invalid_argument("Incompatible coordinate types");
    vec out(points.size(), 0);
    if(points.size()) detail::check(gpp_distance(grid.handle(), points.handle(), num, 1, out.data(), GPP_MEM_HOST));
    return out;
}
inline vec distance(const Points& ipoints, const Points& opoints, int num) {
    if(ipoints.get_coordinate_type() != opoints.get_coordinate_type()) throw std::invalid_argument("Incompatible coordinate types");
    vec out(opoints.size(), 0);
    if(opoints.size()) detail::check(gpp_distance(ipoints.handle(), opoints.handle(), num, 1, out.data(), GPP_MEM_HOST));
    return out;
}
inline vec2 distance(const Grid& igrid, const Grid& ogrid, int num) {
    if(igrid.get_coordinate_type() != ogrid.get_coordinate_type()) throw std::invalid_argument("Incompatible coordinate types");
    vec out(detail::cells(ogrid), 0);
    if(out.size()) detail::check(gpp_distance(igrid.handle(), ogrid.handle(), num, 0, out.data(), GPP_MEM_HOST));
    return detail::unflatten(out, ogrid.size()[0], ogrid.size()[1]);
}
inline vec2 distance(const Points& points, const Grid& grid, int num) {
    if(points.get_coordinate_type() != grid.get_coordinate_type()) throw std::invalid_argument("Incompatible coordinate types");
    vec out(detail::cells(grid), 0);
    if(out.size()) detail::check(gpp_distance(points.handle(), grid.handle(), num, 0, out.data(), GPP_MEM_HOST));
    return detail::unflatten(out, grid.size()[0], grid.size()[1]);
}
namespace detail {
inline vec gridding_flat(gpp_points* to, size_t nout, const Points& points, const vec& values, float radius, int min_num, Statistic statistic, bool nearest) {
    if((int)values.size() != points.size()) throw std::invalid_argument("Points size is not the same as values");
    if(!nearest && (std::isnan(radius) || std::isinf(radius) || radius < 0)) throw std::invalid_argument("radius must be >= 0");
    if(min_num < 0) throw std::invalid_argument("min_num must be >= 0");
    vec out(nout, MV);
    if(nearest) { if(nout || points.size()) check(gpp_gridding_nearest(to, points.handle(), values.data(), min_num, (int)statistic, out.data(), GPP_MEM_HOST)); }
    else if(nout) check(gpp_gridding(to, points.handle(), values.data(), radius, min_num, (int)statistic, out.data(), GPP_MEM_HOST));
    return out;
}
}   // namespace detail
inline vec2 gridding(const Grid& grid, const Points& points, const vec& values, float radius, int min_num, Statistic statistic) {
    return detail::unflatten(detail::gridding_flat(grid.handle(), detail::cells(grid), points, values, radius, min_num, statistic, false), grid.size()[0], grid.size()[1]);
}
inline vec gridding(const Points& opoints, const Points& ipoints, const vec& values, float radius, int min_num, Statistic statistic) {
    return detail::gridding_flat(opoints.handle(), opoints.size(), ipoints, values, radius, min_num, statistic, false);
}
inline vec2 gridding_nearest(const Grid& grid, const Points& points, const vec& values, int min_num, Statistic statistic) {
    return detail::unflatten(detail::gridding_flat(grid.handle(), detail::cells(grid), points, values, 0, min_num, statistic, true), grid.size()[0], grid.size()[1]);
}
inline vec gridding_nearest(const Points& opoints, const Points& ipoints, const vec& values, int min_num, Statistic statistic) {
    return detail::gridding_flat(opoints.handle(), opoints.size(), ipoints, values, 0, min_num, statistic, true);
}

// ---- fill / doping / neighbourhood_search / calc_gradient (src/api/fill.cpp, doping.cpp, neighbourhood_search.cpp,
//      calc_gradient.cpp) ---------------------------------------------------------------------------------------------
enum GradientType { MinMax = 0, LinearRegression = 10 };   // include/gridpp.h:126-129
inline vec2 fill(const Grid& igrid, const vec2& input, const Points& points, const vec& radii, float value, bool outside) {
    size_t Y, X;
    vec v = detail::flatten(input, Y, X);
    if(Y != 0 && !detail::fits(igrid, Y, X)) throw std::invalid_argument("Grid size is not the same as values");
    if((int)radii.size() != points.size()) throw std::invalid_argument("Points size is not the same as radii size");
    vec out(v.size(), MV);
    if(v.size()) detail::check(gpp_fill(igrid.handle(), v.data(), points.handle(), radii.data(), value, outside, out.data(), GPP_MEM_HOST));
    return detail::unflatten(out, Y, X);
}
inline vec2 fill_missing(const vec2& values) {
    size_t Y, X;
    vec v = detail::flatten(values, Y, X);
    vec out(v.size(), MV);
    if(v.size()) detail::check(gpp_fill_missing(v.data(), (int)Y, (int)X, out.data(), GPP_MEM_HOST));
    return detail::unflatten(out, Y, X);
}
namespace detail {
inline vec2 doping(const Grid& igrid, const vec2& background, const Points& points, const vec& observations, const int* halfwidth, const float* radii,
                   size_t nper, float max_elev_diff) {
    size_t Y, X;
    vec v = flatten(background, Y, X);
    if(Y != 0 && !fits(igrid, Y, X)) throw std::invalid_argument("Grid size is not the same as observations");
    if((int)observations.size() != points.size()) throw std::invalid_argument("Points size is not the same as observations size");
    if((int)nper != points.size()) throw std::invalid_argument(halfwidth ? "Points size is not the same as halfwidth size" : "Points size is not the same as radii size");
    vec out(v.size(), MV);
    if(v.size()) check(gpp_doping(igrid.handle(), v.data(), points.handle(), observations.data(), halfwidth, radii, max_elev_diff, out.data(), GPP_MEM_HOST));
    return unflatten(out, Y, X);
}
}   // namespace detail
inline vec2 doping_square(const Grid& igrid, const vec2& background, const Points& points, const vec& observations, const ivec& halfwidth, float max_elev_diff = MV) {
    return detail::doping(igrid, background, points, observations, halfwidth.data(), nullptr, halfwidth.size(), max_elev_diff);
}
inline vec2 doping_circle(const Grid& igrid, const vec2& background, const Points& points, const vec& observations, const vec& radii, float max_elev_diff = MV) {
    return detail::doping(igrid, background, points, observations, nullptr, radii.data(), radii.size(), max_elev_diff);
}
inline vec2 neighbourhood_search(const vec2& array, const vec2& search_array, int halfwidth, float search_target_min, float search_target_max,
                                 float search_delta, const ivec2& apply_array = ivec2()) {
    size_t Y, X, Ys, Xs;
    vec a = detail::flatten(array, Y, X), s = detail::flatten(search_array, Ys, Xs);
    if(Y != Ys || X != Xs) throw std::invalid_argument("search_array must either be the same size as array");
    ivec ap;
    if(apply_array.size() > 0) {
        if(apply_array.size() > 1 && (apply_array.size() != Y || apply_array[0].size() != X)) throw std::invalid_argument("apply_array must either be empty or same size as array");
        for(const auto& r : apply_array) ap.insert(ap.end(), r.begin(), r.end());
    }
    vec out(a.size(), MV);
    if(a.size()) detail::check(gpp_neighbourhood_search(a.data(), s.data(), (int)Y, (int)X, halfwidth, search_target_min, search_target_max, search_delta,
                                                        ap.size() == a.size() ? ap.data() : nullptr, out.data(), GPP_MEM_HOST));
    return detail::unflatten(out, Y, X);
}
inline vec2 calc_gradient(const vec2& base, const vec2& values, GradientType gradient_type, int halfwidth, int min_num = 2, float min_range = MV,
                          float default_gradient = 0) {
    size_t Y, X, Yv, Xv;
    vec b = detail::flatten(base, Y, X), v = detail::flatten(values, Yv, Xv);
    if(Y == 0) throw std::invalid_argument("base input has no size");
    if(Y != Yv || X != Xv) throw std::invalid_argument("base is not the same size as values");
    vec out(b.size(), MV);
    detail::check(gpp_calc_gradient(b.data(), v.data(), (int)Y, (int)X, (int)gradient_type, halfwidth, min_num, min_range, default_gradient, out.data(), GPP_MEM_HOST));
    return detail::unflatten(out, Y, X);
}

// ---- bilinear (include/gridpp.h:902-930; src/api/bilinear.cpp:26-135) -------------------------------------------
namespace detail {
// values [T][Y][X] flattened; size rule of src/api/util.cpp:427-432
inline vec bilinear_flat(const Grid& igrid, gpp_points* to, size_t nq, const vec& v, size_t T, size_t Y, size_t X, bool empty) {
    if(!empty && ((int)Y != igrid.size()[0] || (int)X != igrid.size()[1])) throw std::invalid_argument("Grid size is not the same as values");
    vec out(T * nq, MV);
    if(T * nq == 0 || igrid.size()[0] == 0 || igrid.size()[1] == 0) return out;   // bilinear.cpp:37-39
    if(empty) throw std::invalid_argument("Grid size is not the same as values");
    check(gpp_bilinear(igrid.handle(), to, v.data(), (int)T, out.data(), GPP_MEM_HOST));
    return out;
}
}   // namespace detail
inline vec bilinear(const Grid& igrid, const Points& opoints, const vec2& ivalues) {
    size_t Y, X;
    vec v = detail::flatten(ivalues, Y, X);
    return detail::bilinear_flat(igrid, opoints.handle(), opoints.size(), v, 1, Y, X, ivalues.size() == 0);
}
inline vec2 bilinear(const Grid& igrid, const Points& opoints, const vec3& ivalues) {
    size_t T, Y, X;
    vec v = detail::flatten(ivalues, T, Y, X);
    vec out = detail::bilinear_flat(igrid, opoints.handle(), opoints.size(), v, T, Y, X, T == 0 || Y == 0);
    return detail::unflatten(out, T, opoints.size());
}
inline vec2 bilinear(const Grid& igrid, const Grid& ogrid, const vec2& ivalues) {
    size_t Y, X;
    vec v = detail::flatten(ivalues, Y, X);
    size_t oy = ogrid.size()[0], ox = ogrid.size()[1];
    return detail::unflatten(detail::bilinear_flat(igrid, ogrid.handle(), oy * ox, v, 1, Y, X, ivalues.size() == 0), oy, ox);
}
inline vec3 bilinear(const Grid& igrid, const Grid& ogrid, const vec3& ivalues) {
    size_t T, Y, X;
    vec v = detail::flatten(ivalues, T, Y, X);
    size_t oy = ogrid.size()[0], ox = ogrid.size()[1];
    return detail::unflatten(detail::bilinear_flat(igrid, ogrid.handle(), oy * ox, v, T, Y, X, T == 0 || Y == 0), T, oy, ox);
}

// ---- util (include/gridpp.h:1454-1482) -----------------------------------------------------------------------------
inline float calc_statistic(const vec& array, Statistic statistic) {
    float out = MV;
    detail::check(gpp_calc_statistic(array.data(), 1, (int)array.size(), (int)statistic, &out, GPP_MEM_HOST));
    return out;
}
inline float calc_quantile(const vec& array, float quantile) {
    float out = MV;
    detail::check(gpp_calc_quantile(array.data(), 1, (int)array.size(), &quantile, 1, &out, GPP_MEM_HOST));
    return out;
}
}   // namespace gridpp
