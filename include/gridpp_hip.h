/*
 * gridpp_hip.h -- C-ABI of the MI355X-native gridpp hot path (libgridpp_hip.so).
 *
 * This is the drop-in boundary: plain pointers and sizes, int status codes, no
 * C++ / torch types.  Each entry point names the reference interface it
 * replaces (paths relative to the metno/gridpp repository root).  The C++ host
 * mirror (gridpp_amd/host/gridpp.hpp) and the Python mirror (gridpp_amd/)
 * both bind exactly these symbols; INTEGRATION.md shows the binding a gridpp
 * maintainer would add.
 *
 * Conventions
 *  - every function returns GPP_OK or a negative error code; the message of the
 *    last error on the calling thread is gpp_last_error().
 *      GPP_EINVAL   -> std::invalid_argument / Python ValueError
 *      GPP_ERUNTIME -> std::runtime_error    / Python RuntimeError
 *  - `mem` says where the caller's field arrays live: GPP_MEM_HOST (the library
 *    stages them through HBM) or GPP_MEM_DEVICE (pointers are HBM addresses of
 *    the current device, e.g. torch tensor .data_ptr(); nothing is copied).
 *    Coordinate arrays given to the *_create functions are always host arrays.
 *  - 2-D fields are row-major [Y][X]; 3-D fields are [Y][X][E] with E fastest
 *    (swig/vector.i:385-390).
 *  - the library keeps its scratch buffers and its stream per process: calls are NOT re-entrant.  One call at a time per
 *    process, which is what the reference's python module does anyway (its calls hold the GIL, swig/python/CMakeLists.txt:5-9);
 *    a multi-threaded C++ host serialises its calls with a mutex.  Handles (gpp_points, gpp_field) are immutable once
 *    created, apart from internal caches that a call fills under that same rule.
 *  - all work is enqueued on the library stream (gpp_get_stream) and the call
 *    returns after the stream is synchronised, unless GPP_MEM_DEVICE |
 *    GPP_ASYNC is given.
 */
#ifndef GRIDPP_HIP_H
#define GRIDPP_HIP_H

#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define GPP_OK 0
#define GPP_EINVAL -1
#define GPP_ERUNTIME -2
#define GPP_ENODEVICE -3

#define GPP_MEM_HOST 0
#define GPP_MEM_DEVICE 1
#define GPP_ASYNC 2
/* With GPP_MEM_DEVICE, gpp_neighbourhood_quantile_fast only: the `quantile` argument is in HOST memory although the fields are in HBM -- the
 * scalar quantile of a script (include/gridpp.h:469-486 takes it by value) beside a device-resident cube. */
#define GPP_Q_HOST 8
/* With GPP_MEM_HOST: the float INPUT fields of the call hold float64 values (numpy's default dtype); they are uploaded as
 * they are and cast to float32 on the device -- the rounding the reference's typemap applies on the host
 * (swig/vector.i:42-55).  Outputs stay float32.  Honoured by gpp_optimal_interpolation_full,
 * gpp_optimal_interpolation_ensi, gpp_neighbourhood, gpp_nearest(_levels), gpp_bilinear (every `const float*` argument of
 * these that follows `mem` is then a `const double*`), by gpp_fill_missing, gpp_neighbourhood_search and gpp_calc_gradient
 * (their float fields), and by gpp_neighbourhood_quantile_fast for `input` only. */
#define GPP_HOST_F64 4

/* include/gridpp.h:120-123 */
#define GPP_GEODETIC 0
#define GPP_CARTESIAN 1

/* include/gridpp.h:88-100 */
#define GPP_MEAN 0
#define GPP_MIN 10
#define GPP_MEDIAN 20
#define GPP_MAX 30
#define GPP_QUANTILE 40
#define GPP_STD 50
#define GPP_VARIANCE 60
#define GPP_SUM 70
#define GPP_COUNT 80
#define GPP_RANDOMCHOICE 90

/* ---- runtime ----------------------------------------------------------- */
const char* gpp_last_error(void);
const char* gpp_version(void);            /* include/gridpp.h:15 GRIDPP_VERSION */
int gpp_device_count(int* count);
/* Path overrides: switches that select between implementations with identical results (the tests reach rarely taken paths with them, A/B
 * timings compare them).  The library reads NO environment variable: an override exists only after gpp_set_path_override(name, value)
 * (name = "GPP_...", value NULL clears it; process-wide).  gpp_active_overrides: the names that are set, comma separated, into buf (NUL
 * terminated); returns how many -- a benchmark must run with none.  No counterpart in the reference. */
int gpp_set_path_override(const char* name, const char* value);
int gpp_active_overrides(char* buf, int len);
/* releases the library's large call-to-call device workspaces (kept otherwise for the next call) and the pool of staging buffers that the
 * host-side fields of a call are copied through (buffers of up to 256 MiB, at most 1 GiB in all) */
int gpp_release_workspaces(void);
int gpp_set_device(int device);           /* one process per GPU: call once with LOCAL_RANK */
int gpp_get_stream(void** hip_stream);    /* the hipStream_t all kernels are launched on */
int gpp_synchronize(void);
/* Page-locked host buffers for results (optional): a caller that hands such a buffer as `out` of a GPP_MEM_HOST call gets
 * the device-to-host copy as one DMA transfer instead of a staged copy into pageable memory.  The python mirror returns its
 * large results in such buffers (the reference returns freshly allocated numpy arrays, swig/vector.i:172-180). */
int gpp_host_alloc(size_t bytes, void** out);
int gpp_host_free(void* p);

/* ---- point sets: gridpp::Points / gridpp::Grid / gridpp::KDTree ----------
 * replaces gridpp::Points::Points (src/api/points.cpp:9-31),
 * gridpp::Grid::Grid (src/api/grid.cpp:12-55) and KDTree::KDTree
 * (src/api/kdtree.cpp:6-16): validates coordinates, converts lat/lon to the
 * float32 x,y,z of src/api/util.cpp:596-612 and keeps them resident in HBM.
 * elevs / lafs may be NULL (filled with NaN).  A grid is a point set with
 * nx > 0 (row-major, index = y*nx + x, src/api/grid.cpp:108-114). */
typedef struct gpp_points gpp_points;
int gpp_points_create(const float* lats, const float* lons, const float* elevs, const float* lafs,
                      int n, int coordinate_type, gpp_points** out);
int gpp_grid_create(const float* lats, const float* lons, const float* elevs, const float* lafs,
                    int ny, int nx, int coordinate_type, gpp_points** out);
/* The same from float64 arrays (numpy's default dtype; the reference's typemap casts them to float32 on the host,
 * swig/vector.i:42-55): the cast happens on the device for large sets. */
int gpp_points_create_f64(const double* lats, const double* lons, const double* elevs, const double* lafs,
                          int n, int coordinate_type, gpp_points** out);
int gpp_grid_create_f64(const double* lats, const double* lons, const double* elevs, const double* lafs,
                        int ny, int nx, int coordinate_type, gpp_points** out);
int gpp_points_destroy(gpp_points* p);
int gpp_points_size(const gpp_points* p, int* n, int* ny, int* nx, int* coordinate_type);
/* field: 0 lat, 1 lon, 2 elev, 3 laf, 4 x, 5 y, 6 z; copies n floats to host `out` */
int gpp_points_get(const gpp_points* p, int field, float* out);

/* gridpp::convert_coordinates (src/api/util.cpp:583-615), host arrays */
int gpp_convert_coordinates(const float* lats, const float* lons, int n, int coordinate_type,
                            float* x, float* y, float* z);

/* KDTree::get_neighbours (src/api/kdtree.cpp:39-60): indices (ascending) of the
 * points strictly inside the +-radius box and within `radius` chord distance of
 * (lat, lon).  Writes at most `cap` indices; *count is the full count. */
int gpp_points_get_neighbours(gpp_points* p, float lat, float lon, float radius, int include_match,
                              int* indices, float* distances /* may be NULL */, int cap, int* count);
/* KDTree::get_closest_neighbours (src/api/kdtree.cpp:82-103): the `num` nearest points, nearest first (ties: lower
 * index); indices must hold `num` ints, *count is the number found. */
int gpp_points_get_closest_neighbours(gpp_points* p, float lat, float lon, int num, int include_match,
                                      int* indices, int* count);
/* KDTree::get_nearest_neighbour / Points::get_nearest_neighbour
 * (src/api/kdtree.cpp:82-106, src/api/points.cpp:55-61) for nq query points
 * (host arrays); index -1 when the set is empty or nothing qualifies. */
int gpp_points_nearest_neighbour(gpp_points* p, const float* qlats, const float* qlons, int nq,
                                 int include_match, int* indices);
/* gridpp::nearest(Grid|Points, Points, values) (src/api/nearest.cpp:124-144):
 * out[i] = values[nearest index of query i]; NaN if the source set is empty.
 * values/out follow `mem`. */
int gpp_nearest(gpp_points* from, gpp_points* to, const float* values, float* out, int mem);
/* The overloads with a leading time dimension (src/api/nearest.cpp:32-71,95-122,145-175,198-222):
 * values [nt][size of from] -> out [nt][size of to]. */
int gpp_nearest_levels(gpp_points* from, gpp_points* to, const float* values, int nt, float* out, int mem);

/* Batched KDTree::get_neighbours / get_neighbours_with_distance / get_num_neighbours (src/api/kdtree.cpp:39-64) for nq
 * lookups (host arrays): counts[nq] always; offsets[nq+1] if given; indices / distances (CSR, ascending index inside a
 * location) if given and *total <= cap -- otherwise only the counts and *total are produced, so that the caller can size
 * the buffers and call again. */
int gpp_points_get_neighbours_batch(gpp_points* p, const float* qlats, const float* qlons, int nq, float radius,
                                    int include_match, int* counts, long long* offsets, int* indices, float* distances,
                                    long long cap, long long* total);
/* gridpp::count (src/api/count.cpp:6-66, all four overloads): out[i] = number of points of `from` within `radius` of
 * location i of `to`.  out follows `mem`. */
int gpp_count(gpp_points* from, gpp_points* to, float radius, float* out, int mem);
/* gridpp::distance (src/api/distance.cpp:6-120, all four overloads): out[i] = the largest KDTree::calc_distance
 * (src/api/kdtree.cpp:107-133) between location i of `to` and its `num` nearest points of `from`.  query_first = 1 for
 * the overloads whose output is a Points (they call calc_distance(location, neighbour)), 0 for those whose output is a
 * Grid (calc_distance(neighbour, location)).  GPP_EINVAL if the coordinate types differ. */
int gpp_distance(gpp_points* from, gpp_points* to, int num, int query_first, float* out, int mem);
/* gridpp::gridding (src/api/gridding.cpp:6-63): statistic of the values of the points of `from` within `radius` of every
 * location of `to`; NaN where fewer than min_num (> 0) points are found.  GPP_EINVAL for radius < 0 / NaN, min_num < 0.
 * values / out follow `mem`. */
int gpp_gridding(gpp_points* to, gpp_points* from, const float* values, float radius, int min_num, int statistic,
                 float* out, int mem);
/* gridpp::gridding_nearest (src/api/gridding.cpp:65-131): every point of `from` is assigned to its nearest location of
 * `to`; statistic of what each location received (in input order), NaN for none / fewer than min_num. */
int gpp_gridding_nearest(gpp_points* to, gpp_points* from, const float* values, int min_num, int statistic, float* out, int mem);

/* gridpp::fill (src/api/fill.cpp:6-41): cells of `igrid` within radii[i] of point i get `value` (outside = 0), or keep
 * `input` while every other cell gets `value` (outside = 1).  radii: host array, one per point.  input/out follow `mem`. */
int gpp_fill(gpp_points* igrid, const float* input, gpp_points* points, const float* radii, float value, int outside,
             float* out, int mem);
/* gridpp::fill_missing (src/api/fill.cpp:43-134): linear interpolation across runs of missing values along rows and
 * along columns, averaged. */
int gpp_fill_missing(const float* values, int ny, int nx, float* out, int mem);
/* gridpp::doping_square (halfwidth != NULL) / doping_circle (radii != NULL) (src/api/doping.cpp:5-93): grid cells in the
 * index window around the nearest grid point of observation i / within radii[i] of it take observations[i]; later
 * observations win; cells whose elevation differs from the observation's by more than max_elev_diff (if valid) are left.
 * halfwidth / radii: host arrays.  background / observations / out follow `mem`. */
int gpp_doping(gpp_points* igrid, const float* background, gpp_points* points, const float* observations,
               const int* halfwidth, const float* radii, float max_elev_diff, float* out, int mem);
/* gridpp::neighbourhood_search (src/api/neighbourhood_search.cpp:7-113); apply_array may be NULL (follows `mem`). */
int gpp_neighbourhood_search(const float* array, const float* search_array, int ny, int nx, int halfwidth,
                             float search_target_min, float search_target_max, float search_delta,
                             const int* apply_array, float* out, int mem);
/* gridpp::calc_gradient (src/api/calc_gradient.cpp:7-126); gradient types as include/gridpp.h:126-129. */
#define GPP_GRADIENT_MINMAX 0
#define GPP_GRADIENT_LINEAR_REGRESSION 10
int gpp_calc_gradient(const float* base, const float* values, int ny, int nx, int gradient_type, int halfwidth, int num_min,
                      float min_range, float default_gradient, float* out, int mem);

/* gridpp::bilinear(Grid, Points|Grid, vec2|vec3) (src/api/bilinear.cpp:26-135; per location :322-403, weights
 * :137-320): `values` holds nt time levels of the input grid, [nt][ny][nx]; out is [nt][size of `to`].  A location
 * outside the grid, or whose box has a missing corner, takes the nearest grid point's value; all NaN if the input
 * grid is empty.  GPP_ERUNTIME ("Problem with bilinear interpolation...") when the weights of a box leave [0, 1]
 * (the reference throws std::runtime_error, :309-313).  values/out follow `mem`. */
int gpp_bilinear(gpp_points* igrid, gpp_points* to, const float* values, int nt, float* out, int mem);
/* Grid::get_box (src/api/grid.cpp:149-229) for nq lookups (host arrays): inside[i] and boxes[4*i..] = Y1, X1, Y2, X2
 * (-1 when no box encloses the point). */
int gpp_grid_get_box(gpp_points* grid, const float* qlats, const float* qlons, int nq, int* inside, int* boxes);
/* gridpp::point_in_rectangle (src/api/util.cpp:571-582): corners A, B, C, D as (lat, lon) pairs. */
int gpp_point_in_rectangle(const float corners_latlon[8], float lat, float lon, int* inside);

/* ---- structure functions (src/api/structure.cpp) -----------------------------
 * Scalar forms of BarnesStructure, CressmanStructure, SoarStructure, ToarStructure,
 * PowerlawStructure, LinearStructure (structure.cpp:143-167,287-299,317-341,467-491,
 * 618-642,765-789), MultipleStructure (:90-138: horizontal / vertical / laf factors taken
 * from three structures), the CrossValidation wrapper (:910-944) and the spatially varying
 * forms (per-grid-point h, v, w looked up at the first point of corr(p1, p2), :168-214). */
#define GPP_SK_BARNES 0
#define GPP_SK_CRESSMAN 1
#define GPP_SK_SOAR 2
#define GPP_SK_TOAR 3
#define GPP_SK_POWERLAW 4
#define GPP_SK_LINEAR 5
#define GPP_ST_HAS_LOC 1   /* `loc` holds the localization distance (else derived from kind, h, min_rho) */
#define GPP_ST_CV 2        /* CrossValidation: corr_background = 0 within cv_dist */
typedef struct gpp_structure {
    int kind;        /* kernel of the horizontal factor (GPP_SK_*) */
    float h, v, w;   /* scales: horizontal [m], vertical [m], land-area-fraction */
    float min_rho;   /* m_min_rho of the horizontal structure; see gpp_structure_min_rho */
    int kind_v;      /* kernel of the vertical factor + 1; 0 = same as `kind` (MultipleStructure sets these) */
    int kind_w;      /* kernel of the laf factor + 1;      0 = same as `kind` */
    float loc;       /* localization distance if flags & GPP_ST_HAS_LOC */
    float cv_dist;   /* CrossValidation distance if flags & GPP_ST_CV */
    int flags;
    struct gpp_field* field;   /* spatially varying form: h, v, w fields (gpp_field_create); NULL = scalar */
    /* MultipleStructure(sh, sv, sw) with spatially varying parts (structure.cpp:90-138): `field` = sh's (its h and the localization
     * distance), field_v = sv's (its v), field_w = sw's (its w); NULL = that factor is scalar (v / w above) */
    struct gpp_field* field_v;
    struct gpp_field* field_w;
} gpp_structure;
/* BarnesStructure(Grid, vec2 h, vec2 v, vec2 w, min_rho) and the Soar/Toar/Powerlaw/Linear equivalents
 * (structure.cpp:168-184,342-358,...): h, v, w are host arrays with one value per point of `grid`
 * (which must outlive the field). */
typedef struct gpp_field gpp_field;
int gpp_field_create(gpp_points* grid, const float* h, const float* v, const float* w, int kind, float min_rho, gpp_field** out);
int gpp_field_destroy(gpp_field* field);
/* m_min_rho of the scalar constructors from hmax (NaN: default 0.0013); validates h >= 0, hmax >= 0 */
int gpp_structure_min_rho(int kind, float h, float hmax, float* min_rho);
/* StructureFunction::localization_distance(Point) (structure.cpp:87-89,271-282,454-459,604-610,755-757,902-904);
 * (lat, lon) only matters for the spatially varying forms */
int gpp_structure_localization_distance(const gpp_structure* s, float lat, float lon, float* dist);
/* corr (background = 0) / corr_background (background = 1) for one pair of points given as
 * (x, y, z, elev, laf, lat, lon) -- runs the same device code as the OI kernels */
int gpp_structure_corr(const gpp_structure* s, const float p1[7], const float p2[7], int background, float* rho);

/* gridpp::staticcorr_points (src/api/corr_points.cpp:26-131): out [size of points][size of knots] = corr_background(point,
 * knot) for the knots a point keeps (inside its localization radius, rho > 0, the max_points largest if there are more; 0
 * elsewhere).  Scalar structure functions.  out follows `mem`. */
int gpp_staticcorr_points(gpp_points* points, gpp_points* knots, const gpp_structure* structure, int max_points, float* out, int mem);

/* ---- optimal interpolation ------------------------------------------------
 * replaces gridpp::optimal_interpolation_full (src/api/oi.cpp:138-341, Points
 * form; the Grid form :342-412 is the same call on a grid handle) and through
 * it gridpp::optimal_interpolation (src/api/oi.cpp:26-136: bvariance == NULL
 * and bvariance_at_points == NULL mean "all ones", out_variance may be NULL).
 * background/bvariance/out/out_variance have bgrid-size elements; obs,
 * obs_variance, background_at_points, bvariance_at_points have points-size
 * elements.  Errors: GPP_EINVAL as oi.cpp:152-186; GPP_ERUNTIME if a local
 * (P+R) matrix is singular (arma::inv throws, oi.cpp:315). */
int gpp_optimal_interpolation_full(gpp_points* bgrid, const float* background, const float* bvariance,
                                   gpp_points* points, const float* obs, const float* obs_variance,
                                   const float* background_at_points, const float* bvariance_at_points,
                                   const gpp_structure* structure, int max_points, int allow_extrapolation,
                                   float* out, float* out_variance, int mem);

/* ---- ensemble optimal interpolation (EnSI) ----------------------------------------
 * replaces gridpp::optimal_interpolation_ensi (src/api/oi_ensi.cpp:114-568, Points
 * form; the Grid form :33-112 is the same call on a grid handle).  background and out
 * are [bgrid-size][ne], background_at_points is [points-size][ne]; obs and sigmas have
 * points-size elements.  Members that are invalid anywhere in the field are left
 * untouched (oi_ensi.cpp:187-201). */
int gpp_optimal_interpolation_ensi(gpp_points* bgrid, const float* background, int ne, gpp_points* points,
                                   const float* obs, const float* sigmas, const float* background_at_points,
                                   const gpp_structure* structure, int max_points, int allow_extrapolation,
                                   float* out, int mem);
int gpp_ensi_last_kernel_ms(float* ms);
/* Statistics of the calling thread's last gpp_optimal_interpolation_ensi: condition_passthrough = grid points left at their background
 * values because their E x E system is singular or not finite -- the count behind the reference's warning "Condition number error in N
 * points. Using raw values in those points." (src/api/oi_ensi.cpp:153,386-390,557-561), which the mirrors print; real_part_passthrough
 * = the reference's second counter (:154,423-426,562-566: an empty real part of the matrix square root), which cannot happen for the
 * symmetric square root the kernels form and is always 0. */
typedef struct gpp_ensi_stats {
    long long cells;
    long long condition_passthrough;
    long long real_part_passthrough;
    float kernel_ms;
} gpp_ensi_stats;
int gpp_ensi_last_stats(gpp_ensi_stats* stats);
/* 1: the Jacobi sweeps of the per-cell eigenproblem run to convergence (reference-grade last bits, ~2x the time);
 * 0 (default): they stop at |off-diagonal| <= 0.040 (E - 1) and a perturbation series supplies the rest (since round 4 this
 * mode meets the plain 1e-5 measure on every value of the randomised soak as well: worst 2.5e-6, the same as with converged sweeps).
 * Per calling thread. */
int gpp_ensi_set_convergence(int to_convergence);

/* gridpp::optimal_interpolation_ensi_multi_ebe / _ebesc / _utem (include/gridpp.h:311-441, src/api/oi_ensi_multi.cpp:329-1311;
 * Points overloads, a Grid is its row-major flattening).  variant: 1 = ebe, 2 = ebesc, 3 = utem.  bratios [bgrid-size];
 * background / background_corr / out [bgrid-size][ne]; pobs [points-size][ne] (ebe, ebesc) or [points-size] (utem); pratios
 * [points-size]; pbackground / pbackground_corr [points-size][ne].  background_corr / pbackground_corr are ignored by ebesc
 * (may be NULL).  Members invalid in any of the fields are left untouched (:395-418). */
int gpp_optimal_interpolation_ensi_multi(int variant, gpp_points* bgrid, const float* bratios, const float* background,
                                         const float* background_corr, int ne, gpp_points* points, const float* pobs,
                                         const float* pratios, const float* pbackground, const float* pbackground_corr,
                                         const gpp_structure* structure, int max_points, int allow_extrapolation, float* out, int mem);   /* hipEvent time of the last EnSI kernel (diagnostics / bench) */

/* ---- multi-GPU helpers (no counterpart in the reference, whose parallelism is OpenMP, src/api/oi.cpp:221) --------
 * One process per GPU.  The output grid is cut into contiguous row tiles (gpp_row_tile), every rank builds its tile's Grid and
 * the full Points, rank 0 owns the observation values of a step and broadcasts them (gpp_comm_broadcast: ncclBroadcast over
 * xGMI on the library stream), the neighbourhood filters get their halfwidth-row halos from the neighbouring ranks
 * (gpp_comm_halo_exchange: ncclSend / ncclRecv).  There is no other cross-tile dependence.  RCCL is loaded on first use. */
int gpp_row_tile(int ny, int rank, int world, int* row0, int* row1);
int gpp_comm_unique_id(char* id128);                              /* rank 0; the caller distributes the 128 bytes */
int gpp_comm_init(int rank, int world, const char* id128);        /* collective; after gpp_set_device */
int gpp_comm_rank(int* rank, int* world);                         /* (0, 1) without a communicator */
int gpp_comm_broadcast(void* device_buf, size_t bytes, int root); /* in place, device memory, library stream */
int gpp_comm_broadcast_host(void* host_buf, size_t bytes, int root); /* host memory, staged through HBM */
int gpp_comm_halo_exchange(const float* tile, int rows, size_t row_floats, int halfwidth, float* padded, int* top_rows);
int gpp_comm_destroy(void);

/* ---- neighbourhood filters (src/api/neighbourhood.cpp) -------------------------
 * input is [ny][nx] (is3d == 0, ne must be 1) or [ny][nx][ne] (is3d == 1); out is
 * [ny][nx].  Empty input (any extent 0) returns GPP_OK and writes nothing
 * (neighbourhood.cpp:33-34).  Windows are clipped at the domain edge; NaN / inf
 * values are ignored (util.cpp:16-18). */
/* gridpp::neighbourhood(vec2|vec3, halfwidth, statistic) (neighbourhood.cpp:12-242) */
int gpp_neighbourhood(const float* input, int ny, int nx, int ne, int is3d, int halfwidth, int statistic,
                      float* out, int mem);
/* gridpp::neighbourhood_brute_force (statistic != GPP_QUANTILE) and
 * gridpp::neighbourhood_quantile (statistic == GPP_QUANTILE, exact order statistics)
 * (neighbourhood.cpp:528-539,557-654; util.cpp:19-178) */
int gpp_neighbourhood_brute_force(const float* input, int ny, int nx, int ne, int halfwidth, int statistic,
                                  float quantile, float* out, int mem);
/* gridpp::neighbourhood_quantile_fast, all four overloads (neighbourhood.cpp:296-527):
 * quantile is one value (nq == 1) or a [ny][nx] field (nq == ny*nx); thresholds[nt]. */
int gpp_neighbourhood_quantile_fast(const float* input, int ny, int nx, int ne, int is3d,
                                    const float* quantile, int nq, int halfwidth,
                                    const float* thresholds, int nt, float* out, int mem);

/* ---- per-row statistics (src/api/util.cpp) -------------------------------------
 * array is [rows][len]; one result per row. */
/* gridpp::calc_statistic(vec|vec2, statistic) (util.cpp:19-110,208-215) */
int gpp_calc_statistic(const float* array, long rows, int len, int statistic, float* out, int mem);
/* gridpp::calc_quantile(vec|vec2, q) and (vec3, vec2 q) (util.cpp:111-207): nq == 1 or nq == rows */
int gpp_calc_quantile(const float* array, long rows, int len, const float* quantile, long nq, float* out, int mem);
/* gridpp::calc_even_quantiles (util.cpp:261-338) and, with only_valid = 1,
 * gridpp::get_neighbourhood_thresholds (neighbourhood.cpp:243-295) over the n
 * values of a flattened field.  out holds num floats; *count = number written. */
int gpp_calc_even_quantiles(const float* values, long n, int num, int only_valid, float* out, int* count, int mem);

/* per-call statistics of the last OI call on this thread (diagnostics / bench) */
typedef struct gpp_oi_stats {
    long long cells;          /* grid cells processed */
    long long cells_updated;  /* cells with at least one usable observation */
    long long solves;         /* local (P+R) factorisations actually performed */
    long long fallback_tiles; /* tiles k_oi_union handed to k_oi (or all tiles when the call was redone with the pivoted LU) */
    float kernel_ms;          /* hipEvent time of the OI kernel(s) on the library stream */
    float union_kernel_ms;    /* of which k_oi_union, first pass (one factorisation per tile); 0 when that kernel was not used */
    long long fallback_subtiles; /* work items (4 cells, or whole tiles) k_oi_union's list passes left to k_oi */
    long long big_cells;      /* grid points with more than 62 usable observations (done by k_oi_big) */
} gpp_oi_stats;
int gpp_oi_last_stats(gpp_oi_stats* stats);

/* Asynchronous optimal interpolation (no counterpart in the reference, whose calls return their result; this is for a caller that streams
 * analyses -- one per observation set -- through one GPU, e.g. a rank of the multi-GPU tiling, src/api/oi.cpp:221-338 has no state between
 * calls).  gpp_optimal_interpolation_full(..., mem = GPP_MEM_DEVICE | GPP_ASYNC) enqueues the call on the library stream and returns; every
 * such call is completed, in order, by ONE gpp_wait() of the same thread, which returns the status of that call (the results are in `out`
 * when it returns GPP_OK; gpp_oi_last_stats() then describes that call).  Only a call in the steady state of a repeated analysis (the same
 * Grid / Points handles as the call before) is really deferred; any other runs synchronously at once and its gpp_wait() returns immediately.
 * Until the wait returns the caller keeps inputs, outputs and handles alive and unchanged; at most 4 calls are deferred at a time (further
 * ones run synchronously).  gpp_pending() tells how many waits are outstanding.  A GPP_ASYNC call that returns an error (bad arguments, out
 * of memory: nothing was enqueued) is queued as a completed call too, so a caller may pair one gpp_wait() with every submission without looking
 * at the return code of the submission; that wait returns the same code and message again. */
int gpp_wait(void);
int gpp_pending(int* count);

#ifdef __cplusplus
}
#endif
#endif
