/*
 * gridpp_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * A plain-C, single-threaded CPU restatement of the metno/gridpp hot path
 * (optimal interpolation, Barnes structure function, radius / nearest
 * neighbour search, neighbourhood filters, EnSI).  It exists so that the HIP
 * kernels can be checked against the reference algorithm.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
 * library; the product path (gridpp_amd + libgridpp_hip.so) never does.
 *
 * Every function cites the reference file:line it restates (paths relative to
 * the upstream repository root).  No reference source text is copied: the
 * reference is C++ on nested std::vector + Boost R-tree + Armadillo, this file
 * is C on flat arrays with a linear scan and an in-file LU / Jacobi.
 *
 * Third-party arithmetic that is NOT in the reference tree:
 *   - Boost.Geometry index::rtree (1.71/1.72 in the reference CI / wheels):
 *     contributes no arithmetic, only WHICH indices and in what ORDER.  Here:
 *     linear scan in index order.  Order only matters for exact-float ties in
 *     the top-max_points cut (src/api/oi.cpp:266 uses an unstable std::sort),
 *     where this oracle DEFINES the tie-break: higher rho first, then lower
 *     observation index.
 *   - Armadillo inv()/eig_sym()/rcond() -> LAPACK dgetrf+dgetri / dsyev:
 *     restated as partial-pivot LU inverse and cyclic Jacobi, all in double.
 *
 * Parity pin: tests/test_oracle_golden.py checks this file against the
 * known-answer values of the reference's own unit tests (tests/golden/ JSON files,
 * each entry citing the reference test file:line).  The reference itself is
 * unbuildable in this image (needs Boost, Armadillo, LAPACK, SWIG).
 */
#include <math.h>
#ifdef _OPENMP
#include <omp.h>
#endif
#include <stdlib.h>
#include <string.h>
#include <float.h>

#define ORC_OK 0
#define ORC_EINVAL -1
#define ORC_ESINGULAR -2

/* include/gridpp.h:55 */
static const double orc_radius_earth = 6.378137e6;

/* src/api/util.cpp:16-18 (value != NAN is always true) */
static int orc_valid(float v) { return !isnan(v) && !isinf(v); }
int orc_is_valid(float v) { return orc_valid(v); }

/* ------------------------------------------------------------------------ */
/* coordinates: src/api/util.cpp:583-624                                      */
/* ------------------------------------------------------------------------ */
int orc_convert_coordinates(const float* lats, const float* lons, int n, int type,
                            float* x, float* y, float* z) {
    for(int i = 0; i < n; i++) {
        float lat = lats[i], lon = lons[i];
        int ok_lat = (type == 1) ? orc_valid(lat) : (orc_valid(lat) && lat >= -90.001 && lat <= 90.001);
        if(!ok_lat || !orc_valid(lon)) return ORC_EINVAL;
        if(type == 1) { /* Cartesian: util.cpp:601-604 */
            x[i] = lon; y[i] = lat; z[i] = 0;
        } else {        /* Geodetic: util.cpp:606-612, double trig, float store */
            double lonr = M_PI / 180 * lon;
            double latr = M_PI / 180 * lat;
            x[i] = (float)(cos(latr) * cos(lonr) * orc_radius_earth);
            y[i] = (float)(cos(latr) * sin(lonr) * orc_radius_earth);
            z[i] = (float)(sin(latr) * orc_radius_earth);
        }
    }
    return ORC_OK;
}

/* src/api/kdtree.cpp:192-194 -- float arithmetic, no FMA (x86-64 baseline build) */
/* (built with -ffp-contract=off and no -march flag, see oracle/Makefile) */
static float orc_straight_distance(float x0, float y0, float z0, float x1, float y1, float z1) {
    return sqrtf((x0 - x1) * (x0 - x1) + (y0 - y1) * (y0 - y1) + (z0 - z1) * (z0 - z1));
}
float orc_calc_straight_distance(float x0, float y0, float z0, float x1, float y1, float z1) {
    return orc_straight_distance(x0, y0, z0, x1, y1, z1);
}

/* src/api/kdtree.cpp:107-136 */
float orc_calc_distance(float lat1, float lon1, float lat2, float lon2, int type) {
    if(type == 1) {
        float dx = lon1 - lon2, dy = lat1 - lat2;
        return sqrtf(dx * dx + dy * dy);
    }
    if(lat1 == lat2 && lon1 == lon2) return 0;
    /* deg2rad returns float (kdtree.cpp:195-197) */
    double lat1r = (float)(lat1 * M_PI / 180), lat2r = (float)(lat2 * M_PI / 180);
    double lon1r = (float)(lon1 * M_PI / 180), lon2r = (float)(lon2 * M_PI / 180);
    double ratio = cos(lat1r)*cos(lon1r)*cos(lat2r)*cos(lon2r) + cos(lat1r)*sin(lon1r)*cos(lat2r)*sin(lon2r) + sin(lat1r)*sin(lat2r);
    return (float)(acos(ratio) * 6.378137e6);
}

/* ------------------------------------------------------------------------ */
/* Barnes structure function: src/api/structure.cpp                          */
/* ------------------------------------------------------------------------ */
/* structure.cpp:26-34: v float, exponent and exp in double, result float */
float orc_barnes_rho(float dist, float length) {
    if(!orc_valid(length) || length == 0) return 1;
    if(!orc_valid(dist)) return 0;
    float v = dist / length;
    return (float)exp(-0.5 * v * v);
}
/* structure.cpp:143-167 (scalar ctor): m_min_rho (float) */
float orc_barnes_min_rho(float h, float hmax) {
    if(orc_valid(hmax)) return (float)exp(pow((double)(hmax / h), 2) / -2);
    return 0.0013f; /* structure.cpp:5 */
}
/* structure.cpp:280-282: sqrt(-2*log(m_min_rho)) * h with float overloads
 * (<math.h> is pulled in through the Boost.Math headers of gridpp.h).  The
 * double-precision reading gives the same float for every value the reference
 * tests pin (tests/test_barnes_structure.py:46-66,85-97). */
float orc_barnes_localization_distance(float h, float min_rho) {
    return sqrtf(-2 * logf(min_rho)) * h;
}
/* structure.cpp:185-230, non-spatial branch :215-228 */
float orc_barnes_corr(float x1, float y1, float z1, float e1, float l1,
                      float x2, float y2, float z2, float e2, float l2,
                      float h, float v, float w, float loc_dist) {
    float hdist = orc_straight_distance(x1, y1, z1, x2, y2, z2);
    if(hdist > loc_dist) return 0;
    float rho = orc_barnes_rho(hdist, h);
    if(orc_valid(e1) && orc_valid(e2)) rho *= orc_barnes_rho(e1 - e2, v);
    if(orc_valid(l1) && orc_valid(l2)) rho *= orc_barnes_rho(l1 - l2, w);
    return rho;
}

/* ------------------------------------------------------------------------ */
/* generic structure functions: src/api/structure.cpp:26-86 (rho kernels),     */
/* :90-138 (MultipleStructure), :287-312 (Cressman), :317-459 (SOAR), :467-612 */
/* (TOAR), :618-759 (Powerlaw), :765-904 (Linear), :910-944 (CrossValidation). */
/* Scalar forms only.  float/double follow the C++ overloads the reference     */
/* gets with <math.h> in scope (exp(float) -> expf, abs(float) -> fabsf).       */
/* ------------------------------------------------------------------------ */
enum { SK_BARNES = 0, SK_CRESSMAN = 1, SK_SOAR = 2, SK_TOAR = 3, SK_POWERLAW = 4, SK_LINEAR = 5 };
typedef struct {
    int kh, kv, kw;      /* kernel of the horizontal / vertical / laf factor */
    float h, v, w;       /* scales */
    float loc;           /* localization distance of the horizontal structure */
    int cv; float cv_dist;   /* CrossValidation */
} orc_struct;

float orc_rho(int kind, float dist, float length) {
    if(kind == SK_LINEAR) {                                   /* structure.cpp:76-86 */
        if(!orc_valid(length) || length < 0) return 1;
        if(!orc_valid(dist)) return 0;
        float absdiff = fabsf(dist);
        if(absdiff > 1) absdiff = 1;
        return (1 - (1 - length) * absdiff);
    }
    if(!orc_valid(length) || length == 0) return 1;
    if(!orc_valid(dist)) return 0;
    if(kind == SK_BARNES) { float v = dist / length; return (float)exp(-0.5 * v * v); }
    if(kind == SK_CRESSMAN) {                                 /* :35-44 */
        if(dist >= length) return 0;
        return (length * length - dist * dist) / (length * length + dist * dist);
    }
    float v = dist / length;
    if(kind == SK_SOAR) return (1 + v) * expf(-v);            /* :46-54 */
    if(kind == SK_TOAR) return (1 + v + (v * v) / 3) * expf(-v);   /* :56-64 */
    if(kind == SK_POWERLAW) return (float)(1 / (1 + 0.5 * v * v)); /* :66-74 */
    return NAN;
}
/* m_min_rho of the scalar constructors (:154-157,328-331,478-481,629-632,776-779) */
float orc_structure_min_rho(int kind, float h, float hmax) {
    if(!orc_valid(hmax) || kind == SK_LINEAR || kind == SK_CRESSMAN) return 0.0013f;
    if(kind == SK_BARNES) return (float)exp(pow((double)(hmax / h), 2) / -2);
    if(kind == SK_SOAR) return (1 + hmax / h) * expf(-hmax / h);
    if(kind == SK_TOAR) return (float)((1 + hmax / h + pow((double)(hmax / h), 2) / 3) * expf(-hmax / h));
    if(kind == SK_POWERLAW) return (float)(1 / (1 + 0.5 * pow((double)(hmax / h), 2)));
    return 0.0013f;
}
/* localization_distance(h) (:280-282, 7-12 [Cressman: base class, = h], 454-459, 604-610, 755-757, 902-904) */
float orc_structure_localization_distance(int kind, float h, float min_rho) {
    if(kind == SK_BARNES) return sqrtf(-2 * logf(min_rho)) * h;
    if(kind == SK_CRESSMAN) return h;
    if(kind == SK_SOAR) { float lm = logf(min_rho); return (-lm + logf(-lm)) * h; }
    if(kind == SK_TOAR) { float lm = logf(min_rho); float ll = logf(-logf(min_rho)); return (float)(((double)(-lm + ll) + 0.5 * ll) * h); }
    if(kind == SK_POWERLAW) return sqrtf(2 * (1 - min_rho) / min_rho) * h;
    return 0;
}
/* corr / corr_background of a (Multiple / CrossValidation-wrapped) scalar structure */
float orc_corr_g(const orc_struct* s, float x1, float y1, float z1, float e1, float l1,
                 float x2, float y2, float z2, float e2, float l2, int background) {
    float hdist = orc_straight_distance(x1, y1, z1, x2, y2, z2);
    if(background && s->cv && hdist <= s->cv_dist) return 0;           /* :918-925 */
    if(s->kh != SK_CRESSMAN && hdist > s->loc) return 0;               /* e.g. :216-217; Cressman has no cut (:300-312) */
    float rho = orc_rho(s->kh, hdist, s->h);
    if(orc_valid(e1) && orc_valid(e2)) rho *= orc_rho(s->kv, e1 - e2, s->v);
    if(orc_valid(l1) && orc_valid(l2)) rho *= orc_rho(s->kw, l1 - l2, s->w);
    return rho;
}
float orc_corr_generic(int kh, int kv, int kw, float h, float v, float w, float loc, int cv, float cv_dist, int background,
                       float x1, float y1, float z1, float e1, float l1, float x2, float y2, float z2, float e2, float l2) {
    orc_struct s = {kh, kv, kw, h, v, w, loc, cv, cv_dist};
    return orc_corr_g(&s, x1, y1, z1, e1, l1, x2, y2, z2, e2, l2, background);
}

/* ------------------------------------------------------------------------ */
/* radius / nearest-neighbour search: src/api/kdtree.cpp:39-106,241-270       */
/* ------------------------------------------------------------------------ */
/* strictly-inside box test (kdtree.cpp:46,53: index::within) then
 * within_radius (kdtree.cpp:247-260) */
static int orc_in_radius(float qx, float qy, float qz, float px, float py, float pz,
                         float radius, int include_match) {
    float lox = qx - radius, hix = qx + radius;
    float loy = qy - radius, hiy = qy + radius;
    float loz = qz - radius, hiz = qz + radius;
    if(!(px > lox && px < hix && py > loy && py < hiy && pz > loz && pz < hiz)) return 0;
    float d = orc_straight_distance(px, py, pz, qx, qy, qz);
    if(include_match) return d <= radius;
    return d <= radius && d > 0;
}
/* returns count; indices in ascending index order (the oracle's traversal order) */
int orc_get_neighbours(const float* px, const float* py, const float* pz, int n,
                       float qx, float qy, float qz, float radius, int include_match, int* out) {
    int c = 0;
    for(int i = 0; i < n; i++)
        if(orc_in_radius(qx, qy, qz, px[i], py[i], pz[i], radius, include_match)) out[c++] = i;
    return c;
}
/* kdtree.cpp:82-106: k nearest by chord distance.  Ties: the R-tree's order is
 * unspecified; the oracle takes the lowest index.  Comparison is done on the
 * float squared distance like Boost's comparable_distance. */
int orc_nearest_neighbour(const float* px, const float* py, const float* pz, int n,
                          float qx, float qy, float qz, int include_match) {
    int best = -1; float bestd = 0;
    for(int i = 0; i < n; i++) {
        if(!include_match && px[i] == qx && py[i] == qy && pz[i] == qz) continue; /* kdtree.cpp:265-270 */
        float dx = px[i] - qx, dy = py[i] - qy, dz = pz[i] - qz;
        float s = dx * dx + dy * dy + dz * dz;
        if(best < 0 || s < bestd) { best = i; bestd = s; }
    }
    return best;
}

/* ------------------------------------------------------------------------ */
/* dense double helpers (stand in for Armadillo -> LAPACK)                   */
/* ------------------------------------------------------------------------ */
/* inverse via LU with partial pivoting (dgetrf + dgetri semantics); returns
 * ORC_ESINGULAR on an exactly zero pivot (arma::inv then throws). a is n*n
 * row-major, overwritten by the inverse. */
static int orc_inv(double* a, int n) {
    int* piv = (int*)malloc(sizeof(int) * n);
    double* inv = (double*)calloc((size_t)n * n, sizeof(double));
    for(int i = 0; i < n; i++) inv[i * n + i] = 1;
    for(int k = 0; k < n; k++) {
        int p = k; double m = fabs(a[k * n + k]);
        for(int i = k + 1; i < n; i++) if(fabs(a[i * n + k]) > m) { m = fabs(a[i * n + k]); p = i; }
        if(m == 0 || isnan(m)) { free(piv); free(inv); return ORC_ESINGULAR; }
        piv[k] = p;
        if(p != k) for(int j = 0; j < n; j++) {
            double t = a[k * n + j]; a[k * n + j] = a[p * n + j]; a[p * n + j] = t;
            t = inv[k * n + j]; inv[k * n + j] = inv[p * n + j]; inv[p * n + j] = t;
        }
        double d = a[k * n + k];
        for(int i = k + 1; i < n; i++) {
            double f = a[i * n + k] / d;
            if(f == 0) continue;
            for(int j = k; j < n; j++) a[i * n + j] -= f * a[k * n + j];
            for(int j = 0; j < n; j++) inv[i * n + j] -= f * inv[k * n + j];
        }
    }
    for(int k = n - 1; k >= 0; k--) {
        double d = a[k * n + k];
        for(int j = 0; j < n; j++) inv[k * n + j] /= d;
        for(int i = 0; i < k; i++) {
            double f = a[i * n + k];
            if(f == 0) continue;
            for(int j = 0; j < n; j++) inv[i * n + j] -= f * inv[k * n + j];
        }
    }
    memcpy(a, inv, sizeof(double) * n * n);
    free(piv); free(inv);
    return ORC_OK;
}

/* symmetric eigen-decomposition by cyclic Jacobi (stand-in for dsyev).
 * a: n*n symmetric (destroyed), w: eigenvalues, v: eigenvectors in columns. */
static void orc_eig_sym(double* a, int n, double* w, double* v) {
    for(int i = 0; i < n; i++) for(int j = 0; j < n; j++) v[i * n + j] = (i == j);
    for(int sweep = 0; sweep < 100; sweep++) {
        double off = 0;
        for(int i = 0; i < n; i++) for(int j = i + 1; j < n; j++) off += a[i * n + j] * a[i * n + j];
        if(off < 1e-300) break;
        for(int p = 0; p < n; p++) for(int q = p + 1; q < n; q++) {
            double apq = a[p * n + q];
            if(apq == 0) continue;
            double theta = (a[q * n + q] - a[p * n + p]) / (2 * apq);
            double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1));
            double c = 1 / sqrt(t * t + 1), s = t * c;
            for(int k = 0; k < n; k++) {
                double akp = a[k * n + p], akq = a[k * n + q];
                a[k * n + p] = c * akp - s * akq; a[k * n + q] = s * akp + c * akq;
            }
            for(int k = 0; k < n; k++) {
                double apk = a[p * n + k], aqk = a[q * n + k];
                a[p * n + k] = c * apk - s * aqk; a[q * n + k] = s * apk + c * aqk;
            }
            for(int k = 0; k < n; k++) {
                double vkp = v[k * n + p], vkq = v[k * n + q];
                v[k * n + p] = c * vkp - s * vkq; v[k * n + q] = s * vkp + c * vkq;
            }
        }
    }
    for(int i = 0; i < n; i++) w[i] = a[i * n + i];
}

/* ------------------------------------------------------------------------ */
/* candidate selection shared by OI and EnSI                                  */
/* src/api/oi.cpp:229-285 / src/api/oi_ensi.cpp:213-269                       */
/* ------------------------------------------------------------------------ */
typedef struct { float rho; int idx; } orc_pair;
static int orc_pair_cmp(const void* a, const void* b) {
    const orc_pair* p = (const orc_pair*)a; const orc_pair* q = (const orc_pair*)b;
    if(p->rho > q->rho) return -1;
    if(p->rho < q->rho) return 1;
    return (p->idx > q->idx) - (p->idx < q->idx); /* oracle-defined tie-break */
}
static int orc_pair_idx_cmp(const void* a, const void* b) {
    const orc_pair* p = (const orc_pair*)a; const orc_pair* q = (const orc_pair*)b;
    return (p->idx > q->idx) - (p->idx < q->idx);
}
/* Returns number selected; sel[] = observation indices, srho[] = rho (float). */
/* cand == NULL: linear scan over all observations (the reference's R-tree returns the same set; its order is unspecified
   and only matters for exact rho ties).  cand != NULL: the same test on a pre-filtered list in ascending index order (the
   cell list of the CPU baseline below, any order) -- a superset of the observations in range; the kept set is put back into
   index order where the reference keeps candidate order, so the result is identical. */
static int orc_select_from(const int* cand, int ncand, float gx, float gy, float gz, float ge, float gl,
                      int nS, const float* ox, const float* oy, const float* oz,
                      const float* oe, const float* ol,
                      const float* pobs, const float* pbg /* may be NULL: EnSI */,
                      float h, float v, float w, float loc, int max_points,
                      orc_pair* work, int* sel, float* srho) {
    int n = 0;
    const int nloop = cand ? ncand : nS;
    for(int k = 0; k < nloop; k++) {
        const int s = cand ? cand[k] : k;
        if(!orc_in_radius(gx, gy, gz, ox[s], oy[s], oz[s], loc, 1)) continue;      /* oi.cpp:233 */
        float rho = orc_barnes_corr(gx, gy, gz, ge, gl, ox[s], oy[s], oz[s], oe[s], ol[s], h, v, w, loc); /* :250 */
        if(!orc_valid(pobs[s])) continue;                                            /* :252 */
        if(pbg && !orc_valid(pbg[s])) continue;
        if(rho > 0) { work[n].rho = rho; work[n].idx = s; n++; }                     /* :253-255 */
    }
    if(max_points > 0 && n > max_points) {                                           /* :262-273 */
        qsort(work, n, sizeof(orc_pair), orc_pair_cmp);
        n = max_points;
    }
    else if(cand) qsort(work, n, sizeof(orc_pair), orc_pair_idx_cmp);   /* kept in candidate order: that of the linear scan */
    for(int i = 0; i < n; i++) { sel[i] = work[i].idx; srho[i] = work[i].rho; }
    return n;
}
static int orc_select(float gx, float gy, float gz, float ge, float gl,
                      int nS, const float* ox, const float* oy, const float* oz,
                      const float* oe, const float* ol,
                      const float* pobs, const float* pbg,
                      float h, float v, float w, float loc, int max_points,
                      orc_pair* work, int* sel, float* srho) {
    return orc_select_from(NULL, 0, gx, gy, gz, ge, gl, nS, ox, oy, oz, oe, ol, pobs, pbg, h, v, w, loc, max_points, work, sel, srho);
}

/* the local analysis of one grid point from its selected observations (oi.cpp:289-337) */
static int orc_oi_solve_cell(int y, int lS, const int* sel, const float* srho,
                         const float* background, const float* bvariance,
                         const float* ox, const float* oy, const float* oz, const float* oelev, const float* olaf,
                         const float* pobs, const float* pratios, const float* pbackground,
                         float h, float v, float w, float loc, int allow_extrapolation, float* out, float* out_var,
                         double** scratch, size_t* scratch_cap) {
    /* caller-owned work space (one per thread), grown on demand: A [lS x lS], d [lS], G [lS] */
    const size_t need = (size_t)lS * lS + 2 * (size_t)lS;
    if(*scratch_cap < need) { free(*scratch); *scratch = (double*)malloc(sizeof(double) * need); *scratch_cap = need; }
    double* A = *scratch;
    double* d = A + (size_t)lS * lS;
    double* G = d + lS;
    for(int i = 0; i < lS; i++) {                                               /* :297-314 */
        int si = sel[i];
        d[i] = (double)pobs[si] - (double)pbackground[si];                      /* lObs - lY in double */
        G[i] = (double)srho[i];
        for(int j = 0; j < lS; j++) {
            int sj = sel[j];
            float c = orc_barnes_corr(ox[si], oy[si], oz[si], oelev[si], olaf[si],
                                      ox[sj], oy[sj], oz[sj], oelev[sj], olaf[sj], h, v, w, loc);
            A[i * lS + j] = (double)c;
        }
        A[i * lS + i] += (double)pratios[si];                                   /* lP + lR */
    }
    if(orc_inv(A, lS) != ORC_OK) return ORC_ESINGULAR; /* :315 */
    double dx = 0, a00 = 0; float maxInc = 0, minInc = 0;
    for(int j = 0; j < lS; j++) {
        double gsr = 0;
        for(int i = 0; i < lS; i++) gsr += G[i] * A[i * lS + j];                /* lGSR = lG * inv */
        dx += gsr * d[j];                                                       /* :316 */
        a00 += gsr * G[j];                                                      /* :336 */
    }
    for(int j = 0; j < lS; j++) {                                               /* :319-320 */
        float dj = (float)d[j];
        if(j == 0 || dj > maxInc) maxInc = dj;
        if(j == 0 || dj < minInc) minInc = dj;
    }
    float increment = (float)dx;                                                /* :317 */
    if(!allow_extrapolation) {                                                  /* :318-334 */
        if(maxInc > 0 && increment > maxInc) increment = maxInc;
        else if(maxInc < 0 && increment > 0) increment = maxInc;
        else if(minInc < 0 && increment < minInc) increment = minInc;
        else if(minInc > 0 && increment < 0) increment = minInc;
    }
    out[y] = background[y] + increment;                                         /* :335 */
    out_var[y] = (float)((double)bvariance[y] * (1 - a00));                     /* :337 */
    return ORC_OK;
}
/* ------------------------------------------------------------------------ */
/* optimal_interpolation_full (Points): src/api/oi.cpp:138-341                */
/* Arrays are flat; obs x/y/z from orc_convert_coordinates.                   */
/* out / out_var must be preallocated [nY]; they are initialised to            */
/* background / bvariance (oi.cpp:198-199).  Range [y0,y1) lets the CPU        */
/* baseline time a bounded sample.                                             */
/* ------------------------------------------------------------------------ */
int orc_oi_full_range(int y0, int y1,
                const float* gx, const float* gy, const float* gz, const float* gelev, const float* glaf,
                const float* background, const float* bvariance,
                int nS, const float* ox, const float* oy, const float* oz, const float* oelev, const float* olaf,
                const float* pobs, const float* obs_variance, const float* pbackground, const float* bvariance_at_points,
                float h, float v, float w, float min_rho,
                int max_points, int allow_extrapolation,
                float* out, float* out_var) {
    if(max_points < 0) return ORC_EINVAL;
    for(int y = y0; y < y1; y++) { out[y] = background[y]; out_var[y] = bvariance[y]; }
    if(nS == 0) return ORC_OK;                                                     /* oi.cpp:189-190 */
    float* pratios = (float*)malloc(sizeof(float) * nS);
    for(int s = 0; s < nS; s++) pratios[s] = obs_variance[s] / bvariance_at_points[s]; /* :192-195 */
    float loc = orc_barnes_localization_distance(h, min_rho);                       /* :229 */
    orc_pair* work = (orc_pair*)malloc(sizeof(orc_pair) * nS);
    int* sel = (int*)malloc(sizeof(int) * nS);
    float* srho = (float*)malloc(sizeof(float) * nS);
    double* scratch = NULL; size_t scratch_cap = 0;
    int rc = ORC_OK;
    for(int y = y0; y < y1; y++) {
        if(!orc_valid(background[y])) continue;                                     /* :223 */
        int lS = orc_select(gx[y], gy[y], gz[y], gelev[y], glaf[y], nS, ox, oy, oz, oelev, olaf,
                            pobs, pbackground, h, v, w, loc, max_points, work, sel, srho);
        if(lS == 0) continue;                                                       /* :234,284 */
        rc = orc_oi_solve_cell(y, lS, sel, srho, background, bvariance, ox, oy, oz, oelev, olaf, pobs, pratios, pbackground,
                               h, v, w, loc, allow_extrapolation, out, out_var, &scratch, &scratch_cap);
        if(rc != ORC_OK) break;
    }
    free(pratios); free(work); free(sel); free(srho); free(scratch);
    return rc;
}
int orc_oi_full(int nY,
                const float* gx, const float* gy, const float* gz, const float* gelev, const float* glaf,
                const float* background, const float* bvariance,
                int nS, const float* ox, const float* oy, const float* oz, const float* oelev, const float* olaf,
                const float* pobs, const float* obs_variance, const float* pbackground, const float* bvariance_at_points,
                float h, float v, float w, float min_rho,
                int max_points, int allow_extrapolation,
                float* out, float* out_var) {
    return orc_oi_full_range(0, nY, gx, gy, gz, gelev, glaf, background, bvariance, nS, ox, oy, oz, oelev, olaf,
                             pobs, obs_variance, pbackground, bvariance_at_points, h, v, w, min_rho,
                             max_points, allow_extrapolation, out, out_var);
}
/* ------------------------------------------------------------------------ */
/* CPU baseline of bench.py: the same loop as orc_oi_full_range with (a) the  */
/* radius query answered from a cell list instead of a linear scan -- the     */
/* algorithm class of the reference's R-tree query (kdtree.cpp:39-60): only   */
/* the observations of the bins that overlap the query box are tested, and    */
/* the kept ones are put back into index order, so every number computed     */
/* from them is identical to the linear scan (tests/test_oracle_baseline.py) --   */
/* and (b) the grid points spread over OpenMP threads with a dynamic schedule */
/* (the reference: `#pragma omp parallel for`, oi.cpp:221).                    */
/* nthreads <= 0: OMP default.  use_cell_list = 0: linear scan.               */
/* ------------------------------------------------------------------------ */
int orc_oi_full_omp(int nthreads, int use_cell_list, int y0, int y1,
                const float* gx, const float* gy, const float* gz, const float* gelev, const float* glaf,
                const float* background, const float* bvariance,
                int nS, const float* ox, const float* oy, const float* oz, const float* oelev, const float* olaf,
                const float* pobs, const float* obs_variance, const float* pbackground, const float* bvariance_at_points,
                float h, float v, float w, float min_rho,
                int max_points, int allow_extrapolation,
                float* out, float* out_var) {
    if(max_points < 0) return ORC_EINVAL;
    for(int y = y0; y < y1; y++) { out[y] = background[y]; out_var[y] = bvariance[y]; }
    if(nS == 0) return ORC_OK;
    float* pratios = (float*)malloc(sizeof(float) * nS);
    for(int s = 0; s < nS; s++) pratios[s] = obs_variance[s] / bvariance_at_points[s];
    const float loc = orc_barnes_localization_distance(h, min_rho);
    /* cell list on the two axes with the widest extent; bin edge = half the query radius */
    int ax = 0, ay = 1, nbx = 1, nby = 1;
    float amin = 0, bmin = 0, inv = 0;
    int* bin_start = NULL; int* bin_items = NULL;
    if(use_cell_list) {
        const float* c[3] = {ox, oy, oz};
        float lo[3], hi[3];
        for(int k = 0; k < 3; k++) { lo[k] = hi[k] = c[k][0]; for(int s = 1; s < nS; s++) { if(c[k][s] < lo[k]) lo[k] = c[k][s]; if(c[k][s] > hi[k]) hi[k] = c[k][s]; } }
        int order[3] = {0, 1, 2};
        for(int a = 0; a < 3; a++) for(int b = a + 1; b < 3; b++) if(hi[order[b]] - lo[order[b]] > hi[order[a]] - lo[order[a]]) { int t = order[a]; order[a] = order[b]; order[b] = t; }
        ax = order[0]; ay = order[1];
        const float edge = loc > 0 ? loc * 0.5f : 1.0f;
        inv = 1.0f / edge; amin = lo[ax]; bmin = lo[ay];
        nbx = (int)((hi[ax] - lo[ax]) * inv) + 1; nby = (int)((hi[ay] - lo[ay]) * inv) + 1;
        if((long)nbx * nby > 4000000L) { nbx = nby = 1; inv = 0; }
        bin_start = (int*)calloc((size_t)nbx * nby + 1, sizeof(int));
        bin_items = (int*)malloc(sizeof(int) * nS);
        int* bin_of = (int*)malloc(sizeof(int) * nS);
        for(int s = 0; s < nS; s++) {
            int bx = (int)((c[ax][s] - amin) * inv), by = (int)((c[ay][s] - bmin) * inv);
            if(bx < 0) bx = 0; if(bx >= nbx) bx = nbx - 1; if(by < 0) by = 0; if(by >= nby) by = nby - 1;
            bin_of[s] = by * nbx + bx; bin_start[bin_of[s] + 1]++;
        }
        for(int k = 0; k < nbx * nby; k++) bin_start[k + 1] += bin_start[k];
        int* fill = (int*)malloc(sizeof(int) * (size_t)nbx * nby);
        for(int k = 0; k < nbx * nby; k++) fill[k] = bin_start[k];
        for(int s = 0; s < nS; s++) bin_items[fill[bin_of[s]]++] = s;      /* ascending index inside a bin */
        free(fill); free(bin_of);
    }
    int rc = ORC_OK;
#ifdef _OPENMP
    if(nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel
    {
        orc_pair* work = (orc_pair*)malloc(sizeof(orc_pair) * nS);
        int* sel = (int*)malloc(sizeof(int) * nS);
        float* srho = (float*)malloc(sizeof(float) * nS);
        int* cand = (int*)malloc(sizeof(int) * nS);
        double* scratch = NULL; size_t scratch_cap = 0;
#pragma omp for schedule(dynamic, 64)
        for(int y = y0; y < y1; y++) {
            if(rc != ORC_OK) continue;
            if(!orc_valid(background[y])) continue;
            int lS;
            if(use_cell_list) {
                const float g[3] = {gx[y], gy[y], gz[y]};
                int bx0 = (int)floorf((g[ax] - loc - amin) * inv), bx1 = (int)floorf((g[ax] + loc - amin) * inv);
                int by0 = (int)floorf((g[ay] - loc - bmin) * inv), by1 = (int)floorf((g[ay] + loc - bmin) * inv);
                if(bx0 < 0) bx0 = 0; if(by0 < 0) by0 = 0; if(bx1 >= nbx) bx1 = nbx - 1; if(by1 >= nby) by1 = nby - 1;
                int nc = 0;
                for(int by = by0; by <= by1; by++)
                    for(int k = bin_start[by * nbx + bx0]; k < bin_start[by * nbx + bx1 + 1]; k++) cand[nc++] = bin_items[k];
                lS = orc_select_from(cand, nc, gx[y], gy[y], gz[y], gelev[y], glaf[y], nS, ox, oy, oz, oelev, olaf,
                                     pobs, pbackground, h, v, w, loc, max_points, work, sel, srho);
            }
            else lS = orc_select(gx[y], gy[y], gz[y], gelev[y], glaf[y], nS, ox, oy, oz, oelev, olaf,
                                 pobs, pbackground, h, v, w, loc, max_points, work, sel, srho);
            if(lS == 0) continue;
            int r = orc_oi_solve_cell(y, lS, sel, srho, background, bvariance, ox, oy, oz, oelev, olaf, pobs, pratios, pbackground,
                                      h, v, w, loc, allow_extrapolation, out, out_var, &scratch, &scratch_cap);
            if(r != ORC_OK) {
#pragma omp critical
                rc = r;
            }
        }
        free(work); free(sel); free(srho); free(cand); free(scratch);
    }
    free(pratios); free(bin_start); free(bin_items);
    return rc;
}
int orc_omp_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
/* Threads of the loops the REFERENCE runs under `#pragma omp parallel for` in the neighbourhood family                   */
/* (neighbourhood.cpp:101 the window loop of the summed-area table, :453 the threshold loop and :483 the row loop of     */
/* quantile_fast).  1 (the default) keeps every oracle function the plain serial loop the tests compare against; the     */
/* cpu_baseline leg of bench.py raises it for its timed calls (same bits: the iterations are independent).               */
static int orc_nbh_threads = 1;
int orc_set_neighbourhood_threads(int n) {
    int old = orc_nbh_threads;
    orc_nbh_threads = n < 1 ? 1 : n;
    return old;
}
/* ------------------------------------------------------------------------ */
/* optimal_interpolation_full with a generic scalar structure (same loop as   */
/* orc_oi_full_range; P is filled with corr(obs_i, obs_j) and may be           */
/* non-symmetric, e.g. SOAR/TOAR/Cressman vertical factors on signed           */
/* elevation differences, so the general inverse is what matters)              */
/* ------------------------------------------------------------------------ */
int orc_oi_full_generic(int nY,
                const float* gx, const float* gy, const float* gz, const float* gelev, const float* glaf,
                const float* background, const float* bvariance,
                int nS, const float* ox, const float* oy, const float* oz, const float* oelev, const float* olaf,
                const float* pobs, const float* obs_variance, const float* pbackground, const float* bvariance_at_points,
                int kh, int kv, int kw, float h, float v, float w, float loc, int cv, float cv_dist,
                int max_points, int allow_extrapolation, float* out, float* out_var,
                /* spatially varying form (structure.cpp:168-214): parameters at the FIRST point of corr(p1, p2);
                   per background point [nY] and per observation [nS]; all NULL for the scalar forms */
                const float* c_h, const float* c_v, const float* c_w, const float* c_R,
                const float* o_h, const float* o_v, const float* o_w, const float* o_R) {
    if(max_points < 0) return ORC_EINVAL;
    orc_struct st = {kh, kv, kw, h, v, w, loc, cv, cv_dist};
    for(int y = 0; y < nY; y++) { out[y] = background[y]; out_var[y] = bvariance[y]; }
    if(nS == 0) return ORC_OK;
    float* pratios = (float*)malloc(sizeof(float) * nS);
    for(int s = 0; s < nS; s++) pratios[s] = obs_variance[s] / bvariance_at_points[s];
    orc_pair* work = (orc_pair*)malloc(sizeof(orc_pair) * nS);
    int rc = ORC_OK;
    for(int y = 0; y < nY && rc == ORC_OK; y++) {
        if(!orc_valid(background[y])) continue;
        int n = 0;
        orc_struct sc = st;   /* the structure as seen from this grid point */
        if(c_h) { sc.h = c_h[y]; sc.v = c_v[y]; sc.w = c_w[y]; sc.loc = c_R[y]; }
        for(int s = 0; s < nS; s++) {
            if(!orc_in_radius(gx[y], gy[y], gz[y], ox[s], oy[s], oz[s], sc.loc, 1)) continue;
            float rho = orc_corr_g(&sc, gx[y], gy[y], gz[y], gelev[y], glaf[y], ox[s], oy[s], oz[s], oelev[s], olaf[s], 1);
            if(!orc_valid(pobs[s]) || !orc_valid(pbackground[s])) continue;
            if(rho > 0) { work[n].rho = rho; work[n].idx = s; n++; }
        }
        if(max_points > 0 && n > max_points) { qsort(work, n, sizeof(orc_pair), orc_pair_cmp); n = max_points; }
        int lS = n;
        if(lS == 0) continue;
        double* A = (double*)malloc(sizeof(double) * lS * lS);
        double* d = (double*)malloc(sizeof(double) * lS);
        double* G = (double*)malloc(sizeof(double) * lS);
        for(int i = 0; i < lS; i++) {
            int si = work[i].idx;
            d[i] = (double)pobs[si] - (double)pbackground[si];
            G[i] = (double)work[i].rho;
            orc_struct so = st;   /* the structure as seen from observation i (oi.cpp:304-310) */
            if(o_h) { so.h = o_h[si]; so.v = o_v[si]; so.w = o_w[si]; so.loc = o_R[si]; }
            for(int j = 0; j < lS; j++) {
                int sj = work[j].idx;
                A[i * lS + j] = (double)orc_corr_g(&so, ox[si], oy[si], oz[si], oelev[si], olaf[si],
                                                  ox[sj], oy[sj], oz[sj], oelev[sj], olaf[sj], 0);
            }
            A[i * lS + i] += (double)pratios[si];
        }
        if(orc_inv(A, lS) != ORC_OK) { rc = ORC_ESINGULAR; free(A); free(d); free(G); break; }
        double dx = 0, a00 = 0; float maxInc = 0, minInc = 0;
        for(int j = 0; j < lS; j++) {
            double gsr = 0;
            for(int i = 0; i < lS; i++) gsr += G[i] * A[i * lS + j];
            dx += gsr * d[j];
            a00 += gsr * G[j];
        }
        for(int j = 0; j < lS; j++) {
            float dj = (float)d[j];
            if(j == 0 || dj > maxInc) maxInc = dj;
            if(j == 0 || dj < minInc) minInc = dj;
        }
        float increment = (float)dx;
        if(!allow_extrapolation) {
            if(maxInc > 0 && increment > maxInc) increment = maxInc;
            else if(maxInc < 0 && increment > 0) increment = maxInc;
            else if(minInc < 0 && increment < minInc) increment = minInc;
            else if(minInc > 0 && increment < 0) increment = minInc;
        }
        out[y] = background[y] + increment;
        out_var[y] = (float)((double)bvariance[y] * (1 - a00));
        free(A); free(d); free(G);
    }
    free(pratios); free(work);
    return rc;
}

/* Diagnostic for parity tests: the selected observation indices of one cell
 * (in selection order) and the gap between the last kept and first dropped rho
 * (0 => the top-max_points cut straddles an exact float tie, where the
 * reference's own result is implementation-defined, oi.cpp:266). */
int orc_oi_selection(float gx, float gy, float gz, float ge, float gl,
                     int nS, const float* ox, const float* oy, const float* oz, const float* oe, const float* ol,
                     const float* pobs, const float* pbg, float h, float v, float w, float min_rho,
                     int max_points, int* sel, int* tie_at_cut) {
    float loc = orc_barnes_localization_distance(h, min_rho);
    orc_pair* work = (orc_pair*)malloc(sizeof(orc_pair) * (nS + 1));
    float* srho = (float*)malloc(sizeof(float) * (nS + 1));
    int* all = (int*)malloc(sizeof(int) * (nS + 1));
    int n = orc_select(gx, gy, gz, ge, gl, nS, ox, oy, oz, oe, ol, pobs, pbg, h, v, w, loc, 0, work, all, srho);
    *tie_at_cut = 0;
    if(max_points > 0 && n > max_points) {
        qsort(work, n, sizeof(orc_pair), orc_pair_cmp);
        if(work[max_points - 1].rho == work[max_points].rho) *tie_at_cut = 1;
        n = max_points;
    }
    for(int i = 0; i < n; i++) sel[i] = work[i].idx;
    free(work); free(srho); free(all);
    return n;
}

/* ------------------------------------------------------------------------ */
/* statistics: src/api/util.cpp:19-178                                        */
/* ------------------------------------------------------------------------ */
enum { ST_MEAN = 0, ST_MIN = 10, ST_MEDIAN = 20, ST_MAX = 30, ST_QUANTILE = 40, ST_STD = 50,
       ST_VARIANCE = 60, ST_SUM = 70, ST_COUNT = 80, ST_RANDOMCHOICE = 90 };

static int orc_fcmp(const void* a, const void* b) {
    float x = *(const float*)a, y = *(const float*)b;
    return (x > y) - (x < y);
}
/* util.cpp:111-178; returns ORC_EINVAL via *err for q outside [0,1] */
float orc_calc_quantile(const float* array, int T, float quantile, int* err) {
    if(err) *err = ORC_OK;
    if(quantile < 0 || quantile > 1) { if(err) *err = ORC_EINVAL; return NAN; }
    if(!orc_valid(quantile)) return NAN;
    if(T == 0) return NAN;
    if(quantile == 0 || quantile == 1) {
        float m = NAN;
        for(int i = 0; i < T; i++) {
            float val = array[i];
            if(!orc_valid(val)) continue;
            else if(!orc_valid(m)) m = val;
            else if(quantile == 0 ? (val < m) : (val > m)) m = val;
        }
        return m;
    }
    float* clean = (float*)malloc(sizeof(float) * T);
    int N = 0;
    for(int i = 0; i < T; i++) if(orc_valid(array[i])) clean[N++] = array[i];
    float value = NAN;
    if(N > 0) {
        qsort(clean, N, sizeof(float), orc_fcmp);
        /* quantile * (N-1): float * int -> float, then floor/ceil (util.cpp:160-161) */
        float pos = quantile * (N - 1);
        int lowerIndex = (int)floorf(pos);
        int upperIndex = (int)ceilf(pos);
        float lowerQuantile = (float)lowerIndex / (N - 1);
        float upperQuantile = (float)upperIndex / (N - 1);
        float lowerValue = clean[lowerIndex], upperValue = clean[upperIndex];
        if(lowerIndex == upperIndex) value = lowerValue;
        else {
            float f = (quantile - lowerQuantile) / (upperQuantile - lowerQuantile);
            value = lowerValue + (upperValue - lowerValue) * f;
        }
    }
    free(clean);
    return value;
}
/* util.cpp:19-110 (RandomChoice omitted: depends on rand()) */
float orc_calc_statistic(const float* array, int n, int statistic) {
    float value = NAN;
    if(statistic == ST_MEAN || statistic == ST_SUM || statistic == ST_COUNT) {
        float total = 0; int count = 0;
        for(int i = 0; i < n; i++) if(orc_valid(array[i])) { total += array[i]; count++; }
        if(statistic == ST_COUNT) value = count;
        else if(count > 0) value = (statistic == ST_MEAN) ? total / count : total;
    }
    else if(statistic == ST_STD || statistic == ST_VARIANCE) {
        float total = 0, total2 = 0, K = NAN; int count = 0;
        for(int i = 0; i < n; i++) if(orc_valid(array[i])) {
            if(!orc_valid(K)) K = array[i];
            total += array[i] - K;
            total2 += (array[i] - K) * (array[i] - K);
            count++;
        }
        if(count > 0) {
            float mean = total / count, mean2 = total2 / count;
            float var = mean2 - mean * mean;
            if(var < 0) var = 0;
            value = (statistic == ST_STD) ? sqrtf(var) : var;
        }
    }
    else {
        float q = (statistic == ST_MIN) ? 0 : (statistic == ST_MEDIAN) ? 0.5f : (statistic == ST_MAX) ? 1 : NAN;
        if(isnan(q)) return NAN;
        value = orc_calc_quantile(array, n, q, NULL);
    }
    return value;
}

/* util.cpp:339-414 */
static int orc_lower_index(float x, const float* v, int n) {
    int index = -1;
    for(int i = 0; i < n; i++) {
        float c = v[i];
        if(orc_valid(c)) {
            if(c < x) index = i;
            else if(c == x) { index = i; break; }
            else if(c > x) break;
        }
    }
    return index;
}
static int orc_upper_index(float x, const float* v, int n) {
    int index = -1;
    for(int i = n - 1; i >= 0; i--) {
        float c = v[i];
        if(orc_valid(c)) {
            if(c > x) index = i;
            else if(c == x) { index = i; break; }
            else if(c < x) break;
        }
    }
    return index;
}
float orc_interpolate(float x, const float* iX, const float* iY, int n) {
    if(!orc_valid(x)) return NAN;
    if(n == 0) return NAN;
    if(x > iX[n - 1]) return iY[n - 1];
    if(x < iX[0]) return iY[0];
    int i0 = orc_lower_index(x, iX, n), i1 = orc_upper_index(x, iX, n);
    float x0 = iX[i0], x1 = iX[i1], y0 = iY[i0], y1 = iY[i1];
    if(x0 == x1) {
        if(i0 == 0 && i1 == n - 1) return (y0 + y1) / 2;
        else if(i0 == 0) return y1;
        else if(i1 == n - 1) return y0;
        else return (y0 + y1) / 2;
    }
    return y0 + (y1 - y0) * (x - x0) / (x1 - x0);
}

/* util.cpp:261-338; values need not be sorted; returns count written to out (<= num) */
int orc_calc_even_quantiles(const float* values, int size, int num, float* out) {
    int nq = 0;
    if(num == 0 || size == 0) return 0;
    float* sorted = (float*)malloc(sizeof(float) * size);
    memcpy(sorted, values, sizeof(float) * size);
    qsort(sorted, size, sizeof(float), orc_fcmp);
    if(num >= size) {
        out[nq++] = sorted[0];
        for(int i = 1; i < size; i++) if(sorted[i] != sorted[i - 1]) out[nq++] = sorted[i];
        free(sorted); return nq;
    }
    float lowest = sorted[0], highest = sorted[size - 1];
    int count_lower = 0;
    for(int i = 0; i < size; i++) { if(sorted[i] != lowest) break; count_lower++; }
    out[nq++] = lowest;
    if(num == 2) { if(lowest != highest) out[nq++] = highest; free(sorted); return nq; }
    int repeated = count_lower < size && count_lower > size / num;
    if(repeated) out[nq++] = sorted[count_lower];
    float last_added = out[nq - 1];
    float* uniq = (float*)malloc(sizeof(float) * size);
    int nu = 0;
    for(int i = 0; i < size; i++)
        if(sorted[i] > last_added && (nu == 0 || sorted[i] != uniq[nu - 1])) uniq[nu++] = sorted[i];
    if(nu > 0) {
        int num_left = num - nq;
        for(int i = 1; i <= num_left; i++) {
            float f = (float)i / num_left;
            int index = (int)(nu * f - 1);   /* size_t * float -> float, -1, truncation (util.cpp:322) */
            if(index >= 0) out[nq++] = uniq[index];
        }
    }
    free(sorted); free(uniq);
    return nq;
}
/* neighbourhood.cpp:243-295: valid values of a 2-D/3-D field (flattened) */
int orc_get_neighbourhood_thresholds(const float* input, long n, int num_thresholds, float* out) {
    if(num_thresholds <= 0) return ORC_EINVAL;
    float* all = (float*)malloc(sizeof(float) * (n > 0 ? n : 1));
    int c = 0;
    for(long i = 0; i < n; i++) if(orc_valid(input[i])) all[c++] = input[i];
    int r = orc_calc_even_quantiles(all, c, num_thresholds, out);
    free(all);
    return r;
}

/* ------------------------------------------------------------------------ */
/* neighbourhood (2-D): src/api/neighbourhood.cpp:28-242                      */
/* ------------------------------------------------------------------------ */
static int orc_nbh_brute(const float* in, int nY, int nX, int nE, int hw, int statistic, float quantile, float* out);

int orc_neighbourhood(const float* in, int nY, int nX, int hw, int statistic, float* out) {
    if(hw < 0) return ORC_EINVAL;
    if(statistic == ST_QUANTILE) return ORC_EINVAL;
    if(nY == 0 || nX == 0) return ORC_OK;
    for(long i = 0; i < (long)nY * nX; i++) out[i] = NAN;
    if(statistic == ST_MEAN || statistic == ST_SUM || statistic == ST_COUNT) {
        /* summed-area table, double values + int counts (:45-99) */
        double* values = (double*)calloc((size_t)nY * nX, sizeof(double));
        int* counts = (int*)calloc((size_t)nY * nX, sizeof(int));
#define V(i,j) values[(size_t)(i) * nX + (j)]
#define Cn(i,j) counts[(size_t)(i) * nX + (j)]
        for(int i = 0; i < nY; i++) for(int j = 0; j < nX; j++) {
            float value = in[(size_t)i * nX + j];
            int ok = orc_valid(value);
            if(j == 0 && i == 0) { if(ok) { V(i,j) = value; Cn(i,j) = 1; } }
            else if(j == 0) { V(i,j) = ok ? V(i-1,j) + value : V(i-1,j); Cn(i,j) = Cn(i-1,j) + ok; }
            else if(i == 0) { V(i,j) = ok ? V(i,j-1) + value : V(i,j-1); Cn(i,j) = Cn(i,j-1) + ok; }
            else {
                V(i,j) = ok ? V(i,j-1) + V(i-1,j) - V(i-1,j-1) + value : V(i,j-1) + V(i-1,j) - V(i-1,j-1);
                Cn(i,j) = Cn(i,j-1) + Cn(i-1,j) - Cn(i-1,j-1) + ok;
            }
        }
#pragma omp parallel for num_threads(orc_nbh_threads) if(orc_nbh_threads > 1)                                        /* :101 */
        for(int i = 0; i < nY; i++) for(int j = 0; j < nX; j++) {        /* :101-144 */
            int i1 = i + hw < nY - 1 ? i + hw : nY - 1;
            int j1 = j + hw < nX - 1 ? j + hw : nX - 1;
            int i0 = i - hw - 1, j0 = j - hw - 1;
            double v11 = V(i1,j1), v00 = 0, v10 = 0, v01 = 0;
            int c11 = Cn(i1,j1), c00 = 0, c10 = 0, c01 = 0;
            if(i0 >= 0 && j0 >= 0) { v00 = V(i0,j0); v10 = V(i1,j0); v01 = V(i0,j1); c00 = Cn(i0,j0); c10 = Cn(i1,j0); c01 = Cn(i0,j1); }
            else if(j0 >= 0) { v10 = V(i1,j0); c10 = Cn(i1,j0); }
            else if(i0 >= 0) { v01 = V(i0,j1); c01 = Cn(i0,j1); }
            double value = v11 + v00 - v10 - v01;
            int count = c11 + c00 - c10 - c01;
            float* o = &out[(size_t)i * nX + j];
            if(statistic == ST_COUNT) *o = count;
            else if(count > 0) { if(statistic == ST_MEAN) value /= count; *o = (float)value; }
        }
#undef V
#undef Cn
        free(values); free(counts);
    }
    else if(statistic == ST_MIN || statistic == ST_MAX) {
        /* :146-210 -- the edge rows use the full window, interior rows the
         * column-sliver then row pass; both equal the window min/max of the
         * valid values, which is what is computed here directly. */
        for(int i = 0; i < nY; i++) for(int j = 0; j < nX; j++) {
            float m = NAN;
            int ia = i - hw > 0 ? i - hw : 0, ib = i + hw < nY - 1 ? i + hw : nY - 1;
            int ja = j - hw > 0 ? j - hw : 0, jb = j + hw < nX - 1 ? j + hw : nX - 1;
            for(int ii = ia; ii <= ib; ii++) for(int jj = ja; jj <= jb; jj++) {
                float val = in[(size_t)ii * nX + jj];
                if(!orc_valid(val)) continue;
                if(!orc_valid(m)) m = val;
                else if(statistic == ST_MIN ? val < m : val > m) m = val;
            }
            out[(size_t)i * nX + j] = m;
        }
    }
    else if(statistic == ST_STD || statistic == ST_VARIANCE) {           /* :211-235 */
        float* mean = (float*)malloc(sizeof(float) * nY * nX);
        float* mean2 = (float*)malloc(sizeof(float) * nY * nX);
        float* in2 = (float*)malloc(sizeof(float) * nY * nX);
        for(long i = 0; i < (long)nY * nX; i++) in2[i] = in[i] * in[i];
        orc_neighbourhood(in, nY, nX, hw, ST_MEAN, mean);
        orc_neighbourhood(in2, nY, nX, hw, ST_MEAN, mean2);
        for(long i = 0; i < (long)nY * nX; i++) {
            float var = mean2[i] - mean[i] * mean[i];
            out[i] = (statistic == ST_STD) ? sqrtf(var) : var;
        }
        free(mean); free(mean2); free(in2);
    }
    else return orc_nbh_brute(in, nY, nX, 1, hw, statistic, 0, out);     /* :236-238 */
    return ORC_OK;
}
/* neighbourhood.cpp:12-27: member statistic first, then the 2-D filter */
int orc_neighbourhood3(const float* in, int nY, int nX, int nE, int hw, int statistic, float* out) {
    if(nY == 0 || nX == 0) return ORC_OK;
    float* flat = (float*)malloc(sizeof(float) * nY * nX);
    for(long c = 0; c < (long)nY * nX; c++) flat[c] = orc_calc_statistic(in + c * nE, nE, statistic);
    int rc = orc_neighbourhood(flat, nY, nX, hw, statistic, out);
    free(flat);
    return rc;
}
/* neighbourhood.cpp:557-654 (brute force, statistic or exact quantile); nE=1 for 2-D */
static int orc_nbh_brute(const float* in, int nY, int nX, int nE, int hw, int statistic, float quantile, float* out) {
    if(hw < 0) return ORC_EINVAL;
    if(nY == 0 || nX == 0 || nE == 0) return ORC_OK;
    size_t cap = (size_t)(2 * hw + 1) * (2 * hw + 1) * nE;
    if(cap > (size_t)nY * nX * nE) cap = (size_t)nY * nX * nE;
    float* hood = (float*)malloc(sizeof(float) * cap);
    int rc = ORC_OK;
    for(int i = 0; i < nY && rc == ORC_OK; i++) for(int j = 0; j < nX; j++) {
        int ia = i - hw > 0 ? i - hw : 0, ib = i + hw < nY - 1 ? i + hw : nY - 1;
        int ja = j - hw > 0 ? j - hw : 0, jb = j + hw < nX - 1 ? j + hw : nX - 1;
        int n = 0;
        for(int ii = ia; ii <= ib; ii++) for(int jj = ja; jj <= jb; jj++) for(int e = 0; e < nE; e++)
            hood[n++] = in[((size_t)ii * nX + jj) * nE + e];
        if(statistic == ST_QUANTILE) {
            int err; out[(size_t)i * nX + j] = orc_calc_quantile(hood, n, quantile, &err);
            if(err != ORC_OK) { rc = err; break; }
        }
        else out[(size_t)i * nX + j] = orc_calc_statistic(hood, n, statistic);
    }
    free(hood);
    return rc;
}
int orc_neighbourhood_brute_force(const float* in, int nY, int nX, int nE, int hw, int statistic, float* out) {
    return orc_nbh_brute(in, nY, nX, nE, hw, statistic, 0, out);
}
int orc_neighbourhood_quantile(const float* in, int nY, int nX, int nE, float quantile, int hw, float* out) {
    return orc_nbh_brute(in, nY, nX, nE, hw, ST_QUANTILE, quantile, out);
}

/* neighbourhood.cpp:296-409 (2-D, nE==1 with is3d=0) and :411-527 (3-D).
 * quantile: nq==1 scalar, else nY*nX field. */
int orc_neighbourhood_quantile_fast(const float* in, int nY, int nX, int nE, int is3d,
                                    const float* quantile, int nq, int hw,
                                    const float* thresholds, int nT, float* out) {
    if(hw < 0) return ORC_EINVAL;
    if(nY == 0 || nX == 0 || nE == 0) return ORC_OK;
    if(!(nq == 1) && !(nq == nY * nX)) return ORC_EINVAL;
    for(int i = 0; i < nq; i++) if(orc_valid(quantile[i]) && (quantile[i] < 0 || quantile[i] > 1)) return ORC_EINVAL;
    size_t C = (size_t)nY * nX;
    for(size_t c = 0; c < C; c++) out[c] = NAN;
    if(nT == 0) return ORC_OK;
    float* stats = (float*)malloc(sizeof(float) * C * nT);
    /* :453 -- one thread per threshold in the reference (the neighbourhood() inside then runs serially: nested regions are off) */
#pragma omp parallel for num_threads(orc_nbh_threads) if(orc_nbh_threads > 1)
    for(int t = 0; t < nT; t++) {
        float* temp = (float*)malloc(sizeof(float) * C);
        for(size_t c = 0; c < C; c++) {
            int sum = 0, count = 0;
            for(int e = 0; e < nE; e++) {
                float val = in[c * nE + e];
                if(orc_valid(val)) { if(val <= thresholds[t]) sum++; count++; }
            }
            temp[c] = count > 0 ? (float)sum / count : NAN;
        }
        orc_neighbourhood(temp, nY, nX, hw, ST_MEAN, stats + (size_t)t * C);
        free(temp);
    }
    /* :483 -- rows in parallel */
#pragma omp parallel for num_threads(orc_nbh_threads) if(orc_nbh_threads > 1)
    for(int yrow = 0; yrow < nY; yrow++) {
    float* yarray = (float*)malloc(sizeof(float) * nT);
    for(size_t c = (size_t)yrow * nX; c < (size_t)(yrow + 1) * nX; c++) {
        float q = (nq == 1) ? quantile[0] : quantile[c];
        int missing = 0;
        for(int t = 0; t < nT; t++) {
            float s = stats[(size_t)t * C + c];
            float sum = 0; int count = 0;
            if(is3d) { for(int e = 0; e < nE; e++) if(orc_valid(s)) { sum += s; count++; } } /* :494-499 */
            else if(orc_valid(s)) { sum = s; count = 1; }                                   /* :378-381 */
            if(count > 0) {
                float yv = sum / count;
                if(yv > 1) yv = 1; else if(yv < 0) yv = 0;
                yarray[t] = yv;
            }
            else { yarray[t] = NAN; missing = 1; }
        }
        if(!missing) {
            if(q == 1 && yarray[0] == 1) out[c] = thresholds[0];
            else if(q == 0 && yarray[nT - 1] == 0) out[c] = thresholds[nT - 1];
            else out[c] = orc_interpolate(q, yarray, thresholds, nT);
        }
    }
    free(yarray);
    }
    free(stats);
    return ORC_OK;
}

/* ------------------------------------------------------------------------ */
/* nearest(Grid, Points, vec2): src/api/nearest.cpp:124-144                  */
/* ------------------------------------------------------------------------ */
int orc_nearest(const float* gx, const float* gy, const float* gz, int nG, const float* values,
                const float* qx, const float* qy, const float* qz, int nQ, float* out, int* out_idx) {
    for(int i = 0; i < nQ; i++) {
        if(nG == 0) { out[i] = NAN; if(out_idx) out_idx[i] = -1; continue; }
        int idx = orc_nearest_neighbour(gx, gy, gz, nG, qx[i], qy[i], qz[i], 1);
        out[i] = values[idx];
        if(out_idx) out_idx[i] = idx;
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------------ */
/* count / gridding / gridding_nearest: src/api/count.cpp:6-66,              */
/* src/api/gridding.cpp:6-131                                               */
/* ------------------------------------------------------------------------ */
/* count: number of points of the input set within `radius` of every output location (get_num_neighbours with
 * include_match = true, src/api/kdtree.cpp:61-64) */
int orc_count(const float* px, const float* py, const float* pz, int n, const float* qx, const float* qy, const float* qz,
              int nq, float radius, float* out) {
    for(int i = 0; i < nq; i++) {
        int c = 0;
        for(int j = 0; j < n; j++) c += orc_in_radius(qx[i], qy[i], qz[i], px[j], py[j], pz[j], radius, 1);
        out[i] = c;
    }
    return ORC_OK;
}
/* gridding.cpp:6-63: statistic of the values of the input points within `radius` of every output location, MV when
 * fewer than min_num (if min_num > 0).  Neighbours are taken in index order (the R-tree's order is unspecified). */
int orc_gridding(const float* px, const float* py, const float* pz, const float* values, int n, const float* qx,
                 const float* qy, const float* qz, int nq, float radius, int min_num, int statistic, float* out) {
    if(!orc_valid(radius) || radius < 0 || min_num < 0) return ORC_EINVAL;
    float* curr = (float*)malloc(sizeof(float) * (n > 0 ? n : 1));
    for(int i = 0; i < nq; i++) {
        int c = 0;
        for(int j = 0; j < n; j++)
            if(orc_in_radius(qx[i], qy[i], qz[i], px[j], py[j], pz[j], radius, 1)) curr[c++] = values[j];
        out[i] = (min_num <= 0 || c >= min_num) ? orc_calc_statistic(curr, c, statistic) : NAN;
    }
    free(curr);
    return ORC_OK;
}
/* gridding.cpp:65-131: every input point goes to its nearest output location; statistic of what each location received
 * (in input order), MV for locations that received nothing or fewer than min_num. */
int orc_gridding_nearest(const float* ox, const float* oy, const float* oz, int no, const float* px, const float* py,
                         const float* pz, const float* values, int n, int min_num, int statistic, float* out) {
    if(min_num < 0) return ORC_EINVAL;
    int* target = (int*)malloc(sizeof(int) * (n > 0 ? n : 1));
    float* curr = (float*)malloc(sizeof(float) * (n > 0 ? n : 1));
    for(int s = 0; s < n; s++) target[s] = orc_nearest_neighbour(ox, oy, oz, no, px[s], py[s], pz[s], 1);
    for(int i = 0; i < no; i++) {
        int c = 0;
        for(int s = 0; s < n; s++) if(target[s] == i) curr[c++] = values[s];
        out[i] = (c > 0 && (min_num <= 0 || c >= min_num)) ? orc_calc_statistic(curr, c, statistic) : NAN;
    }
    free(target); free(curr);
    return ORC_OK;
}

/* ------------------------------------------------------------------------ */
/* staticcorr_points: src/api/corr_points.cpp:26-131                          */
/* ------------------------------------------------------------------------ */
/* out [nY][nS], zero except at the knots a point keeps: those inside its localization radius with
 * corr_background > 0, the max_points largest of them if there are more (ties -> lower knot index; the reference's
 * std::sort on rho alone leaves ties unspecified).  NO test of the reference covers this function: parity unpinned. */
int orc_staticcorr_points(int nY, const float* gx, const float* gy, const float* gz, const float* gelev, const float* glaf,
                          int nS, const float* ox, const float* oy, const float* oz, const float* oelev, const float* olaf,
                          int kh, int kv, int kw, float h, float v, float w, float loc, int cv, float cv_dist,
                          int max_points, float* out) {
    if(max_points < 0) return ORC_EINVAL;
    orc_struct st = {kh, kv, kw, h, v, w, loc, cv, cv_dist};
    for(size_t i = 0; i < (size_t)nY * nS; i++) out[i] = 0;
    orc_pair* work = (orc_pair*)malloc(sizeof(orc_pair) * (nS > 0 ? nS : 1));
    for(int y = 0; y < nY; y++) {
        int c = 0;
        for(int j = 0; j < nS; j++) {
            if(!orc_in_radius(gx[y], gy[y], gz[y], ox[j], oy[j], oz[j], loc, 1)) continue;
            float rho = orc_corr_g(&st, gx[y], gy[y], gz[y], gelev[y], glaf[y], ox[j], oy[j], oz[j], oelev[j], olaf[j], 1);
            if(rho > 0) { work[c].rho = rho; work[c].idx = j; c++; }
        }
        int keep = c;
        if(max_points > 0 && c > max_points) { qsort(work, c, sizeof(orc_pair), orc_pair_cmp); keep = max_points; }
        for(int i = 0; i < keep; i++) out[(size_t)y * nS + work[i].idx] = work[i].rho;
    }
    free(work);
    return ORC_OK;
}

/* ------------------------------------------------------------------------ */
/* distance: src/api/distance.cpp:6-120                                      */
/* ------------------------------------------------------------------------ */
/* out[i] = the largest calc_distance (kdtree.cpp:107-133) from output location i to its `num` nearest input points
 * (nearest by float32 squared chord, ties -> lower index).  query_first: argument order of calc_distance -- the
 * Grid->Points and Points->Points overloads pass (location, neighbour), the two ->Grid overloads (neighbour, location). */
int orc_distance(const float* px, const float* py, const float* pz, const float* plat, const float* plon, int n,
                 const float* qx, const float* qy, const float* qz, const float* qlat, const float* qlon, int nq, int num, int type,
                 int query_first, float* out) {
    float* key = (float*)malloc(sizeof(float) * (n > 0 ? n : 1));
    char* taken = (char*)malloc(n > 0 ? n : 1);
    for(int i = 0; i < nq; i++) {
        for(int j = 0; j < n; j++) {
            float dx = px[j] - qx[i], dy = py[j] - qy[i], dz = pz[j] - qz[i];
            key[j] = dx * dx + dy * dy + dz * dz;
            taken[j] = 0;
        }
        float max_dist = 0;
        for(int k = 0; k < num && k < n; k++) {
            int best = -1;
            for(int j = 0; j < n; j++) if(!taken[j] && (best < 0 || key[j] < key[best])) best = j;
            taken[best] = 1;
            float d = query_first ? orc_calc_distance(qlat[i], qlon[i], plat[best], plon[best], type)
                                  : orc_calc_distance(plat[best], plon[best], qlat[i], qlon[i], type);
            if(d > max_dist) max_dist = d;
        }
        out[i] = max_dist;
    }
    free(key); free(taken);
    return ORC_OK;
}

/* ------------------------------------------------------------------------ */
/* fill / fill_missing (src/api/fill.cpp:6-134), doping_square / doping_circle */
/* (src/api/doping.cpp:5-93), neighbourhood_search                             */
/* (src/api/neighbourhood_search.cpp:7-113), calc_gradient                     */
/* (src/api/calc_gradient.cpp:7-126)                                           */
/* ------------------------------------------------------------------------ */
/* fill.cpp:6-41: grid cells within radii[i] of point i get `value` (outside = 0), or keep the input while all the
 * others get `value` (outside = 1).  gx.. = the grid's search coordinates, px.. the points'. */
int orc_fill(const float* gx, const float* gy, const float* gz, int nG, const float* input, const float* px,
             const float* py, const float* pz, const float* radii, int nP, float value, int outside, float* out) {
    for(int i = 0; i < nP; i++) if(radii[i] < 0) return ORC_EINVAL;
    for(int c = 0; c < nG; c++) out[c] = outside ? value : input[c];
    for(int i = 0; i < nP; i++)
        for(int c = 0; c < nG; c++)
            if(orc_in_radius(px[i], py[i], pz[i], gx[c], gy[c], gz[c], radii[i], 1)) out[c] = outside ? input[c] : value;
    return ORC_OK;
}
/* fill.cpp:43-134: linear interpolation across runs of missing values along rows and along columns, averaged */
int orc_fill_missing(const float* values, int Y, int X, float* out) {
    float* ry = (float*)malloc(sizeof(float) * Y * X);
    float* rx = (float*)malloc(sizeof(float) * Y * X);
    for(int i = 0; i < Y * X; i++) ry[i] = rx[i] = NAN;
    for(int y = 0; y < Y; y++) {
        int last = 0, next = -1;
        for(int x = 0; x < X; x++) {
            float curr = values[y * X + x];
            if(!orc_valid(curr)) {
                if(next < x) for(next = x; next < X; next++) if(orc_valid(values[y * X + next])) break;
                if(next >= X) continue;
                float vl = values[y * X + last], vn = values[y * X + next];
                ry[y * X + x] = (vl) + (vn - vl) * (x - last) / (next - last);
            }
            else { last = x; ry[y * X + x] = curr; }
        }
    }
    for(int x = 0; x < X; x++) {
        int last = 0, next = -1;
        for(int y = 0; y < Y; y++) {
            float curr = values[y * X + x];
            if(!orc_valid(curr)) {
                if(next < y) for(next = y; next < Y; next++) if(orc_valid(values[next * X + x])) break;
                if(next >= Y) continue;
                float vl = values[last * X + x], vn = values[next * X + x];
                rx[y * X + x] = (vl) + (vn - vl) * (y - last) / (next - last);
            }
            else { last = y; rx[y * X + x] = curr; }
        }
    }
    for(int i = 0; i < Y * X; i++) {
        int count = 0; float total = 0;
        if(orc_valid(ry[i])) { total += ry[i]; count++; }
        if(orc_valid(rx[i])) { total += rx[i]; count++; }
        out[i] = count > 0 ? total / count : NAN;
    }
    free(ry); free(rx);
    return ORC_OK;
}
/* doping.cpp:5-48: a (2 hw + 1)^2 window of grid cells around the nearest grid point of every observation takes its
 * value (later observations overwrite earlier ones); cells whose elevation differs by more than max_elev_diff are left */
int orc_doping_square(const float* gx, const float* gy, const float* gz, const float* gelev, int Y, int X,
                      const float* background, const float* px, const float* py, const float* pz, const float* pelev,
                      const float* obs, const int* halfwidth, int nP, float max_elev_diff, float* out) {
    if(orc_valid(max_elev_diff) && max_elev_diff < 0) return ORC_EINVAL;
    for(int i = 0; i < nP; i++) if(halfwidth[i] < 0) return ORC_EINVAL;
    for(int c = 0; c < Y * X; c++) out[c] = background[c];
    int check = orc_valid(max_elev_diff);
    for(int i = 0; i < nP; i++) {
        int nn = orc_nearest_neighbour(gx, gy, gz, Y * X, px[i], py[i], pz[i], 1);
        int iy = nn / X, ix = nn % X;
        for(int yy = (iy - halfwidth[i] > 0 ? iy - halfwidth[i] : 0); yy <= (iy + halfwidth[i] < Y - 1 ? iy + halfwidth[i] : Y - 1); yy++)
            for(int xx = (ix - halfwidth[i] > 0 ? ix - halfwidth[i] : 0); xx <= (ix + halfwidth[i] < X - 1 ? ix + halfwidth[i] : X - 1); xx++) {
                if(check && fabsf(pelev[i] - gelev[yy * X + xx]) > max_elev_diff) continue;
                out[yy * X + xx] = obs[i];
            }
    }
    return ORC_OK;
}
/* doping.cpp:50-93: the same with a radius per observation */
int orc_doping_circle(const float* gx, const float* gy, const float* gz, const float* gelev, int nG,
                      const float* background, const float* px, const float* py, const float* pz, const float* pelev,
                      const float* obs, const float* radii, int nP, float max_elev_diff, float* out) {
    if(orc_valid(max_elev_diff) && max_elev_diff < 0) return ORC_EINVAL;
    for(int i = 0; i < nP; i++) if(radii[i] < 0) return ORC_EINVAL;
    for(int c = 0; c < nG; c++) out[c] = background[c];
    int check = orc_valid(max_elev_diff);
    for(int i = 0; i < nP; i++)
        for(int c = 0; c < nG; c++) {
            if(!orc_in_radius(px[i], py[i], pz[i], gx[c], gy[c], gz[c], radii[i], 1)) continue;
            if(check && fabsf(pelev[i] - gelev[c]) > max_elev_diff) continue;
            out[c] = obs[i];
        }
    return ORC_OK;
}
/* neighbourhood_search.cpp:7-113; apply may be NULL (no apply array) */
int orc_neighbourhood_search(const float* array, const float* search, int nY, int nX, int halfwidth, float tmin, float tmax,
                             float delta, const int* apply, float* out) {
    if(tmin > tmax || halfwidth < 0) return ORC_EINVAL;
    for(int y = 0; y < nY; y++) for(int x = 0; x < nX; x++) {
        const int c = y * nX + x;
        float nearest_target = NAN;
        int ny_ = 0, nx_ = 0, counter = 0;
        float accum = 0;
        if(!orc_valid(search[c])) { out[c] = array[c]; continue; }
        if(apply && apply[c] == 0) { out[c] = array[c]; continue; }
        for(int yy = (y - halfwidth > 0 ? y - halfwidth : 0); yy <= (y + halfwidth < nY - 1 ? y + halfwidth : nY - 1); yy++)
            for(int xx = (x - halfwidth > 0 ? x - halfwidth : 0); xx <= (x + halfwidth < nX - 1 ? x + halfwidth : nX - 1); xx++) {
                const int n = yy * nX + xx;
                if(!orc_valid(search[n]) || !orc_valid(array[n])) continue;
                if(!apply || apply[c] == 1) {
                    if(search[n] >= tmin && search[n] <= tmax) { counter++; accum = accum + array[n]; }
                    else if(counter > 0) continue;
                    else if(fabsf(search[n] - search[c]) >= delta) {
                        if(!orc_valid(nearest_target)) { nearest_target = search[n]; ny_ = yy; nx_ = xx; }
                        else {
                            float cur = fminf(fabsf(search[n] - tmin), fabsf(search[n] - tmax));
                            float best = fminf(fabsf(nearest_target - tmin), fabsf(nearest_target - tmax));
                            if(cur < best) { nearest_target = search[n]; ny_ = yy; nx_ = xx; }
                        }
                    }
                }
            }
        if(counter > 0) out[c] = accum / counter;
        else if(orc_valid(nearest_target)) out[c] = array[ny_ * nX + nx_];
        else out[c] = array[c];
    }
    return ORC_OK;
}
/* calc_gradient.cpp:7-126; gradient_type 0 = MinMax, 10 = LinearRegression.  (MinMax's range test is written with an
 * unqualified abs() in the reference; it is taken as the float absolute value here.) */
int orc_calc_gradient(const float* base, const float* values, int nY, int nX, int gradient_type, int halfwidth, int num_min,
                      float min_range, float default_gradient, float* out) {
    if(halfwidth <= 0 || (orc_valid(min_range) && min_range < 0) || num_min < 0 || nY == 0) return ORC_EINVAL;
    const int n = nY * nX;
    for(int i = 0; i < n; i++) out[i] = default_gradient;
    if(gradient_type == 0) {
        for(int y = 0; y < nY; y++) for(int x = 0; x < nX; x++) {
            float cmax = NAN, cmin = NAN;
            int imax = 0, imin = 0, count = 0;
            for(int yy = (y - halfwidth > 0 ? y - halfwidth : 0); yy <= (y + halfwidth < nY - 1 ? y + halfwidth : nY - 1); yy++)
                for(int xx = (x - halfwidth > 0 ? x - halfwidth : 0); xx <= (x + halfwidth < nX - 1 ? x + halfwidth : nX - 1); xx++) {
                    float b = base[yy * nX + xx];
                    if(!orc_valid(b) || !orc_valid(values[yy * nX + xx])) continue;
                    if(!orc_valid(cmax) || b > cmax) { cmax = b; imax = yy * nX + xx; }
                    if(!orc_valid(cmin) || b < cmin) { cmin = b; imin = yy * nX + xx; }
                    count++;
                }
            if(count < num_min || !orc_valid(cmax) || !orc_valid(cmin) || fabsf(cmax - cmin) <= min_range) continue;
            out[y * nX + x] = (values[imax] - values[imin]) / (cmax - cmin);
        }
        return ORC_OK;
    }
    float *b0 = (float*)calloc((size_t)n * 10, sizeof(float)), *v0 = b0 + n, *bb = b0 + 2 * n, *bv = b0 + 3 * n, *ok = b0 + 4 * n;
    float *mX = b0 + 5 * n, *mY = b0 + 6 * n, *mXX = b0 + 7 * n, *mXY = b0 + 8 * n, *cnt = b0 + 9 * n;
    for(int i = 0; i < n; i++) {
        b0[i] = v0[i] = bb[i] = bv[i] = NAN; ok[i] = 0;
        if(orc_valid(base[i]) && orc_valid(values[i])) {
            bb[i] = (float)pow(base[i], 2); bv[i] = base[i] * values[i]; ok[i] = 1; b0[i] = base[i]; v0[i] = values[i];
        }
    }
    orc_neighbourhood(b0, nY, nX, halfwidth, ST_MEAN, mX);
    orc_neighbourhood(v0, nY, nX, halfwidth, ST_MEAN, mY);
    orc_neighbourhood(bb, nY, nX, halfwidth, ST_MEAN, mXX);
    orc_neighbourhood(bv, nY, nX, halfwidth, ST_MEAN, mXY);
    orc_neighbourhood(ok, nY, nX, halfwidth, ST_SUM, cnt);
    for(int i = 0; i < n; i++) {
        if(cnt[i] >= num_min && orc_valid(mXX[i]) && orc_valid(mXY[i]) && orc_valid(mX[i]) && mXX[i] - mX[i] * mX[i] != 0) {
            int valid_range = 1;
            if(orc_valid(min_range)) {
                float range = sqrtf(mXX[i] - mX[i] * mX[i]);
                if(!orc_valid(range) || range < min_range) valid_range = 0;
            }
            if(valid_range) out[i] = (mXY[i] - mX[i] * mY[i]) / (mXX[i] - mX[i] * mX[i]);
        }
    }
    free(b0);
    return ORC_OK;
}

/* ------------------------------------------------------------------------ */
/* bilinear(Grid, Points|Grid, vec2|vec3): src/api/bilinear.cpp:26-135        */
/* ------------------------------------------------------------------------ */
/* src/api/util.cpp:561-582: signed line tests of m against the edges AB, AD, BC, CD (float arithmetic); a point is
 * inside when the four signs fit a clockwise or a counter-clockwise quadrilateral. */
static float orc_edge_side(float plat, float plon, float qlat, float qlon, float mlat, float mlon) {
    float vlon = qlon - plon;            /* vect2d: lon = dlon, lat = -dlat */
    float vlat = -1 * (qlat - plat);
    float c = -1 * (vlat * plon + vlon * plat);
    return (vlat * mlon + vlon * mlat) + c;
}
int orc_point_in_rectangle(float alat, float alon, float blat, float blon, float clat, float clon,
                           float dlat, float dlon, float mlat, float mlon) {
    float d1 = orc_edge_side(alat, alon, blat, blon, mlat, mlon);
    float d2 = orc_edge_side(alat, alon, dlat, dlon, mlat, mlon);
    float d3 = orc_edge_side(blat, blon, clat, clon, mlat, mlon);
    float d4 = orc_edge_side(clat, clon, dlat, dlon, mlat, mlon);
    int cw = 0 >= d1 && 0 >= d4 && 0 <= d2 && 0 >= d3;
    int ccw = 0 <= d1 && 0 <= d4 && 0 >= d2 && 0 <= d3;
    return cw || ccw;
}

/* src/api/grid.cpp:149-229: the grid box around (lat, lon), searched in the four quadrants of the nearest grid
 * point in the order (+1,-1), (+1,+1), (-1,-1), (-1,+1) = (ydir, xdir).  nn = flat index of the nearest point. */
int orc_get_box(const float* glat, const float* glon, int nY, int nX, int nn, float lat, float lon,
                int* Y1, int* X1, int* Y2, int* X2) {
    *Y1 = *Y2 = *X1 = *X2 = -1;
    if(nn < 0 || nX <= 1 || nY <= 1) return 0;
    int Y = nn / nX, X = nn % nX;
    for(int it = 0; it < 4; it++) {
        int xdir = -1 + 2 * (it % 2);
        int ydir = -1 + 2 * (it < 2);
        if((Y == 0 && ydir == -1) || (Y == nY - 1 && ydir == 1) || (X == 0 && xdir == -1) || (X == nX - 1 && xdir == 1))
            continue;
        int a = Y * nX + X, b = (Y + ydir) * nX + X, c = (Y + ydir) * nX + X + xdir, d = Y * nX + X + xdir;
        if(orc_point_in_rectangle(glat[a], glon[a], glat[b], glon[b], glat[c], glon[c], glat[d], glon[d], lat, lon)) {
            *X1 = xdir == 1 ? X : X - 1;
            *X2 = *X1 + 1;
            *Y1 = ydir == 1 ? Y : Y - 1;
            *Y2 = *Y1 + 1;
            return 1;
        }
    }
    return 0;
}

/* src/api/bilinear.cpp:154-157 */
static int orc_bl_in_range(float v) {
    float tol = 0.01;
    return v >= -tol && v < 1 + tol;
}

/* src/api/bilinear.cpp:138-153 (parallelogram: 2x2 linear system, float) */
static void orc_bl_parallelogram(float x, float y, float X1, float X2, float X3, float Y1, float Y2, float Y3,
                                 float* t, float* s) {
    float A = X2 - X1, B = X3 - X1, C = Y2 - Y1, D = Y3 - Y1;
    float det = 1 / (A * D - B * C);
    *s = det * ((x - X1) * (D) + (y - Y1) * (-B));
    *t = det * ((x - X1) * (-C) + (y - Y1) * (A));
}

/* src/api/bilinear.cpp:159-267 (general quadrilateral: roots of the quadratic in double; coefficient differences
 * are formed in float first, as the reference's `double a = -x0 + x2` does) */
static void orc_bl_general(float x, float y, float x0, float x1, float x2, float x3, float y0, float y1, float y2,
                           float y3, float* t_out, float* s_out) {
    double a = -x0 + x2, b = -x0 + x1, c = x0 - x1 - x2 + x3, d = x - x0;
    double e = -y0 + y2, f = -y0 + y1, g = y0 - y1 - y2 + y3, h = y - y0;
    double alpha = NAN, beta = NAN;
    double Y1 = y1, Y2 = y3, Y3 = y0, Y4 = y2, X1 = x1, X2 = x3, X3 = x0, X4 = x2;
    double X31 = X3 - X1, X21 = X2 - X1, Y42 = Y4 - Y2, Y21 = Y2 - Y1, Y31 = Y3 - Y1, Y43 = Y4 - Y3, X42 = X4 - X2,
           X43 = X4 - X3;
    double qa = 2 * c * e - 2 * a * g, qb = 2 * c * f - 2 * b * g;
    double lin1 = b * e - a * f + d * g - c * h, lin2 = b * e - a * f - d * g + c * h;
    double root = sqrt(-4 * (c * e - a * g) * (d * f - b * h) + pow(lin1, 2));
    if(qa != 0 && qb != 0) {
        alpha = -(lin1 + root) / qa;
        beta = (lin2 + root) / qb;
        if(!orc_bl_in_range(alpha)) alpha = -(lin1 - root) / qa;
        if(!orc_bl_in_range(beta)) beta = (lin2 - root) / qb;
    }
    else if(qb == 0) {
        alpha = -(lin1 + root) / qa;
        if(!orc_bl_in_range(alpha)) alpha = -(lin1 - root) / qa;
        float s = alpha, t;
        if(Y3 + Y43 * s - Y1 - Y21 * s == 0) t = (x - X1 - X21 * s) / (X3 + X43 * s - X1 - X21 * s);
        else t = (y - Y1 - Y21 * s) / (Y3 + Y43 * s - Y1 - Y21 * s);
        beta = 1 - t;
    }
    else if(qa == 0) {
        beta = (lin2 + root) / qb;   /* the reference's retry uses the same root (:245-246) */
        float t = 1 - beta, s;
        if(Y2 + Y42 * t - Y1 - Y31 * t == 0) s = (x - X1 - X31 * t) / (X2 + X42 * t - X1 - X31 * t);
        else s = (y - Y1 - Y31 * t) / (Y2 + Y42 * t - Y1 - Y31 * t);
        alpha = s;
    }
    *s_out = alpha;
    *t_out = 1 - beta;
}

/* src/api/bilinear.cpp:269-320.  Returns 0, or 1 when s / t fall outside [0, 1] after the +-0.15 snap (the
 * reference throws std::runtime_error there); s and t are returned for the message. */
static int orc_bl_weights(float x, float y, float x0, float x1, float x2, float x3, float y0, float y1, float y2,
                          float y3, float* s_out, float* t_out) {
    float Y1 = y1, Y2 = y3, Y3 = y0, Y4 = y2, X1 = x1, X2 = x3, X3 = x0, X4 = x2;
    float s = NAN, t = NAN;
    int vertical = fabsf((X3 - X1) * (Y4 - Y2) - (X4 - X2) * (Y3 - Y1)) <= 1e-4;
    int horizontal = fabsf((X2 - X1) * (Y4 - Y3) - (X4 - X3) * (Y2 - Y1)) <= 1e-4;
    if(vertical && horizontal) orc_bl_parallelogram(x, y, X1, X2, X3, Y1, Y2, Y3, &t, &s);
    else orc_bl_general(x, y, x0, x1, x2, x3, y0, y1, y2, y3, &t, &s);
    if(t >= 1 && t <= 1.15) t = 1;
    if(t <= 0 && t >= -0.15) t = 0;
    if(s >= 1 && s <= 1.15) s = 1;
    if(s <= 0 && s >= -0.15) s = 0;
    *s_out = s; *t_out = t;
    return !(s >= 0 && s <= 1 && t >= 0 && t <= 1);
}

/* values [nT][nY][nX], out [nT][nQ]; glat/glon the grid's float32 lat/lon, gx/gy/gz its search coordinates.
 * Returns ORC_OK, or ORC_ESINGULAR when a box is too distorted (bad_s / bad_t receive the offending weights). */
int orc_bilinear(const float* glat, const float* glon, const float* gx, const float* gy, const float* gz, int nY, int nX,
                 const float* values, int nT, const float* qlat, const float* qlon, const float* qx, const float* qy,
                 const float* qz, int nQ, float* out, float* bad_s, float* bad_t) {
    int nG = nY * nX;
    for(int i = 0; i < nQ; i++) {
        if(nG == 0) { for(int k = 0; k < nT; k++) out[(size_t)k * nQ + i] = NAN; continue; }
        int nn = orc_nearest_neighbour(gx, gy, gz, nG, qx[i], qy[i], qz[i], 1);
        int I1, J1, I2, J2;
        int inside = orc_get_box(glat, glon, nY, nX, nn, qlat[i], qlon[i], &I1, &J1, &I2, &J2);
        for(int k = 0; k < nT; k++) {
            const float* v = values + (size_t)k * nG;
            float res = v[nn];
            if(inside) {
                float v0 = v[I1 * nX + J1], v1 = v[I2 * nX + J1], v2 = v[I1 * nX + J2], v3 = v[I2 * nX + J2];
                if(orc_valid(v0) && orc_valid(v1) && orc_valid(v2) && orc_valid(v3)) {
                    float s, t;
                    if(orc_bl_weights(qlon[i], qlat[i], glon[I1 * nX + J1], glon[I2 * nX + J1], glon[I1 * nX + J2],
                                      glon[I2 * nX + J2], glat[I1 * nX + J1], glat[I2 * nX + J1], glat[I1 * nX + J2],
                                      glat[I2 * nX + J2], &s, &t)) {
                        if(bad_s) *bad_s = s;
                        if(bad_t) *bad_t = t;
                        return ORC_ESINGULAR;
                    }
                    float P1 = v1, P2 = v3, P3 = v0, P4 = v2;
                    res = P1 * (1 - s) * (1 - t) + P2 * s * (1 - t) + P3 * (1 - s) * t + P4 * s * t;
                }
            }
            out[(size_t)k * nQ + i] = res;
        }
    }
    return ORC_OK;
}

/* ------------------------------------------------------------------------ */
/* optimal_interpolation_ensi (Points): src/api/oi_ensi.cpp:114-568           */
/* background [nY][nE], pbackground [nS][nE], out [nY][nE]                    */
/* ------------------------------------------------------------------------ */
/* gst == NULL: scalar BarnesStructure(h, v, w, min_rho); else any structure function (structure.cpp:287-944), optionally with
 * spatially varying scales: the structure as seen from the grid point (localization_distance(p1) :213, corr_background(p1, p2)
 * :250 take the parameters at the FIRST point), c_h / c_v / c_w / c_R per background point [nY] or NULL */
static int orc_oi_ensi_core(int y0, int y1, int nY, int nE,
                const float* gx, const float* gy, const float* gz, const float* gelev, const float* glaf,
                const float* background,
                int nS, const float* ox, const float* oy, const float* oz, const float* oelev, const float* olaf,
                const float* pobs, const float* psigmas, const float* pbackground,
                float h, float v, float w, float min_rho,
                int max_points, int allow_extrapolation, float* out,
                const orc_struct* gst, const float* c_h, const float* c_v, const float* c_w, const float* c_R) {
    if(max_points < 0) return ORC_EINVAL;
    for(size_t i = (size_t)y0 * nE; i < (size_t)y1 * nE; i++) out[i] = background[i];
    if(nS == 0) return ORC_OK;
    /* gY / gYhat: :166-178 */
    float* gYp = (float*)malloc(sizeof(float) * nS * nE);
    float* gYhat = (float*)malloc(sizeof(float) * nS);
    for(int i = 0; i < nS; i++) {
        float mean = orc_calc_statistic(pbackground + (size_t)i * nE, nE, ST_MEAN);
        for(int e = 0; e < nE; e++) {
            float value = pbackground[(size_t)i * nE + e];
            gYp[(size_t)i * nE + e] = (orc_valid(value) && orc_valid(mean)) ? value - mean : value;
        }
        gYhat[i] = mean;
    }
    /* validEns: :187-201 -- members valid over the WHOLE field */
    int* validEns = (int*)malloc(sizeof(int) * nE);
    int nV = 0;
    for(int e = 0; e < nE; e++) {
        int bad = 0;
        for(int y = 0; y < nY && !bad; y++) if(!orc_valid(background[(size_t)y * nE + e])) bad = 1;
        if(!bad) validEns[nV++] = e;
    }
    float loc = orc_barnes_localization_distance(h, min_rho);
    orc_pair* work = (orc_pair*)malloc(sizeof(orc_pair) * nS);
    int* sel = (int*)malloc(sizeof(int) * nS);
    float* srho = (float*)malloc(sizeof(float) * nS);
    double* Pinv = (double*)malloc(sizeof(double) * nV * nV);
    double* P = (double*)malloc(sizeof(double) * nV * nV);
    double* Aw = (double*)malloc(sizeof(double) * nV * nV);
    double* eval = (double*)malloc(sizeof(double) * nV);
    double* evec = (double*)malloc(sizeof(double) * nV * nV);
    double* W = (double*)malloc(sizeof(double) * nV * nV);
    double* wv = (double*)malloc(sizeof(double) * nV);
    double* X = (double*)malloc(sizeof(double) * nV);
    for(int y = y0; y < y1; y++) {
        int lS;
        if(!gst)
            lS = orc_select(gx[y], gy[y], gz[y], gelev[y], glaf[y], nS, ox, oy, oz, oelev, olaf,
                            pobs, NULL, h, v, w, loc, max_points, work, sel, srho);   /* :213-269 */
        else {
            orc_struct sc = *gst;
            if(c_h) { sc.h = c_h[y]; sc.v = c_v[y]; sc.w = c_w[y]; sc.loc = c_R[y]; }
            int n = 0;
            for(int sI = 0; sI < nS; sI++) {
                if(!orc_in_radius(gx[y], gy[y], gz[y], ox[sI], oy[sI], oz[sI], sc.loc, 1)) continue;               /* :213-233 */
                float rho = orc_corr_g(&sc, gx[y], gy[y], gz[y], gelev[y], glaf[y], ox[sI], oy[sI], oz[sI], oelev[sI], olaf[sI], 1);   /* :250 */
                if(!orc_valid(pobs[sI])) continue;                                                                   /* :235 */
                if(rho > 0) { work[n].rho = rho; work[n].idx = sI; n++; }
            }
            if(max_points > 0 && n > max_points) { qsort(work, n, sizeof(orc_pair), orc_pair_cmp); n = max_points; } /* :262-273 */
            for(int i = 0; i < n; i++) { sel[i] = work[i].idx; srho[i] = work[i].rho; }
            lS = n;
        }
        if(lS == 0) continue;
        if(nV == 0) continue;
        double* lY = (double*)malloc(sizeof(double) * lS * nV);     /* lS x nV, arma column-major: lY[e*lS+i] */
        double* C = (double*)malloc(sizeof(double) * nV * lS);
        double* Rinv = (double*)malloc(sizeof(double) * lS);
        double* dvec = (double*)malloc(sizeof(double) * lS);
        for(int i = 0; i < lS; i++) {                                /* :282-302 */
            int idx = sel[i];
            for(int e = 0; e < nV; e++) lY[(size_t)e * lS + i] = (double)gYp[(size_t)idx * nE + validEns[e]];
            float s2 = psigmas[idx] * psigmas[idx];                 /* float product (:300) */
            Rinv[i] = (double)srho[i] / (double)s2;
            dvec[i] = (double)pobs[idx] - (double)gYhat[idx];
        }
        for(int e = 0; e < nV; e++) for(int i = 0; i < lS; i++) C[(size_t)e * lS + i] = lY[(size_t)e * lS + i] * Rinv[i]; /* :380 */
        float diag = 1 / 1.0f * (nV - 1);                            /* :383, delta = 1 */
        for(int a = 0; a < nV; a++) for(int b = 0; b < nV; b++) {    /* :385 */
            double s = 0;
            for(int i = 0; i < lS; i++) s += C[(size_t)a * lS + i] * lY[(size_t)b * lS + i];
            Pinv[a * nV + b] = s + (a == b ? (double)diag : 0.0);
        }
        /* rcond <= 0 -> passthrough (:386-390): only for singular / NaN matrices */
        memcpy(P, Pinv, sizeof(double) * nV * nV);
        if(orc_inv(P, nV) != ORC_OK) { free(lY); free(C); free(Rinv); free(dvec); continue; } /* :398 */
        for(int i = 0; i < nV * nV; i++) Aw[i] = (nV - 1) * P[i];    /* :401 */
        orc_eig_sym(Aw, nV, eval, evec);
        for(int a = 0; a < nV; a++) for(int b = 0; b < nV; b++) {    /* :419-421 W = V sqrt(L) V' */
            double s = 0;
            for(int k = 0; k < nV; k++) s += evec[a * nV + k] * sqrt(eval[k]) * evec[b * nV + k];
            W[a * nV + b] = s;
        }
        for(int a = 0; a < nV; a++) {                                /* :427-437 w = P C (lObs - lYhat) */
            double s = 0;
            for(int i = 0; i < lS; i++) {
                double pc = 0;
                for(int b = 0; b < nV; b++) pc += P[a * nV + b] * C[(size_t)b * lS + i];
                s += pc * dvec[i];
            }
            wv[a] = s;
        }
        for(int a = 0; a < nV; a++) for(int b = 0; b < nV; b++) W[a * nV + b] += wv[a];   /* :440-444 */
        float total = 0; int count = 0;                              /* :447-462 */
        for(int e = 0; e < nV; e++) {
            float value = background[(size_t)y * nE + validEns[e]];
            if(orc_valid(value)) { X[e] = value; total += value; count++; }
        }
        float ensMean = total / count;
        for(int e = 0; e < nV; e++) X[e] -= ensMean;                 /* double - float */
        for(int e = 0; e < nV; e++) {                                /* :505-553 */
            float tot = 0;
            for(int k = 0; k < nV; k++) tot += X[k] * W[k * nV + e]; /* float += double product */
            float currIncrement = tot;
            if(!allow_extrapolation) {
                /* :523-524 uses lY[e] (LINEAR index into the lS x nV column-major matrix) */
                double lYe = lY[e];
                float maxInc = 0, minInc = 0;
                for(int i = 0; i < lS; i++) {
                    float dv = (float)((double)pobs[sel[i]] - (lYe + (double)gYhat[sel[i]]));
                    if(i == 0 || dv > maxInc) maxInc = dv;
                    if(i == 0 || dv < minInc) minInc = dv;
                }
                float memberIncrement = currIncrement - X[e];
                if(maxInc > 0 && memberIncrement > maxInc) currIncrement = maxInc + X[e];
                else if(maxInc < 0 && memberIncrement > 0) currIncrement = 0 + X[e];
                else if(minInc < 0 && memberIncrement < minInc) currIncrement = minInc + X[e];
                else if(minInc > 0 && memberIncrement < 0) currIncrement = 0 + X[e];
            }
            out[(size_t)y * nE + validEns[e]] = ensMean + currIncrement;
        }
        free(lY); free(C); free(Rinv); free(dvec);
    }
    free(gYp); free(gYhat); free(validEns); free(work); free(sel); free(srho);
    free(Pinv); free(P); free(Aw); free(eval); free(evec); free(W); free(wv); free(X);
    return ORC_OK;
}
int orc_oi_ensi_range(int y0, int y1, int nY, int nE,
                const float* gx, const float* gy, const float* gz, const float* gelev, const float* glaf,
                const float* background,
                int nS, const float* ox, const float* oy, const float* oz, const float* oelev, const float* olaf,
                const float* pobs, const float* psigmas, const float* pbackground,
                float h, float v, float w, float min_rho,
                int max_points, int allow_extrapolation, float* out) {
    return orc_oi_ensi_core(y0, y1, nY, nE, gx, gy, gz, gelev, glaf, background, nS, ox, oy, oz, oelev, olaf, pobs, psigmas, pbackground,
                            h, v, w, min_rho, max_points, allow_extrapolation, out, NULL, NULL, NULL, NULL, NULL);
}
/* optimal_interpolation_ensi with any structure function / spatially varying scales (see orc_oi_ensi_core) */
int orc_oi_ensi_generic(int nY, int nE,
                const float* gx, const float* gy, const float* gz, const float* gelev, const float* glaf,
                const float* background,
                int nS, const float* ox, const float* oy, const float* oz, const float* oelev, const float* olaf,
                const float* pobs, const float* psigmas, const float* pbackground,
                int kh, int kv, int kw, float h, float v, float w, float loc, int cv, float cv_dist,
                int max_points, int allow_extrapolation, float* out,
                const float* c_h, const float* c_v, const float* c_w, const float* c_R) {
    orc_struct st = {kh, kv, kw, h, v, w, loc, cv, cv_dist};
    return orc_oi_ensi_core(0, nY, nY, nE, gx, gy, gz, gelev, glaf, background, nS, ox, oy, oz, oelev, olaf, pobs, psigmas, pbackground,
                            h, v, w, 0.0f, max_points, allow_extrapolation, out, &st, c_h, c_v, c_w, c_R);
}
int orc_oi_ensi(int nY, int nE,
                const float* gx, const float* gy, const float* gz, const float* gelev, const float* glaf,
                const float* background,
                int nS, const float* ox, const float* oy, const float* oz, const float* oelev, const float* olaf,
                const float* pobs, const float* psigmas, const float* pbackground,
                float h, float v, float w, float min_rho,
                int max_points, int allow_extrapolation, float* out) {
    return orc_oi_ensi_range(0, nY, nY, nE, gx, gy, gz, gelev, glaf, background, nS, ox, oy, oz, oelev, olaf,
                             pobs, psigmas, pbackground, h, v, w, min_rho, max_points, allow_extrapolation, out);
}

/* ------------------------------------------------------------------------ */
/* optimal_interpolation_ensi_multi_{ebe, ebesc, utem} (Points overloads):    */
/* src/api/oi_ensi_multi.cpp:329-628, :630-860, :862-1311.                     */
/* variant 1 = ebe, 2 = ebesc, 3 = utem.  background / background_corr         */
/* [nY][nE]; pbackground / pbackground_corr [nS][nE]; pobs [nS][nE] (ebe,      */
/* ebesc) or [nS] (utem); bratios [nY]; pratios [nS]; out [nY][nE].            */
/* The reference indexes `lInnov(i, ei)` with the ORIGINAL member index into a */
/* matrix with nValidEns columns (:563-567, :770-773): with an invalid member  */
/* in front of a valid one that is out of bounds (Armadillo throws) ->         */
/* ORC_ESINGULAR here (RuntimeError in the mirrors).                           */
/* ------------------------------------------------------------------------ */
#define ORC_MIN_STD 0.0013f
int orc_oi_ensi_multi(int variant, int nY, int nE,
                const float* gx, const float* gy, const float* gz, const float* gelev, const float* glaf,
                const float* bratios, const float* background, const float* background_corr,
                int nS, const float* ox, const float* oy, const float* oz, const float* oelev, const float* olaf,
                const float* pobs, const float* pratios, const float* pbackground, const float* pbackground_corr,
                float h, float v, float w, float min_rho,
                int max_points, int allow_extrapolation, float* out) {
    if(max_points < 0) return ORC_EINVAL;
    for(size_t i = 0; i < (size_t)nY * nE; i++) out[i] = background[i];
    if(nS == 0) return ORC_OK;
    const int corr = variant != 2;
    int* validEns = (int*)malloc(sizeof(int) * (nE + 1));
    int nV = 0;
    for(int e = 0; e < nE; e++) {                                         /* :395-418 / :691-712 / :930-950 */
        int bad = 0;
        for(int y = 0; y < nY && !bad; y++) {
            if(!orc_valid(background[(size_t)y * nE + e])) bad = 1;
            if(corr && !orc_valid(background_corr[(size_t)y * nE + e])) bad = 1;
        }
        for(int i = 0; i < nS && !bad; i++) {
            if(!orc_valid(pbackground[(size_t)i * nE + e])) bad = 1;
            if(corr && !orc_valid(pbackground_corr[(size_t)i * nE + e])) bad = 1;
        }
        if(!bad) validEns[nV++] = e;
    }
    if(nV == 0) { free(validEns); return ORC_OK; }
    int oob = (variant != 3) && validEns[nV - 1] != nV - 1;
    /* per-observation ensemble quantities */
    float* gZ = (float*)calloc((size_t)nS * nV, sizeof(float));          /* ebe: gZ_R (:421-444); utem: gY_corr (:985-1002) */
    float* gY = (float*)calloc((size_t)nS * nV, sizeof(float));          /* utem: gY (:976-984) */
    float* gYhat = (float*)calloc(nS, sizeof(float));
    float* row = (float*)malloc(sizeof(float) * nV);
    float* obs0 = (float*)malloc(sizeof(float) * nS);                    /* the value whose validity selects an observation */
    const float const_fact = 1 / sqrt(nV - 1);                           /* :968 (float) */
    for(int i = 0; i < nS; i++) {
        obs0[i] = variant == 3 ? pobs[i] : pobs[(size_t)i * nE];
        if(variant == 3) {
            for(int e = 0; e < nV; e++) row[e] = pbackground[(size_t)i * nE + validEns[e]];
            float mean = orc_calc_statistic(row, nV, ST_MEAN);
            for(int e = 0; e < nV; e++) gY[(size_t)i * nV + e] = orc_valid(mean) ? row[e] - mean : 0;
            gYhat[i] = mean;
        }
        if(corr) {
            for(int e = 0; e < nV; e++) row[e] = pbackground_corr[(size_t)i * nE + validEns[e]];
            float mean = orc_calc_statistic(row, nV, ST_MEAN), std = orc_calc_statistic(row, nV, ST_STD);
            if(orc_valid(mean) && orc_valid(std) && std > ORC_MIN_STD) {
                for(int e = 0; e < nV; e++) {
                    if(variant == 1) gZ[(size_t)i * nV + e] = 1 / sqrt(nV - 1) * (row[e] - mean) / std;     /* double expression -> float */
                    else gZ[(size_t)i * nV + e] = const_fact * (row[e] - mean) / std;                         /* float expression */
                }
            }
        }
    }
    float loc = orc_barnes_localization_distance(h, min_rho);
    orc_pair* work = (orc_pair*)malloc(sizeof(orc_pair) * nS);
    int* sel = (int*)malloc(sizeof(int) * nS);
    float* srho = (float*)malloc(sizeof(float) * nS);
    int rc = ORC_OK;
    for(int y = 0; y < nY && rc == ORC_OK; y++) {
        float ratio = bratios[y];
        int lS = orc_select(gx[y], gy[y], gz[y], gelev[y], glaf[y], nS, ox, oy, oz, oelev, olaf,
                            obs0, NULL, h, v, w, loc, max_points, work, sel, srho);
        if(lS == 0) continue;
        if(oob) { rc = ORC_ESINGULAR; break; }
        if(variant != 3) {
            double* A = (double*)malloc(sizeof(double) * lS * lS);
            double* rl = (double*)malloc(sizeof(double) * lS);
            double* K = (double*)malloc(sizeof(double) * lS);
            double* xL = (double*)calloc(nV, sizeof(double));
            if(variant == 1) {                                                 /* :531-541 */
                for(int e = 0; e < nV; e++) row[e] = background_corr[(size_t)y * nE + validEns[e]];
                float mean = orc_calc_statistic(row, nV, ST_MEAN), std = orc_calc_statistic(row, nV, ST_STD);
                if(orc_valid(mean) && orc_valid(std) && std > ORC_MIN_STD)
                    for(int e = 0; e < nV; e++) xL[e] = 1 / sqrt(nV - 1) * (row[e] - mean) / std;
            }
            for(int i = 0; i < lS; i++) {
                int si = sel[i];
                double rz = 1.0;
                if(variant == 1) { rz = 0; for(int e = 0; e < nV; e++) rz += xL[e] * (double)gZ[(size_t)si * nV + e]; }
                rl[i] = (double)srho[i] * rz;                                  /* :581 / lCorr1D */
                for(int j = 0; j < lS; j++) {
                    int sj = sel[j];
                    float c = orc_barnes_corr(ox[si], oy[si], oz[si], oelev[si], olaf[si], ox[sj], oy[sj], oz[sj], oelev[sj], olaf[sj], h, v, w, loc);
                    double zz = 1.0;
                    if(variant == 1) { zz = 0; for(int e = 0; e < nV; e++) zz += (double)gZ[(size_t)si * nV + e] * (double)gZ[(size_t)sj * nV + e]; }
                    A[i * lS + j] = (double)c * zz;                            /* :584 */
                }
                A[i * lS + i] += (double)pratios[si];                          /* + lR_dd */
            }
            if(orc_inv(A, lS) != ORC_OK) { rc = ORC_ESINGULAR; free(A); free(rl); free(K); free(xL); break; }
            for(int j = 0; j < lS; j++) { double s = 0; for(int i = 0; i < lS; i++) s += rl[i] * A[i * lS + j]; K[j] = s; }   /* :586 */
            for(int e = 0; e < nV; e++) {
                double s = 0; float maxInc = 0, minInc = 0;
                for(int i = 0; i < lS; i++) {
                    float inn = pobs[(size_t)sel[i] * nE + validEns[e]] - pbackground[(size_t)sel[i] * nE + validEns[e]];   /* :565 */
                    s += K[i] * (double)inn;
                    if(i == 0 || inn > maxInc) maxInc = inn;
                    if(i == 0 || inn < minInc) minInc = inn;
                }
                double dx = ratio * s;                                         /* :588 */
                if(!allow_extrapolation) {                                     /* :593-615 */
                    float increment = dx;
                    if(maxInc > 0 && increment > maxInc) increment = maxInc;
                    else if(maxInc < 0 && increment > 0) increment = 0;
                    else if(minInc < 0 && increment < minInc) increment = minInc;
                    else if(minInc > 0 && increment < 0) increment = 0;
                    dx = increment;
                }
                out[(size_t)y * nE + validEns[e]] = background[(size_t)y * nE + validEns[e]] + dx;   /* :620 */
            }
            free(A); free(rl); free(K); free(xL);
        }
        else {
            double* lY = (double*)malloc(sizeof(double) * lS * nV);           /* column-major lS x nV */
            double* lYc = (double*)malloc(sizeof(double) * lS * nV);
            double* Rinv = (double*)malloc(sizeof(double) * lS);
            double* dvec = (double*)malloc(sizeof(double) * lS);
            double* Pinv = (double*)malloc(sizeof(double) * nV * nV);
            double* P = (double*)malloc(sizeof(double) * nV * nV);
            double* Aw = (double*)malloc(sizeof(double) * nV * nV);
            double* eval = (double*)malloc(sizeof(double) * nV);
            double* evec = (double*)malloc(sizeof(double) * nV * nV);
            double* W = (double*)malloc(sizeof(double) * nV * nV);
            double* wv = (double*)malloc(sizeof(double) * nV);
            double* X = (double*)malloc(sizeof(double) * nV);
            double* Xc = (double*)malloc(sizeof(double) * nV);
            for(int i = 0; i < lS; i++) {
                int idx = sel[i];
                for(int e = 0; e < nV; e++) { lY[(size_t)e * lS + i] = (double)gY[(size_t)idx * nV + e]; lYc[(size_t)e * lS + i] = (double)gZ[(size_t)idx * nV + e]; }
                Rinv[i] = (double)srho[i] / (double)pratios[idx];             /* :1078 */
                dvec[i] = (double)pobs[idx] - (double)gYhat[idx];
            }
            for(int a = 0; a < nV; a++) for(int b = 0; b < nV; b++) {          /* :1082-1085 */
                double s = 0;
                for(int i = 0; i < lS; i++) s += lYc[(size_t)a * lS + i] * Rinv[i] * lYc[(size_t)b * lS + i];
                Pinv[a * nV + b] = s + (a == b ? 1.0 : 0.0);
            }
            memcpy(P, Pinv, sizeof(double) * nV * nV);
            if(orc_inv(P, nV) == ORC_OK) {
                for(int i = 0; i < nV * nV; i++) Aw[i] = (nV - 1) * P[i];      /* :1098 */
                orc_eig_sym(Aw, nV, eval, evec);
                for(int a = 0; a < nV; a++) for(int b = 0; b < nV; b++) {
                    double s = 0;
                    for(int k = 0; k < nV; k++) s += evec[a * nV + k] * sqrt(eval[k]) * evec[b * nV + k];
                    W[a * nV + b] = s;
                }
                for(int a = 0; a < nV; a++) {                                   /* :1131-1139 */
                    double s = 0;
                    for(int i = 0; i < lS; i++) {
                        double pc = 0;
                        for(int b = 0; b < nV; b++) pc += P[a * nV + b] * lYc[(size_t)b * lS + i] * Rinv[i];
                        s += pc * dvec[i];
                    }
                    wv[a] = s;
                }
                float* vals = row;                                              /* :1141-1179 */
                float* valc = (float*)malloc(sizeof(float) * nV);
                float total = 0, totalc = 0;
                for(int e = 0; e < nV; e++) {
                    vals[e] = background[(size_t)y * nE + validEns[e]]; valc[e] = background_corr[(size_t)y * nE + validEns[e]];
                    total += vals[e]; totalc += valc[e];
                }
                float ensMean = total / nV, ensStd = orc_calc_statistic(vals, nV, ST_STD);
                float ensMeanC = totalc / nV, ensStdC = orc_calc_statistic(valc, nV, ST_STD);
                for(int e = 0; e < nV; e++) {
                    X[e] = (double)vals[e] - ensMean;
                    float value_corr = (float)(double)valc[e];                  /* X_corr(e) holds the float in a double */
                    Xc[e] = (ensStdC <= ORC_MIN_STD) ? 0 : const_fact * (value_corr - ensMeanC) / ensStdC;
                }
                for(int a = 0; a < nV; a++) for(int b = 0; b < nV; b++) W[a * nV + b] = ensStd * W[a * nV + b] + ratio * wv[a];   /* :1181-1185 */
                for(int e = 0; e < nV; e++) {
                    float tot = 0;
                    for(int k = 0; k < nV; k++) tot += Xc[k] * W[k * nV + e];   /* :1229-1233 */
                    float currIncrement = tot;
                    if(!allow_extrapolation) {
                        double lYe = lY[e];                                     /* LINEAR index, :1240 */
                        float maxInc = 0, minInc = 0;
                        for(int i = 0; i < lS; i++) {
                            float dv = (float)((double)pobs[sel[i]] - (lYe + (double)gYhat[sel[i]]));
                            if(i == 0 || dv > maxInc) maxInc = dv;
                            if(i == 0 || dv < minInc) minInc = dv;
                        }
                        float memberIncrement = currIncrement - X[e];
                        if(maxInc > 0 && memberIncrement > maxInc) currIncrement = maxInc + X[e];
                        else if(maxInc < 0 && memberIncrement > 0) currIncrement = 0 + X[e];
                        else if(minInc < 0 && memberIncrement < minInc) currIncrement = minInc + X[e];
                        else if(minInc > 0 && memberIncrement < 0) currIncrement = 0 + X[e];
                    }
                    out[(size_t)y * nE + validEns[e]] = ensMean + currIncrement;
                }
                free(valc);
            }
            free(lY); free(lYc); free(Rinv); free(dvec); free(Pinv); free(P); free(Aw); free(eval); free(evec); free(W); free(wv); free(X); free(Xc);
        }
    }
    free(validEns); free(gZ); free(gY); free(gYhat); free(row); free(obs0); free(work); free(sel); free(srho);
    return rc;
}
