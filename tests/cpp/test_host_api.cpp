// C++ caller of the host mirror (gridpp_amd/host/gridpp.hpp): the reference's own known-answer cases, written the way
// a gridpp.h user writes them (tests/test_optimal_interpolation.py:50-63, tests/test_neighbourhood.py:75-88,
// tests/test_barnes_structure.py, tests/test_kdtree.py:149-161).  Built and run by tests/test_gpu_cpp_host.py.
#include <cstdio>
#include <cstdlib>
#include "gridpp.hpp"

#define CHECK(cond) do { if(!(cond)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); return 1; } } while(0)

int main() {
    using namespace gridpp;
    // OI analytic case
    vec2 y = {{0, 0, 0}}, x = {{0, 2500, 10000}};
    Grid grid(y, x, y, y, Cartesian);
    Points points(vec{0}, vec{2500}, vec{0}, vec{0}, Cartesian);
    BarnesStructure structure(2500);
    vec2 background = {{0, 0, 0}};
    vec2 out = optimal_interpolation(grid, background, points, vec{1}, vec{0.1f}, vec{0}, structure, 10);
    CHECK(std::fabs(out[0][0] - std::exp(-0.5) / 1.1) < 1e-6);
    CHECK(std::fabs(out[0][1] - 1 / 1.1) < 1e-6);
    CHECK(std::fabs(out[0][2] - std::exp(-4.5) / 1.1) < 1e-6);
    vec2 variance;
    vec2 bvar = {{1, 1, 1}};
    optimal_interpolation_full(grid, background, bvar, points, vec{1}, vec{0.1f}, vec{0}, vec{1}, structure, 10, variance);
    CHECK(std::fabs(variance[0][1] - 0.1 / 1.1) < 1e-6);
    // invalid arguments -> std::invalid_argument
    bool threw = false;
    try { optimal_interpolation(grid, background, points, vec{1}, vec{0.1f}, vec{0}, structure, -1); } catch(const std::invalid_argument&) { threw = true; }
    CHECK(threw);
    threw = false;
    try { Points bad(vec{91}, vec{0}); } catch(const std::invalid_argument&) { threw = true; }
    CHECK(threw);
    // radius edge
    Points p3(vec{0, 1000, 2000}, vec{0, 0, 0}, vec{0, 0, 0}, vec{0, 0, 0}, Cartesian);
    CHECK(p3.get_neighbours(0, 0, 1000) == ivec{0});
    CHECK((p3.get_neighbours(0, 0, 1001) == ivec{0, 1}));
    CHECK(p3.get_nearest_neighbour(900, 0) == 1);
    // neighbourhood 5x5 with two NaNs
    vec2 values(5, vec(5));
    for(int i = 0; i < 5; i++) for(int j = 0; j < 5; j++) values[i][j] = i * 5 + j;
    values[1][3] = NAN; values[2][4] = NAN;
    vec2 m = neighbourhood(values, 1, Mean);
    CHECK(m[2][2] == 12.5f);
    CHECK(std::fabs(m[0][4] - 5.3333f) < 1e-4);
    CHECK(neighbourhood(values, 1, Max)[2][2] == 18);
    CHECK(neighbourhood_quantile(values, 0.5f, 1)[2][3] == 13);
    vec thr = get_neighbourhood_thresholds(values, 100);
    CHECK(neighbourhood_quantile_fast(values, 0.5f, 1, thr)[2][2] == 12);
    {   // deprecated aliases (include/gridpp.h:710-716): the ensemble forms of the three calls above
        vec3 ens(5, vec2(5, vec(1)));
        for(int i = 0; i < 5; i++) for(int j = 0; j < 5; j++) ens[i][j][0] = values[i][j];
        CHECK(neighbourhood_ens(ens, 1, Mean)[2][2] == 12.5f);
        CHECK(neighbourhood_quantile_ens(ens, 0.5f, 1)[2][3] == 13);
        CHECK(neighbourhood_quantile_ens_fast(ens, 0.5f, 1, thr)[2][2] == 12);
    }
    CHECK(calc_statistic(vec{0, 1, NAN}, Mean) == 0.5f);
    CHECK(calc_quantile(vec{0, NAN, 2}, 0.5f) == 1);
    // EnSI pass-through (tests/test_optimal_interpolation_ens.py:9-35)
    Points g1(vec{0}, vec{0});
    vec2 bg3 = {{0, 0, 0}};
    Points p2(vec{0, 0.1f}, vec{0, 0.1f});
    vec2 o3 = optimal_interpolation_ensi(g1, bg3, p2, vec{NAN, 0}, vec{1, 1}, vec2{{0, 0, 0}, {0, 0, 0}}, BarnesStructure(500000), 10);
    CHECK(o3[0][0] == 0 && o3[0][2] == 0);
    // nearest with time levels and the grid / points combinations (tests/test_nearest.py:58-104)
    {
        vec2 la1 = {{30, 30, 30}, {40, 40, 40}, {50, 50, 50}}, lo1 = {{0, 10, 20}, {0, 10, 20}, {0, 10, 20}};
        vec2 la2 = {{30, 30}, {50, 50}}, lo2 = {{0, 20}, {0, 20}};
        Grid g1(la1, lo1), g2(la2, lo2);
        vec2 v = {{0, 1, 2}, {3, 4, 5}, {6, 7, 8}};
        vec2 n2 = nearest(g1, g2, v);
        CHECK(n2[0][0] == 0 && n2[0][1] == 2 && n2[1][0] == 6 && n2[1][1] == 8);
        vec3 n3 = nearest(g1, g2, vec3{v, v});
        CHECK(n3.size() == 2 && n3[1][1][1] == 8);
        Points ip(vec{0, 5, 10}, vec{0, 5, 10}), op(vec{-1, 6}, vec{-1, 6});
        vec2 pp = nearest(ip, op, vec2{{0, 1, 2}, {9, 6, 1}});
        CHECK(pp[0][0] == 0 && pp[0][1] == 1 && pp[1][0] == 9 && pp[1][1] == 6);
        CHECK(nearest(ip, g2, vec{0, 1, 2}).size() == 2);
        CHECK(nearest(ip, g2, vec2{{0, 1, 2}}).size() == 1);
        CHECK(nearest(g1, op, vec3{v, v})[1].size() == 2);
    }
    // count / gridding (tests/test_count.py:28-40, tests/test_gridding.py:25-50)
    {
        vec2 gy = {{0, 1}, {0, 1}, {0, 1}}, gx = {{0, 0}, {0.5f, 0.5f}, {1, 1}}, z2 = {{0, 0}, {0, 0}, {0, 0}};
        Grid gg(gy, gx, z2, z2, Cartesian);
        Points gp(vec{-0.2f, 0.5f, 1}, vec{-0.2f, 0.5f, 1}, vec{0, 0, 0}, vec{0, 0, 0}, Cartesian);
        vec2 gs = gridding(gg, gp, vec{1, 2, 3}, 0.6f, 0, Sum);
        CHECK(gs[0][0] == 1 && std::isnan(gs[0][1]) && gs[1][0] == 2 && gs[1][1] == 5 && gs[2][1] == 3);
        vec2 gc = gridding(gg, gp, vec{1, 2, 3}, 0.6f, 0, Count);
        CHECK(gc[0][1] == 0 && gc[1][1] == 2);
        vec2 cn = count(gp, gg, 0.6f);
        CHECK(cn[1][1] == 2 && cn[0][1] == 0);
        vec2 dd = distance(gp, gg, 1);          // tests/test_distance.py: planar distance to the nearest input point
        CHECK(dd.size() == 3 && dd[2][1] == 0 && std::fabs(dd[1][1] - 0.5f) < 1e-6 && std::fabs(dd[0][0] - std::sqrt(0.08f)) < 1e-5);
        vec2 gn = gridding_nearest(gg, gp, vec{1, 2, 3}, 1, Sum);
        CHECK(gn.size() == 3 && gn[0].size() == 2);
        threw = false;
        try { gridding(gg, gp, vec{1}, 0.6f, 0, Sum); } catch(const std::invalid_argument&) { threw = true; }
        CHECK(threw);
    }
    // fill / neighbourhood_search / calc_gradient (tests/test_fill_missing.py:8-17, tests/test_neighbourhood_search.py:52-54,
    // tests/test_calc_gradient.py:17-26)
    {
        vec2 ns = neighbourhood_search(vec2{{0, 1, 2}}, vec2{{0.5f, 0.5f, 1}}, 1, 0.7f, 1, 0.1f);
        CHECK(ns[0][0] == 0 && ns[0][1] == 2 && ns[0][2] == 2);
        vec2 gr = calc_gradient(vec2{{0, 1, 2}}, vec2{{0, 1, 2}}, LinearRegression, 5, 0, 0, -11);
        CHECK(std::fabs(gr[0][0] - 1) < 1e-6 && std::fabs(gr[0][2] - 1) < 1e-6);
        vec2 fm = fill_missing(vec2{{0, 1, 2}, {3, NAN, 5}, {6, 7, 8}});
        CHECK(fm[1][1] == 4);
    }
    // bilinear (tests/test_bilinear.py:134-156, tests/test_grid.py:23-30)
    {
        vec2 la1 = {{0, 0}, {1, 1}}, lo1 = {{0, 1}, {0, 1}};
        vec2 la2 = {{0, 0, 0}, {0.5f, 0.5f, 0.5f}, {1, 1, 1}}, lo2 = {{0, 0.5f, 1}, {0, 0.5f, 1}, {0, 0.5f, 1}};
        Grid g1(la1, lo1), g2(la2, lo2);
        vec2 v = {{0, 1}, {2, 3}};
        vec2 b = bilinear(g1, g2, v);
        CHECK(b[0][1] == 0.5f && b[1][1] == 1.5f && b[2][2] == 3 && b[1][0] == 1);
        vec3 b3 = bilinear(g1, g2, vec3{v, v, v});
        CHECK(b3.size() == 3 && b3[2][1][2] == 2);
        vec bp = bilinear(g1, Points(vec{0.5f}, vec{0.25f}), v);
        CHECK(bp.size() == 1 && bp[0] == 1.25f);
        Grid g3(vec2{{0, 0, 0}, {1, 1, 1}}, vec2{{0, 1, 2}, {0.25f, 1.25f, 2.25f}});
        int Y1, X1, Y2, X2;
        CHECK(g3.get_box(0.4f, 1.25f, Y1, X1, Y2, X2) && Y1 == 0 && X1 == 1 && Y2 == 1 && X2 == 2);
        threw = false;
        try { bilinear(g1, g2, vec2{{0, 1, 2}}); } catch(const std::invalid_argument&) { threw = true; }
        CHECK(threw);
    }
    // optimal_interpolation_ensi_multi_* (include/gridpp.h:311-441): one observation on a grid point, unit ratios, no ensemble spread in
    // the observation term -> ebesc moves every member half way (K = rho / (1 + pratio) = 1 / 2 at the observed point); the Grid and
    // Points overloads agree
    {
        vec2 la = {{0, 0}, {0.01f, 0.01f}}, lo = {{0, 0.01f}, {0, 0.01f}};
        Grid g(la, lo);
        Points gp(vec{0, 0, 0.01f, 0.01f}, vec{0, 0.01f, 0, 0.01f});
        Points ob(vec{0}, vec{0});
        vec3 bg3 = {{{1, 2, 3}, {1, 2, 3}}, {{1, 2, 3}, {1, 2, 3}}};
        vec2 bg2 = {{1, 2, 3}, {1, 2, 3}, {1, 2, 3}, {1, 2, 3}};
        vec2 pobs = {{3, 4, 5}}, pbg = {{1, 2, 3}};
        BarnesStructure st(10000);
        vec3 a3 = optimal_interpolation_ensi_multi_ebesc(g, vec2{{1, 1}, {1, 1}}, bg3, ob, pobs, vec{1}, pbg, st, 5);
        vec2 a2 = optimal_interpolation_ensi_multi_ebesc(gp, vec{1, 1, 1, 1}, bg2, ob, pobs, vec{1}, pbg, st, 5);
        CHECK(std::fabs(a3[0][0][0] - 2) < 1e-6 && std::fabs(a3[0][0][2] - 4) < 1e-6);
        CHECK(a3[1][1][1] == a2[3][1] && a3[0][1][0] == a2[1][0] && a3[1][1][1] > 2 && a3[1][1][1] < 3);
        vec3 u3 = optimal_interpolation_ensi_multi_utem(g, vec2{{1, 1}, {1, 1}}, bg3, bg3, ob, vec{3}, vec{1}, pbg, pbg, st, 5);
        CHECK(u3.size() == 2 && u3[0][0].size() == 3 && std::isfinite(u3[0][0][0]));
        vec3 e3 = optimal_interpolation_ensi_multi_ebe(g, vec2{{1, 1}, {1, 1}}, bg3, bg3, ob, pobs, vec{1}, pbg, pbg, st, 5);
        // perfectly correlated 3-member ensembles: Z Z^T = n / (n - 1) = 1.5 (population std, 1 / sqrt(n - 1) factor), K = 1.5 / 2.5
        CHECK(std::fabs(e3[0][0][0] - 2.2f) < 1e-5);
        threw = false;
        try { optimal_interpolation_ensi_multi_ebesc(gp, vec{1, 1, 1}, bg2, ob, pobs, vec{1}, pbg, st, 5); } catch(const std::invalid_argument&) { threw = true; }
        CHECK(threw);
    }
    // multi-GPU helpers through RCCL with a world of one: the same code path as N ranks (ncclCommInitRank, ncclBroadcast on the library
    // stream), checked here as far as one GPU allows; row tiles partition the rows
    {
        int r0, r1, covered = 0;
        for(int r = 0; r < 3; r++) { multi::row_tile(10, r, 3, r0, r1); CHECK(r0 == covered); covered = r1; }
        CHECK(covered == 10);
        multi::init(0, 1, multi::unique_id());
        vec2 la = {{0, 0}, {0.01f, 0.01f}}, lo = {{0, 0.01f}, {0, 0.01f}};
        Grid g(la, lo);
        Points ob(vec{0}, vec{0});
        vec pobs = {1}, prat = {1}, pbg = {0};
        BarnesStructure st(10000);
        vec2 t = multi::optimal_interpolation(g, vec2{{0, 0}, {0, 0}}, ob, pobs, prat, pbg, st, 5);
        vec2 u = optimal_interpolation(g, vec2{{0, 0}, {0, 0}}, ob, pobs, prat, pbg, st, 5);
        CHECK(t[0][0] == u[0][0] && t[1][1] == u[1][1] && std::fabs(t[0][0] - 0.5f) < 1e-6);
        multi::destroy();
    }
    // debug level, messages, and the EnSI pass-through warning (include/gridpp.h:1394-1430, src/api/oi_ensi.cpp:557-561)
    {
        CHECK(get_debug_level() == 0);
        set_debug_level(2);
        CHECK(get_debug_level() == 2);
        set_debug_level(0);
        bool thrown = false;
        try { error("expected"); } catch(const std::runtime_error& e) { thrown = std::string(e.what()) == "expected"; }
        CHECK(thrown);
        // one valid member (the others are NaN somewhere): Pinv is the zero matrix, every grid point with an observation in range
        // keeps its background values and is counted
        vec2 la = {{0, 0}, {0.01f, 0.01f}}, lo = {{0, 0.01f}, {0, 0.01f}};
        Grid g(la, lo);
        Points ob(vec{0.005f}, vec{0.005f});
        const float nanv = std::nanf("");
        vec3 bg3 = {{{1, 2, nanv}, {3, 4, 5}}, {{6, nanv, 8}, {9, 10, 11}}};
        vec2 pbg = {{0.5f, 0.6f, 0.7f}};
        vec3 o3 = optimal_interpolation_ensi(g, bg3, ob, vec{1}, vec{1}, pbg, BarnesStructure(10000), 5);
        gpp_ensi_stats est;
        CHECK(gpp_ensi_last_stats(&est) == GPP_OK);
        CHECK(est.cells == 4 && est.condition_passthrough == 4 && est.real_part_passthrough == 0);
        CHECK(o3[0][1][0] == 3 && o3[1][1][2] == 11);
    }
    // get_statistic (include/gridpp.h:1410, src/api/gridpp.cpp:11-43)
    if(get_statistic("mean") != Mean || get_statistic("randomchoice") != RandomChoice || get_statistic("quantile") != Quantile ||
       get_statistic("variance") != Unknown || get_statistic("x") != Unknown) { std::printf("FAIL get_statistic\n"); return 1; }
    std::printf("gridpp.hpp host API: all checks passed (version %s)\n", version().c_str());
    return 0;
}
